// FP8 operand preparation for the fp8 GEMM path (BASELINE.json configs[3]): per-tensor amax and
// scaled conversion to OCP e4m3 / e5m2 (gfx950 encodings), with an optional transposed copy (the
// weight-gradient and input-gradient GEMMs want their operands k-major, i.e. transposed) and an
// optional column sum (bias gradient).  The reference has no counterpart: it is fp32 only.
#include "common.h"

namespace {

constexpr float E4M3_MAX = 448.0f;
constexpr float E5M2_MAX = 57344.0f;

template <typename T>
__global__ __launch_bounds__(256) void amax_kernel(const T* __restrict__ src, long rows, long cols, long ld,
                                                   float* __restrict__ amax, int vec) {
  float m = 0.f;
  const long total = rows * cols;
  if (vec) {
    // contiguous matrix, 16-byte aligned, element count a multiple of the vector width: flat 16-byte loads
    constexpr int VW = 16 / (int)sizeof(T);
    const long nv = total / VW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
      float v[VW];
      const uint4 u = reinterpret_cast<const uint4*>(src)[i];
      if constexpr (sizeof(T) == 4) {
        v[0] = __uint_as_float(u.x); v[1] = __uint_as_float(u.y); v[2] = __uint_as_float(u.z); v[3] = __uint_as_float(u.w);
      } else {
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(w[j] << 16); v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
      }
#pragma unroll
      for (int j = 0; j < VW; ++j) m = fmaxf(m, fabsf(v[j]));
    }
  } else {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
      const long r = i / cols, c = i - r * cols;
      m = fmaxf(m, fabsf(to_f32(src[r * ld + c])));
    }
  }
  m = wave_max(m);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    // non-negative floats order like their bit patterns
    atomicMax(reinterpret_cast<unsigned int*>(amax), __float_as_uint(m));
  }
}

// two floats -> two fp8 bytes in the low half of the result (round to nearest even)
template <int FMT>
__device__ __forceinline__ unsigned cvt2(float a, float b) {
  if (FMT == FAVIT_E4M3) return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffffu;
  return (unsigned)__builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false) & 0xffffu;
}

// One workgroup converts a 64 x 64 tile.  Thread t owns row (t >> 2), columns 16*(t & 3) .. +15:
// 16 fp8 bytes = one 16-byte store of dst; the transposed copy goes through a padded LDS tile.
template <typename T, int FMT>
__global__ __launch_bounds__(256) void quantize_kernel(const T* __restrict__ src, long rows, long cols, long ld,
                                                       uint8_t* __restrict__ dst, long ld_dst,
                                                       uint8_t* __restrict__ dst_t, long ld_t,
                                                       const float* __restrict__ amax, float* __restrict__ scale_inv,
                                                       float* __restrict__ colsum, float* __restrict__ amax_next,
                                                       float* __restrict__ amax_clear, int amax_n) {
  __shared__ uint8_t tile[64][64 + 4];
  __shared__ float csum[4][64];
  __shared__ float wmax[4];
  constexpr float FMAX = FMT == FAVIT_E4M3 ? E4M3_MAX : E5M2_MAX;
  // amax = the maximum over amax_n (1, or FAVIT_FP8_AMAX_SLOTS with delayed scaling) partial maxima: the previous
  // call spread its atomics over that many addresses (tens of thousands of workgroups on ONE address serialise in
  // one L2 channel: measured 20 -> 83 us for this kernel); they were final at its end, so plain (L1-cached) loads do
  float am = 0.f;
  for (int i = threadIdx.x & 63; i < amax_n; i += 64) am = fmaxf(am, amax[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
  const float scale = am > 0.f ? __fdiv_rn(FMAX, am) : 1.0f;           // IEEE division: reproducible scales
  if (blockIdx.x == 0 && blockIdx.y == 0) {
    if (threadIdx.x == 0) scale_inv[0] = am > 0.f ? __fdiv_rn(am, FMAX) : 1.0f;
    // delayed scaling: clear the slots the call after next accumulates into
    if (amax_clear && threadIdx.x < FAVIT_FP8_AMAX_SLOTS) amax_clear[threadIdx.x] = 0.f;
  }
  const long r0 = (long)blockIdx.y * 64, c0 = (long)blockIdx.x * 64;
  const int tr = threadIdx.x >> 2, tc = (threadIdx.x & 3) * 16;
  const long r = r0 + tr;
  float v[16];
  {
    const T* rowp = src + r * ld + c0 + tc;
    if (r < rows && c0 + tc + 16 <= cols && ((reinterpret_cast<uintptr_t>(rowp) & 15) == 0)) {
      constexpr int NV = 16 * (int)sizeof(T) / 16;        // 16-byte loads for 16 elements
      unsigned w[4 * NV];
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const uint4 u = reinterpret_cast<const uint4*>(rowp)[q];
        w[4 * q] = u.x; w[4 * q + 1] = u.y; w[4 * q + 2] = u.z; w[4 * q + 3] = u.w;
      }
      if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __uint_as_float(w[j]);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[2 * j] = __uint_as_float(w[j] << 16); v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const long c = c0 + tc + j;
        v[j] = (r < rows && c < cols) ? to_f32(src[r * ld + c]) : 0.f;
      }
    }
  }
  if (amax_next) {
    // delayed scaling: this tensor's amax for the NEXT call of the same site, taken in the pass that quantises it
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) m = fmaxf(m, fabsf(v[j]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  }
  if (colsum) {
    // column sums of the UNQUANTISED values: 16 rows per wave reduced by DPP-free shuffles, then LDS
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float s = v[j];
      s += __shfl_xor(s, 4);
      s += __shfl_xor(s, 8);
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if ((threadIdx.x & 63) < 4) csum[threadIdx.x >> 6][tc + j] = s;
    }
  }
  unsigned w[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float f[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = fminf(fmaxf(v[4 * q + j] * scale, -FMAX), FMAX);
    w[q] = cvt2<FMT>(f[0], f[1]) | (cvt2<FMT>(f[2], f[3]) << 16);
  }
  if (dst && r < rows) {
    uint8_t* o = dst + r * ld_dst + c0 + tc;
    if (c0 + tc + 16 <= cols && ((reinterpret_cast<uintptr_t>(o) & 15) == 0)) {
      *reinterpret_cast<uint4*>(o) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (c0 + tc + j < cols) o[j] = (uint8_t)(w[j >> 2] >> (8 * (j & 3)));
    }
  }
  if (dst_t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<unsigned*>(&tile[tr][tc + 4 * q]) = w[q];
  }
  __syncthreads();
  if (amax_next && threadIdx.x == 0) {
    const float m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    const unsigned slot = (blockIdx.y * gridDim.x + blockIdx.x) & (FAVIT_FP8_AMAX_SLOTS - 1);
    if (m > 0.f) atomicMax(reinterpret_cast<unsigned int*>(amax_next) + slot, __float_as_uint(m));
  }
  if (colsum && threadIdx.x < 64) {
    const long c = c0 + threadIdx.x;
    if (c < cols) atomicAdd(colsum + c, csum[0][threadIdx.x] + csum[1][threadIdx.x] + csum[2][threadIdx.x] + csum[3][threadIdx.x]);
  }
  if (dst_t) {
    // thread t writes 16 consecutive source rows of source column (t >> 2): one 16-byte store of dst_t.
    // Source rows past `rows` hold zeros (the loads above were predicated), which zero-fills the pad.
    const int oc = threadIdx.x >> 2, orow = (threadIdx.x & 3) * 16;
    const long c = c0 + oc;
    if (c < cols) {
      unsigned o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        o[q] = (unsigned)tile[orow + 4 * q][oc] | ((unsigned)tile[orow + 4 * q + 1][oc] << 8) |
               ((unsigned)tile[orow + 4 * q + 2][oc] << 16) | ((unsigned)tile[orow + 4 * q + 3][oc] << 24);
      uint8_t* p = dst_t + c * ld_t + r0 + orow;
      if (r0 + orow + 16 <= ld_t && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
        *reinterpret_cast<uint4*>(p) = make_uint4(o[0], o[1], o[2], o[3]);
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (r0 + orow + j < ld_t) p[j] = (uint8_t)(o[j >> 2] >> (8 * (j & 3)));
      }
    }
  }
}

}  // namespace

extern "C" int favit_fp8_amax(const void* src, int src_dtype, int64_t rows, int64_t cols, int64_t ld_src, float* amax,
                              void* stream) {
  if (!src || !amax || rows <= 0 || cols <= 0 || ld_src < cols) return FAVIT_ERR_INVALID;
  hipStream_t st = as_stream(stream);
  const long total = rows * cols;
  long nb = (total + 256 * 16 - 1) / (256 * 16);
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  const int esz = src_dtype == FAVIT_F32 ? 4 : 2;
  const int vec = (ld_src == cols) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && ((total * esz) % 16 == 0);
  if (src_dtype == FAVIT_F32)
    hipLaunchKernelGGL(amax_kernel<float>, dim3((unsigned)nb), dim3(256), 0, st, (const float*)src, rows, cols, ld_src, amax, vec);
  else if (src_dtype == FAVIT_BF16)
    hipLaunchKernelGGL(amax_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, st, (const bf16_t*)src, rows, cols, ld_src, amax, vec);
  else
    return FAVIT_ERR_INVALID;
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

extern "C" int favit_fp8_quantize(const void* src, int src_dtype, int64_t rows, int64_t cols, int64_t ld_src, void* dst,
                                  int64_t ld_dst, void* dst_t, int64_t ld_t, int fmt, const float* amax,
                                  float* scale_inv, float* colsum, float* amax_next, float* amax_clear, void* stream) {
  // delayed scaling (amax_next given): amax, amax_next and amax_clear are arrays of FAVIT_FP8_AMAX_SLOTS floats
  const int amax_n = amax_next ? FAVIT_FP8_AMAX_SLOTS : 1;
  if ((amax_next == nullptr) != (amax_clear == nullptr)) return FAVIT_ERR_INVALID;
  if (!src || !amax || !scale_inv || rows <= 0 || cols <= 0 || ld_src < cols) return FAVIT_ERR_INVALID;
  if (!dst && !dst_t) return FAVIT_ERR_INVALID;
  if (dst && ld_dst < cols) return FAVIT_ERR_INVALID;
  if (dst_t && ld_t < rows) return FAVIT_ERR_INVALID;
  if (fmt != FAVIT_E4M3 && fmt != FAVIT_E5M2) return FAVIT_ERR_INVALID;
  hipStream_t st = as_stream(stream);
  // the grid also covers the zero pad of the transposed copy (source rows rows..ld_t-1)
  const long rcover = dst_t ? (ld_t > rows ? ld_t : rows) : rows;
  dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rcover + 63) / 64));
#define FAVIT_Q(T, F)                                                                                              \
  hipLaunchKernelGGL((quantize_kernel<T, F>), grid, dim3(256), 0, st, (const T*)src, (long)rows, (long)cols,        \
                     (long)ld_src, (uint8_t*)dst, (long)ld_dst, (uint8_t*)dst_t, (long)ld_t, amax, scale_inv, colsum, \
                     amax_next, amax_clear, amax_n)
  if (src_dtype == FAVIT_F32) {
    if (fmt == FAVIT_E4M3) FAVIT_Q(float, FAVIT_E4M3); else FAVIT_Q(float, FAVIT_E5M2);
  } else if (src_dtype == FAVIT_BF16) {
    if (fmt == FAVIT_E4M3) FAVIT_Q(bf16_t, FAVIT_E4M3); else FAVIT_Q(bf16_t, FAVIT_E5M2);
  } else {
    return FAVIT_ERR_INVALID;
  }
#undef FAVIT_Q
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}
