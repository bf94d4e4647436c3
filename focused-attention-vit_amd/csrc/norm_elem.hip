// HBM-bound row / elementwise kernels of the encoder path (gfx950):
// LayerNorm fwd/bwd, row reduction, cast, dropout, patchify, CLS/pos prologue,
// cross-entropy and AdamW.  All are one-pass, 16-B-per-lane coalesced; LayerNorm keeps
// the row in registers (one 64-lane wave per token row, shuffle reductions).
#include "common.h"

extern "C" unsigned* favit_health_ptr_(void);

namespace {

// ---------------------------------------------------------------------------------
// LayerNorm: one wave per row, NV float4 chunks per lane (D <= 256*NV)
// ---------------------------------------------------------------------------------
template <typename OutT>
__device__ __forceinline__ void store4(OutT* p, float a, float b, float c, float d) {
  if constexpr (sizeof(OutT) == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
  } else {
    bf16x4 o = {(bf16_t)a, (bf16_t)b, (bf16_t)c, (bf16_t)d};
    *reinterpret_cast<bf16x4*>(p) = o;
  }
}
template <typename InT>
__device__ __forceinline__ float4 load4(const InT* p) {
  if constexpr (sizeof(InT) == 4) {
    return *reinterpret_cast<const float4*>(p);
  } else {
    const bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
    return make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
  }
}

// fp8 mode (round 4): the bf16 tensor a LayerNorm pass produces -- the normalised rows that feed the qkv / fc1 GEMMs,
// the low-precision copy of the stream gradient that feeds the fc2 / proj input-gradient GEMMs -- leaves the pass
// QUANTISED as well, with the consumer site's delayed scale, instead of being re-read by a stand-alone quantising pass
// (favit_fp8_quantize: 18.9 us per 768-wide tensor at cfg4, 48 of them per step).  Same protocol and the same
// arithmetic as that kernel (fp8.hip): the scale is FMAX / max(amax[0..255]) of the site's previous call, the values
// are the ROUNDED bf16 ones, their maximum goes to amax_next (atomics spread over the slots), workgroup 0 writes
// scale_inv and clears amax_clear -- so the bytes, the scale and the history are bit-identical to the two-pass form.
struct LnQ8 {
  uint8_t* q;                 // [rows, D] fp8 (NULL: off)
  const float* amax;          // FAVIT_FP8_AMAX_SLOTS partial maxima measured by the previous call
  float* scale_inv;
  float* amax_next;
  float* amax_clear;
  int fmt;                    // FAVIT_E4M3 / FAVIT_E5M2
};

__device__ __forceinline__ float lnq8_begin(const LnQ8& q8, int lane) {
  const float fmax = q8.fmt == FAVIT_E4M3 ? 448.0f : 57344.0f;
  float am = 0.f;
  for (int i = lane; i < FAVIT_FP8_AMAX_SLOTS; i += 64) am = fmaxf(am, q8.amax[i]);
  am = wave_max(am);
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) q8.scale_inv[0] = am > 0.f ? __fdiv_rn(am, fmax) : 1.0f;
    if (threadIdx.x < FAVIT_FP8_AMAX_SLOTS) q8.amax_clear[threadIdx.x] = 0.f;
  }
  return am > 0.f ? __fdiv_rn(fmax, am) : 1.0f;
}
// four values as the bf16 copy holds them -> four fp8 bytes; m tracks max |value|
__device__ __forceinline__ unsigned lnq8_pack(float a, float b, float c, float d, float scale, int fmt, float& m) {
  const float fmax = fmt == FAVIT_E4M3 ? 448.0f : 57344.0f;
  a = (float)(bf16_t)a; b = (float)(bf16_t)b; c = (float)(bf16_t)c; d = (float)(bf16_t)d;
  m = fmaxf(fmaxf(m, fmaxf(fabsf(a), fabsf(b))), fmaxf(fabsf(c), fabsf(d)));
  a = fminf(fmaxf(a * scale, -fmax), fmax); b = fminf(fmaxf(b * scale, -fmax), fmax);
  c = fminf(fmaxf(c * scale, -fmax), fmax); d = fminf(fmaxf(d * scale, -fmax), fmax);
  if (fmt == FAVIT_E4M3)
    return ((unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffffu) |
           ((unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(c, d, 0, false) << 16);
  return ((unsigned)__builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false) & 0xffffu) |
         ((unsigned)__builtin_amdgcn_cvt_pk_bf8_f32(c, d, 0, false) << 16);
}
__device__ __forceinline__ void lnq8_end(const LnQ8& q8, float m, int lane, int wave) {
  m = wave_max(m);
  if (lane == 0 && m > 0.f)
    atomicMax(reinterpret_cast<unsigned int*>(q8.amax_next) + ((blockIdx.x * 4 + wave) & (FAVIT_FP8_AMAX_SLOTS - 1)),
              __float_as_uint(m));
}

template <typename OutT, int NV, bool Q8 = false>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, long ldx,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, OutT* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, long rows,
                                                     int D, float eps, LnQ8 q8) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float qscale = 1.0f, qmax = 0.f;
  if constexpr (Q8) qscale = lnq8_begin(q8, lane);
  // (one row per wave and launch without the quantisation; with it the grid is capped and a wave walks several rows,
  // so that the scale's round trip and the amax atomic are paid once per wave: 56 -> see tools/lnq8_bench.py)
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
  const float* xr = x + row * ldx;
  const int nchunk = D >> 2;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int i4 = lane + 64 * c;
    v[c] = (i4 < nchunk) ? *reinterpret_cast<const float4*>(xr + 4 * i4) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
  }
  const float mu = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int i4 = lane + 64 * c;
    if (i4 < nchunk) {
      const float a = v[c].x - mu, b = v[c].y - mu, cc = v[c].z - mu, d = v[c].w - mu;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
  OutT* yr = y + row * (long)D;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int i4 = lane + 64 * c;
    if (i4 < nchunk) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * i4);
      const float4 b = *reinterpret_cast<const float4*>(beta + 4 * i4);
      const float o0 = (v[c].x - mu) * rs * g.x + b.x, o1 = (v[c].y - mu) * rs * g.y + b.y;
      const float o2 = (v[c].z - mu) * rs * g.z + b.z, o3 = (v[c].w - mu) * rs * g.w + b.w;
      store4<OutT>(yr + 4 * i4, o0, o1, o2, o3);
      if constexpr (Q8)
        *reinterpret_cast<unsigned*>(q8.q + row * (long)D + 4 * i4) = lnq8_pack(o0, o1, o2, o3, qscale, q8.fmt, qmax);
    }
  }
  }
  if constexpr (Q8) lnq8_end(q8, qmax, lane, wave);
}

// sum over the 32 lanes of an aligned half wave, in the VALU (DPP within rows of 16, v_permlane16_swap across them)
__device__ __forceinline__ float half_wave_sum(float v) {
  v = dpp_sum8(v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));  // row_ror:8
  return v + lane_xor16(v);
}

// Forward for D <= 512: TWO rows per wave, 32 lanes per row, NV float4 per lane.  At D = 384 every lane carries three
// vectors (the one-row-per-wave kernel above leaves half of its second vector idle and has 1.5 KB in flight per wave),
// and the two reductions cost five VALU exchanges each instead of six ds_bpermute round trips.
template <typename OutT, int NV, bool Q8 = false>
__global__ __launch_bounds__(256) void ln_fwd_half_kernel(const float* __restrict__ x, long ldx,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, OutT* __restrict__ y,
                                                          float* __restrict__ mean, float* __restrict__ rstd, long rows,
                                                          int D, float eps, LnQ8 q8) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l32 = lane & 31;
  float qscale = 1.0f, qmax = 0.f;
  if constexpr (Q8) qscale = lnq8_begin(q8, lane);
  for (long pair = (long)blockIdx.x * 4 + wave; 2 * pair < rows; pair += (long)gridDim.x * 4) {
  const long row0 = pair * 2 + (lane >> 5);
  const bool live = row0 < rows;
  const long row = live ? row0 : rows - 1;               // a dead half recomputes the last row and stores nothing
  const float* xr = x + row * ldx;
  const int nchunk = D >> 2;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int i4 = l32 + 32 * c;
    v[c] = (i4 < nchunk) ? *reinterpret_cast<const float4*>(xr + 4 * i4) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
  }
  const float mu = half_wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int i4 = l32 + 32 * c;
    if (i4 < nchunk) {
      const float a = v[c].x - mu, b = v[c].y - mu, cc = v[c].z - mu, d = v[c].w - mu;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rs = rsqrtf(half_wave_sum(q) / (float)D + eps);
  if (live) {
    if (l32 == 0) {
      mean[row] = mu;
      rstd[row] = rs;
    }
    OutT* yr = y + row * (long)D;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const int i4 = l32 + 32 * c;
      if (i4 < nchunk) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * i4);
        const float4 b = *reinterpret_cast<const float4*>(beta + 4 * i4);
        const float o0 = (v[c].x - mu) * rs * g.x + b.x, o1 = (v[c].y - mu) * rs * g.y + b.y;
        const float o2 = (v[c].z - mu) * rs * g.z + b.z, o3 = (v[c].w - mu) * rs * g.w + b.w;
        store4<OutT>(yr + 4 * i4, o0, o1, o2, o3);
        if constexpr (Q8)
          *reinterpret_cast<unsigned*>(q8.q + row * (long)D + 4 * i4) = lnq8_pack(o0, o1, o2, o3, qscale, q8.fmt, qmax);
      }
    }
  }
  }
  if constexpr (Q8) lnq8_end(q8, qmax, lane, wave);
}

template <typename DyT, typename LpT, int NV, bool Q8 = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const DyT* __restrict__ dy, const float* __restrict__ x, long ldx,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ dres, float* __restrict__ dx,
                                                     long lddx, LpT* __restrict__ dx_lp,
                                                     float* __restrict__ dgamma_part, float* __restrict__ dbeta_part,
                                                     long rows, int D, uint32_t lp_thresh, float lp_scale,
                                                     uint64_t lp_seed, const unsigned long long* lp_epoch, LnQ8 q8) {
  float qscale = 1.0f, qmax = 0.f;
  if constexpr (Q8) qscale = lnq8_begin(q8, threadIdx.x & 63);
  if (lp_thresh) lp_seed = favit_eff_seed(lp_seed, lp_epoch);
  __shared__ float red[4][2][256 * NV > 2048 ? 2048 : 256 * NV];   // [wave][gamma|beta][D padded]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = D >> 2;
  float4 gam[NV], ag[NV], ab[NV];
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int i4 = lane + 64 * c;
    gam[c] = (i4 < nchunk) ? *reinterpret_cast<const float4*>(gamma + 4 * i4) : make_float4(0.f, 0.f, 0.f, 0.f);
    ag[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float invD = 1.0f / (float)D;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const float mu = mean[row], rs = rstd[row];
    const float* xr = x + row * ldx;
    const DyT* dyr = dy + row * (long)D;
    float4 xh[NV], g[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const int i4 = lane + 64 * c;
      if (i4 < nchunk) {
        const float4 xv = *reinterpret_cast<const float4*>(xr + 4 * i4);
        const float4 d = load4<DyT>(dyr + 4 * i4);
        xh[c] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
        g[c] = make_float4(d.x * gam[c].x, d.y * gam[c].y, d.z * gam[c].z, d.w * gam[c].w);
        s1 += (g[c].x + g[c].y) + (g[c].z + g[c].w);
        s2 += (g[c].x * xh[c].x + g[c].y * xh[c].y) + (g[c].z * xh[c].z + g[c].w * xh[c].w);
        ag[c].x += d.x * xh[c].x; ag[c].y += d.y * xh[c].y; ag[c].z += d.z * xh[c].z; ag[c].w += d.w * xh[c].w;
        ab[c].x += d.x; ab[c].y += d.y; ab[c].z += d.z; ab[c].w += d.w;
      } else {
        xh[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        g[c] = xh[c];
      }
    }
    const float c1 = wave_sum(s1) * invD, c2 = wave_sum(s2) * invD;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const int i4 = lane + 64 * c;
      if (i4 < nchunk) {
        float o0 = rs * (g[c].x - c1 - xh[c].x * c2), o1 = rs * (g[c].y - c1 - xh[c].y * c2);
        float o2 = rs * (g[c].z - c1 - xh[c].z * c2), o3 = rs * (g[c].w - c1 - xh[c].w * c2);
        if (dres) {
          const float4 r = *reinterpret_cast<const float4*>(dres + row * lddx + 4 * i4);
          o0 += r.x; o1 += r.y; o2 += r.z; o3 += r.w;
        }
        *reinterpret_cast<float4*>(dx + row * lddx + 4 * i4) = make_float4(o0, o1, o2, o3);
        if (dx_lp) {
          if (lp_thresh) {       // the low-precision copy only feeds a branch whose OUTPUT was dropped: apply that mask here
            const uint64_t e = (uint64_t)(row * (long)D + 4 * i4);       // (a multiple of 4: D is)
            bool k0, k1, k2, k3;
            favit_keep2(lp_seed, e, lp_thresh, k0, k1);
            favit_keep2(lp_seed, e + 2, lp_thresh, k2, k3);
            o0 = k0 ? o0 * lp_scale : 0.f;
            o1 = k1 ? o1 * lp_scale : 0.f;
            o2 = k2 ? o2 * lp_scale : 0.f;
            o3 = k3 ? o3 * lp_scale : 0.f;
          }
          store4<LpT>(dx_lp + row * (long)D + 4 * i4, o0, o1, o2, o3);
          if constexpr (Q8)
            *reinterpret_cast<unsigned*>(q8.q + row * (long)D + 4 * i4) = lnq8_pack(o0, o1, o2, o3, qscale, q8.fmt, qmax);
        }
      }
    }
  }
  if constexpr (Q8) lnq8_end(q8, qmax, lane, wave);
  // fold the 4 waves of the workgroup, then one partial row per workgroup
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int i4 = lane + 64 * c;
    if (i4 < nchunk) {
      *reinterpret_cast<float4*>(&red[wave][0][4 * i4]) = ag[c];
      *reinterpret_cast<float4*>(&red[wave][1][4 * i4]) = ab[c];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < D; i += 256) {
    dgamma_part[(long)blockIdx.x * D + i] = (red[0][0][i] + red[1][0][i]) + (red[2][0][i] + red[3][0][i]);
    dbeta_part[(long)blockIdx.x * D + i] = (red[0][1][i] + red[1][1][i]) + (red[2][1][i] + red[3][1][i]);
  }
}

__global__ __launch_bounds__(1024) void reduce_rows_kernel(const float* __restrict__ in, long ld,
                                                            float* __restrict__ out, float* __restrict__ out1,
                                                            long rows, int cols, int accumulate) {
  // block = 64 columns x 16 row groups; blockIdx.y = 1 selects the second stacked matrix -> out1;
  // gridDim.z > 1 splits the rows (then the result is added atomically; accumulate mode only).
  __shared__ float red[16][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cx;
  const float* src = in + (long)blockIdx.y * rows * ld;
  float* dst = blockIdx.y ? out1 : out;
  const long per = (rows + gridDim.z - 1) / gridDim.z;
  const long rbeg = (long)blockIdx.z * per, rend = min(rows, rbeg + per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < cols) {
    long r = rbeg + ry;
    for (; r + 48 < rend; r += 64) {                  // 4 independent loads in flight
      s0 += src[r * ld + col];
      s1 += src[(r + 16) * ld + col];
      s2 += src[(r + 32) * ld + col];
      s3 += src[(r + 48) * ld + col];
    }
    for (; r < rend; r += 16) s0 += src[r * ld + col];
  }
  red[ry][cx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ry == 0 && col < cols) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][cx];
    if (gridDim.z > 1) atomicAdd(dst + col, t);
    else dst[col] = accumulate ? dst[col] + t : t;
  }
}

// Several stacked-pair reductions in ONE launch (the dgamma / dbeta partials of up to REDUCE_MULTI_MAX LayerNorm
// backward passes: 24 launches of ~5 us per cfg2 step otherwise).  Entry e: in [2][rows][cols] -> out0, out1 [cols],
// ADDED to the destinations (rows split over gridDim.z / n blocks, fp32 atomics as in the single form).
constexpr int REDUCE_MULTI_MAX = 32;
struct ReduceMulti {
  const float* in[REDUCE_MULTI_MAX];
  float* out0[REDUCE_MULTI_MAX];
  float* out1[REDUCE_MULTI_MAX];
};
__global__ __launch_bounds__(1024) void reduce_rows_multi_kernel(ReduceMulti rm, long rows, int cols, int zsplit) {
  __shared__ float red[16][64];
  const int e = blockIdx.z / zsplit, z = blockIdx.z % zsplit;
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cx;
  const float* src = rm.in[e] + (long)blockIdx.y * rows * cols;
  float* dst = blockIdx.y ? rm.out1[e] : rm.out0[e];
  const long per = (rows + zsplit - 1) / zsplit;
  const long rbeg = (long)z * per, rend = min(rows, rbeg + per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < cols) {
    long r = rbeg + ry;
    for (; r + 48 < rend; r += 64) {
      s0 += src[r * cols + col];
      s1 += src[(r + 16) * cols + col];
      s2 += src[(r + 32) * cols + col];
      s3 += src[(r + 48) * cols + col];
    }
    for (; r < rend; r += 16) s0 += src[r * cols + col];
  }
  red[ry][cx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ry == 0 && col < cols) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][cx];
    atomicAdd(dst + col, t);
  }
}

// ---------------------------------------------------------------------------------
template <typename S, typename T>
__global__ void cast_kernel(const S* __restrict__ src, T* __restrict__ dst, long n) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = load4<S>(src + 4 * i);
    store4<T>(dst + 4 * i, v.x, v.y, v.z, v.w);
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dst[i] = from_f32<T>(to_f32(src[i]));
}

template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, long n, uint32_t thresh, float scale,
                               uint64_t seed, const unsigned long long* epoch) {
  seed = favit_eff_seed(seed, epoch);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = favit_keep(seed, (uint64_t)i, thresh) ? from_f32<T>(to_f32(x[i]) * scale) : from_f32<T>(0.f);
}

// ---------------------------------------------------------------------------------
// patchify: 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)'  (models/vit.py:38-39)
// ---------------------------------------------------------------------------------
template <typename T>
__global__ void patchify_fwd_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int C, int HW, int P) {
  const int g = HW / P;
  const long K = (long)P * P * C;
  const long total = (long)B * g * g * K;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / K;
    const int e = (int)(i - row * K);
    const int c = e % C, pp = e / C, p2 = pp % P, p1 = pp / P;
    const int b = (int)(row / (g * g)), pr = (int)(row % (g * g)), ph = pr / g, pw = pr % g;
    out[i] = from_f32<T>(img[(((long)b * C + c) * HW + (ph * P + p1)) * HW + pw * P + p2]);
  }
}

// Strip version: one workgroup owns the P image rows of one (image, patch-row) -- C x P x HW elements.
// It reads them with coalesced 16-byte loads into LDS (already converted to T) and writes the g patches
// of the strip, K = P*P*C contiguous elements each, with 16-byte stores; the (p1 p2 c) shuffle happens
// between LDS and registers.  Needs HW % 4 == 0 and K % VW == 0 (VW = elements per 16 bytes).
template <typename T>
__global__ __launch_bounds__(256) void patchify_fwd_strip_kernel(const float* __restrict__ img, T* __restrict__ out, int B,
                                                                 int C, int HW, int P) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* s = reinterpret_cast<T*>(smem_raw);                  // [C][P][HW]
  constexpr int VW = 16 / (int)sizeof(T);
  const int g = HW / P, ph = blockIdx.x % g, b = blockIdx.x / g;
  const int rowq = HW / 4, nq = C * P * rowq;             // float4 quads of the strip
  for (int q = threadIdx.x; q < nq; q += 256) {
    const int x4 = q % rowq, r = q / rowq, p1 = r % P, c = r / P;
    const float4 v = *reinterpret_cast<const float4*>(img + (((long)b * C + c) * HW + (ph * P + p1)) * HW + 4 * x4);
    T* d = s + (c * P + p1) * HW + 4 * x4;
    d[0] = from_f32<T>(v.x); d[1] = from_f32<T>(v.y); d[2] = from_f32<T>(v.z); d[3] = from_f32<T>(v.w);
  }
  __syncthreads();
  const int K = P * P * C, nv = g * K / VW;
  for (int i = threadIdx.x; i < nv; i += 256) {
    const int e0 = (i * VW) % K, pw = (i * VW) / K;
    T v[VW];
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      const int e = e0 + j, c = e % C, pp = e / C, p2 = pp % P, p1 = pp / P;
      v[j] = s[(c * P + p1) * HW + pw * P + p2];
    }
    *reinterpret_cast<uint4*>(out + ((long)b * g * g + ph * g + pw) * K + e0) = *reinterpret_cast<const uint4*>(v);
  }
}

// Three-channel strip version (every model on the path: RGB).  The interleave (c fastest) is done in REGISTERS while
// the strip is loaded -- a thread takes the same four pixels of the three colour planes (3 coalesced 16-byte loads) and
// writes their 12 interleaved values as three 4-element vectors -- so the LDS image [p1][x][c] already holds every patch
// row (P*3 contiguous elements) in output order and the second phase is 16-byte LDS reads -> 16-byte stores with no
// per-element index arithmetic (the generic strip kernel's 8 scalar LDS reads + div/mod per output vector made it
// VALU / LDS-issue bound: 101 us for the 231 MB of a 256-image batch; HBM rate would be ~46 us).
template <typename T>
__global__ __launch_bounds__(256) void patchify_fwd_rgb_kernel(const float* __restrict__ img, T* __restrict__ out, int B,
                                                               int HW, int P) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* s = reinterpret_cast<T*>(smem_raw);                  // [P][HW][3]
  constexpr int VW = 16 / (int)sizeof(T);
  typedef T vec4_t __attribute__((ext_vector_type(4)));
  const int g = HW / P, ph = blockIdx.x % g, b = blockIdx.x / g;
  const int rowq = HW / 4, nq = P * rowq;                 // groups of four pixels in the strip
  const long plane = (long)HW * HW;
  const float* base = img + (long)b * 3 * plane + (long)(ph * P) * HW;
  for (int q = threadIdx.x; q < nq; q += 256) {
    const int x4 = q % rowq, p1 = q / rowq;
    const float* src = base + (long)p1 * HW + 4 * x4;
    const float4 r = *reinterpret_cast<const float4*>(src);
    const float4 gg = *reinterpret_cast<const float4*>(src + plane);
    const float4 bl = *reinterpret_cast<const float4*>(src + 2 * plane);
    vec4_t* d = reinterpret_cast<vec4_t*>(s + ((long)p1 * HW + 4 * x4) * 3);
    d[0] = (vec4_t){from_f32<T>(r.x), from_f32<T>(gg.x), from_f32<T>(bl.x), from_f32<T>(r.y)};
    d[1] = (vec4_t){from_f32<T>(gg.y), from_f32<T>(bl.y), from_f32<T>(r.z), from_f32<T>(gg.z)};
    d[2] = (vec4_t){from_f32<T>(bl.z), from_f32<T>(r.w), from_f32<T>(gg.w), from_f32<T>(bl.w)};
  }
  __syncthreads();
  const int PR = P * 3, K = P * PR, vpr = PR / VW;        // elements / vectors per patch row
  const int nv = g * P * vpr;
  for (int i = threadIdx.x; i < nv; i += 256) {
    const int v = i % vpr, rr = i / vpr, p1 = rr % P, pw = rr / P;
    const uint4 val = *reinterpret_cast<const uint4*>(s + ((long)p1 * HW + pw * P) * 3 + v * VW);
    *reinterpret_cast<uint4*>(out + ((long)b * g * g + ph * g + pw) * K + p1 * PR + v * VW) = val;
  }
}

__global__ void patchify_bwd_kernel(const float* __restrict__ dpatch, float* __restrict__ dimg, int B, int C, int HW,
                                    int P) {
  const int g = HW / P;
  const long K = (long)P * P * C;
  const long total = (long)B * C * HW * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xx = (int)(i % HW), yy = (int)((i / HW) % HW), c = (int)((i / ((long)HW * HW)) % C);
    const int b = (int)(i / ((long)HW * HW * C));
    const int ph = yy / P, p1 = yy % P, pw = xx / P, p2 = xx % P;
    dimg[i] = dpatch[((long)b * g * g + ph * g + pw) * K + (p1 * P + p2) * C + c];
  }
}

// x[b,0,:] = cls + pos[0]; x[b,1+n,:] = tok[b,n,:] + pos[1+n]   (models/vit.py:292-296)
__global__ void embed_prologue_fwd_kernel(const float* __restrict__ tok, const float* __restrict__ cls,
                                          const float* __restrict__ pos, float* __restrict__ x, int B, int N, int D) {
  const int d4n = D >> 2;
  const long total = (long)B * (N + 1) * d4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int d4 = (int)(i % d4n);
    const long r = i / d4n;
    const int l = (int)(r % (N + 1));
    const long b = r / (N + 1);
    float4 v = (l == 0) ? *reinterpret_cast<const float4*>(cls + 4 * d4)
                        : *reinterpret_cast<const float4*>(tok + ((b * N + (l - 1)) * (long)D) + 4 * d4);
    if (pos) {
      const float4 p = *reinterpret_cast<const float4*>(pos + (long)l * D + 4 * d4);
      v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    *reinterpret_cast<float4*>(x + r * (long)D + 4 * d4) = v;
  }
}

// dtok = dx[:,1:], dcls = sum_b dx[b,0], dpos[l] = sum_b dx[b,l]
template <typename T>
__global__ void embed_prologue_bwd_kernel(const float* __restrict__ dx, T* __restrict__ dtok, float* __restrict__ dcls,
                                          float* __restrict__ dpos, int B, int N, int D) {
  const long LD = (long)(N + 1) * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < LD; i += (long)gridDim.x * blockDim.x) {
    const int l = (int)(i / D), d = (int)(i % D);
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
      const float v = dx[(long)b * LD + i];
      s += v;
      if (l > 0 && dtok) dtok[((long)b * N + (l - 1)) * D + d] = from_f32<T>(v);
    }
    if (dpos) dpos[i] = s;
    if (l == 0 && dcls) dcls[d] = s;
  }
}

// Chunked version: grid.y splits the batch; every thread owns one float4 of a token row position (l, d4),
// walks its chunk of images with 16-byte loads / stores and adds its partial sums with fp32 atomics
// (dcls / dpos are zeroed by the caller of this kernel).  Needs D % 4 == 0.
template <typename T>
__global__ __launch_bounds__(256) void embed_prologue_bwd_chunk_kernel(const float* __restrict__ dx, T* __restrict__ dtok,
                                                                       float* __restrict__ dcls, float* __restrict__ dpos,
                                                                       int B, int N, int D, int bchunk) {
  const int d4n = D >> 2;
  const long nq = (long)(N + 1) * d4n;
  const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const int l = (int)(q / d4n), d = 4 * (int)(q % d4n);
  const int b0 = blockIdx.y * bchunk, b1 = min(B, b0 + bchunk);
  const long LD = (long)(N + 1) * D;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int b = b0; b < b1; ++b) {
    const float4 v = *reinterpret_cast<const float4*>(dx + (long)b * LD + (long)l * D + d);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    if (l > 0 && dtok) {
      T* o = dtok + ((long)b * N + (l - 1)) * D + d;
      if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4*>(o) = v;
      } else {
        bf16x4 t;
        t[0] = (bf16_t)v.x; t[1] = (bf16_t)v.y; t[2] = (bf16_t)v.z; t[3] = (bf16_t)v.w;
        *reinterpret_cast<bf16x4*>(o) = t;
      }
    }
  }
  if (dpos) {
    float* p = dpos + (long)l * D + d;
    atomicAdd(p, s.x); atomicAdd(p + 1, s.y); atomicAdd(p + 2, s.z); atomicAdd(p + 3, s.w);
  }
  if (l == 0 && dcls) {
    atomicAdd(dcls + d, s.x); atomicAdd(dcls + d + 1, s.y); atomicAdd(dcls + d + 2, s.z); atomicAdd(dcls + d + 3, s.w);
  }
}

// mean cross-entropy rows: loss_rows[b] = lse - logit[label]; dlogits = (softmax - onehot) * grad_scale
__global__ __launch_bounds__(256) void cross_entropy_kernel(const float* __restrict__ logits,
                                                            const int64_t* __restrict__ labels,
                                                            float* __restrict__ loss_rows, float* __restrict__ dlogits,
                                                            int B, int C, float grad_scale, unsigned* __restrict__ health) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= B) return;
  const float* lr = logits + (long)row * C;
  float m = -INFINITY;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, lr[c]);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += __expf(lr[c] - m);
  s = wave_sum(s);
  const float lse = m + __logf(s);
  const int64_t lab64 = labels[row];
  const bool lab_ok = lab64 >= 0 && lab64 < C;       // out of range (e.g. ignore_index): NaN row, never an OOB read
  const int lab = lab_ok ? (int)lab64 : -1;
  if (lane == 0) {
    const float lrow = lab_ok ? lse - lr[lab] : __builtin_nanf("");
    loss_rows[row] = lrow;
    // health word (favit_set_health_word): a non-finite loss row of an in-range label = non-finite logits
    if (health && lab_ok && !isfinite(lrow) && !(atomicOr(health, 1u) & 1u)) health[1] = health[3] + 1;
  }
  if (dlogits) {
    const float inv = 1.0f / s;
    for (int c = lane; c < C; c += 64)
      dlogits[(long)row * C + c] = (__expf(lr[c] - m) * inv - (c == lab ? 1.f : 0.f)) * grad_scale;
  }
}

// torch.optim.AdamW semantics (decoupled weight decay), one flat fp32 chunk
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, bf16_t* __restrict__ p_lp, long n, float lr, float b1, float b2,
                             float eps, float wd, float bc1, float bc2, float gscale, unsigned* __restrict__ health) {
  const float step = lr / bc1, rbc2 = rsqrtf(bc2);
  bool bad_g = false, bad_p = false;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    pi -= step * mi / (sqrtf(vi) * rbc2 + eps);
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (p_lp) p_lp[i] = (bf16_t)pi;
    bad_g |= !isfinite(gi);
    bad_p |= !isfinite(pi);
  }
  // Health word (favit_set_health_word; null = off): the update reads every gradient and writes every parameter
  // anyway, so noticing the first non-finite one costs two compares per element and no memory traffic.  [0] flags
  // (1 loss row, 2 gradient, 4 updated parameter), [1] / [2] value of the AdamW launch counter [3] (+1) when the
  // flag bits 1 / (2|4) were first set, [3] AdamW launches so far.  A clean run performs no atomic at all.
  if (health) {
    const unsigned f = (bad_g ? 2u : 0u) | (bad_p ? 4u : 0u);
    if (f && !(atomicOr(health, f) & 6u)) health[2] = health[3] + 1;
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(health + 3, 1u);   // (stream order: one AdamW launch at a time)
  }
}

inline int grid_for(long n, int block = 256, int cap = 4096) {
  long g = (n + block - 1) / block;
  if (g < 1) g = 1;
  return (int)(g > cap ? cap : g);
}

}  // namespace

#define LN_DISPATCH_NV(D, MACRO)            \
  do {                                      \
    const int nv__ = ((D) + 255) / 256;     \
    if (nv__ <= 1) { MACRO(1); }            \
    else if (nv__ <= 2) { MACRO(2); }       \
    else if (nv__ <= 3) { MACRO(3); }       \
    else if (nv__ <= 4) { MACRO(4); }       \
    else if (nv__ <= 8) { MACRO(8); }       \
    else return FAVIT_ERR_UNSUPPORTED;      \
  } while (0)

namespace {
int layernorm_fwd_impl(const float* x, int64_t ldx, const float* gamma, const float* beta, void* y, int y_dtype, float* mean,
                       float* rstd, int64_t rows, int32_t D, float eps, const LnQ8& q8, void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || rows <= 0 || D <= 0) return FAVIT_ERR_INVALID;
  if ((D & 3) || (ldx & 3)) return FAVIT_ERR_ALIGN;
  hipStream_t st = as_stream(stream);
  if (D <= 512 && getenv("FAVIT_LN_ONE_ROW") == nullptr) {          // two rows per wave
    const long nb2 = (rows + 7) / 8;
    const dim3 grid2((unsigned)(q8.q && nb2 > 2048 ? 2048 : nb2));
#define LN_FWD2(NV)                                                                                                 \
    do {                                                                                                            \
      if (q8.q)                                                                                                     \
        hipLaunchKernelGGL((ln_fwd_half_kernel<bf16_t, NV, true>), grid2, dim3(256), 0, st, x, (long)ldx, gamma,    \
                           beta, (bf16_t*)y, mean, rstd, (long)rows, D, eps, q8);                                   \
      else if (y_dtype == FAVIT_F32)                                                                                \
        hipLaunchKernelGGL((ln_fwd_half_kernel<float, NV>), grid2, dim3(256), 0, st, x, (long)ldx, gamma, beta,     \
                           (float*)y, mean, rstd, (long)rows, D, eps, q8);                                          \
      else                                                                                                          \
        hipLaunchKernelGGL((ln_fwd_half_kernel<bf16_t, NV>), grid2, dim3(256), 0, st, x, (long)ldx, gamma, beta,    \
                           (bf16_t*)y, mean, rstd, (long)rows, D, eps, q8);                                         \
    } while (0)
    const int nv2 = (D + 127) / 128;
    if (nv2 <= 1) LN_FWD2(1); else if (nv2 == 2) LN_FWD2(2); else if (nv2 == 3) LN_FWD2(3); else LN_FWD2(4);
#undef LN_FWD2
    FAVIT_CHECK_LAUNCH();
    return FAVIT_OK;
  }
  const long nb1 = (rows + 3) / 4;
  const dim3 grid((unsigned)(q8.q && nb1 > 2048 ? 2048 : nb1));
#define LN_FWD(NV)                                                                                              \
  if (q8.q)                                                                                                     \
    hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, NV, true>), grid, dim3(256), 0, st, x, (long)ldx, gamma, beta,    \
                       (bf16_t*)y, mean, rstd, (long)rows, D, eps, q8);                                         \
  else if (y_dtype == FAVIT_F32)                                                                                \
    hipLaunchKernelGGL((ln_fwd_kernel<float, NV>), grid, dim3(256), 0, st, x, (long)ldx, gamma, beta, (float*)y, \
                       mean, rstd, (long)rows, D, eps, q8);                                                     \
  else                                                                                                          \
    hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, NV>), grid, dim3(256), 0, st, x, (long)ldx, gamma, beta,          \
                       (bf16_t*)y, mean, rstd, (long)rows, D, eps, q8)
  LN_DISPATCH_NV(D, LN_FWD);
#undef LN_FWD
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

int layernorm_bwd_impl(const void* dy, int dy_dtype, const float* x, int64_t ldx, const float* gamma, const float* mean,
                       const float* rstd, const float* dres, float* dx, int64_t lddx, void* dx_lp, int lp_dtype,
                       float* dgamma_part, float* dbeta_part, int32_t nparts, float* dgamma, float* dbeta,
                       int32_t accumulate, int64_t rows, int32_t D, float lp_dropout_p, uint64_t lp_dropout_seed,
                       const LnQ8& q8, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma_part || !dbeta_part || rows <= 0 || D <= 0 ||
      nparts <= 0)
    return FAVIT_ERR_INVALID;
  if (dbeta_part != dgamma_part + (long)nparts * D) return FAVIT_ERR_INVALID;   // [2][nparts][D] workspace
  if ((D & 3) || (ldx & 3) || (lddx & 3)) return FAVIT_ERR_ALIGN;
  if (dy_dtype != lp_dtype && dx_lp) return FAVIT_ERR_UNSUPPORTED;
  if (lp_dropout_p < 0.f || lp_dropout_p >= 1.f) return FAVIT_ERR_INVALID;
  if (q8.q && (!dx_lp || dy_dtype != FAVIT_BF16)) return FAVIT_ERR_UNSUPPORTED;
  const uint32_t lp_thresh = dx_lp ? dropout_threshold(lp_dropout_p) : 0u;
  const float lp_scale = 1.0f / (1.0f - lp_dropout_p);
  hipStream_t st = as_stream(stream);
  const dim3 grid((unsigned)nparts);
#define LN_BWD(NV)                                                                                                 \
  if (q8.q)                                                                                                        \
    hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, bf16_t, NV, true>), grid, dim3(256), 0, st, (const bf16_t*)dy, x,    \
                       (long)ldx, gamma, mean, rstd, dres, dx, (long)lddx, (bf16_t*)dx_lp, dgamma_part,            \
                       dbeta_part, (long)rows, D, lp_thresh, lp_scale, lp_dropout_seed, favit_dropout_epoch_ptr_(), q8); \
  else if (dy_dtype == FAVIT_F32)                                                                                  \
    hipLaunchKernelGGL((ln_bwd_kernel<float, float, NV>), grid, dim3(256), 0, st, (const float*)dy, x, (long)ldx,  \
                       gamma, mean, rstd, dres, dx, (long)lddx, (float*)dx_lp, dgamma_part, dbeta_part, (long)rows, \
                       D, lp_thresh, lp_scale, lp_dropout_seed, favit_dropout_epoch_ptr_(), q8);                   \
  else                                                                                                             \
    hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, bf16_t, NV>), grid, dim3(256), 0, st, (const bf16_t*)dy, x,          \
                       (long)ldx, gamma, mean, rstd, dres, dx, (long)lddx, (bf16_t*)dx_lp, dgamma_part,            \
                       dbeta_part, (long)rows, D, lp_thresh, lp_scale, lp_dropout_seed, favit_dropout_epoch_ptr_(), q8)
  LN_DISPATCH_NV(D, LN_BWD);
#undef LN_BWD
  FAVIT_CHECK_LAUNCH();
  if (dgamma && dbeta) {
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((D + 63) / 64, 2, accumulate ? 8 : 1), dim3(1024), 0, st, dgamma_part,
                       (long)D, dgamma, dbeta, (long)nparts, D, accumulate);
    FAVIT_CHECK_LAUNCH();
  }
  return FAVIT_OK;
}

bool lnq8_args_ok(const void* q, int fmt, const float* amax, float* scale_inv, float* amax_next, float* amax_clear) {
  return q && amax && scale_inv && amax_next && amax_clear && (fmt == FAVIT_E4M3 || fmt == FAVIT_E5M2) &&
         amax != amax_next && amax != amax_clear && amax_next != amax_clear;
}
}  // namespace

extern "C" int favit_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, void* y,
                                   int y_dtype, float* mean, float* rstd, int64_t rows, int32_t D, float eps,
                                   void* stream) {
  const LnQ8 off = {nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  return layernorm_fwd_impl(x, ldx, gamma, beta, y, y_dtype, mean, rstd, rows, D, eps, off, stream);
}

extern "C" int favit_layernorm_fwd_q8(const float* x, int64_t ldx, const float* gamma, const float* beta, void* y,
                                      float* mean, float* rstd, int64_t rows, int32_t D, float eps, void* q, int fmt,
                                      const float* amax, float* scale_inv, float* amax_next, float* amax_clear,
                                      void* stream) {
  if (!lnq8_args_ok(q, fmt, amax, scale_inv, amax_next, amax_clear)) return FAVIT_ERR_INVALID;
  const LnQ8 q8 = {reinterpret_cast<uint8_t*>(q), amax, scale_inv, amax_next, amax_clear, fmt};
  return layernorm_fwd_impl(x, ldx, gamma, beta, y, FAVIT_BF16, mean, rstd, rows, D, eps, q8, stream);
}

extern "C" int favit_layernorm_bwd(const void* dy, int dy_dtype, const float* x, int64_t ldx, const float* gamma,
                                   const float* mean, const float* rstd, const float* dres, float* dx, int64_t lddx,
                                   void* dx_lp, int lp_dtype, float* dgamma_part, float* dbeta_part, int32_t nparts,
                                   float* dgamma, float* dbeta, int32_t accumulate, int64_t rows, int32_t D,
                                   float lp_dropout_p, uint64_t lp_dropout_seed, void* stream) {
  const LnQ8 off = {nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  return layernorm_bwd_impl(dy, dy_dtype, x, ldx, gamma, mean, rstd, dres, dx, lddx, dx_lp, lp_dtype, dgamma_part,
                            dbeta_part, nparts, dgamma, dbeta, accumulate, rows, D, lp_dropout_p, lp_dropout_seed, off,
                            stream);
}

extern "C" int favit_layernorm_bwd_q8(const void* dy, const float* x, int64_t ldx, const float* gamma, const float* mean,
                                      const float* rstd, const float* dres, float* dx, int64_t lddx, void* dx_lp,
                                      float* dgamma_part, float* dbeta_part, int32_t nparts, float* dgamma, float* dbeta,
                                      int32_t accumulate, int64_t rows, int32_t D, float lp_dropout_p,
                                      uint64_t lp_dropout_seed, void* q, int fmt, const float* amax, float* scale_inv,
                                      float* amax_next, float* amax_clear, void* stream) {
  if (!lnq8_args_ok(q, fmt, amax, scale_inv, amax_next, amax_clear) || !dx_lp) return FAVIT_ERR_INVALID;
  const LnQ8 q8 = {reinterpret_cast<uint8_t*>(q), amax, scale_inv, amax_next, amax_clear, fmt};
  return layernorm_bwd_impl(dy, FAVIT_BF16, x, ldx, gamma, mean, rstd, dres, dx, lddx, dx_lp, FAVIT_BF16, dgamma_part,
                            dbeta_part, nparts, dgamma, dbeta, accumulate, rows, D, lp_dropout_p, lp_dropout_seed, q8,
                            stream);
}

// ---------------------------------------------------------------------------------
// Classification head (nn.Linear(D, num_classes) on the normalised CLS row: models/vit.py:259,306,
// models/vit_mhla.py:156,249, models/sppp_mhla.py:240,318): M = batch rows, N = a few classes.  As a GEMM it is ONE
// 128x128 tile -- a single workgroup walking K in 64-deep steps, 13-21 us per launch, three launches plus two casts per
// step (3.7 % of the cfg1 step).  Here: exact fp32 dot products on the VALU, one launch forward, one backward, no
// atomics (deterministic), 3-4 us each.
// ---------------------------------------------------------------------------------
namespace {
constexpr int SL_MAX_N = 64;

__global__ __launch_bounds__(256) void small_linear_fwd_kernel(const float* __restrict__ x, long ldx,
                                                               const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y,
                                                               int N, int K) {
  const int m = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* xr = x + (long)m * ldx;
  for (int n = wave; n < N; n += 4) {
    const float* wr = w + (long)n * K;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s = fmaf(xr[k], wr[k], s);
    s = wave_sum(s);
    if (lane == 0) y[(long)m * N + n] = s + (bias ? bias[n] : 0.f);
  }
}

// blocks [0, M): dx rows (if dx); blocks [M, M + N): one weight row each (+ its bias gradient)
__global__ __launch_bounds__(256) void small_linear_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               long ldx, const float* __restrict__ w,
                                                               float* __restrict__ dx, float* __restrict__ dw,
                                                               float* __restrict__ db, int accumulate, int M, int N,
                                                               int K) {
  __shared__ float sh[SL_MAX_N > 256 ? SL_MAX_N : 256];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < M) {
    if (!dx) return;
    const int m = blockIdx.x;
    if (tid < N) sh[tid] = dy[(long)m * N + tid];
    __syncthreads();
    for (int k = tid; k < K; k += 256) {
      float s = 0.f;
      for (int n = 0; n < N; ++n) s = fmaf(sh[n], w[(long)n * K + k], s);
      dx[(long)m * K + k] = s;
    }
    return;
  }
  const int n = blockIdx.x - M;
  float bsum = 0.f;
  for (int k0 = 0; k0 < K; k0 += 256) {
    const int k = k0 + tid;
    float s = 0.f;
    for (int m0 = 0; m0 < M; m0 += 256) {              // dy[:, n] of 256 rows through LDS, x read coalesced along k
      __syncthreads();
      const int mm = m0 + tid;
      sh[tid] = mm < M ? dy[(long)mm * N + n] : 0.f;
      __syncthreads();
      const int cnt = min(256, M - m0);
      if (k < K)
        for (int i = 0; i < cnt; ++i) s = fmaf(sh[i], x[(long)(m0 + i) * ldx + k], s);
      if (k0 == 0 && tid == 0)
        for (int i = 0; i < cnt; ++i) bsum += sh[i];
    }
    if (k < K) dw[(long)n * K + k] = accumulate ? dw[(long)n * K + k] + s : s;
  }
  if (db && tid == 0) db[n] = accumulate ? db[n] + bsum : bsum;
}
}  // namespace

extern "C" int favit_small_linear_fwd(const float* x, int64_t ldx, const float* w, const float* bias, float* y, int32_t M,
                                      int32_t N, int32_t K, void* stream) {
  if (!x || !w || !y || M <= 0 || N <= 0 || K <= 0 || ldx < K) return FAVIT_ERR_INVALID;
  if (N > SL_MAX_N) return FAVIT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(small_linear_fwd_kernel, dim3(M), dim3(256), 0, as_stream(stream), x, (long)ldx, w, bias, y, N, K);
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

extern "C" int favit_small_linear_bwd(const float* dy, const float* x, int64_t ldx, const float* w, float* dx, float* dw,
                                      float* db, int32_t accumulate, int32_t M, int32_t N, int32_t K, void* stream) {
  if (!dy || !x || !w || !dw || M <= 0 || N <= 0 || K <= 0 || ldx < K) return FAVIT_ERR_INVALID;
  if (N > SL_MAX_N) return FAVIT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(small_linear_bwd_kernel, dim3(M + N), dim3(256), 0, as_stream(stream), dy, x, (long)ldx, w, dx, dw, db,
                     accumulate, M, N, K);
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

extern "C" int favit_reduce_rows(const float* in, int64_t ld, float* out, int64_t rows, int32_t cols,
                                 int32_t accumulate, void* stream) {
  if (!in || !out || rows <= 0 || cols <= 0) return FAVIT_ERR_INVALID;
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((cols + 63) / 64, 1), dim3(1024), 0, as_stream(stream), in, (long)ld, out,
                     (float*)nullptr, (long)rows, cols, accumulate);
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

extern "C" int favit_reduce_rows_multi(int32_t n, const float* const* in, float* const* out0, float* const* out1,
                                       int64_t rows, int32_t cols, void* stream) {
  if (n <= 0 || n > REDUCE_MULTI_MAX || !in || !out0 || !out1 || rows <= 0 || cols <= 0) return FAVIT_ERR_INVALID;
  ReduceMulti rm;
  for (int i = 0; i < n; ++i) {
    if (!in[i] || !out0[i] || !out1[i]) return FAVIT_ERR_INVALID;
    rm.in[i] = in[i]; rm.out0[i] = out0[i]; rm.out1[i] = out1[i];
  }
  const int zsplit = rows >= 512 ? 8 : 1;
  hipLaunchKernelGGL(reduce_rows_multi_kernel, dim3((cols + 63) / 64, 2, (unsigned)(n * zsplit)), dim3(1024), 0,
                     as_stream(stream), rm, (long)rows, cols, zsplit);
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

extern "C" int favit_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream) {
  if (!src || !dst || n < 0) return FAVIT_ERR_INVALID;
  if (n == 0) return FAVIT_OK;
  if ((reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(dst) & 7)) return FAVIT_ERR_ALIGN;
  hipStream_t st = as_stream(stream);
  const int g = grid_for((n + 3) / 4);
  if (src_dtype == FAVIT_F32 && dst_dtype == FAVIT_BF16)
    hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(g), dim3(256), 0, st, (const float*)src, (bf16_t*)dst, (long)n);
  else if (src_dtype == FAVIT_BF16 && dst_dtype == FAVIT_F32)
    hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(g), dim3(256), 0, st, (const bf16_t*)src, (float*)dst, (long)n);
  else if (src_dtype == FAVIT_F32 && dst_dtype == FAVIT_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), dim3(g), dim3(256), 0, st, (const float*)src, (float*)dst, (long)n);
  else if (src_dtype == FAVIT_BF16 && dst_dtype == FAVIT_BF16)
    hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, (long)n);
  else
    return FAVIT_ERR_INVALID;
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

extern "C" int favit_dropout(const void* x, void* y, int dtype, int64_t n, float p, uint64_t seed, void* stream) {
  if (!x || !y || n < 0 || p < 0.f || p >= 1.f) return FAVIT_ERR_INVALID;
  if (n == 0) return FAVIT_OK;
  hipStream_t st = as_stream(stream);
  const uint32_t th = dropout_threshold(p);
  const float scale = 1.0f / (1.0f - p);
  if (dtype == FAVIT_F32)
    hipLaunchKernelGGL((dropout_kernel<float>), dim3(grid_for(n)), dim3(256), 0, st, (const float*)x, (float*)y, (long)n, th, scale, seed, favit_dropout_epoch_ptr_());
  else if (dtype == FAVIT_BF16)
    hipLaunchKernelGGL((dropout_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, (long)n, th, scale, seed, favit_dropout_epoch_ptr_());
  else
    return FAVIT_ERR_INVALID;
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

extern "C" int favit_patchify_fwd(const float* img, void* out, int out_dtype, int32_t B, int32_t C, int32_t HW,
                                  int32_t P, void* stream) {
  if (!img || !out || B <= 0 || C <= 0 || HW <= 0 || P <= 0 || HW % P) return FAVIT_ERR_INVALID;
  hipStream_t st = as_stream(stream);
  const long total = (long)B * C * HW * HW;
  {
    const int esz = out_dtype == FAVIT_F32 ? 4 : 2, vw = 16 / esz;
    const size_t lds = (size_t)C * P * HW * esz;
    const long Kp = (long)P * P * C;
    // RGB fast path: (P*3) % vw == 0 keeps every patch row a whole number of 16-byte vectors
    if (C == 3 && (out_dtype == FAVIT_F32 || out_dtype == FAVIT_BF16) && (HW % 4) == 0 && ((P * 3) % vw) == 0 &&
        lds <= 64 * 1024 && (reinterpret_cast<uintptr_t>(img) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
      const dim3 grid((unsigned)(B * (HW / P)));
      if (out_dtype == FAVIT_F32)
        hipLaunchKernelGGL((patchify_fwd_rgb_kernel<float>), grid, dim3(256), lds, st, img, (float*)out, B, HW, P);
      else
        hipLaunchKernelGGL((patchify_fwd_rgb_kernel<bf16_t>), grid, dim3(256), lds, st, img, (bf16_t*)out, B, HW, P);
      FAVIT_CHECK_LAUNCH();
      return FAVIT_OK;
    }
    if ((out_dtype == FAVIT_F32 || out_dtype == FAVIT_BF16) && (HW % 4) == 0 && (Kp % vw) == 0 && lds <= 64 * 1024 &&
        (reinterpret_cast<uintptr_t>(img) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
      const dim3 grid((unsigned)(B * (HW / P)));
      if (out_dtype == FAVIT_F32)
        hipLaunchKernelGGL((patchify_fwd_strip_kernel<float>), grid, dim3(256), lds, st, img, (float*)out, B, C, HW, P);
      else
        hipLaunchKernelGGL((patchify_fwd_strip_kernel<bf16_t>), grid, dim3(256), lds, st, img, (bf16_t*)out, B, C, HW, P);
      FAVIT_CHECK_LAUNCH();
      return FAVIT_OK;
    }
  }
  if (out_dtype == FAVIT_F32)
    hipLaunchKernelGGL((patchify_fwd_kernel<float>), dim3(grid_for(total, 256, 8192)), dim3(256), 0, st, img, (float*)out, B, C, HW, P);
  else if (out_dtype == FAVIT_BF16)
    hipLaunchKernelGGL((patchify_fwd_kernel<bf16_t>), dim3(grid_for(total, 256, 8192)), dim3(256), 0, st, img, (bf16_t*)out, B, C, HW, P);
  else
    return FAVIT_ERR_INVALID;
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

extern "C" int favit_patchify_bwd(const float* dpatch, float* dimg, int32_t B, int32_t C, int32_t HW, int32_t P,
                                  void* stream) {
  if (!dpatch || !dimg || B <= 0 || C <= 0 || HW <= 0 || P <= 0 || HW % P) return FAVIT_ERR_INVALID;
  const long total = (long)B * C * HW * HW;
  hipLaunchKernelGGL(patchify_bwd_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, as_stream(stream), dpatch, dimg, B, C, HW, P);
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

extern "C" int favit_embed_prologue_fwd(const float* tok, const float* cls, const float* pos, float* x, int32_t B,
                                        int32_t N, int32_t D, void* stream) {
  if (!tok || !cls || !x || B <= 0 || N <= 0 || D <= 0) return FAVIT_ERR_INVALID;
  if (D & 3) return FAVIT_ERR_ALIGN;
  const long total = (long)B * (N + 1) * (D / 4);
  hipLaunchKernelGGL(embed_prologue_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), tok, cls, pos, x, B, N, D);
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

extern "C" int favit_embed_prologue_bwd(const float* dx, void* dtok, int dtok_dtype, float* dcls, float* dpos,
                                        int32_t B, int32_t N, int32_t D, void* stream) {
  if (!dx || B <= 0 || N <= 0 || D <= 0) return FAVIT_ERR_INVALID;
  const long total = (long)(N + 1) * D;
  hipStream_t st = as_stream(stream);
  if ((D & 3) == 0 && B >= 32 && (dtok_dtype == FAVIT_F32 || dtok_dtype == FAVIT_BF16) &&
      (reinterpret_cast<uintptr_t>(dx) & 15) == 0 && (!dtok || (reinterpret_cast<uintptr_t>(dtok) & 15) == 0)) {
    // large batches: split the batch over grid.y (the one-thread-per-element kernel below walks all B
    // images serially with 4-byte accesses: 119 us at B = 256, L = 197, D = 384)
    if (dpos) (void)favit_zero_async(dpos, sizeof(float) * total, st);
    if (dcls) (void)favit_zero_async(dcls, sizeof(float) * D, st);
    const int chunks = B >= 128 ? 16 : 4, bchunk = (B + chunks - 1) / chunks;
    const dim3 grid((unsigned)((total / 4 + 255) / 256), (unsigned)chunks);
    if (dtok_dtype == FAVIT_F32)
      hipLaunchKernelGGL((embed_prologue_bwd_chunk_kernel<float>), grid, dim3(256), 0, st, dx, (float*)dtok, dcls, dpos, B, N, D, bchunk);
    else
      hipLaunchKernelGGL((embed_prologue_bwd_chunk_kernel<bf16_t>), grid, dim3(256), 0, st, dx, (bf16_t*)dtok, dcls, dpos, B, N, D, bchunk);
    FAVIT_CHECK_LAUNCH();
    return FAVIT_OK;
  }
  if (dtok_dtype == FAVIT_F32)
    hipLaunchKernelGGL((embed_prologue_bwd_kernel<float>), dim3(grid_for(total)), dim3(256), 0, st, dx, (float*)dtok, dcls, dpos, B, N, D);
  else if (dtok_dtype == FAVIT_BF16)
    hipLaunchKernelGGL((embed_prologue_bwd_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, st, dx, (bf16_t*)dtok, dcls, dpos, B, N, D);
  else
    return FAVIT_ERR_INVALID;
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

extern "C" int favit_cross_entropy(const float* logits, const int64_t* labels, float* loss_rows, float* dlogits,
                                   int32_t B, int32_t C, float grad_scale, void* stream) {
  if (!logits || !labels || !loss_rows || B <= 0 || C <= 0) return FAVIT_ERR_INVALID;
  hipLaunchKernelGGL(cross_entropy_kernel, dim3((B + 3) / 4), dim3(256), 0, as_stream(stream), logits, labels, loss_rows, dlogits, B, C, grad_scale, favit_health_ptr_());
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

extern "C" int favit_adamw(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr,
                           float beta1, float beta2, float eps, float weight_decay, float bias_c1, float bias_c2,
                           float grad_scale, void* stream) {
  if (!p || !g || !m || !v || n < 0) return FAVIT_ERR_INVALID;
  if (n == 0) return FAVIT_OK;
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), p, g, m, v, (bf16_t*)p_bf16, (long)n, lr, beta1, beta2, eps, weight_decay, bias_c1, bias_c2, grad_scale, favit_health_ptr_());
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

extern "C" int favit_abi_version(void) { return FAVIT_ABI_VERSION; }

// process-wide dropout epoch word (device pointer or null), read by every launcher of a dropout-drawing kernel
static const unsigned long long* g_drop_epoch = nullptr;
extern "C" const unsigned long long* favit_dropout_epoch_ptr_(void) { return g_drop_epoch; }
extern "C" int favit_set_dropout_epoch(const uint64_t* device_word) {
  if (reinterpret_cast<uintptr_t>(device_word) & 7) return FAVIT_ERR_ALIGN;
  g_drop_epoch = reinterpret_cast<const unsigned long long*>(device_word);
  return FAVIT_OK;
}

// process-wide health word (4 x uint32 on the device, or null): see adamw_kernel
static unsigned* g_health = nullptr;
extern "C" unsigned* favit_health_ptr_(void) { return g_health; }
extern "C" int favit_set_health_word(uint32_t* device_words) {
  if (reinterpret_cast<uintptr_t>(device_words) & 15) return FAVIT_ERR_ALIGN;
  g_health = device_words;
  return FAVIT_OK;
}

extern "C" const char* favit_strerror(int code) {
  switch (code) {
    case FAVIT_OK: return "ok";
    case FAVIT_ERR_INVALID: return "invalid argument";
    case FAVIT_ERR_UNSUPPORTED: return "unsupported shape/dtype combination";
    case FAVIT_ERR_ALIGN: return "pointer or leading-dimension alignment requirement violated";
    case FAVIT_ERR_LAUNCH: return "HIP kernel launch failed";
    default: return "unknown favit error";
  }
}
