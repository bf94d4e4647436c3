// MFMA GEMM with fused epilogue for gfx950 (see include/favit.h: favit_gemm).
//
//   C[m,n] = epilogue( alpha * sum_k A[m,k] * B[n,k] )
//
// One 256-thread workgroup (4 waves, 2x2) owns a 128x128 output tile; each wave owns a
// 64x64 sub-tile.  Operands are staged global -> registers -> LDS (double buffered, one
// barrier per K-step) and consumed from LDS as MFMA fragments:
//   bf16 : v_mfma_f32_16x16x32_bf16, BK = 64.
//          k-major operand  -> LDS image [row][64] (128-B rows), 16-B chunk XOR swizzle,
//                              fragment = one ds_read_b128
//          mn-major operand -> LDS image [k][128] (256-B rows), 32-B XOR swizzle,
//                              fragment = two ds_read_b64_tr_b16 (hardware transpose)
//   f32  : v_mfma_f32_32x32x2_f32 (exact fp32 fma chain), BK = 16, LDS image [k][132]
//          for both layouts, fragment = ds_read_b32.
// The MFMA is issued with the B-side fragment as the "A" operand so that every lane ends
// up with 4 consecutive n for one m: the accumulators go to LDS as float4 and the
// epilogue (bias / GELU / dGELU / residual / atomics) runs on full coalesced rows.
#include "common.h"

namespace {

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int NTHREADS = 256;
constexpr int EPI_LD = 128;                          // fp32 epilogue image [128][128], 16-B slots XOR-swizzled by row
constexpr int EPI_BYTES = BM * EPI_LD * 4;           // 65536
constexpr int BK16 = 64;                             // bf16 K step
constexpr int OP16_BYTES = 128 * 64 * 2;             // one bf16 operand tile (either image) = 16 KiB
constexpr int BK32 = 16;                             // f32 K step
constexpr int F32_LD = 132;
constexpr int OP32_BYTES = BK32 * F32_LD * 4;        // 8448
constexpr int LDS_BYTES = EPI_BYTES;                 // >= 4*OP16_BYTES (65536) and 4*OP32_BYTES

struct KParams {
  const void* A;
  const void* B;
  void* C;
  const float* bias;
  const void* aux_in;
  void* aux_out;
  const float* residual;
  float* a_rowsum;
  long M, N, K;
  long lda, ldb, ldc, ld_aux_in, ld_aux_out, ld_res;
  long sAo, sAi, sBo, sBi, sCo, sCi;
  long k_per_split;
  int batch_inner;
  int act;
  int atomic;
  int a_vec, b_vec, c_vec;
  int tiles_n;
  float alpha;
  uint32_t drop_thresh;
  float drop_scale;
  uint64_t drop_seed;
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  // blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a contiguous
  // run of tile ids so that tiles sharing an A panel hit the same L2 (speed only).
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// --------------------------------------------------------------------------------------
// bf16 operand staging
// --------------------------------------------------------------------------------------
__device__ __forceinline__ int hsw(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

template <bool KMAJOR>
__device__ __forceinline__ void stage_load16(const bf16_t* __restrict__ base, long ld, long i0, long I, long k0,
                                             long kend, bool vec, uint4 (&regs)[4], int tid) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int c = tid + NTHREADS * p;
    long i, k;
    const bf16_t* ptr;
    bool full, any;
    if (KMAJOR) {
      i = i0 + (c >> 3);
      k = k0 + (c & 7) * 8;
      ptr = base + i * ld + k;
      any = (i < I) && (k < kend);
      full = any && (k + 8 <= kend) && vec;
    } else {
      k = k0 + (c >> 4);
      i = i0 + (c & 15) * 8;
      ptr = base + k * ld + i;
      any = (k < kend) && (i < I);
      full = any && (i + 8 <= I) && vec;
    }
    uint4 v = make_uint4(0, 0, 0, 0);
    if (full) {
      v = *reinterpret_cast<const uint4*>(ptr);
    } else if (any) {
      const long lim = KMAJOR ? (kend - k) : (I - i);
      unsigned short e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = (j < lim) ? reinterpret_cast<const unsigned short*>(ptr)[j] : (unsigned short)0;
      v.x = e[0] | ((unsigned)e[1] << 16);
      v.y = e[2] | ((unsigned)e[3] << 16);
      v.z = e[4] | ((unsigned)e[5] << 16);
      v.w = e[6] | ((unsigned)e[7] << 16);
    }
    regs[p] = v;
  }
}

template <bool KMAJOR>
__device__ __forceinline__ void stage_store16(char* lds, const uint4 (&regs)[4], int tid) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int c = tid + NTHREADS * p;
    int byte;
    if (KMAJOR) {
      const int row = c >> 3, kc = c & 7;
      byte = row * 128 + ((kc ^ ((row >> 1) & 7)) << 4);
    } else {
      const int krow = c >> 4, ic = c & 15;
      byte = krow * 256 + ((ic ^ (hsw(krow) << 1)) << 4);
    }
    *reinterpret_cast<uint4*>(lds + byte) = regs[p];
  }
}

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

// fragment for the 16 rows starting at r0, k-substep ks (32 wide)
template <bool KMAJOR>
__device__ __forceinline__ bf16x8 load_frag16(const char* lds, int r0, int ks, int lane) {
  if (KMAJOR) {
    const int row = r0 + (lane & 15);
    const int kc = ks * 4 + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
  } else {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int krow = ks * 32 + 8 * g + q;
    const int byte = krow * 256 + ((((r0 + 4 * p) * 2)) ^ (hsw(krow) << 5));
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + byte));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + byte + 4 * 256));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(bf16x8, r);
  }
}

// --------------------------------------------------------------------------------------
// epilogue (shared): accumulators are already in the fp32 LDS image epi[128][128]; the
// 16-B slot index of a row is XORed with (row & 7) so that the accumulator float4 writes
// (8 consecutive rows, same column) and the row reads are both bank-conflict free.
// --------------------------------------------------------------------------------------
__device__ __forceinline__ int epi_off(int row, int col) {
  return row * EPI_LD + ((((col >> 2) ^ (row & 7)) << 2) | (col & 3));
}
template <typename InT, typename OutT>
__device__ __forceinline__ void run_epilogue(const KParams& p, const float* epi, long m0, long n0, OutT* C,
                                             bool first_split, int tid) {
  const bool full_tile = (m0 + BM <= p.M) && (n0 + BN <= p.N);
  if (p.c_vec && !p.atomic && full_tile) {
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int c = tid + NTHREADS * it;
      const int row = c >> 5, c4 = (c & 31) * 4;
      const long m = m0 + row, n = n0 + c4;
      float4 v = *reinterpret_cast<const float4*>(epi + epi_off(row, c4));
      float a[4] = {v.x * p.alpha, v.y * p.alpha, v.z * p.alpha, v.w * p.alpha};
      if (p.bias) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
        a[0] += b.x; a[1] += b.y; a[2] += b.z; a[3] += b.w;
      }
      if (p.aux_out) {
        OutT* ao = reinterpret_cast<OutT*>(p.aux_out) + m * p.ld_aux_out + n;
        if constexpr (sizeof(OutT) == 4) {
          *reinterpret_cast<float4*>(ao) = make_float4(a[0], a[1], a[2], a[3]);
        } else {
          bf16x4 o = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3]};
          *reinterpret_cast<bf16x4*>(ao) = o;
        }
      }
      if (p.act == FAVIT_ACT_GELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = gelu_f(a[j]);
      } else if (p.act == FAVIT_ACT_DGELU) {
        const InT* ai = reinterpret_cast<const InT*>(p.aux_in) + m * p.ld_aux_in + n;
        float x[4];
        if constexpr (sizeof(InT) == 4) {
          const float4 t = *reinterpret_cast<const float4*>(ai);
          x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
        } else {
          const bf16x4 t = *reinterpret_cast<const bf16x4*>(ai);
          x[0] = (float)t[0]; x[1] = (float)t[1]; x[2] = (float)t[2]; x[3] = (float)t[3];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] *= dgelu_f(x[j]);
      }
      if (p.drop_thresh) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          a[j] = favit_keep(p.drop_seed, (uint64_t)(m * p.N + n + j), p.drop_thresh) ? a[j] * p.drop_scale : 0.f;
      }
      if (p.residual) {
        const float4 r = *reinterpret_cast<const float4*>(p.residual + m * p.ld_res + n);
        a[0] += r.x; a[1] += r.y; a[2] += r.z; a[3] += r.w;
      }
      OutT* co = C + m * p.ldc + n;
      if constexpr (sizeof(OutT) == 4) {
        *reinterpret_cast<float4*>(co) = make_float4(a[0], a[1], a[2], a[3]);
      } else {
        bf16x4 o = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3]};
        *reinterpret_cast<bf16x4*>(co) = o;
      }
    }
  } else {
    // scalar path: ragged tiles, unaligned leading dims, or fp32 atomics (split-K /
    // accumulate).  Lanes walk one row contiguously: an atomic wave-instruction adds
    // 256 contiguous bytes.
    for (int it = 0; it < 64; ++it) {
      const int c = tid + NTHREADS * it;
      const int row = c >> 7, col = c & 127;
      const long m = m0 + row, n = n0 + col;
      if (m >= p.M || n >= p.N) continue;
      float v = epi[epi_off(row, col)] * p.alpha;
      if (first_split && p.bias) v += p.bias[n];
      if (p.aux_out) reinterpret_cast<OutT*>(p.aux_out)[m * p.ld_aux_out + n] = from_f32<OutT>(v);
      if (p.act == FAVIT_ACT_GELU) v = gelu_f(v);
      else if (p.act == FAVIT_ACT_DGELU)
        v *= dgelu_f(to_f32(reinterpret_cast<const InT*>(p.aux_in)[m * p.ld_aux_in + n]));
      if (p.drop_thresh) v = favit_keep(p.drop_seed, (uint64_t)(m * p.N + n), p.drop_thresh) ? v * p.drop_scale : 0.f;
      if (first_split && p.residual) v += p.residual[m * p.ld_res + n];
      if (p.atomic) {
        if constexpr (sizeof(OutT) == 4) atomicAdd(reinterpret_cast<float*>(C) + m * p.ldc + n, v);
      } else {
        C[m * p.ldc + n] = from_f32<OutT>(v);
      }
    }
  }
}

// --------------------------------------------------------------------------------------
// bf16 kernel
// --------------------------------------------------------------------------------------
template <bool AK, bool BKM, typename OutT>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_kernel(KParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const long m0 = (long)(tile / p.tiles_n) * BM;
  const long n0 = (long)(tile % p.tiles_n) * BN;
  const int z = blockIdx.z;
  const long zo = z / p.batch_inner, zi = z % p.batch_inner;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A) + zo * p.sAo + zi * p.sAi;
  const bf16_t* Bm = reinterpret_cast<const bf16_t*>(p.B) + zo * p.sBo + zi * p.sBi;
  OutT* C = reinterpret_cast<OutT*>(p.C) + zo * p.sCo + zi * p.sCi;

  const long kbeg = (long)blockIdx.y * p.k_per_split;
  const long kend = min(p.K, kbeg + p.k_per_split);
  const int nk = (int)((kend - kbeg + BK16 - 1) / BK16);

  auto ldsA = [&](int b) { return smem + (2 * b) * OP16_BYTES; };
  auto ldsB = [&](int b) { return smem + (2 * b + 1) * OP16_BYTES; };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fused bias-gradient: row sums of the (mn-major) A operand
  const bool do_rowsum = (!AK) && (p.a_rowsum != nullptr) && (n0 == 0);
  float rs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  uint4 ra[4], rb[4];
  if (nk > 0) {
    stage_load16<AK>(A, p.lda, m0, p.M, kbeg, kend, p.a_vec, ra, tid);
    stage_load16<BKM>(Bm, p.ldb, n0, p.N, kbeg, kend, p.b_vec, rb, tid);
    stage_store16<AK>(ldsA(0), ra, tid);
    stage_store16<BKM>(ldsB(0), rb, tid);
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (do_rowsum) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned w[4] = {ra[q].x, ra[q].y, ra[q].z, ra[q].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          rs[2 * j] += bf16_bits_to_f32((unsigned short)(w[j] & 0xffff));
          rs[2 * j + 1] += bf16_bits_to_f32((unsigned short)(w[j] >> 16));
        }
      }
    }
    if (kt + 1 < nk) {
      const long k0 = kbeg + (long)(kt + 1) * BK16;
      stage_load16<AK>(A, p.lda, m0, p.M, k0, kend, p.a_vec, ra, tid);
      stage_load16<BKM>(Bm, p.ldb, n0, p.N, k0, kend, p.b_vec, rb, tid);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = load_frag16<AK>(ldsA(cur), wr * 64 + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = load_frag16<BKM>(ldsB(cur), wc * 64 + j * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      stage_store16<AK>(ldsA(cur ^ 1), ra, tid);
      stage_store16<BKM>(ldsB(cur ^ 1), rb, tid);
    }
    __syncthreads();
  }

  // accumulators -> fp32 LDS image (the staging buffers are dead after the last barrier)
  float* epi = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = wr * 64 + i * 16 + (lane & 15);
      const int n = wc * 64 + j * 16 + 4 * (lane >> 4);
      *reinterpret_cast<f32x4*>(epi + epi_off(m, n)) = acc[i][j];
    }
  __syncthreads();
  run_epilogue<bf16_t, OutT>(p, epi, m0, n0, C, blockIdx.y == 0, tid);

  if (do_rowsum) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);   // [16][128]
    const int ic = tid & 15, part = tid >> 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) red[part * 128 + ic * 8 + j] = rs[j];
    __syncthreads();
    if (tid < 128) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) s += red[q * 128 + tid];
      const long m = m0 + tid;
      if (m < p.M) atomicAdd(p.a_rowsum + m, s);
    }
  }
}

// --------------------------------------------------------------------------------------
// f32 kernel (exact fp32: v_mfma_f32_32x32x2_f32)
// --------------------------------------------------------------------------------------
template <bool KMAJOR>
__device__ __forceinline__ void stage_load32(const float* __restrict__ base, long ld, long i0, long I, long k0,
                                             long kend, bool vec, float4 (&regs)[2], int tid) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int c = tid + NTHREADS * p;
    long i, k;
    const float* ptr;
    bool full, any;
    long lim;
    if (KMAJOR) {
      i = i0 + (c >> 2);
      k = k0 + (c & 3) * 4;
      ptr = base + i * ld + k;
      any = (i < I) && (k < kend);
      full = any && (k + 4 <= kend) && vec;
      lim = kend - k;
    } else {
      k = k0 + (c >> 5);
      i = i0 + (c & 31) * 4;
      ptr = base + k * ld + i;
      any = (k < kend) && (i < I);
      full = any && (i + 4 <= I) && vec;
      lim = I - i;
    }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (full) {
      v = *reinterpret_cast<const float4*>(ptr);
    } else if (any) {
      v.x = ptr[0];
      v.y = (1 < lim) ? ptr[1] : 0.f;
      v.z = (2 < lim) ? ptr[2] : 0.f;
      v.w = (3 < lim) ? ptr[3] : 0.f;
    }
    regs[p] = v;
  }
}

template <bool KMAJOR>
__device__ __forceinline__ void stage_store32(float* lds, const float4 (&regs)[2], int tid) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int c = tid + NTHREADS * p;
    if (KMAJOR) {
      const int row = c >> 2, kc = (c & 3) * 4;
      lds[(kc + 0) * F32_LD + row] = regs[p].x;
      lds[(kc + 1) * F32_LD + row] = regs[p].y;
      lds[(kc + 2) * F32_LD + row] = regs[p].z;
      lds[(kc + 3) * F32_LD + row] = regs[p].w;
    } else {
      const int krow = c >> 5, ic = (c & 31) * 4;
      *reinterpret_cast<float4*>(lds + krow * F32_LD + ic) = regs[p];
    }
  }
}

template <bool AK, bool BKM>
__global__ __launch_bounds__(NTHREADS) void gemm_f32_kernel(KParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const long m0 = (long)(tile / p.tiles_n) * BM;
  const long n0 = (long)(tile % p.tiles_n) * BN;
  const int z = blockIdx.z;
  const long zo = z / p.batch_inner, zi = z % p.batch_inner;
  const float* A = reinterpret_cast<const float*>(p.A) + zo * p.sAo + zi * p.sAi;
  const float* Bm = reinterpret_cast<const float*>(p.B) + zo * p.sBo + zi * p.sBi;
  float* C = reinterpret_cast<float*>(p.C) + zo * p.sCo + zi * p.sCi;

  const long kbeg = (long)blockIdx.y * p.k_per_split;
  const long kend = min(p.K, kbeg + p.k_per_split);
  const int nk = (int)((kend - kbeg + BK32 - 1) / BK32);

  auto ldsA = [&](int b) { return reinterpret_cast<float*>(smem + (2 * b) * OP32_BYTES); };
  auto ldsB = [&](int b) { return reinterpret_cast<float*>(smem + (2 * b + 1) * OP32_BYTES); };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const bool do_rowsum = (!AK) && (p.a_rowsum != nullptr) && (n0 == 0);
  float rs[4] = {0.f, 0.f, 0.f, 0.f};

  float4 ra[2], rb[2];
  if (nk > 0) {
    stage_load32<AK>(A, p.lda, m0, p.M, kbeg, kend, p.a_vec, ra, tid);
    stage_load32<BKM>(Bm, p.ldb, n0, p.N, kbeg, kend, p.b_vec, rb, tid);
    stage_store32<AK>(ldsA(0), ra, tid);
    stage_store32<BKM>(ldsB(0), rb, tid);
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (do_rowsum) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        rs[0] += ra[q].x; rs[1] += ra[q].y; rs[2] += ra[q].z; rs[3] += ra[q].w;
      }
    }
    if (kt + 1 < nk) {
      const long k0 = kbeg + (long)(kt + 1) * BK32;
      stage_load32<AK>(A, p.lda, m0, p.M, k0, kend, p.a_vec, ra, tid);
      stage_load32<BKM>(Bm, p.ldb, n0, p.N, k0, kend, p.b_vec, rb, tid);
    }
    const float* la = ldsA(cur) + wr * 64 + (lane & 31);
    const float* lb = ldsB(cur) + wc * 64 + (lane & 31);
#pragma unroll
    for (int s = 0; s < BK32 / 2; ++s) {
      const int kk = 2 * s + (lane >> 5);
      const float a0 = la[kk * F32_LD], a1 = la[kk * F32_LD + 32];
      const float b0 = lb[kk * F32_LD], b1 = lb[kk * F32_LD + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, a0, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a1, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, a1, acc[1][1], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      stage_store32<AK>(ldsA(cur ^ 1), ra, tid);
      stage_store32<BKM>(ldsB(cur ^ 1), rb, tid);
    }
    __syncthreads();
  }

  float* epi = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = wr * 64 + i * 32 + (lane & 31);
      const int nb = wc * 64 + j * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        f32x4 v = {acc[i][j][4 * rg], acc[i][j][4 * rg + 1], acc[i][j][4 * rg + 2], acc[i][j][4 * rg + 3]};
        *reinterpret_cast<f32x4*>(epi + epi_off(m, nb + 8 * rg)) = v;
      }
    }
  __syncthreads();
  run_epilogue<float, float>(p, epi, m0, n0, C, blockIdx.y == 0, tid);

  if (do_rowsum) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);   // [8][128]
    const int ic = tid & 31, part = tid >> 5;
#pragma unroll
    for (int j = 0; j < 4; ++j) red[part * 128 + ic * 4 + j] = rs[j];
    __syncthreads();
    if (tid < 128) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) s += red[q * 128 + tid];
      const long m = m0 + tid;
      if (m < p.M) atomicAdd(p.a_rowsum + m, s);
    }
  }
}

__global__ void zero_c_kernel(float* C, long M, long N, long ldc, long sCo, long sCi, int batch_inner) {
  const long z = blockIdx.z;
  float* c = C + (z / batch_inner) * sCo + (z % batch_inner) * sCi;
  const long total = M * N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    c[(i / N) * ldc + (i % N)] = 0.f;
}

template <typename K>
int launch(K kernel, const KParams& kp, dim3 grid, hipStream_t st) {
  static bool attr_set = false;   // one flag per kernel instantiation
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL(kernel, grid, dim3(NTHREADS), LDS_BYTES, st, kp);
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace

extern "C" int favit_gemm(const favit_gemm_t* g, void* stream) {
  if (!g || !g->A || !g->B || !g->C) return FAVIT_ERR_INVALID;
  if (g->M <= 0 || g->N <= 0 || g->K < 0) return FAVIT_ERR_INVALID;
  if (g->in_dtype != FAVIT_F32 && g->in_dtype != FAVIT_BF16) return FAVIT_ERR_INVALID;
  if (g->out_dtype != FAVIT_F32 && g->out_dtype != FAVIT_BF16) return FAVIT_ERR_INVALID;
  if (g->in_dtype == FAVIT_F32 && g->out_dtype != FAVIT_F32) return FAVIT_ERR_UNSUPPORTED;
  if (g->act == FAVIT_ACT_DGELU && !g->aux_in) return FAVIT_ERR_INVALID;
  if (g->a_rowsum && g->a_kmajor) return FAVIT_ERR_UNSUPPORTED;
  const int batch = g->batch > 0 ? g->batch : 1;
  const int batch_inner = g->batch_inner > 0 ? g->batch_inner : 1;
  if (g->a_rowsum && batch != 1) return FAVIT_ERR_UNSUPPORTED;
  hipStream_t st = as_stream(stream);

  const int bk = g->in_dtype == FAVIT_BF16 ? BK16 : BK32;
  const long tiles_m = (g->M + BM - 1) / BM, tiles_n = (g->N + BN - 1) / BN;
  const long tiles = tiles_m * tiles_n * batch;
  long splits = g->split_k;
  const bool can_split = (g->out_dtype == FAVIT_F32) && g->act == FAVIT_ACT_NONE && !g->aux_out &&
                         !(g->dropout_p > 0.f);
  if (splits <= 0) {
    splits = 1;
    // automatic split-K only for the weight-gradient shape (both operands mn-major, K = tokens):
    // forward / input-gradient GEMMs stay single-pass and therefore bitwise deterministic.
    if (can_split && !g->a_kmajor && !g->b_kmajor && tiles < 256 && g->K >= 8 * bk) {
      splits = (512 + tiles - 1) / tiles;
      const long max_splits = g->K / (4 * bk);
      if (splits > max_splits) splits = max_splits;
      if (splits < 1) splits = 1;
    }
  }
  if (splits > 1 && !can_split) return FAVIT_ERR_UNSUPPORTED;
  long kps = (g->K + splits - 1) / splits;
  kps = ((kps + bk - 1) / bk) * bk;
  if (kps <= 0) kps = bk;
  splits = (g->K + kps - 1) / kps;
  if (splits < 1) splits = 1;
  const int atomic = (splits > 1 || g->accumulate) ? 1 : 0;
  if (atomic && g->out_dtype != FAVIT_F32) return FAVIT_ERR_UNSUPPORTED;

  KParams kp;
  kp.A = g->A; kp.B = g->B; kp.C = g->C;
  kp.bias = g->bias; kp.aux_in = g->aux_in; kp.aux_out = g->aux_out; kp.residual = g->residual;
  kp.a_rowsum = g->a_rowsum;
  kp.M = g->M; kp.N = g->N; kp.K = g->K;
  kp.lda = g->lda; kp.ldb = g->ldb; kp.ldc = g->ldc;
  kp.ld_aux_in = g->ld_aux_in; kp.ld_aux_out = g->ld_aux_out; kp.ld_res = g->ld_res;
  kp.sAo = g->sAo; kp.sAi = g->sAi; kp.sBo = g->sBo; kp.sBi = g->sBi; kp.sCo = g->sCo; kp.sCi = g->sCi;
  kp.k_per_split = kps;
  kp.batch_inner = batch_inner;
  kp.act = g->act;
  kp.atomic = atomic;
  kp.tiles_n = (int)tiles_n;
  kp.alpha = g->alpha;
  if (g->dropout_p < 0.f || g->dropout_p >= 1.f) return FAVIT_ERR_INVALID;
  kp.drop_thresh = dropout_threshold(g->dropout_p);
  kp.drop_scale = 1.0f / (1.0f - g->dropout_p);
  kp.drop_seed = g->dropout_seed;
  if (kp.drop_thresh && (splits > 1 || batch != 1)) return FAVIT_ERR_UNSUPPORTED;

  const int in_vec = g->in_dtype == FAVIT_BF16 ? 8 : 4;   // elements per 16-B load
  auto strides_ok = [&](long so, long si, int v) { return batch == 1 || ((so % v) == 0 && (si % v) == 0); };
  kp.a_vec = aligned(g->A, 16) && (g->lda % in_vec) == 0 && strides_ok(g->sAo, g->sAi, in_vec);
  kp.b_vec = aligned(g->B, 16) && (g->ldb % in_vec) == 0 && strides_ok(g->sBo, g->sBi, in_vec);
  const size_t osz = g->out_dtype == FAVIT_BF16 ? 2 : 4;
  const size_t isz = g->in_dtype == FAVIT_BF16 ? 2 : 4;
  bool cv = aligned(g->C, 4 * osz) && (g->ldc % 4) == 0 && strides_ok(g->sCo, g->sCi, 4);
  if (g->bias) cv = cv && aligned(g->bias, 16);
  if (g->aux_out) cv = cv && aligned(g->aux_out, 4 * osz) && (g->ld_aux_out % 4) == 0 && batch == 1;
  if (g->aux_in) cv = cv && aligned(g->aux_in, 4 * isz) && (g->ld_aux_in % 4) == 0 && batch == 1;
  if (g->residual) cv = cv && aligned(g->residual, 16) && (g->ld_res % 4) == 0 && batch == 1;
  if ((g->aux_out || g->aux_in || g->residual) && batch != 1) return FAVIT_ERR_UNSUPPORTED;
  kp.c_vec = cv ? 1 : 0;

  if (splits > 1 && !g->accumulate) {
    const long total = g->M * g->N;
    const int zb = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(zero_c_kernel, dim3(zb, 1, batch), dim3(256), 0, st, reinterpret_cast<float*>(g->C), g->M,
                       g->N, g->ldc, g->sCo, g->sCi, batch_inner);
    FAVIT_CHECK_LAUNCH();
  }

  dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)splits, (unsigned)batch);
  const int layout = (g->a_kmajor ? 2 : 0) | (g->b_kmajor ? 1 : 0);
  if (g->in_dtype == FAVIT_BF16) {
    if (g->out_dtype == FAVIT_BF16) {
      switch (layout) {
        case 3: return launch(gemm_bf16_kernel<true, true, bf16_t>, kp, grid, st);
        case 2: return launch(gemm_bf16_kernel<true, false, bf16_t>, kp, grid, st);
        case 1: return launch(gemm_bf16_kernel<false, true, bf16_t>, kp, grid, st);
        default: return launch(gemm_bf16_kernel<false, false, bf16_t>, kp, grid, st);
      }
    } else {
      switch (layout) {
        case 3: return launch(gemm_bf16_kernel<true, true, float>, kp, grid, st);
        case 2: return launch(gemm_bf16_kernel<true, false, float>, kp, grid, st);
        case 1: return launch(gemm_bf16_kernel<false, true, float>, kp, grid, st);
        default: return launch(gemm_bf16_kernel<false, false, float>, kp, grid, st);
      }
    }
  } else {
    switch (layout) {
      case 3: return launch(gemm_f32_kernel<true, true>, kp, grid, st);
      case 2: return launch(gemm_f32_kernel<true, false>, kp, grid, st);
      case 1: return launch(gemm_f32_kernel<false, true>, kp, grid, st);
      default: return launch(gemm_f32_kernel<false, false>, kp, grid, st);
    }
  }
}
