// MFMA GEMM with fused epilogue for gfx950 (see include/favit.h: favit_gemm).
//
//   C[m,n] = epilogue( alpha * sum_k A[m,k] * B[n,k] )
//
// The bf16 kernels share the fragment / epilogue code below:
//   * "p4" (the hot one): 256x128 tile, 8 waves, BK = 32, three LDS stages filled by
//     global_load_lds with a counted vmcnt, wave-private transpose epilogue, 2 workgroups per CU;
//     also runs the grouped weight-gradient launch (chunks / K-splits / per-XCD queues) and the fp8 operands
//   * "p7" 256x256 (K >= 512, N % 256 == 0), "pp" ping-pong (K >= 4096), "s64" / "s64k2" 64-row tiles for
//     launches with fewer tiles than CUs, "s64ln" (LayerNorm fused, off), "pd" (persistent, deferred epilogue: off)
//   * "glds": 128x128 tile, direct-to-LDS double buffer (small and batched problems)
//   * the register-staged 128x128 kernel (any shape / alignment; the fallback)
// and two exact-fp32 kernels (the parity mode): "p4f", the p4 structure with v_mfma_f32_32x32x2_f32 for the large
// problems, and the register-staged 128x128 one.  In the 128x128 kernels a 256-thread workgroup
// (4 waves, 2x2) owns the tile, each wave a 64x64 sub-tile; operands are consumed from LDS as
// MFMA fragments:
//   bf16 : v_mfma_f32_16x16x32_bf16, BK = 64.
//          k-major operand  -> LDS image [row][64] (128-B rows), 16-B chunk XOR swizzle,
//                              fragment = one ds_read_b128
//          mn-major operand -> LDS image [k][128] (256-B rows), 32-B XOR swizzle,
//                              fragment = two ds_read_b64_tr_b16 (hardware transpose)
//   f32  : v_mfma_f32_32x32x2_f32 (exact fp32 fma chain), BK = 16, LDS image [k][132]
//          for both layouts, fragment = ds_read_b32 (p4f: see its own header).
// The MFMA is issued with the B-side fragment as the "A" operand so that every lane ends
// up with 4 consecutive n for one m: the accumulators go to LDS as float4 and the
// epilogue (bias / GELU / dGELU / residual / atomics) runs on full coalesced rows.
#include "common.h"
#include <type_traits>
#include <stdlib.h>
#include <string.h>

namespace {
// name of the kernel family the last favit_gemm / grouped launch of this host thread dispatched to
// (favit_gemm_last_kernel: tests and bench.py read it to assert / report which kernel really ran)
thread_local const char* g_last_kernel = "none";
thread_local int g_last_grouped_splits = 0;   // K-splits the last grouped weight-gradient launch of this thread used

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
  const unsigned long long* drop_epoch;   // favit_set_dropout_epoch word (or null): mixed into drop_seed in the kernel
  int ntiles;          // output tiles per (split, batch)
  int xcd_split;       // 1: 1-D grid of ntiles*nsplit blocks, all tiles of a split on one XCD
  int store_policy;    // cache policy of the epilogue's output stores (store16_policy)
  int rowsum_store;    // 1: a_rowsum is a private slab slot of this split (plain store), 0: atomicAdd
  int q_block0, q_tile0;   // p4: blocks >= q_block0 (> 0) run 64x128 quarter tiles of the full tiles from q_tile0 on
#ifdef FAVIT_PROBE
  int dbg;     // probe build only (make probe; tools/): 1 = skip epilogue, 2 = skip main loop
  unsigned long long* probe;   // probe build only: per-wave cycle stamps of the pp kernel (favit_probe_buffer)
#endif
  const float* scale_a;   // fp8 operands: device dequantisation factors (or null)
  const float* scale_b;
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  // blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a contiguous
  // run of tile ids so that tiles sharing an A panel hit the same L2 (speed only).
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// (tile, split) of this workgroup.  Split-K launches use a 1-D grid in which the hardware's
// round-robin block->XCD deal (blocks b and b+8 share an XCD) is inverted so that ALL output tiles
// of one K-split run on the same XCD: they stream the same token rows of both operands, and with
// the default order every XCD fetched its own copy (measured: 29 % L2 hit rate, vs 95 % for the
// un-split kernels).  nsplit is a multiple of 8 in this mode.  Speed only, never correctness.
__device__ __forceinline__ void tile_and_split(const KParams& p, int& tile, int& split) {
  if (p.xcd_split) {
    const int h = blockIdx.x, xcd = h & 7, idx = h >> 3;
    split = xcd + 8 * (idx / p.ntiles);
    tile = idx % p.ntiles;
  } else {
    tile = xcd_remap(blockIdx.x, gridDim.x);
    split = blockIdx.y;
  }
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

// k-major fragment for the 16 rows starting at r0, k-substep ks (32 wide) of a BK = 64 image
template <bool KMAJOR>
__device__ __forceinline__ bf16x8 load_frag16(const char* lds, int r0, int ks, int lane) {
  static_assert(KMAJOR, "mn-major fragments go through load_frags4 (tr_read_pair)");
  const int row = r0 + (lane & 15);
  const int kc = ks * 4 + (lane >> 4);
  return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
}

// mn-major fragments without the compiler's LDS-DMA alias wait.  hipcc cannot tell that a
// ds_read_b64_tr_b16 (an intrinsic without memory operands) does not touch the stage a global_load_lds
// issued a moment ago is still filling, and puts `s_waitcnt vmcnt(0)` in front of the first transposed
// read of every K-step: the prefetch it was meant to overlap is drained before any MFMA issues
// (measured: NN 20 % slower than NT on identical shapes, 1.09 us per K-step in the weight-gradient
// kernel).  Issued as inline asm the reads carry no such dependency; ordering against the DMA is what
// the kernel's own counted vmcnt + barrier already establish, and their completion is waited for
// explicitly (tr_fence: every raw register pair is an in/out operand of the s_waitcnt, so no consumer
// can be scheduled above it).
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;

__device__ __forceinline__ void tr_read_pair(const char* lds, int r0, int ks, int lane, u32x2_t& lo, u32x2_t& hi) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int krow = ks * 32 + 8 * g + q;
  const int byte = krow * 256 + ((((r0 + 4 * p) * 2)) ^ (hsw(krow) << 5));
  typedef __attribute__((address_space(3))) const char* lds_cptr;
  const unsigned addr = (unsigned)(uintptr_t)(lds_cptr)(lds + byte);
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(hi) : "v"(addr) : "memory");
}

__device__ __forceinline__ void tr_fence(u32x2_t (&l)[4], u32x2_t (&h)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3]), "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3])
               :
               : "memory");
}

__device__ __forceinline__ void tr_fence2(u32x2_t (&l)[4], u32x2_t (&h)[4], u32x2_t (&l2)[4], u32x2_t (&h2)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3]), "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]),
                 "+v"(l2[0]), "+v"(l2[1]), "+v"(l2[2]), "+v"(l2[3]), "+v"(h2[0]), "+v"(h2[1]), "+v"(h2[2]), "+v"(h2[3])
               :
               : "memory");
}

__device__ __forceinline__ bf16x8 tr_pack(const u32x2_t& lo, const u32x2_t& hi) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  const u32x4_t r = {lo.x, lo.y, hi.x, hi.y};
  return __builtin_bit_cast(bf16x8, r);
}

// the 4 fragments of a 64-row operand slice (rows r0, r0+16, ...), k-substep ks of a BK = 64 or BK = 32 image
template <bool KMAJOR, bool BK32>
__device__ __forceinline__ void load_frags4(const char* lds, int r0, int ks, int lane, bf16x8 (&f)[4]) {
  if constexpr (KMAJOR) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (BK32) {
        const int row = r0 + i * 16 + (lane & 15), c = lane >> 4;
        f[i] = *reinterpret_cast<const bf16x8*>(lds + row * 64 + ((c ^ ((-(row >> 2)) & 3)) << 4));
      } else {
        const int row = r0 + i * 16 + (lane & 15), kc = ks * 4 + (lane >> 4);
        f[i] = *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
      }
    }
  } else {
    u32x2_t l[4], h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) tr_read_pair(lds, r0 + i * 16, ks, lane, l[i], h[i]);
    tr_fence(l, h);
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = tr_pack(l[i], h[i]);
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
template <typename InT> __device__ __forceinline__ float epi_gelu(float v) {
  if constexpr (sizeof(InT) == 2) return gelu_fast(v); else return gelu_f(v);
}
template <typename InT> __device__ __forceinline__ float epi_dgelu(float v) {
  if constexpr (sizeof(InT) == 2) return dgelu_fast(v); else return dgelu_f(v);
}
// GELU and its derivative together (FAVIT_ACT_GELU_SAVEGRAD): Phi and phi share the exponential
template <typename InT> __device__ __forceinline__ void epi_gelu_both(float v, float& h, float& g) {
  if constexpr (sizeof(InT) == 2) {
    float c, d;
    gelu_terms_fast(v, c, d);
    h = v * c;
    g = fmaf(v, d, c);
  } else {
    h = gelu_f(v);
    g = dgelu_f(v);
  }
}

template <typename InT, typename OutT, int TBM = BM, int NT = NTHREADS>
__device__ __forceinline__ void run_epilogue(const KParams& p, const float* epi, long m0, long n0, OutT* C,
                                             bool first_split, int tid) {
  const bool full_tile = (m0 + TBM <= p.M) && (n0 + BN <= p.N);
  if (p.c_vec && !p.atomic && full_tile) {
#pragma unroll 4
    for (int it = 0; it < TBM * BN / 4 / NT; ++it) {
      const int c = tid + NT * it;
      const int row = c >> 5, c4 = (c & 31) * 4;
      const long m = m0 + row, n = n0 + c4;
      float4 v = *reinterpret_cast<const float4*>(epi + epi_off(row, c4));
      float a[4] = {v.x * p.alpha, v.y * p.alpha, v.z * p.alpha, v.w * p.alpha};
      if (p.bias) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
        a[0] += b.x; a[1] += b.y; a[2] += b.z; a[3] += b.w;
      }
      float ax[4] = {a[0], a[1], a[2], a[3]};            // what aux_out receives: the pre-activation, or GELU'
      if (p.act == FAVIT_ACT_GELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = epi_gelu<InT>(a[j]);
      } else if (p.act == FAVIT_ACT_GELU_SAVEGRAD) {
#pragma unroll
        for (int j = 0; j < 4; ++j) epi_gelu_both<InT>(a[j], a[j], ax[j]);
      } else if (p.act == FAVIT_ACT_DGELU || p.act == FAVIT_ACT_MULAUX) {
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
        for (int j = 0; j < 4; ++j) a[j] *= (p.act == FAVIT_ACT_MULAUX) ? x[j] : epi_dgelu<InT>(x[j]);
      }
      if (p.aux_out) {
        OutT* ao = reinterpret_cast<OutT*>(p.aux_out) + m * p.ld_aux_out + n;
        if constexpr (sizeof(OutT) == 4) {
          *reinterpret_cast<float4*>(ao) = make_float4(ax[0], ax[1], ax[2], ax[3]);
        } else {
          bf16x4 o = {(bf16_t)ax[0], (bf16_t)ax[1], (bf16_t)ax[2], (bf16_t)ax[3]};
          *reinterpret_cast<bf16x4*>(ao) = o;
        }
      }
      if (p.drop_thresh) {
        const uint64_t sd = favit_eff_seed(p.drop_seed, p.drop_epoch);
        const uint64_t e0 = (uint64_t)(m * p.N + n);
        if ((e0 & 1) == 0) {                               // (always, when N is even: n is a multiple of 4)
#pragma unroll
          for (int j = 0; j < 4; j += 2) {
            bool k0, k1;
            favit_keep2(sd, e0 + j, p.drop_thresh, k0, k1);
            a[j] = k0 ? a[j] * p.drop_scale : 0.f;
            a[j + 1] = k1 ? a[j + 1] * p.drop_scale : 0.f;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) a[j] = favit_keep(sd, e0 + j, p.drop_thresh) ? a[j] * p.drop_scale : 0.f;
        }
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
    for (int it = 0; it < TBM * BN / NT; ++it) {
      const int c = tid + NT * it;
      const int row = c >> 7, col = c & 127;
      const long m = m0 + row, n = n0 + col;
      if (m >= p.M || n >= p.N) continue;
      float v = epi[epi_off(row, col)] * p.alpha;
      if (first_split && p.bias) v += p.bias[n];
      float vx = v;
      if (p.act == FAVIT_ACT_GELU) v = epi_gelu<InT>(v);
      else if (p.act == FAVIT_ACT_GELU_SAVEGRAD) epi_gelu_both<InT>(v, v, vx);
      else if (p.act == FAVIT_ACT_DGELU)
        v *= epi_dgelu<InT>(to_f32(reinterpret_cast<const InT*>(p.aux_in)[m * p.ld_aux_in + n]));
      else if (p.act == FAVIT_ACT_MULAUX)
        v *= to_f32(reinterpret_cast<const InT*>(p.aux_in)[m * p.ld_aux_in + n]);
      if (p.aux_out) reinterpret_cast<OutT*>(p.aux_out)[m * p.ld_aux_out + n] = from_f32<OutT>(vx);
      if (p.drop_thresh) v = favit_keep(favit_eff_seed(p.drop_seed, p.drop_epoch), (uint64_t)(m * p.N + n), p.drop_thresh) ? v * p.drop_scale : 0.f;
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

  int tile, split;
  tile_and_split(p, tile, split);
  const long m0 = (long)(tile / p.tiles_n) * BM;
  const long n0 = (long)(tile % p.tiles_n) * BN;
  const int z = blockIdx.z;
  const long zo = z / p.batch_inner, zi = z % p.batch_inner;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A) + zo * p.sAo + zi * p.sAi;
  const bf16_t* Bm = reinterpret_cast<const bf16_t*>(p.B) + zo * p.sBo + zi * p.sBi;
  OutT* C = reinterpret_cast<OutT*>(p.C) + zo * p.sCo + zi * p.sCi;

  const long kbeg = (long)split * p.k_per_split;
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
      load_frags4<AK, false>(ldsA(cur), wr * 64, ks, lane, af);
      load_frags4<BKM, false>(ldsB(cur), wc * 64, ks, lane, bfr);
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
  run_epilogue<bf16_t, OutT>(p, epi, m0, n0, C, split == 0, tid);

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
// bf16 kernel, direct-to-LDS staging (global_load_lds_dwordx4): no VGPR round trip and no
// ds_write pass.  A wave-instruction lands 1 KiB contiguously (wave-uniform LDS base +
// lane*16), so the LDS image stays linear and the XOR swizzle is applied to the per-lane
// SOURCE address (and again on the fragment read).  Used when every K-range is a multiple of
// 64 and operands are 16-B aligned; rows past the M/N edge are clamped (their outputs are
// never stored).  The bias-gradient row sums come from one extra MFMA per A fragment against
// an all-ones fragment.
// --------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// source pointer of this lane for 1-KiB piece q (0..15) of a 128-wide operand image
template <bool KMAJOR>
__device__ __forceinline__ const bf16_t* glds_src(const bf16_t* base, long ld, long i0, long I, long k0, int q, int lane) {
  if (KMAJOR) {
    const int row = 8 * q + (lane >> 3), pc = lane & 7;
    const int c = pc ^ ((row >> 1) & 7);
    long i = i0 + row;
    i = i < I ? i : I - 1;
    return base + i * ld + k0 + c * 8;
  } else {
    const int krow = 4 * q + (lane >> 4), pc = lane & 15;
    const int c = pc ^ (hsw(krow) << 1);
    long i = i0 + c * 8;
    const long imax = (I - 8) & ~7L;
    i = i < imax ? i : imax;
    return base + (k0 + krow) * ld + i;
  }
}

template <bool KMAJOR>
__device__ __forceinline__ void glds_setup(const bf16_t* base, long ld, long i0, long I, long k0, int wave, int lane,
                                           const bf16_t* (&src)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = wave * 4 + j;
    if (KMAJOR) {
      const int row = 8 * q + (lane >> 3), pc = lane & 7;
      const int c = pc ^ ((row >> 1) & 7);
      long i = i0 + row;
      i = i < I ? i : I - 1;
      src[j] = base + i * ld + k0 + c * 8;
    } else {
      const int krow = 4 * q + (lane >> 4), pc = lane & 15;
      const int c = pc ^ (hsw(krow) << 1);
      long i = i0 + c * 8;
      const long imax = (I - 8) & ~7L;
      i = i < imax ? i : imax;
      src[j] = base + (k0 + krow) * ld + i;
    }
  }
}

template <bool KMAJOR>
__device__ __forceinline__ void glds_issue(const bf16_t* (&src)[4], char* tile, int wave, long ld) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    __builtin_amdgcn_global_load_lds((gptr_t)src[j], (lptr_t)(tile + (wave * 4 + j) * 1024), 16, 0, 0);
    src[j] += KMAJOR ? BK16 : BK16 * ld;
  }
}

template <bool AK, bool BKM, typename OutT>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_glds_kernel(KParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  int tile, split;
  tile_and_split(p, tile, split);
  const long m0 = (long)(tile / p.tiles_n) * BM;
  const long n0 = (long)(tile % p.tiles_n) * BN;
  const int z = blockIdx.z;
  const long zo = z / p.batch_inner, zi = z % p.batch_inner;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A) + zo * p.sAo + zi * p.sAi;
  const bf16_t* Bm = reinterpret_cast<const bf16_t*>(p.B) + zo * p.sBo + zi * p.sBi;
  OutT* C = reinterpret_cast<OutT*>(p.C) + zo * p.sCo + zi * p.sCi;

  const long kbeg = (long)split * p.k_per_split;
  const long kend = min(p.K, kbeg + p.k_per_split);
  const int nk = (int)((kend - kbeg) / BK16);

  auto ldsA = [&](int b) { return smem + (2 * b) * OP16_BYTES; };
  auto ldsB = [&](int b) { return smem + (2 * b + 1) * OP16_BYTES; };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bool do_rowsum = (!AK) && (p.a_rowsum != nullptr) && (n0 == 0) && (wc == 0);
  f32x4 racc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) racc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)1.0f;

  const bf16_t* sa[4];
  const bf16_t* sb[4];
  glds_setup<AK>(A, p.lda, m0, p.M, kbeg, wave, lane, sa);
  glds_setup<BKM>(Bm, p.ldb, n0, p.N, kbeg, wave, lane, sb);
  if (nk > 0) {
    glds_issue<AK>(sa, ldsA(0), wave, p.lda);
    glds_issue<BKM>(sb, ldsB(0), wave, p.ldb);
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      glds_issue<AK>(sa, ldsA(cur ^ 1), wave, p.lda);
      glds_issue<BKM>(sb, ldsB(cur ^ 1), wave, p.ldb);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
      load_frags4<AK, false>(ldsA(cur), wr * 64, ks, lane, af);
      load_frags4<BKM, false>(ldsB(cur), wc * 64, ks, lane, bfr);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      if (do_rowsum) {
#pragma unroll
        for (int i = 0; i < 4; ++i) racc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[i], racc[i], 0, 0, 0);
      }
    }
    __syncthreads();
  }

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
  run_epilogue<bf16_t, OutT>(p, epi, m0, n0, C, split == 0, tid);

  if (do_rowsum && lane < 16) {
    // D[i][j] of the ones-MFMA = rowsum(A[m0 + wr*64 + 16*t + j]) for every i: take row 0
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long m = m0 + wr * 64 + i * 16 + lane;
      if (m < p.M) atomicAdd(p.a_rowsum + m, racc[i][0]);
    }
  }
}

// --------------------------------------------------------------------------------------
// bf16 kernel "p4": 256x128 tile, 8 waves, BK = 32, three 24-KiB LDS stages (72 KiB -> TWO
// workgroups per CU) and an epilogue that runs straight from the accumulator registers (no
// LDS staging, no barrier): one workgroup's prologue / epilogue / store drain overlaps the
// other's MFMA loop.  Measured on the K=384 shapes of this path the LDS-staged epilogue of a
// lone workgroup cost as much as the whole main loop.
//   k-major image : [rows][32 k]  (64-B rows), 16-B chunk c stored at c ^ ((-(row>>2)) & 3)
//   mn-major image: [32 k][128 i] (256-B rows), same 32-B XOR swizzle as the BK=64 image
// (Storing straight from the MFMA layout -- 32-B pieces per row -- measured 1.7 TB/s on the
// two-output fc1 epilogue; the wave-private LDS transpose below writes full 128/256-B row segments.)
// Measured alternatives (tools/dma_probe.py, tools/overlap_probe.py, round 1):
//  * the operand stream of this tile alone (no MFMA) moves 13.9 TB/s L2->LDS; with BK = 64 (whole
//    128-B lines per row and step) and three 48-KiB stages it moves 19.3 TB/s, but 144 KiB of LDS
//    leave one workgroup (2 waves/SIMD) per CU, whose LDS-read + MFMA phase then takes 1.0 us per
//    48-KiB step -- a persistent BK = 64 variant ran the main loop at 850 TF vs 950 TF here;
//  * an HBM store or load stream running beside the main loop adds its full time (concurrent
//    fill + main loop = sum of both): epilogue traffic and the L2-bound loop share the L2/fabric
//    path, so a phase stagger between the two co-resident workgroups buys nothing (measured).
// --------------------------------------------------------------------------------------
constexpr int P4_BM = 256;
constexpr int P4_BK = 32;
constexpr int P4_THREADS = 512;
constexpr int P4_A_BYTES = 256 * 64;               // 16 KiB
constexpr int P4_B_BYTES = 128 * 64;               // 8 KiB
constexpr int P4_STAGE = P4_A_BYTES + P4_B_BYTES;  // 24 KiB
constexpr int P4_LDS = 3 * P4_STAGE;               // 73728
constexpr int S64_A_BYTES = 64 * 128;                // A image of the 64-row tiles (BK = 64): 8 KiB
constexpr int S64_STAGE = S64_A_BYTES + OP16_BYTES;  // 24 KiB

__device__ __forceinline__ int ksw32(int row) { return (-(row >> 2)) & 3; }

// source pointer for 1-KiB piece q of a BK=32 image.  K-major: piece = 16 rows x 64 B (any number
// of rows); mn-major: piece = 4 k-rows x 256 B of a 128-wide sub-image (8 pieces).
template <bool KMAJOR>
__device__ __forceinline__ const bf16_t* glds_src32(const bf16_t* base, long ld, long i0, long I, long k0, int q, int lane) {
  if (KMAJOR) {
    const int row = 16 * q + (lane >> 2), pc = lane & 3;
    const int c = pc ^ ksw32(row);
    long i = i0 + row;
    i = i < I ? i : I - 1;
    return base + i * ld + k0 + c * 8;
  } else {
    const int krow = 4 * q + (lane >> 4), pc = lane & 15;
    const int c = pc ^ (hsw(krow) << 1);
    long i = i0 + c * 8;
    const long imax = (I - 8) & ~7L;
    i = i < imax ? i : imax;
    return base + (k0 + krow) * ld + i;
  }
}

// k-major fragment (16 rows from r0, all 32 k of a BK = 32 stage)
template <bool KMAJOR>
__device__ __forceinline__ bf16x8 load_frag32(const char* lds, int r0, int lane) {
  static_assert(KMAJOR, "mn-major fragments go through load_frags4 (tr_read_pair)");
  const int row = r0 + (lane & 15);
  const int c = lane >> 4;
  return *reinterpret_cast<const bf16x8*>(lds + row * 64 + ((c ^ ksw32(row)) << 4));
}

// 16-byte output store with a cache policy: 0 plain, 1 non-temporal (nt), 2 write-through (sc1).
// GEMM outputs are written once and not re-read by this kernel; keeping them out of the XCD L2
// protects the A/B operand lines that co-resident workgroups still share.
__device__ __forceinline__ void store16_policy(void* ptr, uint4 v, int policy) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const u32x4 t = {v.x, v.y, v.z, v.w};
  if (policy == 1) {
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4*>(ptr));
  } else if (policy == 2) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(t) : "memory");
  } else if (policy == 3) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(ptr), "v"(t) : "memory");
  } else if (policy == 4) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(ptr), "v"(t) : "memory");
  } else {
    *reinterpret_cast<u32x4*>(ptr) = t;
  }
}

constexpr int WEPI_LD = 68;                         // padded fp32 row
constexpr int WEPI_BYTES = 32 * WEPI_LD * 4;        // 8704 B per wave

// rows [16*Q, 16*(Q+NI)) of the wave's 64x64 accumulator tile (NI = 1 or 2 groups of 16 rows), already deposited in
// the wave's LDS scratch wl[16 * NI][WEPI_LD]: bias / activation / dropout / residual, then whole row segments out
template <typename InT, typename OutT, int Q, int NI>
__device__ __forceinline__ void wave_epilogue_tail(const KParams& p, OutT* C, long mbase, long nbase, int lane, float* wl,
                                                   bool first_split, float alpha) {
  const bool fast = p.c_vec && (nbase + 64 <= p.N);
  {
    // (same wave wrote and reads: the compiler's lgkmcnt wait orders them; no barrier needed)
    bool done = false;
    if constexpr (sizeof(OutT) == 4) {
      if (p.atomic) {
        // split-K / accumulate: fp32 atomics, one 256-B contiguous row segment per wave-instruction
        const long n = nbase + lane;
        const float bv = (first_split && p.bias && n < p.N) ? p.bias[n] : 0.f;
        for (int row = 0; row < 16 * NI; ++row) {
          const long m = mbase + Q * 16 + row;
          if (m < p.M && n < p.N) {
            float v = fmaf(wl[row * WEPI_LD + lane], alpha, bv);
            if (first_split && p.residual) v += p.residual[m * p.ld_res + n];
            atomicAdd(reinterpret_cast<float*>(C) + m * p.ldc + n, v);
          }
        }
        done = true;
      }
    }
    if (!done) {
    constexpr int CPL = sizeof(OutT) == 2 ? 8 : 4;          // columns per lane (16 B of output)
    constexpr int LPR = 64 / CPL;                           // lanes per row
    constexpr int RPI = 64 / LPR;                           // rows per iteration
    constexpr int NIT = 16 * NI / RPI;
    const int lr = lane / LPR, lc = (lane % LPR) * CPL;
    const long n = nbase + lc;
    if (fast) {
      // The loads of a row group (residual rows, saved pre-activations) are issued one group AHEAD of the
      // stores: the compiler cannot move a load above a store that might alias it, and a load placed after a
      // store waits (vmcnt is in-order and counts stores) for that store's whole round trip to HBM.
      float4 bv[CPL / 4];
#pragma unroll
      for (int c4 = 0; c4 < CPL / 4; ++c4)
        bv[c4] = p.bias ? *reinterpret_cast<const float4*>(p.bias + n + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
      f32x4 res[2][CPL / 4];                                // ping-pong: the loads of row group it + 1 are issued
      typedef typename std::conditional<sizeof(InT) == 4, f32x4, bf16x4>::type aux4_t;
      aux4_t aux[2][CPL / 4];                               // before the stores of row group it
      auto preload = [&](int it) {
        long m = mbase + Q * 16 + it * RPI + lr;
        m = m < p.M ? m : p.M - 1;                          // rows past M: load a valid row, never store
#pragma unroll
        for (int c4 = 0; c4 < CPL / 4; ++c4) {
          if constexpr (sizeof(OutT) == 4) {               // (bf16 outputs with a residual are rare: loaded in place below)
            if (p.residual) res[it & 1][c4] = *reinterpret_cast<const f32x4*>(p.residual + m * p.ld_res + n + 4 * c4);
          }
          if (p.act == FAVIT_ACT_DGELU || p.act == FAVIT_ACT_MULAUX)
            aux[it & 1][c4] = *reinterpret_cast<const aux4_t*>(reinterpret_cast<const InT*>(p.aux_in) + m * p.ld_aux_in + n + 4 * c4);
        }
      };
      preload(0);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        if (it + 1 < NIT) preload(it + 1);
        const int row = it * RPI + lr;
        const long m = mbase + Q * 16 + row;
        float a[CPL];
#pragma unroll
        for (int c4 = 0; c4 < CPL / 4; ++c4) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(wl + row * WEPI_LD + lc + 4 * c4);
          a[4 * c4] = fmaf(t[0], alpha, bv[c4].x); a[4 * c4 + 1] = fmaf(t[1], alpha, bv[c4].y);
          a[4 * c4 + 2] = fmaf(t[2], alpha, bv[c4].z); a[4 * c4 + 3] = fmaf(t[3], alpha, bv[c4].w);
        }
        if (m >= p.M) continue;
        auto store_vec = [&](OutT* dst, const float (&sv)[CPL]) {
          uint4 raw;
          if constexpr (sizeof(OutT) == 4) {
            raw = make_uint4(__float_as_uint(sv[0]), __float_as_uint(sv[1]), __float_as_uint(sv[2]), __float_as_uint(sv[3]));
          } else {
            bf16x8 o;
#pragma unroll
            for (int c = 0; c < 8; ++c) o[c] = (bf16_t)sv[c];
            raw = __builtin_bit_cast(uint4, o);
          }
#ifdef FAVIT_PROBE
          if ((p.dbg & 0x800) && raw.x != 0x12345678u) return;       // dbg 0x800: everything but the global stores (timing only)
#endif
          store16_policy(dst, raw, p.store_policy);
        };
        if (p.act == FAVIT_ACT_GELU_SAVEGRAD) {
          float gd[CPL];
#pragma unroll
          for (int c = 0; c < CPL; ++c) epi_gelu_both<InT>(a[c], a[c], gd[c]);
          if (p.aux_out) store_vec(reinterpret_cast<OutT*>(p.aux_out) + m * p.ld_aux_out + n, gd);
        } else if (p.aux_out) {
          store_vec(reinterpret_cast<OutT*>(p.aux_out) + m * p.ld_aux_out + n, a);
        }
        if (p.act == FAVIT_ACT_GELU) {
#pragma unroll
          for (int c = 0; c < CPL; ++c) a[c] = epi_gelu<InT>(a[c]);
        } else if (p.act == FAVIT_ACT_DGELU) {
#pragma unroll
          for (int c4 = 0; c4 < CPL / 4; ++c4)
#pragma unroll
            for (int c = 0; c < 4; ++c) a[4 * c4 + c] *= epi_dgelu<InT>((float)aux[it & 1][c4][c]);
        } else if (p.act == FAVIT_ACT_MULAUX) {
#pragma unroll
          for (int c4 = 0; c4 < CPL / 4; ++c4)
#pragma unroll
            for (int c = 0; c < 4; ++c) a[4 * c4 + c] *= (float)aux[it & 1][c4][c];
        }
        if (p.drop_thresh) {
          const uint64_t sd = favit_eff_seed(p.drop_seed, p.drop_epoch);
          const uint64_t e0 = (uint64_t)(m * p.N + n);
          if ((e0 & 1) == 0) {                             // (always, when N is even: n is a multiple of 4): one draw per pair
#pragma unroll
            for (int c = 0; c < CPL; c += 2) {
              bool k0, k1;
#ifdef FAVIT_PROBE
              if (p.dbg & 0x20000) { k0 = ((e0 + c) & 14) != 0; k1 = k0; } else   // dbg 0x20000: a mask without the draw (timing only)
#endif
              favit_keep2(sd, e0 + c, p.drop_thresh, k0, k1);
              a[c] = k0 ? a[c] * p.drop_scale : 0.f;
              a[c + 1] = k1 ? a[c + 1] * p.drop_scale : 0.f;
            }
          } else {
#pragma unroll
            for (int c = 0; c < CPL; ++c) a[c] = favit_keep(sd, e0 + c, p.drop_thresh) ? a[c] * p.drop_scale : 0.f;
          }
        }
        if (p.residual) {
#pragma unroll
          for (int c4 = 0; c4 < CPL / 4; ++c4) {
            f32x4 r;
            if constexpr (sizeof(OutT) == 4) r = res[it & 1][c4];
            else r = *reinterpret_cast<const f32x4*>(p.residual + m * p.ld_res + n + 4 * c4);
            a[4 * c4] += r[0]; a[4 * c4 + 1] += r[1]; a[4 * c4 + 2] += r[2]; a[4 * c4 + 3] += r[3];
          }
        }
        store_vec(C + m * p.ldc + n, a);
      }
    } else {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int row = it * RPI + lr;
        const long m = mbase + Q * 16 + row;
        if (m >= p.M) continue;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          if (n + c >= p.N) continue;
          float v = wl[row * WEPI_LD + lc + c] * alpha;
          if (p.bias) v += p.bias[n + c];
          float vx = v;
          if (p.act == FAVIT_ACT_GELU) v = epi_gelu<InT>(v);
          else if (p.act == FAVIT_ACT_GELU_SAVEGRAD) epi_gelu_both<InT>(v, v, vx);
          else if (p.act == FAVIT_ACT_DGELU)
            v *= epi_dgelu<InT>(to_f32(reinterpret_cast<const InT*>(p.aux_in)[m * p.ld_aux_in + n + c]));
          else if (p.act == FAVIT_ACT_MULAUX)
            v *= to_f32(reinterpret_cast<const InT*>(p.aux_in)[m * p.ld_aux_in + n + c]);
          if (p.aux_out) reinterpret_cast<OutT*>(p.aux_out)[m * p.ld_aux_out + n + c] = from_f32<OutT>(vx);
          if (p.drop_thresh) v = favit_keep(favit_eff_seed(p.drop_seed, p.drop_epoch), (uint64_t)(m * p.N + n + c), p.drop_thresh) ? v * p.drop_scale : 0.f;
          if (p.residual) v += p.residual[m * p.ld_res + n + c];
          C[m * p.ldc + n + c] = from_f32<OutT>(v);
        }
      }
    }
    }  // !done
  }
}

// accumulators of the 16x16x32 MFMA layout (acc[i][j]: rows 16 i + (lane & 15), columns 16 j + 4 (lane >> 4) + e)
template <typename InT, typename OutT, int Q, int NI, int NJ = 4, int CB = 0>
__device__ __forceinline__ void wave_epilogue_rows(const KParams& p, const f32x4 (&acc)[4][NJ], OutT* C, long mbase,
                                              long nbase, int lane, float* wl, bool first_split, float alpha) {
#pragma unroll
  for (int ii = 0; ii < NI; ++ii)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#ifdef FAVIT_PROBE
      if (!(p.dbg & 0x1000))                                       // dbg 0x1000: no accumulator writes to the LDS scratch
#endif
      *reinterpret_cast<f32x4*>(wl + (ii * 16 + (lane & 15)) * WEPI_LD + j * 16 + 4 * (lane >> 4)) = acc[Q + ii][CB * 4 + j];
  wave_epilogue_tail<InT, OutT, Q, NI>(p, C, mbase, nbase, lane, wl, first_split, alpha);
}

template <typename InT, typename OutT>
__device__ __forceinline__ void wave_epilogue(const KParams& p, const f32x4 (&acc)[4][4], OutT* C, long mbase,
                                              long nbase, int lane, float* wl, bool first_split, float alpha) {
  wave_epilogue_rows<InT, OutT, 0, 2>(p, acc, C, mbase, nbase, lane, wl, first_split, alpha);
  wave_epilogue_rows<InT, OutT, 2, 2>(p, acc, C, mbase, nbase, lane, wl, first_split, alpha);
}

// F8 = 0: bf16 operands.  F8 = 1 / 2: fp8 operands (OCP e4m3 B; A e4m3 / e5m2), both k-major: a stage is
// 64 k-values = the same 64-byte rows, so the LDS images, the DMA pieces and the 16-byte fragment reads
// are unchanged; a lane's 16 bytes are two 8-byte MFMA operands (k-order inside a stage is permuted
// identically for A and B, which a contraction does not notice).  Half the L2->LDS and LDS->register
// bytes per flop of the bf16 form; v_mfma_f32_16x16x32_{fp8,bf8}_fp8 runs at the bf16 MFMA rate.
template <int F8>
__device__ __forceinline__ f32x4 mfma_tile(const bf16x8& bfrag, const bf16x8& afrag, f32x4 acc) {
  if constexpr (F8 == 0) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfrag, afrag, acc, 0, 0, 0);
  } else {
    typedef __attribute__((ext_vector_type(2))) long l64x2;
    const l64x2 b2 = __builtin_bit_cast(l64x2, bfrag), a2 = __builtin_bit_cast(l64x2, afrag);
    if constexpr (F8 == 1) {
      acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b2[0], a2[0], acc, 0, 0, 0);
      return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b2[1], a2[1], acc, 0, 0, 0);
    } else {      // first operand = B fragment (e4m3), second = A fragment (e5m2)
      acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(b2[0], a2[0], acc, 0, 0, 0);
      return __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(b2[1], a2[1], acc, 0, 0, 0);
    }
  }
}

template <bool AK, bool BKM, typename OutT, int F8 = 0>
__device__ __forceinline__ void p4_body(const KParams& p, int tile, int split, int z) {
  static_assert(F8 == 0 || (AK && BKM), "fp8 operands are k-major");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ESZ = F8 ? 1 : 2;                    // operand element size
  constexpr int SBK = F8 ? 64 : P4_BK;               // k-values per 64-byte stage row
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  const long m0 = (long)(tile / p.tiles_n) * P4_BM;
  const long n0 = (long)(tile % p.tiles_n) * BN;
  const long zo = z / p.batch_inner, zi = z % p.batch_inner;
  const char* A = reinterpret_cast<const char*>(p.A) + (zo * p.sAo + zi * p.sAi) * ESZ;
  const char* Bm = reinterpret_cast<const char*>(p.B) + (zo * p.sBo + zi * p.sBi) * ESZ;
  OutT* C = reinterpret_cast<OutT*>(p.C) + zo * p.sCo + zi * p.sCi;
  const long kbeg = (long)split * p.k_per_split;
  const long kend = min(p.K, kbeg + p.k_per_split);
#ifdef FAVIT_PROBE
  const int nk = p.dbg == 2 ? 0 : (int)((kend - kbeg) / SBK);
#else
  const int nk = (int)((kend - kbeg) / SBK);
#endif
#ifdef FAVIT_PROBE
  // dbg 16: stagger experiment -- one of the two first-round workgroups of a CU starts (dbg >> 8) us late, so that its
  // epilogue (HBM) falls on the other's main loop (L2 -> LDS + MFMA).  bit 5 selects the guess of which blocks share a CU.
  if ((p.dbg & 16) && blockIdx.x < 512 && blockIdx.y == 0 && blockIdx.z == 0) {
    const int idx = blockIdx.x >> 3;
    const bool late = (p.dbg & 32) ? (idx & 1) : ((idx >> 5) & 1);
    if (late) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      const unsigned long long ticks = (unsigned long long)(p.dbg >> 8) * 100ull;
      while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    }
  }
#endif
#ifdef FAVIT_PROBE
  // dbg 128 + probe buffer: per-workgroup timeline [block][4] = {start, main loop done, end} in 10-ns ticks + placement
  const bool tl = (p.dbg & 128) && p.probe && tid == 0 && blockIdx.y == 0 && blockIdx.z == 0;
  unsigned long long tl0 = 0, tl1 = 0;
  if (tl) tl0 = __builtin_amdgcn_s_memrealtime();
#endif
  // fused bias gradient (row sums of the mn-major A operand) by one extra ones-MFMA per A fragment.  The two waves of
  // a row (wc = 0 / 1) hold the SAME four A fragments: each takes two of them (round 4: eight accumulator registers
  // per wave instead of sixteen in every other wave -- with sixteen the allocator spilled them inside the K loop of
  // the grouped kernel, and the reload's vmcnt(0) drained the DMA ring every k-step: 233 -> 306 us per cfg2 block).
  const bool do_rowsum = (!AK) && (p.a_rowsum != nullptr) && (n0 == 0);
  f32x4 racc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) racc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)1.0f;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // 3 pieces per wave per stage: 2 of A (16 pieces), 1 of B (8 pieces)
  const char* sa[2];
  const char* sb;
  int da[2], db;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int qa = wave * 2 + j;
    if (F8) {
      const int row = 16 * qa + (lane >> 2), c = (lane & 3) ^ ksw32(row);
      long i = m0 + row;
      i = i < p.M ? i : p.M - 1;
      sa[j] = A + i * p.lda + kbeg + c * 16;
      da[j] = qa * 1024;
    } else if (AK) {
      sa[j] = reinterpret_cast<const char*>(glds_src32<true>(reinterpret_cast<const bf16_t*>(A), p.lda, m0, p.M, kbeg, qa, lane));
      da[j] = qa * 1024;
    } else {
      const int sub = qa >> 3, q = qa & 7;
      sa[j] = reinterpret_cast<const char*>(glds_src32<false>(reinterpret_cast<const bf16_t*>(A), p.lda, m0 + sub * 128, p.M, kbeg, q, lane));
      da[j] = sub * (P4_A_BYTES / 2) + q * 1024;
    }
  }
  if (F8) {
    const int row = 16 * wave + (lane >> 2), c = (lane & 3) ^ ksw32(row);
    long i = n0 + row;
    i = i < p.N ? i : p.N - 1;
    sb = Bm + i * p.ldb + kbeg + c * 16;
  } else {
    sb = reinterpret_cast<const char*>(glds_src32<BKM>(reinterpret_cast<const bf16_t*>(Bm), p.ldb, n0, p.N, kbeg, wave, lane));
  }
  db = P4_A_BYTES + wave * 1024;
  const long a_step = AK ? 64 : (long)P4_BK * p.lda * 2;      // bytes per stage
  const long b_step = BKM ? 64 : (long)P4_BK * p.ldb * 2;
#ifdef FAVIT_PROBE
  // dbg 0x2000 (timing only, wrong results): the B operand is NOT streamed -- what a launch would take if a weight
  // slice stayed in LDS and only A were fetched (tools/gemm_bench.py)
  const bool no_b = (p.dbg & 0x2000) != 0;
#else
  constexpr bool no_b = false;
#endif
  auto issue = [&](int buf) {
    char* st = smem + buf * P4_STAGE;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __builtin_amdgcn_global_load_lds((gptr_t)sa[j], (lptr_t)(st + da[j]), 16, 0, 0);
      sa[j] += a_step;
    }
    if (!no_b) {
      __builtin_amdgcn_global_load_lds((gptr_t)sb, (lptr_t)(st + db), 16, 0, 0);
      sb += b_step;
    }
  };

  if (nk > 0) issue(0);
  if (nk > 1) issue(1);
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      if (no_b) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk) issue(cur >= 1 ? cur - 1 : 2);
    const char* st = smem + cur * P4_STAGE;
    const char* la = AK ? st : st + (wr >> 1) * (P4_A_BYTES / 2);
    const int ra = AK ? wr * 64 : (wr & 1) * 64;
    const char* lb = st + P4_A_BYTES;
    bf16x8 af[4], bfr[4];
    // k-major operand first: its compiler-tracked ds_read_b128 are in flight while the transposed reads issue
    if constexpr (!AK && !BKM) {               // weight gradients: all 16 transposed reads in flight, one wait
      u32x2_t al[4], ah[4], bl[4], bh[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) tr_read_pair(la, ra + i * 16, 0, lane, al[i], ah[i]);
#pragma unroll
      for (int j = 0; j < 4; ++j) tr_read_pair(lb, wc * 64 + j * 16, 0, lane, bl[j], bh[j]);
      tr_fence2(al, ah, bl, bh);
#pragma unroll
      for (int i = 0; i < 4; ++i) { af[i] = tr_pack(al[i], ah[i]); bfr[i] = tr_pack(bl[i], bh[i]); }
    } else if constexpr (AK) {
      load_frags4<true, true>(la, ra, 0, lane, af);
      load_frags4<BKM, true>(lb, wc * 64, 0, lane, bfr);
    } else {
      load_frags4<BKM, true>(lb, wc * 64, 0, lane, bfr);
      load_frags4<AK, true>(la, ra, 0, lane, af);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mfma_tile<F8>(bfr[j], af[i], acc[i][j]);
    if (F8 == 0 && do_rowsum) {
      if (wc) {                                  // (wave-uniform)
        racc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[2], racc[0], 0, 0, 0);
        racc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[3], racc[1], 0, 0, 0);
      } else {
        racc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[0], racc[0], 0, 0, 0);
        racc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[1], racc[1], 0, 0, 0);
      }
    }
    cur = cur == 2 ? 0 : cur + 1;
  }
  if (F8 == 0 && do_rowsum && lane < 16) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long m = m0 + wr * 64 + (2 * wc + i) * 16 + lane;
      if (m < p.M) {
        if (p.rowsum_store) p.a_rowsum[m] = racc[i][0];       // exactly one wave of one tile column (n0 == 0) writes row m
        else atomicAdd(p.a_rowsum + m, racc[i][0]);
      }
    }
  }
#ifdef FAVIT_PROBE
  if (p.dbg == 1) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678f) C[0] = from_f32<OutT>(s);
    return;
  }
#endif
  float alpha = p.alpha;
  if (F8) {
    if (p.scale_a) alpha *= p.scale_a[0];
    if (p.scale_b) alpha *= p.scale_b[0];
  }
  __syncthreads();        // every wave is done with the stage buffers; LDS becomes wave-private scratch
#ifdef FAVIT_PROBE
  if (tl) tl1 = __builtin_amdgcn_s_memrealtime();
#endif
  wave_epilogue<bf16_t, OutT>(p, acc, C, m0 + wr * 64, n0 + wc * 64, lane,
                              reinterpret_cast<float*>(smem + wave * WEPI_BYTES), split == 0, alpha);
#ifdef FAVIT_PROBE
  if (p.dbg & 128) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's stores have left
    if (tl) {
      const unsigned long long tl2 = __builtin_amdgcn_s_memrealtime();
      const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));
      unsigned long long* o = p.probe + (size_t)blockIdx.x * 4;
      o[0] = tl0; o[1] = tl1; o[2] = tl2; o[3] = ((unsigned long long)(xcc & 0xF) << 32) | hw;
    }
  }
#endif
}

// The tail of a p4 launch as quarter tiles.  T tiles on 512 workgroup slots leave T mod 512 tiles for a last round
// that occupies a fraction of the CUs for a whole tile time, one latency-bound workgroup per CU (the N = 384 GEMMs
// of the step: 591 tiles, the last 79 alone on the chip for 38 % of the launch).  A latency-bound tile's time is
// its number of k-steps, not its rows, so those tiles are issued as four 64x128 quarters each with 64-deep stages
// (half the barriers; the 24-KiB stage of the 64-row kernel, which is exactly a p4 stage): 8 waves as 4x2 of 16x64,
// one A and two B DMA pieces per wave and stage, the wave-private epilogue of the full tile.  k-major A, K % 64 == 0.
template <bool BKM, typename OutT>
__device__ __forceinline__ void p4_quarter_body(const KParams& p, int q) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(S64_STAGE == P4_STAGE, "the quarter tiles reuse the stage buffers of the full tiles");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int t = p.q_tile0 + (q >> 2);
  const long m0 = (long)(t / p.tiles_n) * P4_BM + 64 * (q & 3);
  const long n0 = (long)(t % p.tiles_n) * BN;
  if (m0 >= p.M) return;                              // uniform for the workgroup, before any barrier
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* Bm = reinterpret_cast<const bf16_t*>(p.B);
  OutT* C = reinterpret_cast<OutT*>(p.C);
  const int nk = (int)(p.K / BK16);

  f32x4 acc[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[0][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bf16_t* sa = glds_src<true>(A, p.lda, m0, p.M, 0, wave, lane);
  const bf16_t* sb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) sb[j] = glds_src<BKM>(Bm, p.ldb, n0, p.N, 0, wave * 2 + j, lane);
  auto issue = [&](int buf) {
    char* st = smem + buf * S64_STAGE;
    __builtin_amdgcn_global_load_lds((gptr_t)sa, (lptr_t)(st + wave * 1024), 16, 0, 0);
    sa += BK16;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __builtin_amdgcn_global_load_lds((gptr_t)sb[j], (lptr_t)(st + S64_A_BYTES + (wave * 2 + j) * 1024), 16, 0, 0);
      sb[j] += BKM ? BK16 : BK16 * p.ldb;
    }
  };
  if (nk > 0) issue(0);
  if (nk > 1) issue(1);
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk) issue(cur >= 1 ? cur - 1 : 2);
    const char* la = smem + cur * S64_STAGE;
    const char* lb = la + S64_A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 af = load_frag16<true>(la, wr * 16, ks, lane);
      bf16x8 bfr[4];
      load_frags4<BKM, false>(lb, wc * 64, ks, lane, bfr);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af, acc[0][j], 0, 0, 0);
    }
    cur = cur == 2 ? 0 : cur + 1;
  }
  __syncthreads();        // every wave is done with the stage buffers; LDS becomes wave-private scratch
  wave_epilogue_rows<bf16_t, OutT, 0, 1>(p, acc, C, m0 + wr * 16, n0 + wc * 64, lane,
                                         reinterpret_cast<float*>(smem + wave * WEPI_BYTES), true, p.alpha);
}

template <bool AK, bool BKM, typename OutT>
__global__ __launch_bounds__(P4_THREADS, 4) void gemm_bf16_p4_kernel(KParams p) {
  if constexpr (AK) {
    if (p.q_block0 > 0) {
      // (quarter tiles FIRST instead -- finishing early, staggering the full tiles behind them -- measured the same)
      const int b = (int)blockIdx.x;
      if (b >= p.q_block0) p4_quarter_body<BKM, OutT>(p, xcd_remap(b - p.q_block0, (int)gridDim.x - p.q_block0));
      else p4_body<AK, BKM, OutT>(p, xcd_remap(b, p.q_block0), 0, 0);
      return;
    }
  }
  int tile, split;
  tile_and_split(p, tile, split);
  p4_body<AK, BKM, OutT>(p, tile, split, blockIdx.z);
}

// --------------------------------------------------------------------------------------
// bf16 kernel "pd" (EXPERIMENTAL, FAVIT_GEMM_PD=1): the 256x128 tile and the 64x64 wave tiles of p4 as ONE persistent
// 8-wave workgroup per CU with a DEFERRED epilogue -- DESIGN.md section 7, item 1.  A workgroup walks its tiles; the
// accumulators of tile t move to a second register set and are written out in four 16-row slices per wave BETWEEN the
// k-steps of tile t + 1 (a wave slices every third k-step; the two waves of a SIMD never in the same step), while the
// DMA ring (three 24-KiB stages) streams on across tile borders.
//   vmcnt is ONE in-order counter for DMA pieces, loads and stores.  Every vector-memory operation of a wave is
//   therefore counted (`issued`), every ring stage remembers the count at its issue (`marks`), and "stage g has landed"
//   = at most (issued - mark[g]) younger operations outstanding.  Nothing is ever loaded into a compiler-managed
//   register asynchronously: the epilogue's operands (residual rows, bias) travel by the same LDS-DMA as the GEMM
//   operands, into wave-private buffers, one slice (three k-steps) ahead; its stores are inline asm (exact count).
//   NT layout, M % 256 == 0, N % 128 == 0, K % 32 == 0, K >= 384.  EPI 0: bf16 out (+ bias); EPI 1: fp32 out + bias +
//   fp32 residual.
// --------------------------------------------------------------------------------------
constexpr int PD_STAGES = 3;
constexpr int PD_RING = PD_STAGES * P4_STAGE;             // 73728
constexpr int PD_SCRATCH = 16 * WEPI_LD * 4;              // 4352 B: one 16-row slice of accumulators, transposed through LDS
constexpr int PD_RBUF = 4096;                             // residual rows of the next slice: 4 DMA pieces [4 rows][64 fp32]
constexpr int PD_BBUF = 1024;                             // bias of the tile: one DMA piece (16 chunks of 4 columns)
constexpr int PD_WAVE = PD_SCRATCH + PD_RBUF + PD_BBUF;   // 9472
constexpr int PD_LDS = PD_RING + 8 * PD_WAVE;             // 149504

__device__ __forceinline__ void pd_wait(int younger) {
  // until at most `younger` operations are outstanding, rounded DOWN to an available immediate (never too few)
  if (younger >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if (younger >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  else if (younger >= 17) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
  else if (younger >= 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
  else if (younger >= 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  else if (younger >= 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
  else if (younger >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (younger >= 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
  else if (younger >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if (younger >= 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if (younger >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (younger >= 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if (younger >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (younger >= 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if (younger >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (younger >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if (younger >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if (younger >= 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// 16-byte non-temporal store as inline asm (exactly one vector-memory instruction; the data registers are read at
// issue).  (An SGPR-base form -- "s"(base) + 32-bit lane offset -- faulted on a null base in the probe build of the
// EPI 1 instantiation, which spills 49 SGPRs; the 64-bit VGPR pointer form is the one store16_policy uses.)
__device__ __forceinline__ void pd_gstore16(void* base, unsigned off, const f32x4& v) {
  void* ptr = reinterpret_cast<char*>(base) + off;
  asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(ptr), "v"(v) : "memory");
}

template <int EPI>
__global__ __launch_bounds__(P4_THREADS, 2) void gemm_bf16_pd_kernel(KParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int ph = wave % 3;                               // this wave slices at k-steps ph, ph + 3, ph + 6, ph + 9
  char* wbase = smem + PD_RING + wave * PD_WAVE;
  float* wl = reinterpret_cast<float*>(wbase);
  char* rbuf = wbase + PD_SCRATCH;
  char* bbuf = rbuf + PD_RBUF;
  const char* A = reinterpret_cast<const char*>(p.A);
  const char* Bm = reinterpret_cast<const char*>(p.B);
  const int nk = (int)(p.K / P4_BK);
  const int nwg = (int)gridDim.x, bid = (int)blockIdx.x;
  const int my_tiles = (p.ntiles - bid + nwg - 1) / nwg;  // tiles bid, bid + nwg, ... (nwg % 8 == 0: all on this XCD's share)
  typedef typename std::conditional<EPI == 0, bf16_t, float>::type OutT;
  OutT* C = reinterpret_cast<OutT*>(p.C);

  auto coords = [&](int ordinal, long& m0, long& n0) __attribute__((always_inline)) {
    const int t = xcd_remap(bid + ordinal * nwg, p.ntiles);
    m0 = (long)(t / p.tiles_n) * P4_BM;
    n0 = (long)(t % p.tiles_n) * BN;
  };

  // ---- issue side of the ring: runs two stages ahead of the consumer, across tile borders ----
  int issued = 0, islot = 0, is_tile = 0, is_k = 0;
  unsigned long long marks = 0;   // three 16-bit fields: `issued` (mod 2^16) right after the DMA of the stage in ring slot 0..2
  const char* sa[2];
  const char* sb;
  auto set_issue_tile = [&](int ordinal) __attribute__((always_inline)) {
    long m0, n0;
    coords(ordinal, m0, n0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      sa[j] = reinterpret_cast<const char*>(glds_src32<true>(reinterpret_cast<const bf16_t*>(A), p.lda, m0, p.M, 0, wave * 2 + j, lane));
    sb = reinterpret_cast<const char*>(glds_src32<true>(reinterpret_cast<const bf16_t*>(Bm), p.ldb, n0, p.N, 0, wave, lane));
  };
  auto issue_one = [&]() __attribute__((always_inline)) {
    if (is_tile < my_tiles) {                            // (uniform) else: nothing left to stream
      char* st = smem + islot * P4_STAGE;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        __builtin_amdgcn_global_load_lds((gptr_t)sa[j], (lptr_t)(st + (wave * 2 + j) * 1024), 16, 0, 0);
        sa[j] += 64;
      }
      __builtin_amdgcn_global_load_lds((gptr_t)sb, (lptr_t)(st + P4_A_BYTES + wave * 1024), 16, 0, 0);
      sb += 64;
      issued += 3;
      const int sh = 16 * islot;                         // (scalar shifts on wave-uniform values: no array, no scratch)
      marks = (marks & ~(0xFFFFull << sh)) | ((unsigned long long)(issued & 0xFFFF) << sh);
      if (++is_k == nk) {
        is_k = 0;
        if (++is_tile < my_tiles) set_issue_tile(is_tile);
      }
    }
    islot = islot == PD_STAGES - 1 ? 0 : islot + 1;      // (the slot advances even when nothing is issued: it mirrors the consumer)
  };

  // ---- deferred epilogue: accumulators and coordinates of the PREVIOUS tile; its operands arrive in rbuf / bbuf ----
  f32x4 acc[4][4], prev[4][4];
  long pm0 = 0, pn0 = 0;
  // lane offsets (bytes) inside a slice: EPI 0: row lane / 8, columns 8 (lane % 8) .. of bf16; EPI 1: row lane / 16,
  // columns 4 (lane % 16) .. of fp32 -- the uniform part (tile, wave, slice, row group) goes into the SGPR base
  const unsigned loff_c = EPI == 0 ? (unsigned)(((lane >> 3) * p.ldc + 8 * (lane & 7)) * 2)
                                   : (unsigned)(((lane >> 4) * p.ldc + 4 * (lane & 15)) * 4);
  const long loff_r = ((long)(lane >> 4) * p.ld_res + 4 * (lane & 15)) * 4;
  // residual rows of slice Q of the tile at (m0, n0) -> rbuf: piece `it` = rows 4 it .. 4 it + 3, lane -> its own 16 bytes
  auto load_res = [&](long m0, long n0, int Q) __attribute__((always_inline)) {
    if constexpr (EPI == 1) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const char* src = reinterpret_cast<const char*>(p.residual + (m0 + wr * 64 + Q * 16 + it * 4) * p.ld_res + n0 + wc * 64) + loff_r;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(rbuf + it * 1024), 16, 0, 0);
      }
      issued += 4;
    }
  };
  // bias columns n0 + 64 wc .. + 63 -> bbuf: lane l brings chunk l % 16 (4 columns) to bbuf + 16 l
  auto load_bias = [&](long n0) __attribute__((always_inline)) {
    if (p.bias != nullptr) {
      const char* src = reinterpret_cast<const char*>(p.bias + n0 + wc * 64 + 4 * (lane & 15));
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)bbuf, 16, 0, 0);
      issued += 1;
    }
  };
  auto slice = [&](auto qtag) __attribute__((always_inline)) {
    constexpr int Q = decltype(qtag)::value;
    // accumulators of rows 16 Q .. 16 Q + 15 -> wave-private LDS scratch (transposed reads below)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<f32x4*>(wl + (lane & 15) * WEPI_LD + j * 16 + 4 * (lane >> 4)) = prev[Q][j];
    // (same wave wrote and reads: the compiler's lgkmcnt wait orders them)
    if constexpr (EPI == 0) {
      f32x4 b0 = (f32x4){0.f, 0.f, 0.f, 0.f}, b1 = b0;
      if (p.bias != nullptr) {
        b0 = *reinterpret_cast<const f32x4*>(bbuf + 32 * (lane & 7));
        b1 = *reinterpret_cast<const f32x4*>(bbuf + 32 * (lane & 7) + 16);
      }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int row = it * 8 + (lane >> 3), col = 8 * (lane & 7);
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(wl + row * WEPI_LD + col);
        const f32x4 t1 = *reinterpret_cast<const f32x4*>(wl + row * WEPI_LD + col + 4);
        bf16x8 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          o[c] = (bf16_t)fmaf(t0[c], p.alpha, b0[c]);
          o[4 + c] = (bf16_t)fmaf(t1[c], p.alpha, b1[c]);
        }
        pd_gstore16(C + (pm0 + wr * 64 + Q * 16 + it * 8) * p.ldc + pn0 + wc * 64, loff_c, __builtin_bit_cast(f32x4, o));
      }
      issued += 2;
    } else {
      f32x4 b0 = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (p.bias != nullptr) b0 = *reinterpret_cast<const f32x4*>(bbuf + 16 * (lane & 15));
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 4 + (lane >> 4), col = 4 * (lane & 15);
        const f32x4 t = *reinterpret_cast<const f32x4*>(wl + row * WEPI_LD + col);
        const f32x4 r = *reinterpret_cast<const f32x4*>(rbuf + it * 1024 + 16 * lane);
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = fmaf(t[c], p.alpha, b0[c]) + r[c];
        pd_gstore16(C + (pm0 + wr * 64 + Q * 16 + it * 4) * p.ldc + pn0 + wc * 64, loff_c, o);
      }
      issued += 4;
      // (every read of rbuf has been consumed by a store above: the next slice's rows may land in it)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (Q < 3) load_res(pm0, pn0, Q + 1);              // the next slice's residual rows: three k-steps of lead
    }
  };
  auto slice_n = [&](int j) __attribute__((always_inline)) {
    if (j == 0) slice(std::integral_constant<int, 0>());
    else if (j == 1) slice(std::integral_constant<int, 1>());
    else if (j == 2) slice(std::integral_constant<int, 2>());
    else slice(std::integral_constant<int, 3>());
  };

  set_issue_tile(0);
  issue_one();
  issue_one();
  int cslot = 0;
  for (int ti = 0; ti < my_tiles; ++ti) {
    long m0, n0;
    coords(ti, m0, n0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < nk; ++k) {
      // this wave's pieces of the stage in slot cslot have landed: at most (issued - mark) younger operations in flight
      pd_wait((issued - (int)((marks >> (16 * cslot)) & 0xFFFF)) & 0xFFFF);
      __builtin_amdgcn_s_barrier();                      // ... everybody's; and everybody is done with the previous stage
#ifdef FAVIT_PROBE
      const bool no_epi = (p.dbg & 0x10000) != 0;         // probe build: no slices, no epilogue operands (timing only)
#else
      constexpr bool no_epi = false;
#endif
      if (ti > 0 && !no_epi) {
        const int d = k - ph;
        if (d >= 0 && d < 12 && d % 3 == 0) slice_n(d / 3);
      }
      if (k == nk - 3 + ph && !no_epi) {                 // operands of THIS tile's first slice (runs in the next tile's step ph,
        load_bias(n0);                                   // or in the drain): the previous tile's last slice has read both buffers
        load_res(m0, n0, 0);
      }
      issue_one();                                       // two stages ahead, into the slot of the stage consumed last
      const char* st = smem + cslot * P4_STAGE;
      bf16x8 af[4], bfr[4];
      load_frags4<true, true>(st, wr * 64, 0, lane, af);
      load_frags4<true, true>(st + P4_A_BYTES, wc * 64, 0, lane, bfr);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      cslot = cslot == PD_STAGES - 1 ? 0 : cslot + 1;
    }
    // tile border: the finished accumulators become the deferred epilogue's input
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) prev[i][j] = acc[i][j];
    pm0 = m0; pn0 = n0;
  }
  // drain: the last tile's epilogue, nothing left to overlap it with
#ifdef FAVIT_PROBE
  if (p.dbg & 0x10000) {                                  // (keeps the accumulators alive)
    float sres = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) sres += prev[i][j][0] + prev[i][j][1] + prev[i][j][2] + prev[i][j][3];
    if (sres == 12345.678f) C[0] = (OutT)sres;
    return;
  }
#endif
  for (int j = 0; j < 4; ++j) {
    pd_wait(0);
    slice_n(j);
  }
  pd_wait(0);
}

template <typename Kn>
int launch_pd(Kn kernel, const KParams& kp, int nwg, hipStream_t st) {
  g_last_kernel = "pd";
  favit_ensure_dyn_lds(reinterpret_cast<const void*>(kernel), PD_LDS);
  hipLaunchKernelGGL(kernel, dim3((unsigned)nwg), dim3(P4_THREADS), PD_LDS, st, kp);
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

// fp8 operands (F8 = 1: e4m3 x e4m3, 2: e5m2 A x e4m3 B), NT layout only
template <typename OutT, int F8>
__global__ __launch_bounds__(P4_THREADS, 4) void gemm_fp8_p4_kernel(KParams p) {
  int tile, split;
  tile_and_split(p, tile, split);
  p4_body<true, true, OutT, F8>(p, tile, split, blockIdx.z);
}

// --------------------------------------------------------------------------------------
// bf16 kernel "pp" (ping-pong): the 256x128 tile and the 64x64 wave tiles of p4, but ONE 512-thread
// workgroup per CU whose two halves -- waves 0-3 (rows 0-127) and waves 4-7 (rows 128-255); waves w and
// w+4 share a SIMD -- run one barrier apart: while one half issues its 16 MFMAs of a 32-deep k-step, the
// other half reads its next fragments from LDS and issues the DMA of a later stage, then they swap.  The
// matrix pipe of every SIMD always has one wave in its MFMA cluster; LDS-read latency, DMA issue and the
// barrier are hidden under the partner's cluster instead of relying on occupancy (p4: 4 waves per SIMD,
// MFMA pipe 19-26 % busy, 40-60 % of wave time parked on s_waitcnt / s_barrier).
//   stage  = 64 k-values: A image [256 rows][128 B] + B image ([128 rows][128 B] k-major, or [64 k][256 B]
//            mn-major), 48 KiB, the same swizzled images and 1-KiB DMA pieces as the 128x128 DMA kernel;
//            three stages (144 KiB): stage s+2 is fetched while stage s is consumed;
//   reads  = inline-asm ds_read_b128 / ds_read_b64_tr_b16 + ONE explicit lgkmcnt(0) after the barrier, so
//            the compiler's LDS-DMA alias wait (vmcnt(0) before a read) can never drain the prefetch;
//   barrier bookkeeping (B(n) = n-th hardware barrier; X / Y = barrier before / after a cluster):
//            half 0 runs R(s,0) X C(s,0) Y R(s,1) X C(s,1) Y ..., half 1 the same one barrier later.
//            RAW: every wave waits (counted vmcnt) for its own pieces of stage s+1 before X(s,1); the first
//            read of stage s+1 (half 0's R(s+1,0)) comes after B(4s+4), which both X(s,1) precede.
//            WAR: stage s+2 overwrites stage s-1, last read in half 1's R(s-1,1), complete before its
//            Y(s-1,1) = B(4s+1); half 0 therefore issues the DMA in R(s,1) (after B(4s+2)), half 1 in
//            R(s,0) (after B(4s+1)).
// The epilogue is p4's wave-private one and starts without a barrier: a wave's scratch lies in a stage
// buffer that no wave reads any more (the two that do not hold the last stage).
// --------------------------------------------------------------------------------------
constexpr int PP_THREADS = 768;                     // 8 MFMA waves (two ping-pong halves) + 4 loader waves
constexpr int PP_A_BYTES = 256 * 128;                 // 32 KiB
constexpr int PP_B_BYTES = 128 * 128;                 // 16 KiB
constexpr int PP_STAGE = PP_A_BYTES + PP_B_BYTES;     // 48 KiB
constexpr int PP_LDS = 3 * PP_STAGE;                  // 147456

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((address_space(3))) const char* lds_cptr_t;

// the four k-major fragments of rows r0 + 16 i (r0 a multiple of 16), k-substep ks of a BK = 64 image
__device__ __forceinline__ void pp_kread4(const char* lds, int r0, int ks, int lane, u32x4_t (&f)[4]) {
  const int row = r0 + (lane & 15), kc = ks * 4 + (lane >> 4);
  const unsigned addr = (unsigned)(uintptr_t)(lds_cptr_t)(lds + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
  asm volatile("ds_read_b128 %0, %1" : "=v"(f[0]) : "v"(addr) : "memory");
  asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(f[1]) : "v"(addr) : "memory");
  asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(f[2]) : "v"(addr) : "memory");
  asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(f[3]) : "v"(addr) : "memory");
}

__device__ __forceinline__ void pp_fence_kk(u32x4_t (&a)[4], u32x4_t (&b)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
               :
               : "memory");
}

__device__ __forceinline__ void pp_fence_km(u32x4_t (&a)[4], u32x2_t (&l)[4], u32x2_t (&h)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3]),
                 "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3])
               :
               : "memory");
}

template <bool BKM, typename OutT>
__global__ __launch_bounds__(PP_THREADS) void gemm_bf16_pp_kernel(KParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const long m0 = (long)(tile / p.tiles_n) * P4_BM;
  const long n0 = (long)(tile % p.tiles_n) * BN;
  const int nk = (int)(p.K / BK16);                   // >= 2 (host)
  const int nb = 4 * nk;                              // barrier intervals of the main loop

  if (wave >= 8) {
    // ---- loader wave l: A pieces 8l .. 8l+7 and B pieces 4l .. 4l+3 of every stage, three per interval ----
    const int l = wave - 8;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* Bm = reinterpret_cast<const bf16_t*>(p.B);
    const bf16_t* sa[8];
    const bf16_t* sb[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) sa[j] = glds_src<true>(A, p.lda, m0, p.M, 0, l * 8 + j, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) sb[j] = glds_src<BKM>(Bm, p.ldb, n0, p.N, 0, l * 4 + j, lane);
    const long b_step = BKM ? BK16 : (long)BK16 * p.ldb;
    // group g (0..3) of the stage that goes to ring slot `slot`: A pieces 2g, 2g+1 and B piece g of this wave
    auto issue_group = [&](int slot, int g) {
      char* st = smem + slot * PP_STAGE;
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) {
        if (gg == g) {
          __builtin_amdgcn_global_load_lds((gptr_t)sa[2 * gg], (lptr_t)(st + (l * 8 + 2 * gg) * 1024), 16, 0, 0);
          sa[2 * gg] += BK16;
          __builtin_amdgcn_global_load_lds((gptr_t)sa[2 * gg + 1], (lptr_t)(st + (l * 8 + 2 * gg + 1) * 1024), 16, 0, 0);
          sa[2 * gg + 1] += BK16;
          __builtin_amdgcn_global_load_lds((gptr_t)sb[gg], (lptr_t)(st + PP_A_BYTES + (l * 4 + gg) * 1024), 16, 0, 0);
          sb[gg] += b_step;
        }
      }
    };
    // intervals -7 .. -1 of the schedule below: stage 0 whole, three groups of stage 1
#pragma unroll
    for (int g = 0; g < 4; ++g) issue_group(0, g);
#pragma unroll
    for (int g = 0; g < 3; ++g) issue_group(1, g);
    asm volatile("s_waitcnt vmcnt(9)" ::: "memory");  // own pieces of stage 0 landed
    __builtin_amdgcn_s_barrier();                     // B(0)
    // interval n (between B(n) and B(n+1)): group (n+7)%4 of stage t = (n+7)/4, whose slot is free once B(4t-7)
    // has been passed; before B(4t') the pieces of stage t' must have landed.
    int t = 1, g = 3, slot = 1;
    for (int n = 0; n < nb; ++n) {
#ifdef FAVIT_PROBE
      if (t < nk && !(p.dbg & 4)) issue_group(slot, g);          // dbg 4: no DMA in the loop (timing only)
#else
      if (t < nk) issue_group(slot, g);
#endif
      if ((n & 3) == 3) {                             // next barrier is B(n+1) = B(4t'), t' = (n+1)/4: stage t' must have landed
        const int tp = (n + 1) >> 2;
        if (tp < nk) {
          // younger than stage t' at this point: the three groups of stage t'+1 issued in intervals n-2 .. n
          if (tp + 1 < nk) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      __builtin_amdgcn_s_barrier();                   // B(n+1)
      if (++g == 4) { g = 0; ++t; slot = slot == 2 ? 0 : slot + 1; }
    }
    return;
  }

  // ---- consumer waves: two ping-pong halves of four ----
  const int half = wave >> 2;                         // waves w and w + 4 share a SIMD
  const int wr = half * 2 + ((wave >> 1) & 1), wc = wave & 1;
  OutT* C = reinterpret_cast<OutT*>(p.C);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  __builtin_amdgcn_s_barrier();                       // B(0): stage 0 landed
  if (half == 1) __builtin_amdgcn_s_barrier();        // half 1 runs one barrier behind half 0

#ifdef FAVIT_PROBE
  // in-kernel stamps (probe build): cycles per wave spent in [read section + X wait], [fence + MFMA issue], [Y wait]
  unsigned long long pt_r = 0, pt_c = 0, pt_y = 0, pt0 = 0, pt1 = 0, pt2 = 0;
#define PP_STAMP(x) do { if (p.probe) { x = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(x) :: "memory"); } } while (0)
  PP_STAMP(pt0);
  const unsigned long long pt_begin = pt0;
#else
#define PP_STAMP(x)
#endif
  int cur = 0;                                        // ring slot of stage s
  for (int s = 0; s < nk; ++s) {
    const char* la = smem + cur * PP_STAGE;
    const char* lb = la + PP_A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4_t af[4], bk4[4];
      u32x2_t bl[4], bh[4];
#ifdef FAVIT_PROBE
      if (p.dbg & 8) {                                // dbg 8: no LDS fragment reads (timing only)
#pragma unroll
        for (int i = 0; i < 4; ++i) { af[i] = (u32x4_t){1u, 2u, 3u, 4u}; bk4[i] = af[i]; bl[i] = (u32x2_t){1u, 2u}; bh[i] = bl[i]; }
      } else
#endif
      {
      pp_kread4(la, wr * 64, ks, lane, af);
      if constexpr (BKM) {
        pp_kread4(lb, wc * 64, ks, lane, bk4);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) tr_read_pair(lb, wc * 64 + j * 16, ks, lane, bl[j], bh[j]);
      }
      }
      __builtin_amdgcn_s_barrier();                   // X(s, ks)
      bf16x8 bfr[4];
      if constexpr (BKM) {
        pp_fence_kk(af, bk4);
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = __builtin_bit_cast(bf16x8, bk4[j]);
      } else {
        pp_fence_km(af, bl, bh);
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = tr_pack(bl[j], bh[j]);
      }
      __builtin_amdgcn_sched_barrier(0);
#ifdef FAVIT_PROBE
      PP_STAMP(pt1);
      pt_r += pt1 - pt0;
      __builtin_amdgcn_sched_barrier(0);
#endif
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], __builtin_bit_cast(bf16x8, af[i]), acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
#ifdef FAVIT_PROBE
      PP_STAMP(pt2);
      pt_c += pt2 - pt1;
      __builtin_amdgcn_sched_barrier(0);
#endif
      // Y(s, ks); half 1 omits its very last one: with its extra barrier up front every wave executes 4 nk + 1
      if (!(half == 1 && s == nk - 1 && ks == 1)) __builtin_amdgcn_s_barrier();
#ifdef FAVIT_PROBE
      PP_STAMP(pt0);
      pt_y += pt0 - pt2;
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    cur = cur == 2 ? 0 : cur + 1;
  }
  // `cur` is now the slot after the last stage's; the last stage sits in slot (cur + 2) % 3 and may still be
  // read by half 1.  Wave-private epilogue scratch lives in the two other slots (4 waves each).
  const int free0 = cur, free1 = cur == 2 ? 0 : cur + 1;
  float* wl = reinterpret_cast<float*>(smem + (half ? free1 : free0) * PP_STAGE + (wave & 3) * WEPI_BYTES);
  wave_epilogue<bf16_t, OutT>(p, acc, C, m0 + wr * 64, n0 + wc * 64, lane, wl, true, p.alpha);
#ifdef FAVIT_PROBE
  if (p.probe && lane == 0) {          // [block][wave][5]: read, cluster, y-wait, loop total, whole kernel
    unsigned long long pt_end;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_STAMP(pt_end);
    unsigned long long* o = p.probe + ((size_t)blockIdx.x * 8 + wave) * 5;
    o[0] = pt_r; o[1] = pt_c; o[2] = pt_y; o[3] = pt0 - pt_begin; o[4] = pt_end - pt_begin;
  }
#endif
}

template <typename Kn>
int launch_pp(Kn kernel, const KParams& kp, dim3 grid, hipStream_t st) {
  g_last_kernel = "pp";
  favit_ensure_dyn_lds(reinterpret_cast<const void*>(kernel), PP_LDS);
  hipLaunchKernelGGL(kernel, grid, dim3(PP_THREADS), PP_LDS, st, kp);
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

// --------------------------------------------------------------------------------------
// bf16 kernel "p7": 256x256 tile, SIXTEEN waves (4x4, 64x64 each), BK = 32, three 32-KiB stages
// (96 KiB -> one 1024-thread workgroup per CU: the same 4 waves/SIMD and 128 VGPRs as two p4
// workgroups).  The operand stream per flop drops by a third (L2->LDS intensity 128 vs 85 flop/B),
// which is what bounds the p4 main loop; the price is that all sixteen waves share one barrier and
// nothing overlaps the epilogue.  Measured (tools/gemm_bench.py, BASE=1): +4..9 % on NT problems with
// K >= 768 (8192^3: 1011 -> 1106 TF; ViT-Base qkv / fc1 / fc2), nothing at K = 384 (the epilogue is
// half of those launches), and -10 % with a mn-major B operand -- so it serves NT, K >= 512, N a
// multiple of 256 only.  Epilogue: wave-private LDS transpose in 16-row passes (69.6 KiB).
// --------------------------------------------------------------------------------------
constexpr int P7_THREADS = 1024;
constexpr int P7_BN = 256;
constexpr int P7_B_BYTES = 256 * 64;                 // 16 KiB
constexpr int P7_STAGE = P4_A_BYTES + P7_B_BYTES;    // 32 KiB
constexpr int P7_LDS = 3 * P7_STAGE;                 // 98304
constexpr int WEPI_Q_BYTES = 16 * WEPI_LD * 4;       // 4352 B per wave

template <typename OutT>
__global__ __launch_bounds__(P7_THREADS) void gemm_bf16_p7_kernel(KParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const long m0 = (long)(tile / p.tiles_n) * P4_BM;
  const long n0 = (long)(tile % p.tiles_n) * P7_BN;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* Bm = reinterpret_cast<const bf16_t*>(p.B);
  OutT* C = reinterpret_cast<OutT*>(p.C);
#ifdef FAVIT_PROBE
  const int nk = p.dbg == 2 ? 0 : (int)(p.K / P4_BK);
#else
  const int nk = (int)(p.K / P4_BK);
#endif

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // one 1-KiB piece of A (16 pieces) and one of B (16 pieces) per wave and stage
  const char* sa = reinterpret_cast<const char*>(glds_src32<true>(A, p.lda, m0, p.M, 0, wave, lane));
  const char* sb = reinterpret_cast<const char*>(glds_src32<true>(Bm, p.ldb, n0, p.N, 0, wave, lane));
  const int da = wave * 1024, db = P4_A_BYTES + wave * 1024;
  const long b_step = 64;
  auto issue = [&](int buf) {
    char* st = smem + buf * P7_STAGE;
    __builtin_amdgcn_global_load_lds((gptr_t)sa, (lptr_t)(st + da), 16, 0, 0);
    sa += 64;
    __builtin_amdgcn_global_load_lds((gptr_t)sb, (lptr_t)(st + db), 16, 0, 0);
    sb += b_step;
  };

  if (nk > 0) issue(0);
  if (nk > 1) issue(1);
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk) issue(cur >= 1 ? cur - 1 : 2);
    const char* st = smem + cur * P7_STAGE;
    const char* lb = st + P4_A_BYTES;
    const int rb = wc * 64;
    bf16x8 af[4], bfr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = load_frag32<true>(st, wr * 64 + i * 16, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[j] = load_frag32<true>(lb, rb + j * 16, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    cur = cur == 2 ? 0 : cur + 1;
  }
#ifdef FAVIT_PROBE
  if (p.dbg == 1) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678f) C[0] = from_f32<OutT>(s);
    return;
  }
#endif
  __syncthreads();        // every wave is done with the stage buffers; LDS becomes wave-private scratch
  float* wl = reinterpret_cast<float*>(smem + wave * WEPI_Q_BYTES);
  const long mb = m0 + wr * 64, nb = n0 + wc * 64;
  wave_epilogue_rows<bf16_t, OutT, 0, 1>(p, acc, C, mb, nb, lane, wl, true, p.alpha);
  wave_epilogue_rows<bf16_t, OutT, 1, 1>(p, acc, C, mb, nb, lane, wl, true, p.alpha);
  wave_epilogue_rows<bf16_t, OutT, 2, 1>(p, acc, C, mb, nb, lane, wl, true, p.alpha);
  wave_epilogue_rows<bf16_t, OutT, 3, 1>(p, acc, C, mb, nb, lane, wl, true, p.alpha);
}

template <typename Kn>
int launch_p7(Kn kernel, const KParams& kp, dim3 grid, hipStream_t st) {
  g_last_kernel = "p7";
  favit_ensure_dyn_lds(reinterpret_cast<const void*>(kernel), P7_LDS);
  hipLaunchKernelGGL(kernel, grid, dim3(P7_THREADS), P7_LDS, st, kp);
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

// --------------------------------------------------------------------------------------
// bf16 kernel "s64": 64x128 tile, 4 waves (2x2, 32x64 each), BK = 64, three 24-KiB stages, for problems
// that give the 128x128 kernels fewer than one workgroup per CU (the 17-token SPPP configurations,
// ViT-Tiny, heads): twice the workgroups, a three-deep DMA ring instead of the double buffer, and the
// barrier-free wave-private epilogue.  A is k-major; B either layout.  Such launches are latency-bound
// (6 K-steps at K = 384), so what counts is the length of one workgroup's dependency chain.
// (The opposite choice -- the 256x128 kernel on these problems -- measured 4.68 -> 5.44 ms per SPPP step.)
// --------------------------------------------------------------------------------------
constexpr int S64_BM = 64;
constexpr int S64_LDS = 3 * S64_STAGE;                   // 73728

template <bool BKM, typename OutT>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_s64_kernel(KParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int z = blockIdx.z;
  const long m0 = (long)(tile / p.tiles_n) * S64_BM;
  const long n0 = (long)(tile % p.tiles_n) * BN;
  const long zo = z / p.batch_inner, zi = z % p.batch_inner;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A) + zo * p.sAo + zi * p.sAi;
  const bf16_t* Bm = reinterpret_cast<const bf16_t*>(p.B) + zo * p.sBo + zi * p.sBi;
  OutT* C = reinterpret_cast<OutT*>(p.C) + zo * p.sCo + zi * p.sCi;
  const int nk = (int)(p.K / BK16);

  f32x4 acc[4][4];                        // rows 0..31 of the wave tile live in acc[0..1][*]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // 6 one-KiB pieces per wave and stage: 2 of A (8 pieces = 64 rows), 4 of B (16 pieces)
  const bf16_t* sa[2];
  const bf16_t* sb[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) sa[j] = glds_src<true>(A, p.lda, m0, p.M, 0, wave * 2 + j, lane);
#pragma unroll
  for (int j = 0; j < 4; ++j) sb[j] = glds_src<BKM>(Bm, p.ldb, n0, p.N, 0, wave * 4 + j, lane);
#ifdef FAVIT_PROBE
  // dbg 0x4000 (timing only, wrong results): only HALF of the B pieces are streamed -- what a k-step of a 64x64 tile
  // (16 KB instead of 24 KB per stage) would cost in this kernel's structure
  const bool half_b = (p.dbg & 0x4000) != 0;
#else
  constexpr bool half_b = false;
#endif
  auto issue = [&](int buf) {
    char* st = smem + buf * S64_STAGE;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __builtin_amdgcn_global_load_lds((gptr_t)sa[j], (lptr_t)(st + (wave * 2 + j) * 1024), 16, 0, 0);
      sa[j] += BK16;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (half_b && j >= 2) continue;
      __builtin_amdgcn_global_load_lds((gptr_t)sb[j], (lptr_t)(st + S64_A_BYTES + (wave * 4 + j) * 1024), 16, 0, 0);
      sb[j] += BKM ? BK16 : BK16 * p.ldb;
    }
  };
  if (nk > 0) issue(0);
  if (nk > 1) issue(1);
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) { if (half_b) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk) issue(cur >= 1 ? cur - 1 : 2);
    const char* la = smem + cur * S64_STAGE;
    const char* lb = la + S64_A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[2], bfr[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = load_frag16<true>(la, wr * 32 + i * 16, ks, lane);
      load_frags4<BKM, false>(lb, wc * 64, ks, lane, bfr);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    cur = cur == 2 ? 0 : cur + 1;
  }
  __syncthreads();        // every wave is done with the stage buffers; LDS becomes wave-private scratch
  wave_epilogue_rows<bf16_t, OutT, 0, 2>(p, acc, C, m0 + wr * 32, n0 + wc * 64, lane,
                                         reinterpret_cast<float*>(smem + wave * WEPI_BYTES), true, p.alpha);
}

// --------------------------------------------------------------------------------------
// bf16 kernel "s64k2": the 64x128 tile of s64 with the K loop split between TWO groups of four waves (waves w and
// w + 4 share a SIMD): group g takes k-steps g, g + 2, ... through a ring of its own, so two k-steps of the tile are
// in flight on the CU at any time, and the halves are added through LDS before the (shared) epilogue.  For launches
// with no more tiles than CUs (the N = 384 projections at 17 tokens per image: 102 tiles): such a launch is ONE
// latency-bound workgroup per CU whose time is 6.6 us + 0.33 us per k-step -- and that per-k-step cost is the
// dependency chain barrier -> fragment reads -> MFMAs of a lone wave per SIMD, not bytes: streaming only half of the B
// operand (probe build, FAVIT_GEMM_DBG=0x4000) leaves every launch unchanged (fc2 at 2,176 tokens: 15.8 us both ways).
// Deterministic: wave (g, w) ends up with rows 16 g .. 16 g + 15 of wave tile w as own + partner, one fp32 addition.
// --------------------------------------------------------------------------------------
constexpr int S64K2_THREADS = 512;
constexpr int S64K2_LDS = 2 * S64_LDS;                   // 147456

template <bool BKM, typename OutT>
__global__ __launch_bounds__(S64K2_THREADS) void gemm_bf16_s64k2_kernel(KParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave8 >> 2, wave = wave8 & 3;          // k-step parity of this wave, wave inside its group
  const int wr = wave >> 1, wc = wave & 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const long m0 = (long)(tile / p.tiles_n) * S64_BM;
  const long n0 = (long)(tile % p.tiles_n) * BN;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* Bm = reinterpret_cast<const bf16_t*>(p.B);
  OutT* C = reinterpret_cast<OutT*>(p.C);
  const int nk = (int)(p.K / BK16);
  const int nit = (nk + 1) >> 1;                         // iterations of BOTH groups (one barrier each)
  const int mine = (nk - grp + 1) >> 1;                  // k-steps this group really has
  char* ring = smem + grp * S64_LDS;

  f32x4 acc[4][4];                        // rows 0..31 of the wave tile live in acc[0..1][*]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bf16_t* sa[2];
  const bf16_t* sb[4];
  const long k0 = (long)grp * BK16;
#pragma unroll
  for (int j = 0; j < 2; ++j) sa[j] = glds_src<true>(A, p.lda, m0, p.M, k0, wave * 2 + j, lane);
#pragma unroll
  for (int j = 0; j < 4; ++j) sb[j] = glds_src<BKM>(Bm, p.ldb, n0, p.N, k0, wave * 4 + j, lane);
  auto issue = [&](int buf) {
    char* st = ring + buf * S64_STAGE;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __builtin_amdgcn_global_load_lds((gptr_t)sa[j], (lptr_t)(st + (wave * 2 + j) * 1024), 16, 0, 0);
      sa[j] += 2 * BK16;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_global_load_lds((gptr_t)sb[j], (lptr_t)(st + S64_A_BYTES + (wave * 4 + j) * 1024), 16, 0, 0);
      sb[j] += BKM ? 2 * BK16 : 2 * BK16 * p.ldb;
    }
  };
  if (mine > 0) issue(0);
  if (mine > 1) issue(1);
  int cur = 0;
  for (int it = 0; it < nit; ++it) {
    if (it + 1 < mine) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // (all eight waves: the two groups advance in lock-step)
    if (it + 2 < mine) issue(cur >= 1 ? cur - 1 : 2);
    if (it < mine) {
      const char* la = ring + cur * S64_STAGE;
      const char* lb = la + S64_A_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[2], bfr[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = load_frag16<true>(la, wr * 32 + i * 16, ks, lane);
        load_frags4<BKM, false>(lb, wc * 64, ks, lane, bfr);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      }
    }
    cur = cur == 2 ? 0 : cur + 1;
  }
  __syncthreads();        // every wave is done with the rings; LDS becomes the exchange buffer + wave-private scratch
  // wave (grp, wave) keeps rows 16 grp .. 16 grp + 15 of wave tile `wave`: it hands the OTHER 16 rows to its partner
  float* xch = reinterpret_cast<float*>(smem);           // [8 waves][4 j][64 lanes] f32x4 = 32 KiB
  {
    f32x4* mine_out = reinterpret_cast<f32x4*>(xch) + ((size_t)wave8 * 4) * 64 + lane;
#pragma unroll
    for (int j = 0; j < 4; ++j) mine_out[j * 64] = grp ? acc[0][j] : acc[1][j];
  }
  __syncthreads();
  f32x4 fin[4][4];
  {
    const f32x4* theirs = reinterpret_cast<const f32x4*>(xch) + ((size_t)(wave8 ^ 4) * 4) * 64 + lane;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 t = theirs[j * 64], o = grp ? acc[1][j] : acc[0][j];
      fin[0][j] = (f32x4){o[0] + t[0], o[1] + t[1], o[2] + t[2], o[3] + t[3]};
    }
  }
  __syncthreads();        // the exchange buffer is dead: wave-private epilogue scratch may overwrite it
  wave_epilogue_rows<bf16_t, OutT, 0, 1>(p, fin, C, m0 + wr * 32 + grp * 16, n0 + wc * 64, lane,
                                         reinterpret_cast<float*>(smem + wave8 * WEPI_Q_BYTES), true, p.alpha);
}

template <typename Kn>
int launch_s64k2(Kn kernel, const KParams& kp, dim3 grid, hipStream_t st) {
  g_last_kernel = "s64k2";
  favit_ensure_dyn_lds(reinterpret_cast<const void*>(kernel), S64K2_LDS);
  hipLaunchKernelGGL(kernel, grid, dim3(S64K2_THREADS), S64K2_LDS, st, kp);
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

template <typename Kn>
int launch_s64(Kn kernel, const KParams& kp, dim3 grid, hipStream_t st) {
  g_last_kernel = "s64";
  favit_ensure_dyn_lds(reinterpret_cast<const void*>(kernel), S64_LDS);
  hipLaunchKernelGGL(kernel, grid, dim3(NTHREADS), S64_LDS, st, kp);
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

// --------------------------------------------------------------------------------------
// bf16 kernel "s64ln": LayerNorm fused into the A-operand staging of the 64x128-tile kernel, for the short-token
// configurations (17 / 65 tokens per image: a block's forward is seven launches of 5-15 us, two of them the LayerNorms
// in front of the qkv and fc1 projections).  The workgroup reads its 64 rows of the fp32 residual stream, normalises
// them (two-pass mean / variance as csrc/norm_elem.hip, 16 lanes per row, reductions in the VALU by DPP) and writes
// the bf16 result as the WHOLE-K A panel into LDS in the stage images the fragment reads expect (K = D <= 512); only
// the weight tiles go through the DMA ring (three 16-KiB stages; two for D > 256 so that two workgroups still share a
// CU).  The workgroup of tile column 0 also writes xn / mean / rstd for backward (the weight-gradient launch reads xn).
// Every column tile of a row block recomputes the normalisation (9-12x 98 KB from L2): cheaper than one more launch.
// --------------------------------------------------------------------------------------
struct LnFuse {
  const float* x;        // fp32 residual stream, rows of ldx
  long ldx;
  const float* gamma;
  const float* beta;
  bf16_t* xn;            // [M, D] bf16 out (saved for backward)
  float* mean;
  float* rstd;
  float eps;
};

__device__ __forceinline__ float dpp_sum16(float v) {
  v = dpp_sum8(v);
  // row_mirror (0x140): lane i <- lane 15 - i of its row of 16: the other half's sum of eight
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
}

template <typename OutT, int NI, int NSTB>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_s64ln_kernel(KParams p, LnFuse q) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* panel = smem;                                   // NI stage images of the A operand: [64 rows][64 k] bf16
  char* ring = smem + NI * S64_A_BYTES;                 // NSTB stages of the B operand: [128 rows][64 k]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const long m0 = (long)(tile / p.tiles_n) * S64_BM;
  const long n0 = (long)(tile % p.tiles_n) * BN;
  const bf16_t* Bm = reinterpret_cast<const bf16_t*>(p.B);
  OutT* C = reinterpret_cast<OutT*>(p.C);
  constexpr int D = NI * 64;

  const bf16_t* sb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) sb[j] = glds_src<true>(Bm, p.ldb, n0, p.N, 0, wave * 4 + j, lane);
  auto issue = [&](int buf) {
    char* st = ring + buf * OP16_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_global_load_lds((gptr_t)sb[j], (lptr_t)(st + (wave * 4 + j) * 1024), 16, 0, 0);
      sb[j] += BK16;
    }
  };
  issue(0);                                              // the weight stream starts under the LayerNorm
  if (NSTB == 3 && NI > 1) issue(1);

  // ---- LayerNorm of rows m0 .. m0 + 63: wave w owns rows 16 w .. 16 w + 15, four at a time, 16 lanes per row ----
  const int j16 = lane & 15, rsub = lane >> 4;
  const bool write_xn = (n0 == 0) && q.xn != nullptr;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int r = wave * 16 + it * 4 + rsub;
    long m = m0 + r;
    const bool live = m < p.M;
    if (!live) m = p.M - 1;                              // rows past M: a valid row, never stored
    const float* xr = q.x + m * q.ldx;
    float4 v[NI];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      v[i] = *reinterpret_cast<const float4*>(xr + 4 * (j16 + 16 * i));
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mu = dpp_sum16(s) * (1.0f / (float)D);
    float qq = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float a = v[i].x - mu, b = v[i].y - mu, c = v[i].z - mu, d = v[i].w - mu;
      qq += (a * a + b * b) + (c * c + d * d);
    }
    const float rs = rsqrtf(dpp_sum16(qq) * (1.0f / (float)D) + q.eps);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int col = 4 * (j16 + 16 * i);
      const float4 g = *reinterpret_cast<const float4*>(q.gamma + col);
      const float4 be = *reinterpret_cast<const float4*>(q.beta + col);
      bf16x4 o;
      o[0] = (bf16_t)((v[i].x - mu) * rs * g.x + be.x);
      o[1] = (bf16_t)((v[i].y - mu) * rs * g.y + be.y);
      o[2] = (bf16_t)((v[i].z - mu) * rs * g.z + be.z);
      o[3] = (bf16_t)((v[i].w - mu) * rs * g.w + be.w);
      // stage image i (k = 64 i .. 64 i + 63): 128-byte rows, 16-byte chunk kc at kc ^ ((row >> 1) & 7) (load_frag16)
      *reinterpret_cast<bf16x4*>(panel + i * S64_A_BYTES + r * 128 + ((((j16 >> 1)) ^ ((r >> 1) & 7)) << 4) + (j16 & 1) * 8) = o;
      if (write_xn && live) *reinterpret_cast<bf16x4*>(q.xn + m * (long)D + col) = o;
    }
    if (write_xn && live && j16 == 0) {
      q.mean[m] = mu;
      q.rstd[m] = rs;
    }
  }
  __syncthreads();        // the A panel is complete (and: this wave's loads and stores so far have all returned)

  f32x4 acc[4][4];                        // rows 0..31 of the wave tile live in acc[0..1][*]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int cur = 0;
#pragma unroll
  for (int kt = 0; kt < NI; ++kt) {
    if (NSTB == 3 && kt + 1 < NI) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (NSTB == 3) { if (kt + 2 < NI) issue(cur >= 1 ? cur - 1 : 2); }
    else if (kt + 1 < NI) issue(cur ^ 1);
    const char* la = panel + kt * S64_A_BYTES;
    const char* lb = ring + cur * OP16_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[2], bfr[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = load_frag16<true>(la, wr * 32 + i * 16, ks, lane);
      load_frags4<true, false>(lb, wc * 64, ks, lane, bfr);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    cur = NSTB == 3 ? (cur == 2 ? 0 : cur + 1) : (cur ^ 1);
  }
  __syncthreads();        // every wave is done with panel and ring; LDS becomes wave-private scratch
  wave_epilogue_rows<bf16_t, OutT, 0, 2>(p, acc, C, m0 + wr * 32, n0 + wc * 64, lane,
                                         reinterpret_cast<float*>(smem + wave * WEPI_BYTES), true, p.alpha);
}

// Grouped weight-gradient launch: up to GROUP_MAX dW = dY^T.X problems that share the token dimension (the four
// Linear layers of one transformer block -- or, since round 4, of SEVERAL blocks: functional.flush_wgrads collects the
// problems of up to four blocks, or of the whole encoder at short token counts) run as ONE grid.
//   * chunk = consecutive problems with at most 64 tiles together (one block's four problems at D = 384: 63 tiles =
//     the 64 workgroup slots of an XCD);  unit = (chunk, K-split);  every unit is pinned to ONE XCD (host-side
//     longest-first packing into eight queues), so the tiles that stream the same token rows share that XCD's L2;
//   * the number of K-splits is the minimum of a cost model (grouped_plan): rounds x (k-steps + fixed) + slab traffic.
//     More blocks per launch need fewer splits to fill the chip -- four cfg2 blocks: 2 splits instead of 8, a quarter
//     of the slab bytes and of the prologue / epilogue / reduction launches per flop; twelve cfg3 blocks: NO split at
//     all (no slabs, no reduction kernel, the result is one rounding of one fp32 accumulation chain);
//   * nsplit > 1: partial tiles go to per-split slabs with plain stores and grouped_reduce_kernel adds the slabs in
//     split order (deterministic; no fp32 atomics);  nsplit == 1: the tile epilogue writes dW itself (reading the
//     previous value when the problem accumulates).
constexpr int GROUP_MAX = 48;
constexpr int GROUP_XCD_UNITS = 24;     // units in one XCD's queue, at most
struct TnProb {          // 64 bytes
  const void* A;         // dY [T, M], mn-major
  const void* B;         // X  [T, N], mn-major
  float* C;              // dW, or the problem's place in slab 0
  float* rowsum;         // db, or its place in slab 0 (null: no bias gradient)
  int M, N, lda, ldb, ldc, tiles_n, ntiles;
  int flags;             // 1: the direct epilogue adds the previous value of C (accumulate, nsplit == 1); 2: C / ldc allow 16-byte stores
};
struct GroupParams {
  int count, nchunks, nsplit, mode;       // mode 0: fp32 atomics into C, 1: slabs, 2: direct (nsplit == 1)
  long slab_stride;                        // floats between the slabs of consecutive K-splits
  long k_per_split, K;
  unsigned short chunk_tiles[GROUP_MAX];
  unsigned char chunk_first[GROUP_MAX + 1];
  unsigned char xcd_count[8];
  unsigned char xcd_units[8][GROUP_XCD_UNITS];   // unit = split * nchunks + chunk
  TnProb p[GROUP_MAX];
};
static_assert(sizeof(GroupParams) <= 4000, "kernel arguments are limited to 4 KiB");

__global__ __launch_bounds__(P4_THREADS, 4) void gemm_bf16_p4_grouped_tn_kernel(GroupParams gp) {
  const int h = blockIdx.x, xcd = h & 7;
  int idx = h >> 3;
  // this XCD's queue of units, in order: find the unit and the tile inside its chunk (workgroup-uniform scalars)
  const int nu = gp.xcd_count[xcd];
  int j = 0, c = 0, split = 0;
  for (;; ++j) {
    if (j >= nu) return;                       // this XCD's queue is shorter than the longest one (whole workgroup leaves)
    const int u = gp.xcd_units[xcd][j];
    c = u % gp.nchunks;
    split = u / gp.nchunks;
    const int tiles = gp.chunk_tiles[c];
    if (idx < tiles) break;
    idx -= tiles;
  }
  int i = gp.chunk_first[c];
  while (idx >= gp.p[i].ntiles) { idx -= gp.p[i].ntiles; ++i; }
  i = __builtin_amdgcn_readfirstlane(i);              // (all of this is workgroup-uniform: keep it in SGPRs)
  idx = __builtin_amdgcn_readfirstlane(idx);
  split = __builtin_amdgcn_readfirstlane(split);
  const TnProb& q = gp.p[i];
  KParams kp = {};
  kp.A = q.A; kp.B = q.B;
  kp.M = q.M; kp.N = q.N; kp.K = gp.K;
  kp.lda = q.lda; kp.ldb = q.ldb; kp.ldc = q.ldc;
  kp.k_per_split = gp.k_per_split;
  kp.batch_inner = 1;
  kp.a_vec = kp.b_vec = 1;
  kp.tiles_n = q.tiles_n; kp.ntiles = q.ntiles;
  kp.alpha = 1.0f; kp.drop_scale = 1.0f;
  kp.xcd_split = 1;
  if (gp.mode == 1) {
    // slab mode: this split's partial tile goes to its own slab with plain 16-byte stores (no atomics)
    kp.C = q.C + (long)split * gp.slab_stride;
    kp.a_rowsum = q.rowsum ? q.rowsum + (long)split * gp.slab_stride : nullptr;
    kp.c_vec = 1;
    kp.rowsum_store = 1;
  } else if (gp.mode == 2) {
    // one split: the epilogue writes dW itself; an accumulating problem reads the previous value as a residual.
    // The bias gradient is ADDED to its destination by contract: one fp32 atomic per row (a single addend: exact)
    kp.C = q.C;
    kp.a_rowsum = q.rowsum;
    kp.c_vec = (q.flags & 2) ? 1 : 0;
    if (q.flags & 1) { kp.residual = q.C; kp.ld_res = q.ldc; }
  } else {
    kp.C = q.C;
    kp.a_rowsum = q.rowsum;
    kp.atomic = 1;
  }
#ifdef FAVIT_PROBE
  kp.dbg = gp.count < 0 ? 1 : 0;          // (host: FAVIT_GEMM_DBG=1)
#endif
  p4_body<false, false, float>(kp, idx, split, 0);
}

// Second half of the slab-mode split-K: dst = (accumulate ? dst : 0) + sum over the splits' slabs, in split order
// (a fixed summation order: weight gradients are bitwise reproducible, unlike fp32 atomics).
struct ReduceSeg {       // 32 bytes
  float* dst;            // dW [rows, ld] or db [rows]
  long off;              // offset of this segment inside a slab (floats)
  int rows, cols, ld;    // cols = the vector's length, rows = 1, ld = cols for a bias-gradient segment
  int accumulate;
};
struct ReduceParams {
  const float* ws;
  long slab_stride;
  int nsplit, nseg;
  int blk_off[2 * GROUP_MAX + 1];   // first workgroup of every segment (a segment gets one workgroup per 2048 floats)
  ReduceSeg seg[2 * GROUP_MAX];
};
static_assert(sizeof(ReduceParams) <= 4000, "kernel arguments are limited to 4 KiB");
constexpr int REDUCE_PER_BLOCK = 2;     // float4 per thread

__global__ __launch_bounds__(256) void grouped_reduce_kernel(ReduceParams rp) {
  int lo = 0, hi = rp.nseg - 1;         // segment of this workgroup (uniform binary search over <= 96 entries)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)blockIdx.x >= rp.blk_off[mid]) lo = mid; else hi = mid - 1;
  }
  const ReduceSeg& sg = rp.seg[lo];
  const long n4 = (long)sg.rows * sg.cols / 4;               // rows * cols is a multiple of 4 (host)
  const long q0 = (long)((int)blockIdx.x - rp.blk_off[lo]) * (256 * REDUCE_PER_BLOCK) + threadIdx.x;
#pragma unroll
  for (int it = 0; it < REDUCE_PER_BLOCK; ++it) {
    const long q = q0 + it * 256;
    if (q >= n4) break;
    const long e = 4 * q;
    const float* src = rp.ws + sg.off + e;
    // slabs in groups of eight: the loads of a group are all in flight before the first add (with one load per
    // loop trip every slab was a round trip of its own: 3.0 TB/s at ViT-Base, where the 227 MB of slabs no longer
    // sit in the Infinity Cache); the additions keep the split order, so the result is bit-identical
    const long r = e / sg.cols, c = e - r * sg.cols;   // cols % 4 == 0 (or the segment is one contiguous vector)
    float* d = sg.dst + r * sg.ld + c;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sg.accumulate) o = *reinterpret_cast<const float4*>(d);
    float4 acc = *reinterpret_cast<const float4*>(src);
    int sidx = 1;
    for (; sidx + 7 < rp.nsplit; sidx += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(src + (long)(sidx + u) * rp.slab_stride);
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    if (sidx + 6 < rp.nsplit) {                         // 7 left (8 splits)
      float4 v[7];
#pragma unroll
      for (int u = 0; u < 7; ++u) v[u] = *reinterpret_cast<const float4*>(src + (long)(sidx + u) * rp.slab_stride);
#pragma unroll
      for (int u = 0; u < 7; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
      sidx += 7;
    }
    for (; sidx < rp.nsplit; ++sidx) {
      const float4 v = *reinterpret_cast<const float4*>(src + (long)sidx * rp.slab_stride);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (sg.accumulate) { acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
    *reinterpret_cast<float4*>(d) = acc;
  }
}

template <typename Kn>
int launch_p4(Kn kernel, const KParams& kp, dim3 grid, hipStream_t st, const char* name = "p4") {
  g_last_kernel = name;
#ifdef FAVIT_PROBE
  // FAVIT_GEMM_P4_ONE_PER_CU: claim 100 KiB of LDS so that only one workgroup fits a CU (occupancy experiment)
  static const int lds_bytes = getenv("FAVIT_GEMM_P4_ONE_PER_CU") ? 100 * 1024 : P4_LDS;
#else
  constexpr int lds_bytes = P4_LDS;
#endif
  favit_ensure_dyn_lds(reinterpret_cast<const void*>(kernel), lds_bytes);
  hipLaunchKernelGGL(kernel, grid, dim3(P4_THREADS), lds_bytes, st, kp);
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
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

  int tile, split;
  tile_and_split(p, tile, split);
  const long m0 = (long)(tile / p.tiles_n) * BM;
  const long n0 = (long)(tile % p.tiles_n) * BN;
  const int z = blockIdx.z;
  const long zo = z / p.batch_inner, zi = z % p.batch_inner;
  const float* A = reinterpret_cast<const float*>(p.A) + zo * p.sAo + zi * p.sAi;
  const float* Bm = reinterpret_cast<const float*>(p.B) + zo * p.sBo + zi * p.sBi;
  float* C = reinterpret_cast<float*>(p.C) + zo * p.sCo + zi * p.sCi;

  const long kbeg = (long)split * p.k_per_split;
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
  run_epilogue<float, float>(p, epi, m0, n0, C, split == 0, tid);

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

// --------------------------------------------------------------------------------------
// f32 kernel "p4f" (round 4): the p4 structure for exact fp32 -- 256x128 tile, 8 waves (64x64 each, four
// v_mfma_f32_32x32x2_f32 accumulator blocks), BK = 16 (the same 64-byte stage rows, the same 24-KiB stages filled by
// global_load_lds with counted vmcnt, one raw s_barrier per stage, two workgroups per CU), the wave-private epilogue.
// The fp32 MFMA runs at 1/16 of the bf16 rate, so this kernel is MFMA-bound by construction: a stage is 32 MFMAs of 64
// cycles per wave against 24 KiB of DMA, and a wave reads 4 KiB of fragments per 1,024 MFMA cycles.  It replaces the
// register-staged 128x128 kernel of round 1 (__syncthreads per stage, scalar LDS writes: 0.30-0.41 of the 157 TF peak)
// for the large GEMMs of the fp32 parity mode.
//   k-major image : p4's ([rows][16 k], 64-B rows, 16-B chunk c at c ^ ksw32(row)); a lane reads ONE 16-byte chunk per
//                   block row and 8-deep k group: lanes 0-31 chunk 2t, lanes 32-63 chunk 2t+1, i.e. the MFMA's two
//                   k-slots are k = 8t + s and 8t + 4 + s in step s (A and B permute k identically);
//   mn-major image: [16 k][W] floats (W = 256 / 128), k-rows whose bit 2 is set rotated by 32 words so that the two
//                   lane halves of a fragment read (k-rows 8t + s and 8t + 4 + s) hit different banks.
// --------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ const char* glds_src_f32_mn(const char* base, long ld, long i0, long I, long k0, int q, int lane) {
  constexpr int LPR = W / 4;                           // lanes (16-byte chunks) per k-row
  const int krow = q * (64 / LPR) + lane / LPR;
  const int pc = lane % LPR;
  const int c = (pc - 8 * ((krow >> 2) & 1)) & (LPR - 1);
  long i = i0 + c * 4;
  const long imax = (I - 4) & ~3L;
  i = i < imax ? i : imax;
  return base + ((k0 + krow) * ld + i) * 4;
}

template <bool AK, bool BKM, int TBM, int NW>
__device__ __forceinline__ void p4f_body(const KParams& p, int tile, int split) {
  static_assert(TBM == 256 || TBM == 128, "256x128 tiles, or 128x128 ones where those balance better");
  // NW = 4 (2x2 waves of 128 / 64 rows, 174-182 VGPRs, two waves per SIMD from two workgroups) was measured: 8192^3
  // 139.8 TF against 132.5 with eight waves, but every shape of the path slower (fc2 forward 587 us against 501, qkv 531
  // against 445: four waves take twice as long over the epilogue); the bare MFMA loop (tools/mfma_peak.py) holds
  // 154.6 TF with one or two waves per SIMD and 123.6 TF with four, v_mfma_f32_16x16x4_f32 155.0 TF with four -- but a
  // 16x16x4 form of this loop (one 16-byte chunk per 16-row block and stage, the bf16 epilogue shared) measured the same
  // or slower inside the kernel: qkv / fc1 forward 429.5 us against 384.5 in the step, 8192^3 129.6 TF against 132.5.
  static_assert(NW == 8 || NW == 4, "eight waves as 4x2, or four as 2x2 with twice the rows per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WR = TBM / (NW / 2);                  // rows per wave (the wave tile is WR x 64)
  constexpr int NI = WR / 32;                         // 32-row accumulator blocks per wave
  constexpr int PA = TBM / 16 / NW, PB = 8 / NW;      // 1-KiB DMA pieces per wave and stage
  constexpr int A_BYTES = TBM * 64;
  constexpr int STAGE = A_BYTES + P4_B_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int h = lane >> 5, l31 = lane & 31;

  const long m0 = (long)(tile / p.tiles_n) * TBM;
  const long n0 = (long)(tile % p.tiles_n) * BN;
  const char* A = reinterpret_cast<const char*>(p.A);
  const char* Bm = reinterpret_cast<const char*>(p.B);
  float* C = reinterpret_cast<float*>(p.C);
  const long kbeg = (long)split * p.k_per_split;
  const long kend = min(p.K, kbeg + p.k_per_split);
  const int nk = (int)((kend - kbeg) / BK32);

  f32x16 acc[NI][2];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const bool do_rowsum = (!AK) && (p.a_rowsum != nullptr) && (n0 == 0) && (wc == 0);
  float rs[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) rs[i] = 0.f;

  const char* sa[PA];
  const char* sb[PB];
#pragma unroll
  for (int j = 0; j < PA; ++j) {
    const int qa = wave * PA + j;
    if (AK) sa[j] = reinterpret_cast<const char*>(glds_src32<true>(reinterpret_cast<const bf16_t*>(A), 2 * p.lda, m0, p.M, 2 * kbeg, qa, lane));
    else sa[j] = glds_src_f32_mn<TBM>(A, p.lda, m0, p.M, kbeg, qa, lane);
  }
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    const int qb = wave * PB + j;
    if (BKM) sb[j] = reinterpret_cast<const char*>(glds_src32<true>(reinterpret_cast<const bf16_t*>(Bm), 2 * p.ldb, n0, p.N, 2 * kbeg, qb, lane));
    else sb[j] = glds_src_f32_mn<128>(Bm, p.ldb, n0, p.N, kbeg, qb, lane);
  }
  const long a_step = AK ? 64 : (long)BK32 * p.lda * 4;      // bytes per stage
  const long b_step = BKM ? 64 : (long)BK32 * p.ldb * 4;
  auto issue = [&](int buf) {
    char* st = smem + buf * STAGE;
#pragma unroll
    for (int j = 0; j < PA; ++j) {
      __builtin_amdgcn_global_load_lds((gptr_t)sa[j], (lptr_t)(st + (wave * PA + j) * 1024), 16, 0, 0);
      sa[j] += a_step;
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      __builtin_amdgcn_global_load_lds((gptr_t)sb[j], (lptr_t)(st + A_BYTES + (wave * PB + j) * 1024), 16, 0, 0);
      sb[j] += b_step;
    }
  };

  if (nk > 0) issue(0);
  if (nk > 1) issue(1);
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {                                  // the newest stage (PA + PB pieces of this wave) may still fly
      if constexpr (PA + PB == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if constexpr (PA + PB == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if constexpr (PA + PB == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk) issue(cur >= 1 ? cur - 1 : 2);
    const char* la = smem + cur * STAGE;
    const char* lb = la + A_BYTES;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x4 a[NI], b[2];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        if (AK) {
          const int row = wr * WR + i * 32 + l31;
          a[i] = *reinterpret_cast<const f32x4*>(la + row * 64 + (((2 * t + h) ^ ksw32(row)) << 4));
        } else {
          const float* f = reinterpret_cast<const float*>(la) + (8 * t + 4 * h) * TBM + ((wr * WR + i * 32 + l31 + 32 * h) & (TBM - 1));
          a[i] = (f32x4){f[0], f[TBM], f[2 * TBM], f[3 * TBM]};
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (BKM) {
          const int row = wc * 64 + j * 32 + l31;
          b[j] = *reinterpret_cast<const f32x4*>(lb + row * 64 + (((2 * t + h) ^ ksw32(row)) << 4));
        } else {
          const float* f = reinterpret_cast<const float*>(lb) + (8 * t + 4 * h) * 128 + ((wc * 64 + j * 32 + l31 + 32 * h) & 127);
          b[j] = (f32x4){f[0], f[128], f[256], f[384]};
        }
      }
      // (s_setprio 1 around the 8 NI MFMAs of a group, so that the arbiter stays with one wave: 8192^3 132.5 -> 108.6 TF)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[0][s], a[i][s], acc[i][0], 0, 0, 0);
          acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[1][s], a[i][s], acc[i][1], 0, 0, 0);
        }
      if (do_rowsum) {                              // (wave-uniform) bias gradient: the column sums of the mn-major A
#pragma unroll
        for (int i = 0; i < NI; ++i) rs[i] += (a[i][0] + a[i][1]) + (a[i][2] + a[i][3]);
      }
    }
    cur = cur == 2 ? 0 : cur + 1;
  }
  if (do_rowsum) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float tot = rs[i] + __shfl_xor(rs[i], 32);
      const long m = m0 + wr * WR + i * 32 + l31;
      if (h == 0 && m < p.M) {
        if (p.rowsum_store) p.a_rowsum[m] = tot;
        else atomicAdd(p.a_rowsum + m, tot);
      }
    }
  }
  __syncthreads();        // every wave is done with the stage buffers; LDS becomes wave-private scratch
  float* wl = reinterpret_cast<float*>(smem + wave * WEPI_BYTES);
  // deposit a block row (32 rows x 64 columns): a lane holds row (lane & 31), columns 32 j + 8 rg + 4 (lane >> 5) + e
  auto deposit = [&](const f32x16& c0, const f32x16& c1) __attribute__((always_inline)) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const f32x4 v0 = {c0[4 * rg], c0[4 * rg + 1], c0[4 * rg + 2], c0[4 * rg + 3]};
      const f32x4 v1 = {c1[4 * rg], c1[4 * rg + 1], c1[4 * rg + 2], c1[4 * rg + 3]};
      *reinterpret_cast<f32x4*>(wl + l31 * WEPI_LD + 8 * rg + 4 * h) = v0;
      *reinterpret_cast<f32x4*>(wl + l31 * WEPI_LD + 32 + 8 * rg + 4 * h) = v1;
    }
  };
  const long mb = m0 + wr * WR, nb = n0 + wc * 64;
  deposit(acc[0][0], acc[0][1]);
  wave_epilogue_tail<float, float, 0, 2>(p, C, mb, nb, lane, wl, split == 0, p.alpha);
  if constexpr (NI >= 2) {
    deposit(acc[1][0], acc[1][1]);
    wave_epilogue_tail<float, float, 2, 2>(p, C, mb, nb, lane, wl, split == 0, p.alpha);
  }
  if constexpr (NI == 4) {
    deposit(acc[2][0], acc[2][1]);
    wave_epilogue_tail<float, float, 4, 2>(p, C, mb, nb, lane, wl, split == 0, p.alpha);
    deposit(acc[3][0], acc[3][1]);
    wave_epilogue_tail<float, float, 6, 2>(p, C, mb, nb, lane, wl, split == 0, p.alpha);
  }
}

template <bool AK, bool BKM, int TBM>
__global__ __launch_bounds__(P4_THREADS, 4) void gemm_f32_p4_kernel(KParams p) {
  // (A phase stagger -- the first-round workgroup in a CU's second slot starting 0.5 / 1 / 1.5 half tiles late, so that
  // the two workgroups of a CU never sit in their epilogues together -- changed no shape by more than 1 %: measured,
  // removed.)
  int tile, split;
  tile_and_split(p, tile, split);
  p4f_body<AK, BKM, TBM, 8>(p, tile, split);
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
  g_last_kernel = "t128";
  favit_ensure_dyn_lds(reinterpret_cast<const void*>(kernel), LDS_BYTES);
  hipLaunchKernelGGL(kernel, grid, dim3(NTHREADS), LDS_BYTES, st, kp);
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// Kernel-selection switches (tools/README.md), read once per process.  Every selectable kernel computes the
// same result; the work-skipping probe switch (FAVIT_GEMM_DBG) exists only in the `make probe` build.
struct GemmKnobs {
  int dbg, store_policy;
  bool force128, no_p4, no_p7, no_s64, no_s64k2, no_pp, no_quarter, no_f32p4;
  long quarter_max;
  GemmKnobs() {
    const char* e;
#ifdef FAVIT_PROBE
    dbg = (e = getenv("FAVIT_GEMM_DBG")) ? atoi(e) : 0;
#else
    dbg = 0;
#endif
    store_policy = (e = getenv("FAVIT_GEMM_STORE")) ? atoi(e) : 1;
    force128 = getenv("FAVIT_GEMM_TILE128") != nullptr;
    no_p4 = getenv("FAVIT_GEMM_NO_P4") != nullptr;
    no_p7 = getenv("FAVIT_GEMM_NO_P7") != nullptr;
    no_s64 = getenv("FAVIT_GEMM_NO_S64") != nullptr;
    no_s64k2 = getenv("FAVIT_GEMM_NO_S64K2") != nullptr;
    no_f32p4 = getenv("FAVIT_GEMM_NO_F32P4") != nullptr;
    no_pp = getenv("FAVIT_GEMM_NO_PP") != nullptr;
    no_quarter = getenv("FAVIT_GEMM_NO_QUARTER") != nullptr;
    quarter_max = (e = getenv("FAVIT_GEMM_QUARTER_MAX")) ? atol(e) : 128;       // tail rounds up to 25 % of the slots
  }
};
static const GemmKnobs& knobs() {
  static const GemmKnobs k;
  return k;
}

#ifdef FAVIT_PROBE
unsigned long long* g_probe_buffer = nullptr;
#endif

}  // namespace

#ifdef FAVIT_PROBE
// probe build only (not declared in include/favit.h): device buffer of >= grid * 8 * 5 uint64 for the pp stamps
extern "C" void favit_probe_buffer(void* buf) { g_probe_buffer = reinterpret_cast<unsigned long long*>(buf); }
#endif

extern "C" const char* favit_gemm_last_kernel(void) { return g_last_kernel; }

extern "C" int favit_gemm(const favit_gemm_t* g, void* stream) {
  if (!g || !g->A || !g->B || !g->C) return FAVIT_ERR_INVALID;
  if (g->M <= 0 || g->N <= 0 || g->K < 0) return FAVIT_ERR_INVALID;
  if (g->in_dtype != FAVIT_F32 && g->in_dtype != FAVIT_BF16 && g->in_dtype != FAVIT_FP8) return FAVIT_ERR_INVALID;
  if (g->out_dtype != FAVIT_F32 && g->out_dtype != FAVIT_BF16) return FAVIT_ERR_INVALID;
  const bool fp8 = g->in_dtype == FAVIT_FP8;
  if (fp8) {
    if (!g->a_kmajor || !g->b_kmajor || g->a_rowsum || (g->batch > 1)) return FAVIT_ERR_UNSUPPORTED;
    if (g->fp8_fmt & ~1) return FAVIT_ERR_UNSUPPORTED;            // B must be e4m3
    if (g->K <= 0 || (g->K % 64) != 0) return FAVIT_ERR_UNSUPPORTED;
    if (!aligned(g->A, 16) || !aligned(g->B, 16) || (g->lda % 16) != 0 || (g->ldb % 16) != 0) return FAVIT_ERR_ALIGN;
  }
  if (g->in_dtype == FAVIT_F32 && g->out_dtype != FAVIT_F32) return FAVIT_ERR_UNSUPPORTED;
  if (g->act < FAVIT_ACT_NONE || g->act > FAVIT_ACT_MULAUX) return FAVIT_ERR_INVALID;
  if ((g->act == FAVIT_ACT_DGELU || g->act == FAVIT_ACT_MULAUX) && !g->aux_in) return FAVIT_ERR_INVALID;
  if (g->act == FAVIT_ACT_GELU_SAVEGRAD && !g->aux_out) return FAVIT_ERR_INVALID;
  if (g->a_rowsum && g->a_kmajor) return FAVIT_ERR_UNSUPPORTED;
  const int batch = g->batch > 0 ? g->batch : 1;
  const int batch_inner = g->batch_inner > 0 ? g->batch_inner : 1;
  if (g->a_rowsum && batch != 1) return FAVIT_ERR_UNSUPPORTED;
  hipStream_t st = as_stream(stream);

  const int bk = g->in_dtype == FAVIT_F32 ? BK32 : BK16;
  const long tiles_m = (g->M + BM - 1) / BM, tiles_n = (g->N + BN - 1) / BN;
  const long tiles = tiles_m * tiles_n * batch;
  long splits = g->split_k;
  const bool can_split = (g->out_dtype == FAVIT_F32) && g->act == FAVIT_ACT_NONE && !g->aux_out &&
                         !(g->dropout_p > 0.f);
  if (splits <= 0) {
    splits = 1;
    // automatic split-K only for the weight-gradient shape (both operands mn-major, K = tokens):
    // forward / input-gradient GEMMs stay single-pass and therefore bitwise deterministic.
    if (can_split && ((!g->a_kmajor && !g->b_kmajor) || fp8) && tiles < 256 && g->K >= 8 * bk) {
      splits = (512 + tiles - 1) / tiles;
      const long max_splits = g->K / (4 * bk);
      if (splits > max_splits) splits = max_splits;
      if (splits < 1) splits = 1;
    }
  }
  // exact-fp32 problems the DMA kernel takes (p4f; alignment is checked at the dispatch below).  It is MFMA-bound, so
  // what counts is how evenly the work falls on the 256 CUs and how many rows of the last tile row are padding:
  // 256- or 128-row tiles by that measure, and a weight gradient is split into as many k-ranges as fill 512 slots.
  const bool f32p4_shape = g->in_dtype == FAVIT_F32 && !fp8 && batch == 1 && !knobs().no_f32p4 && (g->K % BK32) == 0 &&
                           g->K >= 4 * BK32 && (g->a_kmajor || (g->M >= 4 && (g->M % 4) == 0)) &&
                           (g->b_kmajor || (g->N >= 4 && (g->N % 4) == 0));
  int f32_tbm = 256;
  long t4f = ((g->M + 255) / 256) * tiles_n;
  if (f32p4_shape) {
    const bool auto_split = g->split_k <= 0 && splits > 1;
    double best = -1.0;
    for (int tbm = 256; tbm >= 128; tbm -= 128) {
      const long tm = (g->M + tbm - 1) / tbm, t = tm * tiles_n;
      long sp = splits;
      if (auto_split) {
        sp = (512 + t / 2) / t;                                               // 72 KiB of LDS either way: 2 workgroups per CU
        const long max_splits = g->K / (4 * 64);
        sp = sp > max_splits ? max_splits : sp;
        sp = sp < 1 ? 1 : sp;
      }
      const double per_cu = (double)t * sp / 256.0;                           // tiles per CU: the busiest CU takes the ceiling
      const double balance = per_cu / (double)(long)(per_cu + 0.999999);
      const double useful = (double)g->M / (double)(tm * tbm);
      const double score = balance * useful * (tbm == 256 ? 1.02 : 1.0);      // (tie: the larger tile)
      if (score > best) { best = score; f32_tbm = tbm; t4f = t; if (auto_split) splits = sp; }
    }
  }
  if (splits > 1 && !can_split) return FAVIT_ERR_UNSUPPORTED;
  // p4-eligible weight-gradient GEMM: pick splits = 8*s (one group of splits per XCD) that fills
  // the 64 workgroup slots of an XCD (32 CUs x 2) best, s <= 4 to bound the atomic traffic.
  bool xcd_split = false;
  if (g->split_k <= 0 && splits > 1 && (g->in_dtype == FAVIT_BF16 || fp8) && batch == 1 && g->M >= 256) {
    const long t4 = ((g->M + 255) / 256) * tiles_n;
    double best = -1.0;
    long best_s = 1;
    for (long s = 1; s <= 4; ++s) {
      const long w = t4 * s;
      const double util = (double)w / (double)(((w + 63) / 64) * 64);
      if (util > best + 0.02) { best = util; best_s = s; }
    }
    if (g->K / (8 * best_s) >= 4 * 64) { splits = 8 * best_s; xcd_split = true; }
  }
  long kps = (g->K + splits - 1) / splits;
  kps = ((kps + 63) / 64) * 64;
  if (kps <= 0) kps = 64;
  if (!xcd_split) {
    splits = (g->K + kps - 1) / kps;
    if (splits < 1) splits = 1;
  } else if ((splits - 1) * kps >= g->K) {
    xcd_split = false;                  // a split would be empty: fall back to the plain mapping
    splits = (g->K + kps - 1) / kps;
  }
  const int atomic = (splits > 1 || g->accumulate) ? 1 : 0;
  if (atomic && g->out_dtype != FAVIT_F32) return FAVIT_ERR_UNSUPPORTED;

  KParams kp;
  memset(&kp, 0, sizeof(kp));
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
  kp.ntiles = (int)(tiles_m * tiles_n);
  kp.xcd_split = 0;
  kp.q_block0 = 0;
  kp.q_tile0 = 0;
  kp.alpha = g->alpha;
  kp.scale_a = fp8 ? g->scale_a : nullptr;
  kp.scale_b = fp8 ? g->scale_b : nullptr;
#ifdef FAVIT_PROBE
  kp.dbg = knobs().dbg;
  kp.probe = g_probe_buffer;
#endif
  // epilogue outputs / residual / aux reads are touched once: non-temporal keeps them from evicting
  // the operand panels the co-resident workgroups share in L2 (fc2: 140 -> 114 us)
  kp.store_policy = knobs().store_policy;
  kp.rowsum_store = 0;
  if (g->dropout_p < 0.f || g->dropout_p >= 1.f) return FAVIT_ERR_INVALID;
  kp.drop_thresh = dropout_threshold(g->dropout_p);
  kp.drop_scale = 1.0f / (1.0f - g->dropout_p);
  kp.drop_seed = g->dropout_seed;
  kp.drop_epoch = favit_dropout_epoch_ptr_();
  if (kp.drop_thresh && (splits > 1 || batch != 1)) return FAVIT_ERR_UNSUPPORTED;

  const int in_vec = fp8 ? 16 : (g->in_dtype == FAVIT_BF16 ? 8 : 4);   // elements per 16-B load
  auto strides_ok = [&](long so, long si, int v) { return batch == 1 || ((so % v) == 0 && (si % v) == 0); };
  kp.a_vec = aligned(g->A, 16) && (g->lda % in_vec) == 0 && strides_ok(g->sAo, g->sAi, in_vec);
  kp.b_vec = aligned(g->B, 16) && (g->ldb % in_vec) == 0 && strides_ok(g->sBo, g->sBi, in_vec);
  const size_t osz = g->out_dtype == FAVIT_BF16 ? 2 : 4;
  const size_t isz = g->in_dtype == FAVIT_F32 ? 4 : 2;            // aux_in is bf16 beside fp8 operands
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

  if (fp8) {
    // every fp8 problem runs the 256x128 DMA kernel (rows past M / N are clamped, the epilogue guards them)
    const long t8 = ((g->M + 255) / 256) * tiles_n;
    dim3 grid8((unsigned)t8, (unsigned)splits, 1u);
    kp.ntiles = (int)t8;
    if (xcd_split) {
      kp.xcd_split = 1;
      grid8 = dim3((unsigned)(t8 * splits), 1u, 1u);
    }
    if ((kps % 64) != 0) return FAVIT_ERR_UNSUPPORTED;
    const bool a_bf8 = (g->fp8_fmt & 1) != 0;
    if (g->out_dtype == FAVIT_BF16) {
      if (a_bf8) return launch_p4(gemm_fp8_p4_kernel<bf16_t, 2>, kp, grid8, st);
      return launch_p4(gemm_fp8_p4_kernel<bf16_t, 1>, kp, grid8, st);
    }
    if (a_bf8) return launch_p4(gemm_fp8_p4_kernel<float, 2>, kp, grid8, st);
    return launch_p4(gemm_fp8_p4_kernel<float, 1>, kp, grid8, st);
  }

  dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)splits, (unsigned)batch);
  const int layout = (g->a_kmajor ? 2 : 0) | (g->b_kmajor ? 1 : 0);
  const bool glds_ok = g->in_dtype == FAVIT_BF16 && kp.a_vec && kp.b_vec && (g->K % BK16) == 0 && g->K > 0 &&
                       (g->a_kmajor || g->M >= 8) && (g->b_kmajor || g->N >= 8) &&
                       (g->a_kmajor || (g->M % 8) == 0) && (g->b_kmajor || (g->N % 8) == 0);
  const bool force128 = knobs().force128;
  const bool no_p4 = knobs().no_p4;
  // large GEMMs: 256x128 tiles, 2 workgroups per CU, wave-private epilogue (plain or fp32-atomic)
  const long t4 = ((g->M + 255) / 256) * tiles_n;
  // 256x256 tiles: single-pass NT problems with K >= 512 and N a multiple of 256 (the D = 768 forward GEMMs)
  const bool no_p7 = knobs().no_p7;
  if (glds_ok && !force128 && !no_p4 && !no_p7 && splits == 1 && !atomic && batch == 1 && g->a_kmajor && g->b_kmajor &&
      g->M >= 1024 && (g->N % P7_BN) == 0 && (g->K % P4_BK) == 0 && g->K >= 512 &&
      ((g->M + 255) / 256) * (g->N / P7_BN) >= 256) {
    KParams k7 = kp;
    k7.tiles_n = (int)(g->N / P7_BN);
    k7.ntiles = (int)(((g->M + 255) / 256) * k7.tiles_n);
    dim3 grid7((unsigned)k7.ntiles, 1u, 1u);
    if (g->out_dtype == FAVIT_BF16) return launch_p7(gemm_bf16_p7_kernel<bf16_t>, k7, grid7, st);
    return launch_p7(gemm_bf16_p7_kernel<float>, k7, grid7, st);
  }
  // ping-pong kernel: single-pass problems with a k-major A and a VERY long reduction (K >= 4096) that the 256x256
  // kernel cannot take (mn-major B, or N % 256 != 0).  Measured (tools/gemm_bench.py, tools/pp_probe.py): 8192^3
  // 1053-1081 TF against 962 TF for p4 on the same device; but 203 vs 175 us and 153 vs 134 us on the ViT-Base
  // input-gradient GEMMs (K = 3072 / 2304, 870 tiles: one workgroup per CU loses to p4's two), a tie at K = 1536 and
  // a loss at K = 384 -- so every GEMM of the training configurations stays on p4.  FAVIT_GEMM_PP=1 forces it (tests).
  const bool pp_shape = (g->K >= 4096 && t4 >= 256) || getenv("FAVIT_GEMM_PP") != nullptr;   // read per call: tests toggle it
  if (glds_ok && !force128 && !knobs().no_pp && pp_shape && splits == 1 && !atomic && batch == 1 && g->a_kmajor &&
      !g->a_rowsum && g->M >= 256 && (g->K % BK16) == 0 && g->K >= 2 * BK16) {
    dim3 gridp((unsigned)t4, 1u, 1u);
    kp.ntiles = (int)t4;
    if (g->out_dtype == FAVIT_BF16) {
      if (g->b_kmajor) return launch_pp(gemm_bf16_pp_kernel<true, bf16_t>, kp, gridp, st);
      return launch_pp(gemm_bf16_pp_kernel<false, bf16_t>, kp, gridp, st);
    }
    if (g->b_kmajor) return launch_pp(gemm_bf16_pp_kernel<true, float>, kp, gridp, st);
    return launch_pp(gemm_bf16_pp_kernel<false, float>, kp, gridp, st);
  }
  // experimental persistent kernel with the deferred epilogue (FAVIT_GEMM_PD=1; read per call: tests toggle it)
  if (glds_ok && !force128 && splits == 1 && !atomic && batch == 1 && g->a_kmajor && g->b_kmajor && !g->a_rowsum &&
      (g->M % P4_BM) == 0 && (g->N % BN) == 0 && (g->K % P4_BK) == 0 && g->K >= 12 * P4_BK && t4 >= 8 &&
      g->act == FAVIT_ACT_NONE && !g->aux_out && !g->aux_in && kp.drop_thresh == 0 && kp.c_vec && getenv("FAVIT_GEMM_PD") != nullptr) {
    const bool res_f32 = g->out_dtype == FAVIT_F32 && g->residual != nullptr;
    const bool plain_bf16 = g->out_dtype == FAVIT_BF16 && g->residual == nullptr;
    if (res_f32 || plain_bf16) {
      KParams kd = kp;
      kd.ntiles = (int)t4;
      int ncu = 256;
      { const char* e = getenv("FAVIT_GEMM_PD_WGS"); if (e && atoi(e) >= 8 && atoi(e) <= 512) ncu = atoi(e) & ~7; }
      const int nwg = t4 < ncu ? (int)(t4 & ~7L) : ncu;
      if (nwg >= 8) {
        if (res_f32) return launch_pd(gemm_bf16_pd_kernel<1>, kd, nwg, st);
        return launch_pd(gemm_bf16_pd_kernel<0>, kd, nwg, st);
      }
    }
  }
  if (glds_ok && !force128 && !no_p4 && (g->K % P4_BK) == 0 && (kps % P4_BK) == 0 &&
      ((splits == 1 && g->M >= 1024 && t4 * batch >= 256) || (xcd_split && splits > 1))) {
    dim3 grid4((unsigned)t4, (unsigned)splits, (unsigned)batch);
    kp.ntiles = (int)t4;
    if (xcd_split) {
      kp.xcd_split = 1;
      grid4 = dim3((unsigned)(t4 * splits), 1u, 1u);
    }
    // tail round as quarter tiles (p4_quarter_body): single-pass, un-batched, k-major A, a last round that would
    // leave more than 40 % of the 512 workgroup slots empty
    const long tail4 = t4 % 512;
    if (!knobs().no_quarter && g->a_kmajor && splits == 1 && batch == 1 && !atomic && t4 > 512 && tail4 > 0 &&
        tail4 <= knobs().quarter_max && (g->K % BK16) == 0) {
      kp.q_block0 = (int)(t4 - tail4);
      kp.q_tile0 = (int)(t4 - tail4);
      grid4 = dim3((unsigned)(t4 - tail4 + 4 * tail4), 1u, 1u);
    }
    if (g->out_dtype == FAVIT_BF16) {
      switch (layout) {
        case 3: return launch_p4(gemm_bf16_p4_kernel<true, true, bf16_t>, kp, grid4, st);
        case 2: return launch_p4(gemm_bf16_p4_kernel<true, false, bf16_t>, kp, grid4, st);
        case 1: return launch_p4(gemm_bf16_p4_kernel<false, true, bf16_t>, kp, grid4, st);
        default: return launch_p4(gemm_bf16_p4_kernel<false, false, bf16_t>, kp, grid4, st);
      }
    } else {
      switch (layout) {
        case 3: return launch_p4(gemm_bf16_p4_kernel<true, true, float>, kp, grid4, st);
        case 2: return launch_p4(gemm_bf16_p4_kernel<true, false, float>, kp, grid4, st);
        case 1: return launch_p4(gemm_bf16_p4_kernel<false, true, float>, kp, grid4, st);
        default: return launch_p4(gemm_bf16_p4_kernel<false, false, float>, kp, grid4, st);
      }
    }
  }
  // fewer 128x128 tiles than 1.5 per CU: 64-row tiles, three-deep DMA ring (latency-bound launches)
  if (glds_ok && !force128 && !knobs().no_s64 && splits == 1 && !atomic && g->a_kmajor &&
      !g->a_rowsum && tiles < 384 && g->M > 64 && g->K >= 2 * BK16) {
    KParams ks = kp;
    const long tm = (g->M + S64_BM - 1) / S64_BM;
    ks.ntiles = (int)(tm * tiles_n);
    dim3 grids((unsigned)(tm * tiles_n), 1u, (unsigned)batch);
    // no more tiles than CUs and at least four k-steps: the K loop split between two wave groups (s64k2)
    if (!knobs().no_s64k2 && batch == 1 && tm * tiles_n <= 256 && g->K >= 4 * BK16) {
      if (g->out_dtype == FAVIT_BF16) {
        if (g->b_kmajor) return launch_s64k2(gemm_bf16_s64k2_kernel<true, bf16_t>, ks, grids, st);
        return launch_s64k2(gemm_bf16_s64k2_kernel<false, bf16_t>, ks, grids, st);
      }
      if (g->b_kmajor) return launch_s64k2(gemm_bf16_s64k2_kernel<true, float>, ks, grids, st);
      return launch_s64k2(gemm_bf16_s64k2_kernel<false, float>, ks, grids, st);
    }
    if (g->out_dtype == FAVIT_BF16) {
      if (g->b_kmajor) return launch_s64(gemm_bf16_s64_kernel<true, bf16_t>, ks, grids, st);
      return launch_s64(gemm_bf16_s64_kernel<false, bf16_t>, ks, grids, st);
    }
    if (g->b_kmajor) return launch_s64(gemm_bf16_s64_kernel<true, float>, ks, grids, st);
    return launch_s64(gemm_bf16_s64_kernel<false, float>, ks, grids, st);
  }
  if (glds_ok) {
    if (g->out_dtype == FAVIT_BF16) {
      switch (layout) {
        case 3: return launch(gemm_bf16_glds_kernel<true, true, bf16_t>, kp, grid, st);
        case 2: return launch(gemm_bf16_glds_kernel<true, false, bf16_t>, kp, grid, st);
        case 1: return launch(gemm_bf16_glds_kernel<false, true, bf16_t>, kp, grid, st);
        default: return launch(gemm_bf16_glds_kernel<false, false, bf16_t>, kp, grid, st);
      }
    } else {
      switch (layout) {
        case 3: return launch(gemm_bf16_glds_kernel<true, true, float>, kp, grid, st);
        case 2: return launch(gemm_bf16_glds_kernel<true, false, float>, kp, grid, st);
        case 1: return launch(gemm_bf16_glds_kernel<false, true, float>, kp, grid, st);
        default: return launch(gemm_bf16_glds_kernel<false, false, float>, kp, grid, st);
      }
    }
  }
  // exact fp32, large problems: the 256x128 DMA kernel (p4f).  Small ones keep the 128x128 tiles below: with fewer
  // than one 256x128 workgroup per CU the finer tiles balance better and the launch is latency-bound anyway.
  if (f32p4_shape && kp.a_vec && kp.b_vec && (kps % BK32) == 0 && t4f * splits * (f32_tbm / 128) >= 384) {
    KParams kf = kp;
    kf.ntiles = (int)t4f;
    const dim3 gridf((unsigned)t4f, (unsigned)splits, 1u);
    if (f32_tbm == 256) {
      switch (layout) {
        case 3: return launch_p4(gemm_f32_p4_kernel<true, true, 256>, kf, gridf, st, "p4f");
        case 2: return launch_p4(gemm_f32_p4_kernel<true, false, 256>, kf, gridf, st, "p4f");
        case 1: return launch_p4(gemm_f32_p4_kernel<false, true, 256>, kf, gridf, st, "p4f");
        default: return launch_p4(gemm_f32_p4_kernel<false, false, 256>, kf, gridf, st, "p4f");
      }
    }
    switch (layout) {
      case 3: return launch_p4(gemm_f32_p4_kernel<true, true, 128>, kf, gridf, st, "p4f128");
      case 2: return launch_p4(gemm_f32_p4_kernel<true, false, 128>, kf, gridf, st, "p4f128");
      case 1: return launch_p4(gemm_f32_p4_kernel<false, true, 128>, kf, gridf, st, "p4f128");
      default: return launch_p4(gemm_f32_p4_kernel<false, false, 128>, kf, gridf, st, "p4f128");
    }
  }
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

namespace {

// slab layout shared by the size query and the launch: per problem the dense [M, N] partial, then its [M] bias row
long grouped_slab_floats(const favit_gemm_t* gs, int count) {
  long tot = 0;
  for (int i = 0; i < count; ++i) tot += gs[i].M * gs[i].N + ((gs[i].M + 3) / 4) * 4;
  return (tot + 63) / 64 * 64;
}

// FAVIT_GROUPED_S=n (A/B measurements): force n K-splits.  Read once per process.
int grouped_forced_splits() {
  static const int v = [] { const char* e = getenv("FAVIT_GROUPED_S"); return e ? atoi(e) : 0; }();
  return v;
}

// How a grouped launch is laid out: chunks (consecutive problems, <= 64 tiles), the number of K-splits and the
// unit -> XCD queues.  The split count minimises
//     max over XCDs of max(1, queue tiles / 64) x (k x 1.4 us x (1 + k / 4000) + 18 us)   k = 32-token k-steps per split
//   + (nsplit > 1 ? (2 nsplit + 1) x output bytes / 5 TB/s + 5 us : 2 x output bytes / 5 TB/s)   slabs + reduction
// among the split counts that keep a split at or below 6,400 tokens (when any does).  The constants are measured
// (tools/grouped_probe.py, round 4, random data, back-to-back launches): 1.4 us per k-step of a round of 256x128
// tiles with two workgroups per CU and ~18 us of prologue + epilogue per round (228 us for 2 rounds x 68 k-steps at
// cfg3); the k / 4000 term is the DRIFT of long splits -- the 63 tiles of a unit stop sharing operand lines in their
// XCD's L2 as they run apart: four cfg2 blocks in one launch took 1.60 us per k-step with 2 splits of 788 k-steps
// against 1.37 with 8 splits of 197 (1300 vs 1214 us), which is also why the 6,400-token cap exists.  Round 2
// maximised slot utilisation alone and took 16 splits at ViT-Base; round 3 priced the slabs; round 4 adds the
// multi-block launches, where one or two splits fill the chip at short token counts (cfg3: 12 blocks, no split, no
// slabs, no reduction kernel: 19 us per block against 40).
struct GroupPlan {
  int nchunks;
  int chunk_first[GROUP_MAX + 1];
  int chunk_tiles[GROUP_MAX];
  long nsplit, kps;
  int xcd_count[8];
  int xcd_units[8][GROUP_XCD_UNITS];
  long max_queue_tiles;
};

bool grouped_pack(GroupPlan& pl, long nsplit) {
  // longest-first packing of the units into eight queues (units of one chunk are equal: walk chunks by size)
  const int nunits = pl.nchunks * (int)nsplit;
  if (nunits > 255 || nunits > 8 * GROUP_XCD_UNITS) return false;
  int order[GROUP_MAX];
  for (int c = 0; c < pl.nchunks; ++c) order[c] = c;
  for (int a = 1; a < pl.nchunks; ++a)            // insertion sort, descending tiles (stable)
    for (int b = a; b > 0 && pl.chunk_tiles[order[b]] > pl.chunk_tiles[order[b - 1]]; --b) { int t = order[b]; order[b] = order[b - 1]; order[b - 1] = t; }
  long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int x = 0; x < 8; ++x) pl.xcd_count[x] = 0;
  for (int oc = 0; oc < pl.nchunks; ++oc) {
    const int c = order[oc];
    for (long sp = 0; sp < nsplit; ++sp) {
      int best = -1;
      for (int x = 0; x < 8; ++x)
        if (pl.xcd_count[x] < GROUP_XCD_UNITS && (best < 0 || load[x] < load[best])) best = x;
      if (best < 0) return false;
      pl.xcd_units[best][pl.xcd_count[best]++] = (int)(sp * pl.nchunks + c);
      load[best] += pl.chunk_tiles[c];
    }
  }
  pl.max_queue_tiles = 0;
  for (int x = 0; x < 8; ++x) pl.max_queue_tiles = load[x] > pl.max_queue_tiles ? load[x] : pl.max_queue_tiles;
  return true;
}

// Returns false if the problems cannot be grouped (more than GROUP_MAX chunks cannot happen: a chunk holds >= 1 problem).
bool grouped_plan(const favit_gemm_t* gs, int count, GroupPlan& pl) {
  const long K = gs[0].K;
  pl.nchunks = 0;
  long out_floats = 0;
  int cur = 0;
  for (int i = 0; i < count; ++i) {
    const long t = ((gs[i].M + P4_BM - 1) / P4_BM) * ((gs[i].N + BN - 1) / BN);
    if (t > 65535) return false;
    out_floats += gs[i].M * gs[i].N;
    if (pl.nchunks == 0 || cur + t > 64) {
      pl.chunk_first[pl.nchunks] = i;
      pl.chunk_tiles[pl.nchunks] = 0;
      ++pl.nchunks;
      cur = 0;
    }
    cur += (int)t;
    pl.chunk_tiles[pl.nchunks - 1] = cur;
  }
  pl.chunk_first[pl.nchunks] = count;
  static const long cand[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32};
  const int forced = grouped_forced_splits();
  double best = 0.0;
  long best_s = 0, best_kps = 0;
  constexpr long KPS_CAP = 6400;
  bool best_capped = false;
  for (long s0 : cand) {
    const long s = forced > 0 ? (long)forced : s0;
    long kps = (K + s - 1) / s;
    kps = ((kps + P4_BK - 1) / P4_BK) * P4_BK;
    if (s > 1 && (s - 1) * kps >= K) continue;        // the last split would be empty
    GroupPlan trial = pl;
    if (!grouped_pack(trial, s)) continue;
    // (tiles of a queue do not run in strict rounds: 216 tiles on 64 slots take ~3.4 tile times, not 4)
    const double rounds = trial.max_queue_tiles <= 64 ? 1.0 : (double)trial.max_queue_tiles / 64.0;
    const double out_bytes = 4.0 * (double)out_floats;
    const double ksteps = (double)(kps / P4_BK);
    const double cost = rounds * (ksteps * 1.4e-6 * (1.0 + ksteps / 4000.0) + 18.0e-6) +
                        (s > 1 ? (2.0 * (double)s + 1.0) * out_bytes / 5.0e12 + 5.0e-6 : 2.0 * out_bytes / 5.0e12);
    const bool capped = kps <= KPS_CAP;               // a candidate under the cap beats any candidate over it
    if (best_s == 0 || (capped && !best_capped) || (capped == best_capped && cost < best)) {
      best = cost; best_s = s; best_kps = kps; best_capped = capped;
    }
    if (forced > 0) break;
  }
  if (best_s == 0) return false;
  pl.nsplit = best_s;
  pl.kps = best_kps;
  return grouped_pack(pl, best_s);
}

int grouped_tn_impl(const favit_gemm_t* gs, int32_t count, float* ws, int64_t ws_bytes, void* stream) {
  if (!gs || count <= 0 || count > GROUP_MAX) return FAVIT_ERR_INVALID;
  hipStream_t st = as_stream(stream);
  GroupParams gp;
  memset(&gp, 0, sizeof(gp));          // every field defined, whatever is added to the structs later
  gp.count = count;
  const long K = gs[0].K;
  if (K <= 0 || (K % P4_BK) != 0) return FAVIT_ERR_UNSUPPORTED;
  bool direct_ok = true;               // nsplit == 1 needs nothing; 16-byte stores when C allows them
  for (int i = 0; i < count; ++i) {
    const favit_gemm_t* g = gs + i;
    if (!g->A || !g->B || !g->C || g->M <= 0 || g->N <= 0) return FAVIT_ERR_INVALID;
    if (g->K != K || g->in_dtype != FAVIT_BF16 || g->out_dtype != FAVIT_F32 || g->a_kmajor || g->b_kmajor ||
        (g->batch > 1) || g->act != FAVIT_ACT_NONE || g->aux_in || g->aux_out || g->residual || g->bias ||
        g->dropout_p > 0.f || g->alpha != 1.0f)
      return FAVIT_ERR_UNSUPPORTED;
    if (!aligned(g->A, 16) || !aligned(g->B, 16) || (g->lda % 8) || (g->ldb % 8) || (g->M % 8) || (g->N % 8) ||
        g->M < 8 || g->N < 8 || g->lda > INT32_MAX || g->ldb > INT32_MAX || g->ldc > INT32_MAX || g->M > INT32_MAX ||
        g->N > INT32_MAX)
      return FAVIT_ERR_UNSUPPORTED;
    TnProb& q = gp.p[i];
    q.A = g->A; q.B = g->B; q.C = reinterpret_cast<float*>(g->C); q.rowsum = g->a_rowsum;
    q.M = (int)g->M; q.N = (int)g->N; q.lda = (int)g->lda; q.ldb = (int)g->ldb; q.ldc = (int)g->ldc;
    q.tiles_n = (int)((g->N + BN - 1) / BN);
    q.ntiles = (int)(((g->M + P4_BM - 1) / P4_BM) * q.tiles_n);
    q.flags = (g->accumulate ? 1 : 0) | (((g->ldc % 4) == 0 && aligned(g->C, 16)) ? 2 : 0);
  }
  const long stride = grouped_slab_floats(gs, count);
  bool slab_dst_ok = ws != nullptr && aligned(ws, 16);
  for (int i = 0; i < count; ++i)      // the reduction writes 16-byte vectors: unaligned destinations keep the atomic path
    slab_dst_ok = slab_dst_ok && (gs[i].ldc % 4) == 0 && aligned(gs[i].C, 16) && (!gs[i].a_rowsum || aligned(gs[i].a_rowsum, 16));
  GroupPlan pl;
  if (!grouped_plan(gs, count, pl)) return FAVIT_ERR_UNSUPPORTED;
  const long nsplit = pl.nsplit;
  gp.nsplit = (int)nsplit;
  gp.k_per_split = pl.kps;
  gp.K = K;
  gp.nchunks = pl.nchunks;
  for (int c = 0; c < pl.nchunks; ++c) { gp.chunk_tiles[c] = (unsigned short)pl.chunk_tiles[c]; gp.chunk_first[c] = (unsigned char)pl.chunk_first[c]; }
  gp.chunk_first[pl.nchunks] = (unsigned char)count;
  for (int x = 0; x < 8; ++x) {
    gp.xcd_count[x] = (unsigned char)pl.xcd_count[x];
    for (int j = 0; j < pl.xcd_count[x]; ++j) gp.xcd_units[x][j] = (unsigned char)pl.xcd_units[x][j];
  }
  const bool slabs = nsplit > 1 && slab_dst_ok && ws_bytes >= (int64_t)(nsplit * stride * 4);
  gp.mode = nsplit == 1 ? 2 : (slabs ? 1 : 0);
  gp.slab_stride = slabs ? stride : 0;
  ReduceParams rp;
  if (slabs) {
    // Every split writes its partial tiles with plain 16-byte stores into its own slab; a second kernel adds the
    // slabs in split order.  fp32 atomics reach ~1.3 TB/s chip-wide (MI355X_MICROARCH.md) against ~5 TB/s for plain
    // stores + the reduction's reads: measured 318 -> 227 us main loop + epilogue at the cfg2 block shapes and
    // 914 -> 538 us at ViT-Base (tools/grouped_probe.py), and the result no longer depends on arrival order.
    memset(&rp, 0, sizeof(rp));
    rp.ws = ws; rp.slab_stride = stride; rp.nsplit = (int)nsplit; rp.nseg = 0;
    long o = 0;
    int blk = 0;
    auto add_seg = [&](float* dst, long off, long rows, long cols, long ld, int acc) {
      rp.seg[rp.nseg] = ReduceSeg{dst, off, (int)rows, (int)cols, (int)ld, acc};
      rp.blk_off[rp.nseg++] = blk;
      blk += (int)((rows * cols / 4 + 256 * REDUCE_PER_BLOCK - 1) / (256 * REDUCE_PER_BLOCK));
    };
    for (int i = 0; i < count; ++i) {
      TnProb& q = gp.p[i];
      add_seg(reinterpret_cast<float*>(gs[i].C), o, gs[i].M, gs[i].N, gs[i].ldc, gs[i].accumulate ? 1 : 0);
      q.C = ws + o;
      q.ldc = (int)gs[i].N;
      o += gs[i].M * gs[i].N;
      if (gs[i].a_rowsum) {
        // the bias gradient is always ADDED to its destination (the fused column sum's contract); M % 8 == 0
        add_seg(gs[i].a_rowsum, o, 1, gs[i].M, gs[i].M, 1);
        q.rowsum = ws + o;
      }
      o += ((gs[i].M + 3) / 4) * 4;
    }
    rp.blk_off[rp.nseg] = blk;
  } else if (gp.mode == 0) {
    for (int i = 0; i < count; ++i) {
      if (!gs[i].accumulate) {
        const long total = gs[i].M * gs[i].N;
        const int zb = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL(zero_c_kernel, dim3(zb, 1, 1), dim3(256), 0, st, reinterpret_cast<float*>(gs[i].C), gs[i].M,
                           gs[i].N, gs[i].ldc, 0L, 0L, 1);
        FAVIT_CHECK_LAUNCH();
      }
    }
  }
#ifdef FAVIT_PROBE
  if (knobs().dbg == 1) gp.count = -count;            // probe build: FAVIT_GEMM_DBG=1 times the grouped main loop without its epilogue
#endif
  g_last_kernel = "grouped_tn";
  g_last_grouped_splits = (int)nsplit;
  favit_ensure_dyn_lds(reinterpret_cast<const void*>(gemm_bf16_p4_grouped_tn_kernel), P4_LDS);
  hipLaunchKernelGGL(gemm_bf16_p4_grouped_tn_kernel, dim3((unsigned)(8 * pl.max_queue_tiles)), dim3(P4_THREADS), P4_LDS, st, gp);
  FAVIT_CHECK_LAUNCH();
  if (slabs) {
    hipLaunchKernelGGL(grouped_reduce_kernel, dim3((unsigned)rp.blk_off[rp.nseg]), dim3(256), 0, st, rp);
    FAVIT_CHECK_LAUNCH();
  }
  return FAVIT_OK;
}

}  // namespace

extern "C" int favit_gemm_grouped_tn(const favit_gemm_t* gs, int32_t count, void* stream) {
  return grouped_tn_impl(gs, count, nullptr, 0, stream);
}

extern "C" int64_t favit_gemm_grouped_tn_workspace(const favit_gemm_t* gs, int32_t count) {
  if (!gs || count <= 0 || count > GROUP_MAX || gs[0].K <= 0) return 0;
  GroupPlan pl;
  if (!grouped_plan(gs, count, pl)) return 0;
  return pl.nsplit > 1 ? (int64_t)(pl.nsplit * grouped_slab_floats(gs, count) * 4) : 0;
}

extern "C" int favit_gemm_grouped_tn_ws(const favit_gemm_t* gs, int32_t count, void* workspace, int64_t workspace_bytes,
                                        void* stream) {
  return grouped_tn_impl(gs, count, reinterpret_cast<float*>(workspace), workspace_bytes, stream);
}

extern "C" int favit_gemm_grouped_last_splits(void) { return g_last_grouped_splits; }

// LayerNorm + small-M forward GEMM in one launch (gemm_bf16_s64ln_kernel); see include/favit.h.
extern "C" int favit_ln_gemm(const favit_gemm_t* g, const float* x, int64_t ldx, const float* gamma, const float* beta,
                             float eps, void* xn, float* mean, float* rstd, void* stream) {
  if (!g || !g->B || !g->C || !x || !gamma || !beta || !xn || !mean || !rstd) return FAVIT_ERR_INVALID;
  if (g->M <= 0 || g->N <= 0 || g->K <= 0) return FAVIT_ERR_INVALID;
  if (g->in_dtype != FAVIT_BF16 || !g->b_kmajor || g->batch > 1 || g->split_k > 1 || g->accumulate || g->a_rowsum)
    return FAVIT_ERR_UNSUPPORTED;
  if ((g->K % 64) != 0 || g->K > 512 || g->K == 320 || g->K == 448) return FAVIT_ERR_UNSUPPORTED;   // NI in {1,2,3,4,6,8}
  if (g->act < FAVIT_ACT_NONE || g->act > FAVIT_ACT_MULAUX) return FAVIT_ERR_INVALID;
  if ((g->act == FAVIT_ACT_DGELU || g->act == FAVIT_ACT_MULAUX) && !g->aux_in) return FAVIT_ERR_INVALID;
  if (g->act == FAVIT_ACT_GELU_SAVEGRAD && !g->aux_out) return FAVIT_ERR_INVALID;
  if (!aligned(g->B, 16) || (g->ldb % 8) != 0 || (ldx % 4) != 0 || !aligned(x, 16) || !aligned(gamma, 16) ||
      !aligned(beta, 16) || !aligned(xn, 8))
    return FAVIT_ERR_ALIGN;
  if (g->dropout_p < 0.f || g->dropout_p >= 1.f) return FAVIT_ERR_INVALID;
  const long tiles_n = (g->N + BN - 1) / BN, tm = (g->M + S64_BM - 1) / S64_BM;
  // the regime of the 64-row kernel (favit_gemm: fewer than 1.5 128x128 tiles per CU); beyond it LayerNorm is a
  // bandwidth-bound pass of its own and the large-tile kernels take the GEMM
  if (((g->M + BM - 1) / BM) * tiles_n >= 384 || g->M <= 64) return FAVIT_ERR_UNSUPPORTED;
  KParams kp;
  memset(&kp, 0, sizeof(kp));
  kp.B = g->B; kp.C = g->C;
  kp.bias = g->bias; kp.aux_in = g->aux_in; kp.aux_out = g->aux_out; kp.residual = g->residual;
  kp.M = g->M; kp.N = g->N; kp.K = g->K;
  kp.ldb = g->ldb; kp.ldc = g->ldc;
  kp.ld_aux_in = g->ld_aux_in; kp.ld_aux_out = g->ld_aux_out; kp.ld_res = g->ld_res;
  kp.batch_inner = 1;
  kp.act = g->act;
  kp.tiles_n = (int)tiles_n;
  kp.ntiles = (int)(tm * tiles_n);
  kp.alpha = g->alpha;
  kp.store_policy = knobs().store_policy;
  kp.drop_thresh = dropout_threshold(g->dropout_p);
  kp.drop_scale = 1.0f / (1.0f - g->dropout_p);
  kp.drop_seed = g->dropout_seed;
  kp.drop_epoch = favit_dropout_epoch_ptr_();
  kp.a_vec = kp.b_vec = 1;
  const size_t osz = g->out_dtype == FAVIT_BF16 ? 2 : 4;
  bool cv = aligned(g->C, 4 * osz) && (g->ldc % 4) == 0;
  if (g->bias) cv = cv && aligned(g->bias, 16);
  if (g->aux_out) cv = cv && aligned(g->aux_out, 4 * osz) && (g->ld_aux_out % 4) == 0;
  if (g->aux_in) cv = cv && aligned(g->aux_in, 8) && (g->ld_aux_in % 4) == 0;
  if (g->residual) cv = cv && aligned(g->residual, 16) && (g->ld_res % 4) == 0;
  kp.c_vec = cv ? 1 : 0;
  LnFuse q{x, (long)ldx, gamma, beta, reinterpret_cast<bf16_t*>(xn), mean, rstd, eps};
  const int ni = (int)(g->K / 64);
  const int nstb = ni > 4 ? 2 : 3;
  const int lds_bytes = ni * S64_A_BYTES + nstb * OP16_BYTES < 4 * WEPI_BYTES ? 4 * WEPI_BYTES : ni * S64_A_BYTES + nstb * OP16_BYTES;
  const dim3 grid((unsigned)(tm * tiles_n));
  hipStream_t st = as_stream(stream);
  g_last_kernel = "s64ln";
#define FAVIT_LNG(NI_, NSTB_)                                                                                          \
  do {                                                                                                                 \
    if (g->out_dtype == FAVIT_BF16) {                                                                                  \
      favit_ensure_dyn_lds(reinterpret_cast<const void*>(gemm_bf16_s64ln_kernel<bf16_t, NI_, NSTB_>), lds_bytes);      \
      hipLaunchKernelGGL((gemm_bf16_s64ln_kernel<bf16_t, NI_, NSTB_>), grid, dim3(NTHREADS), lds_bytes, st, kp, q);    \
    } else {                                                                                                           \
      favit_ensure_dyn_lds(reinterpret_cast<const void*>(gemm_bf16_s64ln_kernel<float, NI_, NSTB_>), lds_bytes);       \
      hipLaunchKernelGGL((gemm_bf16_s64ln_kernel<float, NI_, NSTB_>), grid, dim3(NTHREADS), lds_bytes, st, kp, q);     \
    }                                                                                                                  \
  } while (0)
  switch (ni) {
    case 1: FAVIT_LNG(1, 3); break;
    case 2: FAVIT_LNG(2, 3); break;
    case 3: FAVIT_LNG(3, 3); break;
    case 4: FAVIT_LNG(4, 3); break;
    case 6: FAVIT_LNG(6, 2); break;
    case 8: FAVIT_LNG(8, 2); break;
    default: return FAVIT_ERR_UNSUPPORTED;
  }
#undef FAVIT_LNG
  (void)nstb;
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}
