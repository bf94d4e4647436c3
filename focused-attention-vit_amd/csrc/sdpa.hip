// Fused scaled-dot-product attention (flash style) for the DENSE attention variants of the path:
// vit.MultiHeadAttention (models/vit.py:95-100), CrossAttention / MultiHeadCrossAttention
// (models/attention.py:63-75, 131-144) and the nn.MultiheadAttention branch of models/vit_mhla.py:57-62.
//
//   S = scale * Q K^T  ->  mask  ->  softmax  ->  dropout  ->  O = P V
//
// No [B*H, Lq, Lk] tensor ever reaches HBM: tiles of the streamed operands sit in LDS, both contractions run on
// MFMA (bf16: v_mfma_f32_16x16x32_bf16; fp32 parity mode: the exact-fp32 v_mfma_f32_16x16x4_f32), the softmax
// runs in registers with wave shuffles (online, one (max, sum) pair per row), and the forward only stores O and
// the row log-sum-exp.  The backward recomputes the probabilities from Q, K and that log-sum-exp (two kernels,
// no atomics, deterministic): dQ per query tile (which also computes delta = rowsum(dO * O) of its rows), then (dK, dV)
// per key tile.
//
// One device function serves the three kernels.  An "owner" tile (64 rows, 16 per wave; the rows whose output is
// accumulated) meets "streamed" rows in blocks of 32 staged in LDS:
//   MODE 0 forward : owner = queries, streamed = K (A) and V (B):   O^T  += V^T . Pd
//   MODE 1 dQ      : owner = queries, streamed = K (A) and V (B):   dQ^T += K^T . dS
//   MODE 2 dK, dV  : owner = keys,    streamed = Q (A) and dO (B):  dK^T += Q^T . dS,  dV^T += dO^T . Pd
// T1 = A_streamed . A_owner^T (scores) and T2 = B_streamed . B_owner^T (dPd) come out of the MFMA with a lane
// holding, for ITS owner row (lane & 15), the streamed rows 4g .. 4g+3 (g = lane >> 4) of both 16-row sub-tiles:
// exactly the k-slice the second MFMA wants as its B operand, so P / dS never leave registers (the transposed A
// operand is read with ds_read_b64_tr_b16 in the same permuted row order).
// Output columns are processed in chunks of 16 * NDT <= 128 (blockIdx.z), so any head dim that is a multiple of
// 16 works (single-head CrossAttention: hd = embed_dim).
#include "common.h"

namespace {

struct View {
  long ld, sb, sh;      // element (b, h, l, d) at base + b*sb + h*sh + l*ld + d
};

struct SdpaArgs {
  const void* q;
  const void* k;
  const void* v;
  const void* dout;
  void* out0;           // MODE 0: o    MODE 1: dq    MODE 2: dk
  void* out1;           //                            MODE 2: dv
  float* lse;           // [B*H, Lq]  (written by MODE 0, read by 1 / 2)
  const float* delta;   // [B*H, Lq]  rowsum(dO * O)
  float* delta_out;     // MODE 1 (round 4): the dQ pass computes delta of its owner rows itself and publishes it here
  const void* o;        //   for the dK / dV pass that follows it on the stream (no sdpa_delta_kernel launch: 6 us x 12
  View vo;              //   per cfg1 step); o = the forward's output
  const uint8_t* mask;
  long m_sb, m_sq;
  View vq, vk, vv, vdo, vo0, vo1;
  int B, H, Lq, Lk, hd, rows_per_block;
  float scale;
  uint32_t thresh;
  float keep_scale;
  uint64_t seed;
  const unsigned long long* epoch;   // favit_set_dropout_epoch word (or null)
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static constexpr int KC = 32;                       // contraction length of one MFMA
  typedef bf16x8 frag;
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  // k-slice g of chunk c of a row (elements 32c + 8g .. +7); elements >= hd read as zero
  static __device__ __forceinline__ frag slice_global(const bf16_t* row, int c, int g, int hd) {
    const int e0 = 32 * c + 8 * g;
    if (e0 + 8 <= hd) return *reinterpret_cast<const frag*>(row + e0);
    frag z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (bf16_t)0.f;
    return z;
  }
  static __device__ __forceinline__ frag slice_lds(const char* row, int c, int g) {
    return *reinterpret_cast<const frag*>(row + (32 * c + 8 * g) * 2);
  }
};
template <> struct Mma<float> {
  static constexpr int KC = 4;
  typedef float frag;
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ frag slice_global(const float* row, int c, int g, int hd) {
    const int e = 4 * c + g;
    return e < hd ? row[e] : 0.f;
  }
  static __device__ __forceinline__ frag slice_lds(const char* row, int c, int g) {
    return *reinterpret_cast<const float*>(row + (4 * c + g) * 4);
  }
};

typedef __attribute__((address_space(3))) s16x4* lds_tr_ptr_t;

// Column (relative to d0) that MFMA row group p4 of output tile dt covers in the bf16 path.  With 4 or 8 tiles the
// tiles are interleaved (tile 4*half + sub holds columns 64 half + 16 (m/4) + 4 sub + m%4) so that lane group g ends up
// with the 16 consecutive columns 64 half + 16 g .. + 15 and stores them as two 16-byte pieces; a
// ds_read_b64_tr_b16 lane supplies its own column address, so the permutation is free.
template <int NDT>
__device__ __forceinline__ int tile_col(int dt, int p4) {
  return NDT % 4 == 0 ? 64 * (dt >> 2) + 16 * p4 + 4 * (dt & 3) : 16 * dt + 4 * p4;
}

// acc[dt] (+)= X^T[d0 + 16 dt ..][rows of the block] . w   with w = the lane's 8 values (sub-tile st = e >> 2,
// row 4g + (e & 3)) of its owner column.  tile = LDS image of the 32 streamed rows (row stride rs bytes).
template <typename T, int NDT>
__device__ __forceinline__ void acc_transposed(f32x4 (&acc)[NDT], const char* tile, int rs, int d0, int lane,
                                               const float (&w)[8], int ne) {
  const int g = lane >> 4, n = lane & 15;
  if constexpr (sizeof(T) == 2) {
    bf16x8 wf;
#pragma unroll
    for (int e = 0; e < 8; ++e) wf[e] = (bf16_t)w[e];
    const int q4 = n >> 2, p4 = n & 3;
    const char* r0 = tile + (4 * g + q4) * rs + d0 * 2;
    const char* r1 = tile + (16 + 4 * g + q4) * rs + d0 * 2;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const int coff = tile_col<NDT>(dt, p4) * 2;
      const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr_t)(r0 + coff));
      const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr_t)(r1 + coff));
      typedef __attribute__((ext_vector_type(8))) short s16x8;
      s16x8 xx;
      xx[0] = lo4[0]; xx[1] = lo4[1]; xx[2] = lo4[2]; xx[3] = lo4[3];
      xx[4] = hi4[0]; xx[5] = hi4[1]; xx[6] = hi4[2]; xx[7] = hi4[3];
      acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xx), wf, acc[dt], 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (e >= ne) break;                              // 16-row blocks (fp32, very wide heads): sub-tile 0 only
      const char* row = tile + (16 * (e >> 2) + 4 * g + (e & 3)) * rs + (d0 + n) * 4;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
        acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(*reinterpret_cast<const float*>(row + 64 * dt), w[e], acc[dt], 0, 0, 0);
    }
  }
}

// rows [s0, s0 + 32) x hd of a streamed matrix -> LDS (zero beyond Ls rows and in the padded columns)
template <typename T>
__device__ __forceinline__ void stage_tile(char* tile, int rs, const T* base, long ld, int s0, int Ls, int hd,
                                           int hd_pad, int tid, int nrows, int nthr) {
  constexpr int EPC = 16 / (int)sizeof(T);            // elements per 16-byte chunk
  const int cpr = hd_pad / EPC;
  for (int c = tid; c < nrows * cpr; c += nthr) {
    const int r = c / cpr, ch = c - r * cpr;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (s0 + r < Ls && (ch + 1) * EPC <= hd) v = *reinterpret_cast<const uint4*>(base + (long)(s0 + r) * ld + ch * EPC);
    *reinterpret_cast<uint4*>(tile + r * rs + ch * 16) = v;
  }
}

// The same staging in two halves, so that the NEXT block's rows are in flight (in registers) while the current block
// is being consumed: up to PF 16-byte chunks per thread and tile.
constexpr int SDPA_PF = 2;
template <typename T>
__device__ __forceinline__ void tile_gload(uint4 (&r)[SDPA_PF], const T* base, long ld, int s0, int Ls, int hd, int cpr,
                                           int nchunks, int tid, int nthr) {
  constexpr int EPC = 16 / (int)sizeof(T);
#pragma unroll
  for (int j = 0; j < SDPA_PF; ++j) {
    const int c = tid + nthr * j;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (c < nchunks) {
      const int row = c / cpr, ch = c - row * cpr;
      if (s0 + row < Ls && (ch + 1) * EPC <= hd) v = *reinterpret_cast<const uint4*>(base + (long)(s0 + row) * ld + ch * EPC);
    }
    r[j] = v;
  }
}
__device__ __forceinline__ void tile_lstore(char* tile, int rs, const uint4 (&r)[SDPA_PF], int cpr, int nchunks, int tid,
                                            int nthr) {
#pragma unroll
  for (int j = 0; j < SDPA_PF; ++j) {
    const int c = tid + nthr * j;
    if (c < nchunks) {
      const int row = c / cpr, ch = c - row * cpr;
      *reinterpret_cast<uint4*>(tile + row * rs + ch * 16) = r[j];
    }
  }
}

template <typename T, int NDT, int MODE>
__global__ __launch_bounds__(320) void sdpa_kernel(SdpaArgs a) {      // 4 or 5 waves: 64 or 80 owner rows per workgroup
  if (a.thresh) a.seed = favit_eff_seed(a.seed, a.epoch);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef Mma<T> M;
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;
  const int z = blockIdx.y, b = z / a.H, h = z % a.H;
  const int d0 = blockIdx.z * (16 * NDT);
  const int Lo = MODE == 2 ? a.Lk : a.Lq;              // owner rows
  const int Ls = MODE == 2 ? a.Lq : a.Lk;              // streamed rows
  const int hd = a.hd;
  const int hd_pad = (hd + M::KC - 1) / M::KC * M::KC;
  const int rs = hd_pad * (int)sizeof(T) + 16;
  const int SB = a.rows_per_block;                     // streamed rows per block: 32, or 16 (fp32, very wide heads)
  const int ne = SB == 32 ? 8 : 4;
  char* tileA = smem;
  char* tileB = smem + SB * rs;

  const T* Q = reinterpret_cast<const T*>(a.q) + b * a.vq.sb + h * a.vq.sh;
  const T* Kp = reinterpret_cast<const T*>(a.k) + b * a.vk.sb + h * a.vk.sh;
  const T* V = reinterpret_cast<const T*>(a.v) + b * a.vv.sb + h * a.vv.sh;
  const T* dO = MODE ? reinterpret_cast<const T*>(a.dout) + b * a.vdo.sb + h * a.vdo.sh : nullptr;
  const T* ownA = MODE == 2 ? Kp : Q;   const long ownA_ld = MODE == 2 ? a.vk.ld : a.vq.ld;
  const T* ownB = MODE == 2 ? V : dO;   const long ownB_ld = MODE == 2 ? a.vv.ld : a.vdo.ld;
  const T* strA = MODE == 2 ? Q : Kp;   const long strA_ld = MODE == 2 ? a.vq.ld : a.vk.ld;
  const T* strB = MODE == 2 ? dO : V;   const long strB_ld = MODE == 2 ? a.vdo.ld : a.vv.ld;

  const int on = blockIdx.x * (nthr >> 2) + wave * 16 + n;      // this lane's owner row (16 rows per wave)
  const int onc = min(on, Lo - 1);
  const T* oa_row = ownA + (long)onc * ownA_ld;
  const T* ob_row = MODE ? ownB + (long)onc * ownB_ld : nullptr;
  const long zq = (long)z * a.Lq;

  f32x4 acc0[NDT], acc1[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) { acc0[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  float m_run = -INFINITY, l_run = 0.f;                // MODE 0: online softmax state of the owner (query) row
  float lse_o = 0.f, delta_o = 0.f;                    // MODE 1: the owner row's statistics
  if (MODE == 1) {
    lse_o = a.lse[zq + onc];
    if (a.delta_out) {
      // delta_i = sum_d dO[i, d] * O[i, d] of the owner row: the four lanes of a row take the k-slices g of every chunk
      const T* o_row = reinterpret_cast<const T*>(a.o) + b * a.vo.sb + h * a.vo.sh + (long)onc * a.vo.ld;
      float dsum = 0.f;
      const int nch = (hd + M::KC - 1) / M::KC;
      for (int c = 0; c < nch; ++c) {
        const typename M::frag fd = M::slice_global(ob_row, c, g, hd), fo = M::slice_global(o_row, c, g, hd);
        if constexpr (sizeof(T) == 2) {
#pragma unroll
          for (int j = 0; j < 8; ++j) dsum = fmaf((float)fd[j], (float)fo[j], dsum);
        } else {
          dsum = fmaf(fd, fo, dsum);
        }
      }
      dsum += __shfl_xor(dsum, 16);
      dsum += __shfl_xor(dsum, 32);
      delta_o = dsum;
      if (g == 0 && on < Lo && blockIdx.z == 0) a.delta_out[zq + on] = dsum;
    } else {
      delta_o = a.delta[zq + onc];
    }
  }
  const uint8_t* mrow_b = a.mask ? a.mask + (long)b * a.m_sb : nullptr;
  const int nchunk = hd_pad / M::KC;

  // The owner rows' fragments do not change over the streamed blocks: with few chunks (bf16, head dim <= 128) they are
  // read once.  (In the loop they were global loads in front of every block's first MFMA, and a wait for them is a
  // wait for every older load as well -- vmcnt is in order -- which would also defeat the prefetch below.)
  constexpr int HOIST = sizeof(T) == 2 ? 4 : 0;
  const bool hoisted = HOIST > 0 && nchunk <= HOIST;
  typename M::frag fo_h[HOIST > 0 ? HOIST : 1], fb_h[HOIST > 0 ? HOIST : 1];
  if (hoisted) {
#pragma unroll
    for (int c = 0; c < HOIST; ++c) {
      if (c < nchunk) {
        fo_h[c] = M::slice_global(oa_row, c, g, hd);
        if (MODE) fb_h[c] = M::slice_global(ob_row, c, g, hd);
      }
    }
  }
  // streamed rows of the NEXT block travel in registers while this block is consumed (small tiles only)
  constexpr int EPC = 16 / (int)sizeof(T);
  const int cpr = hd_pad / EPC, nchunks = SB * cpr;
  const bool pf = hoisted && nchunks <= SDPA_PF * nthr && !a.mask;
  uint4 pra[SDPA_PF], prb[SDPA_PF];
  if (pf) {
    tile_gload<T>(pra, strA, strA_ld, 0, Ls, hd, cpr, nchunks, tid, nthr);
    tile_gload<T>(prb, strB, strB_ld, 0, Ls, hd, cpr, nchunks, tid, nthr);
  }

  for (int s0 = 0; s0 < Ls; s0 += SB) {
    __syncthreads();                                   // previous block's LDS reads are complete
    if (pf) {
      tile_lstore(tileA, rs, pra, cpr, nchunks, tid, nthr);
      tile_lstore(tileB, rs, prb, cpr, nchunks, tid, nthr);
    } else {
      stage_tile<T>(tileA, rs, strA, strA_ld, s0, Ls, hd, hd_pad, tid, SB, nthr);
      stage_tile<T>(tileB, rs, strB, strB_ld, s0, Ls, hd, hd_pad, tid, SB, nthr);
    }
    __syncthreads();
    // MODE 2: the streamed rows' statistics, loaded BEFORE the prefetch is issued (in-order vmcnt)
    float lse_s[8], dl_s[8];
    if (MODE == 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int sr = min(s0 + 16 * (e >> 2) + 4 * g + (e & 3), Ls - 1);
        lse_s[e] = a.lse[zq + sr];
        dl_s[e] = a.delta[zq + sr];
      }
    }
    if (pf && s0 + SB < Ls) {
      tile_gload<T>(pra, strA, strA_ld, s0 + SB, Ls, hd, cpr, nchunks, tid, nthr);
      tile_gload<T>(prb, strB, strB_ld, s0 + SB, Ls, hd, cpr, nchunks, tid, nthr);
    }

    f32x4 t1[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    f32x4 t2[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    const char* ra0 = tileA + n * rs;
    const char* ra1 = tileA + (16 + n) * rs;
    const char* rb0 = tileB + n * rs;
    const char* rb1 = tileB + (16 + n) * rs;
    if (hoisted) {
#pragma unroll
      for (int c = 0; c < HOIST; ++c) {
        if (c < nchunk) {
          t1[0] = M::mma(M::slice_lds(ra0, c, g), fo_h[c], t1[0]);
          if (SB == 32) t1[1] = M::mma(M::slice_lds(ra1, c, g), fo_h[c], t1[1]);
          if (MODE) {
            t2[0] = M::mma(M::slice_lds(rb0, c, g), fb_h[c], t2[0]);
            if (SB == 32) t2[1] = M::mma(M::slice_lds(rb1, c, g), fb_h[c], t2[1]);
          }
        }
      }
    } else {
      for (int c = 0; c < nchunk; ++c) {
        const typename M::frag fo = M::slice_global(oa_row, c, g, hd);
        t1[0] = M::mma(M::slice_lds(ra0, c, g), fo, t1[0]);
        if (SB == 32) t1[1] = M::mma(M::slice_lds(ra1, c, g), fo, t1[1]);
        if (MODE) {
          const typename M::frag fb = M::slice_global(ob_row, c, g, hd);
          t2[0] = M::mma(M::slice_lds(rb0, c, g), fb, t2[0]);
          if (SB == 32) t2[1] = M::mma(M::slice_lds(rb1, c, g), fb, t2[1]);
        }
      }
    }

    // ---- scores of (streamed row s, owner row on); i = query, j = key ----
    float sc[8], keep[8];
    bool ok[8];
    float bm = -INFINITY;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int s = s0 + 16 * (e >> 2) + 4 * g + (e & 3);
      const int i = MODE == 2 ? s : on, j = MODE == 2 ? on : s;
      ok[e] = (s < Ls) && (on < Lo) && (e < ne);
      if (ok[e] && mrow_b && mrow_b[(long)i * a.m_sq + j] == 0) ok[e] = false;      // masked_fill(-inf)
      sc[e] = ok[e] ? t1[e >> 2][e & 3] * a.scale : -INFINITY;
      keep[e] = 1.f;
      if (a.thresh && ok[e]) keep[e] = favit_keep(a.seed, (uint64_t)((zq + i) * (long)a.Lk + j), a.thresh) ? a.keep_scale : 0.f;
      bm = fmaxf(bm, sc[e]);
    }
    float wA[8], wB[8];                                // B operands of the transposed products: dS and Pd
    if (MODE == 0) {
      bm = quad16_max(bm);
      const float m_new = fmaxf(m_run, bm);
      const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;       // a fully masked prefix: no NaN in flight
      const float alpha = __expf(m_run - m_safe);
      float ls = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float p = __expf(sc[e] - m_safe);
        ls += p;
        wB[e] = p * keep[e];
      }
      l_run = l_run * alpha + ls;                      // per-lane partial sum (reduced over g at the end)
      m_run = m_new;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        acc1[dt][0] *= alpha; acc1[dt][1] *= alpha; acc1[dt][2] *= alpha; acc1[dt][3] *= alpha;
      }
      acc_transposed<T, NDT>(acc1, tileB, rs, d0, lane, wB, ne);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float lse_i = lse_o, dl_i = delta_o;
        if (MODE == 2) {
          lse_i = lse_s[e];
          dl_i = dl_s[e];
        }
        const float p = ok[e] ? __expf(sc[e] - lse_i) : 0.f;
        const float dp = t2[e >> 2][e & 3] * keep[e];
        wA[e] = p * (dp - dl_i) * a.scale;             // dS (times the scale of the scores)
        wB[e] = p * keep[e];                           // dropped probabilities
      }
      acc_transposed<T, NDT>(acc0, tileA, rs, d0, lane, wA, ne);
      if (MODE == 2) acc_transposed<T, NDT>(acc1, tileB, rs, d0, lane, wB, ne);
    }
  }

  // ---- store: the lane holds, for owner row `on`, columns d0 + 16 dt + 4g .. + 3 (fp32, or fewer than 4 tiles),
  //      or the 16 consecutive columns d0 + 64 half + 16 g .. of tiles 4 half .. 4 half + 3 (bf16, see tile_col) ----
  auto store = [&](void* base, const View& vw, const f32x4 (&acc)[NDT], float mul) {
    T* orow = reinterpret_cast<T*>(base) + b * vw.sb + h * vw.sh + (long)on * vw.ld;
    if constexpr (sizeof(T) == 2 && NDT % 4 == 0) {
#pragma unroll
      for (int half = 0; half < NDT / 4; ++half) {
        const int d = d0 + 64 * half + 16 * g;
        if (d < hd) {
          bf16x8 lo, hi;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            lo[r] = (bf16_t)(acc[4 * half + 0][r] * mul);
            lo[4 + r] = (bf16_t)(acc[4 * half + 1][r] * mul);
            hi[r] = (bf16_t)(acc[4 * half + 2][r] * mul);
            hi[4 + r] = (bf16_t)(acc[4 * half + 3][r] * mul);
          }
          *reinterpret_cast<bf16x8*>(orow + d) = lo;
          *reinterpret_cast<bf16x8*>(orow + d + 8) = hi;
        }
      }
    } else {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const int d = d0 + 16 * dt + 4 * g;
        if (d < hd) {
          if constexpr (sizeof(T) == 2) {
            bf16x4 ob = {(bf16_t)(acc[dt][0] * mul), (bf16_t)(acc[dt][1] * mul), (bf16_t)(acc[dt][2] * mul), (bf16_t)(acc[dt][3] * mul)};
            *reinterpret_cast<bf16x4*>(orow + d) = ob;
          } else {
            *reinterpret_cast<float4*>(orow + d) = make_float4(acc[dt][0] * mul, acc[dt][1] * mul, acc[dt][2] * mul, acc[dt][3] * mul);
          }
        }
      }
    }
  };
  if (MODE == 0) {
    l_run = quad16_sum(l_run);
    const float inv = 1.0f / l_run;                    // a fully masked row: 0 * inf = NaN, as torch's softmax
    if (on < Lo) {
      store(a.out0, a.vo0, acc1, inv);
      if (g == 0 && blockIdx.z == 0) a.lse[zq + on] = m_run + __logf(l_run);
    }
  } else if (on < Lo) {
    store(a.out0, a.vo0, acc0, 1.0f);
    if (MODE == 2) store(a.out1, a.vo1, acc1, 1.0f);
  }
}

template <typename T, int MODE>
int launch_mode(const SdpaArgs& a_in, hipStream_t st) {
  SdpaArgs a = a_in;
  const int Lo = MODE == 2 ? a.Lk : a.Lq;
  const int kc = Mma<T>::KC;
  const int hd_pad = (a.hd + kc - 1) / kc * kc;
  const int rs = hd_pad * (int)sizeof(T) + 16;
  int lds = 2 * 32 * rs + 1024;            // + slack: unused output columns may read past the last row
  a.rows_per_block = 32;
  if (lds > 160 * 1024 && sizeof(T) == 4) {            // fp32 with a very wide head: 16 streamed rows per block
    a.rows_per_block = 16;
    lds = 2 * 16 * rs + 1024;
  }
  if (lds > 160 * 1024) return FAVIT_ERR_UNSUPPORTED;
  // output columns per workgroup: the largest of 16 / 32 / 64 / 128 that does not exceed the head dim
  const int ndt = a.hd >= 128 ? 8 : a.hd >= 64 ? 4 : a.hd >= 32 ? 2 : 1;
  // 64 owner rows per workgroup (four waves), or 80 (five) where that saves workgroups: 65 tokens (ViT-Tiny on 32x32
  // images) are ONE workgroup of five waves instead of a full one plus one that owns a single row
  int nw = ((Lo + 79) / 80 < (Lo + 63) / 64) ? 5 : 4;
  { const char* e = getenv("FAVIT_SDPA_WAVES"); if (e && (atoi(e) == 4 || atoi(e) == 5)) nw = atoi(e); }
  const dim3 grid((unsigned)((Lo + 16 * nw - 1) / (16 * nw)), (unsigned)(a.B * a.H), (unsigned)((a.hd + 16 * ndt - 1) / (16 * ndt)));
#define FAVIT_SDPA_LAUNCH(NDT)                                                                            \
  do {                                                                                                    \
    if (lds > 65536) favit_ensure_dyn_lds(reinterpret_cast<const void*>(sdpa_kernel<T, NDT, MODE>), lds); \
    hipLaunchKernelGGL((sdpa_kernel<T, NDT, MODE>), grid, dim3(64 * nw), lds, st, a);                     \
  } while (0)
  switch (ndt) {
    case 8: FAVIT_SDPA_LAUNCH(8); break;
    case 4: FAVIT_SDPA_LAUNCH(4); break;
    case 2: FAVIT_SDPA_LAUNCH(2); break;
    default: FAVIT_SDPA_LAUNCH(1); break;
  }
#undef FAVIT_SDPA_LAUNCH
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int check(const favit_sdpa_t* s, bool bwd) {
  if (!s || !s->q || !s->k || !s->v || !s->lse) return FAVIT_ERR_INVALID;
  if (s->B <= 0 || s->H <= 0 || s->Lq <= 0 || s->Lk <= 0 || s->hd <= 0) return FAVIT_ERR_INVALID;
  if (s->dtype != FAVIT_F32 && s->dtype != FAVIT_BF16) return FAVIT_ERR_INVALID;
  if (s->dropout_p < 0.f || s->dropout_p >= 1.f) return FAVIT_ERR_INVALID;
  if (s->hd % 16) return FAVIT_ERR_UNSUPPORTED;
  const int vec = s->dtype == FAVIT_BF16 ? 8 : 4;       // 16-byte row chunks: strides in elements
  const int64_t* str[] = {s->q_str, s->k_str, s->v_str, s->o_str};
  for (const int64_t* p : str)
    if (p[0] % vec || p[1] % vec || p[2] % vec) return FAVIT_ERR_ALIGN;
  if (!al16(s->q) || !al16(s->k) || !al16(s->v)) return FAVIT_ERR_ALIGN;
  if (!bwd) {
    if (!s->o || !al16(s->o)) return FAVIT_ERR_INVALID;
  } else {
    if (!s->o || !s->dout || !s->dq || !s->dk || !s->dv || !s->delta) return FAVIT_ERR_INVALID;
    const int64_t* str2[] = {s->do_str, s->dq_str, s->dk_str, s->dv_str};
    for (const int64_t* p : str2)
      if (p[0] % vec || p[1] % vec || p[2] % vec) return FAVIT_ERR_ALIGN;
    if (!al16(s->dout) || !al16(s->dq) || !al16(s->dk) || !al16(s->dv)) return FAVIT_ERR_ALIGN;
  }
  return FAVIT_OK;
}

inline View view(const int64_t* p) { return View{(long)p[0], (long)p[1], (long)p[2]}; }

SdpaArgs base_args(const favit_sdpa_t* s) {
  SdpaArgs a;
  a.q = s->q; a.k = s->k; a.v = s->v; a.dout = s->dout;
  a.out0 = nullptr; a.out1 = nullptr;
  a.lse = s->lse; a.delta = s->delta;
  a.delta_out = nullptr; a.o = s->o; a.vo = view(s->o_str);
  a.mask = s->mask; a.m_sb = (long)s->m_sb; a.m_sq = (long)s->m_sq;
  a.vq = view(s->q_str); a.vk = view(s->k_str); a.vv = view(s->v_str); a.vdo = view(s->do_str);
  a.vo0 = view(s->o_str); a.vo1 = view(s->o_str);
  a.B = s->B; a.H = s->H; a.Lq = s->Lq; a.Lk = s->Lk; a.hd = s->hd; a.rows_per_block = 64;
  a.scale = s->scale;
  a.thresh = dropout_threshold(s->dropout_p);
  a.keep_scale = 1.0f / (1.0f - s->dropout_p);
  a.seed = s->seed;
  a.epoch = favit_dropout_epoch_ptr_();
  return a;
}

}  // namespace

extern "C" int favit_sdpa_fwd(const favit_sdpa_t* s, void* stream) {
  const int rc = check(s, false);
  if (rc != FAVIT_OK) return rc;
  SdpaArgs a = base_args(s);
  a.out0 = s->o;
  hipStream_t st = as_stream(stream);
  return s->dtype == FAVIT_BF16 ? launch_mode<bf16_t, 0>(a, st) : launch_mode<float, 0>(a, st);
}

extern "C" int favit_sdpa_bwd(const favit_sdpa_t* s, void* stream) {
  int rc = check(s, true);
  if (rc != FAVIT_OK) return rc;
  hipStream_t st = as_stream(stream);
  // (delta = rowsum(dO * O) is computed by the dQ pass for its owner rows and read by the dK / dV pass behind it)
  SdpaArgs a = base_args(s);
  a.out0 = s->dq; a.vo0 = view(s->dq_str);
  a.delta_out = s->delta;
  rc = s->dtype == FAVIT_BF16 ? launch_mode<bf16_t, 1>(a, st) : launch_mode<float, 1>(a, st);
  if (rc != FAVIT_OK) return rc;
  a.delta_out = nullptr;
  a.out0 = s->dk; a.vo0 = view(s->dk_str);
  a.out1 = s->dv; a.vo1 = view(s->dv_str);
  return s->dtype == FAVIT_BF16 ? launch_mode<bf16_t, 2>(a, st) : launch_mode<float, 2>(a, st);
}
