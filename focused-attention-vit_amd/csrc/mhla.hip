// MHLA: windowed "latent" attention of models/mhla.py for gfx950.
//
//  * favit_mhla_fold_{fwd,bwd}: latent_proj (one Linear(hd,hd) shared by K, V and all heads,
//    mhla.py:41,105-106) folded into the qkv projection weights, and the map of the folded
//    gradients back onto qkv.{weight,bias} / latent_proj.{weight,bias}.
//  * favit_mhla_attn_{fwd,bwd}: the banded attention core.  The reference materialises
//    2 x [B,H,L,W,hd] gathered windows (mhla.py:117-126); here a workgroup stages the K~/V~
//    rows of its row block (+ halo, + the two "wrap" edges that the pad rule of
//    mhla.py:72-79 makes reachable) in LDS once, 8 lanes share one query row (hd/8 dims
//    each), the W scores are reduced with wave shuffles and the softmax runs in registers.
//    HBM traffic is the algorithmic 4*B*L*D*e bytes (read q,k~,v~, write o).
//    Backward recomputes the probabilities; dK~/dV~ are gathered per key row from LDS
//    tables of dS / P (deterministic, no atomics).
#include "common.h"
#include <stdlib.h>

namespace {

// closed form of MultiHeadLatentAttention._get_window_indices (mhla.py:46-83)
__device__ __forceinline__ int win_idx(int i, int w, int L, int W, int h) {
  const int lo = max(0, i - h), hi = min(L, i + h + 1), n = hi - lo;
  if (n == W) return lo + w;
  if (lo == 0) return w < n ? w : L - 1;       // short window starting at 0: pad END with L-1
  const int pad = W - n;
  return w < pad ? 0 : lo + (w - pad);         // otherwise: pad FRONT with 0
}

template <typename T, int N>
__device__ __forceinline__ void vload(const T* p, float (&o)[N]) {
  constexpr int BYTES = N * (int)sizeof(T);
  constexpr int NW = BYTES / 4;
  uint32_t w[NW];
  if constexpr (BYTES % 16 == 0) {
#pragma unroll
    for (int c = 0; c < BYTES / 16; ++c) {
      const uint4 u = reinterpret_cast<const uint4*>(p)[c];
      w[4 * c] = u.x; w[4 * c + 1] = u.y; w[4 * c + 2] = u.z; w[4 * c + 3] = u.w;
    }
  } else if constexpr (BYTES == 8) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    w[0] = u.x; w[1] = u.y;
  } else {
    static_assert(BYTES == 4, "unsupported vector width");
    w[0] = *reinterpret_cast<const uint32_t*>(p);
  }
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int j = 0; j < N; ++j) o[j] = __uint_as_float(w[j]);
  } else {
#pragma unroll
    for (int j = 0; j < N / 2; ++j) {
      o[2 * j] = __uint_as_float(w[j] << 16);
      o[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
    }
  }
}

template <typename T, int N>
__device__ __forceinline__ void vstore(T* p, const float (&v)[N]) {
  constexpr int BYTES = N * (int)sizeof(T);
  constexpr int NW = BYTES / 4;
  uint32_t w[NW];
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int j = 0; j < N; ++j) w[j] = __float_as_uint(v[j]);
  } else {
#pragma unroll
    for (int j = 0; j < N / 2; ++j) {
      const bf16_t a = (bf16_t)v[2 * j], b = (bf16_t)v[2 * j + 1];
      w[j] = (uint32_t)__builtin_bit_cast(unsigned short, a) | ((uint32_t)__builtin_bit_cast(unsigned short, b) << 16);
    }
  }
  if constexpr (BYTES % 16 == 0) {
#pragma unroll
    for (int c = 0; c < BYTES / 16; ++c)
      reinterpret_cast<uint4*>(p)[c] = make_uint4(w[4 * c], w[4 * c + 1], w[4 * c + 2], w[4 * c + 3]);
  } else if constexpr (BYTES == 8) {
    *reinterpret_cast<uint2*>(p) = make_uint2(w[0], w[1]);
  } else {
    *reinterpret_cast<uint32_t*>(p) = w[0];
  }
}

__device__ __forceinline__ float sum8(float v) { return dpp_sum8(v); }

// A lane's DPL-element slice kept in its storage form for dot products: bf16 pairs stay packed and
// are consumed by v_dot2c_f32_bf16 (2 MACs per instruction, no unpack); fp32 stays fp32.
template <typename T, int N> struct Slice;
template <int N> struct Slice<float, N> {
  float v[N];
  __device__ __forceinline__ void load(const float* p) { vload<float, N>(p, v); }
  __device__ __forceinline__ float dot(const Slice& o) const {
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < N; ++d) acc = fmaf(v[d], o.v[d], acc);
    return acc;
  }
  __device__ __forceinline__ void unpack(float (&o)[N]) const {
#pragma unroll
    for (int d = 0; d < N; ++d) o[d] = v[d];
  }
};
template <int N> struct Slice<bf16_t, N> {
  uint32_t w[N / 2];
  __device__ __forceinline__ void load(const bf16_t* p) {
    constexpr int BYTES = N * 2;
    if constexpr (BYTES % 16 == 0) {
#pragma unroll
      for (int c = 0; c < BYTES / 16; ++c) {
        const uint4 u = reinterpret_cast<const uint4*>(p)[c];
        w[4 * c] = u.x; w[4 * c + 1] = u.y; w[4 * c + 2] = u.z; w[4 * c + 3] = u.w;
      }
    } else if constexpr (BYTES == 8) {
      const uint2 u = *reinterpret_cast<const uint2*>(p);
      w[0] = u.x; w[1] = u.y;
    } else {
      w[0] = *reinterpret_cast<const uint32_t*>(p);
    }
  }
  __device__ __forceinline__ float dot(const Slice& o) const {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < N / 2; ++j)
      acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, w[j]), __builtin_bit_cast(bf2, o.w[j]), acc, false);
    return acc;
  }
  __device__ __forceinline__ void unpack(float (&o)[N]) const {
#pragma unroll
    for (int j = 0; j < N / 2; ++j) {
      o[2 * j] = __uint_as_float(w[j] << 16);
      o[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
    }
  }
};

// LDS image of rows of one [*, hd] tensor: a contiguous "main" run of rows plus the head
// rows [0, head_n) and the tail rows [tail_lo, L) that the wrap rule can reach.
struct RowImage {
  int main_lo, main_hi, n_main, head_n, tail_lo, n_rows;
  __device__ __forceinline__ void init(int lo, int hi, int edge, int L) {
    main_lo = max(0, lo);
    main_hi = min(L, hi);
    n_main = max(0, main_hi - main_lo);
    head_n = min(edge, L);
    tail_lo = max(0, L - edge);
    n_rows = n_main + head_n + (L - tail_lo);
  }
  __device__ __forceinline__ int slot(int r) const {
    if (r >= main_lo && r < main_hi) return r - main_lo;
    if (r < head_n) return n_main + r;
    return n_main + head_n + (r - tail_lo);
  }
  __device__ __forceinline__ int row_of_slot(int s) const {
    if (s < n_main) return main_lo + s;
    if (s < n_main + head_n) return s - n_main;
    return tail_lo + (s - n_main - head_n);
  }
};

__host__ __device__ inline int row_stride_bytes(int hd, int esz) {
  const int rb = hd * esz;
  return rb + ((rb % 256 == 0) ? 16 : 0);
}

// copy rows of (column block `col0`, hd wide) of src[B*L, ld] into the LDS image
template <typename T>
__device__ __forceinline__ void stage_rows(char* lds, const RowImage& im, const T* src, long ld, long tok0, int col0,
                                           int hd, int rs, int tid, int nthreads) {
  const int cpr = hd * (int)sizeof(T) / 16;       // 16-B chunks per row
  const int total = im.n_rows * cpr;
  for (int c = tid; c < total; c += nthreads) {
    const int s = c / cpr, ch = c - s * cpr;
    const int r = im.row_of_slot(s);
    const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(src + (tok0 + r) * ld + col0) + ch * 16);
    *reinterpret_cast<uint4*>(lds + s * rs + ch * 16) = v;
  }
}

struct AttnArgs {
  const void* qkv;
  const void* dout;
  void* out;      // fwd: o ; bwd: dqkv
  const uint8_t* mask;
  int B, L, H, hd, W;
  int rb;              // backward: key/query rows per workgroup (32 - 2h keeps phase 1 to one pass)
  float inv_sqrt_hd;   // unused (true division is applied), kept for clarity
  uint32_t thresh;
  float keep_scale;
  uint64_t seed;
};

// scores + softmax of one query row (8 lanes own the row; every lane ends with all p[w])
template <typename T, int DPL, int WMAX>
__device__ __forceinline__ void row_softmax(const Slice<T, DPL>& q, const char* ldsK, const RowImage& imK, int rs,
                                            int lane8, int i, int b, int head, const AttnArgs& a, int (&idx)[WMAX],
                                            float (&p)[WMAX]) {
  const int h = a.W >> 1;
  const float sq = sqrtf((float)a.hd);
  float m = -INFINITY;
#pragma unroll
  for (int w = 0; w < WMAX; ++w) {
    float s = -INFINITY;
    idx[w] = 0;
    if (w < a.W) {
      idx[w] = win_idx(i, w, a.L, a.W, h);
      Slice<T, DPL> kf;
      kf.load(reinterpret_cast<const T*>(ldsK + imK.slot(idx[w]) * rs) + lane8 * DPL);
      s = sum8(q.dot(kf)) / sq;                                   // mhla.py:133 (true division)
      if (a.mask && a.mask[((long)b * a.L + i) * a.L + idx[w]] == 0) s = -INFINITY;   // mhla.py:143
    }
    p[w] = s;
    m = fmaxf(m, s);
  }
  float l = 0.f;
#pragma unroll
  for (int w = 0; w < WMAX; ++w) {
    p[w] = (w < a.W) ? __expf(p[w] - m) : 0.f;
    l += p[w];
  }
  const float inv = 1.0f / l;
#pragma unroll
  for (int w = 0; w < WMAX; ++w) p[w] *= inv;
}

constexpr int FWD_QPB = 64;   // query rows per workgroup (forward)
constexpr int BWD_RB = 32;    // key/query rows per workgroup (backward)

template <typename T, int DPL, int WMAX>
__global__ __launch_bounds__(256) void mhla_fwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HD = 8 * DPL;
  const int tid = threadIdx.x, lane8 = tid & 7, qs = tid >> 3;
  const int head = blockIdx.y, b = blockIdx.z;
  const int r0 = blockIdx.x * FWD_QPB, r1 = min(a.L, r0 + FWD_QPB);
  const int h = a.W >> 1, D = a.H * HD;
  const long ld = 3L * D, tok0 = (long)b * a.L;
  const T* qkv = reinterpret_cast<const T*>(a.qkv);
  const int rs = row_stride_bytes(HD, sizeof(T));

  RowImage im;
  im.init(r0 - h, r1 + h, 1, a.L);     // forward only needs keys 0 and L-1 outside the band
  char* ldsK = smem;
  char* ldsV = smem + im.n_rows * rs;
  stage_rows<T>(ldsK, im, qkv, ld, tok0, D + head * HD, HD, rs, tid, 256);
  stage_rows<T>(ldsV, im, qkv, ld, tok0, 2 * D + head * HD, HD, rs, tid, 256);
  __syncthreads();

  for (int base = r0; base < r1; base += 32) {
    const int iq = base + qs;
    const bool valid = iq < r1;
    const int i = valid ? iq : r1 - 1;
    Slice<T, DPL> q;
    q.load(qkv + (tok0 + i) * ld + head * HD + lane8 * DPL);
    int idx[WMAX];
    float p[WMAX];
    row_softmax<T, DPL, WMAX>(q, ldsK, im, rs, lane8, i, b, head, a, idx, p);
    float o[DPL];
#pragma unroll
    for (int d = 0; d < DPL; ++d) o[d] = 0.f;
#pragma unroll
    for (int w = 0; w < WMAX; ++w) {
      if (w < a.W) {
        float pw = p[w];
        if (a.thresh) {   // attention dropout (mhla.py:147), per window slot
          const uint64_t e = (((uint64_t)b * a.H + head) * a.L + i) * a.W + w;
          pw = favit_keep(a.seed, e, a.thresh) ? pw * a.keep_scale : 0.f;
        }
        float vf[DPL];
        vload<T, DPL>(reinterpret_cast<const T*>(ldsV + im.slot(idx[w]) * rs) + lane8 * DPL, vf);
#pragma unroll
        for (int d = 0; d < DPL; ++d) o[d] = fmaf(pw, vf[d], o[d]);
      }
    }
    if (valid) vstore<T, DPL>(reinterpret_cast<T*>(a.out) + (tok0 + i) * (long)D + head * HD + lane8 * DPL, o);
  }
}

template <typename T, int DPL, int WMAX>
__global__ __launch_bounds__(256) void mhla_bwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HD = 8 * DPL;
  const int tid = threadIdx.x, lane8 = tid & 7, qs = tid >> 3;
  const int head = blockIdx.y, b = blockIdx.z;
  const int r0 = blockIdx.x * a.rb, r1 = min(a.L, r0 + a.rb);
  const int L = a.L, W = a.W, h = W >> 1, D = a.H * HD;
  const long ld = 3L * D, tok0 = (long)b * L;
  const T* qkv = reinterpret_cast<const T*>(a.qkv);
  const T* dout = reinterpret_cast<const T*>(a.dout);
  T* dqkv = reinterpret_cast<T*>(a.out);
  const int rs = row_stride_bytes(HD, sizeof(T));

  // rows whose probabilities this block needs: band rows of its keys, plus the rows that
  // wrap onto key L-1 (rows 0..h) / key 0 (rows >= max(h+1, L-h)) if the block owns it.
  const int qm_lo = max(0, r0 - h), qm_hi = min(L, r1 + h), n_qm = qm_hi - qm_lo;
  const int hx_n = (r1 >= L) ? min(h + 1, L) : 0;
  const int tx_lo = max(h + 1, L - h);
  const int tx_n = (r0 == 0 && tx_lo < L) ? (L - tx_lo) : 0;
  const int n_q = n_qm + hx_n + tx_n;                       // <= 32 + 2h + 2(h+1) <= 64
  auto qrow_of = [&](int s) { return s < n_qm ? qm_lo + s : (s < n_qm + hx_n ? s - n_qm : tx_lo + (s - n_qm - hx_n)); };
  auto qslot_of = [&](int r) { return (r >= qm_lo && r < qm_hi) ? r - qm_lo : ((hx_n > 0 && r < hx_n) ? n_qm + r : n_qm + hx_n + (r - tx_lo)); };

  RowImage imK;
  imK.init(r0 - 2 * h, r1 + 2 * h, 2 * h + 1, L);
  char* ldsK = smem;
  char* ldsV = ldsK + imK.n_rows * rs;
  char* ldsQ = ldsV + imK.n_rows * rs;
  char* ldsG = ldsQ + 64 * rs;                               // dO rows
  float* tds = reinterpret_cast<float*>(ldsG + 64 * rs);     // [64][WMAX] dS / sqrt(hd)
  float* tp = tds + 64 * WMAX;                               // [64][WMAX] P after dropout

  stage_rows<T>(ldsK, imK, qkv, ld, tok0, D + head * HD, HD, rs, tid, 256);
  stage_rows<T>(ldsV, imK, qkv, ld, tok0, 2 * D + head * HD, HD, rs, tid, 256);
  {
    const int cpr = HD * (int)sizeof(T) / 16;
    for (int c = tid; c < n_q * cpr; c += 256) {
      const int s = c / cpr, ch = c - s * cpr;
      const int r = qrow_of(s);
      *reinterpret_cast<uint4*>(ldsQ + s * rs + ch * 16) =
          *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(qkv + (tok0 + r) * ld + head * HD) + ch * 16);
      *reinterpret_cast<uint4*>(ldsG + s * rs + ch * 16) =
          *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(dout + (tok0 + r) * (long)D + head * HD) + ch * 16);
    }
  }
  __syncthreads();

  const float sq = sqrtf((float)a.hd);
  // ---- phase 1: per query row: P, dS; dQ for the rows this block owns ----
  for (int pass = 0; pass < 2; ++pass) {
    const int s = pass * 32 + qs;
    if (pass * 32 >= n_q) break;
    const bool valid = s < n_q;
    const int sc = valid ? s : n_q - 1;
    const int i = qrow_of(sc);
    Slice<T, DPL> q, g;
    q.load(reinterpret_cast<const T*>(ldsQ + sc * rs) + lane8 * DPL);
    g.load(reinterpret_cast<const T*>(ldsG + sc * rs) + lane8 * DPL);
    int idx[WMAX];
    float p[WMAX];
    row_softmax<T, DPL, WMAX>(q, ldsK, imK, rs, lane8, i, b, head, a, idx, p);
    float dp[WMAX], pd[WMAX];
    float dot = 0.f;
#pragma unroll
    for (int w = 0; w < WMAX; ++w) {
      dp[w] = 0.f;
      pd[w] = 0.f;
      if (w < W) {
        Slice<T, DPL> vf;
        vf.load(reinterpret_cast<const T*>(ldsV + imK.slot(idx[w]) * rs) + lane8 * DPL);
        float t = sum8(g.dot(vf));                 // d(P_dropped)[w]
        float keep = 1.f;
        if (a.thresh) {
          const uint64_t e = (((uint64_t)b * a.H + head) * L + i) * W + w;
          keep = favit_keep(a.seed, e, a.thresh) ? a.keep_scale : 0.f;
        }
        pd[w] = p[w] * keep;
        dp[w] = t * keep;
        dot = fmaf(p[w], dp[w], dot);
      }
    }
    float dq[DPL];
#pragma unroll
    for (int d = 0; d < DPL; ++d) dq[d] = 0.f;
    const bool own = valid && i >= r0 && i < r1 && sc < n_qm;
#pragma unroll
    for (int w = 0; w < WMAX; ++w) {
      if (w < W) {
        const float ds = p[w] * (dp[w] - dot) / sq;
        if (valid && lane8 == (w & 7)) {
          tds[sc * WMAX + w] = ds;
          tp[sc * WMAX + w] = pd[w];
        }
        float kf[DPL];
        vload<T, DPL>(reinterpret_cast<const T*>(ldsK + imK.slot(idx[w]) * rs) + lane8 * DPL, kf);
#pragma unroll
        for (int d = 0; d < DPL; ++d) dq[d] = fmaf(ds, kf[d], dq[d]);
      }
    }
    if (own) vstore<T, DPL>(dqkv + (tok0 + i) * ld + head * HD + lane8 * DPL, dq);
  }
  __syncthreads();

  // ---- phase 2: per key row j: gather dK~, dV~ from the rows that reference it ----
  {
    const int j = r0 + qs;
    const bool valid = j < r1;
    float dk[DPL], dv[DPL];
#pragma unroll
    for (int d = 0; d < DPL; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
    auto add = [&](int i, int w) {
      const int s = qslot_of(i);
      const float ds = tds[s * WMAX + w], pw = tp[s * WMAX + w];
      float qf[DPL], gf[DPL];
      vload<T, DPL>(reinterpret_cast<const T*>(ldsQ + s * rs) + lane8 * DPL, qf);
      vload<T, DPL>(reinterpret_cast<const T*>(ldsG + s * rs) + lane8 * DPL, gf);
#pragma unroll
      for (int d = 0; d < DPL; ++d) {
        dk[d] = fmaf(ds, qf[d], dk[d]);
        dv[d] = fmaf(pw, gf[d], dv[d]);
      }
    };
    if (valid) {
      for (int i = max(0, j - h); i <= min(L - 1, j + h); ++i) {       // band references
        const int lo = max(0, i - h), n = min(L, i + h + 1) - lo, pad = W - n;
        const int w = (lo == 0 || pad == 0) ? (j - lo) : pad + (j - lo);
        add(i, w);
      }
      if (j == L - 1) {                                                 // END padding of rows with lo == 0
        for (int i = 0; i <= min(h, L - 1); ++i) {
          const int n = min(L, i + h + 1);
          for (int w = n; w < W; ++w) add(i, w);
        }
      }
      if (j == 0) {                                                     // FRONT padding of rows with lo > 0
        for (int i = max(h + 1, L - h); i < L; ++i) {
          const int lo = i - h, n = L - lo, pad = W - n;
          for (int w = 0; w < pad; ++w) add(i, w);
        }
      }
      vstore<T, DPL>(dqkv + (tok0 + j) * ld + D + head * HD + lane8 * DPL, dk);
      vstore<T, DPL>(dqkv + (tok0 + j) * ld + 2 * D + head * HD + lane8 * DPL, dv);
    }
  }
}

// ---------------------------------------------------------------------------------
// MFMA formulation of the banded attention core (bf16, hd in {32, 64, 128}, W <= 15).
// A wave owns 16 query rows.  Their extended key set -- the 16+2h band keys plus the two wrap keys
// 0 and L-1 that the pad rule can reference -- fits 32 "slots", so
//   S[slot][q]  = K_slots . Q^T      (v_mfma_f32_16x16x32_bf16, 2 key tiles x hd/32 k-steps)
//   O^T[d][q]   = V_slots^T . P      (hd/16 MFMAs, K = 32 slots, V^T fragments by ds_read_b64_tr_b16)
// replace ~2*W*hd scalar FMAs per query.  The window rule (duplicates included) becomes a per-slot
// multiplicity: band slots count 1, the wrap slots count the number of pad copies of that row;
// softmax weights are mult*exp(s - max).  The lane that holds S[slot 4g+r][query] after the first
// MFMA also holds exactly the P element the second MFMA wants as its B operand (same permuted slot
// order on both operands), so P never leaves registers.
// ---------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr_t;

struct SlotInfo {          // per lane: the slot geometry of one query row
  int i, lo, hi, n, pad, end_pad, front_pad;
  __device__ __forceinline__ void init(int row, int L, int W, int h) {
    i = row;
    lo = max(0, i - h);
    hi = min(L, i + h + 1);
    n = hi - lo;
    pad = W - n;
    end_pad = (lo == 0) ? pad : 0;
    front_pad = (lo > 0) ? pad : 0;
  }
};

// key index of a slot for the tile that starts at query row t0 (clamped into [0, L-1])
__device__ __forceinline__ int slot_key(int slot, int t0, int h, int L) {
  int j;
  if (slot == 31) j = L - 1;
  else if (slot >= 16 + 2 * h) j = 0;                  // wrap slot 30 and the unused slots: any staged, finite row
  else j = t0 - h + slot;
  return min(max(j, 0), L - 1);
}

template <int HD>
__global__ __launch_bounds__(256) void mhla_fwd_mfma_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RS = HD * 2 + 16;                      // padded row: conflict-free fragment reads
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, qi = lane & 15;
  const int head = blockIdx.y, b = blockIdx.z;
  const int r0 = blockIdx.x * 64, r1 = min(a.L, r0 + 64);
  const int L = a.L, W = a.W, h = W >> 1, D = a.H * HD;
  const long ld = 3L * D, tok0 = (long)b * L;
  const bf16_t* qkv = reinterpret_cast<const bf16_t*>(a.qkv);

  RowImage im;
  im.init(r0 - h, r1 + h, 1, L);
  char* ldsK = smem;
  char* ldsV = smem + im.n_rows * RS;
  stage_rows<bf16_t>(ldsK, im, qkv, ld, tok0, D + head * HD, HD, RS, tid, 256);
  stage_rows<bf16_t>(ldsV, im, qkv, ld, tok0, 2 * D + head * HD, HD, RS, tid, 256);
  __syncthreads();

  const int t0 = r0 + 16 * wave;
  if (t0 >= r1) return;                                // whole wave idle (no barrier after this point)
  const int i = t0 + qi;
  const bool qvalid = i < L;
  SlotInfo si;
  si.init(min(i, L - 1), L, W, h);

  // ---- S = K_slots . Q^T ----
  const bf16_t* qrow = qkv + (tok0 + si.i) * ld + head * HD;
  f32x4 S[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  const int krow0 = im.slot(slot_key(qi, t0, h, L)), krow1 = im.slot(slot_key(16 + qi, t0, h, L));
#pragma unroll
  for (int ks = 0; ks < HD / 32; ++ks) {
    const bf16x8 qf = *reinterpret_cast<const bf16x8*>(qrow + 32 * ks + 8 * g);
    const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(ldsK + krow0 * RS + (32 * ks + 8 * g) * 2);
    const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(ldsK + krow1 * RS + (32 * ks + 8 * g) * 2);
    S[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf, S[0], 0, 0, 0);
    S[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf, S[1], 0, 0, 0);
  }

  // ---- per-slot multiplicity, mask, softmax (lane holds slots 16kt + 4g + r of query qi) ----
  const float inv_sq = 1.0f / sqrtf((float)HD);
  float sc[8], mult[8], kw[8];
  float mx = -INFINITY;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int kt = e >> 2, r = e & 3;
    const int slot = 16 * kt + 4 * g + r;
    int j, mu, w0;                                      // key, multiplicity, first window index
    if (slot < 16 + 2 * h && slot < 30) {
      j = t0 - h + slot;
      mu = (j >= si.lo && j < si.hi) ? 1 : 0;
      w0 = (si.lo == 0 || si.pad == 0) ? (j - si.lo) : si.pad + (j - si.lo);
    } else if (slot == 30) {
      j = 0; mu = si.front_pad; w0 = 0;
    } else if (slot == 31) {
      j = L - 1; mu = si.end_pad; w0 = si.n;
    } else {
      j = 0; mu = 0; w0 = 0;
    }
    if (mu > 0 && a.mask && a.mask[((long)b * L + si.i) * L + j] == 0) mu = 0;     // mhla.py:143
    float kwe = (float)mu;
    if (a.thresh && mu > 0) {                            // dropout acts on every window copy separately
      kwe = 0.f;
      for (int c = 0; c < mu; ++c) {
        const uint64_t idx = (((uint64_t)b * a.H + head) * L + si.i) * W + (w0 + c);
        kwe += favit_keep(a.seed, idx, a.thresh) ? a.keep_scale : 0.f;
      }
    }
    sc[e] = S[kt][r] * inv_sq;
    mult[e] = (float)mu;
    kw[e] = kwe;
    if (mu > 0) mx = fmaxf(mx, sc[e]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float lsum = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = (mult[e] > 0.f) ? __expf(sc[e] - mx) : 0.f;
    lsum = fmaf(mult[e], sc[e], lsum);
  }
  lsum += __shfl_xor(lsum, 16, 64);
  lsum += __shfl_xor(lsum, 32, 64);
  const float inv_l = 1.0f / lsum;
  bf16x8 pf;
#pragma unroll
  for (int e = 0; e < 8; ++e) pf[e] = (bf16_t)(sc[e] * inv_l * kw[e]);

  // ---- O^T = V_slots^T . P ----
  const int q4 = qi >> 2, p4 = qi & 3;
  const int vrow0 = im.slot(slot_key(4 * g + q4, t0, h, L)), vrow1 = im.slot(slot_key(16 + 4 * g + q4, t0, h, L));
  bf16_t* orow = reinterpret_cast<bf16_t*>(a.out) + (tok0 + si.i) * (long)D + head * HD;
#pragma unroll
  for (int dt = 0; dt < HD / 16; ++dt) {
    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(ldsV + vrow0 * RS + (16 * dt + 4 * p4) * 2));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(ldsV + vrow1 * RS + (16 * dt + 4 * p4) * 2));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 vv;
    vv[0] = lo4[0]; vv[1] = lo4[1]; vv[2] = lo4[2]; vv[3] = lo4[3];
    vv[4] = hi4[0]; vv[5] = hi4[1]; vv[6] = hi4[2]; vv[7] = hi4[3];
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vv), pf, o, 0, 0, 0);
    if (qvalid) {
      bf16x4 ob = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
      *reinterpret_cast<bf16x4*>(orow + 16 * dt + 4 * g) = ob;
    }
  }
}

// ---------------------------------------------------------------------------------
// latent_proj fold (weight space, tiny): LDS-tiled, register-blocked small GEMMs so each launch is
// a few microseconds.
//   Weff[s,h] = Wl . Wqkv[s,h],  beff[s,h] = Wl . bqkv[s,h] + bl      (s in {k, v})
// Column D of the [3D, D+1] augmented matrices is the bias.
// ---------------------------------------------------------------------------------
constexpr int FOLD_TC = 64;    // columns per workgroup

// C[r][c] = sum_k At[k][r] * Bt[k][c] for an [HD x 64] tile; At = [HD k][HD r], Bt = [HD k][64] in LDS.
// Thread t owns rows r0 = TR*(t/16).. and columns c0 = 4*(t%16)..: TR x 4 register block, two vector
// LDS reads per k for 4*TR FMAs.
template <int HD>
__device__ __forceinline__ void fold_tile_gemm(const float* At, const float* Bt, float (&acc)[HD / 16][4]) {
  constexpr int TR = HD / 16;
  const int r0 = TR * (threadIdx.x >> 4), c0 = 4 * (threadIdx.x & 15);
#pragma unroll
  for (int i = 0; i < TR; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
#pragma unroll 4
  for (int k = 0; k < HD; ++k) {
    float a[TR];
#pragma unroll
    for (int i = 0; i < TR; ++i) a[i] = At[k * HD + r0 + i];
    const float4 bv = *reinterpret_cast<const float4*>(Bt + k * FOLD_TC + c0);
#pragma unroll
    for (int i = 0; i < TR; ++i) {
      acc[i][0] = fmaf(a[i], bv.x, acc[i][0]);
      acc[i][1] = fmaf(a[i], bv.y, acc[i][1]);
      acc[i][2] = fmaf(a[i], bv.z, acc[i][2]);
      acc[i][3] = fmaf(a[i], bv.w, acc[i][3]);
    }
  }
}

// grid (ceil((D+1)/64), 2H + 1): blockIdx.y < 2H -> one (s,h) block of HD rows; == 2H -> the q rows.
// BWD = false: out = Wl . in (+ bl on the bias column), written as T (and optionally fp32)
// BWD = true : out = Wl^T . in, fp32, optionally accumulated
template <typename T, int HD, bool BWD>
__global__ __launch_bounds__(256) void fold_w_kernel(const float* __restrict__ win, const float* __restrict__ bin,
                                                     const float* __restrict__ wl, const float* __restrict__ bl,
                                                     T* __restrict__ wout, float* __restrict__ wout_f32,
                                                     float* __restrict__ bout, int D, int H, int accumulate) {
  __shared__ __attribute__((aligned(16))) float sAt[HD * HD];
  __shared__ __attribute__((aligned(16))) float sBt[HD * FOLD_TC];
  const int tc = threadIdx.x & 63, tg = threadIdx.x >> 6;
  auto put = [&](long r, int c, float v) {
    if (c < D) {
      if (BWD && accumulate) v += to_f32(wout[r * D + c]);
      wout[r * D + c] = from_f32<T>(v);
      if (wout_f32) wout_f32[r * D + c] = v;
    } else if (c == D) {
      bout[r] = (BWD && accumulate) ? bout[r] + v : v;
    }
  };
  if ((int)blockIdx.y >= 2 * H) {                      // q part: copy / cast, HD rows per block
    const int c = blockIdx.x * FOLD_TC + tc;
    if (c > D) return;
    const int rb = ((int)blockIdx.y - 2 * H) * HD;
#pragma unroll 4
    for (int r = rb + tg; r < rb + HD && r < D; r += 4) put(r, c, (c < D) ? win[(long)r * D + c] : bin[r]);
    return;
  }
  const long base = (long)D + (long)blockIdx.y * HD;   // first row of this (s,h) block
  // At[k][r]: forward needs Wl[r][k] (transpose while staging), backward Wl[k][r] (as stored)
  for (int i = threadIdx.x; i < HD * HD; i += 256) {
    const int k = i / HD, r = i % HD;
    sAt[i] = BWD ? wl[k * HD + r] : wl[r * HD + k];
  }
  for (int k = tg; k < HD; k += 4) {
    const int c = blockIdx.x * FOLD_TC + tc;
    sBt[k * FOLD_TC + tc] = (c < D) ? win[(base + k) * D + c] : (c == D ? bin[base + k] : 0.f);
  }
  __syncthreads();
  float acc[HD / 16][4];
  fold_tile_gemm<HD>(sAt, sBt, acc);
  constexpr int TR = HD / 16;
  const int r0 = TR * (threadIdx.x >> 4), c0 = blockIdx.x * FOLD_TC + 4 * (threadIdx.x & 15);
#pragma unroll
  for (int i = 0; i < TR; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = acc[i][j];
      if (!BWD && c0 + j == D) v += bl[r0 + i];
      put(base + r0 + i, c0 + j, v);
    }
}

// dWl[i][j] += sum_c dWeff_z[i][c] Wqkv_z[j][c] + dbeff_z[i] bqkv_z[j];  dbl[i] += dbeff_z[i].
// grid (2H): one workgroup per (s,h) block z walks the D+1 columns in tiles of 64 (register-blocked
// [HD x HD] accumulator) and adds its partial with fp32 atomics (2H-way contention only).
template <int HD>
__global__ __launch_bounds__(256) void fold_bwd_l_kernel(const float* __restrict__ dweff,
                                                         const float* __restrict__ dbeff,
                                                         const float* __restrict__ wqkv,
                                                         const float* __restrict__ bqkv, float* __restrict__ dwl,
                                                         float* __restrict__ dbl, int D) {
  constexpr int TR = HD / 16;                          // thread block TR x TR of the [HD x HD] output
  constexpr int LDT = HD + 1;                          // padded: transposed staging is conflict-free
  __shared__ float sa[64 * LDT];                       // [c][i]  dWeff tile, transposed
  __shared__ float sb[64 * LDT];                       // [c][j]  Wqkv tile, transposed
  const int tc = threadIdx.x & 63, tg = threadIdx.x >> 6;
  const long base = (long)D + (long)blockIdx.x * HD;
  const int i0 = TR * (threadIdx.x >> 4), j0 = TR * (threadIdx.x & 15);
  const int c = blockIdx.y * 64 + tc;                  // column D = bias column
  float va[HD / 4], vb[HD / 4];
#pragma unroll
  for (int q = 0; q < HD / 4; ++q) {
    const int r = tg + 4 * q;
    va[q] = (c < D) ? dweff[(base + r) * D + c] : (c == D ? dbeff[base + r] : 0.f);
    vb[q] = (c < D) ? wqkv[(base + r) * D + c] : (c == D ? bqkv[base + r] : 0.f);
  }
#pragma unroll
  for (int q = 0; q < HD / 4; ++q) {
    sa[tc * LDT + tg + 4 * q] = va[q];
    sb[tc * LDT + tg + 4 * q] = vb[q];
  }
  __syncthreads();
  float acc[TR][TR];
#pragma unroll
  for (int i = 0; i < TR; ++i)
#pragma unroll
    for (int j = 0; j < TR; ++j) acc[i][j] = 0.f;
#pragma unroll 4
  for (int k = 0; k < 64; ++k) {
    float a[TR], bb[TR];
#pragma unroll
    for (int i = 0; i < TR; ++i) { a[i] = sa[k * LDT + i0 + i]; bb[i] = sb[k * LDT + j0 + i]; }
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
      for (int j = 0; j < TR; ++j) acc[i][j] = fmaf(a[i], bb[j], acc[i][j]);
  }
#pragma unroll
  for (int i = 0; i < TR; ++i)
#pragma unroll
    for (int j = 0; j < TR; ++j) atomicAdd(dwl + (long)(i0 + i) * HD + j0 + j, acc[i][j]);
  if (blockIdx.y == 0 && threadIdx.x < HD) atomicAdd(dbl + threadIdx.x, dbeff[base + threadIdx.x]);
}

template <typename T, bool BWD>
int launch_fold_w(const float* win, const float* bin, const float* wl, const float* bl, T* wout, float* wout_f32,
                  float* bout, int D, int H, int accumulate, hipStream_t st) {
  const int hd = D / H;
  const dim3 grid((D + 1 + FOLD_TC - 1) / FOLD_TC, 2 * H + (D + hd - 1) / hd);
  switch (hd) {
    case 16: hipLaunchKernelGGL((fold_w_kernel<T, 16, BWD>), grid, dim3(256), 0, st, win, bin, wl, bl, wout, wout_f32, bout, D, H, accumulate); break;
    case 32: hipLaunchKernelGGL((fold_w_kernel<T, 32, BWD>), grid, dim3(256), 0, st, win, bin, wl, bl, wout, wout_f32, bout, D, H, accumulate); break;
    case 64: hipLaunchKernelGGL((fold_w_kernel<T, 64, BWD>), grid, dim3(256), 0, st, win, bin, wl, bl, wout, wout_f32, bout, D, H, accumulate); break;
    default: return FAVIT_ERR_UNSUPPORTED;
  }
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

template <typename T, int DPL, int WMAX>
int launch_attn(bool bwd, const AttnArgs& a, hipStream_t st) {
  constexpr int HD = 8 * DPL;
  const int rs = row_stride_bytes(HD, sizeof(T));
  const int h = a.W / 2;
  if (!bwd) {
    const int rows = (FWD_QPB + 2 * h) + 2;
    const size_t lds = (size_t)2 * rows * rs;
    auto k = mhla_fwd_kernel<T, DPL, WMAX>;
    if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((a.L + FWD_QPB - 1) / FWD_QPB, a.H, a.B);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, a);
  } else {
    const int krows = (a.rb + 4 * h) + 2 * (2 * h + 1);
    const size_t lds = (size_t)2 * krows * rs + (size_t)2 * 64 * rs + (size_t)2 * 64 * WMAX * 4;
    if (lds > 160 * 1024) return FAVIT_ERR_UNSUPPORTED;
    auto k = mhla_bwd_kernel<T, DPL, WMAX>;
    if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((a.L + a.rb - 1) / a.rb, a.H, a.B);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, a);
  }
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

template <typename T, int WMAX>
int dispatch_dpl(bool bwd, const AttnArgs& a, hipStream_t st) {
  switch (a.hd) {
    case 16: return launch_attn<T, 2, WMAX>(bwd, a, st);
    case 32: return launch_attn<T, 4, WMAX>(bwd, a, st);
    case 64: return launch_attn<T, 8, WMAX>(bwd, a, st);
    case 128: return launch_attn<T, 16, WMAX>(bwd, a, st);
    default: return FAVIT_ERR_UNSUPPORTED;
  }
}

int attn_entry(bool bwd, const void* qkv, const void* dout, void* out, const uint8_t* mask, int B, int L, int H, int hd,
               int W, int dtype, float p, uint64_t seed, void* stream) {
  if (!qkv || !out || (bwd && !dout) || B <= 0 || L <= 0 || H <= 0 || hd <= 0) return FAVIT_ERR_INVALID;
  if (W <= 0 || (W & 1) == 0) return FAVIT_ERR_INVALID;      // even windows crash the reference (mhla.py:83)
  if (W > 15) return FAVIT_ERR_UNSUPPORTED;
  if (p < 0.f || p >= 1.f) return FAVIT_ERR_INVALID;
  AttnArgs a;
  a.qkv = qkv; a.dout = dout; a.out = out; a.mask = mask;
  a.B = B; a.L = L; a.H = H; a.hd = hd; a.W = W;
  a.rb = BWD_RB - 2 * (W / 2);
  a.inv_sqrt_hd = 0.f;
  a.thresh = dropout_threshold(p);
  a.keep_scale = 1.0f / (1.0f - p);
  a.seed = seed;
  hipStream_t st = as_stream(stream);
  if (!bwd && dtype == FAVIT_BF16 && (hd == 32 || hd == 64 || hd == 128) && getenv("FAVIT_MHLA_VALU") == nullptr) {
    const int h = W / 2;
    const size_t lds = (size_t)2 * (64 + 2 * h + 2) * (hd * 2 + 16);
    dim3 grid((L + 63) / 64, H, B);
    if (hd == 32) hipLaunchKernelGGL(mhla_fwd_mfma_kernel<32>, grid, dim3(256), lds, st, a);
    else if (hd == 64) hipLaunchKernelGGL(mhla_fwd_mfma_kernel<64>, grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL(mhla_fwd_mfma_kernel<128>, grid, dim3(256), lds, st, a);
    FAVIT_CHECK_LAUNCH();
    return FAVIT_OK;
  }
  if (dtype == FAVIT_F32) return W <= 7 ? dispatch_dpl<float, 7>(bwd, a, st) : dispatch_dpl<float, 15>(bwd, a, st);
  if (dtype == FAVIT_BF16) return W <= 7 ? dispatch_dpl<bf16_t, 7>(bwd, a, st) : dispatch_dpl<bf16_t, 15>(bwd, a, st);
  return FAVIT_ERR_INVALID;
}

}  // namespace

extern "C" int favit_mhla_attn_fwd(const void* qkv, void* out, const uint8_t* mask, int32_t B, int32_t L, int32_t H,
                                   int32_t hd, int32_t W, int dtype, float dropout_p, uint64_t seed, void* stream) {
  return attn_entry(false, qkv, nullptr, out, mask, B, L, H, hd, W, dtype, dropout_p, seed, stream);
}

extern "C" int favit_mhla_attn_bwd(const void* qkv, const void* dout, void* dqkv, const uint8_t* mask, int32_t B,
                                   int32_t L, int32_t H, int32_t hd, int32_t W, int dtype, float dropout_p,
                                   uint64_t seed, void* stream) {
  return attn_entry(true, qkv, dout, dqkv, mask, B, L, H, hd, W, dtype, dropout_p, seed, stream);
}

extern "C" int favit_mhla_fold_fwd(const float* wqkv, const float* bqkv, const float* wl, const float* bl, void* weff,
                                   int weff_dtype, float* weff_f32, float* beff, int32_t D, int32_t H, void* stream) {
  if (!wqkv || !bqkv || !wl || !bl || !weff || !beff || D <= 0 || H <= 0 || D % H) return FAVIT_ERR_INVALID;
  hipStream_t st = as_stream(stream);
  if (weff_dtype == FAVIT_F32)
    return launch_fold_w<float, false>(wqkv, bqkv, wl, bl, (float*)weff, weff_f32, beff, D, H, 0, st);
  if (weff_dtype == FAVIT_BF16)
    return launch_fold_w<bf16_t, false>(wqkv, bqkv, wl, bl, (bf16_t*)weff, weff_f32, beff, D, H, 0, st);
  return FAVIT_ERR_INVALID;
}

extern "C" int favit_mhla_fold_bwd(const float* dweff, const float* dbeff, const float* wqkv, const float* bqkv,
                                   const float* wl, float* dwqkv, float* dbqkv, float* dwl, float* dbl, int32_t D,
                                   int32_t H, int32_t accumulate, void* stream) {
  if (!dweff || !dbeff || !wqkv || !bqkv || !wl || !dwqkv || !dbqkv || !dwl || !dbl || D <= 0 || H <= 0 || D % H)
    return FAVIT_ERR_INVALID;
  const int hd = D / H;
  hipStream_t st = as_stream(stream);
  const int rc = launch_fold_w<float, true>(dweff, dbeff, wl, nullptr, dwqkv, nullptr, dbqkv, D, H, accumulate, st);
  if (rc != FAVIT_OK) return rc;
  if (!accumulate) {
    (void)hipMemsetAsync(dwl, 0, sizeof(float) * hd * hd, st);
    (void)hipMemsetAsync(dbl, 0, sizeof(float) * hd, st);
  }
  switch (hd) {
    case 16: hipLaunchKernelGGL(fold_bwd_l_kernel<16>, dim3(2 * H, (D + 64) / 64), dim3(256), 0, st, dweff, dbeff, wqkv, bqkv, dwl, dbl, D); break;
    case 32: hipLaunchKernelGGL(fold_bwd_l_kernel<32>, dim3(2 * H, (D + 64) / 64), dim3(256), 0, st, dweff, dbeff, wqkv, bqkv, dwl, dbl, D); break;
    case 64: hipLaunchKernelGGL(fold_bwd_l_kernel<64>, dim3(2 * H, (D + 64) / 64), dim3(256), 0, st, dweff, dbeff, wqkv, bqkv, dwl, dbl, D); break;
    default: return FAVIT_ERR_UNSUPPORTED;
  }
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}
