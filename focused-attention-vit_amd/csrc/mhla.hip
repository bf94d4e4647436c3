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
  // like slot(), but rows that are not staged at all map to the nearest staged main row (used for
  // slots whose result is discarded, so that no uninitialised LDS is ever read)
  __device__ __forceinline__ int slot_safe(int r) const {
    if (r >= main_lo && r < main_hi) return r - main_lo;
    if (r < head_n) return n_main + r;
    if (r >= tail_lo) return n_main + head_n + (r - tail_lo);
    return (r < main_lo ? 0 : n_main - 1);
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

// Two-phase staging: ALL global loads of a thread are issued (into registers) before the first LDS
// store, so their latencies overlap; the single-loop stage_rows() above made every 16-byte chunk
// wait for its own load (vmcnt(0) per iteration) and dominated the workgroup's lifetime.
template <typename T, int NCH>
struct RowStager {
  uint4 regs[NCH];
  template <typename F>
  __device__ __forceinline__ void load(int nrows, const T* src, long ld, long tok0, int col0, int hd, int tid, F row_of) {
    const int cpr = hd * (int)sizeof(T) / 16;
    const int total = nrows * cpr;
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int c = min(tid + 256 * it, total - 1);      // clamped: unconditional load, store is predicated
      const int s = c / cpr, ch = c - s * cpr;
      regs[it] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(src + (tok0 + row_of(s)) * ld + col0) + ch * 16);
    }
  }
  __device__ __forceinline__ void store(char* lds, int nrows, int hd, int rs, int tid) const {
    const int cpr = hd * (int)sizeof(T) / 16;
    const int total = nrows * cpr;
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int c = tid + 256 * it;
      if (c < total) {
        const int s = c / cpr, ch = c - s * cpr;
        *reinterpret_cast<uint4*>(lds + s * rs + ch * 16) = regs[it];
      }
    }
  }
};

// Two matrices staged with ONE row mapping (K~ and V~ share their rows, Q and dO theirs): the slot -> row index
// arithmetic and the chunk offsets are computed once per 16-byte chunk instead of once per matrix (these kernels
// are bound by their VALU instruction count, not by memory).
template <typename T, int NCH>
struct PairStager {
  uint4 ra[NCH], rb[NCH];
  template <typename F>
  __device__ __forceinline__ void load(int nrows, const T* srcA, long ldA, int colA, const T* srcB, long ldB, int colB,
                                       long tok0, int hd, int tid, F row_of) {
    const int cpr = hd * (int)sizeof(T) / 16;
    const int total = nrows * cpr;
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int c = min(tid + 256 * it, total - 1);      // clamped: unconditional load, store is predicated
      const int s = c / cpr, ch = c - s * cpr;
      const long row = tok0 + row_of(s);
      ra[it] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(srcA + row * ldA + colA) + ch * 16);
      rb[it] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(srcB + row * ldB + colB) + ch * 16);
    }
  }
  __device__ __forceinline__ void store(char* ldsA, char* ldsB, int nrows, int hd, int rs, int tid) const {
    const int cpr = hd * (int)sizeof(T) / 16;
    const int total = nrows * cpr;
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int c = tid + 256 * it;
      if (c < total) {
        const int s = c / cpr, ch = c - s * cpr;
        *reinterpret_cast<uint4*>(ldsA + s * rs + ch * 16) = ra[it];
        *reinterpret_cast<uint4*>(ldsB + s * rs + ch * 16) = rb[it];
      }
    }
  }
  // Branch-free variant: chunks past the end go to a 1-KiB dump area (one 16-byte slot per lane).  With the predicated
  // store() the compiler sinks every load into its store's block and the chunks of a thread make their round trips one
  // after the other (see mhla_bwd_lse_kernel); this keeps all of them in flight.
  __device__ __forceinline__ void store_dump(char* ldsA, char* ldsB, char* dump, int nrows, int hd, int rs, int tid) const {
    const int cpr = hd * (int)sizeof(T) / 16;
    const int total = nrows * cpr;
    char* mine = dump + (tid & 63) * 16;
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int c = tid + 256 * it;
      const int s = c / cpr, ch = c - s * cpr;
      const bool ok = c < total;
      *reinterpret_cast<uint4*>(ok ? ldsA + s * rs + ch * 16 : mine) = ra[it];
      *reinterpret_cast<uint4*>(ok ? ldsB + s * rs + ch * 16 : mine) = rb[it];
    }
  }
};

struct AttnArgs {
  const void* qkv;
  const void* dout;
  void* out;      // fwd: o ; bwd: dqkv
  const uint8_t* mask;
  int B, L, H, hd, W;
  int rb;              // backward: key rows per workgroup (attn_entry sizes it to the LDS regions)
  int qcap;            // backward: rows of the Q / dO images and of the dS / P tables (>= rb + 3h + 1)
#ifdef FAVIT_PROBE
  int dbg;             // probe build only (make probe; tools/): skips work, never in libfavit.so
#endif
  float inv_sqrt_hd;   // unused (true division is applied), kept for clarity
  uint32_t thresh;
  float keep_scale;
  uint64_t seed;
  const unsigned long long* epoch;   // favit_set_dropout_epoch word (or null)
  float* lse_out;      // forward (MFMA kernel): log-sum-exp of every (batch, head, row) [B, H, L], or null
  const float* lse;    // backward, "saved statistics" kernel: the forward's lse ...
  const void* o;       // ... and the forward's output [B*L, D] (delta of the halo rows = dO . O)
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

template <typename T, int DPL, int WMAX>
__global__ __launch_bounds__(256) void mhla_fwd_kernel(AttnArgs a) {
  if (a.thresh) a.seed = favit_eff_seed(a.seed, a.epoch);
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
  {
    constexpr int NCH = ((FWD_QPB + 16) * (HD * (int)sizeof(T) / 16) + 255) / 256;
    RowStager<T, NCH> sk, sv;
    auto rowf = [&](int s) { return im.row_of_slot(s); };
    sk.load(im.n_rows, qkv, ld, tok0, D + head * HD, HD, tid, rowf);
    sv.load(im.n_rows, qkv, ld, tok0, 2 * D + head * HD, HD, tid, rowf);
    sk.store(ldsK, im.n_rows, HD, rs, tid);
    sv.store(ldsV, im.n_rows, HD, rs, tid);
  }
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
  if (a.thresh) a.seed = favit_eff_seed(a.seed, a.epoch);
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
  char* ldsG = ldsQ + a.qcap * rs;                           // dO rows
  float* tds = reinterpret_cast<float*>(ldsG + a.qcap * rs); // [qcap][WMAX] dS / sqrt(hd)
  float* tp = tds + a.qcap * WMAX;                           // [qcap][WMAX] P after dropout

  {
    constexpr int CPR = HD * (int)sizeof(T) / 16;
    constexpr int NKV = (80 * CPR + 255) / 256, NQ = (64 * CPR + 255) / 256;
    RowStager<T, NKV> sk, sv;
    RowStager<T, NQ> sq, sg;
    auto rowk = [&](int s) { return imK.row_of_slot(s); };
    auto rowq = [&](int s) { return qrow_of(s); };
    sk.load(imK.n_rows, qkv, ld, tok0, D + head * HD, HD, tid, rowk);
    sv.load(imK.n_rows, qkv, ld, tok0, 2 * D + head * HD, HD, tid, rowk);
    sq.load(n_q, qkv, ld, tok0, head * HD, HD, tid, rowq);
    sg.load(n_q, dout, (long)D, tok0, head * HD, HD, tid, rowq);
    sk.store(ldsK, imK.n_rows, HD, rs, tid);
    sv.store(ldsV, imK.n_rows, HD, rs, tid);
    sq.store(ldsQ, n_q, HD, rs, tid);
    sg.store(ldsG, n_q, HD, rs, tid);
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
  for (int jb = r0; jb < r1; jb += 32) {
    const int j = jb + qs;
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


// Output tiles of the transposed products (O^T, dQ^T, dK^T, dV^T) in a column order that lets a lane store 32
// contiguous bytes: tile t = 4*half + sub of the MFMA sequence covers the columns d = 64*half + 16*(m/4) + 4*sub + m%4
// (m = MFMA row), i.e. lane group g ends up with the 16 consecutive columns 64*half + 16g .. +15 over sub = 0..3.
// The permutation costs nothing: a ds_read_b64_tr_b16 lane supplies its own column address.
__device__ __forceinline__ int tr_col(int t, int p4) { return 64 * (t >> 2) + 16 * p4 + 4 * (t & 3); }

template <int HD>
__device__ __forceinline__ void store_rows16(bf16_t* row, const f32x4 (&o)[HD / 16], int g) {
#pragma unroll
  for (int half = 0; half < HD / 64; ++half) {
    bf16x8 lo, hi;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      lo[r] = (bf16_t)o[4 * half + 0][r];
      lo[4 + r] = (bf16_t)o[4 * half + 1][r];
      hi[r] = (bf16_t)o[4 * half + 2][r];
      hi[4 + r] = (bf16_t)o[4 * half + 3][r];
    }
    bf16_t* d = row + 64 * half + 16 * g;
    *reinterpret_cast<bf16x8*>(d) = lo;
    *reinterpret_cast<bf16x8*>(d + 8) = hi;
  }
}

template <int HD, bool PLAIN>            // PLAIN: no mask, no dropout (compiled out)
__global__ __launch_bounds__(256) void mhla_fwd_mfma_kernel(AttnArgs a) {
  if (a.thresh) a.seed = favit_eff_seed(a.seed, a.epoch);
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
  bf16x8 qfr[HD / 32];
  {
    constexpr int NCH = (80 * (HD * 2 / 16) + 255) / 256;
    PairStager<bf16_t, NCH> skv;
    auto rowf = [&](int s) { return im.row_of_slot(s); };
    skv.load(im.n_rows, qkv, ld, D + head * HD, qkv, ld, 2 * D + head * HD, tok0, HD, tid, rowf);
    // this lane's query fragments are requested together with the K~ / V~ rows (they used to be loaded after the
    // barrier: one more exposed round trip per workgroup)
    const int tq = min(r0 + 16 * wave + qi, L - 1);
    const bf16_t* qrow0 = qkv + (tok0 + tq) * ld + head * HD;
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) qfr[ks] = *reinterpret_cast<const bf16x8*>(qrow0 + 32 * ks + 8 * g);
    skv.store_dump(ldsK, ldsV, ldsV + im.n_rows * RS, im.n_rows, HD, RS, tid);
  }
  __syncthreads();

#ifdef FAVIT_PROBE
  if (a.dbg == 1) return;
#endif
  const int t0 = r0 + 16 * wave;
  if (t0 >= r1) return;                                // whole wave idle (no barrier after this point)
  const int i = t0 + qi;
  const bool qvalid = i < L;
  SlotInfo si;
  si.init(min(i, L - 1), L, W, h);

  // ---- S = K_slots . Q^T ----
  f32x4 S[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  const int krow0 = im.slot(slot_key(qi, t0, h, L)), krow1 = im.slot(slot_key(16 + qi, t0, h, L));
#pragma unroll
  for (int ks = 0; ks < HD / 32; ++ks) {
    const bf16x8 qf = qfr[ks];
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
    if (!PLAIN && mu > 0 && a.mask && a.mask[((long)b * L + si.i) * L + j] == 0) mu = 0;     // mhla.py:143
    float kwe = (float)mu;
    if (!PLAIN && a.thresh && mu > 0) {                            // dropout acts on every window copy separately
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
  mx = quad16_max(mx);
  float lsum = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = (mult[e] > 0.f) ? __expf(sc[e] - mx) : 0.f;
    lsum = fmaf(mult[e], sc[e], lsum);
  }
  lsum = quad16_sum(lsum);
  const float inv_l = 1.0f / lsum;
  if (a.lse_out && g == 0 && qvalid) a.lse_out[((long)b * a.H + head) * L + i] = mx + __logf(lsum);
  bf16x8 pf;
#pragma unroll
  for (int e = 0; e < 8; ++e) pf[e] = (bf16_t)(sc[e] * inv_l * kw[e]);

  // ---- O^T = V_slots^T . P ----
  const int q4 = qi >> 2, p4 = qi & 3;
  const int vrow0 = im.slot(slot_key(4 * g + q4, t0, h, L)), vrow1 = im.slot(slot_key(16 + 4 * g + q4, t0, h, L));
  bf16_t* orow = reinterpret_cast<bf16_t*>(a.out) + (tok0 + si.i) * (long)D + head * HD;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  f32x4 oo[HD / 16];
#pragma unroll
  for (int dt = 0; dt < HD / 16; ++dt) {
    const int coff = (HD % 64 == 0 ? tr_col(dt, p4) : 16 * dt + 4 * p4) * 2;
    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(ldsV + vrow0 * RS + coff));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(ldsV + vrow1 * RS + coff));
    s16x8 vv;
    vv[0] = lo4[0]; vv[1] = lo4[1]; vv[2] = lo4[2]; vv[3] = lo4[3];
    vv[4] = hi4[0]; vv[5] = hi4[1]; vv[6] = hi4[2]; vv[7] = hi4[3];
    oo[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vv), pf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
  }
#ifdef FAVIT_PROBE
  if (qvalid && (a.dbg != 2 || oo[0][0] == 12345.f)) {
#else
  if (qvalid) {
#endif
    if constexpr (HD % 64 == 0) {
      store_rows16<HD>(orow, oo, g);
    } else {
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) {
        bf16x4 ob = {(bf16_t)oo[dt][0], (bf16_t)oo[dt][1], (bf16_t)oo[dt][2], (bf16_t)oo[dt][3]};
        *reinterpret_cast<bf16x4*>(orow + 16 * dt + 4 * g) = ob;
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// MFMA backward (bf16, hd in {32,64,128}; W <= 7, or W <= 11 when L > 16).
// Phase 1 (query tiles of 16 rows, recomputed like the forward): S, P, dP = V.dO^T, dS, and
//   dQ^T = K_slots^T . dS; dS and the dropped probabilities go to LDS tables [query][32 slots].
// Phase 2 (key tiles of 16 rows owned by this workgroup): every key gathers its column of the
//   tables over the 16+2h band queries (+ the wrap rows for keys 0 / L-1) into a B fragment and
//   dK^T = Q^T . W_dS, dV^T = dO^T . W_P run on MFMA with Q^T / dO^T fragments read by
//   ds_read_b64_tr_b16.  No atomics: halo query rows are recomputed (deterministic).
// ---------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void mhla_bwd_mfma_kernel(AttnArgs a) {
  if (a.thresh) a.seed = favit_eff_seed(a.seed, a.epoch);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RS = HD * 2 + 16;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, qi = lane & 15, q4 = qi >> 2, p4 = qi & 3;
  const int head = blockIdx.y, b = blockIdx.z;
  const int r0 = blockIdx.x * 64, r1 = min(a.L, r0 + 64);
  const int L = a.L, W = a.W, h = W >> 1, D = a.H * HD;
  const long ld = 3L * D, tok0 = (long)b * L;
  const bf16_t* qkv = reinterpret_cast<const bf16_t*>(a.qkv);
  const bf16_t* dout = reinterpret_cast<const bf16_t*>(a.dout);
  bf16_t* dqkv = reinterpret_cast<bf16_t*>(a.out);

  // query tiles: main rows [qm_lo, qm_hi) in tiles of 16, then (if this block owns key L-1) the rows
  // 0.. that END-pad onto it, then (if it owns key 0) the rows tx_lo.. that FRONT-pad onto it
  const int qm_lo = max(0, r0 - h), qm_hi = min(L, r1 + h);
  const int nmt = (qm_hi - qm_lo + 15) >> 4;
  const int has_hx = (r1 >= L) ? 1 : 0;
  const int tx_lo = max(h + 1, L - h);
  const int has_tx = (r0 == 0 && tx_lo < L) ? 1 : 0;
  const int ntiles = nmt + has_hx + has_tx;
  auto tile_t0 = [&](int t) { return t < nmt ? qm_lo + 16 * t : ((has_hx && t == nmt) ? 0 : tx_lo); };
  // table / Q-image slot of query row i for the band (main) part
  auto main_slot = [&](int i) { return min(max(i, qm_lo), qm_hi - 1) - qm_lo; };
  auto row_slot = [&](int i, bool head_kind) {           // slot of a wrap row
    if (i >= qm_lo && i < qm_hi) return i - qm_lo;
    return head_kind ? 16 * nmt + i : 16 * (nmt + has_hx) + (i - tx_lo);
  };

  RowImage imK;
  imK.init(r0 - 2 * h, r1 + 2 * h, 2 * h + 1, L);
  char* ldsK = smem;
  char* ldsV = ldsK + imK.n_rows * RS;
  char* ldsQ = ldsV + imK.n_rows * RS;
  char* ldsG = ldsQ + 16 * ntiles * RS;
  bf16_t* tds = reinterpret_cast<bf16_t*>(ldsG + 16 * ntiles * RS);      // [16*ntiles][32]
  bf16_t* tp = tds + 16 * ntiles * 32;

  {
    constexpr int CPR = HD * 2 / 16;
    constexpr int NKV = (108 * CPR + 255) / 256, NQ = (112 * CPR + 255) / 256;
    RowStager<bf16_t, NKV> sk, sv;
    RowStager<bf16_t, NQ> sq, sg;
    auto rowk = [&](int s) { return imK.row_of_slot(s); };
    auto rowq = [&](int s) { return min(tile_t0(s >> 4) + (s & 15), L - 1); };
    sk.load(imK.n_rows, qkv, ld, tok0, D + head * HD, HD, tid, rowk);
    sv.load(imK.n_rows, qkv, ld, tok0, 2 * D + head * HD, HD, tid, rowk);
    sq.load(16 * ntiles, qkv, ld, tok0, head * HD, HD, tid, rowq);
    sg.load(16 * ntiles, dout, (long)D, tok0, head * HD, HD, tid, rowq);
    sk.store(ldsK, imK.n_rows, HD, RS, tid);
    sv.store(ldsV, imK.n_rows, HD, RS, tid);
    sq.store(ldsQ, 16 * ntiles, HD, RS, tid);
    sg.store(ldsG, 16 * ntiles, HD, RS, tid);
  }
  __syncthreads();

  const float inv_sq = 1.0f / sqrtf((float)HD);
  // ---------------- phase 1: query tiles ----------------
  for (int tile = wave; tile < ntiles; tile += 4) {
    const int t0 = tile_t0(tile);
    const int i = t0 + qi;
    SlotInfo si;
    si.init(min(i, L - 1), L, W, h);
    const int krow0 = imK.slot_safe(slot_key(qi, t0, h, L)), krow1 = imK.slot_safe(slot_key(16 + qi, t0, h, L));
    const char* qimg = ldsQ + (16 * tile + qi) * RS;
    const char* gimg = ldsG + (16 * tile + qi) * RS;
    f32x4 S[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    f32x4 dP[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) {
      const int off = (32 * ks + 8 * g) * 2;
      const bf16x8 qf = *reinterpret_cast<const bf16x8*>(qimg + off);
      const bf16x8 gf = *reinterpret_cast<const bf16x8*>(gimg + off);
      const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(ldsK + krow0 * RS + off);
      const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(ldsK + krow1 * RS + off);
      const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(ldsV + krow0 * RS + off);
      const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(ldsV + krow1 * RS + off);
      S[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf, S[0], 0, 0, 0);
      S[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf, S[1], 0, 0, 0);
      dP[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v0, gf, dP[0], 0, 0, 0);
      dP[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1, gf, dP[1], 0, 0, 0);
    }
    float sc[8], mult[8], kw[8];
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int kt = e >> 2, r = e & 3;
      const int slot = 16 * kt + 4 * g + r;
      int j, mu, w0;
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
      if (mu > 0 && a.mask && a.mask[((long)b * L + si.i) * L + j] == 0) mu = 0;
      float kwe = (float)mu;
      if (a.thresh && mu > 0) {
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
    mx = quad16_max(mx);
    float lsum = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[e] = (mult[e] > 0.f) ? __expf(sc[e] - mx) : 0.f;
      lsum = fmaf(mult[e], sc[e], lsum);
    }
    lsum = quad16_sum(lsum);
    const float inv_l = 1.0f / lsum;
    float dot = 0.f;
    float dpn[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[e] *= inv_l;                                   // pn
      dpn[e] = kw[e] * dP[e >> 2][e & 3];
      dot = fmaf(sc[e], dpn[e], dot);
    }
    dot = quad16_sum(dot);
    bf16x8 dsf;
    bf16x4 pw0, pw1, ds0, ds1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float ds = (sc[e] * dpn[e] - mult[e] * sc[e] * dot) * inv_sq;
      const bf16_t dsb = (bf16_t)ds, pwb = (bf16_t)(sc[e] * kw[e]);
      dsf[e] = dsb;
      if (e < 4) { ds0[e] = dsb; pw0[e] = pwb; } else { ds1[e - 4] = dsb; pw1[e - 4] = pwb; }
    }
    {
      bf16_t* trow = tds + (16 * tile + qi) * 32 + 4 * g;
      bf16_t* prow = tp + (16 * tile + qi) * 32 + 4 * g;
      *reinterpret_cast<bf16x4*>(trow) = ds0;
      *reinterpret_cast<bf16x4*>(trow + 16) = ds1;
      *reinterpret_cast<bf16x4*>(prow) = pw0;
      *reinterpret_cast<bf16x4*>(prow + 16) = pw1;
    }
    // dQ^T = K_slots^T . dS  (only rows this block owns; main tiles only)
    const bool own = tile < nmt && i >= r0 && i < r1;
    const int vrow0 = imK.slot_safe(slot_key(4 * g + q4, t0, h, L)), vrow1 = imK.slot_safe(slot_key(16 + 4 * g + q4, t0, h, L));
    bf16_t* dqrow = dqkv + (tok0 + si.i) * ld + head * HD;
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) {
      const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(ldsK + vrow0 * RS + (16 * dt + 4 * p4) * 2));
      const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(ldsK + vrow1 * RS + (16 * dt + 4 * p4) * 2));
      s16x8 kk;
      kk[0] = lo4[0]; kk[1] = lo4[1]; kk[2] = lo4[2]; kk[3] = lo4[3];
      kk[4] = hi4[0]; kk[5] = hi4[1]; kk[6] = hi4[2]; kk[7] = hi4[3];
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kk), dsf, o, 0, 0, 0);
      if (own) {
        bf16x4 ob = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
        *reinterpret_cast<bf16x4*>(dqrow + 16 * dt + 4 * g) = ob;
      }
    }
  }
  __syncthreads();

  // ---------------- phase 2: key tiles ----------------
  const int k0 = r0 + 16 * wave;
  if (k0 >= r1) return;
  const int j = k0 + qi;                                  // this lane's key (B-operand column)
  const bool has0 = (k0 == 0), hasL = (L - 1 >= k0 && L - 1 < k0 + 16);
  const int nband = 16 + 2 * h;
  const int nhx = hasL ? min(h, L - 1) + 1 : 0;           // rows 0..min(h,L-1) END-pad onto key L-1
  // q-slot -> (query row, table slot, kind): kind 0 band, 1 END-pad row (key L-1), 2 FRONT-pad row (key 0)
  auto qs_row = [&](int qs, int& row, int& tslot, int& kind) {
    if (qs < nband) {
      kind = 0;
      row = k0 - h + qs;
      tslot = main_slot(row);
    } else if (qs < nband + nhx) {
      kind = 1;
      row = qs - nband;
      tslot = row_slot(row, true);
    } else {
      kind = 2;
      row = tx_lo + (qs - nband - nhx);
      if (!has0 || row >= L) { kind = 3; row = 0; tslot = main_slot(k0); return; }
      tslot = row_slot(row, false);
    }
  };
  // B fragments: lane = key j, 8 q-slots 8g..8g+7
  bf16x8 wds, wp;
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    int row, tslot, kind;
    qs_row(8 * g + jj, row, tslot, kind);
    bf16_t vds = (bf16_t)0.f, vp = (bf16_t)0.f;
    if (kind == 0) {
      if (row >= 0 && row < L && j < L && abs(row - j) <= h) {
        const int t0i = qm_lo + 16 * ((row - qm_lo) >> 4);
        const int sl = j - t0i + h;
        vds = tds[tslot * 32 + sl];
        vp = tp[tslot * 32 + sl];
      }
    } else if (kind == 1) {
      if (j == L - 1) { vds = tds[tslot * 32 + 31]; vp = tp[tslot * 32 + 31]; }
    } else if (kind == 2) {
      if (j == 0 && L > 1) { vds = tds[tslot * 32 + 30]; vp = tp[tslot * 32 + 30]; }
    }
    wds[jj] = vds;
    wp[jj] = vp;
  }
  // A fragments: Q^T / dO^T rows of q-slots 8g+q4 and 8g+4+q4 (every lane supplies one row address)
  int ra, rb_, ta, tb, ka, kb;
  qs_row(8 * g + q4, ra, ta, ka);
  qs_row(8 * g + 4 + q4, rb_, tb, kb);
  const bool jvalid = j < r1;
  bf16_t* dkrow = dqkv + (tok0 + min(j, L - 1)) * ld + D + head * HD;
  bf16_t* dvrow = dkrow + D;
#pragma unroll
  for (int dt = 0; dt < HD / 16; ++dt) {
    const int coff = (16 * dt + 4 * p4) * 2;
    const s16x4 qa = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(ldsQ + ta * RS + coff));
    const s16x4 qb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(ldsQ + tb * RS + coff));
    const s16x4 ga = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(ldsG + ta * RS + coff));
    const s16x4 gb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(ldsG + tb * RS + coff));
    s16x8 qq, gg;
    qq[0] = qa[0]; qq[1] = qa[1]; qq[2] = qa[2]; qq[3] = qa[3]; qq[4] = qb[0]; qq[5] = qb[1]; qq[6] = qb[2]; qq[7] = qb[3];
    gg[0] = ga[0]; gg[1] = ga[1]; gg[2] = ga[2]; gg[3] = ga[3]; gg[4] = gb[0]; gg[5] = gb[1]; gg[6] = gb[2]; gg[7] = gb[3];
    f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
    dk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, qq), wds, dk, 0, 0, 0);
    dv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, gg), wp, dv, 0, 0, 0);
    if (jvalid) {
      bf16x4 kb4 = {(bf16_t)dk[0], (bf16_t)dk[1], (bf16_t)dk[2], (bf16_t)dk[3]};
      bf16x4 vb4 = {(bf16_t)dv[0], (bf16_t)dv[1], (bf16_t)dv[2], (bf16_t)dv[3]};
      *reinterpret_cast<bf16x4*>(dkrow + 16 * dt + 4 * g) = kb4;
      *reinterpret_cast<bf16x4*>(dvrow + 16 * dt + 4 * g) = vb4;
    }
  }
}

// ---------------------------------------------------------------------------------
// MFMA backward, second formulation ("two owner passes", bf16, hd in {32,64,128}, W <= 11): no dS / P tables.
// A workgroup owns `rb` (<= 64, balanced over L) consecutive rows as QUERIES and as KEYS; four waves, one
// 16-row tile each per pass.
//   pass 1 (owner = 16 queries, 32 key slots, as the forward): S and dP = dO.V^T on MFMA, softmax with the slot
//           multiplicities, dS -> dQ^T = K_slots^T . dS for the rows the block owns.  Every query tile the block's
//           keys can see (its own rows, the h halo rows on each side, the wrap rows of keys 0 / L-1) leaves only
//           two numbers per row in LDS: lse_i and delta_i = sum_w P_w dP_w.
//   pass 2 (owner = 16 keys, 32 QUERY slots: the 16 + 2h band queries plus the wrap rows): S^T and dP^T are
//           RECOMPUTED on MFMA from the staged Q / dO rows (8 MFMAs at hd = 64 -- cheaper than keeping two
//           [rows][32] tables in LDS and gathering them), P = mult * exp(s - lse_i), dS = P (kw dP - mult
//           delta_i); dK^T = Q_slots^T . dS and dV^T = dO_slots^T . Pd with ds_read_b64_tr_b16 fragments.
// LDS: K~, V~ rows [r0 - 2h, r1 + 2h) + edges, Q, dO query tiles, 8 bytes of statistics per query row: 50 KiB at
// L = 197, W = 7, hd = 64 (three workgroups per CU; the table formulation needed 72 KiB -> two).
// Deterministic (no atomics); halo rows are recomputed.
// ---------------------------------------------------------------------------------
// PLAIN: no mask and no dropout (the training configurations of BASELINE.json): the per-element mask loads and
// dropout draws are compiled out.
template <int HD, bool PLAIN>
__global__ __launch_bounds__(256) void mhla_bwd_mfma2_kernel(AttnArgs a) {
  if (a.thresh) a.seed = favit_eff_seed(a.seed, a.epoch);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RS = HD * 2 + 16;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, qi = lane & 15, q4 = qi >> 2, p4 = qi & 3;
  const int head = blockIdx.y, b = blockIdx.z;
  const int r0 = blockIdx.x * a.rb, r1 = min(a.L, r0 + a.rb);
  const int L = a.L, W = a.W, h = W >> 1, D = a.H * HD;
  const long ld = 3L * D, tok0 = (long)b * L;
  const bf16_t* qkv = reinterpret_cast<const bf16_t*>(a.qkv);
  const bf16_t* dout = reinterpret_cast<const bf16_t*>(a.dout);
  bf16_t* dqkv = reinterpret_cast<bf16_t*>(a.out);

  // query tiles: main rows [qm_lo, qm_hi) in tiles of 16, then (if this block owns key L-1) the rows 0.. that
  // END-pad onto it, then (if it owns key 0) the rows tx_lo.. that FRONT-pad onto it
  const int qm_lo = max(0, r0 - h), qm_hi = min(L, r1 + h);
  const int nmt = (qm_hi - qm_lo + 15) >> 4;
  const int has_hx = (r1 >= L) ? 1 : 0;
  const int tx_lo = max(h + 1, L - h);
  const int has_tx = (r0 == 0 && tx_lo < L) ? 1 : 0;
  const int ntiles = nmt + has_hx + has_tx;
  auto tile_t0 = [&](int t) { return t < nmt ? qm_lo + 16 * t : ((has_hx && t == nmt) ? 0 : tx_lo); };
  auto main_slot = [&](int i) { return min(max(i, qm_lo), qm_hi - 1) - qm_lo; };
  auto row_slot = [&](int i, bool head_kind) {           // image / statistics slot of a wrap row
    if (i >= qm_lo && i < qm_hi) return i - qm_lo;
    return head_kind ? 16 * nmt + i : 16 * (nmt + has_hx) + (i - tx_lo);
  };

  RowImage imK;
  imK.init(r0 - 2 * h, r1 + 2 * h, 2 * h + 1, L);
  char* ldsK = smem;
  char* ldsV = ldsK + imK.n_rows * RS;
  char* ldsQ = ldsV + imK.n_rows * RS;
  char* ldsG = ldsQ + 16 * ntiles * RS;
  float* stat = reinterpret_cast<float*>(ldsG + 16 * ntiles * RS);      // [16*ntiles][2]: lse, delta

  {
    constexpr int CPR = HD * 2 / 16;
    constexpr int NKV = (108 * CPR + 255) / 256, NQ = (112 * CPR + 255) / 256;
    PairStager<bf16_t, NKV> skv;
    PairStager<bf16_t, NQ> sqg;
    auto rowk = [&](int s) { return imK.row_of_slot(s); };
    auto rowq = [&](int s) { return min(tile_t0(s >> 4) + (s & 15), L - 1); };
    skv.load(imK.n_rows, qkv, ld, D + head * HD, qkv, ld, 2 * D + head * HD, tok0, HD, tid, rowk);
    sqg.load(16 * ntiles, qkv, ld, head * HD, dout, (long)D, head * HD, tok0, HD, tid, rowq);
    skv.store(ldsK, ldsV, imK.n_rows, HD, RS, tid);
    sqg.store(ldsQ, ldsG, 16 * ntiles, HD, RS, tid);
  }
  __syncthreads();
#ifdef FAVIT_PROBE
  if (a.dbg & 2) return;                                 // probe: staging only
  const bool probe_nostore = (a.dbg & 1) != 0;
#else
  constexpr bool probe_nostore = false;
#endif

  const float inv_sq = 1.0f / sqrtf((float)HD);
  // ---------------- pass 1: query tiles ----------------
  for (int tile = wave; tile < ntiles; tile += 4) {
    const int t0 = tile_t0(tile);
    const int i = t0 + qi;
    SlotInfo si;
    si.init(min(i, L - 1), L, W, h);
    const int krow0 = imK.slot_safe(slot_key(qi, t0, h, L)), krow1 = imK.slot_safe(slot_key(16 + qi, t0, h, L));
    const char* qimg = ldsQ + (16 * tile + qi) * RS;
    const char* gimg = ldsG + (16 * tile + qi) * RS;
    f32x4 S[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    f32x4 dP[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) {
      const int off = (32 * ks + 8 * g) * 2;
      const bf16x8 qf = *reinterpret_cast<const bf16x8*>(qimg + off);
      const bf16x8 gf = *reinterpret_cast<const bf16x8*>(gimg + off);
      const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(ldsK + krow0 * RS + off);
      const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(ldsK + krow1 * RS + off);
      const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(ldsV + krow0 * RS + off);
      const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(ldsV + krow1 * RS + off);
      S[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf, S[0], 0, 0, 0);
      S[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf, S[1], 0, 0, 0);
      dP[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v0, gf, dP[0], 0, 0, 0);
      dP[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1, gf, dP[1], 0, 0, 0);
    }
    float sc[8], mult[8], kw[8];
    float mx = -INFINITY;
    // interior tile (wave-uniform): every row has its full window of W distinct keys, no wrap slots, no padding
    const bool interior = PLAIN && tile < nmt && t0 >= h && t0 + 15 + h <= L - 1;
    if (interior) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int slot = 16 * (e >> 2) + 4 * g + (e & 3);
        const int d = slot - h - qi;                                    // key - query
        const bool in = (slot < 16 + 2 * h) && (d >= -h) && (d <= h);
        sc[e] = S[e >> 2][e & 3] * inv_sq;
        mult[e] = in ? 1.f : 0.f;
        kw[e] = mult[e];
        if (in) mx = fmaxf(mx, sc[e]);
      }
    } else {
  #pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int kt = e >> 2, r = e & 3;
        const int slot = 16 * kt + 4 * g + r;
        int j, mu, w0;
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
        if (!PLAIN && mu > 0 && a.mask && a.mask[((long)b * L + si.i) * L + j] == 0) mu = 0;
        float kwe = (float)mu;
        if (!PLAIN && a.thresh && mu > 0) {
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
    }
    mx = quad16_max(mx);
    float lsum = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[e] = (mult[e] > 0.f) ? __expf(sc[e] - mx) : 0.f;
      lsum = fmaf(mult[e], sc[e], lsum);
    }
    lsum = quad16_sum(lsum);
    const float inv_l = 1.0f / lsum;
    float dot = 0.f;
    float dpn[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[e] *= inv_l;                                   // single-copy probability
      dpn[e] = kw[e] * dP[e >> 2][e & 3];
      dot = fmaf(sc[e], dpn[e], dot);
    }
    dot = quad16_sum(dot);
    if (g == 0) {
      stat[2 * (16 * tile + qi)] = mx + __logf(lsum);
      stat[2 * (16 * tile + qi) + 1] = dot;
    }
    // dQ^T = K_slots^T . dS  -- only for main tiles that contain rows this block owns
    if (tile < nmt && t0 + 15 >= r0 && t0 < r1) {
      bf16x8 dsf;
#pragma unroll
      for (int e = 0; e < 8; ++e) dsf[e] = (bf16_t)((sc[e] * dpn[e] - mult[e] * sc[e] * dot) * inv_sq);
      const bool own = i >= r0 && i < r1;
      const int vrow0 = imK.slot_safe(slot_key(4 * g + q4, t0, h, L)), vrow1 = imK.slot_safe(slot_key(16 + 4 * g + q4, t0, h, L));
      bf16_t* dqrow = dqkv + (tok0 + si.i) * ld + head * HD;
      if constexpr (HD % 64 == 0) {
        f32x4 oq[HD / 16];
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) {
          const int coff = tr_col(dt, p4) * 2;
          const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(ldsK + vrow0 * RS + coff));
          const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(ldsK + vrow1 * RS + coff));
          s16x8 kk;
          kk[0] = lo4[0]; kk[1] = lo4[1]; kk[2] = lo4[2]; kk[3] = lo4[3];
          kk[4] = hi4[0]; kk[5] = hi4[1]; kk[6] = hi4[2]; kk[7] = hi4[3];
          oq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kk), dsf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
        if (own && !(probe_nostore && oq[0][0] != 12345.f)) store_rows16<HD>(dqrow, oq, g);
      } else {
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) {
          const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(ldsK + vrow0 * RS + (16 * dt + 4 * p4) * 2));
          const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(ldsK + vrow1 * RS + (16 * dt + 4 * p4) * 2));
          s16x8 kk;
          kk[0] = lo4[0]; kk[1] = lo4[1]; kk[2] = lo4[2]; kk[3] = lo4[3];
          kk[4] = hi4[0]; kk[5] = hi4[1]; kk[6] = hi4[2]; kk[7] = hi4[3];
          f32x4 o = {0.f, 0.f, 0.f, 0.f};
          o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kk), dsf, o, 0, 0, 0);
          if (own && !(probe_nostore && o[0] != 12345.f)) {
            bf16x4 ob = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
            *reinterpret_cast<bf16x4*>(dqrow + 16 * dt + 4 * g) = ob;
          }
        }
      }
    }
  }
  __syncthreads();

  // ---------------- pass 2: key tiles ----------------
  const int k0 = r0 + 16 * wave;
  if (k0 >= r1) return;
  const int j = k0 + qi;                                  // this lane's key (B-operand column)
  const int jc = min(j, L - 1);
  const bool has0 = (k0 == 0), hasL = (L - 1 >= k0 && L - 1 < k0 + 16);
  const int nband = 16 + 2 * h;
  const int nhx = hasL ? min(h, L - 1) + 1 : 0;           // rows 0..min(h,L-1) END-pad onto key L-1
  // query slot -> (query row, image / statistics slot, kind): 0 band, 1 END-pad row, 2 FRONT-pad row, 3 unused
  auto qs_row = [&](int qs, int& row, int& tslot, int& kind) {
    if (qs < nband) {
      kind = 0;
      row = k0 - h + qs;
      tslot = main_slot(row);
    } else if (qs < nband + nhx) {
      kind = 1;
      row = qs - nband;
      tslot = row_slot(row, true);
    } else {
      kind = 2;
      row = tx_lo + (qs - nband - nhx);
      if (!has0 || row >= L) { kind = 3; row = 0; tslot = main_slot(k0); return; }
      tslot = row_slot(row, false);
    }
  };
  const bool interior_keys = PLAIN && !has0 && !hasL;       // wave-uniform: band query slots only, multiplicity 0 / 1
  auto img_slot = [&](int qs) {                             // Q / dO image row (and statistics slot) of a query slot
    if (interior_keys) return main_slot(k0 - h + qs);       // (clamped: slots past the band carry multiplicity 0)
    int row, tslot, kind;
    qs_row(qs, row, tslot, kind);
    return tslot;
  };
  // S^T[qs][key] = Q_qs . K_key,  dP^T[qs][key] = dO_qs . V_key   (A = the query slot rows, B = this lane's key row)
  f32x4 T1[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  f32x4 T2[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  {
    const int ts0 = img_slot(qi), ts1 = img_slot(16 + qi);
    const char* kimg = ldsK + imK.slot_safe(jc) * RS;
    const char* vimg = ldsV + imK.slot_safe(jc) * RS;
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) {
      const int off = (32 * ks + 8 * g) * 2;
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kimg + off);
      const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vimg + off);
      const bf16x8 qa = *reinterpret_cast<const bf16x8*>(ldsQ + ts0 * RS + off);
      const bf16x8 qb = *reinterpret_cast<const bf16x8*>(ldsQ + ts1 * RS + off);
      const bf16x8 ga = *reinterpret_cast<const bf16x8*>(ldsG + ts0 * RS + off);
      const bf16x8 gb = *reinterpret_cast<const bf16x8*>(ldsG + ts1 * RS + off);
      T1[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf, T1[0], 0, 0, 0);
      T1[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qb, kf, T1[1], 0, 0, 0);
      T2[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga, vf, T2[0], 0, 0, 0);
      T2[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gb, vf, T2[1], 0, 0, 0);
    }
  }
  // the lane holds query slots 16 st + 4g + r (e = 4 st + r) of its key: exactly the k-slice of the next MFMAs
  bf16x8 wds, wp;
  if (interior_keys) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int qs = 16 * (e >> 2) + 4 * g + (e & 3);
      const int row = k0 - h + qs, d = qs - h - qi;                    // query row, query - key
      const int tslot = min(max(row, qm_lo), qm_hi - 1) - qm_lo;
      const bool in = (qs < nband) && (row < L) && (j < L) && (d >= -h) && (d <= h);
      const float lse = stat[2 * tslot], dl = stat[2 * tslot + 1];
      const float pn = in ? __expf(T1[e >> 2][e & 3] * inv_sq - lse) : 0.f;
      wds[e] = (bf16_t)(pn * (T2[e >> 2][e & 3] - dl) * inv_sq);
      wp[e] = (bf16_t)pn;
    }
  } else {
  #pragma unroll
    for (int e = 0; e < 8; ++e) {
      int row, tslot, kind;
      qs_row(16 * (e >> 2) + 4 * g + (e & 3), row, tslot, kind);
      int mu = 0, w0 = 0;
      if (j < L && row >= 0 && row < L && kind != 3) {
        const int lo = max(0, row - h), hi = min(L, row + h + 1), n = hi - lo, pad = W - n;
        if (kind == 0) {
          if (j >= lo && j < hi) { mu = 1; w0 = (lo == 0 || pad == 0) ? (j - lo) : pad + (j - lo); }
        } else if (kind == 1) {
          if (j == L - 1 && lo == 0) { mu = pad; w0 = n; }              // END padding of the rows whose window starts at 0
        } else {
          if (j == 0 && lo > 0) { mu = pad; w0 = 0; }                   // FRONT padding
        }
        if (!PLAIN && mu > 0 && a.mask && a.mask[((long)b * L + row) * L + j] == 0) mu = 0;
      }
      float kwe = (float)mu;
      if (!PLAIN && a.thresh && mu > 0) {
        kwe = 0.f;
        for (int c = 0; c < mu; ++c) {
          const uint64_t idx = (((uint64_t)b * a.H + head) * L + row) * W + (w0 + c);
          kwe += favit_keep(a.seed, idx, a.thresh) ? a.keep_scale : 0.f;
        }
      }
      const float lse = stat[2 * tslot], dl = stat[2 * tslot + 1];
      const float pn = mu > 0 ? __expf(T1[e >> 2][e & 3] * inv_sq - lse) : 0.f;
      wds[e] = (bf16_t)(pn * (kwe * T2[e >> 2][e & 3] - (float)mu * dl) * inv_sq);
      wp[e] = (bf16_t)(pn * kwe);
    }
  }
  // A fragments: Q^T / dO^T rows of query slots 4g+q4 and 16+4g+q4 (every lane supplies one row address)
  const int ta = img_slot(4 * g + q4), tb = img_slot(16 + 4 * g + q4);
  const bool jvalid = j < r1;
  bf16_t* dkrow = dqkv + (tok0 + jc) * ld + D + head * HD;
  bf16_t* dvrow = dkrow + D;
  f32x4 okk[HD / 16], ovv[HD / 16];
#pragma unroll
  for (int dt = 0; dt < HD / 16; ++dt) {
    const int coff = (HD % 64 == 0 ? tr_col(dt, p4) : 16 * dt + 4 * p4) * 2;
    const s16x4 qa = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(ldsQ + ta * RS + coff));
    const s16x4 qb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(ldsQ + tb * RS + coff));
    const s16x4 ga = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(ldsG + ta * RS + coff));
    const s16x4 gb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(ldsG + tb * RS + coff));
    s16x8 qq, gg;
    qq[0] = qa[0]; qq[1] = qa[1]; qq[2] = qa[2]; qq[3] = qa[3]; qq[4] = qb[0]; qq[5] = qb[1]; qq[6] = qb[2]; qq[7] = qb[3];
    gg[0] = ga[0]; gg[1] = ga[1]; gg[2] = ga[2]; gg[3] = ga[3]; gg[4] = gb[0]; gg[5] = gb[1]; gg[6] = gb[2]; gg[7] = gb[3];
    okk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, qq), wds, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    ovv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, gg), wp, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
  }
  if (jvalid && !(probe_nostore && okk[0][0] != 12345.f)) {
    if constexpr (HD % 64 == 0) {
      store_rows16<HD>(dkrow, okk, g);
      store_rows16<HD>(dvrow, ovv, g);
    } else {
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) {
        bf16x4 kb4 = {(bf16_t)okk[dt][0], (bf16_t)okk[dt][1], (bf16_t)okk[dt][2], (bf16_t)okk[dt][3]};
        bf16x4 vb4 = {(bf16_t)ovv[dt][0], (bf16_t)ovv[dt][1], (bf16_t)ovv[dt][2], (bf16_t)ovv[dt][3]};
        *reinterpret_cast<bf16x4*>(dkrow + 16 * dt + 4 * g) = kb4;
        *reinterpret_cast<bf16x4*>(dvrow + 16 * dt + 4 * g) = vb4;
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// MFMA backward with SAVED statistics (round 3; bf16, hd = 64, W <= 11, L >= 2h + 2).  Same two owner passes as above,
// but the forward hands over lse per (batch, head, row) and its output O, so a query row outside the block never
// needs its softmax recomputed: P = mult * exp(s - lse_i) directly, and delta_i = sum_w P_w dP_w = dO_i . O_i is an
// 64-term dot product for the <= 4h + 1 halo / wrap rows of a block (owned rows keep the exact fp32 sum of pass 1).
//   * pass 1 runs over the OWNED 16-row tiles only (the table-free kernel above also ran the halo tiles: 15-17 query
//     tiles per (batch, head) at L = 197 against 13 here), rows [r0 - h, r1 + h) are staged once for all four images
//     (K~/V~ needed [r0 - 2h, r1 + 2h) before) and blocks are whole tiles: rb = 48 rows = three waves, one wave per
//     tile in both passes (the host picks the tile count per block, attn_entry);
//   * 128-byte LDS rows with the 16-byte chunks XOR-swizzled by the row index instead of 144-byte padded rows:
//     (rb + 2h + 2) K~/V~ rows and (rb + 4h + 1) Q / dO rows + statistics + the dump area = 31.5 KiB at W = 7, three
//     tiles -> five workgroups per CU.
// 272 MB of algorithmic traffic; 314 MB measured (halo rows, lse, the O rows of the halo); DESIGN.md has the split.
// ---------------------------------------------------------------------------------
// PairStager for a run-time thread count and swizzled 128-byte rows: load() issues every request, store() writes LDS.
//  * addresses = a workgroup-uniform base + a 32-bit byte offset per lane (the SGPR-base form of global_load: one
//    multiply-add per chunk instead of 64-bit pointer arithmetic -- staging was a third of the kernel's VALU work);
//  * store() is branch-free: chunks past the end (and rows `keep` rejects) go to a 1-KiB `dump` area, one slot per
//    lane.  A predicated store lets the compiler sink each load into its store's block: one global round trip per
//    chunk, one after the other (which is how the older kernels above stage, as their disassembly shows).
template <int NCH>
struct PairStager128 {
  uint4 ra[NCH], rb[NCH];
  // row_off_a / row_off_b: byte offset of LDS row s from base_a / base_b
  template <typename FA, typename FB>
  __device__ __forceinline__ void load(int nrows, int tid, int nthr, const char* base_a, const char* base_b, FA row_off_a,
                                       FB row_off_b) {
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int c = min(tid + nthr * it, nrows * 8 - 1);     // clamped: unconditional load
      const uint32_t ch = (uint32_t)(c & 7) * 16u;
      ra[it] = *reinterpret_cast<const uint4*>(base_a + (row_off_a(c >> 3) + ch));
      rb[it] = *reinterpret_cast<const uint4*>(base_b + (row_off_b(c >> 3) + ch));
    }
  }
  template <typename FP>
  __device__ __forceinline__ void store(char* lds_a, char* lds_b, char* dump, int nrows, int tid, int nthr, FP keep) const {
    char* mine = dump + (tid & 63) * 16;
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int c = tid + nthr * it;
      const bool ok = c < nrows * 8 && keep(c >> 3);
      const int o = (c >> 3) * 128 + ((((c & 7) ^ (c >> 3)) & 7) << 4);
      *reinterpret_cast<uint4*>(ok ? lds_a + o : mine) = ra[it];
      *reinterpret_cast<uint4*>(ok ? lds_b + o : mine) = rb[it];
    }
  }
};

__device__ __forceinline__ int swz128(int slot, int boff) {        // byte `boff` (< 128) of LDS row `slot`
  return slot * 128 + ((((boff >> 4) ^ slot) & 7) << 4) + (boff & 15);
}

// PLAIN: no mask and no dropout (compiled out).  NOMASK: no mask -- the interior fast paths apply; with dropout
// (NOMASK && !PLAIN: the reference's training setting) they draw the keep decisions from the window position in closed form.
template <bool PLAIN, int NIT, bool NOMASK = PLAIN>   // NIT: 16-byte chunks per thread and image pair (3: two or more tiles per block, W <= 7)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void mhla_bwd_lse_kernel(AttnArgs a) {
  constexpr int HD = 64;
  if (a.thresh) a.seed = favit_eff_seed(a.seed, a.epoch);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, qi = lane & 15, q4 = qi >> 2, p4 = qi & 3;
  const int head = blockIdx.y, b = blockIdx.z;
  const int L = a.L, W = a.W, h = W >> 1, D = a.H * HD;
  const int r0 = blockIdx.x * a.rb, r1 = min(L, r0 + a.rb);
  const long ld = 3L * D, tok0 = (long)b * L;
  const bf16_t* qkv = reinterpret_cast<const bf16_t*>(a.qkv);
  const bf16_t* dout = reinterpret_cast<const bf16_t*>(a.dout);
  const bf16_t* fo = reinterpret_cast<const bf16_t*>(a.o);
  bf16_t* dqkv = reinterpret_cast<bf16_t*>(a.out);
  const float* lse_g = a.lse + ((long)b * a.H + head) * L;

  // Images.  Main rows [qm_lo, qm_hi) of all four matrices share their slots; then K~/V~: key 0, key L-1 (the wrap
  // keys of pass 1); Q/dO: rows 0..h (they END-pad onto key L-1; only staged by the block that owns it) and rows
  // L-h..L-1 (they FRONT-pad onto key 0).
  const int qm_lo = max(0, r0 - h), qm_hi = min(L, r1 + h), nm = qm_hi - qm_lo;
  const bool has_hx = r1 >= L, has_tx = r0 == 0;
  const int tx_lo = L - h;
  const int cap_kv = a.rb + 2 * h + 2, cap_q = a.rb + 4 * h + 1;
  char* ldsK = smem;
  char* ldsV = ldsK + cap_kv * 128;
  char* ldsQ = ldsV + cap_kv * 128;
  char* ldsG = ldsQ + cap_q * 128;
  float* stat = reinterpret_cast<float*>(ldsG + cap_q * 128);          // [cap_q][2]: lse, delta
  const int nkv = nm + 2, nq = nm + 2 * h + 1;
  auto q_row = [&](int s) {                                            // -1: slot not used by this block
    if (s < nm) return qm_lo + s;
    const int e = s - nm;
    if (e <= h) return has_hx ? e : -1;
    return has_tx ? tx_lo + (e - h - 1) : -1;
  };
  {
    PairStager128<NIT> skv, sqg;
    const uint32_t ldb = (uint32_t)ld * 2u, ldg = (uint32_t)D * 2u;      // row pitches in bytes (qkv, dout)
    const char* base_q = reinterpret_cast<const char*>(qkv + tok0 * ld + head * HD);
    const char* base_g = reinterpret_cast<const char*>(dout + tok0 * (long)D + head * HD);
    auto kv_row = [&](int s) { return (uint32_t)(s < nm ? qm_lo + s : (s == nm ? 0 : L - 1)); };
    // (branch-free: the wrap slots of a block that does not use them load rows 0..h / L-h.. all the same)
    auto q_src = [&](int s) { const int e = s - nm; return (uint32_t)(s < nm ? qm_lo + s : (e <= h ? e : tx_lo + (e - h - 1))); };
    skv.load(nkv, tid, nthr, base_q + D * 2, base_q + D * 4, [&](int s) { return kv_row(s) * ldb; },
             [&](int s) { return kv_row(s) * ldb; });
    sqg.load(nq, tid, nthr, base_q, base_g, [&](int s) { return q_src(s) * ldb; }, [&](int s) { return q_src(s) * ldg; });
    // lse of every Q / dO slot: one slot per thread (nq <= 64 * waves), requested with the rows and branch-free as well
    const bool lse_ok = tid < nq && ((tid < nm) | (tid - nm <= h ? has_hx : has_tx));
    const float lse_v = lse_g[lse_ok ? q_src(tid) : 0u];
    char* dump = reinterpret_cast<char*>(stat + 2 * cap_q);
    *(lse_ok ? stat + 2 * tid : reinterpret_cast<float*>(dump) + tid) = lse_v;
    skv.store(ldsK, ldsV, dump, nkv, tid, nthr, [&](int) { return true; });
    sqg.store(ldsQ, ldsG, dump, nq, tid, nthr, [&](int s) { return (s < nm) | (s - nm <= h ? has_hx : has_tx); });
  }
  // delta of the rows this block does not own: their O rows are requested now and used after pass 1.  Candidate x:
  // h rows below r0, h rows from r1 up, the 2h + 1 wrap rows; 8 lanes (16 bytes each) per row.
  constexpr int NXP = NIT == 3 ? 1 : 3;      // passes of nthr / 8 rows (NIT = 3: >= 16 rows per pass, <= 13 candidates;
                                             // 64 threads at W = 11: 21 candidates in passes of 8)
  const int nx = 4 * h + 1;
  auto x_slot = [&](int x) {                                            // -1: nothing to do
    if (x < h) return x < r0 - qm_lo ? x : -1;
    if (x < 2 * h) { const int s = (r1 - qm_lo) + (x - h); return s < nm ? s : -1; }
    const int s = nm + (x - 2 * h);
    return q_row(s) >= 0 ? s : -1;
  };
  uint4 ro[NXP];
#pragma unroll
  for (int k = 0; k < NXP; ++k) {
    const int x = (tid >> 3) + k * (nthr >> 3);
    const int s = x < nx ? x_slot(x) : -1;
    ro[k] = *reinterpret_cast<const uint4*>(fo + (tok0 + (s >= 0 ? q_row(s) : r0)) * (long)D + head * HD + (tid & 7) * 8);
  }
  __syncthreads();
#ifdef FAVIT_PROBE
  if (a.dbg & 2) return;                                 // probe: staging only
  const bool probe_nostore = (a.dbg & 1) != 0;
#else
  constexpr bool probe_nostore = false;
#endif

  const float inv_sq = 0.125f;                           // 1 / sqrt(64)
  // ---------------- pass 1: the owned query tiles ----------------
  if (r0 + 16 * wave < r1) {
    const int t0 = r0 + 16 * wave;
    const int i = t0 + qi;
    SlotInfo si;
    si.init(min(i, L - 1), L, W, h);
    auto kslot = [&](int slot) {                                        // image row of a key slot of this tile
      if (slot == 31) return nm + 1;
      if (slot >= 16 + 2 * h) return nm;                                // slot 30 (key 0) and the unused slots
      return min(max(t0 - h + slot, qm_lo), qm_hi - 1) - qm_lo;
    };
    const int qslot = min(t0 - qm_lo + qi, nm - 1);
    const int krow0 = kslot(qi), krow1 = kslot(16 + qi);
    f32x4 S[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    f32x4 dP[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) {
      const int off = (32 * ks + 8 * g) * 2;
      const bf16x8 qf = *reinterpret_cast<const bf16x8*>(ldsQ + swz128(qslot, off));
      const bf16x8 gf = *reinterpret_cast<const bf16x8*>(ldsG + swz128(qslot, off));
      const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(ldsK + swz128(krow0, off));
      const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(ldsK + swz128(krow1, off));
      const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(ldsV + swz128(krow0, off));
      const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(ldsV + swz128(krow1, off));
      S[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf, S[0], 0, 0, 0);
      S[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf, S[1], 0, 0, 0);
      dP[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v0, gf, dP[0], 0, 0, 0);
      dP[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1, gf, dP[1], 0, 0, 0);
    }
    const float lse_i = stat[2 * qslot];
    float pn[8], mult[8], dpn[8];
    // interior tile (wave-uniform): every row has its full window of W distinct keys, no wrap slots, no padding
    const bool interior = NOMASK && t0 >= h && t0 + 15 + h <= L - 1;
    if (interior) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int slot = 16 * (e >> 2) + 4 * g + (e & 3);
        const int d = slot - h - qi;                                    // key - query
        const bool in = (slot < 16 + 2 * h) && (d >= -h) && (d <= h);
        mult[e] = in ? 1.f : 0.f;
        pn[e] = in ? __expf(S[e >> 2][e & 3] * inv_sq - lse_i) : 0.f;
        float kwe = mult[e];
        if (!PLAIN) {                                                   // dropout: window position of key j is d + h
          const uint64_t idx = (((uint64_t)b * a.H + head) * L + i) * W + (uint64_t)(d + h);
          kwe = (in && favit_keep(a.seed, idx, a.thresh)) ? a.keep_scale : 0.f;
        }
        dpn[e] = kwe * dP[e >> 2][e & 3];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int kt = e >> 2, r = e & 3;
        const int slot = 16 * kt + 4 * g + r;
        int j, mu, w0;
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
        if (!PLAIN && mu > 0 && a.mask && a.mask[((long)b * L + si.i) * L + j] == 0) mu = 0;
        float kwe = (float)mu;
        if (!PLAIN && a.thresh && mu > 0) {
          kwe = 0.f;
          for (int c = 0; c < mu; ++c) {
            const uint64_t idx = (((uint64_t)b * a.H + head) * L + si.i) * W + (w0 + c);
            kwe += favit_keep(a.seed, idx, a.thresh) ? a.keep_scale : 0.f;
          }
        }
        mult[e] = (float)mu;
        pn[e] = mu > 0 ? __expf(S[kt][r] * inv_sq - lse_i) : 0.f;      // single-copy probability
        dpn[e] = kwe * dP[kt][r];
      }
    }
    float dot = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) dot = fmaf(pn[e], dpn[e], dot);
    dot = quad16_sum(dot);
    const bool own = i < r1;
    if (g == 0 && own) stat[2 * qslot + 1] = dot;
    bf16x8 dsf;
#pragma unroll
    for (int e = 0; e < 8; ++e) dsf[e] = (bf16_t)((pn[e] * dpn[e] - mult[e] * pn[e] * dot) * inv_sq);
    const int vrow0 = kslot(4 * g + q4), vrow1 = kslot(16 + 4 * g + q4);
    bf16_t* dqrow = dqkv + (tok0 + si.i) * ld + head * HD;
    f32x4 oq[HD / 16];
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) {
      const int coff = tr_col(dt, p4) * 2;
      const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(ldsK + swz128(vrow0, coff)));
      const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(ldsK + swz128(vrow1, coff)));
      s16x8 kk;
      kk[0] = lo4[0]; kk[1] = lo4[1]; kk[2] = lo4[2]; kk[3] = lo4[3];
      kk[4] = hi4[0]; kk[5] = hi4[1]; kk[6] = hi4[2]; kk[7] = hi4[3];
      oq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kk), dsf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    if (own && !(probe_nostore && oq[0][0] != 12345.f)) store_rows16<HD>(dqrow, oq, g);
  }
  // delta = dO . O of the rows owned by other blocks
#pragma unroll
  for (int k = 0; k < NXP; ++k) {
    const int x = (tid >> 3) + k * (nthr >> 3);
    const int s = x < nx ? x_slot(x) : -1;
    float d = 0.f;
    if (s >= 0) {
      const uint4 gd = *reinterpret_cast<const uint4*>(ldsG + swz128(s, (tid & 7) * 16));
      const uint32_t ow[4] = {ro[k].x, ro[k].y, ro[k].z, ro[k].w}, gw[4] = {gd.x, gd.y, gd.z, gd.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        d = fmaf(__uint_as_float(ow[c] << 16), __uint_as_float(gw[c] << 16), d);
        d = fmaf(__uint_as_float(ow[c] & 0xffff0000u), __uint_as_float(gw[c] & 0xffff0000u), d);
      }
    }
    d = dpp_sum8(d);
    if (s >= 0 && (tid & 7) == 0) stat[2 * s + 1] = d;
  }
  __syncthreads();

  // ---------------- pass 2: the owned key tiles ----------------
  const int k0 = r0 + 16 * wave;
  if (k0 >= r1) return;
  const int j = k0 + qi;                                  // this lane's key (B-operand column)
  const int jc = min(j, L - 1);
  const bool has0 = (k0 == 0), hasL = (L - 1 >= k0 && L - 1 < k0 + 16);
  const int nband = 16 + 2 * h;
  const int nhx = hasL ? h + 1 : 0;                       // rows 0..h END-pad onto key L-1
  auto main_slot = [&](int i) { return min(max(i, qm_lo), qm_hi - 1) - qm_lo; };
  // query slot -> (query row, image / statistics slot, kind): 0 band, 1 END-pad row, 2 FRONT-pad row, 3 unused
  auto qs_row = [&](int qs, int& row, int& tslot, int& kind) {
    if (qs < nband) {
      kind = 0;
      row = k0 - h + qs;
      tslot = main_slot(row);
    } else if (qs < nband + nhx) {
      kind = 1;
      row = qs - nband;
      tslot = nm + row;
    } else {
      const int k = qs - nband - nhx;
      kind = 2;
      row = tx_lo + k;
      if (!has0 || k >= h) { kind = 3; row = 0; tslot = main_slot(k0); return; }
      tslot = nm + h + 1 + k;
    }
  };
  const bool interior_keys = NOMASK && !has0 && !hasL;      // wave-uniform: band query slots only, multiplicity 0 / 1
  auto img_slot = [&](int qs) {                             // Q / dO image row (and statistics slot) of a query slot
    if (interior_keys) return main_slot(k0 - h + qs);       // (clamped: slots past the band carry multiplicity 0)
    int row, tslot, kind;
    qs_row(qs, row, tslot, kind);
    return tslot;
  };
  // S^T[qs][key] = Q_qs . K_key,  dP^T[qs][key] = dO_qs . V_key   (A = the query slot rows, B = this lane's key row)
  f32x4 T1[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  f32x4 T2[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  {
    const int ts0 = img_slot(qi), ts1 = img_slot(16 + qi);
    const int kj = main_slot(jc);
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) {
      const int off = (32 * ks + 8 * g) * 2;
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ldsK + swz128(kj, off));
      const bf16x8 vf = *reinterpret_cast<const bf16x8*>(ldsV + swz128(kj, off));
      const bf16x8 qa = *reinterpret_cast<const bf16x8*>(ldsQ + swz128(ts0, off));
      const bf16x8 qb = *reinterpret_cast<const bf16x8*>(ldsQ + swz128(ts1, off));
      const bf16x8 ga = *reinterpret_cast<const bf16x8*>(ldsG + swz128(ts0, off));
      const bf16x8 gb = *reinterpret_cast<const bf16x8*>(ldsG + swz128(ts1, off));
      T1[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf, T1[0], 0, 0, 0);
      T1[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qb, kf, T1[1], 0, 0, 0);
      T2[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga, vf, T2[0], 0, 0, 0);
      T2[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gb, vf, T2[1], 0, 0, 0);
    }
  }
  // the lane holds query slots 16 st + 4g + r (e = 4 st + r) of its key: exactly the k-slice of the next MFMAs
  bf16x8 wds, wp;
  if (interior_keys) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int qs = 16 * (e >> 2) + 4 * g + (e & 3);
      const int row = k0 - h + qs, d = qs - h - qi;                    // query row, query - key
      const int tslot = min(max(row, qm_lo), qm_hi - 1) - qm_lo;
      const bool in = (qs < nband) && (row >= 0) && (row < L) && (j < L) && (d >= -h) && (d <= h);
      const float lse = stat[2 * tslot], dl = stat[2 * tslot + 1];
      const float pn = in ? __expf(T1[e >> 2][e & 3] * inv_sq - lse) : 0.f;
      float kwe = 1.f;
      if (!PLAIN) {                                         // dropout: this key's position in the query row's window
        const int lo = max(0, row - h), pad = W - (min(L, row + h + 1) - lo);
        const int w0 = (lo == 0 || pad == 0) ? (j - lo) : pad + (j - lo);
        const uint64_t idx = (((uint64_t)b * a.H + head) * L + row) * W + (uint64_t)w0;
        kwe = (in && favit_keep(a.seed, idx, a.thresh)) ? a.keep_scale : 0.f;
      }
      wds[e] = (bf16_t)(pn * (kwe * T2[e >> 2][e & 3] - dl) * inv_sq);
      wp[e] = (bf16_t)(pn * kwe);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int row, tslot, kind;
      qs_row(16 * (e >> 2) + 4 * g + (e & 3), row, tslot, kind);
      int mu = 0, w0 = 0;
      if (j < L && row >= 0 && row < L && kind != 3) {
        const int lo = max(0, row - h), hi = min(L, row + h + 1), n = hi - lo, pad = W - n;
        if (kind == 0) {
          if (j >= lo && j < hi) { mu = 1; w0 = (lo == 0 || pad == 0) ? (j - lo) : pad + (j - lo); }
        } else if (kind == 1) {
          if (j == L - 1 && lo == 0) { mu = pad; w0 = n; }              // END padding of the rows whose window starts at 0
        } else {
          if (j == 0 && lo > 0) { mu = pad; w0 = 0; }                   // FRONT padding
        }
        if (!PLAIN && mu > 0 && a.mask && a.mask[((long)b * L + row) * L + j] == 0) mu = 0;
      }
      float kwe = (float)mu;
      if (!PLAIN && a.thresh && mu > 0) {
        kwe = 0.f;
        for (int c = 0; c < mu; ++c) {
          const uint64_t idx = (((uint64_t)b * a.H + head) * L + row) * W + (w0 + c);
          kwe += favit_keep(a.seed, idx, a.thresh) ? a.keep_scale : 0.f;
        }
      }
      const float lse = stat[2 * tslot], dl = stat[2 * tslot + 1];
      const float pn = mu > 0 ? __expf(T1[e >> 2][e & 3] * inv_sq - lse) : 0.f;
      wds[e] = (bf16_t)(pn * (kwe * T2[e >> 2][e & 3] - (float)mu * dl) * inv_sq);
      wp[e] = (bf16_t)(pn * kwe);
    }
  }
  // A fragments: Q^T / dO^T rows of query slots 4g+q4 and 16+4g+q4 (every lane supplies one row address)
  const int ta = img_slot(4 * g + q4), tb = img_slot(16 + 4 * g + q4);
  const bool jvalid = j < r1;
  bf16_t* dkrow = dqkv + (tok0 + jc) * ld + D + head * HD;
  bf16_t* dvrow = dkrow + D;
  f32x4 okk[HD / 16], ovv[HD / 16];
#pragma unroll
  for (int dt = 0; dt < HD / 16; ++dt) {
    const int coff = tr_col(dt, p4) * 2;
    const s16x4 qa = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(ldsQ + swz128(ta, coff)));
    const s16x4 qb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(ldsQ + swz128(tb, coff)));
    const s16x4 ga = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(ldsG + swz128(ta, coff)));
    const s16x4 gb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(ldsG + swz128(tb, coff)));
    s16x8 qq, gg;
    qq[0] = qa[0]; qq[1] = qa[1]; qq[2] = qa[2]; qq[3] = qa[3]; qq[4] = qb[0]; qq[5] = qb[1]; qq[6] = qb[2]; qq[7] = qb[3];
    gg[0] = ga[0]; gg[1] = ga[1]; gg[2] = ga[2]; gg[3] = ga[3]; gg[4] = gb[0]; gg[5] = gb[1]; gg[6] = gb[2]; gg[7] = gb[3];
    okk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, qq), wds, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    ovv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, gg), wp, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
  }
  if (jvalid && !(probe_nostore && okk[0][0] != 12345.f)) {
    store_rows16<HD>(dkrow, okk, g);
    store_rows16<HD>(dvrow, ovv, g);
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
__device__ __forceinline__ void fold_w_body(const float* __restrict__ win, const float* __restrict__ bin,
                                            const float* __restrict__ wl, const float* __restrict__ bl,
                                            T* __restrict__ wout, float* __restrict__ wout_f32,
                                            float* __restrict__ bout, int D, int H, int accumulate, int bx, int by,
                                            float* smem) {
  float* sAt = smem;                       // [HD * HD]
  float* sBt = smem + HD * HD;             // [HD * FOLD_TC]
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
  if (by >= 2 * H) {                      // q part: copy / cast, HD rows per block
    const int c = bx * FOLD_TC + tc;
    if (c > D) return;
    const int rb = (by - 2 * H) * HD;
#pragma unroll 4
    for (int r = rb + tg; r < rb + HD && r < D; r += 4) put(r, c, (c < D) ? win[(long)r * D + c] : bin[r]);
    return;
  }
  const long base = (long)D + (long)by * HD;   // first row of this (s,h) block
  // At[k][r]: forward needs Wl[r][k] (transpose while staging), backward Wl[k][r] (as stored)
  for (int i = threadIdx.x; i < HD * HD; i += 256) {
    const int k = i / HD, r = i % HD;
    sAt[i] = BWD ? wl[k * HD + r] : wl[r * HD + k];
  }
  for (int k = tg; k < HD; k += 4) {
    const int c = bx * FOLD_TC + tc;
    sBt[k * FOLD_TC + tc] = (c < D) ? win[(base + k) * D + c] : (c == D ? bin[base + k] : 0.f);
  }
  __syncthreads();
  float acc[HD / 16][4];
  fold_tile_gemm<HD>(sAt, sBt, acc);
  constexpr int TR = HD / 16;
  const int r0 = TR * (threadIdx.x >> 4), c0 = bx * FOLD_TC + 4 * (threadIdx.x & 15);
#pragma unroll
  for (int i = 0; i < TR; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = acc[i][j];
      if (!BWD && c0 + j == D) v += bl[r0 + i];
      put(base + r0 + i, c0 + j, v);
    }
}

template <typename T, int HD, bool BWD>
__global__ __launch_bounds__(256) void fold_w_kernel(const float* __restrict__ win, const float* __restrict__ bin,
                                                     const float* __restrict__ wl, const float* __restrict__ bl,
                                                     T* __restrict__ wout, float* __restrict__ wout_f32,
                                                     float* __restrict__ bout, int D, int H, int accumulate) {
  __shared__ __attribute__((aligned(16))) float smem[HD * HD + HD * FOLD_TC];
  fold_w_body<T, HD, BWD>(win, bin, wl, bl, wout, wout_f32, bout, D, H, accumulate, blockIdx.x, blockIdx.y, smem);
}

// dWl[i][j] += sum_c dWeff_z[i][c] Wqkv_z[j][c] + dbeff_z[i] bqkv_z[j];  dbl[i] += dbeff_z[i].
// grid (2H, FOLD_L_SPLIT): a workgroup takes one (s,h) block z and walks HALF of the D+1 columns in tiles of 64 with a
// register-blocked [HD x HD] accumulator, then adds its partial with fp32 atomics -- 2 * 2H partials per layer land on
// the same HD x HD words.  (One tile per workgroup, as before round 3, made that 2H * (D/64 + 1): 312-way contention
// at ViT-Base, 349 us for the twelve layers of a cfg4 step.)
constexpr int FOLD_L_SPLIT = 4;
template <int HD>
__device__ __forceinline__ void fold_bwd_l_body(const float* __restrict__ dweff, const float* __restrict__ dbeff,
                                                const float* __restrict__ wqkv, const float* __restrict__ bqkv,
                                                float* __restrict__ dwl, float* __restrict__ dbl, int D, int bx, int by,
                                                float* smem) {
  constexpr int TR = HD / 16;                          // thread block TR x TR of the [HD x HD] output
  constexpr int LDT = HD + 1;                          // padded: transposed staging is conflict-free
  float* sa = smem;                                    // [64][LDT]  dWeff tile, transposed
  float* sb = smem + 64 * LDT;                         // [64][LDT]  Wqkv tile, transposed
  const int tc = threadIdx.x & 63, tg = threadIdx.x >> 6;
  const long base = (long)D + (long)bx * HD;
  const int i0 = TR * (threadIdx.x >> 4), j0 = TR * (threadIdx.x & 15);
  const int ntile = (D + 64) / 64;                     // tiles of 64 columns over D + 1 (column D = the bias)
  const int per = (ntile + FOLD_L_SPLIT - 1) / FOLD_L_SPLIT;
  const int t_lo = by * per, t_hi = min(ntile, t_lo + per);
  float acc[TR][TR];
#pragma unroll
  for (int i = 0; i < TR; ++i)
#pragma unroll
    for (int j = 0; j < TR; ++j) acc[i][j] = 0.f;
  for (int t = t_lo; t < t_hi; ++t) {
    const int c = t * 64 + tc;
    float va[HD / 4], vb[HD / 4];
#pragma unroll
    for (int q = 0; q < HD / 4; ++q) {
      const int r = tg + 4 * q;
      va[q] = (c < D) ? dweff[(base + r) * D + c] : (c == D ? dbeff[base + r] : 0.f);
      vb[q] = (c < D) ? wqkv[(base + r) * D + c] : (c == D ? bqkv[base + r] : 0.f);
    }
    if (t > t_lo) __syncthreads();                     // the previous tile has been consumed
#pragma unroll
    for (int q = 0; q < HD / 4; ++q) {
      sa[tc * LDT + tg + 4 * q] = va[q];
      sb[tc * LDT + tg + 4 * q] = vb[q];
    }
    __syncthreads();
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
  }
  if (t_lo < t_hi) {
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
      for (int j = 0; j < TR; ++j) atomicAdd(dwl + (long)(i0 + i) * HD + j0 + j, acc[i][j]);
  }
  if (by == 0 && threadIdx.x < HD) atomicAdd(dbl + threadIdx.x, dbeff[base + threadIdx.x]);
}

// The whole backward of the fold in ONE launch (these kernels are a few workgroups each and purely
// latency-bound): blocks [0, nw) map the folded weight gradient back onto qkv.{weight,bias}
// (Wl^T . dWeff), the remaining blocks accumulate the latent_proj gradient.
template <int HD>
__global__ __launch_bounds__(256) void fold_bwd_all_kernel(const float* __restrict__ dweff,
                                                           const float* __restrict__ dbeff,
                                                           const float* __restrict__ wqkv,
                                                           const float* __restrict__ bqkv,
                                                           const float* __restrict__ wl, float* __restrict__ dwqkv,
                                                           float* __restrict__ dbqkv, float* __restrict__ dwl,
                                                           float* __restrict__ dbl, int D, int H, int accumulate,
                                                           int gwx, int nw, int glx) {
  constexpr int SW = HD * HD + HD * FOLD_TC, SL = 2 * 64 * (HD + 1);
  __shared__ __attribute__((aligned(16))) float smem[SW > SL ? SW : SL];
  const int b = blockIdx.x;
  if (b < nw) {
    fold_w_body<float, HD, true>(dweff, dbeff, wl, nullptr, dwqkv, nullptr, dbqkv, D, H, accumulate, b % gwx, b / gwx, smem);
  } else {
    const int r = b - nw;
    fold_bwd_l_body<HD>(dweff, dbeff, wqkv, bqkv, dwl, dbl, D, r % glx, r / glx, smem);
  }
}

// The backward folds of up to FOLD_BWD_MAX layers in ONE launch (blockIdx.y = layer): each is ~100 small
// workgroups and latency-bound (22.9 us x 12 per cfg2 step as separate launches), the gradients they produce
// (qkv.{weight,bias}, latent_proj.{weight,bias}) feed nothing in the backward chain.  Accumulating form only.
constexpr int FOLD_BWD_MAX = 16;
struct FoldBwdBatch {
  const float* dweff[FOLD_BWD_MAX];
  const float* dbeff[FOLD_BWD_MAX];
  const float* wqkv[FOLD_BWD_MAX];
  const float* bqkv[FOLD_BWD_MAX];
  const float* wl[FOLD_BWD_MAX];
  float* dwqkv[FOLD_BWD_MAX];
  float* dbqkv[FOLD_BWD_MAX];
  float* dwl[FOLD_BWD_MAX];
  float* dbl[FOLD_BWD_MAX];
};
template <int HD>
__global__ __launch_bounds__(256) void fold_bwd_multi_kernel(FoldBwdBatch fb, int D, int H, int gwx, int nw, int glx) {
  constexpr int SW = HD * HD + HD * FOLD_TC, SL = 2 * 64 * (HD + 1);
  __shared__ __attribute__((aligned(16))) float smem[SW > SL ? SW : SL];
  const int b = blockIdx.x, z = blockIdx.y;
  if (b < nw) {
    fold_w_body<float, HD, true>(fb.dweff[z], fb.dbeff[z], fb.wl[z], nullptr, fb.dwqkv[z], nullptr, fb.dbqkv[z], D, H, 1,
                                 b % gwx, b / gwx, smem);
  } else {
    const int r = b - nw;
    fold_bwd_l_body<HD>(fb.dweff[z], fb.dbeff[z], fb.wqkv[z], fb.bqkv[z], fb.dwl[z], fb.dbl[z], D, r % glx, r / glx, smem);
  }
}

// Forward folds of up to FOLD_MAX independent blocks (the layers of one encoder) in ONE launch:
// the fold only depends on parameters, so all of them can run before the first block.
constexpr int FOLD_MAX = 32;
struct FoldBatch {
  const float* wqkv[FOLD_MAX];
  const float* bqkv[FOLD_MAX];
  const float* wl[FOLD_MAX];
  const float* bl[FOLD_MAX];
  void* weff[FOLD_MAX];
  float* beff[FOLD_MAX];
};

template <typename T, int HD>
__global__ __launch_bounds__(256) void fold_w_multi_kernel(FoldBatch fb, int D, int H) {
  __shared__ __attribute__((aligned(16))) float smem[HD * HD + HD * FOLD_TC];
  const int z = blockIdx.z;
  fold_w_body<T, HD, false>(fb.wqkv[z], fb.bqkv[z], fb.wl[z], fb.bl[z], reinterpret_cast<T*>(fb.weff[z]), nullptr,
                            fb.beff[z], D, H, 0, blockIdx.x, blockIdx.y, smem);
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
    const size_t lds = (size_t)2 * krows * rs + (size_t)2 * a.qcap * rs + (size_t)2 * a.qcap * WMAX * 4;
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

// the saved-statistics backward (mhla_bwd_lse_kernel) and the forward that feeds it
bool lse_path_ok(int L, int hd, int W, int dtype) {
  const int h = W / 2;
  return dtype == FAVIT_BF16 && hd == 64 && W > 0 && (W & 1) && (W <= 7 || (W <= 11 && L > 16)) && L >= 2 * h + 2 &&
         getenv("FAVIT_MHLA_VALU") == nullptr && getenv("FAVIT_MHLA_NO_LSE") == nullptr;
}

int attn_entry(bool bwd, const void* qkv, const void* dout, void* out, const uint8_t* mask, int B, int L, int H, int hd,
               int W, int dtype, float p, uint64_t seed, void* stream, const void* fwd_o = nullptr,
               const float* lse_in = nullptr, float* lse_out = nullptr) {
  if (!qkv || !out || (bwd && !dout) || B <= 0 || L <= 0 || H <= 0 || hd <= 0) return FAVIT_ERR_INVALID;
  if (W <= 0 || (W & 1) == 0) return FAVIT_ERR_INVALID;      // even windows crash the reference (mhla.py:83)
  if (W > 15) return FAVIT_ERR_UNSUPPORTED;
  if (p < 0.f || p >= 1.f) return FAVIT_ERR_INVALID;
  AttnArgs a;
  a.qkv = qkv; a.dout = dout; a.out = out; a.mask = mask;
  a.B = B; a.L = L; a.H = H; a.hd = hd; a.W = W;
  a.lse_out = lse_out; a.lse = lse_in; a.o = fwd_o;
  if (lse_out && (bwd || !lse_path_ok(L, hd, W, dtype))) return FAVIT_ERR_UNSUPPORTED;
  {
    // backward row block: as many key rows as the fixed LDS regions allow (64 query / dO rows incl.
    // the wrap rows, 80 K~/V~ rows incl. halo and edges), then balanced over the blocks of L
    const int h = W / 2;
    int rb_max = 64 - 3 * h - 1;
    if (80 - 8 * h - 2 < rb_max) rb_max = 80 - 8 * h - 2;
    if (rb_max < 1) return FAVIT_ERR_UNSUPPORTED;
    // One block if the sequence fits.  Otherwise blocks whose band rows (rb + 2h) fill exactly one 32-row
    // pass of phase 1, balanced over L: less LDS per workgroup (6 resident instead of 4 at hd = 64, W = 7)
    // outweighs the extra halo staging (measured at L = 197: rb 25 -> 127 us, rb 50 -> 133 us, rb 33 -> 158 us).
    int target = rb_max;
    if (L > rb_max && 32 - 2 * h >= 8 && 32 - 2 * h < rb_max) target = 32 - 2 * h;
    const int nblk = (L + target - 1) / target;
    a.rb = (L + nblk - 1) / nblk;
    const char* e = getenv("FAVIT_MHLA_RB");
    if (e && atoi(e) > 0 && atoi(e) <= rb_max) a.rb = atoi(e);
    a.qcap = a.rb + 3 * h + 1;                 // band rows + the wrap rows a first / last block adds
    if (a.qcap > 64) a.qcap = 64;
  }
#ifdef FAVIT_PROBE
  { const char* e = getenv("FAVIT_MHLA_DBG"); a.dbg = e ? atoi(e) : 0; }
#endif
  a.inv_sqrt_hd = 0.f;
  a.thresh = dropout_threshold(p);
  a.keep_scale = 1.0f / (1.0f - p);
  a.seed = seed;
  a.epoch = favit_dropout_epoch_ptr_();
  hipStream_t st = as_stream(stream);
  if (!bwd && dtype == FAVIT_BF16 && (hd == 32 || hd == 64 || hd == 128) && getenv("FAVIT_MHLA_VALU") == nullptr) {
    const int h = W / 2;
    const size_t lds = (size_t)2 * (64 + 2 * h + 2) * (hd * 2 + 16) + 1024;      // K~ / V~ images + the staging dump area
    dim3 grid((L + 63) / 64, H, B);
    const bool plain = (mask == nullptr) && (a.thresh == 0);
    if (hd == 32) { if (plain) hipLaunchKernelGGL((mhla_fwd_mfma_kernel<32, true>), grid, dim3(256), lds, st, a); else hipLaunchKernelGGL((mhla_fwd_mfma_kernel<32, false>), grid, dim3(256), lds, st, a); }
    else if (hd == 64) { if (plain) hipLaunchKernelGGL((mhla_fwd_mfma_kernel<64, true>), grid, dim3(256), lds, st, a); else hipLaunchKernelGGL((mhla_fwd_mfma_kernel<64, false>), grid, dim3(256), lds, st, a); }
    else { if (plain) hipLaunchKernelGGL((mhla_fwd_mfma_kernel<128, true>), grid, dim3(256), lds, st, a); else hipLaunchKernelGGL((mhla_fwd_mfma_kernel<128, false>), grid, dim3(256), lds, st, a); }
    FAVIT_CHECK_LAUNCH();
    return FAVIT_OK;
  }
  // bf16 backward with the forward's statistics (mhla_bwd_lse_kernel): whole-tile row blocks, one wave per 16 rows
  if (bwd && lse_in && fwd_o) {
    if (!lse_path_ok(L, hd, W, dtype)) return FAVIT_ERR_UNSUPPORTED;
    const int h = W / 2;
    // three tiles per block: 31.5 KiB of LDS = five workgroups per CU (measured at L = 197: 65.9 us against 71.2 with
    // four tiles / 39.8 KiB / four per CU and 68.3 with two)
    int nw = (L + 15) / 16 < 3 ? (L + 15) / 16 : 3;
    { const char* e = getenv("FAVIT_MHLA_LSE_WAVES"); if (e && atoi(e) >= 1 && atoi(e) <= 4 && atoi(e) <= (L + 15) / 16) nw = atoi(e); }
    AttnArgs a3 = a;
    a3.rb = 16 * nw;
    const int cap_kv = a3.rb + 2 * h + 2, cap_q = a3.rb + 4 * h + 1;
    const size_t lds = (size_t)(2 * cap_kv + 2 * cap_q) * 128 + (size_t)cap_q * 8 + 1024;     // images, statistics, dump
    dim3 grid((L + a3.rb - 1) / a3.rb, H, B);
    const bool plain = mask == nullptr && a3.thresh == 0;
    const bool three = cap_kv * 8 <= 3 * 64 * nw && cap_q * 8 <= 3 * 64 * nw;         // chunks per thread and image pair
    if (three) {
      if (plain) hipLaunchKernelGGL((mhla_bwd_lse_kernel<true, 3>), grid, dim3(64 * nw), lds, st, a3);
      else if (mask == nullptr) hipLaunchKernelGGL((mhla_bwd_lse_kernel<false, 3, true>), grid, dim3(64 * nw), lds, st, a3);
      else hipLaunchKernelGGL((mhla_bwd_lse_kernel<false, 3>), grid, dim3(64 * nw), lds, st, a3);
    } else {
      if (plain) hipLaunchKernelGGL((mhla_bwd_lse_kernel<true, 5>), grid, dim3(64 * nw), lds, st, a3);
      else if (mask == nullptr) hipLaunchKernelGGL((mhla_bwd_lse_kernel<false, 5, true>), grid, dim3(64 * nw), lds, st, a3);
      else hipLaunchKernelGGL((mhla_bwd_lse_kernel<false, 5>), grid, dim3(64 * nw), lds, st, a3);
    }
    FAVIT_CHECK_LAUNCH();
    return FAVIT_OK;
  }
  // bf16 backward on MFMA, "two owner passes" formulation (mhla_bwd_mfma2_kernel): balanced row blocks of <= 64,
  // no dS / P tables.  FAVIT_MHLA_BWD_TABLES selects the older table formulation (72 KiB of LDS), FAVIT_MHLA_VALU
  // the 8-lanes-per-row kernel; all three pass the same tests.
  if (bwd && dtype == FAVIT_BF16 && (hd == 32 || hd == 64 || hd == 128) && (W <= 7 || (W <= 11 && L > 16)) &&
      getenv("FAVIT_MHLA_VALU") == nullptr && getenv("FAVIT_MHLA_BWD_MFMA") == nullptr && getenv("FAVIT_MHLA_BWD_TABLES") == nullptr) {
    const int h = W / 2;
    const int rs = hd * 2 + 16;
    // Row blocks: balanced over L, at most 64 rows, and -- if a finer split gets there -- small enough for FOUR
    // workgroups per CU (<= 40 KiB of LDS): measured at L = 197, hd = 64: 4 blocks of 50 rows (45 KiB, 3 / CU) 100 us,
    // 5 blocks of 40 (38 KiB, 4 / CU) 84 us, 6 blocks 92 us (more halo rows per owned row).
    auto geometry = [&](int nb, int& rbv, int& kr, int& nt) {
      rbv = (L + nb - 1) / nb;
      const int nbe = (L + rbv - 1) / rbv;
      kr = (rbv + 4 * h) + 2 * (2 * h + 1);
      // query tiles: the main rows + the wrap tiles of keys 0 / L-1 (both only when one block holds the whole sequence)
      nt = (rbv + 2 * h + 15) / 16 + (nbe >= 2 ? 1 : 2);
      return (size_t)2 * kr * rs + (size_t)2 * 16 * nt * rs + (size_t)16 * nt * 8;
    };
    const int nb0 = (L + 63) / 64;
    int nblk = nb0, rbv, krows, ntl;
    for (int nb = nb0; nb <= 2 * nb0 && nb <= L; ++nb) {
      if (geometry(nb, rbv, krows, ntl) <= 40 * 1024) { nblk = nb; break; }
    }
    { const char* e = getenv("FAVIT_MHLA_BLOCKS"); if (e && atoi(e) > 0 && (L + atoi(e) - 1) / atoi(e) <= 64) nblk = atoi(e); }
    AttnArgs a2 = a;
    (void)geometry(nblk, rbv, krows, ntl);
    a2.rb = rbv;
    nblk = (L + a2.rb - 1) / a2.rb;
#ifdef FAVIT_PROBE
    // FAVIT_MHLA_LDS_PAD=<bytes>: claim more LDS than needed (occupancy experiment: fewer workgroups per CU)
    const size_t lds_pad = getenv("FAVIT_MHLA_LDS_PAD") ? (size_t)atoi(getenv("FAVIT_MHLA_LDS_PAD")) : 0;
#else
    constexpr size_t lds_pad = 0;
#endif
    const size_t lds = (size_t)2 * krows * rs + (size_t)2 * 16 * ntl * rs + (size_t)16 * ntl * 8 + lds_pad;
    if (lds <= 160 * 1024 && krows <= 108 && 16 * ntl <= 112) {
      dim3 grid(nblk, H, B);
      const bool plain = (mask == nullptr) && (a2.thresh == 0);
#define FAVIT_MFMA2(HDV, PL)                                                                                  \
  do {                                                                                                        \
    if (lds > 65536) favit_ensure_dyn_lds(reinterpret_cast<const void*>(mhla_bwd_mfma2_kernel<HDV, PL>), (int)lds); \
    hipLaunchKernelGGL((mhla_bwd_mfma2_kernel<HDV, PL>), grid, dim3(256), lds, st, a2);                       \
  } while (0)
      if (hd == 32) { if (plain) FAVIT_MFMA2(32, true); else FAVIT_MFMA2(32, false); }
      else if (hd == 64) { if (plain) FAVIT_MFMA2(64, true); else FAVIT_MFMA2(64, false); }
      else { if (plain) FAVIT_MFMA2(128, true); else FAVIT_MFMA2(128, false); }
#undef FAVIT_MFMA2
      FAVIT_CHECK_LAUNCH();
      return FAVIT_OK;
    }
  }
  // The table formulation of the MFMA backward (opt-in: FAVIT_MHLA_BWD_MFMA=1 or FAVIT_MHLA_BWD_TABLES=1): measured
  // 189 us at the bench shape against 131 us for the 8-lanes-per-row kernel (72 KiB of LDS allow only two
  // workgroups per CU and every phase is latency-bound).
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

extern "C" int favit_mhla_attn_lse_supported(int32_t L, int32_t hd, int32_t W, int dtype) {
  return lse_path_ok(L, hd, W, dtype) ? 1 : 0;
}

extern "C" int favit_mhla_attn_fwd_lse(const void* qkv, void* out, float* lse, const uint8_t* mask, int32_t B, int32_t L,
                                       int32_t H, int32_t hd, int32_t W, int dtype, float dropout_p, uint64_t seed,
                                       void* stream) {
  if (!lse) return FAVIT_ERR_INVALID;
  return attn_entry(false, qkv, nullptr, out, mask, B, L, H, hd, W, dtype, dropout_p, seed, stream, nullptr, nullptr, lse);
}

extern "C" int favit_mhla_attn_bwd_lse(const void* qkv, const void* dout, const void* o, const float* lse, void* dqkv,
                                       const uint8_t* mask, int32_t B, int32_t L, int32_t H, int32_t hd, int32_t W,
                                       int dtype, float dropout_p, uint64_t seed, void* stream) {
  if (!o || !lse) return FAVIT_ERR_INVALID;
  return attn_entry(true, qkv, dout, dqkv, mask, B, L, H, hd, W, dtype, dropout_p, seed, stream, o, lse, nullptr);
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

extern "C" int favit_mhla_fold_fwd_multi(int32_t n, const float* const* wqkv, const float* const* bqkv,
                                         const float* const* wl, const float* const* bl, void* const* weff,
                                         int weff_dtype, float* const* beff, int32_t D, int32_t H, void* stream) {
  if (n <= 0 || n > FOLD_MAX || !wqkv || !bqkv || !wl || !bl || !weff || !beff || D <= 0 || H <= 0 || D % H)
    return FAVIT_ERR_INVALID;
  FoldBatch fb;
  for (int i = 0; i < n; ++i) {
    if (!wqkv[i] || !bqkv[i] || !wl[i] || !bl[i] || !weff[i] || !beff[i]) return FAVIT_ERR_INVALID;
    fb.wqkv[i] = wqkv[i]; fb.bqkv[i] = bqkv[i]; fb.wl[i] = wl[i]; fb.bl[i] = bl[i];
    fb.weff[i] = weff[i]; fb.beff[i] = beff[i];
  }
  const int hd = D / H;
  const dim3 grid((D + 1 + FOLD_TC - 1) / FOLD_TC, 2 * H + (D + hd - 1) / hd, n);
  hipStream_t st = as_stream(stream);
#define FAVIT_FOLD_MULTI(T)                                                                                   \
  switch (hd) {                                                                                               \
    case 16: hipLaunchKernelGGL((fold_w_multi_kernel<T, 16>), grid, dim3(256), 0, st, fb, D, H); break;       \
    case 32: hipLaunchKernelGGL((fold_w_multi_kernel<T, 32>), grid, dim3(256), 0, st, fb, D, H); break;       \
    case 64: hipLaunchKernelGGL((fold_w_multi_kernel<T, 64>), grid, dim3(256), 0, st, fb, D, H); break;       \
    default: return FAVIT_ERR_UNSUPPORTED;                                                                    \
  }
  if (weff_dtype == FAVIT_F32) { FAVIT_FOLD_MULTI(float) }
  else if (weff_dtype == FAVIT_BF16) { FAVIT_FOLD_MULTI(bf16_t) }
  else return FAVIT_ERR_INVALID;
#undef FAVIT_FOLD_MULTI
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

extern "C" int favit_mhla_fold_bwd(const float* dweff, const float* dbeff, const float* wqkv, const float* bqkv,
                                   const float* wl, float* dwqkv, float* dbqkv, float* dwl, float* dbl, int32_t D,
                                   int32_t H, int32_t accumulate, void* stream) {
  if (!dweff || !dbeff || !wqkv || !bqkv || !wl || !dwqkv || !dbqkv || !dwl || !dbl || D <= 0 || H <= 0 || D % H)
    return FAVIT_ERR_INVALID;
  const int hd = D / H;
  hipStream_t st = as_stream(stream);
  if (!accumulate) {
    (void)favit_zero_async(dwl, sizeof(float) * hd * hd, st);
    (void)favit_zero_async(dbl, sizeof(float) * hd, st);
  }
  const int gwx = (D + 1 + FOLD_TC - 1) / FOLD_TC, gwy = 2 * H + (D + hd - 1) / hd;      // fold_w grid
  const int glx = 2 * H, gly = FOLD_L_SPLIT;                                             // fold_bwd_l grid
  const int nw = gwx * gwy;
  const dim3 grid((unsigned)(nw + glx * gly));
  switch (hd) {
    case 16: hipLaunchKernelGGL(fold_bwd_all_kernel<16>, grid, dim3(256), 0, st, dweff, dbeff, wqkv, bqkv, wl, dwqkv, dbqkv, dwl, dbl, D, H, accumulate, gwx, nw, glx); break;
    case 32: hipLaunchKernelGGL(fold_bwd_all_kernel<32>, grid, dim3(256), 0, st, dweff, dbeff, wqkv, bqkv, wl, dwqkv, dbqkv, dwl, dbl, D, H, accumulate, gwx, nw, glx); break;
    case 64: hipLaunchKernelGGL(fold_bwd_all_kernel<64>, grid, dim3(256), 0, st, dweff, dbeff, wqkv, bqkv, wl, dwqkv, dbqkv, dwl, dbl, D, H, accumulate, gwx, nw, glx); break;
    default: return FAVIT_ERR_UNSUPPORTED;
  }
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

extern "C" int favit_mhla_fold_bwd_multi(int32_t n, const float* const* dweff, const float* const* dbeff,
                                         const float* const* wqkv, const float* const* bqkv, const float* const* wl,
                                         float* const* dwqkv, float* const* dbqkv, float* const* dwl, float* const* dbl,
                                         int32_t D, int32_t H, void* stream) {
  if (n <= 0 || n > FOLD_BWD_MAX || !dweff || !dbeff || !wqkv || !bqkv || !wl || !dwqkv || !dbqkv || !dwl || !dbl ||
      D <= 0 || H <= 0 || D % H)
    return FAVIT_ERR_INVALID;
  // dwqkv[i] == dbqkv[i] == NULL for EVERY layer: the qkv projection is frozen (fine-tuning), only the latent_proj
  // gradients are produced and the workgroups of the qkv part are not launched
  const bool lat_only = dwqkv[0] == nullptr;
  FoldBwdBatch fb;
  for (int i = 0; i < n; ++i) {
    if (!dweff[i] || !dbeff[i] || !wqkv[i] || !bqkv[i] || !wl[i] || !dwl[i] || !dbl[i]) return FAVIT_ERR_INVALID;
    if (lat_only ? (dwqkv[i] || dbqkv[i]) : (!dwqkv[i] || !dbqkv[i])) return FAVIT_ERR_INVALID;
    fb.dweff[i] = dweff[i]; fb.dbeff[i] = dbeff[i]; fb.wqkv[i] = wqkv[i]; fb.bqkv[i] = bqkv[i]; fb.wl[i] = wl[i];
    fb.dwqkv[i] = dwqkv[i]; fb.dbqkv[i] = dbqkv[i]; fb.dwl[i] = dwl[i]; fb.dbl[i] = dbl[i];
  }
  const int hd = D / H;
  const int gwx = (D + 1 + FOLD_TC - 1) / FOLD_TC, gwy = 2 * H + (D + hd - 1) / hd;      // fold_w grid
  const int glx = 2 * H, gly = FOLD_L_SPLIT;                                             // fold_bwd_l grid
  const int nw = lat_only ? 0 : gwx * gwy;
  const dim3 grid((unsigned)(nw + glx * gly), (unsigned)n);
  hipStream_t st = as_stream(stream);
  switch (hd) {
    case 16: hipLaunchKernelGGL(fold_bwd_multi_kernel<16>, grid, dim3(256), 0, st, fb, D, H, gwx, nw, glx); break;
    case 32: hipLaunchKernelGGL(fold_bwd_multi_kernel<32>, grid, dim3(256), 0, st, fb, D, H, gwx, nw, glx); break;
    case 64: hipLaunchKernelGGL(fold_bwd_multi_kernel<64>, grid, dim3(256), 0, st, fb, D, H, gwx, nw, glx); break;
    default: return FAVIT_ERR_UNSUPPORTED;
  }
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}
