// SPPP (superpixel patch pooling) device kernels for gfx950: the reference runs these as
// per-image / per-patch Python loops with one device sync per patch
// (models/sppp.py:91-128,192-223; models/sppp_mhla.py:226-262,286-297).  The label map
// [B,HW,HW] int64 is an input (SLIC itself is out of scope).  All integer results
// (dominant label, first-appearance rank, grouping) are bit-exact restatements.
#include "common.h"

namespace {

// ---- dominant label per patch: max count, ties -> smallest label (sppp.py:117-120) ----
// One wave per patch.  The labels of a patch are few (a superpixel map has 1-4 distinct labels per 16x16 patch), so
// the wave peels them off one at a time: take the label of the first lane that still has an uncounted pixel, count
// its occurrences with ballots, mark them counted.  Cost ~ (distinct labels) x (pixels / 64) wave steps instead of
// the pixels^2 / 64 comparisons of a per-pixel count (188 -> ~20 us at 128 images of 224 x 224, P = 16).
__global__ __launch_bounds__(64) void dominant_label_kernel(const int64_t* __restrict__ seg, int64_t* __restrict__ dom,
                                                            int HW, int P) {
  extern __shared__ int64_t lab[];       // P*P labels of this patch (patches with more than 64 x 16 pixels)
  const int g = HW / P, n = P * P;
  const int patch = blockIdx.x, b = blockIdx.y;
  const int ph = patch / g, pw = patch % g;
  const int lane = threadIdx.x;
  const int64_t* img = seg + (long)b * HW * HW;
  constexpr int PER = 16;                // pixels per lane held in registers (n <= 1024); larger patches use the LDS copy
  const bool in_regs = n <= 64 * PER;
  int64_t v[PER];
  unsigned todo = 0;                     // bit k: v[k] holds an uncounted pixel
  if (in_regs) {
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int e = lane + 64 * k;
      if (e < n) { v[k] = img[(long)(ph * P + e / P) * HW + pw * P + e % P]; todo |= 1u << k; }
      else v[k] = 0;
    }
    int best_c = -1;
    int64_t best_l = 0;
    while (__ballot(todo != 0)) {
      const int leader = __ffsll((long long)__ballot(todo != 0)) - 1;
      const int k0 = __ffs((int)todo) - 1;                              // meaningful on the leader lane only
      int64_t mine = 0;
#pragma unroll
      for (int k = 0; k < PER; ++k) if (k == k0) mine = v[k];
      const int64_t l = __shfl(mine, leader, 64);
      int c = 0;
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const bool hit = ((todo >> k) & 1u) && v[k] == l;
        c += __popcll(__ballot(hit));
        if (hit) todo &= ~(1u << k);
      }
      if (c > best_c || (c == best_c && l < best_l)) { best_c = c; best_l = l; }
    }
    if (lane == 0) dom[(long)b * g * g + patch] = best_l;
    return;
  }
  for (int e = lane; e < n; e += 64) lab[e] = img[(long)(ph * P + e / P) * HW + pw * P + e % P];
  __syncthreads();
  int best_c = -1;
  int64_t best_l = 0;
  for (int e = lane; e < n; e += 64) {
    const int64_t l = lab[e];
    int c = 0;
    for (int f = 0; f < n; ++f) c += (lab[f] == l);
    if (c > best_c || (c == best_c && l < best_l)) { best_c = c; best_l = l; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const int oc = __shfl_xor(best_c, o, 64);
    const int64_t ol = __shfl_xor(best_l, o, 64);
    if (oc > best_c || (oc == best_c && ol < best_l)) { best_c = oc; best_l = ol; }
  }
  if (lane == 0) dom[(long)b * g * g + patch] = best_l;
}

// ---- token rank = first-appearance order of dominant labels (sppp.py:124-126) ----
__global__ __launch_bounds__(256) void rank_kernel(const int64_t* __restrict__ dom, int32_t* __restrict__ patch_rank,
                                                   int32_t* __restrict__ n_tokens, int32_t* __restrict__ perm,
                                                   int32_t* __restrict__ offs, int N) {
  extern __shared__ int32_t sh[];       // first[N], isfirst[N], rank[N]
  int32_t* first = sh;
  int32_t* isfirst = sh + N;
  int32_t* rk = sh + 2 * N;
  const int b = blockIdx.x;
  const int64_t* d = dom + (long)b * N;
  for (int p = threadIdx.x; p < N; p += 256) {
    const int64_t l = d[p];
    int f = p;
    for (int q = 0; q < p; ++q)
      if (d[q] == l) { f = q; break; }
    first[p] = f;
    isfirst[p] = (f == p);
  }
  __syncthreads();
  for (int p = threadIdx.x; p < N; p += 256) {
    int r = 0;
    for (int q = 0; q < first[p]; ++q) r += isfirst[q];
    rk[p] = r;
    patch_rank[(long)b * N + p] = r;
  }
  __syncthreads();
  int R = 0;
  for (int q = 0; q < N; ++q) R += isfirst[q];
  if (threadIdx.x == 0) n_tokens[b] = R;
  // counting sort by rank (raster order inside a group)
  for (int r = threadIdx.x; r <= N; r += 256) {
    int c = 0;
    for (int q = 0; q < N; ++q) c += (rk[q] < r);
    offs[(long)b * (N + 1) + r] = c;
  }
  for (int p = threadIdx.x; p < N; p += 256) {
    int pos = 0;
    for (int q = 0; q < N; ++q) pos += (rk[q] < rk[p]) || (rk[q] == rk[p] && q < p);
    perm[(long)b * N + pos] = p;
  }
}

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- pooling (sppp.py:209-216): one workgroup per (token r, image b) ----
__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ emb, const int32_t* __restrict__ perm,
                                                       const int32_t* __restrict__ offs, float* __restrict__ out,
                                                       int32_t* __restrict__ argmax, int kind, int N, int R, int D) {
  extern __shared__ float shf[];        // [4] reduction scratch + weights[N]
  float* red = shf;
  float* wts = shf + 4;
  const int r = blockIdx.x, b = blockIdx.y;
  const int beg = offs[(long)b * (N + 1) + r], end = offs[(long)b * (N + 1) + r + 1];
  const int cnt = end - beg;
  const int32_t* pm = perm + (long)b * N + beg;
  const float* e = emb + (long)b * N * D;
  float* o = out + ((long)b * R + r) * D;
  if (cnt <= 0) {
    for (int d = threadIdx.x; d < D; d += 256) o[d] = 0.f;
    return;
  }
  if (kind == FAVIT_POOL_ATTENTION) {
    for (int k = 0; k < cnt; ++k) {
      float s = 0.f;
      for (int d = threadIdx.x; d < D; d += 256) s += e[(long)pm[k] * D + d];
      s = block_sum(s, red);
      if (threadIdx.x == 0) wts[k] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = -INFINITY;
      for (int k = 0; k < cnt; ++k) m = fmaxf(m, wts[k]);
      float l = 0.f;
      for (int k = 0; k < cnt; ++k) { wts[k] = expf(wts[k] - m); l += wts[k]; }
      for (int k = 0; k < cnt; ++k) wts[k] /= l;
    }
    __syncthreads();
  }
  for (int d = threadIdx.x; d < D; d += 256) {
    if (kind == FAVIT_POOL_MEAN) {
      float s = 0.f;
      for (int k = 0; k < cnt; ++k) s += e[(long)pm[k] * D + d];
      o[d] = s / (float)cnt;
    } else if (kind == FAVIT_POOL_MAX) {
      float m = e[(long)pm[0] * D + d];
      int am = pm[0];
      for (int k = 1; k < cnt; ++k) {
        const float v = e[(long)pm[k] * D + d];
        if (v > m) { m = v; am = pm[k]; }
      }
      o[d] = m;
      if (argmax) argmax[((long)b * R + r) * D + d] = am;
    } else {
      float s = 0.f;
      for (int k = 0; k < cnt; ++k) s = fmaf(wts[k], e[(long)pm[k] * D + d], s);
      o[d] = s;
    }
  }
}

__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ emb,
                                                       const int32_t* __restrict__ perm,
                                                       const int32_t* __restrict__ offs,
                                                       const int32_t* __restrict__ argmax, float* __restrict__ demb,
                                                       int kind, int N, int R, int D) {
  extern __shared__ float shf[];
  float* red = shf;
  float* wts = shf + 4;          // w_p
  float* dsv = shf + 4 + N;      // ds_p
  const int r = blockIdx.x, b = blockIdx.y;
  const int beg = offs[(long)b * (N + 1) + r], end = offs[(long)b * (N + 1) + r + 1];
  const int cnt = end - beg;
  if (cnt <= 0) return;
  const int32_t* pm = perm + (long)b * N + beg;
  const float* e = emb + (long)b * N * D;
  const float* go = dout + ((long)b * R + r) * D;
  float* de = demb + (long)b * N * D;
  if (kind == FAVIT_POOL_MEAN) {
    const float inv = 1.0f / (float)cnt;
    for (int k = 0; k < cnt; ++k)
      for (int d = threadIdx.x; d < D; d += 256) de[(long)pm[k] * D + d] = go[d] * inv;
  } else if (kind == FAVIT_POOL_MAX) {
    for (int k = 0; k < cnt; ++k)
      for (int d = threadIdx.x; d < D; d += 256)
        de[(long)pm[k] * D + d] = (argmax[((long)b * R + r) * D + d] == pm[k]) ? go[d] : 0.f;
  } else {
    // out = sum_p w_p e_p, w = softmax_p(sum_d e_pd):  de_pd = w_p*go_d + w_p*(dw_p - sum_q w_q dw_q)
    for (int k = 0; k < cnt; ++k) {
      float s = 0.f, t = 0.f;
      for (int d = threadIdx.x; d < D; d += 256) {
        const float v = e[(long)pm[k] * D + d];
        s += v;
        t = fmaf(go[d], v, t);
      }
      s = block_sum(s, red);
      t = block_sum(t, red);
      if (threadIdx.x == 0) { wts[k] = s; dsv[k] = t; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = -INFINITY;
      for (int k = 0; k < cnt; ++k) m = fmaxf(m, wts[k]);
      float l = 0.f;
      for (int k = 0; k < cnt; ++k) { wts[k] = expf(wts[k] - m); l += wts[k]; }
      float dot = 0.f;
      for (int k = 0; k < cnt; ++k) { wts[k] /= l; dot = fmaf(wts[k], dsv[k], dot); }
      for (int k = 0; k < cnt; ++k) dsv[k] = wts[k] * (dsv[k] - dot);
    }
    __syncthreads();
    for (int k = 0; k < cnt; ++k)
      for (int d = threadIdx.x; d < D; d += 256) de[(long)pm[k] * D + d] = fmaf(wts[k], go[d], dsv[k]);
  }
}

// ---- centroids per LABEL (sppp_mhla.py:226-262): exact integer coordinate sums ----
// One workgroup per image.  Per pixel three 64-bit LDS atomics into one of NC privatised copies of the [S][3] table
// (copy = lane % NC; rows padded to an odd number of 64-bit words so that the copies of one (label, component) fall
// into different banks): the wave-wide shuffle reductions per (wave, label) group that round 2 used cost ~80 VALU
// instructions per group and made this 87 us for 128 images of 224 x 224 (the same finding as in the SLIC assignment,
// DESIGN.md section 4).  Eight pixels per thread are loaded before any is processed.  Integer sums: order-independent.
// NC = 16 for S <= 64, 1 above (the table would not fit).
__global__ __launch_bounds__(1024) void centroid_kernel(const int64_t* __restrict__ seg, float* __restrict__ cent,
                                                        int HW, int S, int NC) {
  extern __shared__ unsigned long long acc[];     // [NC][3 * S + 1]: count, sum x, sum y per label
  const int b = blockIdx.x;
  const int stride = 3 * S + 1;
  for (int i = threadIdx.x; i < NC * stride; i += 1024) acc[i] = 0ull;
  __syncthreads();
  const int64_t* img = seg + (long)b * HW * HW;
  const long n = (long)HW * HW;
  unsigned long long* mine = acc + (threadIdx.x % NC) * stride;
  constexpr int U = 8;
  for (long base = 0; base < n; base += (long)U * 1024) {
    int64_t l[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = base + (long)u * 1024 + threadIdx.x;
      l[u] = i < n ? img[i] : (int64_t)-1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (l[u] >= 0 && l[u] < S) {
        const long i = base + (long)u * 1024 + threadIdx.x;
        unsigned long long* d = mine + 3 * l[u];
        atomicAdd(d, 1ull);
        atomicAdd(d + 1, (unsigned long long)(i % HW));
        atomicAdd(d + 2, (unsigned long long)(i / HW));
      }
    }
  }
  __syncthreads();
  for (int s = threadIdx.x; s < S; s += 1024) {
    unsigned long long c = 0, sx = 0, sy = 0;
    for (int k = 0; k < NC; ++k) {
      c += acc[k * stride + 3 * s];
      sx += acc[k * stride + 3 * s + 1];
      sy += acc[k * stride + 3 * s + 2];
    }
    float cx = 0.5f, cy = 0.5f;
    if (c > 0) {
      cx = (float)((double)sx / ((double)HW * (double)c));
      cy = (float)((double)sy / ((double)HW * (double)c));
    }
    cent[((long)b * S + s) * 2 + 0] = cx;
    cent[((long)b * S + s) * 2 + 1] = cy;
  }
}

// ---- centroid positional encoding (sppp.py:267-300) ----
__global__ void posenc_kernel(const float* __restrict__ x, const float* __restrict__ cent, float* __restrict__ y, int B,
                              int L, int D, int n_cent) {
  const long total = (long)B * L * D;
  const int half = D / 2;
  const float step = -logf(10000.0f) / (float)half;
  const bool prepend = n_cent < L;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int d = (int)(i % D), l = (int)((i / D) % L), b = (int)(i / ((long)D * L));
    if (!cent) {   // classic index sinusoid (sppp.py:257-266): pe[l,2i]=sin(l*div_i), pe[l,2i+1]=cos(l*div_i)
      const float div = expf((float)(d & ~1) * (-logf(10000.0f) / (float)D));
      y[i] = x[i] + ((d & 1) ? cosf((float)l * div) : sinf((float)l * div));
      continue;
    }
    float cx = 0.5f, cy = 0.5f;
    const int ci = prepend ? l - 1 : l;
    if (ci >= 0) {
      cx = cent[((long)b * n_cent + ci) * 2];
      cy = cent[((long)b * n_cent + ci) * 2 + 1];
    }
    float pe;
    if (d < half) pe = sinf(cx * expf((float)d * step));
    else pe = cosf(cy * expf((float)(d - half) * step));
    y[i] = x[i] + pe;
  }
}

}  // namespace

extern "C" int favit_sppp_map_patches(const int64_t* seg, int32_t* patch_rank, int32_t* n_tokens, int32_t* perm,
                                      int32_t* offs, int64_t* dom_ws, int32_t B, int32_t HW, int32_t P, void* stream) {
  if (!seg || !patch_rank || !n_tokens || !perm || !offs || !dom_ws || B <= 0 || HW <= 0 || P <= 0 || HW % P)
    return FAVIT_ERR_INVALID;
  const int g = HW / P, N = g * g;
  if (P * P > 4096 || N > 4096) return FAVIT_ERR_UNSUPPORTED;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(dominant_label_kernel, dim3(N, B), dim3(64), sizeof(int64_t) * P * P, st, seg, dom_ws, HW, P);
  FAVIT_CHECK_LAUNCH();
  hipLaunchKernelGGL(rank_kernel, dim3(B), dim3(256), sizeof(int32_t) * 3 * N, st, dom_ws, patch_rank, n_tokens, perm, offs, N);
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

extern "C" int favit_sppp_pool_fwd(const float* emb, const int32_t* perm, const int32_t* offs, float* out,
                                   int32_t* argmax, int32_t kind, int32_t B, int32_t N, int32_t R, int32_t D,
                                   void* stream) {
  if (!emb || !perm || !offs || !out || B <= 0 || N <= 0 || R <= 0 || D <= 0) return FAVIT_ERR_INVALID;
  if (kind < 0 || kind > 2) return FAVIT_ERR_INVALID;
  if (kind == FAVIT_POOL_MAX && !argmax) return FAVIT_ERR_INVALID;
  hipLaunchKernelGGL(pool_fwd_kernel, dim3(R, B), dim3(256), sizeof(float) * (4 + N), as_stream(stream), emb, perm, offs,
                     out, argmax, kind, N, R, D);
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

extern "C" int favit_sppp_pool_bwd(const float* dout, const float* emb, const int32_t* patch_rank, const int32_t* perm,
                                   const int32_t* offs, const int32_t* argmax, float* demb, int32_t kind, int32_t B,
                                   int32_t N, int32_t R, int32_t D, void* stream) {
  (void)patch_rank;
  if (!dout || !emb || !perm || !offs || !demb || B <= 0 || N <= 0 || R <= 0 || D <= 0) return FAVIT_ERR_INVALID;
  if (kind < 0 || kind > 2) return FAVIT_ERR_INVALID;
  if (kind == FAVIT_POOL_MAX && !argmax) return FAVIT_ERR_INVALID;
  hipLaunchKernelGGL(pool_bwd_kernel, dim3(R, B), dim3(256), sizeof(float) * (4 + 2 * N), as_stream(stream), dout, emb,
                     perm, offs, argmax, demb, kind, N, R, D);
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

extern "C" int favit_sppp_centroids(const int64_t* seg, float* cent, int32_t B, int32_t HW, int32_t S, void* stream) {
  if (!seg || !cent || B <= 0 || HW <= 0 || S <= 0) return FAVIT_ERR_INVALID;
  if (S > 2048) return FAVIT_ERR_UNSUPPORTED;
  const int nc = S <= 64 ? 16 : 1;
  hipLaunchKernelGGL(centroid_kernel, dim3(B), dim3(1024), sizeof(unsigned long long) * (size_t)nc * (3 * S + 1), as_stream(stream),
                     seg, cent, HW, S, nc);
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

extern "C" int favit_sppp_posenc_fwd(const float* x, const float* cent, float* y, int32_t B, int32_t L, int32_t D,
                                     int32_t n_cent, void* stream) {
  if (!x || !y || B <= 0 || L <= 0 || D <= 0 || (D & 1)) return FAVIT_ERR_INVALID;
  if (cent && n_cent != L && n_cent != L - 1) return FAVIT_ERR_INVALID;     // the only shapes the reference accepts (sppp.py:271-299)
  const long total = (long)B * L * D;
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(posenc_kernel, dim3(grid), dim3(256), 0, as_stream(stream), x, cent, y, B, L, D, n_cent);
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}
