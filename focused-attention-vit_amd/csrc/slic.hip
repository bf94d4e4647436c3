// SLIC superpixels on the device: replaces the per-image  D2H -> skimage.segmentation.slic -> H2D  hop of
// SuperpixelSegmentation.segment (reference models/sppp.py:44-74) so that the SPPP front end
// (label map -> patch mapping -> pooling -> centroids, csrc/sppp.hip) never leaves the GPU.
//
// scikit-image is a third-party dependency the reference does not pin and the image does not ship: PARITY WITH
// skimage IS UNPINNED.  What is built here is the published algorithm (Achanta et al., SLIC, as scikit-image
// parametrises it: sigma pre-smoothing, CIELAB, grid seeds, 2*step search windows, distance
// spatial^2 / step^2 + (dLab / compactness)^2, max_num_iter iterations, connectivity enforcement with
// min_size_factor 0.5) with every decision after the colour conversion in INTEGER arithmetic, so the result is
// deterministic and reproducible bit for bit by the CPU restatement oracle/slic_oracle.py:
//   stage 1  favit_slic_features : per-image min-max rescale of the input to [0, 1] over all three channels
//            (scikit-image >= 0.19 does this first, "to make choice of compactness insensitive to input image scale";
//            the reference feeds mean/std-NORMALISED tensors, values ~ -2 .. 2.6, so without it the Lab conversion and
//            the meaning of compactness = 0.1 would differ from the reference's call), gaussian blur (reflect boundary,
//            radius int(4 sigma + 0.5)) + sRGB -> CIELAB, quantised to 1/16 units (int16 x 4 per pixel).  Float work:
//            compared with a tolerance.  The blur is linear with weights summing to 1, so the rescale is applied to
//            the blurred value.  minmax = NULL skips the rescale (the behaviour of scikit-image < 0.19).
//   stage 2  favit_slic_cluster  : k-means in (y, x, L, a, b), fixed point (1/16 pixel, 1/16 Lab), int64 distances
//            16^2 * spatial^2 + coef * dq^2 with coef = round(step^2 / compactness^2), first-minimum ties, centres =
//            truncated integer means.  One assignment launch over every pixel of every image + one centre update per
//            iteration (round 2 used one workgroup per image: 6.0 ms for 128 images of 224x224, half the chip idle);
//            per-cluster sums reduced per wave, then per workgroup in LDS, then 64-bit global atomics.  Integer
//            work: bit-exact, independent of the order of the atomics.
//   stage 3  favit_slic_connect  : 4-connected components of the cluster map by union-find over every pixel in
//            parallel (the smaller root wins, so a component's root is its smallest pixel index whatever the order of
//            the unions; round 2 iterated min-propagation in one workgroup per image: 10.5 ms), components in raster
//            order of their first pixel; those smaller than min_size take the label of an already-labelled neighbour
//            of their first pixel (x+1, x-1, y+1, y-1; the last one found wins), the others get consecutive labels
//            from 0.  Integer work: bit-exact.
#include "common.h"

namespace {

constexpr int SLIC_THREADS = 1024;
constexpr int SLIC_MAXK = 64;          // cluster centres per image
constexpr int SLIC_MAXC = 2048;        // connected components per image handled by stage 3

__device__ __forceinline__ int reflect_idx(int i, int n) {
  // scipy.ndimage 'reflect': (d c b a | a b c d | d c b a)
  while (i < 0 || i >= n) i = i < 0 ? -i - 1 : 2 * n - 1 - i;
  return i;
}

__device__ __forceinline__ float srgb_to_linear(float c) {
  return c > 0.04045f ? powf((c + 0.055f) / 1.055f, 2.4f) : c / 12.92f;
}
__device__ __forceinline__ float lab_f(float t) { return t > 0.008856f ? cbrtf(t) : 7.787f * t + 16.0f / 116.0f; }

// per-image minimum and maximum over all channels: one 1024-thread workgroup per image (3*H*W floats, ~150 per thread)
__global__ __launch_bounds__(1024) void slic_minmax_kernel(const float* __restrict__ img, float* __restrict__ minmax, long n) {
  const float* im = img + (long)blockIdx.x * n;
  float lo = INFINITY, hi = -INFINITY;
  for (long i = threadIdx.x; i < n; i += 1024) {
    const float v = im[i];
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  }
  lo = -wave_max(-lo);
  hi = wave_max(hi);
  __shared__ float slo[16], shi[16];
  if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) { lo = fminf(lo, slo[w]); hi = fmaxf(hi, shi[w]); }
    minmax[2 * blockIdx.x] = lo;
    minmax[2 * blockIdx.x + 1] = hi;
  }
}

// Gaussian weights of the separable pre-smoothing, computed once on the host (radius <= 32)
struct BlurTaps { float w[65]; int radius; };

// pass 1: horizontal blur of the three colour planes into tmp (fp32 [B,3,H,W]); radius 0: not launched
__global__ __launch_bounds__(256) void slic_blur_x_kernel(const float* __restrict__ img, float* __restrict__ tmp, long n,
                                                          int W, BlurTaps t) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % W);
  const float* row = img + (i - x);
  float acc = 0.f;
  for (int d = -t.radius; d <= t.radius; ++d) acc = fmaf(t.w[d + t.radius], row[reflect_idx(x + d, W)], acc);
  tmp[i] = acc;
}

// pass 2: vertical blur (of tmp, or nothing when radius = 0), per-image min-max rescale, sRGB -> CIELAB, quantise
__global__ __launch_bounds__(256) void slic_features_kernel(const float* __restrict__ src, short* __restrict__ feat, int B,
                                                            int H, int W, BlurTaps t, const float* __restrict__ minmax) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  const long HW = (long)H * W;
  if (p >= (long)B * HW) return;
  const int b = (int)(p / HW);
  const int y = (int)((p - b * HW) / W), x = (int)((p - b * HW) % W);
  const float* im = src + (long)b * 3 * HW;
  float rgb[3] = {0.f, 0.f, 0.f};
  if (t.radius == 0) {
    for (int c = 0; c < 3; ++c) rgb[c] = im[c * HW + (long)y * W + x];
  } else {
    for (int d = -t.radius; d <= t.radius; ++d) {
      const long o = (long)reflect_idx(y + d, H) * W + x;
      const float w = t.w[d + t.radius];
      for (int c = 0; c < 3; ++c) rgb[c] = fmaf(w, im[c * HW + o], rgb[c]);
    }
  }
  if (minmax) {                      // image -= min; image /= (max - min)  unless the image is constant
    const float lo = minmax[2 * b], hi = minmax[2 * b + 1];
    const float range = hi - lo;
    for (int c = 0; c < 3; ++c) {
      rgb[c] -= lo;
      if (range != 0.f) rgb[c] = __fdiv_rn(rgb[c], range);
    }
  }
  const float r = srgb_to_linear(rgb[0]), g = srgb_to_linear(rgb[1]), bl = srgb_to_linear(rgb[2]);
  const float X = (0.412453f * r + 0.357580f * g + 0.180423f * bl) / 0.95047f;
  const float Y = 0.212671f * r + 0.715160f * g + 0.072169f * bl;
  const float Z = (0.019334f * r + 0.119193f * g + 0.950227f * bl) / 1.08883f;
  const float fx = lab_f(X), fy = lab_f(Y), fz = lab_f(Z);
  const float lab[3] = {116.0f * fy - 16.0f, 500.0f * (fx - fy), 200.0f * (fy - fz)};
  short* o = feat + p * 4;
  for (int c = 0; c < 3; ++c) {
    float q = rintf(lab[c] * 16.0f);
    q = fminf(fmaxf(q, -8191.f), 8191.f);       // 13 bits + sign (CIELAB x 16 of an sRGB colour stays inside +-2048):
                                                // differences fit 24-bit multiplies, three squares fit 32 bits
    o[c] = (short)q;
  }
  o[3] = 0;
}

// sum of v over the lanes in `mask` (all lanes execute; lanes outside contribute 0), entirely in the VALU: four DPP
// steps inside each row of 16 lanes (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror), then the
// gfx950 row swaps v_permlane16_swap / v_permlane32_swap across the four rows.  (__shfl_xor is ds_bpermute_b32: six
// round trips through the LDS crossbar per sum, five sums per cluster group and wave.)
__device__ __forceinline__ int dpp_sum16_i(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);
  return v;
}
__device__ __forceinline__ int masked_wave_sum(int v, bool in) {
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
  int s = dpp_sum16_i(in ? v : 0);
  unsigned u = (unsigned)s;
  u32x2 r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  u += ((threadIdx.x >> 4) & 1) ? r.x : r.y;
  r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  u += ((threadIdx.x >> 5) & 1) ? r.x : r.y;
  return (int)u;
}

// ---- stage 2: k-means, one launch pair per iteration, every pixel of every image in parallel ----
// workspace per image: sums[K][6] int64 (y16, x16, l, a, b, count), then cen[K][5] int32.  All sums are exact
// integers, so the order in which workgroups add them does not matter: bit-identical to the one-workgroup-per-image
// form this replaces (and to oracle/slic_oracle.py).  The old kernel stopped at the first iteration that changed no
// label; further iterations of a fixed point reproduce it, so running all of them gives the same labels.
__host__ __device__ __forceinline__ size_t slic_ws_stride(int K) {       // bytes per image, a multiple of 8
  return (size_t)K * 48 + (((size_t)K * 20 + 7) & ~(size_t)7) + 8;      // + the arrival counter of the assignment pass
}
__device__ __forceinline__ unsigned* slic_arrivals(void* ws, int b, int K) {
  return reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + (size_t)(b + 1) * slic_ws_stride(K) - 8);
}
__device__ __forceinline__ long long* slic_sums(void* ws, int b, int K) {
  return reinterpret_cast<long long*>(reinterpret_cast<char*>(ws) + (size_t)b * slic_ws_stride(K));
}
__device__ __forceinline__ int* slic_cen(void* ws, int b, int K) {
  return reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + (size_t)b * slic_ws_stride(K) + (size_t)K * 48);
}

__global__ __launch_bounds__(64) void slic_seed_kernel(const short* __restrict__ feat, const int* __restrict__ init_yx,
                                                       void* ws, int K, int H, int W) {
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid >= K) return;
  const short* f = feat + (long)b * H * W * 4;
  const int cy = init_yx[2 * tid], cx = init_yx[2 * tid + 1];
  const short* q = f + ((long)cy * W + cx) * 4;
  int* cen = slic_cen(ws, b, K) + tid * 5;
  cen[0] = cy * 16; cen[1] = cx * 16; cen[2] = q[0]; cen[3] = q[1]; cen[4] = q[2];
  long long* sm = slic_sums(ws, b, K) + tid * 6;
#pragma unroll
  for (int c = 0; c < 6; ++c) sm[c] = 0;
  if (tid == 0) *slic_arrivals(ws, b, K) = 0u;
}

// FAST (H, W <= 2047 and coef < 2^32, i.e. every real call): every difference fits 16 bits and every sum of squares 31
// bits, so a candidate costs two packed subtractions, two v_dot2_i32_i16, one 24-bit multiply and ONE v_mad_u64_u32
// (v_mul_lo_u32 / v_mad_i64_i32 of the general form run at a quarter of the VALU rate) -- the same integers as the 64-bit
// form, which stays for out-of-range arguments.  The kernel is VALU-bound: ~11 of 16 centres pass the window test for
// an average pixel of a 224 x 224 image (the 2 step x 2 step search window is part of the algorithm).
// A workgroup of 256 threads owns 1024 consecutive pixels, four per thread (p = base + j*256 + tid: coalesced): the
// four feature loads are in flight together and up to eight workgroups share a CU.  The first form, one pixel per
// thread in 1024-thread workgroups (two per CU), spent its time in the serial chain centre load -> barrier -> pixel
// load -> loop -> reduction -> barrier -> atomics of each workgroup: 163 us per iteration for 57 MB.
constexpr int ASG_THREADS = 256, ASG_PPT = 4, ASG_TILE = ASG_THREADS * ASG_PPT;
typedef short s16x2_t __attribute__((ext_vector_type(2)));
template <bool FAST>
__global__ __launch_bounds__(ASG_THREADS) void slic_assign_kernel(const short* __restrict__ feat, uint8_t* __restrict__ labels,
                                                                  void* ws, int K, int H, int W, int step, long long coef
#ifdef FAVIT_PROBE
                                                                  , int dbg      // probe build: 1 no candidate loop, 2 no
#endif                                                                           // reduction, 4 no global atomics (timing)
) {
#ifndef FAVIT_PROBE
  constexpr int dbg = 0;
#endif
  __shared__ int cen[SLIC_MAXK][5];      // y16, x16, l, a, b
  // FAST form of a centre: packed 16-bit pairs (y16, x16) and (l, a), b, and the window origin (cy - 2 step, cx - 2 step):
  // the differences are two v_pk_sub_i16, the two sums of squares two v_dot2_i32_i16 (every value fits 16 bits and
  // every sum 31 bits for H, W <= 2047 and features within +-8191), the window test two unsigned compares
  __shared__ __attribute__((aligned(16))) int cpk[SLIC_MAXK][8];
  __shared__ int sums[SLIC_MAXK][6];     // this workgroup's <= 1024 pixels: fits 32 bits
  extern __shared__ int bins[];          // [16 copies][K * 6 + 1]
  const int b = blockIdx.y, tid = threadIdx.x;
  const int HW = H * W;
  const int bstride = K * 6 + 1;
  const short* f = feat + (long)b * HW * 4;
  uint8_t* lab = labels + (long)b * HW;
  for (int i = tid; i < K * 5; i += ASG_THREADS) cen[i / 5][i % 5] = slic_cen(ws, b, K)[i];
  for (int i = tid; i < 16 * bstride; i += ASG_THREADS) bins[i] = 0;
  if (FAST) {
    for (int k = tid; k < K; k += ASG_THREADS) {
      const int* c = slic_cen(ws, b, K) + 5 * k;
      cpk[k][0] = (c[0] & 0xffff) | (c[1] << 16);
      cpk[k][1] = (c[2] & 0xffff) | (c[3] << 16);
      cpk[k][2] = c[4];
      cpk[k][3] = (c[0] >> 4) - 2 * step;
      cpk[k][4] = (c[1] >> 4) - 2 * step;
    }
  }
  typedef __attribute__((ext_vector_type(4))) short short4_t;
  const unsigned coef32 = (unsigned)coef;
  const unsigned win = 4u * (unsigned)step;
  __syncthreads();
  // A thread owns ASG_PPT CONSECUTIVE pixels (p = base + ASG_PPT * tid + j: a wave still reads 2 KiB contiguously).
  // Neighbouring pixels mostly share their cluster, so the per-cluster sums are first folded inside the thread
  // (equal neighbours: entry j into entry j - 1, from the right, so a run ends up in its first pixel) and only then
  // reduced across the wave, slot by slot: slot 0 has every lane, slots 1..3 only the lanes where a new run starts.
  // The wave-level reduction was 60 % of the kernel when it ran once per pixel slot (probe switches).
  // (More pixels per workgroup -- fewer of the contended 64-bit global atomics at its end -- measured slower: 4096 per
  // workgroup leave 1,664 workgroups for 2,048 slots and the k-means pass went 1.56 -> 2.01 ms.)
  short4_t qv[ASG_PPT];
  int bk[ASG_PPT], sy[ASG_PPT], sx[ASG_PPT], sl[ASG_PPT], sa[ASG_PPT], sb[ASG_PPT], sc[ASG_PPT];
  const int p0 = blockIdx.x * ASG_TILE + ASG_PPT * tid;
#pragma unroll
  for (int j = 0; j < ASG_PPT; ++j)
    qv[j] = p0 + j < HW ? *reinterpret_cast<const short4_t*>(f + (long)(p0 + j) * 4) : (short4_t){0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < ASG_PPT; ++j) {
    const int p = p0 + j;
    const bool live = p < HW;
    int best_k = 255, y = 0, x = 0;
    const int q0 = qv[j][0], q1 = qv[j][1], q2 = qv[j][2];
    if (live) {
      y = p / W; x = p - y * W;
      unsigned long long best = ~0ull;
      if (dbg & 1) { best_k = (x * 4 / W) + 4 * (y * 4 / H); best = 0; }
      else if (FAST) {
        const s16x2_t pyx = {(short)(16 * y), (short)(16 * x)};
        const s16x2_t pla = {(short)q0, (short)q1};
        // branch-free body (a centre outside the window gets the distance ~0, which never wins the strict <): the
        // compiler unrolls it and issues the LDS reads of four centres together instead of one dependent
        // read -> test -> branch chain per centre
#pragma unroll 4
        for (int k = 0; k < K; ++k) {
          const int c_yx = cpk[k][0], c_la = cpk[k][1], c_b = cpk[k][2], ylo = cpk[k][3], xlo = cpk[k][4];
          // y in [cy - 2 step, cy + 2 step]  <=>  (unsigned)(y - (cy - 2 step)) <= 4 step
          const bool inwin = ((unsigned)(y - ylo) <= win) & ((unsigned)(x - xlo) <= win);
          const s16x2_t dyx = pyx - __builtin_bit_cast(s16x2_t, c_yx);
          const s16x2_t dla = pla - __builtin_bit_cast(s16x2_t, c_la);
          const int db = q2 - c_b;
          const unsigned sp = (unsigned)__builtin_amdgcn_sdot2(dyx, dyx, 0, false);
          const unsigned cq = (unsigned)__builtin_amdgcn_sdot2(dla, dla, __mul24(db, db), false);
          unsigned long long d = (unsigned long long)coef32 * cq + sp;
          d = inwin ? d : ~0ull;
          if (d < best) { best = d; best_k = k; }
        }
      } else {
        for (int k = 0; k < K; ++k) {
          const int cy = cen[k][0] >> 4, cx = cen[k][1] >> 4;               // int(centre), centres are >= 0
          if (y < cy - 2 * step || y > cy + 2 * step || x < cx - 2 * step || x > cx + 2 * step) continue;
          // 32-bit differences, 32x32 -> 64-bit products: the same integers as 64-bit arithmetic throughout
          const int dy = 16 * y - cen[k][0], dx = 16 * x - cen[k][1];
          const int dl = q0 - cen[k][2], da = q1 - cen[k][3], db = q2 - cen[k][4];
          const long long sp = (long long)dy * dy + (long long)dx * dx;
          const long long cq = (long long)dl * dl + (long long)da * da + (long long)db * db;
          const unsigned long long d = (unsigned long long)(sp + coef * cq);
          if (d < best) { best = d; best_k = k; }
        }
      }
      if (best_k == 255) best_k = lab[p];                                    // no window covers the pixel: keep
      else lab[p] = (uint8_t)best_k;
    }
    bk[j] = live ? best_k : -1;                   // -1: no pixel (never equal to a cluster, never reduced)
    sy[j] = 16 * y; sx[j] = 16 * x; sl[j] = q0; sa[j] = q1; sb[j] = q2; sc[j] = live ? 1 : 0;
  }
  // fold equal neighbours inside the thread, from the right
#pragma unroll
  for (int j = ASG_PPT - 1; j > 0; --j) {
    if (bk[j] >= 0 && bk[j] == bk[j - 1]) {
      sy[j - 1] += sy[j]; sx[j - 1] += sx[j]; sl[j - 1] += sl[j]; sa[j - 1] += sa[j]; sb[j - 1] += sb[j]; sc[j - 1] += sc[j];
      bk[j] = -1;
    }
  }
  // per-cluster sums: privatised LDS bins, copy = lane % 16 (rows padded to an odd number of words: the 16 copies of
  // one (cluster, component) fall into 16 different banks; the four lanes that share a copy are the only same-address
  // collisions of an instruction).  Six ds_add_u32 per run instead of six wave-wide reductions per (wave, cluster)
  // group -- those reductions, ~80 VALU instructions per group, were 60 % of the kernel (probe switches).
  if (!(dbg & 2)) {
    int* mine = bins + (tid & 15) * bstride;
#pragma unroll
    for (int j = 0; j < ASG_PPT; ++j) {
      if (bk[j] >= 0 && bk[j] < K) {
        int* d = mine + bk[j] * 6;
        atomicAdd(d + 0, sy[j]); atomicAdd(d + 1, sx[j]); atomicAdd(d + 2, sl[j]);
        atomicAdd(d + 3, sa[j]); atomicAdd(d + 4, sb[j]); atomicAdd(d + 5, sc[j]);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < K * 6; i += ASG_THREADS) {
    int t = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) t += bins[c * bstride + i];
    sums[i / 6][i % 6] = t;
  }
  __syncthreads();
  if (dbg & 4) return;
  for (int i = tid; i < K * 6; i += ASG_THREADS)
    if (sums[i / 6][i % 6] != 0)
      atomicAdd(reinterpret_cast<unsigned long long*>(slic_sums(ws, b, K)) + i, (unsigned long long)(long long)sums[i / 6][i % 6]);
  // The centre update rides on the LAST workgroup of the image to arrive (round 4: one launch per iteration instead
  // of two -- ten 4.5-us launches per call): every workgroup adds its sums with agent-scope atomics (performed at the
  // memory side, coherent across the XCDs' L2s), waits for their acknowledgement (the barrier's vmcnt(0)) and then
  // takes a ticket; the one that draws the last ticket reads the totals with atomics as well, divides, and clears the
  // sums and the counter for the next iteration.  Exact integers as before: the order of arrival changes nothing.
  // (NO __threadfence(): an agent-scope fence writes back and invalidates the L2 -- with one per workgroup the pass
  // went from 65 us to 570 us per iteration.)
  __shared__ unsigned last_flag;
  __syncthreads();
  if (tid == 0)
    last_flag = __hip_atomic_fetch_add(slic_arrivals(ws, b, K), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!last_flag) return;
  if (tid < K) {
    // (atomic reads: the sums were written by other workgroups' L2 atomics; never through this CU's vector cache)
    unsigned long long* sm = reinterpret_cast<unsigned long long*>(slic_sums(ws, b, K)) + tid * 6;
    int* cn = slic_cen(ws, b, K) + tid * 5;
    long long v[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) v[c] = (long long)atomicExch(sm + c, 0ull);
    if (v[5] > 0) {
#pragma unroll
      for (int c = 0; c < 5; ++c) cn[c] = (int)(v[c] / v[5]);                // truncation toward zero
    }
  }
  if (tid == 0) *slic_arrivals(ws, b, K) = 0u;
}

// ---- stage 3: connected components by union-find (link the larger root under the smaller: the root of a component is
// its smallest pixel index, whatever the order of the unions), every pixel of every image in parallel ----
__device__ __forceinline__ int cc_find(const int* comp, int x) {
  int r = comp[x];
  while (r != x) { x = r; r = comp[x]; }
  return x;
}
__device__ __forceinline__ void cc_union(int* comp, int a, int b) {
  while (true) {
    a = cc_find(comp, a);
    b = cc_find(comp, b);
    if (a == b) return;
    if (a > b) { const int t = a; a = b; b = t; }
    const int old = atomicMin(&comp[b], a);       // b was a root: hang it under a (or under something smaller still)
    if (old == b) return;
    b = old;                                      // somebody re-parented b meanwhile: merge that parent with a
  }
}

// One workgroup per image row: comp[p] = first pixel of p's horizontal run of equal labels (an inclusive max-scan of
// the run-start positions in LDS: no atomics, no pointer chasing), aux[p] = 0.  Replaces the per-pixel "comp[p] = p"
// + union with the left neighbour (~50,000 atomicMin chains per image).
__global__ __launch_bounds__(256) void cc_runs_kernel(const uint8_t* __restrict__ labels, int* __restrict__ comp_ws,
                                                      int* __restrict__ aux_ws, int H, int W) {
  __shared__ int start[2][256];
  const int y = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const long base = (long)b * H * W + (long)y * W;
  const uint8_t* lab = labels + base;
  for (int x0 = 0; x0 < W; x0 += 256) {                        // rows wider than 256: chunks; a run crossing a chunk
    const int x = x0 + tid;                                    // border starts again there (joined by cc_merge below)
    int v = -1;
    if (x < W) v = (tid == 0 || lab[x - 1] != lab[x]) ? x : -1;
    start[0][tid] = v;
    __syncthreads();
    int cur = 0;
    for (int o = 1; o < 256; o <<= 1) {
      int m = start[cur][tid];
      if (tid >= o) m = max(m, start[cur][tid - o]);
      start[cur ^ 1][tid] = m;
      cur ^= 1;
      __syncthreads();
    }
    if (x < W) {
      comp_ws[base + x] = y * W + start[cur][tid];
      aux_ws[base + x] = 0;
    }
    __syncthreads();
  }
}

// Unions across run borders only: (p, p - W) of equal label where p or p - W starts its run -- if neither does, the
// pair (p - 1, p - W - 1) joins the same two runs -- and, for rows wider than one chunk, (p, p - 1) at chunk borders.
// The root of a component is still its smallest pixel index whatever the order of the unions.
__global__ __launch_bounds__(256) void cc_merge_kernel(const uint8_t* __restrict__ labels, int* __restrict__ comp_ws, int H, int W) {
  const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  const int HW = H * W;
  if (p >= HW) return;
  const uint8_t* lab = labels + (long)b * HW;
  int* comp = comp_ws + (long)b * HW;
  const int y = p / W, x = p - y * W;
  const uint8_t l = lab[p];
  if (x > 0 && (x & 255) == 0 && lab[p - 1] == l) cc_union(comp, p, p - 1);
  if (y > 0 && lab[p - W] == l) {
    const bool s_here = x == 0 || (x & 255) == 0 || lab[p - 1] != l;
    const bool s_up = x == 0 || (x & 255) == 0 || lab[p - W - 1] != l;
    if (s_here || s_up) cc_union(comp, p, p - W);
  }
}

// comp[p] = root; component sizes at the roots, one atomic per (wave, root) instead of one per pixel (64 consecutive
// pixels belong to one to three components: ~50,000 atomics on a few dozen addresses per image before)
__global__ __launch_bounds__(256) void cc_compress_kernel(int* __restrict__ comp_ws, int* __restrict__ aux_ws, int HW) {
  const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  const bool live = p < HW;
  int* comp = comp_ws + (long)b * HW;
  int r = -1;
  if (live) {
    r = cc_find(comp, p);
    comp[p] = r;                                  // values only move toward the root: concurrent finds stay correct
  }
  unsigned long long todo = __ballot(live);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int rr = __shfl(r, leader, 64);
    const unsigned long long grp = __ballot(live && r == rr);
    if ((threadIdx.x & 63) == leader) atomicAdd(&aux_ws[(long)b * HW + rr], __popcll(grp));
    todo &= ~grp;
  }
}

// per image: roots in raster order, consecutive labels for components >= min_size, a small component takes the label
// of an already-labelled neighbour of its first pixel (x+1, x-1, y+1, y-1; the last one found wins).  Leaves the final
// label of every component in aux[root].  Everything the serial pass reads (component sizes, the ranks of the four
// neighbour components) is gathered in parallel into LDS first: the pass itself touches no global memory (it used to
// chase aux[] / comp[] through L2 once per component: 0.5 ms for 128 images).
__global__ __launch_bounds__(SLIC_THREADS) void cc_relabel_kernel(const int* __restrict__ comp_ws, int* __restrict__ aux_ws,
                                                                  int* __restrict__ n_regions, int H, int W, int min_size) {
  __shared__ int n_roots;
  __shared__ int root_px[SLIC_MAXC];     // unsorted roots, then component sizes in rank order
  __shared__ int root_sorted[SLIC_MAXC]; // sorted by pixel index
  __shared__ int final_lab[SLIC_MAXC];
  __shared__ short nb_rank[SLIC_MAXC][4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int HW = H * W;
  const int* comp = comp_ws + (long)b * HW;
  int* aux = aux_ws + (long)b * HW;      // component sizes, then the final label of every root
  // roots in raster order: rank = number of roots with a smaller pixel index = a prefix count over the pixels (wave
  // ballots + a scan of the 16 wave totals per 1024 pixels).  (Round 2 collected the roots unsorted and ranked them by
  // counting, O(n^2): 0.6 ms per 128 images once noise images produce ~2000 components each.)
  __shared__ int wave_cnt[SLIC_THREADS / 64];
  if (tid == 0) n_roots = 0;
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  for (int p0 = 0; p0 < HW; p0 += SLIC_THREADS) {
    const int p = p0 + tid;
    const bool is_root = p < HW && comp[p] == p;
    const unsigned long long m = __ballot(is_root);
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int before = n_roots, total = 0;
    for (int w = 0; w < SLIC_THREADS / 64; ++w) {
      if (w < wave) before += wave_cnt[w];
      total += wave_cnt[w];
    }
    const int r = before + __popcll(m & ((1ull << lane) - 1ull));
    if (is_root && r < SLIC_MAXC) root_sorted[r] = p;
    __syncthreads();
    if (tid == 0) n_roots += total;
    __syncthreads();
  }
  if (n_roots > SLIC_MAXC) {             // more components than the tables hold: the cluster map is the answer
    if (tid == 0) n_regions[b] = -1;
    return;
  }
  const int n = n_roots;
  for (int r = tid; r < n; r += SLIC_THREADS) {
    const int root = root_sorted[r];
    root_px[r] = aux[root];                          // size (root_px is free after the sort)
    const int y = root / W, x = root - y * W;
    const int nb[4] = {x + 1 < W ? root + 1 : -1, x > 0 ? root - 1 : -1, y + 1 < H ? root + W : -1, y > 0 ? root - W : -1};
    for (int i = 0; i < 4; ++i) {
      int rk = -1;
      if (nb[i] >= 0) {
        const int ro = comp[nb[i]];
        if (ro < root) {                             // labelled earlier: its rank by binary search of the sorted roots
          int lo = 0, hi = n - 1;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (root_sorted[mid] < ro) lo = mid + 1; else hi = mid;
          }
          rk = lo;
        }
      }
      nb_rank[r][i] = (short)rk;
    }
  }
  __syncthreads();
  if (tid == 0) {
    int next = 0;
    for (int r = 0; r < n; ++r) {
      int fl;
      if (root_px[r] >= min_size) {
        fl = next++;
      } else {
        int adjacent = 0;
        for (int i = 0; i < 4; ++i)
          if (nb_rank[r][i] >= 0) adjacent = final_lab[nb_rank[r][i]];
        fl = adjacent;
      }
      final_lab[r] = fl;
    }
    n_regions[b] = next;
  }
  __syncthreads();
  for (int r = tid; r < n; r += SLIC_THREADS) aux[root_sorted[r]] = final_lab[r];
}

__global__ __launch_bounds__(256) void cc_output_kernel(const uint8_t* __restrict__ labels, const int* __restrict__ comp_ws,
                                                        const int* __restrict__ aux_ws, const int* __restrict__ n_regions,
                                                        long long* __restrict__ out, int HW) {
  const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const long o = (long)b * HW + p;
  out[o] = n_regions[b] < 0 ? (long long)labels[o] : (long long)aux_ws[(long)b * HW + comp_ws[o]];
}

}  // namespace

extern "C" int64_t favit_slic_features_workspace(int32_t B, int32_t H, int32_t W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return (int64_t)sizeof(float) * (2 * (int64_t)B + 3 * (int64_t)B * H * W);    // min / max per image + the blur pass
}

extern "C" int favit_slic_features(const float* img, int16_t* feat, int32_t B, int32_t H, int32_t W, float sigma,
                                   int32_t rescale, float* ws, void* stream) {
  if (!img || !feat || !ws || B <= 0 || H <= 0 || W <= 0 || sigma < 0.f) return FAVIT_ERR_INVALID;
  BlurTaps t;
  t.radius = sigma > 0.f ? (int)(4.0f * sigma + 0.5f) : 0;
  if (t.radius > 32) return FAVIT_ERR_UNSUPPORTED;
  {
    double wsum = 0.0;
    for (int k = -t.radius; k <= t.radius; ++k) wsum += exp(-0.5 * k * k / ((double)sigma * sigma + (sigma > 0.f ? 0.0 : 1.0)));
    for (int k = -t.radius; k <= t.radius; ++k)
      t.w[k + t.radius] = (float)(exp(-0.5 * k * k / ((double)sigma * sigma + (sigma > 0.f ? 0.0 : 1.0))) / wsum);
  }
  hipStream_t st = as_stream(stream);
  const long n = (long)B * H * W;
  float* minmax = rescale ? ws : nullptr;
  float* tmp = ws + 2 * (long)B;
  if (minmax) {
    hipLaunchKernelGGL(slic_minmax_kernel, dim3((unsigned)B), dim3(1024), 0, st, img, minmax, 3L * H * W);
    FAVIT_CHECK_LAUNCH();
  }
  const float* src = img;
  if (t.radius > 0) {
    hipLaunchKernelGGL(slic_blur_x_kernel, dim3((unsigned)((3 * n + 255) / 256)), dim3(256), 0, st, img, tmp, 3 * n, W, t);
    FAVIT_CHECK_LAUNCH();
    src = tmp;
  }
  hipLaunchKernelGGL(slic_features_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src,
                     reinterpret_cast<short*>(feat), B, H, W, t, (const float*)minmax);
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

extern "C" int64_t favit_slic_cluster_workspace(int32_t K, int32_t B) {
  if (K <= 0 || B <= 0) return 0;
  return (int64_t)B * (int64_t)slic_ws_stride(K);   // per image and centre: 6 int64 sums + 5 int32 coordinates
}

extern "C" int favit_slic_cluster(const int16_t* feat, uint8_t* labels, const int32_t* init_yx, int32_t K, int32_t B, int32_t H,
                                  int32_t W, int32_t step, int64_t coef, int32_t iters, void* ws, void* stream) {
  if (!feat || !labels || !init_yx || !ws || B <= 0 || H <= 0 || W <= 0 || K <= 0 || step <= 0 || coef < 0 || iters < 0)
    return FAVIT_ERR_INVALID;
  if (K > SLIC_MAXK || (long)H * W > (1L << 26)) return FAVIT_ERR_UNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(ws) & 7) return FAVIT_ERR_ALIGN;
  hipStream_t st = as_stream(stream);
  const long HW = (long)H * W;
  if (favit_zero_async(labels, (size_t)B * HW, st) != hipSuccess) return FAVIT_ERR_LAUNCH;
  hipLaunchKernelGGL(slic_seed_kernel, dim3((unsigned)B), dim3(64), 0, st, reinterpret_cast<const short*>(feat), init_yx, ws, K, H, W);
  FAVIT_CHECK_LAUNCH();
  const dim3 grid((unsigned)((HW + ASG_TILE - 1) / ASG_TILE), (unsigned)B);
#ifdef FAVIT_PROBE
  const int dbg = getenv("FAVIT_SLIC_DBG") ? atoi(getenv("FAVIT_SLIC_DBG")) : 0;
#define FAVIT_SLIC_DBG_ARG , dbg
#else
#define FAVIT_SLIC_DBG_ARG
#endif
  for (int it = 0; it < iters; ++it) {
    if (H <= 2047 && W <= 2047 && coef < (1LL << 32))
      hipLaunchKernelGGL(slic_assign_kernel<true>, grid, dim3(ASG_THREADS), (size_t)16 * (K * 6 + 1) * 4, st, reinterpret_cast<const short*>(feat), labels,
                         ws, K, H, W, step, (long long)coef FAVIT_SLIC_DBG_ARG);
    else
      hipLaunchKernelGGL(slic_assign_kernel<false>, grid, dim3(ASG_THREADS), (size_t)16 * (K * 6 + 1) * 4, st, reinterpret_cast<const short*>(feat), labels,
                         ws, K, H, W, step, (long long)coef FAVIT_SLIC_DBG_ARG);
    FAVIT_CHECK_LAUNCH();
  }
  return FAVIT_OK;
}

extern "C" int favit_slic_connect(const uint8_t* labels, int32_t* ws_comp, int32_t* ws_aux, int64_t* out, int32_t* n_regions,
                                  int32_t B, int32_t H, int32_t W, int32_t min_size, void* stream) {
  if (!labels || !ws_comp || !ws_aux || !out || !n_regions || B <= 0 || H <= 0 || W <= 0 || min_size < 0) return FAVIT_ERR_INVALID;
  if ((long)H * W > (1L << 30)) return FAVIT_ERR_UNSUPPORTED;
  hipStream_t st = as_stream(stream);
  const int HW = H * W;
  const dim3 grid((unsigned)((HW + 255) / 256), (unsigned)B);
  hipLaunchKernelGGL(cc_runs_kernel, dim3((unsigned)H, (unsigned)B), dim3(256), 0, st, labels, ws_comp, ws_aux, H, W);
  FAVIT_CHECK_LAUNCH();
  hipLaunchKernelGGL(cc_merge_kernel, grid, dim3(256), 0, st, labels, ws_comp, H, W);
  FAVIT_CHECK_LAUNCH();
  hipLaunchKernelGGL(cc_compress_kernel, grid, dim3(256), 0, st, ws_comp, ws_aux, HW);
  FAVIT_CHECK_LAUNCH();
  hipLaunchKernelGGL(cc_relabel_kernel, dim3((unsigned)B), dim3(SLIC_THREADS), 0, st, ws_comp, ws_aux, n_regions, H, W, min_size);
  FAVIT_CHECK_LAUNCH();
  hipLaunchKernelGGL(cc_output_kernel, grid, dim3(256), 0, st, labels, ws_comp, ws_aux, n_regions,
                     reinterpret_cast<long long*>(out), HW);
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}
