// Shared device/host helpers for libfavit (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <utility>
#include <vector>
#include "favit.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define FAVIT_WAVE 64

#define FAVIT_CHECK_LAUNCH()                         \
  do {                                               \
    hipError_t e__ = hipGetLastError();              \
    if (e__ != hipSuccess) return FAVIT_ERR_LAUNCH;  \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of a (kernel, device) pair (any number of devices per
// process, any number of host threads).  Callers pass run-time sizes (sdpa: head dim; mhla backward: L / rows per
// block), so the LARGEST size set so far is remembered per pair and the attribute is raised when a later launch of
// the same instantiation asks for more (a stale smaller limit makes that launch fail, in call-order-dependent ways).
struct FavitDynLds { const void* kernel; int dev; int bytes; };
static inline void favit_ensure_dyn_lds(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::vector<FavitDynLds> done;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  for (auto& e : done)
    if (e.kernel == kernel && e.dev == dev) {
      if (bytes > e.bytes) {
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        e.bytes = bytes;
      }
      return;
    }
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  done.push_back({kernel, dev, bytes});
}

// Zero-fill as a KERNEL, never hipMemsetAsync.  A hipMemsetAsync captured into a HIP graph (train.GraphedStep) came
// back wrong from the SECOND replay on (ROCm 7.2, gfx950): dword 2 of every 16 bytes of the destination held one
// arbitrary constant (3e19, -6e20, 1e25: another value in every process) instead of zero, while the eager call and the
// first replay were right (tools/graph_memset_probe.py shows it without any kernel of this library).  The cls_token /
// pos_embed gradients (atomics on top of that "zero") therefore overflowed AdamW's second moment and silently froze
// both parameters in every graph-replayed step -- and when the constant happened to be a NaN pattern the loss went NaN
// (the two unexplained NaN losses of round 3).  Found with tools/poison.py; pinned by
// tests/test_gpu_kernels.py::test_zero_fills_survive_graph_replays.
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void favit_zero_kernel(unsigned char* __restrict__ p, long bytes) {
  // p + head is 16-byte aligned; body in 16-byte stores, head / tail bytewise
  const long head = (16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15;
  const long h = head < bytes ? head : bytes;
  const long n16 = (bytes - h) >> 4;
  uint4* b = reinterpret_cast<uint4*>(p + h);
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long)gridDim.x * blockDim.x;
  for (long i = tid; i < n16; i += nth) b[i] = make_uint4(0u, 0u, 0u, 0u);
  const long tail0 = h + (n16 << 4);
  if (tid < h) p[tid] = 0;
  if (tid < bytes - tail0) p[tail0 + tid] = 0;
}
static inline hipError_t favit_zero_async(void* ptr, size_t bytes, hipStream_t st) {
  if (!ptr || bytes == 0) return hipSuccess;
  long blocks = (long)((bytes / 16 + 255) / 256);
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(favit_zero_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<unsigned char*>(ptr), (long)bytes);
  return hipGetLastError();
}

template <typename T> struct dtype_of;
template <> struct dtype_of<float> { static constexpr int value = FAVIT_F32; };
template <> struct dtype_of<bf16_t> { static constexpr int value = FAVIT_BF16; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// exact-erf GELU and its derivative (nn.GELU() default, models/vit.py:120)
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float dgelu_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// Fast GELU for the bf16 path: erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below bf16
// resolution), one v_exp + one v_rcp instead of libm erff.  phi/Phi share the exponential.
__device__ __forceinline__ void gelu_terms_fast(float x, float& cdf, float& pdf) {
  const float ax = fabsf(x);
  const float e = __expf(-0.5f * x * x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float erfa = 1.0f - poly * t * e;            // erf(|x|/sqrt2)
  cdf = 0.5f * (1.0f + copysignf(erfa, x));
  pdf = 0.39894228040143267794f * e;
}
__device__ __forceinline__ float gelu_fast(float x) {
  float c, p;
  gelu_terms_fast(x, c, p);
  return x * c;
}
__device__ __forceinline__ float dgelu_fast(float x) {
  float c, p;
  gelu_terms_fast(x, c, p);
  return fmaf(x, p, c);
}

// Counter-based RNG for dropout: the same draw is recomputed in backward, so no mask is stored.  32-bit arithmetic
// throughout (the splitmix64 finaliser used before cost ~28 VALU operations per element -- two 64-bit multiplies -- and
// made every dropout epilogue VALU-bound): an affine map of the index whose odd multiplier and offset come from the seed
// (masks of different seeds are not shifted copies of each other), then a 32-bit multiply-xorshift finaliser (the
// "lowbias32" constants): for a fixed seed the draws of 2^32 consecutive indices are a permutation of the 32-bit values.
// Round 4: ONE 32-bit draw serves the TWO elements 2i and 2i + 1 (its low / high 16 bits against a 16-bit threshold, so
// p is realised to 1 / 65536): the quarter-rate multiplies of the draw were the bulk of a dropout epilogue's VALU work
// (fc1 forward at cfg2: +23 us per launch with dropout 0.1, dH +19 us; now +8 / +10), and the vector epilogues walk pairs.
// tests/test_gpu_kernels.py checks keep rates and the absence of correlation across lags, strides and seeds.
__device__ __forceinline__ uint32_t favit_mix_u32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t favit_rand_u32(uint64_t seed, uint64_t idx) {
  const uint32_t k1 = ((uint32_t)seed * 0x9E3779B1u) | 1u, k2 = (uint32_t)(seed >> 32) * 0x85EBCA77u;
  return favit_mix_u32((uint32_t)idx * k1 + k2 + (uint32_t)(idx >> 32) * 0xC2B2AE3Du);
}
// Dropout epoch (favit_set_dropout_epoch, include/favit.h): a device word every dropout-drawing kernel mixes into its
// by-value seed, so that a step captured ONCE in a HIP graph (seeds frozen into the kernel arguments) still draws
// fresh masks at every replay -- the graph's first node increments the word.  NULL (the default): seeds as given.
// Forward and backward kernels of one replay read the same value, so recomputed masks stay consistent.
extern "C" const unsigned long long* favit_dropout_epoch_ptr_(void);
__device__ __forceinline__ uint64_t favit_eff_seed(uint64_t seed, const unsigned long long* epoch) {
  return epoch ? seed + (uint64_t)(*epoch) * 0x9E3779B97F4A7C15ull : seed;
}
// keep with probability 1-p: element idx takes the low (even idx) or high (odd idx) half of the draw of idx >> 1
__device__ __forceinline__ bool favit_keep(uint64_t seed, uint64_t idx, uint32_t thresh) {
  const uint32_t h = favit_rand_u32(seed, idx >> 1);
  return ((idx & 1) ? (h >> 16) : (h & 0xffffu)) >= (thresh >> 16);
}
// the same for the elements idx_even and idx_even + 1 (idx_even is even): one draw
__device__ __forceinline__ void favit_keep2(uint64_t seed, uint64_t idx_even, uint32_t thresh, bool& k0, bool& k1) {
  const uint32_t h = favit_rand_u32(seed, idx_even >> 1), t = thresh >> 16;
  k0 = (h & 0xffffu) >= t;
  k1 = (h >> 16) >= t;
}
static inline uint32_t dropout_threshold(float p) {
  double t = (double)p * 4294967296.0;
  if (t <= 0) return 0u;
  if (t >= 4294967295.0) return 4294967295u;
  return (uint32_t)t;
}

// sum over the 8 consecutive lanes of an aligned group, entirely in the VALU (DPP): quad_perm
// [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror -- no ds_bpermute round trips through LDS.
__device__ __forceinline__ float dpp_sum8(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  return v;
}

// Exchange with lane ^ 16 / lane ^ 32 in the VALU (gfx950 v_permlane16_swap / v_permlane32_swap: the odd rows of
// the first operand are swapped with the even rows of the second).  __shfl_xor(v, 16) compiles to ds_bpermute_b32,
// a round trip through the LDS crossbar (~100 cycles of dependent latency per reduction step in the softmax).
__device__ __forceinline__ float lane_xor16(float v) {
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
  const unsigned u = __float_as_uint(v);
  const u32x2 r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(((threadIdx.x >> 4) & 1) ? r.x : r.y);
}
__device__ __forceinline__ float lane_xor32(float v) {
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
  const unsigned u = __float_as_uint(v);
  const u32x2 r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(((threadIdx.x >> 5) & 1) ? r.x : r.y);
}
// reductions over the four lanes {l, l^16, l^32, l^48} (the 4 k-slices of one MFMA column)
__device__ __forceinline__ float quad16_max(float v) { v = fmaxf(v, lane_xor16(v)); return fmaxf(v, lane_xor32(v)); }
__device__ __forceinline__ float quad16_sum(float v) { v += lane_xor16(v); return v + lane_xor32(v); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
