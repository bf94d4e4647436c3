// Probe kernels for tools/overlap_probe.py (not part of libfavit).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int POLICY>
__global__ void fill_kernel(u32x4* dst, long n16, int wgs_limit) {
  const u32x4 v = {1u, 2u, 3u, 4u};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) {
    if (POLICY == 1) __builtin_nontemporal_store(v, dst + i);
    else dst[i] = v;
  }
}

__global__ void read_kernel(const u32x4* src, long n16, unsigned* out) {
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) {
    const u32x4 t = __builtin_nontemporal_load(src + i);
    acc += t;
  }
  if (acc.x + acc.y + acc.z + acc.w == 0x12345u) out[0] = 1;
}

extern "C" int probe_fill(void* dst, long bytes, int policy, int blocks, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (policy == 1) hipLaunchKernelGGL(fill_kernel<1>, dim3(blocks), dim3(256), 0, st, (u32x4*)dst, bytes / 16, 0);
  else hipLaunchKernelGGL(fill_kernel<0>, dim3(blocks), dim3(256), 0, st, (u32x4*)dst, bytes / 16, 0);
  return 0;
}
extern "C" int probe_read(const void* src, long bytes, void* out, int blocks, void* stream) {
  hipLaunchKernelGGL(read_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, bytes / 16, (unsigned*)out);
  return 0;
}

// ---- L2 -> LDS DMA probe: the operand traffic of a 256x128 GEMM tile without the math ----
typedef __attribute__((address_space(3))) void* lptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <int CHUNK, int NSTAGE>
__global__ __launch_bounds__(512) void dma_probe(const char* A, const char* W, long lda, long ldw, long M, int tiles_n,
                                                 int ksteps, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ROWS = 384, STAGE = ROWS * CHUNK, PIECES = STAGE / 1024, PPW = PIECES / 8;
  constexpr int LPR = CHUNK / 16, RPP = 64 / LPR;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const long m0 = (long)(tile / tiles_n) * 256, n0 = (long)(tile % tiles_n) * 128;
  const char* src[PPW];
  int dst[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int q = wave * PPW + j, row = q * RPP + lane / LPR, c = lane % LPR;
    long m = m0 + row; m = m < M ? m : M - 1;
    src[j] = row < 256 ? A + m * lda + c * 16 : W + (n0 + row - 256) * ldw + c * 16;
    dst[j] = q * 1024;
  }
  auto issue = [&](int buf) {
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      __builtin_amdgcn_global_load_lds((gptr_t)src[j], (lptr_t)(smem + buf * STAGE + dst[j]), 16, 0, 0);
      src[j] += CHUNK;
    }
  };
  unsigned acc = 0;
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s) if (s < ksteps) issue(s);
  int cur = 0;
  for (int kt = 0; kt < ksteps; ++kt) {
    if (kt + NSTAGE - 1 <= ksteps) {
      if (NSTAGE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (PPW * (NSTAGE - 2) == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if (PPW * (NSTAGE - 2) == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (PPW * (NSTAGE - 2) == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (kt + NSTAGE - 1 < ksteps) issue((cur + NSTAGE - 1) % NSTAGE);
    acc += *reinterpret_cast<const unsigned*>(smem + cur * STAGE + tid * 16);
    cur = (cur + 1) % NSTAGE;
  }
  if (acc == 0x13572468u) sink[0] = acc;
}

extern "C" int probe_dma(const void* A, const void* W, long lda, long ldw, long M, int tiles_n, int ntiles, int ksteps,
                         int chunk, int nstage, int lds_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  static unsigned* sink = nullptr;
  if (!sink) hipMalloc(&sink, 64);
#define LAUNCH(C, S)                                                                                          \
  do {                                                                                                        \
    hipFuncSetAttribute(reinterpret_cast<const void*>(dma_probe<C, S>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                        lds_bytes);                                                                           \
    hipLaunchKernelGGL((dma_probe<C, S>), dim3(ntiles), dim3(512), lds_bytes, st, (const char*)A, (const char*)W, \
                       lda, ldw, M, tiles_n, ksteps, sink);                                                   \
  } while (0)
  if (chunk == 64 && nstage == 3) LAUNCH(64, 3);
  else if (chunk == 64 && nstage == 4) LAUNCH(64, 4);
  else if (chunk == 128 && nstage == 2) LAUNCH(128, 2);
  else if (chunk == 128 && nstage == 3) LAUNCH(128, 3);
  else return -1;
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- do two workgroups on one CU share operand lines through the CU's L1? ----
// 512 persistent workgroups (two per CU).  Each reads its placement (XCC / SE / SH / CU ids), takes slot 0 or 1 of its CU
// from a table, and streams the operand traffic of `ntile` 256x128x(32*ksteps) GEMM tiles (16 KiB of A + 8 KiB of B per
// k-step, three stages, like the p4 kernel).  mode 0: the two workgroups of a CU stream DIFFERENT A tiles;
// mode 1: the SAME A tile (adjacent N tiles of one M panel).  census[key] counts workgroups per CU.
__global__ __launch_bounds__(512) void dma_share_probe(const char* A, const char* W, long a_tile_bytes, long w_tile_bytes,
                                                       int ntile, int ksteps, int mode, unsigned* census, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ unsigned s_key, s_slot;
  constexpr int STAGE = 384 * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));        // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));      // HW_REG_XCC_ID
    const unsigned key = ((xcc & 0xF) << 8) | ((hw >> 8) & 0xFF);         // cu_id[11:8] sh_id[12] se_id[15:13]
    s_key = key;
    s_slot = atomicAdd(census + key, 1u);
  }
  __syncthreads();
  const unsigned key = s_key, slot = s_slot;
  unsigned acc = 0;
  for (int t = 0; t < ntile; ++t) {
    long a_idx = (mode & 1) ? ((long)key * ntile + t) : (((long)key * 2 + (slot & 1)) * ntile + t);
    if (mode & 2) a_idx = (mode & 1) ? ((long)(key & 0xFF) + t) % 16 : ((long)(key & 0xFF) * 2 + (slot & 1) + t) % 16;   // L2-resident set
    const long w_idx = ((long)key * 2 + (slot & 1)) % 64;
    const char* a0 = A + a_idx * a_tile_bytes;
    const char* w0 = W + w_idx * w_tile_bytes;
    const char* src[3];
    int dst[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int q = wave * 3 + j, row = q * 16 + lane / 4, c = lane % 4;       // 24 pieces of 16 rows x 64 B
      src[j] = row < 256 ? a0 + (long)row * (ksteps * 64) + c * 16 : w0 + (long)(row - 256) * (ksteps * 64) + c * 16;
      dst[j] = q * 1024;
    }
    auto issue = [&](int buf) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        __builtin_amdgcn_global_load_lds((gptr_t)src[j], (lptr_t)(smem + buf * STAGE + dst[j]), 16, 0, 0);
        src[j] += 64;
      }
    };
    issue(0);
    if (ksteps > 1) issue(1);
    int cur = 0;
    for (int kt = 0; kt < ksteps; ++kt) {
      if (kt + 1 < ksteps) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + 2 < ksteps) issue(cur >= 1 ? cur - 1 : 2);
      acc += *reinterpret_cast<const unsigned*>(smem + cur * STAGE + tid * 16);
      cur = cur == 2 ? 0 : cur + 1;
    }
    __syncthreads();
  }
  if (acc == 0x13572468u) sink[0] = acc;
}

extern "C" int probe_dma_share(const void* A, const void* W, long a_tile_bytes, long w_tile_bytes, int ntile, int ksteps,
                               int mode, void* census, void* stream) {
  static unsigned* sink = nullptr;
  if (!sink) hipMalloc(&sink, 64);
  hipFuncSetAttribute(reinterpret_cast<const void*>(dma_share_probe), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 24576);
  hipLaunchKernelGGL(dma_share_probe, dim3(512), dim3(512), 3 * 24576, (hipStream_t)stream, (const char*)A, (const char*)W,
                     a_tile_bytes, w_tile_bytes, ntile, ksteps, mode, (unsigned*)census, sink);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- what the matrix cores deliver on this device under load: a bare v_mfma_f32_16x16x32_bf16 loop ----
// Every wave keeps `NACC` independent accumulators and issues MFMAs back to back on register operands (random bf16
// data, no memory traffic in the loop).  out[block] = {shader cycles, 100 MHz ticks} of wave 0 around the loop, so
// the host can report the clock the chip held (guide: "DVFS give-back" item 6).
typedef __attribute__((ext_vector_type(8))) __bf16 pbf16x8;
typedef __attribute__((ext_vector_type(4))) float pf32x4;

template <int NACC>
__global__ __launch_bounds__(256) void mfma_peak_kernel(const pbf16x8* __restrict__ seed, int iters, unsigned long long* out,
                                                        float* sink) {
  const int tid = threadIdx.x;
  pbf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = seed[(tid * 8 + i) & 4095]; b[i] = seed[(tid * 8 + 4 + i) & 4095]; }
  pf32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (pf32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) sink[0] = s;
  if (tid == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
}

extern "C" int probe_mfma_peak(const void* seed, int iters, int blocks, void* out, void* sink, void* stream) {
  hipLaunchKernelGGL(mfma_peak_kernel<16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const pbf16x8*)seed, iters,
                     (unsigned long long*)out, (float*)sink);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- the same for the exact-fp32 instruction of the parity mode: v_mfma_f32_32x32x2_f32, four 16-register accumulators ----
typedef __attribute__((ext_vector_type(16))) float pf32x16;
__global__ __launch_bounds__(256) void mfma_peak_f32_kernel(const float* __restrict__ seed, int iters, unsigned long long* out,
                                                            float* sink) {
  const int tid = threadIdx.x;
  float a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = seed[(tid * 8 + i) & 4095]; b[i] = seed[(tid * 8 + 4 + i) & 4095]; }
  pf32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s], a[s], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[(s + 1) & 3], a[s], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s], a[(s + 1) & 3], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[(s + 2) & 3], a[s], acc[3], 0, 0, 0);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += acc[i][r];
  if (sum == 12345.678f) sink[0] = sum;
  if (tid == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
}

extern "C" int probe_mfma_peak_f32(const void* seed, int iters, int blocks, void* out, void* sink, void* stream) {
  hipLaunchKernelGGL(mfma_peak_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)seed, iters,
                     (unsigned long long*)out, (float*)sink);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- and for v_mfma_f32_16x16x4_f32 (the same 64 flops per cycle and SIMD in 8-pass instructions, sixteen 4-register accumulators) ----
__global__ __launch_bounds__(256) void mfma_peak_f32s_kernel(const float* __restrict__ seed, int iters, unsigned long long* out,
                                                             float* sink) {
  const int tid = threadIdx.x;
  float a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = seed[(tid * 8 + i) & 4095]; b[i] = seed[(tid * 8 + 4 + i) & 4095]; }
  pf32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (pf32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[i & 3], a[i >> 2], acc[i], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (sum == 12345.678f) sink[0] = sum;
  if (tid == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
}

extern "C" int probe_mfma_peak_f32s(const void* seed, int iters, int blocks, void* out, void* sink, void* stream) {
  hipLaunchKernelGGL(mfma_peak_f32s_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)seed, iters,
                     (unsigned long long*)out, (float*)sink);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
