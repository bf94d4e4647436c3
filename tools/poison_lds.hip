// LDS poisoning for the uninitialised-read hunt (tools/poison.py): fills the whole 160 KiB of LDS of every CU with a
// NaN bit pattern (0x7FC07FC0: a NaN as fp32 and as two bf16), so that the next kernel on the stream starts on LDS
// whose every byte is poison.  A kernel that consumes an LDS row it never staged (halo rows, ragged tiles, dump
// slots multiplied by a zero weight: NaN * 0 = NaN) then fails deterministically instead of depending on what the
// previous kernel of some other process left on that CU.
//
// One workgroup claims all 160 KiB, so at most one is resident per CU; every workgroup holds its CU for ~8 us after
// filling, so the 256 workgroups of the first round land on 256 different CUs (the stream is idle when the poison
// kernel starts: it is launched in stream order in front of every libfavit call); a second round repeats it.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/poison_lds.hip -o tools/libpoison.so
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
constexpr int LDS_BYTES = 160 * 1024;

__global__ __launch_bounds__(1024) void poison_lds_kernel(unsigned pattern, unsigned hold_ticks, unsigned* census) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  for (int i = threadIdx.x; i < LDS_BYTES / 4; i += 1024) lds[i] = pattern;
  __syncthreads();
  if (census && threadIdx.x == 0) {
    // HW_REG_HW_ID (id 4) / XCC_ID (id 20): which CU this workgroup ran on (tools/poison.py --census)
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));
    census[blockIdx.x] = ((xcc & 0xF) << 16) | ((hw >> 8) & 0xFF);    // xcc | se_id, sh_id, cu_id bits (HW_ID[15:8])
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
  while (__builtin_amdgcn_s_memrealtime() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(16);
  // read back one word so that the fill cannot be dropped
  if (lds[(threadIdx.x * 37) % (LDS_BYTES / 4)] != pattern) __builtin_trap();
}
}  // namespace

extern "C" int favit_tool_poison_lds(void* stream, unsigned pattern, void* census) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(poison_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            LDS_BYTES) != hipSuccess)
      return -1;
    attr_set = true;
  }
  hipLaunchKernelGGL(poison_lds_kernel, dim3(512), dim3(1024), LDS_BYTES, reinterpret_cast<hipStream_t>(stream), pattern,
                     800u, reinterpret_cast<unsigned*>(census));
  return hipGetLastError() == hipSuccess ? 0 : -4;
}
