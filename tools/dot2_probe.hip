// v_dot2_i32_i16 / v_pk_sub_i16 on gfx950 against scalar arithmetic (csrc/slic.hip uses them in the k-means distance)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef short s16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const int* a, const int* b, const int* c, int* o_dot, int* o_ref, int* o_sub) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const s16x2 x = __builtin_bit_cast(s16x2, a[i]), y = __builtin_bit_cast(s16x2, b[i]);
  const s16x2 d = x - y;
  o_sub[i] = __builtin_bit_cast(int, d);
  o_dot[i] = __builtin_amdgcn_sdot2(d, d, c[i], false);
  const int d0 = (int)x[0] - (int)y[0], d1 = (int)x[1] - (int)y[1];
  o_ref[i] = d0 * d0 + d1 * d1 + c[i];
}
int main() {
  const int n = 1 << 16;
  int *a, *b, *c, *od, *orf, *os;
  hipMallocManaged(&a, n * 4); hipMallocManaged(&b, n * 4); hipMallocManaged(&c, n * 4);
  hipMallocManaged(&od, n * 4); hipMallocManaged(&orf, n * 4); hipMallocManaged(&os, n * 4);
  srand(1);
  for (int i = 0; i < n; ++i) {
    short v[4];
    for (int j = 0; j < 4; ++j) v[j] = (short)((rand() % 16383) - 8191);
    a[i] = (unsigned short)v[0] | ((unsigned)(unsigned short)v[1] << 16);
    b[i] = (unsigned short)v[2] | ((unsigned)(unsigned short)v[3] << 16);
    c[i] = rand() % 1000000;
  }
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, a, b, c, od, orf, os);
  hipDeviceSynchronize();
  int bad = 0;
  for (int i = 0; i < n; ++i)
    if (od[i] != orf[i]) {
      if (bad < 5) printf("i=%d a=%08x b=%08x c=%d sub=%08x dot=%d ref=%d\n", i, a[i], b[i], c[i], os[i], od[i], orf[i]);
      ++bad;
    }
  printf("dot2 mismatches: %d of %d\n", bad, n);
  return bad != 0;
}
