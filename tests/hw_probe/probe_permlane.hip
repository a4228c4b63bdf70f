// probe: semantics of v_permlane16_swap_b32 as exposed by __builtin_amdgcn_permlane16_swap (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
  const unsigned lane = threadIdx.x;
  const unsigned a = 100 + lane, b = 200 + lane;      // a = first operand, b = second operand
  const u2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[lane] = r[0]; out[64 + lane] = r[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 128 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int row = 0; row < 4; row++) printf("row %d: r0 = %u..%u   r1 = %u..%u\n", row, h[16 * row], h[16 * row + 15], h[64 + 16 * row], h[64 + 16 * row + 15]);
  return 0;
}
