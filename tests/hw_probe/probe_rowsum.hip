// probe: row16_sum (DPP) and rows4_sum (v_permlane16/32_swap) of gct2_common.h against the __shfl_xor butterflies they replace
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-value -Wno-inline-asm -I../../include -I../../gan-class-transfer2_amd/csrc probe_rowsum.hip -o probe_rowsum
#include "gct2_common.h"
#include <cstdio>
typedef unsigned u2v __attribute__((ext_vector_type(2)));
__global__ void k32(unsigned* out) {
  const unsigned lane = threadIdx.x;
  const unsigned a = 100 + lane, b = 200 + lane;
  const u2v r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[lane] = r[0]; out[64 + lane] = r[1];
  unsigned xa = a, xb = b;
  asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(xa), "+v"(xb));
  out[128 + lane] = xa; out[192 + lane] = xb;
}
__global__ void k(const float* in, float* out) {
  const int l = threadIdx.x;
  float t = in[l];
  float a = t;
  a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64); a += __shfl_xor(a, 8, 64);
  float b = t;
  b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
  out[l] = a; out[64 + l] = row16_sum(t); out[128 + l] = b; out[192 + l] = rows4_sum(t);
}
int main() {
  float h[64], o[256], *di, *dout;
  for (int i = 0; i < 64; i++) h[i] = 1.0f + 0.37f * i + 0.001f * i * i;
  hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  {
    unsigned hu[256], *du; hipMalloc(&du, sizeof(hu));
    hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, du);
    hipMemcpy(hu, du, sizeof(hu), hipMemcpyDeviceToHost);
    printf("permlane32_swap builtin: r0 lanes 0,31,32,63 = %u %u %u %u   r1 = %u %u %u %u\n", hu[0], hu[31], hu[32], hu[63], hu[64], hu[95], hu[96], hu[127]);
    printf("permlane32_swap asm    : a' lanes 0,31,32,63 = %u %u %u %u   b' = %u %u %u %u\n", hu[128], hu[159], hu[160], hu[191], hu[192], hu[223], hu[224], hu[255]);
  }
  int bad16 = 0, bad4 = 0;
  for (int i = 0; i < 64; i++) { bad16 += o[i] != o[64 + i]; bad4 += o[128 + i] != o[192 + i]; }
  printf("row16_sum: %d of 64 lanes differ from the xor butterfly (lane 0: %g vs %g, lane 17: %g vs %g)\n", bad16, o[0], o[64], o[17], o[64 + 17]);
  printf("rows4_sum: %d of 64 lanes differ (lane 0: %g vs %g, lane 40: %g vs %g)\n", bad4, o[128], o[192], o[128 + 40], o[192 + 40]);
  return bad16 + bad4 ? 1 : 0;
}
