// Hardware probe (test infrastructure): checks, with exact small-integer data, the gfx950 lane maps this
// repo's kernels rely on: mfma_f32_16x16x32_bf16 A/B/C fragments and ds_read_tr16_b64.
// Build: hipcc --offload-arch=gfx950 -O2 probe_mfma_tr.hip -o probe_mfma_tr ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

// A[16][32], B[32][16] (row-major, integer-valued) -> C[16][16]
__global__ void probe_mfma(const float* A, const float* B, float* C) {
  int l = threadIdx.x;
  bf16x8 fa, fb;
  for (int j = 0; j < 8; j++) {
    int k = 8 * (l >> 4) + j;
    fa[j] = (__bf16)A[(l & 15) * 32 + k];
    fb[j] = (__bf16)B[k * 16 + (l & 15)];
  }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c, 0, 0, 0);
  for (int j = 0; j < 4; j++) C[((l >> 4) * 4 + j) * 16 + (l & 15)] = c[j];
}

// T[32][16] (k rows, 16 cols) in LDS; each lane l = 16g+i should end with T[8g+j][i], j=0..7
__global__ void probe_tr(const float* T, float* out) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[32 * 16];
  int l = threadIdx.x;
  for (int i = l; i < 32 * 16; i += 64) lds[i] = (__bf16)T[i];
  __syncthreads();
  int g = l >> 4, q = (l & 15) >> 2, p = l & 3;
  bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lds + (8 * g + q) * 16 + 4 * p));
  bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lds + (8 * g + 4 + q) * 16 + 4 * p));
  for (int j = 0; j < 4; j++) { out[l * 8 + j] = (float)v0[j]; out[l * 8 + 4 + j] = (float)v1[j]; }
}

int main() {
  std::vector<float> A(16 * 32), B(32 * 16), C(256), Cr(256, 0.f), T(512), O(512);
  for (int i = 0; i < 16; i++) for (int k = 0; k < 32; k++) A[i * 32 + k] = (float)((i * 7 + k * 3) % 5 - 2);
  for (int k = 0; k < 32; k++) for (int j = 0; j < 16; j++) B[k * 16 + j] = (float)((k * 5 + j * 11) % 7 - 3);
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) for (int k = 0; k < 32; k++) Cr[i * 16 + j] += A[i * 32 + k] * B[k * 16 + j];
  for (int i = 0; i < 512; i++) T[i] = (float)(i % 251);
  float *dA, *dB, *dC, *dT, *dO;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 1024); hipMalloc(&dT, 2048); hipMalloc(&dO, 2048);
  hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
  hipMemcpy(dT, T.data(), 2048, hipMemcpyHostToDevice);
  probe_mfma<<<1, 64>>>(dA, dB, dC); probe_tr<<<1, 64>>>(dT, dO);
  hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost); hipMemcpy(O.data(), dO, 2048, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; i++) if (C[i] != Cr[i]) bad++;
  printf("mfma_16x16x32_bf16 layout mismatches: %d\n", bad);
  int bad2 = 0;
  for (int l = 0; l < 64; l++) for (int j = 0; j < 8; j++) if (O[l * 8 + j] != T[(8 * (l >> 4) + j) * 16 + (l & 15)]) bad2++;
  printf("ds_read_tr16_b64 layout mismatches: %d\n", bad2);
  if (bad2) { for (int l = 0; l < 20; l++) { printf("lane %d:", l); for (int j = 0; j < 8; j++) printf(" %g", O[l*8+j]); printf("\n"); } }
  return (bad || bad2) ? 1 : 0;
}
