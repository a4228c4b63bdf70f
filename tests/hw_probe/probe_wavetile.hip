// probe: LDS-read + MFMA + barrier loop of a 256 x 256 x 32 stage (both operands T images, as in wgrad256p_kernel) with
//   A: 8 waves of 128 x 64 (two waves per SIMD, 12 fragments per 32 MFMAs)            - the shipped decomposition
//   B: 4 waves of 128 x 128 (one wave per SIMD, 16 fragments per 64 MFMAs, 256 accumulator registers)
// no global traffic inside the loop, random operands (DVFS: zeros would flatter).  Prints TFLOP/s of both on the whole chip.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include -I../../gan-class-transfer2_amd/csrc probe_wavetile.hip -o probe_wavetile
#include "gct2_common.h"
#include <cstdio>
#include <vector>

constexpr int IMG = 32 * 256;

__device__ __forceinline__ void fill(char* lds, int bytes, int tid, int nthreads, unsigned seed) {
  unsigned short* p = reinterpret_cast<unsigned short*>(lds);
  for (int i = tid; i < bytes / 2; i += nthreads) {
    unsigned h = (i + seed) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    // bf16 in [-1, 1): sign | exponent 0x7e or lower | 7 mantissa bits
    p[i] = (unsigned short)(((h & 1u) << 15) | ((0x7au + ((h >> 1) & 3u)) << 7) | ((h >> 8) & 0x7fu));
  }
}

__global__ __launch_bounds__(512, 2) void loop_a(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) char l0[4 * IMG], l1[4 * IMG], l2[4 * IMG], l3[4 * IMG];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 3, wm = wave >> 2;
  fill(l0, 4 * IMG, tid, 512, 1); fill(l1, 4 * IMG, tid, 512, 2); fill(l2, 4 * IMG, tid, 512, 3); fill(l3, 4 * IMG, tid, 512, 4);
  __syncthreads();
  f32x4_t acc[8][4];
  for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  auto compute = [&](const char* base) {
    const char* bimg = base + wm * IMG;
    const char* simg = base + (2 + (wn >> 1)) * IMG;
    int ql = lane;
    asm volatile("" : "+v"(ql));
    u32x4_t sf[4];
#pragma unroll
    for (int j = 0; j < 4; j++) sf[j] = timg_frag(simg, (wn & 1) * 64 + j * 16, 0, ql);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const u32x4_t bf = timg_frag(bimg, i * 16, 0, ql);
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = mfma16<__bf16>(sf[j], bf, acc[i][j]);
    }
  };
  const bool live = iters > 0;
  for (int it = 0; it < iters; it++) {
    if (live) compute(l0); __builtin_amdgcn_s_barrier();
    if (live) compute(l1); __builtin_amdgcn_s_barrier();
    if (live) compute(l2); __builtin_amdgcn_s_barrier();
    if (live) compute(l3); __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
  for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 512 + tid] = s;
}

__global__ __launch_bounds__(256, 1) void loop_b(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) char l0[4 * IMG], l1[4 * IMG], l2[4 * IMG], l3[4 * IMG];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wm = wave >> 1;
  fill(l0, 4 * IMG, tid, 256, 1); fill(l1, 4 * IMG, tid, 256, 2); fill(l2, 4 * IMG, tid, 256, 3); fill(l3, 4 * IMG, tid, 256, 4);
  __syncthreads();
  f32x4_t acc[8][8];
  for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  auto compute = [&](const char* base) {
    const char* bimg = base + wm * IMG;
    const char* simg = base + (2 + wn) * IMG;
    int ql = lane;
    asm volatile("" : "+v"(ql));
    u32x4_t sf[8];
#pragma unroll
    for (int j = 0; j < 8; j++) sf[j] = timg_frag(simg, j * 16, 0, ql);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const u32x4_t bf = timg_frag(bimg, i * 16, 0, ql);
#pragma unroll
      for (int j = 0; j < 8; j++) acc[i][j] = mfma16<__bf16>(sf[j], bf, acc[i][j]);
    }
  };
  const bool live = iters > 0;
  for (int it = 0; it < iters; it++) {
    if (live) compute(l0); __builtin_amdgcn_s_barrier();
    if (live) compute(l1); __builtin_amdgcn_s_barrier();
    if (live) compute(l2); __builtin_amdgcn_s_barrier();
    if (live) compute(l3); __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
  for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 256 + tid] = s;
}

int main() {
  const int blocks = 256, iters = 2000;                 // 8000 stages per work-group
  float* d; hipMalloc(&d, (size_t)blocks * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double flop = 2.0 * 256 * 256 * 32 * 4.0 * iters * blocks;
  for (int rep = 0; rep < 3; rep++) {
    float ms;
    hipEventRecord(e0); hipLaunchKernelGGL(loop_a, dim3(blocks), dim3(512), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("A  8 waves x 128x64 : %8.3f ms  %7.1f TFLOP/s\n", ms, flop / ms / 1e9);
    hipEventRecord(e0); hipLaunchKernelGGL(loop_b, dim3(blocks), dim3(256), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("B  4 waves x 128x128: %8.3f ms  %7.1f TFLOP/s\n", ms, flop / ms / 1e9);
  }
  if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
  return 0;
}
