// probe: LDS-read + MFMA + barrier loop of a 256 x 256 x 32 stage (both operands T images, as in wgrad256p_kernel) with
//   A: 8 waves of 128 x 64 (two waves per SIMD, 12 fragments per 32 MFMAs)            - the shipped decomposition
//   B: 4 waves of 128 x 128 (one wave per SIMD, 16 fragments per 64 MFMAs, 256 accumulator registers)
// no global traffic inside the loop, random operands (DVFS: zeros would flatter).  Prints TFLOP/s of both on the whole chip.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-value -Wno-inline-asm -I../../include -I../../gan-class-transfer2_amd/csrc probe_wavetile.hip -o probe_wavetile
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

// C: loop A plus the staging of wgrad256p_kernel (4 LDS-DMA instructions per wave and stage, stage s+3 issued while stage s is
// multiplied, counted vmcnt, raw barrier) from a 32-KiB-per-work-group source that stays in L2: the cost of the DMA MECHANISM
// (issue slots, LDS write port) without memory latency.  D: the same with a 64-MiB-per-launch source walked linearly (HBM / MALL).
typedef __attribute__((address_space(3))) void lds_void_t;
__device__ __forceinline__ void dma16_hidden(__amdgpu_buffer_rsrc_t rsrc, char* lds_piece, unsigned voff) {
  const unsigned lds_addr = (unsigned)(uintptr_t)(lds_void_t*)lds_piece;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc) : "memory", "m0");
}
#define VMCNT_ONLY(n) ((((n) & 0xF) | 0x70 | 0xF00 | ((((n) >> 4) & 3) << 14)))
template <bool STREAM>
__global__ __launch_bounds__(512, 2) void loop_c(float* out, const char* src, int iters) {
  __shared__ __attribute__((aligned(16))) char l0[4 * IMG], l1[4 * IMG], l2[4 * IMG], l3[4 * IMG];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 3, wm = wave >> 2;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)0x80000000u, 0x00020000);
  unsigned base = STREAM ? (unsigned)blockIdx.x * (unsigned)(iters * 4) * 32768u % 0x7f000000u : (unsigned)blockIdx.x * 32768u;
  const unsigned lane_off = (unsigned)(wave * 1024 + lane * 16);
  unsigned stage_no = 0;
  auto issue = [&](char* tgt) {
    const unsigned o = base + (STREAM ? stage_no * 32768u : 0u) + lane_off;
#pragma unroll
    for (int g = 0; g < 4; g++) dma16_hidden(rs, tgt + g * IMG + wave * 1024, o + g * 8192u);
    stage_no++;
  };
  f32x4_t acc[8][4];
  for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  auto compute = [&](const char* b2) {
    const char* bimg = b2 + wm * IMG;
    const char* simg = b2 + (2 + (wn >> 1)) * IMG;
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
  issue(l0); issue(l1); issue(l2);
  __builtin_amdgcn_s_waitcnt(VMCNT_ONLY(8));
  __builtin_amdgcn_s_barrier();
  auto stage = [&](const char* cur, char* tgt) {
    issue(tgt);
    if (live) compute(cur);
    __builtin_amdgcn_s_waitcnt(VMCNT_ONLY(8));
    __builtin_amdgcn_s_barrier();
  };
  for (int it = 0; it < iters; it++) { stage(l0, l3); stage(l1, l0); stage(l2, l1); stage(l3, l2); }
  __builtin_amdgcn_s_waitcnt(VMCNT_ONLY(0));
  float s = 0.f;
  for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 512 + tid] = s;
}

__global__ void fill_src(unsigned short* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)(i * 2654435761ull); h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (unsigned short)(((h & 1u) << 15) | ((0x7au + ((h >> 1) & 3u)) << 7) | ((h >> 8) & 0x7fu));
  }
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
  {
    const int it2 = 64;                                   // 256 stages per work-group: 8 MiB streamed by each, 2 GiB per launch
    char* src; hipMalloc(&src, (size_t)0x7f000000u + (1u << 20)); hipLaunchKernelGGL(fill_src, dim3(4096), dim3(256), 0, 0, (unsigned short*)src, ((size_t)0x7f000000u + (1u << 20)) / 2); hipDeviceSynchronize();
    const double flop2 = 2.0 * 256 * 256 * 32 * 4.0 * it2 * blocks;
    for (int rep = 0; rep < 3; rep++) {
      float ms;
      hipEventRecord(e0); hipLaunchKernelGGL(loop_a, dim3(blocks), dim3(512), 0, 0, d, it2); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      printf("A  (short run)              : %8.3f ms  %7.1f TFLOP/s\n", ms, flop2 / ms / 1e9);
      hipEventRecord(e0); hipLaunchKernelGGL(loop_c<false>, dim3(blocks), dim3(512), 0, 0, d, src, it2); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      printf("C  A + LDS-DMA, L2-resident : %8.3f ms  %7.1f TFLOP/s\n", ms, flop2 / ms / 1e9);
      hipEventRecord(e0); hipLaunchKernelGGL(loop_c<true>, dim3(blocks), dim3(512), 0, 0, d, src, it2); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      printf("D  A + LDS-DMA, streamed    : %8.3f ms  %7.1f TFLOP/s  (%.2f TB/s)\n", ms, flop2 / ms / 1e9, 32768.0 * 4 * it2 * blocks / ms / 1e9);
    }
  }
  if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
  return 0;
}
