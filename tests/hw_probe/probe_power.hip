// probe: what bounds an LDS-fed bf16 MFMA loop on this chip - structure or power?  Four loops x {random, zero} operands, each run
// back to back for >= 1.5 s and then once more with the in-kernel clock stamped (s_memtime / s_memrealtime around the loop,
// MI355X_MICROARCH.md "DVFS give-back" item 6):
//   R  registers only: 2 waves per SIMD, 32 independent v_mfma_f32_16x16x32_bf16 per iteration, operands fixed in registers
//   A  the wgrad256q stage without global traffic: 8 waves of 128 x 64, 24 transposed LDS reads per 32 MFMAs, raw barrier per stage
//   P  A with the reads of the next stage's fragments issued under this stage's multiplies (second fragment set; no barrier needed
//      because nothing is written): the best case of any re-timing of A
//   N  A without the barrier
// Prints PFLOP/s on the whole chip (256 work-groups), the in-kernel clock, and MFMA-pipe utilisation = 16 cycles x MFMAs / loop cycles.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-value -Wno-inline-asm -I../../include -I../../gan-class-transfer2_amd/csrc probe_power.hip -o probe_power
#include "gct2_common.h"
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int IMG = 32 * 256;

__device__ __forceinline__ unsigned short rnd_bf16(unsigned i, unsigned seed, int zero) {
  unsigned h = (i + seed) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  return zero ? 0 : (unsigned short)(((h & 1u) << 15) | ((0x7au + ((h >> 1) & 3u)) << 7) | ((h >> 8) & 0x7fu));
}
__device__ __forceinline__ void fill(char* lds, int bytes, int tid, int nthreads, unsigned seed, int zero) {
  unsigned short* p = reinterpret_cast<unsigned short*>(lds);
  for (int i = tid; i < bytes / 2; i += nthreads) p[i] = rnd_bf16(i, seed, zero);
}
struct Stamp { unsigned long long t0, r0, t1, r1; };
#define STAMP_BEGIN const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime()
#define STAMP_END(buf) do { const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
    if ((tid & 63) == 0) buf[blockIdx.x * 8 + (tid >> 6)] = Stamp{t0, r0, t1, r1}; } while (0)

__global__ __launch_bounds__(512, 2) void loop_r(float* out, Stamp* st, int iters, int zero) {
  const int tid = threadIdx.x;
  u32x4_t sf[4], bf[8];
  for (int j = 0; j < 4; j++) for (int e = 0; e < 4; e++)
    sf[j][e] = rnd_bf16(tid * 64 + j * 8 + e * 2, 7, zero) | ((unsigned)rnd_bf16(tid * 64 + j * 8 + e * 2 + 1, 7, zero) << 16);
  for (int i = 0; i < 8; i++) for (int e = 0; e < 4; e++)
    bf[i][e] = rnd_bf16(tid * 64 + i * 8 + e * 2, 9, zero) | ((unsigned)rnd_bf16(tid * 64 + i * 8 + e * 2 + 1, 9, zero) << 16);
  f32x4_t acc[8][4];
  for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  STAMP_BEGIN;
  for (int it = 0; it < 4 * iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      asm volatile("" : "+v"(bf[i]));
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = mfma16<__bf16>(sf[j], bf[i], acc[i][j]);
    }
  }
  STAMP_END(st);
  float s = 0.f;
  for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 512 + tid] = s;
}

template <int MODE>      // 0: A (barrier per stage)   1: N (no barrier)   2: P (next stage's fragments prefetched, no barrier)
__global__ __launch_bounds__(512, 2) void loop_a(float* out, Stamp* st, int iters, int zero) {
  __shared__ __attribute__((aligned(16))) char l0[4 * IMG], l1[4 * IMG], l2[4 * IMG], l3[4 * IMG];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 3, wm = wave >> 2;
  fill(l0, 4 * IMG, tid, 512, 1, zero); fill(l1, 4 * IMG, tid, 512, 2, zero); fill(l2, 4 * IMG, tid, 512, 3, zero); fill(l3, 4 * IMG, tid, 512, 4, zero);
  __syncthreads();
  f32x4_t acc[8][4];
  for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  int ql = lane;
  asm volatile("" : "+v"(ql));
  auto compute = [&](const char* base) {
    const char* bimg = base + wm * IMG;
    const char* simg = base + (2 + (wn >> 1)) * IMG;
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
  // P: all 12 fragments of a stage live in registers; while stage s is multiplied, stage s+1's are read into the other set
  u32x4_t fa[12], fb[12];
  auto load = [&](u32x4_t* f, const char* base) {
    const char* bimg = base + wm * IMG;
    const char* simg = base + (2 + (wn >> 1)) * IMG;
#pragma unroll
    for (int j = 0; j < 4; j++) f[j] = timg_frag(simg, (wn & 1) * 64 + j * 16, 0, ql);
#pragma unroll
    for (int i = 0; i < 8; i++) f[4 + i] = timg_frag(bimg, i * 16, 0, ql);
  };
  auto mul_load = [&](const u32x4_t* f, u32x4_t* nf, const char* nbase) {
    const char* bimg = nbase + wm * IMG;
    const char* simg = nbase + (2 + (wn >> 1)) * IMG;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (i < 4) nf[i] = timg_frag(simg, (wn & 1) * 64 + i * 16, 0, ql);
      nf[4 + i] = timg_frag(bimg, i * 16, 0, ql);
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = mfma16<__bf16>(f[j], f[4 + i], acc[i][j]);
    }
  };
  const bool live = iters > 0;
  STAMP_BEGIN;
  if (MODE == 2) {
    load(fa, l0);
    for (int it = 0; it < iters; it++) {
      if (live) mul_load(fa, fb, l1);
      if (live) mul_load(fb, fa, l2);
      if (live) mul_load(fa, fb, l3);
      if (live) mul_load(fb, fa, l0);
    }
  } else {
    for (int it = 0; it < iters; it++) {
      if (live) compute(l0); if (MODE == 0) __builtin_amdgcn_s_barrier();
      if (live) compute(l1); if (MODE == 0) __builtin_amdgcn_s_barrier();
      if (live) compute(l2); if (MODE == 0) __builtin_amdgcn_s_barrier();
      if (live) compute(l3); if (MODE == 0) __builtin_amdgcn_s_barrier();
    }
  }
  STAMP_END(st);
  float s = 0.f;
  for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (MODE == 2) s += __builtin_bit_cast(float, fa[0][0]) * 1e-30f;
  out[blockIdx.x * 512 + tid] = s;
}

template <typename F>
static void run(const char* name, F launch, Stamp* dst, int iters, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double flop = 2.0 * 256 * 256 * 32 * 4.0 * iters * blocks;
  float ms = 0.f, total = 0.f;
  while (total < 1500.f) {                         // settle the clock
    hipEventRecord(e0); for (int k = 0; k < 20; k++) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); total += ms;
  }
  hipEventRecord(e0); for (int k = 0; k < 20; k++) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1); ms /= 20;
  std::vector<Stamp> h(blocks * 8);
  hipMemcpy(h.data(), dst, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
  std::vector<double> clk, cyc;
  for (auto& s : h) if (s.r1 > s.r0) { clk.push_back((double)(s.t1 - s.t0) / (double)(s.r1 - s.r0) * 0.1); cyc.push_back((double)(s.t1 - s.t0)); }
  std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
  const double c = clk.empty() ? 0 : clk[clk.size() / 2], cy = cyc.empty() ? 0 : cyc[cyc.size() / 2];
  const double stages = 4.0 * iters;
  printf("%-44s %8.3f ms  %6.3f PFLOP/s  clock %5.3f GHz  %7.1f cycles per stage  MFMA pipe %5.3f\n", name, ms, flop / ms / 1e12, c, cy / stages,
         stages * 1024.0 / cy);
}

int main() {
  const int blocks = 256, iters = 500;             // 2000 stages per work-group: ~1.5 ms per launch
  float* d; hipMalloc(&d, (size_t)blocks * 512 * 4);
  Stamp* st; hipMalloc(&st, (size_t)blocks * 8 * sizeof(Stamp));
  for (int zero = 0; zero < 2; zero++) {
    const char* z = zero ? "zeros " : "random";
    char nm[96];
    snprintf(nm, sizeof nm, "R registers only, %s", z);
    run(nm, [&] { hipLaunchKernelGGL(loop_r, dim3(blocks), dim3(512), 0, 0, d, st, iters, zero); }, st, iters, blocks);
    snprintf(nm, sizeof nm, "A LDS reads + barrier per stage, %s", z);
    run(nm, [&] { hipLaunchKernelGGL(loop_a<0>, dim3(blocks), dim3(512), 0, 0, d, st, iters, zero); }, st, iters, blocks);
    snprintf(nm, sizeof nm, "N LDS reads, no barrier, %s", z);
    run(nm, [&] { hipLaunchKernelGGL(loop_a<1>, dim3(blocks), dim3(512), 0, 0, d, st, iters, zero); }, st, iters, blocks);
    snprintf(nm, sizeof nm, "P next stage's reads under the MFMAs, %s", z);
    run(nm, [&] { hipLaunchKernelGGL(loop_a<2>, dim3(blocks), dim3(512), 0, 0, d, st, iters, zero); }, st, iters, blocks);
  }
  if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
  return 0;
}
