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


// ---- the same questions for v_mfma_f32_32x32x16_bf16 (half the matrix instructions per FLOP, 24 instead of 8 free issue cycles per
// instruction): R32 registers only; A32 / P32 the 256 x 256 x 32 stage with 8 waves of 128 x 64 (WAVES = 8) or 4 waves of 128 x 128
// (WAVES = 4, one wave per SIMD, 256 accumulator registers).  Fragment of 32 columns x 16 rows: lane l reads rows
// 16 kk + 8 (l >> 5) + ((l >> 2) & 3) (+ 4), 32-byte chunk 2 f + ((l >> 4) & 1), stored at chunk ^ 2 (row & 3): conflict-free for both
// 32-lane halves of ds_read_b64_tr_b16 (4 rows x 2 chunks -> 8 distinct 32-byte bank groups).
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
__device__ __forceinline__ f32x16_t mfma32(u32x4_t a, u32x4_t b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ u32x4_t frag32(const char* img, int f, int kk, int lane) {
  const int k0 = 16 * kk + 8 * (lane >> 5) + ((lane >> 2) & 3);
  const int chunk = 2 * f + ((lane >> 4) & 1);
  const int o0 = k0 * 256 + ((chunk ^ (2 * (k0 & 3))) << 5) + (lane & 3) * 8;       // row k0 + 4 has the same swizzle
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(img + o0));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(img + o0 + 4 * 256));
  const u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
  return u32x4_t{l2[0], l2[1], h2[0], h2[1]};
}

__global__ __launch_bounds__(512, 2) void loop_r32(float* out, Stamp* st, int iters, int zero) {
  const int tid = threadIdx.x;
  u32x4_t sf[2], bf[4];
  for (int j = 0; j < 2; j++) for (int e = 0; e < 4; e++)
    sf[j][e] = rnd_bf16(tid * 64 + j * 8 + e * 2, 7, zero) | ((unsigned)rnd_bf16(tid * 64 + j * 8 + e * 2 + 1, 7, zero) << 16);
  for (int i = 0; i < 4; i++) for (int e = 0; e < 4; e++)
    bf[i][e] = rnd_bf16(tid * 64 + i * 8 + e * 2, 9, zero) | ((unsigned)rnd_bf16(tid * 64 + i * 8 + e * 2 + 1, 9, zero) << 16);
  f32x16_t acc[4][2];
  for (int i = 0; i < 4; i++) for (int j = 0; j < 2; j++) for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
  STAMP_BEGIN;
  for (int it = 0; it < 8 * iters; it++) {            // 8 MFMAs of 32 cycles per trip = half a stage
#pragma unroll
    for (int i = 0; i < 4; i++) {
      asm volatile("" : "+v"(bf[i]));
#pragma unroll
      for (int j = 0; j < 2; j++) acc[i][j] = mfma32(sf[j], bf[i], acc[i][j]);
    }
  }
  STAMP_END(st);
  float s = 0.f;
  for (int i = 0; i < 4; i++) for (int j = 0; j < 2; j++) for (int e = 0; e < 16; e++) s += acc[i][j][e];
  out[blockIdx.x * 512 + tid] = s;
}

template <int MODE, int WAVES>      // MODE 0: reads, multiplies, barrier per stage   2: next k-step's fragments read under this one's multiplies
__global__ __launch_bounds__(WAVES * 64, 2) void loop_a32(float* out, Stamp* st, int iters, int zero) {
  constexpr int NJ = WAVES == 8 ? 2 : 4;              // 32-column fragments of the small operand per wave
  constexpr int NT = WAVES * 64;
  __shared__ __attribute__((aligned(16))) char l0[4 * IMG], l1[4 * IMG], l2[4 * IMG], l3[4 * IMG];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = WAVES == 8 ? wave >> 2 : wave >> 1;
  const int simg = WAVES == 8 ? 2 + ((wave & 3) >> 1) : 2 + (wave & 1);
  const int sf0 = WAVES == 8 ? 2 * (wave & 1) : 0;    // first fragment of the small image
  fill(l0, 4 * IMG, tid, NT, 1, zero); fill(l1, 4 * IMG, tid, NT, 2, zero); fill(l2, 4 * IMG, tid, NT, 3, zero); fill(l3, 4 * IMG, tid, NT, 4, zero);
  __syncthreads();
  f32x16_t acc[4][NJ];
  for (int i = 0; i < 4; i++) for (int j = 0; j < NJ; j++) for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
  int ql = lane;
  asm volatile("" : "+v"(ql));
  auto load = [&](u32x4_t* f, const char* base, int kk) {
#pragma unroll
    for (int j = 0; j < NJ; j++) f[j] = frag32(base + simg * IMG, sf0 + j, kk, ql);
#pragma unroll
    for (int i = 0; i < 4; i++) f[NJ + i] = frag32(base + wm * IMG, i, kk, ql);
  };
  auto mul = [&](const u32x4_t* f) {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < NJ; j++) acc[i][j] = mfma32(f[j], f[NJ + i], acc[i][j]);
  };
  auto mul_load = [&](const u32x4_t* f, u32x4_t* nf, const char* nbase, int nkk) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (i < NJ) nf[i] = frag32(nbase + simg * IMG, sf0 + i, nkk, ql);
      nf[NJ + i] = frag32(nbase + wm * IMG, i, nkk, ql);
#pragma unroll
      for (int j = 0; j < NJ; j++) acc[i][j] = mfma32(f[j], f[NJ + i], acc[i][j]);
    }
  };
  const bool live = iters > 0;
  u32x4_t fa[NJ + 4], fb[NJ + 4];
  STAMP_BEGIN;
  if (MODE == 2) {
    load(fa, l0, 0);
    for (int it = 0; it < iters; it++) {
      if (live) mul_load(fa, fb, l0, 1);
      if (live) mul_load(fb, fa, l1, 0);
      if (live) mul_load(fa, fb, l1, 1);
      if (live) mul_load(fb, fa, l2, 0);
      if (live) mul_load(fa, fb, l2, 1);
      if (live) mul_load(fb, fa, l3, 0);
      if (live) mul_load(fa, fb, l3, 1);
      if (live) mul_load(fb, fa, l0, 0);
    }
  } else {
    const char* ring[4] = {l0, l1, l2, l3};
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int b = 0; b < 4; b++) {
        if (live) { load(fa, ring[b], 0); load(fb, ring[b], 1); mul(fa); mul(fb); }
        __builtin_amdgcn_s_barrier();
      }
    }
  }
  STAMP_END(st);
  float s = 0.f;
  for (int i = 0; i < 4; i++) for (int j = 0; j < NJ; j++) for (int e = 0; e < 16; e++) s += acc[i][j][e];
  s += __builtin_bit_cast(float, fa[0][0]) * 1e-30f;
  out[blockIdx.x * NT + tid] = s;
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
  hipMemset(dst, 0, (size_t)blocks * 8 * sizeof(Stamp));      // kernels with 4 waves leave half the entries untouched
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
    snprintf(nm, sizeof nm, "R32 registers only, 32x32x16, %s", z);
    run(nm, [&] { hipLaunchKernelGGL(loop_r32, dim3(blocks), dim3(512), 0, 0, d, st, iters, zero); }, st, iters, blocks);
    snprintf(nm, sizeof nm, "A32 8 waves, reads + barrier, %s", z);
    run(nm, [&] { hipLaunchKernelGGL((loop_a32<0, 8>), dim3(blocks), dim3(512), 0, 0, d, st, iters, zero); }, st, iters, blocks);
    snprintf(nm, sizeof nm, "P32 8 waves, reads under MFMAs, %s", z);
    run(nm, [&] { hipLaunchKernelGGL((loop_a32<2, 8>), dim3(blocks), dim3(512), 0, 0, d, st, iters, zero); }, st, iters, blocks);
    snprintf(nm, sizeof nm, "A32 4 waves 128x128, reads + barrier, %s", z);
    run(nm, [&] { hipLaunchKernelGGL((loop_a32<0, 4>), dim3(blocks), dim3(256), 0, 0, d, st, iters, zero); }, st, iters, blocks);
    snprintf(nm, sizeof nm, "P32 4 waves 128x128, reads under MFMAs, %s", z);
    run(nm, [&] { hipLaunchKernelGGL((loop_a32<2, 4>), dim3(blocks), dim3(256), 0, 0, d, st, iters, zero); }, st, iters, blocks);
  }
  if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
  return 0;
}
