// probe: how fast does ONE CU take a 32-KiB GEMM stage into LDS, and what does that cost the multiplies running beside it?
//   fill D : LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction)            - what every MFMA kernel of this library uses
//   fill Z : the same through a descriptor with zero records (zero fill, nothing fetched)   - the mechanism without the bytes
//   fill V : buffer_load_dwordx4 into registers, ds_write_b128 one iteration later          - the register-staged path
// by 8 waves (32 pieces of 1 KiB per stage, 4 per wave) or by waves 0-3 only (8 per wave) while waves 4-7 - one per SIMD - run 64
// independent v_mfma_f32_16x16x32_bf16 per stage (1024 cycles of the matrix core), from registers only or with the 48 transposed
// fragment reads of a 256 x 256 x 32 stage issued between them.  No barriers: every wave free-runs `iters` stages and stamps its own
// loop; source = a 64-KiB window per work-group (L2-resident, larger than the 32-KiB L1).  Prints cycles per stage for the filling
// waves and for the multiplying waves.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-value -Wno-inline-asm -I../../include -I../../gan-class-transfer2_amd/csrc probe_fill.hip -o probe_fill
#include "gct2_common.h"
#include <cstdio>
#include <vector>
#include <algorithm>

typedef __attribute__((address_space(3))) void lds_void_t;
constexpr int STAGE = 32 * 1024, NST = 5, WINDOW = 64 * 1024;

__device__ __forceinline__ void dma16_hidden(__amdgpu_buffer_rsrc_t rsrc, char* lds_piece, unsigned voff) {
  const unsigned lds_addr = (unsigned)(uintptr_t)(lds_void_t*)lds_piece;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc) : "memory", "m0");
}
__device__ __forceinline__ unsigned short rnd_bf16(unsigned i, unsigned seed) {
  unsigned h = (i + seed) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  return (unsigned short)(((h & 1u) << 15) | ((0x7au + ((h >> 1) & 3u)) << 7) | ((h >> 8) & 0x7fu));
}
struct Stamp { unsigned long long t0, t1; };

// FILL: 0 = D, 1 = V, 2 = Z, 3 = nobody fills;  NFILL: filling waves (8 or 4);  MUL: 0 = nobody multiplies, 1 = waves 4-7 from registers,
// 2 = waves 4-7 with 48 transposed reads per stage
template <int FILL, int NFILL, int MUL>
__global__ __launch_bounds__(512, 2) void fill_kernel(const char* src, float* out, Stamp* st, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[NST * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int P = 32 / NFILL;                                   // pieces per filling wave and stage
  const bool filler = FILL != 3 && wave < NFILL, multiplier = MUL != 0 && wave >= 4;
  const char* win = src + (size_t)blockIdx.x * WINDOW;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(win), 0, FILL == 2 ? 0 : WINDOW, 0x00020000);
  for (int i = tid; i < NST * STAGE / 4; i += 512) reinterpret_cast<unsigned*>(lds)[i] = rnd_bf16(2 * i, 3) | ((unsigned)rnd_bf16(2 * i + 1, 3) << 16);
  __syncthreads();
  float result = 0.f;
  unsigned long long t0 = 0, t1 = 0;
  const bool live = iters > 0;
  if (filler) {
    t0 = __builtin_amdgcn_s_memtime();
    if (FILL == 0 || FILL == 2) {
      for (int it = 0; it < iters; it++) {
        char* slot = lds + (it % NST) * STAGE + wave * P * 1024;
        const unsigned base = (unsigned)(((it & 1) * STAGE + wave * P * 1024) + lane * 16);
#pragma unroll
        for (int k = 0; k < P; k++) dma16_hidden(rs, slot + k * 1024, base + k * 1024);
        if (P == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // two more stages in flight
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      u32x4_t ra[P], rb[P];
      auto load = [&](u32x4_t* r, int it) {
        const unsigned base = (unsigned)(((it & 1) * STAGE + wave * P * 1024) + lane * 16);
#pragma unroll
        for (int k = 0; k < P; k++) r[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + k * 1024, 0, 0);
      };
      auto store = [&](const u32x4_t* r, int it) {
        char* slot = lds + (it % NST) * STAGE + wave * P * 1024 + lane * 16;
#pragma unroll
        for (int k = 0; k < P; k++) *reinterpret_cast<u32x4_t*>(slot + k * 1024) = r[k];
      };
      load(ra, 0);
      for (int it = 0; it + 1 < iters; it += 2) {
        if (live) { load(rb, it + 1); store(ra, it); }
        if (live) { load(ra, it + 2); store(rb, it + 1); }
      }
      result += __builtin_bit_cast(float, ra[0][0]) * 1e-30f;
    }
    t1 = __builtin_amdgcn_s_memtime();
  } else if (multiplier) {
    u32x4_t sf[4], bf[8];
    for (int j = 0; j < 4; j++) for (int e = 0; e < 4; e++) sf[j][e] = rnd_bf16(tid * 64 + j * 8 + e * 2, 7) | ((unsigned)rnd_bf16(tid * 64 + j * 8 + e * 2 + 1, 7) << 16);
    for (int i = 0; i < 8; i++) for (int e = 0; e < 4; e++) bf[i][e] = rnd_bf16(tid * 64 + i * 8 + e * 2, 9) | ((unsigned)rnd_bf16(tid * 64 + i * 8 + e * 2 + 1, 9) << 16);
    f32x4_t acc[8][4];
    for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    int ql = lane;
    asm volatile("" : "+v"(ql));
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 2 * iters; it++) {                       // one wave per SIMD: two trips of 32 multiplies = one stage's 64
      const char* img = lds + (it % NST) * STAGE;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (MUL == 2) {                                            // 24 transposed reads per trip: this row's big fragment, a small one every other row
          bf[i] = timg_frag(img + (wave & 1) * 8192, i * 16, 0, ql);
          if (i < 4) sf[i] = timg_frag(img + 16384 + (wave & 1) * 8192, i * 16, 0, ql);
        } else asm volatile("" : "+v"(bf[i]));
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = mfma16<__bf16>(sf[j], bf[i], acc[i][j]);
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) result += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  }
  if (lane == 0) st[blockIdx.x * 8 + wave] = Stamp{t0, t1};
  out[blockIdx.x * 512 + tid] = result + __builtin_bit_cast(float, reinterpret_cast<unsigned*>(lds)[tid]) * 1e-30f;
}

template <typename F>
static void run(const char* name, F launch, Stamp* dst, int iters, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0.f, total = 0.f;
  while (total < 300.f) { hipEventRecord(e0); for (int k = 0; k < 5; k++) launch(); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); total += ms; }
  hipMemset(dst, 0, (size_t)blocks * 8 * sizeof(Stamp));
  hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<Stamp> h(blocks * 8);
  hipMemcpy(h.data(), dst, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
  std::vector<double> fill, mul;
  for (int b = 0; b < blocks; b++) for (int w = 0; w < 8; w++) {
    const Stamp& s = h[b * 8 + w];
    if (s.t1 > s.t0) (w < 4 ? fill : mul).push_back((double)(s.t1 - s.t0) / iters);
  }
  auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  printf("%-64s %8.3f ms   waves 0-3: %7.1f   waves 4-7: %7.1f cycles per stage\n", name, ms, med(fill), med(mul));
}

int main() {
  const int blocks = 256, iters = 2000;
  char* src; hipMalloc(&src, (size_t)blocks * WINDOW); hipMemset(src, 0x3c, (size_t)blocks * WINDOW);
  float* d; hipMalloc(&d, (size_t)blocks * 512 * 4);
  Stamp* st; hipMalloc(&st, (size_t)blocks * 8 * sizeof(Stamp));
#define RUN(F, N, M, text) run(text, [&] { hipLaunchKernelGGL((fill_kernel<F, N, M>), dim3(blocks), dim3(512), 0, 0, src, d, st, iters); }, st, iters, blocks)
  RUN(0, 8, 0, "D  LDS-DMA, 8 waves fill, nobody multiplies");
  RUN(2, 8, 0, "Z  LDS-DMA zero fill, 8 waves, nobody multiplies");
  RUN(1, 8, 0, "V  registers + ds_write_b128, 8 waves, nobody multiplies");
  RUN(0, 4, 0, "D  LDS-DMA, waves 0-3 fill, nobody multiplies");
  RUN(1, 4, 0, "V  registers + ds_write_b128, waves 0-3, nobody multiplies");
  RUN(3, 4, 1, "-  nobody fills, waves 4-7 multiply from registers");
  RUN(3, 4, 2, "-  nobody fills, waves 4-7 multiply with fragment reads");
  RUN(0, 4, 1, "D  waves 0-3 fill, waves 4-7 multiply from registers");
  RUN(2, 4, 1, "Z  waves 0-3 zero fill, waves 4-7 multiply from registers");
  RUN(1, 4, 1, "V  waves 0-3 fill, waves 4-7 multiply from registers");
  RUN(0, 4, 2, "D  waves 0-3 fill, waves 4-7 multiply with fragment reads");
  RUN(2, 4, 2, "Z  waves 0-3 zero fill, waves 4-7 multiply with fragment reads");
  RUN(1, 4, 2, "V  waves 0-3 fill, waves 4-7 multiply with fragment reads");
  if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
  return 0;
}
