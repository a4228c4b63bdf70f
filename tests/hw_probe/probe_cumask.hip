// hipExtStreamCreateWithCUMask on MI355X: which CUs (XCC, shader engine, CU id) a mask bit selects.  For a few mask patterns, 8192
// one-wave work-groups are launched on the masked stream; each records hwreg(XCC_ID) and hwreg(HW_ID); the host prints how many distinct
// CUs ran work, per XCC.  ("probe the mask -> XCD mapping first".)   hipcc --offload-arch=gfx950 -o probe_cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>

__global__ void where_kernel(uint32_t* out) {
  uint32_t xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  // hold the slot a little so that the grid spreads over every CU the mask allows
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < 200) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
  hipStream_t s;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%-28s hipExtStreamCreateWithCUMask failed\n", name); return; }
  const int n = 8192;
  uint32_t* d;
  hipMalloc(&d, 2 * n * sizeof(uint32_t));
  hipLaunchKernelGGL(where_kernel, dim3(n), dim3(64), 0, s, d);
  hipStreamSynchronize(s);
  std::vector<uint32_t> h(2 * n);
  hipMemcpy(h.data(), d, 2 * n * sizeof(uint32_t), hipMemcpyDeviceToHost);
  std::set<uint32_t> per_xcc[16];
  for (int i = 0; i < n; i++) {
    const uint32_t xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
    const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
  }
  int bits = 0;
  for (uint32_t w : mask) bits += __builtin_popcount(w);
  printf("%-28s %3d bits set -> CUs used per XCC:", name, bits);
  int tot = 0;
  for (int x = 0; x < 8; x++) { printf(" %2zu", per_xcc[x].size()); tot += (int)per_xcc[x].size(); }
  printf("   total %d\n", tot);
  hipFree(d);
  hipStreamDestroy(s);
}

int main() {
  std::vector<uint32_t> m(8, 0);
  auto fill = [&](auto pred) { for (int i = 0; i < 256; i++) { if (pred(i)) m[i / 32] |= 1u << (i % 32); else m[i / 32] &= ~(1u << (i % 32)); } };
  fill([](int i) { return true; });            run("all 256", m);
  fill([](int i) { return i < 32; });          run("bits 0-31", m);
  fill([](int i) { return i < 64; });          run("bits 0-63", m);
  fill([](int i) { return i < 128; });         run("bits 0-127", m);
  fill([](int i) { return i >= 64; });         run("bits 64-255", m);
  fill([](int i) { return i % 8 < 2; });       run("bits with i % 8 < 2", m);
  fill([](int i) { return i % 8 == 0; });      run("bits with i % 8 == 0", m);
  fill([](int i) { return i % 4 == 0; });      run("bits with i % 4 == 0", m);
  fill([](int i) { return (i / 8) % 4 == 0; }); run("bits with (i/8) % 4 == 0", m);
  return 0;
}
