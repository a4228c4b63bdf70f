// probe: what a cross-stream hand-over costs the stream that records it, and whether a kernel's own completion signal can carry it.
// Chain stream: N kernels of ~20 us back to back.  Variants:
//   A  nothing between them                                                       (the floor)
//   B  hipEventRecord between them (disable-timing | release-to-device events)    (what the engine's step plan does per layer)
//   C  the same events passed as stopEvent of hipExtLaunchKernelGGL               (no separate marker packet)
//   D/E = B/C with a second stream that waits for every event and runs a short kernel behind it (the weight-gradient side stream)
// Prints us per chain kernel (wall, HIP events around the whole sequence on the chain stream) and, for D/E, the time until the side stream
// has drained too.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 probe_marker.hip -o probe_marker
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>

__global__ void spin_kernel(float* out, int iters) {
  float a = threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; i++) a = a * 1.0001f + 0.5f;
  if (a == 12345.f) out[0] = a;
}

int main() {
  const int N = 24, REP = 20;
  float* d; hipMalloc(&d, 1024);
  hipStream_t chain, side; hipStreamCreate(&chain); hipStreamCreate(&side);
  std::vector<hipEvent_t> ev(N);
  for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventReleaseToDevice);
  hipEvent_t t0, t1, t2; hipEventCreate(&t0); hipEventCreate(&t1); hipEventCreate(&t2);
  const dim3 grid(512), block(256);
  const int iters = 20000, side_iters = 4000;
  auto run = [&](int mode) {
    float best = 1e9f, best_side = 1e9f;
    for (int rep = 0; rep < REP; rep++) {
      hipDeviceSynchronize();
      hipEventRecord(t0, chain);
      for (int k = 0; k < N; k++) {
        const bool ext = mode == 2 || mode == 4;
        if (ext) hipExtLaunchKernelGGL(spin_kernel, grid, block, 0, chain, nullptr, ev[k], 0, d, iters);
        else hipLaunchKernelGGL(spin_kernel, grid, block, 0, chain, d, iters);
        if (mode == 1 || mode == 3) hipEventRecord(ev[k], chain);
        if (mode >= 3) {
          hipStreamWaitEvent(side, ev[k], 0);
          hipLaunchKernelGGL(spin_kernel, dim3(64), block, 0, side, d, side_iters);
        }
      }
      hipEventRecord(t1, chain);
      if (mode >= 3) { hipEventRecord(t2, side); hipEventSynchronize(t2); }
      hipEventSynchronize(t1);
      float ms; hipEventElapsedTime(&ms, t0, t1);
      if (ms < best) best = ms;
      if (mode >= 3) { hipStreamWaitEvent(chain, t2, 0); float ms2; hipEventElapsedTime(&ms2, t0, t2); if (ms2 < best_side) best_side = ms2; }
    }
    const char* names[] = {"A  kernels back to back", "B  hipEventRecord behind every kernel", "C  the event as the kernel's stopEvent (hipExtLaunchKernelGGL)",
                           "D  B + a side stream waiting for every event", "E  C + a side stream waiting for every event"};
    printf("%-66s %7.2f us per chain kernel", names[mode], best * 1e3f / N);
    if (mode >= 3) printf("   side stream drained after %8.1f us", best_side * 1e3f);
    printf("\n");
  };
  for (int m = 0; m < 5; m++) run(m);
  if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
  return 0;
}
