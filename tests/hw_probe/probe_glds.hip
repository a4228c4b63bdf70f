// Hardware probe (test infrastructure): buffer_load_dwordx4 ... lds (LDS-DMA) semantics on gfx950:
//  (1) lane l of a wave-instruction lands at lds_base + 16*l  (2) an out-of-range voffset writes ZEROS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void probe(const unsigned* src, unsigned nbytes, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) ((unsigned*)smem)[i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  // wave w fills piece w (1 KiB); lanes permuted on the SOURCE side: lane l reads chunk (l ^ 5); lanes 7 and 40 are out of range
  unsigned voff = (unsigned)(w * 1024 + ((l ^ 5) * 16));
  if (l == 7 || l == 40) voff = 0x80000000u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(smem + w * 1024), 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += blockDim.x) out[i] = ((unsigned*)smem)[i];
}
int main() {
  std::vector<unsigned> h(512), o(512);
  for (int i = 0; i < 512; i++) h[i] = 1000 + i;
  unsigned *d, *dout;
  hipMalloc(&d, 2048); hipMalloc(&dout, 2048);
  hipMemcpy(d, h.data(), 2048, hipMemcpyHostToDevice);
  probe<<<1, 128, 8192>>>(d, 2048, dout);
  hipMemcpy(o.data(), dout, 2048, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < 2; w++) for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++) {
    unsigned got = o[w * 256 + l * 4 + j];
    unsigned exp = (l == 7 || l == 40) ? 0u : h[w * 256 + (l ^ 5) * 4 + j];
    if (got != exp) { if (bad < 8) printf("w%d l%d j%d got %u exp %u\n", w, l, j, got, exp); bad++; }
  }
  printf("glds mismatches: %d\n", bad);
  return bad ? 1 : 0;
}
