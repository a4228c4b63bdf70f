// Weight-gradient GEMM of the 4x4/stride-2 convolutions (autodiff of train.py:148-153,161-166) on gfx950.
//
//   dW[gc][cs] += sum_r big[pix_big(r, tap(gc))][cb(gc)] * small[r][cs],   gc = tap*Cb + cb  (16*Cb rows)
//
// r walks the SMALL grid [B,Hs,Ws]; pix_big(r,(kh,kw)) = (2sh+kh-1, 2sw+kw-1) on the BIG grid (zero outside).
//   Conv2D          : big = layer input x (Cb = Cin),  small = dz (Cs = Cout)  -> dW (4,4,Cin,Cout)
//   Conv2DTranspose : big = dz (Cb = Cout),            small = x  (Cs = Cin)   -> dW (4,4,Cout,Cin)
// Both operands have the REDUCTION index r as their slow (row) index in memory, so both LDS tiles are
// "T images" ([64 r][128 channels]) consumed through ds_read_tr16_b64.  128 x 128 output tile per
// 256-thread workgroup, 64 rows of r per step, register-staged double-buffered LDS, split over r across
// gridDim.z with fp32 atomics into the (pre-zeroed / running) gradient arena.
#include "gct2_common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradParams p) {
  constexpr int IMG = 64 * 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* b_img[2]; char* s_img[2];
  b_img[0] = smem; b_img[1] = smem + 2 * IMG;
  s_img[0] = smem + IMG; s_img[1] = smem + 3 * IMG;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wm = wave >> 1;
  const int Hs = p.Hs, Ws = p.Ws, Cb = p.Cb, Cs = p.Cs;
  const int Hb = 2 * Hs, Wb = 2 * Ws;
  const int R = p.B * Hs * Ws;
  const int GC = 16 * Cb;
  const int tiles_n = (Cs + 127) / 128;
  const int gc0 = (blockIdx.x / tiles_n) * 128, cs0 = (blockIdx.x % tiles_n) * 128;
  // r range of this split, in whole 64-row steps
  const int steps_total = (R + 63) / 64;
  const int steps_per = (steps_total + p.rsplit - 1) / p.rsplit;
  const int step_lo = blockIdx.z * steps_per;
  const int step_hi = min(steps_total, step_lo + steps_per);
  if (step_lo >= step_hi) return;

  const T* __restrict__ big = reinterpret_cast<const T*>(p.big);
  const T* __restrict__ small = reinterpret_cast<const T*>(p.small);

  const int c = tid & 15, rr0 = tid >> 4;          // chunk (8 channels) and first row of this thread
  const int gc = gc0 + c * 8;
  const bool gc_ok = gc < GC;
  const int tap = gc_ok ? gc / Cb : 0, cb = gc_ok ? gc - tap * Cb : 0;
  const int kh = tap >> 2, kw = tap & 3;
  const bool cs_ok = (cs0 + c * 8) < Cs;

  // incremental (b, sh, sw) decode of r, advanced by 64 rows per step
  const int adv_w = 64 % Ws, q1 = 64 / Ws, adv_h = q1 % Hs, adv_b = q1 / Hs;
  int rb[4], rh[4], rw[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int r = step_lo * 64 + rr0 + 16 * i;
    rw[i] = r % Ws; const int t = r / Ws; rh[i] = t % Hs; rb[i] = t / Hs;
  }

  u32x4_t b_reg[4], s_reg[4];
  const u32x4_t zero4 = {0u, 0u, 0u, 0u};
  auto gload = [&](int step) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int r = step * 64 + rr0 + 16 * i;
      const int h = 2 * rh[i] + kh - 1, w = 2 * rw[i] + kw - 1;
      const bool okb = gc_ok && r < R && (unsigned)h < (unsigned)Hb && (unsigned)w < (unsigned)Wb;
      b_reg[i] = zero4;
      if (okb) b_reg[i] = gload128(big + ((size_t)(rb[i] * Hb + h) * Wb + w) * p.ldbig + cb);
      s_reg[i] = zero4;
      if (cs_ok && r < R) s_reg[i] = gload128(small + (size_t)r * p.ldsmall + cs0 + c * 8);
      // advance this row by 64 for the next step
      rw[i] += adv_w; rh[i] += adv_h; rb[i] += adv_b;
      if (rw[i] >= Ws) { rw[i] -= Ws; rh[i]++; }
      if (rh[i] >= Hs) { rh[i] -= Hs; rb[i]++; }
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      lds_write128(b_img[buf], timg_off(rr0 + 16 * i, c), b_reg[i]);
      lds_write128(s_img[buf], timg_off(rr0 + 16 * i, c), s_reg[i]);
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  gload(step_lo);
  sstore(0);
  __syncthreads();
  for (int step = step_lo; step < step_hi; step++) {
    const int buf = (step - step_lo) & 1;
    if (step + 1 < step_hi) gload(step + 1);
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      u32x4_t bf[4], sf[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        bf[i] = timg_frag(b_img[buf], wm * 64 + i * 16, kk, lane);
        sf[i] = timg_frag(s_img[buf], wn * 64 + i * 16, kk, lane);
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = mfma16<T>(bf[i], sf[j], acc[i][j]);
    }
    if (step + 1 < step_hi) sstore(buf ^ 1);
    __syncthreads();
  }

  // lane holds dW[gc = .. + 4*(lane>>4) + r][cs = .. + (lane&15)]
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = gc0 + wm * 64 + i * 16 + 4 * (lane >> 4) + r;
      if (row >= GC) continue;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int col = cs0 + wn * 64 + j * 16 + (lane & 15);
        if (col < Cs) atomicAdd(p.dw + (size_t)row * Cs + col, acc[i][j][r]);
      }
    }
  }
}

}  // namespace

bool wgrad_mfma_supported(int dtype, const WgradParams& p) {
  if (dtype != GCT2_BF16 && dtype != GCT2_F16) return false;
  if (p.Cb % 8 || p.Cs % 8 || p.ldbig % 8 || p.ldsmall % 8) return false;
  if ((uintptr_t)p.big % 16 || (uintptr_t)p.small % 16) return false;
  return true;
}

int wgrad_mfma(int dtype, WgradParams p, hipStream_t s) {
  const int R = p.B * p.Hs * p.Ws;
  const int tiles = ((16 * p.Cb + 127) / 128) * ((p.Cs + 127) / 128);
  const int steps_total = (R + 63) / 64;
  // aim at ~768 workgroups (3 per CU) but keep >= 4 steps of 64 rows per split
  int rsplit = (768 + tiles - 1) / tiles;
  rsplit = max(1, min(rsplit, steps_total / 4));
  p.rsplit = rsplit;
  dim3 grid(tiles, 1, rsplit);
  const size_t lds = 4 * 64 * 256;
  auto launch = [&](auto kern) {
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
  };
  if (dtype == GCT2_BF16) launch(wgrad_kernel<__bf16>);
  else launch(wgrad_kernel<_Float16>);
  return gct2_check_launch("wgrad_mfma");
}
