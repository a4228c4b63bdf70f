// The image layer: DownShuffle_0 = Conv2D(128, 4, 2, 'same', relu) on the 3-channel noised image (train.py:184, 161-166)
// and its weight gradient.  Cin <= 4 cannot use the 8-channel chunks of tapgemm_mfma.hip, so the reduction index is
// re-packed as k' = tap*4 + channel (16 taps x 4 slots = 64 = exactly one BK step, slot 3 zero for RGB):
//   forward : out[m][n]   = sum_k' A[m][k'] * W'[k'][n]        A = im2col row gathered with 2-byte loads (HBM-light:
//                                                               the image is 1/43 of the output bytes)
//   wgrad   : dW'[k'][n] += sum_r A[r][k'] * dz[r][n]           reduction over all output pixels, split over workgroups
// Both reuse the LDS images / MFMA fragments of gct2_common.h; fp32 accumulation.
#include "gct2_common.h"

namespace {

// 8 reduction slots (taps 2c, 2c+1; 4 channel slots each) of output pixel (b, sh, sw) as one 16-byte chunk
template <typename T>
__device__ __forceinline__ u32x4_t gather_chunk(const T* __restrict__ x, int ldx, int Cin, int pixbase, int sh, int sw, int c,
                                                int Hb, int Wb, bool row_ok) {
  const bool VEC8 = ldx >= 4 && (ldx & 3) == 0 && ((uintptr_t)x & 7) == 0;   // wave-uniform
  const int kh = c >> 1, kw0 = 2 * (c & 1);
  const int h = 2 * sh + kh - 1;
  uint32_t out[4] = {0u, 0u, 0u, 0u};
  if (row_ok && (unsigned)h < (unsigned)Hb) {
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int w = 2 * sw + kw0 + t - 1;
      if ((unsigned)w < (unsigned)Wb) {
        const T* px = x + (size_t)(pixbase + h * Wb + w) * ldx;
        if (VEC8) {   // 4 channel slots in one 8-byte load (the view is 8-byte aligned and at least 4 elements wide)
          const u32x2_t v2 = *reinterpret_cast<const u32x2_t*>(px);
          out[2 * t] = v2[0];
          out[2 * t + 1] = Cin == 4 ? v2[1] : (Cin == 3 ? (v2[1] & 0xffffu) : 0u);
          if (Cin == 1) out[2 * t] &= 0xffffu;
        } else {
          uint16_t v[4] = {0, 0, 0, 0};
#pragma unroll
          for (int ch = 0; ch < 4; ch++)
            if (ch < Cin) v[ch] = __builtin_bit_cast(uint16_t, px[ch]);
          out[2 * t] = (uint32_t)v[0] | ((uint32_t)v[1] << 16);
          out[2 * t + 1] = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
        }
      }
    }
  }
  return u32x4_t{out[0], out[1], out[2], out[3]};
}

template <typename T>
__global__ __launch_bounds__(256, 4) void rgb_fwd_kernel(TapGemmParams p) {   // <= 128 registers: four work-groups per CU (this layer is store-bound: occupancy = bytes in flight)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* a_img = smem;                 // N image: 128 pixels x 64 k'
  char* w_img = smem + 128 * 128;     // T image: 64 k' x 128 n
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wm = wave >> 1;
  const int Hs = p.Hs, Ws = p.Ws, Cin = p.K, N = p.N;
  const int Hb = 2 * Hs, Wb = 2 * Ws;
  const int M = p.B * Hs * Ws;
  const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
  const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ w = reinterpret_cast<const T*>(p.w);
  {
    const int c = tid & 7;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int row = (tid >> 3) + 32 * i, m = m0 + row;
      const int sw = m % Ws, t = m / Ws, sh = t % Hs, b = t / Hs;
      lds_write128(a_img, nimg_off(row, c), gather_chunk<T>(x, p.ldx, Cin, b * Hb * Wb, sh, sw, c, Hb, Wb, m < M));
    }
    const int c16 = tid & 15;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int kr = (tid >> 4) + 16 * i, tap = kr >> 2, ch = kr & 3;
      u32x4_t v = {0u, 0u, 0u, 0u};
      if (ch < Cin && (n0 + c16 * 8) < N) v = gload128(w + ((size_t)(tap * Cin + ch)) * N + n0 + c16 * 8);
      lds_write128(w_img, timg_off(kr, c16), v);
    }
  }
  __syncthreads();
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk < 2; kk++) {
    u32x4_t wf[4], af[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      wf[i] = timg_frag(w_img, wn * 64 + i * 16, kk, lane);
      af[i] = nimg_frag(a_img, wm * 64 + i * 16, kk, lane);
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = mfma16<T>(wf[i], af[j], acc[i][j]);
  }
  T* __restrict__ y = reinterpret_cast<T*>(p.y);
  const bool wide = ((uintptr_t)p.y % 16 == 0) && (p.ldy % 8 == 0);     // 16-byte stores possible (block-uniform)
  if (wide) {
    // v_permlane16_swap pairs the n-fragments 2k / 2k+1: afterwards a lane owns 8 consecutive channels of its pixel (see the
    // epilogue of tapgemm_kernel): half the store instructions of this store-bound layer
    const bool staged = n0 + 128 <= N;                          // whole 128-channel tile (block-uniform)
    if (staged) __syncthreads();                                // every wave is done with the operand images: smem becomes the output stage
    const int eg = lane >> 4;
    const int nlane = wn * 64 + 16 * (eg & 1) + 4 * (eg & ~1);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      f32x4_t v0[2], v1[2];
#pragma unroll
      for (int ip = 0; ip < 2; ip++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          float xa = acc[2 * ip][j][r], xb = acc[2 * ip + 1][j][r];
          asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(xa), "+v"(xb));
          v0[ip][r] = xa;
          v1[ip][r] = xb;
        }
      }
      const int ml = wm * 64 + j * 16 + (lane & 15);
      const int m = m0 + ml;
      if (m >= M && !staged) continue;
#pragma unroll
      for (int ip = 0; ip < 2; ip++) {
        const int n = n0 + nlane + 32 * ip;
        if (n >= N) continue;
        f32x4_t a = v0[ip], c = v1[ip];
        if (p.bias) {
          a += *reinterpret_cast<const f32x4_t*>(p.bias + n);
          c += *reinterpret_cast<const f32x4_t*>(p.bias + n + 4);
        }
        if (p.relu) {
#pragma unroll
          for (int r = 0; r < 4; r++) { a[r] = fmaxf(a[r], 0.f); c[r] = fmaxf(c[r], 0.f); }
        }
        const u32x4_t o = {pack2<T>(a[0], a[1]), pack2<T>(a[2], a[3]), pack2<T>(c[0], c[1]), pack2<T>(c[2], c[3])};
        if (staged) lds_write128(smem, ml * 256 + ((((nlane + 32 * ip) >> 3) ^ (ml & 15)) << 4), o);
        else *reinterpret_cast<u32x4_t*>(y + (size_t)m * p.ldy + n) = o;
      }
    }
    if (staged) {
      // the tile leaves through LDS so that 16 consecutive lanes store the 256 contiguous bytes of ONE pixel (a lane's own
      // fragments are 64-byte runs of 16 different pixels per instruction: half cache lines, measured 1.9 TB/s on this store-bound layer)
      __syncthreads();
      const int c16 = tid & 15;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int pr = (tid >> 4) + 16 * k;
        if (m0 + pr < M) {
          const u32x4_t o = lds_read128(smem, pr * 256 + ((c16 ^ (pr & 15)) << 4));
          *reinterpret_cast<u32x4_t*>(y + (size_t)(m0 + pr) * p.ldy + n0 + c16 * 8) = o;
          if (p.bits) p.bits[(size_t)(m0 + pr) * p.ldbits + (n0 >> 3) + c16] = (unsigned char)relu_bits8<T>(o);   // 16 lanes: 16 consecutive bytes
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int m = m0 + wm * 64 + j * 16 + (lane & 15);
    if (m >= M) continue;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int n = n0 + wn * 64 + i * 16 + 4 * (lane >> 4);
      if (n >= N) continue;
      f32x4_t v = acc[i][j];
      if (p.bias) v += *reinterpret_cast<const f32x4_t*>(p.bias + n);
      if (p.relu) {
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = fmaxf(v[r], 0.f);
      }
      u32x2_t o = {pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
      *reinterpret_cast<u32x2_t*>(y + (size_t)m * p.ldy + n) = o;
    }
  }
}

// dW'[k'][n] += sum_r A[r][k'] dz[r][n]; p.big = image x (Cb <= 4 channels), p.small = dz (Cs = N channels)
template <typename T>
__global__ __launch_bounds__(256) void rgb_wgrad_kernel(WgradParams p) {
  constexpr int IMG = 64 * 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* a_img[2]; char* s_img[2];
  a_img[0] = smem; a_img[1] = smem + 2 * IMG;
  s_img[0] = smem + IMG; s_img[1] = smem + 3 * IMG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Hs = p.Hs, Ws = p.Ws, Cin = p.Cb, N = p.Cs;
  const int Hb = 2 * Hs, Wb = 2 * Ws;
  const int R = p.B * Hs * Ws;
  const int n0 = blockIdx.y * 128;
  const int steps_total = (R + 63) / 64;
  const int steps_per = (steps_total + gridDim.x - 1) / gridDim.x;
  const int step_lo = blockIdx.x * steps_per, step_hi = min(steps_total, step_lo + steps_per);
  // (the launcher sizes the grid so that every work-group row has steps: each one owns a slab)
  if (step_lo >= step_hi) return;
  const T* __restrict__ x = reinterpret_cast<const T*>(p.big);
  const T* __restrict__ dz = reinterpret_cast<const T*>(p.small);

  const int ac = tid & 7, arow0 = tid >> 3;          // A: 64 rows x 8 chunks -> 2 per thread
  const int sc = tid & 15, srow0 = tid >> 4;         // dz: 64 rows x 16 chunks -> 4 per thread
  const bool s_ok = (n0 + sc * 8) < N;
  // Two register sets: the loads of step s+2 are issued before step s is multiplied and written to LDS one step later, so that
  // TWO steps of dz (2 x 16 KiB per work-group, 64 KiB per CU) are in flight towards HBM instead of one - this launch only streams dz
  // (67 MB at config 3) and one step in flight kept it at 2.3 TB/s (r05).
  u32x4_t a_reg[2][2], s_reg[2][4];
  auto gload = [&](int step, int set) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int r = step * 64 + arow0 + 32 * i;
      const int sw = r % Ws, t = r / Ws, sh = t % Hs, b = t / Hs;
      a_reg[set][i] = gather_chunk<T>(x, p.ldbig, Cin, b * Hb * Wb, sh, sw, ac, Hb, Wb, r < R);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int r = step * 64 + srow0 + 16 * i;
      s_reg[set][i] = u32x4_t{0u, 0u, 0u, 0u};
      if (s_ok && r < R) s_reg[set][i] = gload128(dz + (size_t)r * p.ldsmall + n0 + sc * 8);
    }
  };
  auto sstore = [&](int buf, int set) {
#pragma unroll
    for (int i = 0; i < 2; i++) lds_write128(a_img[buf], timg_off(arow0 + 32 * i, ac), a_reg[set][i]);
#pragma unroll
    for (int i = 0; i < 4; i++) lds_write128(s_img[buf], timg_off(srow0 + 16 * i, sc), s_reg[set][i]);
  };
  f32x4_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; i++) { acc[i][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[i][1] = acc[i][0]; }
  auto compute = [&](int buf) {
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      u32x4_t af[4], sf[2];
#pragma unroll
      for (int i = 0; i < 4; i++) af[i] = timg_frag(a_img[buf], i * 16, kk, lane);
#pragma unroll
      for (int j = 0; j < 2; j++) sf[j] = timg_frag(s_img[buf], wave * 32 + j * 16, kk, lane);
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = mfma16<T>(af[i], sf[j], acc[i][j]);
    }
  };
  gload(step_lo, 0);
  sstore(0, 0);
  if (step_lo + 1 < step_hi) gload(step_lo + 1, 1);
  __syncthreads();
  // step s is in LDS buffer (s - step_lo) & 1, step s+1 in register set (s + 1 - step_lo) & 1; two steps per trip: static register names
  for (int step = step_lo; step < step_hi; step += 2) {
    if (step + 2 < step_hi) gload(step + 2, 0);
    compute(0);
    if (step + 1 < step_hi) sstore(1, 1);
    __syncthreads();
    if (step + 1 >= step_hi) break;
    if (step + 3 < step_hi) gload(step + 3, 1);
    compute(1);
    if (step + 2 < step_hi) sstore(0, 0);
    __syncthreads();
  }
  // lane holds dW'[k' = 16 i + 4 (lane>>4) + r][n = n0 + 32 wave + 16 j + (lane&15)]; channel slot = r.
  // p.ws: every work-group row (blockIdx.x = its pixel range) stores its partial tensor into its own slab, summed in slab order by
  // wgrad_reduce_kernel or by the fused optimizer step (bit-reproducible); without scratch: fp32 atomics into dw.
  float* __restrict__ out = p.ws ? p.ws + (size_t)blockIdx.x * 16 * Cin * N : p.dw;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int tap = 4 * i + (lane >> 4);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (r >= Cin) continue;
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int n = n0 + wave * 32 + j * 16 + (lane & 15);
        if (n >= N) continue;
        if (p.ws) out[(size_t)(tap * Cin + r) * N + n] = acc[i][j][r];
        else atomicAdd(out + (size_t)(tap * Cin + r) * N + n, acc[i][j][r]);
      }
    }
  }
}

}  // namespace

bool rgb_fwd_supported(int dtype, const TapGemmParams& p) {
  if (dtype != GCT2_BF16 && dtype != GCT2_F16) return false;
  if (p.K > 4 || p.N % 8 || p.ldy % 4) return false;
  if ((uintptr_t)p.x % 2 || (uintptr_t)p.w % 16 || (uintptr_t)p.y % 8) return false;
  if (p.bias && (uintptr_t)p.bias % 16) return false;
  return true;
}
// the staged 16-byte epilogue (whole 128-channel tiles) also writes the ReLU bit plane of p.bits
bool rgb_fwd_writes_bits(const TapGemmParams& p) { return (uintptr_t)p.y % 16 == 0 && p.ldy % 8 == 0 && p.N % 128 == 0; }
int rgb_fwd(int dtype, const TapGemmParams& p, hipStream_t s) {
  const int M = p.B * p.Hs * p.Ws;
  dim3 grid((M + 127) / 128, (p.N + 127) / 128);
  const size_t lds = 128 * 128 + 64 * 256;
  if (dtype == GCT2_BF16) hipLaunchKernelGGL(rgb_fwd_kernel<__bf16>, grid, dim3(256), lds, s, p);
  else hipLaunchKernelGGL(rgb_fwd_kernel<_Float16>, grid, dim3(256), lds, s, p);
  return gct2_check_launch("rgb_fwd");
}
bool rgb_wgrad_supported(int dtype, const WgradParams& p) {
  if (dtype != GCT2_BF16 && dtype != GCT2_F16) return false;
  if (p.Cb > 4 || p.Cs % 8 || p.ldsmall % 8 || (uintptr_t)p.small % 16) return false;
  return true;
}
int wgrad_reduce(const float* ws, float* dw, size_t n, int nsplit, int accumulate, hipStream_t s);   // wgrad_mfma.hip

int rgb_wgrad(const gct2_ctx& c, int dtype, WgradParams p, hipStream_t s, WgradSlabs* defer) {
  if (defer) *defer = WgradSlabs{nullptr, 0, 0};
  const int R = p.B * p.Hs * p.Ws;
  const int steps_total = (R + 63) / 64;
  const int ntiles = (p.Cs + 127) / 128;
  // 512 work-groups (two per CU): measured 1024 -> 48 us (atomics of 1024 partial tiles), 512 -> 38 us, 256 -> 48 us
  int splits = max(1, min(512 / ntiles, steps_total / 4));
  const int per = (steps_total + splits - 1) / splits;
  splits = (steps_total + per - 1) / per;                       // every work-group row non-empty (each one owns a slab)
  dim3 grid(splits, ntiles);
  const size_t n = (size_t)16 * p.Cb * p.Cs;
  size_t ws_bytes = 0;
  float* ws = c.wgrad_scratch(&ws_bytes);
  p.ws = (ws && n % 4 == 0 && (uintptr_t)p.dw % 16 == 0 && n * sizeof(float) * splits <= ws_bytes) ? ws : nullptr;
  if (!p.ws && !p.accumulate) (void)hipMemsetAsync(p.dw, 0, n * sizeof(float), s);   // the atomic path adds into dw
  const size_t lds = 4 * 64 * 256;
  if (dtype == GCT2_BF16) hipLaunchKernelGGL(rgb_wgrad_kernel<__bf16>, grid, dim3(256), lds, s, p);
  else hipLaunchKernelGGL(rgb_wgrad_kernel<_Float16>, grid, dim3(256), lds, s, p);
  if (int e = gct2_check_launch("rgb_wgrad")) return e;
  // the slabs are never handed to the fused optimizer launch (`defer`): this tensor is tiny and the slabs are many, one thread per
  // element walking 512 slabs is latency-bound (29 us at the very end of the step); wgrad_reduce's wide form sums them in ~5 us
  if (p.ws) return wgrad_reduce(p.ws, p.dw, n, splits, p.accumulate, s);
  return GCT2_OK;
}
