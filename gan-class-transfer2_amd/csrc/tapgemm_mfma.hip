// Implicit-GEMM "tap GEMM" for the 4x4 / stride-2 convolutions of train.py:145-169 on gfx950.
//
//   out[m][n] = sum_{tap} sum_{c<K} src[pix(m,tap)][c] * W[tap](c,n)        (fp32 accumulate, MFMA)
//
// FORM_CONV  (Conv2D forward, train.py:161-166; Conv2DTranspose input gradient):
//     m walks the SMALL grid (the output), 16 taps read the BIG grid at (2sh+kh-1, 2sw+kw-1),
//     weights are the Keras kernel viewed as [tap][K][N] (N contiguous)  -> "T image" in LDS.
// FORM_CONVT (Conv2DTranspose forward, train.py:148-153; Conv2D input gradient):
//     one launch-z per output parity phase (ph,pw); m walks the SMALL grid (the source), the output pixel
//     is (2sh+ph, 2sw+pw); each phase has 2x2 taps kh = 1-ph+2a, source row sh+ph-a,
//     weights are the Keras kernel viewed as [tap][N][K] (K contiguous)  -> "N image" in LDS.
//
// Tile: BM x BN outputs per 256-thread workgroup (4 waves, each 64x64 = 4x4 MFMA 16x16x32 tiles), BK = 64
// channels of one tap per step, register-staged double-buffered LDS (one barrier per step).
// MFMA orientation: A operand = weights (rows = n), B operand = activations (cols = m), so every lane ends
// with 4 consecutive output channels of one pixel -> 8-byte NHWC stores.
#include "gct2_common.h"
#include <algorithm>

namespace {

constexpr int BK = 64;

template <typename T, int FORM, int BM, int BN, int EPI>
__global__ __launch_bounds__(256) void tapgemm_kernel(TapGemmParams p) {
  static_assert((BM / 64) * (BN / 64) == 4, "4 waves of 64x64");
  static_assert(FORM == FORM_CONVT || BN == 128, "T image is 128 columns wide");
  constexpr int WAVES_N = BN / 64;
  constexpr int NA = BM / 32;                      // 16-byte chunks per thread, activation tile
  constexpr int NW = (FORM == FORM_CONV) ? 4 : BN / 32;
  constexpr int A_BYTES = BM * 128;
  constexpr int W_BYTES = (FORM == FORM_CONV) ? 64 * 256 : BN * 128;
  constexpr int NTAPS = (FORM == FORM_CONV) ? 16 : 4;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* a_img[2]; char* w_img[2];
  a_img[0] = smem; a_img[1] = smem + A_BYTES + W_BYTES;
  w_img[0] = smem + A_BYTES; w_img[1] = smem + 2 * A_BYTES + W_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WAVES_N, wm = wave / WAVES_N;
  const int Hs = p.Hs, Ws = p.Ws, K = p.K, N = p.N;
  const int M = p.B * Hs * Ws;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int ph = (FORM == FORM_CONVT) ? (int)((blockIdx.z & 3) >> 1) : 0;
  const int pw = (FORM == FORM_CONVT) ? (int)(blockIdx.z & 1) : 0;
  const int kslice = (FORM == FORM_CONVT) ? (int)(blockIdx.z >> 2) : (int)blockIdx.z;
  const int Hsrc = (FORM == FORM_CONV) ? 2 * Hs : Hs, Wsrc = (FORM == FORM_CONV) ? 2 * Ws : Ws;
  const T* __restrict__ xsrc = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ wsrc = reinterpret_cast<const T*>(p.w);

  // ---- per-thread staging descriptors (fixed over the whole K loop) ----
  const int a_chunk = tid & 7, a_row0 = tid >> 3;
  int a_h[NA], a_w[NA], a_pix[NA];                 // source pixel base per staged row; a_pix < 0: row >= M
#pragma unroll
  for (int i = 0; i < NA; i++) {
    const int m = m0 + a_row0 + 32 * i;
    if (m < M) {
      const int sw = m % Ws, t = m / Ws, sh = t % Hs, b = t / Hs;
      a_pix[i] = b * Hsrc * Wsrc;
      a_h[i] = (FORM == FORM_CONV) ? 2 * sh - 1 : sh + ph;
      a_w[i] = (FORM == FORM_CONV) ? 2 * sw - 1 : sw + pw;
    } else {
      a_pix[i] = -1; a_h[i] = 0; a_w[i] = 0;
    }
  }
  const int nk = (K + BK - 1) / BK;
  const int niter = NTAPS * nk;

  u32x4_t a_reg[NA], w_reg[NW];
  const u32x4_t zero4 = {0u, 0u, 0u, 0u};

  auto gload = [&](int it) {
    const int tap = it / nk, c0 = (it - tap * nk) * BK;
    int dh, dw_, tap16;
    if (FORM == FORM_CONV) { dh = tap >> 2; dw_ = tap & 3; tap16 = tap; }
    else {
      const int a = tap >> 1, c = tap & 1;
      dh = -a; dw_ = -c;
      tap16 = (1 - ph + 2 * a) * 4 + (1 - pw + 2 * c);
    }
    const bool cok = (c0 + a_chunk * 8) < K;
#pragma unroll
    for (int i = 0; i < NA; i++) {
      const int h = a_h[i] + dh, w = a_w[i] + dw_;
      const bool ok = cok && a_pix[i] >= 0 && (unsigned)h < (unsigned)Hsrc && (unsigned)w < (unsigned)Wsrc;
      a_reg[i] = zero4;
      if (ok) a_reg[i] = gload128(xsrc + (size_t)(a_pix[i] + h * Wsrc + w) * p.ldx + c0 + a_chunk * 8);
    }
    if (FORM == FORM_CONV) {                       // [tap][K][N]: 64 k-rows x 16 chunks of 8 n
      const int c = tid & 15;
#pragma unroll
      for (int i = 0; i < NW; i++) {
        const int kr = (tid >> 4) + 16 * i;
        const bool ok = (c0 + kr) < K && (n0 + c * 8) < N;
        w_reg[i] = zero4;
        if (ok) w_reg[i] = gload128(wsrc + ((size_t)tap16 * K + c0 + kr) * N + n0 + c * 8);
      }
    } else {                                       // [tap][N][K]: BN n-rows x 8 chunks of 8 k
      const int c = tid & 7;
#pragma unroll
      for (int i = 0; i < NW; i++) {
        const int nr = (tid >> 3) + 32 * i;
        const bool ok = (n0 + nr) < N && (c0 + c * 8) < K;
        w_reg[i] = zero4;
        if (ok) w_reg[i] = gload128(wsrc + ((size_t)tap16 * N + n0 + nr) * K + c0 + c * 8);
      }
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NA; i++) lds_write128(a_img[buf], nimg_off(a_row0 + 32 * i, a_chunk), a_reg[i]);
    if (FORM == FORM_CONV) {
#pragma unroll
      for (int i = 0; i < NW; i++) lds_write128(w_img[buf], timg_off((tid >> 4) + 16 * i, tid & 15), w_reg[i]);
    } else {
#pragma unroll
      for (int i = 0; i < NW; i++) lds_write128(w_img[buf], nimg_off((tid >> 3) + 32 * i, tid & 7), w_reg[i]);
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // split-K: this workgroup reduces iterations [it_lo, it_hi) only and leaves an fp32 partial slab
  const int it_per = (niter + p.ksplit - 1) / p.ksplit;
  const int it_lo = kslice * it_per, it_hi = min(niter, it_lo + it_per);
  if (it_lo < it_hi) {
    gload(it_lo);
    sstore(0);
  }
  __syncthreads();
  for (int it = it_lo; it < it_hi; it++) {
    const int buf = (it - it_lo) & 1;
    if (it + 1 < it_hi) gload(it + 1);
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      u32x4_t wf[4], af[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        wf[i] = (FORM == FORM_CONV) ? timg_frag(w_img[buf], wn * 64 + i * 16, kk, lane)
                                    : nimg_frag(w_img[buf], wn * 64 + i * 16, kk, lane);
        af[i] = nimg_frag(a_img[buf], wm * 64 + i * 16, kk, lane);
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = mfma16<T>(wf[i], af[j], acc[i][j]);
    }
    if (it + 1 < it_hi) sstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds out[m = .. + (lane&15)][n = .. + 4*(lane>>4) + r], r = 0..3 ----
  T* __restrict__ yout = reinterpret_cast<T*>(p.y);
  const T* __restrict__ actp = reinterpret_cast<const T*>(p.act);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int m = m0 + wm * 64 + j * 16 + (lane & 15);
    if (m >= M) continue;
    size_t opix;
    if (FORM == FORM_CONV) opix = (size_t)m;
    else {
      const int sw = m % Ws, t = m / Ws, sh = t % Hs, b = t / Hs;
      opix = ((size_t)b * (2 * Hs) + 2 * sh + ph) * (2 * Ws) + 2 * sw + pw;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int n = n0 + wn * 64 + i * 16 + 4 * (lane >> 4);
      if (n >= N) continue;
      f32x4_t v = acc[i][j];
      if (p.ksplit > 1) {   // partial sum: the finalize kernel adds the slabs and applies the epilogue
        const size_t npix = (size_t)M * (FORM == FORM_CONVT ? 4 : 1);
        *reinterpret_cast<f32x4_t*>(p.ws + ((size_t)kslice * npix + opix) * N + n) = v;
        continue;
      }
      if (EPI == EPI_BIAS_ACT) {
        if (p.bias) {
          const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(p.bias + n);
          v += bv;
        }
        if (p.relu) {
#pragma unroll
          for (int r = 0; r < 4; r++) v[r] = fmaxf(v[r], 0.f);
        }
      } else {
        if (actp) {
          const u32x2_t a2 = *reinterpret_cast<const u32x2_t*>(actp + opix * p.ldact + n);
          if (!(unpack_lo<T>(a2[0]) > 0.f)) v[0] = 0.f;
          if (!(unpack_hi<T>(a2[0]) > 0.f)) v[1] = 0.f;
          if (!(unpack_lo<T>(a2[1]) > 0.f)) v[2] = 0.f;
          if (!(unpack_hi<T>(a2[1]) > 0.f)) v[3] = 0.f;
        }
        if (p.accumulate) {
          const u32x2_t o2 = *reinterpret_cast<const u32x2_t*>(yout + opix * p.ldy + n);
          v[0] += unpack_lo<T>(o2[0]); v[1] += unpack_hi<T>(o2[0]);
          v[2] += unpack_lo<T>(o2[1]); v[3] += unpack_hi<T>(o2[1]);
        }
      }
      u32x2_t o = {pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
      *reinterpret_cast<u32x2_t*>(yout + opix * p.ldy + n) = o;
    }
  }
}

// sums the split-K slabs and applies the epilogue the GEMM kernel skipped; 4 channels per thread
template <typename T, int EPI>
__global__ __launch_bounds__(256) void tapgemm_finalize_kernel(TapGemmParams p, size_t npix) {
  const int N = p.N, n4 = N >> 2;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= npix * n4) return;
  const size_t opix = idx / n4;
  const int n = (int)(idx - opix * n4) * 4;
  f32x4_t v = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < p.ksplit; s++) v += *reinterpret_cast<const f32x4_t*>(p.ws + ((size_t)s * npix + opix) * N + n);
  T* __restrict__ yout = reinterpret_cast<T*>(p.y);
  if (EPI == EPI_BIAS_ACT) {
    if (p.bias) v += *reinterpret_cast<const f32x4_t*>(p.bias + n);
    if (p.relu) {
#pragma unroll
      for (int r = 0; r < 4; r++) v[r] = fmaxf(v[r], 0.f);
    }
  } else {
    if (p.act) {
      const u32x2_t a2 = *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const T*>(p.act) + opix * p.ldact + n);
      if (!(unpack_lo<T>(a2[0]) > 0.f)) v[0] = 0.f;
      if (!(unpack_hi<T>(a2[0]) > 0.f)) v[1] = 0.f;
      if (!(unpack_lo<T>(a2[1]) > 0.f)) v[2] = 0.f;
      if (!(unpack_hi<T>(a2[1]) > 0.f)) v[3] = 0.f;
    }
    if (p.accumulate) {
      const u32x2_t o2 = *reinterpret_cast<const u32x2_t*>(yout + opix * p.ldy + n);
      v[0] += unpack_lo<T>(o2[0]); v[1] += unpack_hi<T>(o2[0]);
      v[2] += unpack_lo<T>(o2[1]); v[3] += unpack_hi<T>(o2[1]);
    }
  }
  u32x2_t o = {pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
  *reinterpret_cast<u32x2_t*>(yout + opix * p.ldy + n) = o;
}

template <typename T, int FORM, int BM, int BN, int EPI>
int launch(TapGemmParams p, hipStream_t s) {
  const int M = p.B * p.Hs * p.Ws;
  constexpr int PH = FORM == FORM_CONVT ? 4 : 1;
  const int tiles = ((M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * PH;
  const int niter = (FORM == FORM_CONV ? 16 : 4) * ((p.K + BK - 1) / BK);
  const size_t npix = (size_t)M * PH;
  // small-M layers (bottleneck of the U-Net) cannot fill 256 CUs with output tiles: split the reduction
  p.ksplit = 1;
  p.ws = nullptr;
  size_t ws_bytes = 0;
  float* ws = gct2_workspace(&ws_bytes);
  if (ws && tiles < 192 && niter >= 4) {
    int want = (512 + tiles - 1) / tiles;
    const size_t slab = npix * p.N * sizeof(float);
    want = (int)std::min<size_t>((size_t)want, ws_bytes / slab);
    want = std::min(want, niter / 2);
    if (want >= 2) {
      const int per = (niter + want - 1) / want;
      p.ksplit = (niter + per - 1) / per;
      p.ws = ws;
    }
  }
  dim3 grid((M + BM - 1) / BM, (p.N + BN - 1) / BN, PH * p.ksplit);
  constexpr int A_BYTES = BM * 128;
  constexpr int W_BYTES = (FORM == FORM_CONV) ? 64 * 256 : BN * 128;
  const size_t lds = 2 * (A_BYTES + W_BYTES);
  auto kern = tapgemm_kernel<T, FORM, BM, BN, EPI>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
  if (p.ksplit > 1) {
    const size_t total = npix * (p.N >> 2);
    hipLaunchKernelGGL((tapgemm_finalize_kernel<T, EPI>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, npix);
  }
  return gct2_check_launch("tapgemm_mfma");
}

template <typename T>
int dispatch(int form, int epi, const TapGemmParams& p, hipStream_t s) {
  if (form == FORM_CONV) {
    return epi == EPI_BIAS_ACT ? launch<T, FORM_CONV, 128, 128, EPI_BIAS_ACT>(p, s)
                               : launch<T, FORM_CONV, 128, 128, EPI_MASK>(p, s);
  }
  const bool narrow = p.N <= 64;
  if (epi == EPI_BIAS_ACT)
    return narrow ? launch<T, FORM_CONVT, 256, 64, EPI_BIAS_ACT>(p, s) : launch<T, FORM_CONVT, 128, 128, EPI_BIAS_ACT>(p, s);
  return narrow ? launch<T, FORM_CONVT, 256, 64, EPI_MASK>(p, s) : launch<T, FORM_CONVT, 128, 128, EPI_MASK>(p, s);
}

}  // namespace

// true when the MFMA path can take this problem (16-byte aligned rows, whole 8-channel chunks)
bool tapgemm_mfma_supported(int dtype, const TapGemmParams& p) {
  if (dtype != GCT2_BF16 && dtype != GCT2_F16) return false;
  if (p.K % 8 || p.N % 8 || p.ldx % 8 || p.ldy % 4) return false;
  if (p.act && p.ldact % 4) return false;
  if ((uintptr_t)p.x % 16 || (uintptr_t)p.w % 16 || (uintptr_t)p.y % 8) return false;
  if (p.act && (uintptr_t)p.act % 8) return false;
  if (p.bias && (uintptr_t)p.bias % 16) return false;
  return true;
}

int tapgemm_mfma(int dtype, int form, int epi, const TapGemmParams& p, hipStream_t s) {
  if (dtype == GCT2_BF16) return dispatch<__bf16>(form, epi, p, s);
  return dispatch<_Float16>(form, epi, p, s);
}
