// Direct (one thread per output) kernels for the same tap GEMMs as tapgemm_mfma.hip / wgrad_mfma.hip.
// They carry (a) the reference's default fp32 arithmetic (train.py:34,38: mixed_precision = False), where
// the 16-bit MFMA path does not apply, and (b) any shape the MFMA path rejects (channel counts that are
// not multiples of 8, e.g. the 3-channel image fed to DownShuffle_0, train.py:184,292).
// fp32 accumulation in k-order; no LDS, coalesced along the output-channel index where the layout allows.
#include "gct2_common.h"

namespace {

template <typename T, int FORM, int EPI>
__global__ __launch_bounds__(256) void direct_tapgemm_kernel(TapGemmParams p) {
  const int Hs = p.Hs, Ws = p.Ws, K = p.K, N = p.N;
  const size_t M = (size_t)p.B * Hs * Ws;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * N) return;
  const int n = (int)(idx % N);
  const int m = (int)(idx / N);
  const int ph = (FORM == FORM_CONVT) ? (int)(blockIdx.z >> 1) : 0;
  const int pw = (FORM == FORM_CONVT) ? (int)(blockIdx.z & 1) : 0;
  const int sw = m % Ws, t = m / Ws, sh = t % Hs, b = t / Hs;
  const int Hsrc = (FORM == FORM_CONV) ? 2 * Hs : Hs, Wsrc = (FORM == FORM_CONV) ? 2 * Ws : Ws;
  const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ w = reinterpret_cast<const T*>(p.w);
  float acc = 0.f;
  constexpr int NT = (FORM == FORM_CONV) ? 4 : 2;
  for (int a = 0; a < NT; a++) {
    for (int c = 0; c < NT; c++) {
      int h, ww, tap16;
      if (FORM == FORM_CONV) { h = 2 * sh + a - 1; ww = 2 * sw + c - 1; tap16 = a * 4 + c; }
      else { h = sh + ph - a; ww = sw + pw - c; tap16 = (1 - ph + 2 * a) * 4 + (1 - pw + 2 * c); }
      if ((unsigned)h >= (unsigned)Hsrc || (unsigned)ww >= (unsigned)Wsrc) continue;
      const T* xr = x + ((size_t)(b * Hsrc + h) * Wsrc + ww) * p.ldx;
      if (FORM == FORM_CONV) {
        const T* wr = w + (size_t)tap16 * K * N + n;
        for (int k = 0; k < K; k++) acc = fmaf(to_f32(xr[k]), to_f32(wr[(size_t)k * N]), acc);
      } else {
        const T* wr = w + ((size_t)tap16 * N + n) * K;
        for (int k = 0; k < K; k++) acc = fmaf(to_f32(xr[k]), to_f32(wr[k]), acc);
      }
    }
  }
  size_t opix;
  if (FORM == FORM_CONV) opix = (size_t)m;
  else opix = ((size_t)b * (2 * Hs) + 2 * sh + ph) * (2 * Ws) + 2 * sw + pw;
  T* y = reinterpret_cast<T*>(p.y) + opix * p.ldy + n;
  if (EPI == EPI_BIAS_ACT) {
    if (p.bias) acc += p.bias[n];
    if (p.relu) acc = fmaxf(acc, 0.f);
  } else {
    if (p.act) {
      const float a = to_f32(reinterpret_cast<const T*>(p.act)[opix * p.ldact + n]);
      if (!(a > 0.f)) acc = 0.f;
    }
    if (p.accumulate) acc += to_f32(*y);
  }
  *y = from_f32<T>(acc);
}

template <typename T>
__global__ __launch_bounds__(256) void direct_wgrad_kernel(WgradParams p) {
  const int Hs = p.Hs, Ws = p.Ws, Cb = p.Cb, Cs = p.Cs;
  const int Hb = 2 * Hs, Wb = 2 * Ws;
  const int R = p.B * Hs * Ws;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)16 * Cb * Cs) return;
  const int cs = (int)(idx % Cs);
  const int gc = (int)(idx / Cs);
  const int tap = gc / Cb, cb = gc - tap * Cb, kh = tap >> 2, kw = tap & 3;
  const int per = (R + p.rsplit - 1) / p.rsplit;
  const int r_lo = blockIdx.z * per, r_hi = min(R, r_lo + per);
  const T* __restrict__ big = reinterpret_cast<const T*>(p.big);
  const T* __restrict__ small = reinterpret_cast<const T*>(p.small);
  float acc = 0.f;
  for (int r = r_lo; r < r_hi; r++) {
    const int sw = r % Ws, t = r / Ws, sh = t % Hs, b = t / Hs;
    const int h = 2 * sh + kh - 1, w = 2 * sw + kw - 1;
    if ((unsigned)h >= (unsigned)Hb || (unsigned)w >= (unsigned)Wb) continue;
    acc = fmaf(to_f32(big[((size_t)(b * Hb + h) * Wb + w) * p.ldbig + cb]),
               to_f32(small[(size_t)r * p.ldsmall + cs]), acc);
  }
  if (r_lo < r_hi) atomicAdd(p.dw + idx, acc);
}

template <typename T>
int launch_tapgemm(int form, int epi, const TapGemmParams& p, hipStream_t s) {
  const size_t total = (size_t)p.B * p.Hs * p.Ws * p.N;
  dim3 grid((unsigned)((total + 255) / 256), 1, form == FORM_CONVT ? 4 : 1);
  if (form == FORM_CONV) {
    if (epi == EPI_BIAS_ACT) hipLaunchKernelGGL((direct_tapgemm_kernel<T, FORM_CONV, EPI_BIAS_ACT>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((direct_tapgemm_kernel<T, FORM_CONV, EPI_MASK>), grid, dim3(256), 0, s, p);
  } else {
    if (epi == EPI_BIAS_ACT) hipLaunchKernelGGL((direct_tapgemm_kernel<T, FORM_CONVT, EPI_BIAS_ACT>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((direct_tapgemm_kernel<T, FORM_CONVT, EPI_MASK>), grid, dim3(256), 0, s, p);
  }
  return gct2_check_launch("direct_tapgemm");
}

template <typename T>
int launch_wgrad(WgradParams p, hipStream_t s) {
  const size_t total = (size_t)16 * p.Cb * p.Cs;
  const int R = p.B * p.Hs * p.Ws;
  // enough threads to fill the chip: ~256k threads, each summing >= 64 rows
  int rsplit = (int)((262144 + total - 1) / total);
  rsplit = max(1, min(rsplit, (R + 63) / 64));
  p.rsplit = rsplit;
  dim3 grid((unsigned)((total + 255) / 256), 1, rsplit);
  if (!p.accumulate) (void)hipMemsetAsync(p.dw, 0, total * sizeof(float), s);      // the kernel adds with atomics
  hipLaunchKernelGGL(direct_wgrad_kernel<T>, grid, dim3(256), 0, s, p);
  return gct2_check_launch("direct_wgrad");
}

}  // namespace

int tapgemm_direct(int dtype, int form, int epi, const TapGemmParams& p, hipStream_t s) {
  switch (dtype) {
    case GCT2_F32: return launch_tapgemm<float>(form, epi, p, s);
    case GCT2_BF16: return launch_tapgemm<__bf16>(form, epi, p, s);
    default: return launch_tapgemm<_Float16>(form, epi, p, s);
  }
}

int wgrad_direct(int dtype, const WgradParams& p, hipStream_t s) {
  switch (dtype) {
    case GCT2_F32: return launch_wgrad<float>(p, s);
    case GCT2_BF16: return launch_wgrad<__bf16>(p, s);
    default: return launch_wgrad<_Float16>(p, s);
  }
}

// ---- stride-1 'same' convolutions with a KS x KS kernel (KS odd): Block's Conv2D(filters, 3, 1, 'same', relu) (train.py:131-139)
// and the bias-free 1 x 1 projection Dense(input_channels) of Residual's residual=True mode (train.py:104-112; a Dense on a rank-4
// tensor is a 1 x 1 convolution, its (Cin, Cout) kernel the same memory as (1, 1, Cin, Cout)).  Both are off in the reference's
// defaults (train.py:20, 26): one thread per output element, fp32 accumulation, every dtype.
namespace {

struct ConvS1Params {
  const void* x; int ldx;        // source view [B,H,W,K]
  const void* w;                 // (KS,KS,Cin,Cout)
  const float* bias;
  const void* act; int ldact;    // dgrad: mask source on the output grid (may be null)
  void* y; int ldy;              // output view [B,H,W,N]
  int B, H, W, K, N, KS;
  int relu, accumulate;
  int Cin, Cout;                 // kernel tensor dims (forward: K = Cin, N = Cout; input gradient: K = Cout, N = Cin)
};

// DGRAD = false: y[b,h,w,n] = act(bias[n] + sum_{kh,kw,k} x[b,h+kh-p,w+kw-p,k] w[kh,kw,k,n])
// DGRAD = true : y[b,h,w,n] (+)= mask * sum_{kh,kw,k} x[b,h-kh+p,w-kw+p,k] w[kh,kw,n,k]      (x = dz, n = input channel)
template <typename T, bool DGRAD>
__global__ __launch_bounds__(256) void direct_conv_s1_kernel(ConvS1Params p) {
  const size_t M = (size_t)p.B * p.H * p.W;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * p.N) return;
  const int n = (int)(idx % p.N);
  const size_t m = idx / p.N;
  const int w0 = (int)(m % p.W), h0 = (int)((m / p.W) % p.H), b = (int)(m / ((size_t)p.W * p.H));
  const int pad = (p.KS - 1) / 2;
  const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ wt = reinterpret_cast<const T*>(p.w);
  float acc = 0.f;
  for (int kh = 0; kh < p.KS; kh++) {
    const int h = DGRAD ? h0 - kh + pad : h0 + kh - pad;
    if ((unsigned)h >= (unsigned)p.H) continue;
    for (int kw = 0; kw < p.KS; kw++) {
      const int ww = DGRAD ? w0 - kw + pad : w0 + kw - pad;
      if ((unsigned)ww >= (unsigned)p.W) continue;
      const T* xr = x + ((size_t)(b * p.H + h) * p.W + ww) * p.ldx;
      const T* wr = wt + (size_t)(kh * p.KS + kw) * p.Cin * p.Cout;
      if (DGRAD) {
        for (int k = 0; k < p.K; k++) acc = fmaf(to_f32(xr[k]), to_f32(wr[(size_t)n * p.Cout + k]), acc);
      } else {
        for (int k = 0; k < p.K; k++) acc = fmaf(to_f32(xr[k]), to_f32(wr[(size_t)k * p.Cout + n]), acc);
      }
    }
  }
  T* y = reinterpret_cast<T*>(p.y) + m * p.ldy + n;
  if (!DGRAD) {
    if (p.bias) acc += p.bias[n];
    if (p.relu) acc = fmaxf(acc, 0.f);
  } else {
    if (p.act && !(to_f32(reinterpret_cast<const T*>(p.act)[m * p.ldact + n]) > 0.f)) acc = 0.f;
    if (p.accumulate) acc += to_f32(*y);
  }
  *y = from_f32<T>(acc);
}

// dw[kh,kw,i,o] += sum_{b,h,w} x[b,h+kh-p,w+kw-p,i] dz[b,h,w,o]; the pixel range is split over blockIdx.z, fp32 atomics
template <typename T>
__global__ __launch_bounds__(256) void direct_conv_s1_wgrad_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dz, int lddz,
                                                                    float* __restrict__ dw, int B, int H, int W, int Cin, int Cout, int KS,
                                                                    int rsplit) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)KS * KS * Cin * Cout) return;
  const int o = (int)(idx % Cout);
  const int i = (int)((idx / Cout) % Cin);
  const int tap = (int)(idx / ((size_t)Cout * Cin)), kh = tap / KS, kw = tap % KS;
  const int pad = (KS - 1) / 2;
  const int R = B * H * W, per = (R + rsplit - 1) / rsplit;
  const int r_lo = blockIdx.z * per, r_hi = min(R, r_lo + per);
  float acc = 0.f;
  for (int r = r_lo; r < r_hi; r++) {
    const int w0 = r % W, t = r / W, h0 = t % H, b = t / H;
    const int h = h0 + kh - pad, ww = w0 + kw - pad;
    if ((unsigned)h >= (unsigned)H || (unsigned)ww >= (unsigned)W) continue;
    acc = fmaf(to_f32(x[((size_t)(b * H + h) * W + ww) * ldx + i]), to_f32(dz[(size_t)r * lddz + o]), acc);
  }
  if (r_lo < r_hi) atomicAdd(dw + idx, acc);
}

template <typename T>
int conv_s1_t(bool dgrad, const ConvS1Params& p, hipStream_t s) {
  const size_t total = (size_t)p.B * p.H * p.W * p.N;
  dim3 grid((unsigned)((total + 255) / 256));
  if (dgrad) hipLaunchKernelGGL((direct_conv_s1_kernel<T, true>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((direct_conv_s1_kernel<T, false>), grid, dim3(256), 0, s, p);
  return gct2_check_launch("conv2d_s1");
}
template <typename T>
int conv_s1_wgrad_t(const void* x, int ldx, const void* dz, int lddz, float* dw, int B, int H, int W, int Cin, int Cout, int KS, int accumulate,
                    hipStream_t s) {
  const size_t total = (size_t)KS * KS * Cin * Cout;
  const int R = B * H * W;
  int rsplit = (int)((262144 + total - 1) / total);
  rsplit = max(1, min(rsplit, (R + 63) / 64));
  if (!accumulate) (void)hipMemsetAsync(dw, 0, total * sizeof(float), s);
  hipLaunchKernelGGL(direct_conv_s1_wgrad_kernel<T>, dim3((unsigned)((total + 255) / 256), 1, rsplit), dim3(256), 0, s,
                     reinterpret_cast<const T*>(x), ldx, reinterpret_cast<const T*>(dz), lddz, dw, B, H, W, Cin, Cout, KS, rsplit);
  return gct2_check_launch("conv2d_s1_wgrad");
}

}  // namespace

int conv_s1_direct(int dtype, bool dgrad, const void* x, int ldx, const void* w, const float* bias, const void* act, int ldact, void* y, int ldy,
                   int B, int H, int W, int K, int N, int KS, int relu, int accumulate, hipStream_t s) {
  ConvS1Params p{x, ldx, w, bias, act, ldact, y, ldy, B, H, W, K, N, KS, relu, accumulate, dgrad ? N : K, dgrad ? K : N};
  switch (dtype) {
    case GCT2_F32: return conv_s1_t<float>(dgrad, p, s);
    case GCT2_BF16: return conv_s1_t<__bf16>(dgrad, p, s);
    default: return conv_s1_t<_Float16>(dgrad, p, s);
  }
}
int conv_s1_wgrad_direct(int dtype, const void* x, int ldx, const void* dz, int lddz, float* dw, int B, int H, int W, int Cin, int Cout, int KS,
                         int accumulate, hipStream_t s) {
  switch (dtype) {
    case GCT2_F32: return conv_s1_wgrad_t<float>(x, ldx, dz, lddz, dw, B, H, W, Cin, Cout, KS, accumulate, s);
    case GCT2_BF16: return conv_s1_wgrad_t<__bf16>(x, ldx, dz, lddz, dw, B, H, W, Cin, Cout, KS, accumulate, s);
    default: return conv_s1_wgrad_t<_Float16>(x, ldx, dz, lddz, dw, B, H, W, Cin, Cout, KS, accumulate, s);
  }
}
