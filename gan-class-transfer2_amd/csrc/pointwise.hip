// HBM-bound kernels of the train step: RNG + noising (train.py:224-234), Dense(3) head (train.py:198-202),
// fp32 MSE and its gradient (train.py:262-272), bias gradients, Keras Adam (train.py:75) and the
// mixed-precision loss-scale state machine (train.py:82-83).  All are streaming kernels: 16-byte accesses
// where the layout allows, grids capped at a few workgroups per CU with grid-stride loops.
#include "gct2_common.h"
#include <algorithm>

namespace {

constexpr int kMaxBlocks = 2048;   // 256 CUs x 8

// ---- Philox4x32-10 (Salmon et al. 2011), counter = (idx, stream), key = seed -------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
  const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
  const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
  c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
}
__device__ __forceinline__ void philox4x32_10(uint64_t seed, uint64_t stream_id, uint64_t idx, uint32_t (&out)[4]) {
  uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32)};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; r++) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) out[i] = c[i];
}

__global__ void rng_uniform_int_kernel(uint64_t seed, uint64_t stream_id, uint64_t offset, int32_t* out, size_t n,
                                       int lo, uint32_t range) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t e = offset + i;
  uint32_t r[4];
  philox4x32_10(seed, stream_id, e >> 2, r);
  out[i] = lo + (int32_t)__umulhi(r[e & 3], range);
}

// Box-Muller on the hardware transcendental units: v_log_f32 (log2), v_sin_f32 / v_cos_f32 (argument in revolutions, so
// u2 goes in as is).  Absolute error ~1e-6 on a unit normal - a noise source, not a parity quantity (TF's own stream
// cannot be reproduced) - and ~5x fewer instructions than libm's logf / sinf / cosf.
__device__ __forceinline__ void box_muller(uint32_t r0, uint32_t r1, float& zc, float& zs) {
  const float u1 = ((float)(r0 >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = ((float)(r1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float rad = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // -2 ln u1 = -2 ln2 log2 u1
  zc = rad * __builtin_amdgcn_cosf(u2);
  zs = rad * __builtin_amdgcn_sinf(u2);
}
// element e uses counter e>>2; lanes (0,1) of the counter feed elements 4c,4c+1, lanes (2,3) feed 4c+2,4c+3
__device__ __forceinline__ float philox_normal(uint64_t seed, uint64_t stream_id, uint64_t e) {
  uint32_t r[4];
  philox4x32_10(seed, stream_id, e >> 2, r);
  const int pair = (int)((e >> 1) & 1);
  float zc, zs;
  box_muller(r[2 * pair], r[2 * pair + 1], zc, zs);
  return (e & 1) ? zs : zc;
}
__global__ void rng_normal_kernel(uint64_t seed, uint64_t stream_id, uint64_t offset, float* out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = philox_normal(seed, stream_id, offset + i);
}

// noised = x sqrt(a) + eps sqrt(1 - a), a = alpha_dash(t) (train.py:85-93, 229-234).  fp32 / bf16: fp32 arithmetic, one rounding at
// the store.  fp16 = Keras' mixed_float16 policy, where x, eps and t are fp16 tensors (train.py:38, 227, 292): every operation
// rounds to fp16 (t/(steps+1), 1-t, the square, *0.25, both square roots, both products, the sum).
template <typename T> struct NoiseCoef {
  float sa, sb;
  __device__ __forceinline__ NoiseCoef(int t_int, int steps1) {
#pragma clang fp contract(off)      // every fp16 operation rounds on its own (no fused multiply-add): that IS the model
    if constexpr (sizeof(T) == 2 && !__is_same(T, __bf16)) {
      const _Float16 t = (_Float16)((_Float16)(float)t_int / (_Float16)(float)steps1);
      const _Float16 om = (_Float16)1.0f - t;
      const _Float16 a = (_Float16)(om * om) * (_Float16)0.25f;
      sa = (float)(_Float16)sqrtf((float)a);
      sb = (float)(_Float16)sqrtf((float)(_Float16)((_Float16)1.0f - a));
    } else {
      const float t = (float)t_int * (1.0f / (float)steps1);
      const float a = (1.f - t) * (1.f - t) * 0.25f;
      sa = sqrtf(a); sb = sqrtf(1.f - a);
    }
  }
  __device__ __forceinline__ T mix(float x, float eps) const {
#pragma clang fp contract(off)
    if constexpr (sizeof(T) == 2 && !__is_same(T, __bf16)) {
      const _Float16 xs = (_Float16)x * (_Float16)sa, es = (_Float16)eps * (_Float16)sb;     // x is exact in fp16 (u8/128 - 1)
      return (T)(xs + es);
    } else {
      return from_f32<T>(x * sa + eps * sb);
    }
  }
};

// noising with eps drawn on the fly from the same stream positions rng_normal_kernel would use.  One thread per Philox
// counter = 4 consecutive elements of the flattened [pixel][channel] image: one Philox call, two logs and two sin/cos pairs
// make four normals (philox_normal spends a whole call per element); values are bit-identical to it.
template <typename T>
__global__ void noise_rng_kernel(const float* __restrict__ x, const int32_t* __restrict__ t_int, uint64_t seed, uint64_t stream_id,
                                 uint64_t offset, float* __restrict__ eps_out, T* __restrict__ out, int ldout,
                                 T* __restrict__ out2, int ldout2, size_t n, int HW, int C, int steps1) {
  const uint64_t c0 = offset >> 2;
  const size_t ncounters = (size_t)(((offset + n + 3) >> 2) - c0);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < ncounters; t += stride) {
    const uint64_t ctr = c0 + t;
    uint32_t r[4];
    philox4x32_10(seed, stream_id, ctr, r);
    float nrm[4];
    box_muller(r[0], r[1], nrm[0], nrm[1]);
    box_muller(r[2], r[3], nrm[2], nrm[3]);
    // elements 4 ctr .. 4 ctr + 3 of the flattened image: (pixel, channel, batch) advance incrementally (one division per thread)
    const uint64_t e0 = 4 * ctr;
    const uint64_t first = e0 < offset ? offset : e0;
    if (first >= offset + n) continue;
    const uint32_t i0 = (uint32_t)(first - offset);
    uint32_t pix = i0 / (uint32_t)C, c = i0 - pix * (uint32_t)C;
    uint32_t b = pix / (uint32_t)HW, rem = pix - b * (uint32_t)HW;
    NoiseCoef<T> nc(1, steps1);
    bool have = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint64_t e = e0 + k;
      if (e < first || e >= offset + n) continue;
      if (!have) {
        nc = NoiseCoef<T>(t_int[b], steps1);
        have = true;
      }
      const size_t i = (size_t)(e - offset);
      if (eps_out) eps_out[i] = nrm[k];
      const T v = nc.mix(x[i], nrm[k]);
      out[(size_t)pix * ldout + c] = v;
      if (out2) out2[(size_t)pix * ldout2 + c] = v;
      if (++c == (uint32_t)C) {
        c = 0; ++pix;
        if (++rem == (uint32_t)HW) { rem = 0; ++b; have = false; }
      }
    }
  }
}

// r06: the train step's shape of the same kernel - 3 channels into the packed 4-slot image, 16-bit storage, nothing else written, stream
// position a multiple of 12.  A thread owns 12 consecutive elements = THREE Philox counters = FOUR whole pixels: three 16-byte loads of x,
// the same counters / normals / mix as noise_rng_kernel (bit-identical), and the four pixels leave as two 16-byte stores (slot 3 = 0,
// which is what the packed image holds there) instead of twelve 2-byte ones.  11 -> 5 us at config 3, on the step's critical path.
template <typename T>
__global__ __launch_bounds__(256) void noise_rng_px4_kernel(const float* __restrict__ x, const int32_t* __restrict__ t_int, uint64_t seed,
                                                            uint64_t stream_id, uint64_t ctr0, T* __restrict__ out, size_t ngroups, int HW,
                                                            int steps1) {
  static_assert(sizeof(T) == 2, "16-bit storage types only");
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < ngroups; t += stride) {
    const uint32_t b = (uint32_t)((4 * t) / (size_t)HW);          // HW % 4 == 0: the four pixels belong to one image
    const NoiseCoef<T> nc(t_int[b], steps1);
    const f32x4_t* xs = reinterpret_cast<const f32x4_t*>(x) + 3 * t;
    const f32x4_t x0 = xs[0], x1 = xs[1], x2 = xs[2];
    const float xv[12] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3], x2[0], x2[1], x2[2], x2[3]};
    uint16_t v[12];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      uint32_t r[4];
      philox4x32_10(seed, stream_id, ctr0 + 3 * t + j, r);
      float nrm[4];
      box_muller(r[0], r[1], nrm[0], nrm[1]);
      box_muller(r[2], r[3], nrm[2], nrm[3]);
#pragma unroll
      for (int k = 0; k < 4; k++) v[4 * j + k] = __builtin_bit_cast(uint16_t, nc.mix(xv[4 * j + k], nrm[k]));
    }
    u32x4_t o0, o1;
    o0[0] = (uint32_t)v[0] | ((uint32_t)v[1] << 16);  o0[1] = (uint32_t)v[2];
    o0[2] = (uint32_t)v[3] | ((uint32_t)v[4] << 16);  o0[3] = (uint32_t)v[5];
    o1[0] = (uint32_t)v[6] | ((uint32_t)v[7] << 16);  o1[1] = (uint32_t)v[8];
    o1[2] = (uint32_t)v[9] | ((uint32_t)v[10] << 16); o1[3] = (uint32_t)v[11];
    u32x4_t* dst = reinterpret_cast<u32x4_t*>(out) + 2 * t;
    dst[0] = o0;
    dst[1] = o1;
  }
}

template <typename T>
__global__ void noise_kernel(const float* __restrict__ x, const int32_t* __restrict__ t_int, const float* __restrict__ eps,
                             T* __restrict__ out, int ldout, T* __restrict__ out2, int ldout2, size_t npix, int HW, int C,
                             int steps1) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t total = npix * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t pix = i / C;
    const int c = (int)(i - pix * C);
    const int b = (int)(pix / HW);
    const NoiseCoef<T> nc(t_int[b], steps1);                   // alpha_dash, train.py:87-93
    const T v = nc.mix(x[i], eps[i]);                          // train.py:231-234
    out[pix * ldout + c] = v;
    if (out2) out2[pix * ldout2 + c] = v;
  }
}

// ---- Dense(3) head -------------------------------------------------------------------------------------
// (keras_f16_point, the mixed_float16 rounding points of the head: gct2_common.h)
template <typename T>
__global__ void dense_fwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ b,
                                 float* __restrict__ y, int M, int Cin, int Cout) {
  extern __shared__ float wsm[];                  // [Cin][4]
  for (int i = threadIdx.x; i < Cin * 4; i += blockDim.x) {
    const int o = i & 3, k = i >> 2;
    wsm[i] = (o < Cout) ? w[k * Cout + o] : 0.f;
  }
  __syncthreads();
  const int stride = gridDim.x * blockDim.x;
  for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += stride) {
    const T* xr = x + (size_t)m * ldx;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int k = 0; k < Cin; k++) {
      const float v = to_f32(xr[k]);
      a0 = fmaf(v, wsm[4 * k + 0], a0); a1 = fmaf(v, wsm[4 * k + 1], a1);
      a2 = fmaf(v, wsm[4 * k + 2], a2); a3 = fmaf(v, wsm[4 * k + 3], a3);
    }
    const float acc[4] = {a0, a1, a2, a3};
    for (int o = 0; o < Cout; o++) y[(size_t)m * Cout + o] = keras_f16_point<T>(acc[o] + (b ? b[o] : 0.f));
  }
}

// one workgroup walks tiles of PIX pixels: dx per pixel, and per-(i,o) partial sums of dw/db over the tile
template <typename T, int PIX>
__global__ __launch_bounds__(256) void dense_bwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ w,
                                                        const float* __restrict__ dy, T* __restrict__ dx, int lddx,
                                                        float* __restrict__ dw, float* __restrict__ db, int M, int Cin,
                                                        int Cout, int Cmask) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* wsm = reinterpret_cast<float*>(smem_raw);          // [Cin][4]
  float* dys = wsm + Cin * 4;                               // [PIX][4]
  T* xs = reinterpret_cast<T*>(dys + PIX * 4);              // [PIX][Cin]
  const int tid = threadIdx.x;
  for (int i = tid; i < Cin * 4; i += 256) {
    const int o = i & 3, k = i >> 2;
    wsm[i] = (o < Cout) ? w[k * Cout + o] : 0.f;
  }
  const int nout = (Cin + 1) * Cout;                        // dw entries + db entries: entry e = tid + 256 k belongs to thread tid
  constexpr int ENT = 8;                                    // (Cin + 1) * Cout <= 2048
  float wacc[ENT];
#pragma unroll
  for (int k = 0; k < ENT; k++) wacc[k] = 0.f;
  const int ntiles = (M + PIX - 1) / PIX;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int mbase = tile * PIX;
    __syncthreads();
    for (int i = tid; i < PIX * Cin; i += 256) {
      const int pm = i / Cin, k = i - pm * Cin;
      xs[i] = (mbase + pm < M) ? x[(size_t)(mbase + pm) * ldx + k] : from_f32<T>(0.f);
    }
    for (int i = tid; i < PIX * 4; i += 256) {
      const int pm = i >> 2, o = i & 3;
      dys[i] = (o < Cout && mbase + pm < M) ? keras_f16_point<T>(dy[(size_t)(mbase + pm) * Cout + o]) : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < PIX * Cmask; i += 256) {
      const int pm = i / Cmask, k = i - pm * Cmask;
      if (mbase + pm >= M) continue;
      float g = dys[4 * pm] * wsm[4 * k] + dys[4 * pm + 1] * wsm[4 * k + 1] + dys[4 * pm + 2] * wsm[4 * k + 2] +
                dys[4 * pm + 3] * wsm[4 * k + 3];
      if (!(to_f32(xs[pm * Cin + k]) > 0.f)) g = 0.f;
      dx[(size_t)(mbase + pm) * lddx + k] = from_f32<T>(g);
    }
#pragma unroll
    for (int k = 0; k < ENT; k++) {
      const int e = tid + 256 * k;
      if (e >= nout) break;
      const int my_i = e / Cout, my_o = e - my_i * Cout;
      float a = wacc[k];
      if (my_i < Cin) {
        for (int pm = 0; pm < PIX; pm++) a = fmaf(to_f32(xs[pm * Cin + my_i]), dys[4 * pm + my_o], a);
      } else {
        for (int pm = 0; pm < PIX; pm++) a += dys[4 * pm + my_o];
      }
      wacc[k] = a;
    }
  }
#pragma unroll
  for (int k = 0; k < ENT; k++) {
    const int e = tid + 256 * k;
    if (e >= nout) break;
    const int my_i = e / Cout, my_o = e - my_i * Cout;
    if (my_i < Cin) atomicAdd(dw + my_i * Cout + my_o, wacc[k]);
    else if (db) atomicAdd(db + my_o, wacc[k]);
  }
}

// ---- MSE -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                  float* __restrict__ dpred, float* __restrict__ partials, size_t n,
                                                  const float* __restrict__ loss_scale_ptr) {
  const float scale = (loss_scale_ptr ? *loss_scale_ptr : 1.f) * 2.0f / (float)n;
  float acc = 0.f;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float d = pred[i] - target[i];
    acc = fmaf(d, d, acc);
    dpred[i] = d * scale;
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ float ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ __launch_bounds__(256) void mse_finish_kernel(const float* __restrict__ partials, int nparts, float* loss, float inv_n) {
  double acc = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += (double)partials[i];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ double ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) *loss = (float)((ws[0] + ws[1] + ws[2] + ws[3]) * (double)inv_n);
}

// ---- fused head for the train step: Dense(3) forward + fp32 MSE + its gradient + Dense backward, ONE pass over R_0 -------
// (train.py:198-202, 262-272 and their autodiff).  Per tile of 256 pixels: the 16-bit rows are copied to LDS with
// 16-byte loads (a tile is one contiguous byte range because consecutive pixels are `ld` apart), thread = pixel
// computes pred / dpred / the masked input gradient row (ds_read_b128; row stride ld*2 bytes is an odd multiple of
// 16 for ld = 72, hence conflict-free), the gradient rows leave through LDS as whole 16-byte chunks, and threads
// (i, o) accumulate dw[i][o] += x[pix][i] * dpred[pix][o] over the tile.
template <typename T, int PIX>
__global__ __launch_bounds__(256) void dense_head_train_kernel(const T* __restrict__ x, int ld, const float* __restrict__ w,
                                                               const float* __restrict__ bias, const float* __restrict__ target,
                                                               float* __restrict__ pred_out, T* __restrict__ dx, int lddx,
                                                               float* __restrict__ dw, float* __restrict__ db,
                                                               float* __restrict__ partials, int M, int Cin, int Cout, int Cmask,
                                                               const float* __restrict__ loss_scale_ptr, float* __restrict__ db_dx) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int nchunk = ld >> 3;                                    // 16-byte chunks per pixel row
  T* xs = reinterpret_cast<T*>(smem_raw);                        // [PIX][ld]
  T* dxs = xs + PIX * ld;                                        // [PIX][Cmask]
  float* dps = reinterpret_cast<float*>(dxs + PIX * Cmask);      // [PIX][4]
  float* wsm = dps + PIX * 4;                                    // [ld][4]  (rows >= Cin are zero)
  const int tid = threadIdx.x;
  for (int i = tid; i < ld * 4; i += 256) {
    const int o = i & 3, k = i >> 2;
    wsm[i] = (o < Cout && k < Cin) ? w[k * Cout + o] : 0.f;
  }
  const float gscale = (loss_scale_ptr ? *loss_scale_ptr : 1.f) * 2.0f / ((float)M * (float)Cout);
  const int nout = (Cin + 1) * Cout;
  const int my_i = tid / Cout, my_o = tid - my_i * Cout;
  float wacc = 0.f, lacc = 0.f, bacc = 0.f;   // bacc: column sum of the gradient rows for channel tid (db of the layer below)
  const int ntiles = (M + PIX - 1) / PIX;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int mbase = tile * PIX;
    const int npx = min(PIX, M - mbase);
    __syncthreads();
    {   // stage the tile: npx * nchunk chunks, contiguous in memory
      const u32x4_t* src = reinterpret_cast<const u32x4_t*>(x + (size_t)mbase * ld);
      u32x4_t* dst = reinterpret_cast<u32x4_t*>(xs);
      for (int i = tid; i < npx * nchunk; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    if (tid < npx) {
      const int m = mbase + tid;
      float a[4] = {0.f, 0.f, 0.f, 0.f};
      const u32x4_t* row = reinterpret_cast<const u32x4_t*>(xs + tid * ld);
      for (int c = 0; c < nchunk; c++) {
        const u32x4_t v = row[c];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float lo = unpack_lo<T>(v[q]), hi = unpack_hi<T>(v[q]);
          const f32x4_t w0 = *reinterpret_cast<const f32x4_t*>(wsm + 4 * (8 * c + 2 * q));
          const f32x4_t w1 = *reinterpret_cast<const f32x4_t*>(wsm + 4 * (8 * c + 2 * q + 1));
#pragma unroll
          for (int o = 0; o < 4; o++) a[o] = fmaf(hi, w1[o], fmaf(lo, w0[o], a[o]));
        }
      }
      float dp[4] = {0.f, 0.f, 0.f, 0.f};
      for (int o = 0; o < Cout; o++) {
        const float pr = keras_f16_point<T>(a[o] + (bias ? bias[o] : 0.f));
        const float d = pr - target[(size_t)m * Cout + o];
        if (pred_out) pred_out[(size_t)m * Cout + o] = pr;
        lacc = fmaf(d, d, lacc);
        dp[o] = keras_f16_point<T>(d * gscale);
      }
      *reinterpret_cast<f32x4_t*>(dps + 4 * tid) = f32x4_t{dp[0], dp[1], dp[2], dp[3]};
      u32x4_t* drow = reinterpret_cast<u32x4_t*>(dxs + tid * Cmask);
      for (int c = 0; c < (Cmask >> 3); c++) {
        const u32x4_t v = row[c];
        u32x4_t g4;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const f32x4_t w0 = *reinterpret_cast<const f32x4_t*>(wsm + 4 * (8 * c + 2 * q));
          const f32x4_t w1 = *reinterpret_cast<const f32x4_t*>(wsm + 4 * (8 * c + 2 * q + 1));
          float g0 = dp[0] * w0[0] + dp[1] * w0[1] + dp[2] * w0[2] + dp[3] * w0[3];
          float g1 = dp[0] * w1[0] + dp[1] * w1[1] + dp[2] * w1[2] + dp[3] * w1[3];
          if (!(unpack_lo<T>(v[q]) > 0.f)) g0 = 0.f;
          if (!(unpack_hi<T>(v[q]) > 0.f)) g1 = 0.f;
          g4[q] = pack2<T>(g0, g1);
        }
        drow[c] = g4;
      }
    } else if (tid < PIX) {
      *reinterpret_cast<f32x4_t*>(dps + 4 * tid) = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    {   // gradient rows out: Cmask/8 chunks per pixel
      const int cpr = Cmask >> 3;
      for (int i = tid; i < npx * cpr; i += 256) {
        const int pm = i / cpr, c = i - pm * cpr;
        *reinterpret_cast<u32x4_t*>(dx + (size_t)(mbase + pm) * lddx + c * 8) = reinterpret_cast<const u32x4_t*>(dxs)[i];
      }
    }
    if (tid < nout) {
      if (my_i < Cin) {
        for (int pm = 0; pm < npx; pm++) wacc = fmaf(to_f32(xs[pm * ld + my_i]), dps[4 * pm + my_o], wacc);
      } else {
        for (int pm = 0; pm < npx; pm++) wacc += dps[4 * pm + my_o];
      }
    }
    if (db_dx && tid >= 256 - Cmask) {           // the last Cmask threads (idle above) sum the stored gradient rows
      const int c = tid - (256 - Cmask);
      for (int pm = 0; pm < npx; pm++) bacc += to_f32(dxs[pm * Cmask + c]);
    }
  }
  if (tid < nout) {
    if (my_i < Cin) atomicAdd(dw + my_i * Cout + my_o, wacc);
    else if (db) atomicAdd(db + my_o, wacc);
  }
  if (db_dx && tid >= 256 - Cmask) atomicAdd(db_dx + (tid - (256 - Cmask)), bacc);
  for (int off = 32; off > 0; off >>= 1) lacc += __shfl_down(lacc, off, 64);
  __shared__ float lws[4];
  if ((tid & 63) == 0) lws[tid >> 6] = lacc;
  __syncthreads();
  if (tid == 0) partials[blockIdx.x] = lws[0] + lws[1] + lws[2] + lws[3];
}

// ---- the same fused head on the matrix cores ------------------------------------------------------------------------------
// One wave = 16 pixels per trip, operands straight from global memory in MFMA fragment layout (no LDS staging):
//   pred[o][pix]  = sum_c W[c][o] x[pix][c]         A = W^T (the fp32 kernel as a three-term 16-bit sum, rows o < Cout), B = x rows
//   g[c][pix]     = sum_o W[c][o] d[pix][o]         A = W rows in the permuted order below, B = d = pred - target (hi + lo)
// so the 16-bit operands carry the fp32 kernel and the fp32 residual as multi-term sums (forward ~fp32; backward ~2^-17, far
// below the 16-bit rounding of the stored gradient).
// Output row rho = 4g + r of backward MFMA ct is channel 32 (ct>>1) + 8 g + 4 (ct&1) + r: exactly the channels whose x values
// this lane already holds as its forward B fragments, so the ReLU mask and the 16-byte gradient store need no shuffle.
// dW / db / db_dx / loss are accumulated per lane and leave as one partial row per work-group (HEAD_ROW floats) for
// dense_head_finish_kernel: no atomics, fixed summation order.
// partial row layout HEAD_ROW: gct2_common.h
template <typename T>
__global__ __launch_bounds__(256, 2) void dense_head_mfma_kernel(const T* __restrict__ x, int ld, const float* __restrict__ w,
                                                                 const float* __restrict__ bias, const float* __restrict__ target,
                                                                 float* __restrict__ pred_out, T* __restrict__ dx, int lddx,
                                                                 float* __restrict__ part, int M, int Cin, int Cout,
                                                                 const float* __restrict__ loss_scale_ptr,
                                                                 const T* __restrict__ x2, int ldx2) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, q = lane & 15;
  const float gscale = (loss_scale_ptr ? *loss_scale_ptr : 1.f) * 2.0f / ((float)M * (float)Cout);
  const float inv_gscale = 1.0f / gscale;
  auto split = [](float v, T& hi, T& lo) { hi = from_f32<T>(v); lo = from_f32<T>(v - to_f32(hi)); };
  auto pk = [](T a, T b) { return (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16); };
  // forward A operands: row o = q, reduction slots = channels 32 kk + 8 g + j
  u32x4_t a_f[3][3];                                            // three-term split: pred keeps fp32 accuracy
#pragma unroll
  for (int kk = 0; kk < 3; kk++) {
    T hi[8], mid[8], lo[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int c = 32 * kk + 8 * g + j;
      const float v = (q < Cout && c < Cin && (kk < 2 || g == 0)) ? w[c * Cout + q] : 0.f;
      split(v, hi[j], mid[j]);
      const float rest = (v - to_f32(hi[j])) - to_f32(mid[j]);
      lo[j] = from_f32<T>(rest);
    }
    a_f[kk][0] = u32x4_t{pk(hi[0], hi[1]), pk(hi[2], hi[3]), pk(hi[4], hi[5]), pk(hi[6], hi[7])};
    a_f[kk][1] = u32x4_t{pk(mid[0], mid[1]), pk(mid[2], mid[3]), pk(mid[4], mid[5]), pk(mid[6], mid[7])};
    a_f[kk][2] = u32x4_t{pk(lo[0], lo[1]), pk(lo[2], lo[3]), pk(lo[4], lo[5]), pk(lo[6], lo[7])};
  }
  // backward A operands (lanes g == 0 only): row rho = q -> channel ch; slots {Whi0..2, Whi0..2, 0, 0} and {Wlo0..2, 0...}
  u32x4_t a_b[4][2];
#pragma unroll
  for (int ct = 0; ct < 4; ct++) {
    const int ch = 32 * (ct >> 1) + 8 * (q >> 2) + 4 * (ct & 1) + (q & 3);
    T hi[3], lo[3];
#pragma unroll
    for (int o = 0; o < 3; o++) split((g == 0 && o < Cout) ? w[ch * Cout + o] : 0.f, hi[o], lo[o]);
    const T z = from_f32<T>(0.f);
    a_b[ct][0] = u32x4_t{pk(hi[0], hi[1]), pk(hi[2], hi[0]), pk(hi[1], hi[2]), 0u};
    a_b[ct][1] = u32x4_t{pk(lo[0], lo[1]), pk(lo[2], z), 0u, 0u};
  }
  float bv[3];
#pragma unroll
  for (int o = 0; o < 3; o++) bv[o] = (bias && o < Cout) ? bias[o] : 0.f;
  const uint32_t tail_keep = Cin - 64;                         // valid channels in the third fragment (pad channels may hold anything)

  float wacc[3][8][3];
#pragma unroll
  for (int kk = 0; kk < 3; kk++)
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
      for (int o = 0; o < 3; o++) wacc[kk][j][o] = 0.f;
  float bacc[2][8];
#pragma unroll
  for (int kk = 0; kk < 2; kk++)
#pragma unroll
    for (int j = 0; j < 8; j++) bacc[kk][j] = 0.f;
  float dbacc[3] = {0.f, 0.f, 0.f}, lacc = 0.f;

  const int ngroups = (M + 15) >> 4;
  const int gstride = gridDim.x * 4;
  auto load = [&](int grp, u32x4_t (&xf)[3]) {
    const int p = grp * 16 + q;
    xf[0] = xf[1] = xf[2] = u32x4_t{0u, 0u, 0u, 0u};
    if (grp < ngroups && p < M) {
      const T* row = x + (size_t)p * ld + 8 * g;
      xf[0] = gload128(row);
      xf[1] = gload128(row + 32);
      if (g == 0) {
        if (x2) {   // channels 64.. live in a packed side buffer (<= 4 of them, 8-byte aligned rows)
          const u32x2_t v2 = *reinterpret_cast<const u32x2_t*>(x2 + (size_t)p * ldx2);
          xf[2] = u32x4_t{v2[0], v2[1], 0u, 0u};
        } else xf[2] = gload128(row + 64);
      }
    }
  };
  u32x4_t cur[3], nxt[3];
  int grp = blockIdx.x * 4 + wave;
  load(grp, cur);
  for (; grp < ngroups; grp += gstride) {
    load(grp + gstride, nxt);
    const int p = grp * 16 + q;
    const bool pv = p < M;
    if (g == 0) {                                              // zero the pad channels >= Cin of the third fragment
#pragma unroll
      for (int j2 = 0; j2 < 4; j2++) {
        const uint32_t keep = tail_keep >= 2u * j2 + 2u ? 0xffffffffu : (tail_keep == 2u * j2 + 1u ? 0xffffu : 0u);
        cur[2][j2] &= keep;
      }
    }
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 3; kk++) {
      acc = mfma16<T>(a_f[kk][2], cur[kk], acc);            // smallest terms first
      acc = mfma16<T>(a_f[kk][1], cur[kk], acc);
      acc = mfma16<T>(a_f[kk][0], cur[kk], acc);
    }
    // lanes g == 0: acc[o] = pred[pixel q][o]
    float d[3] = {0.f, 0.f, 0.f};
    if (g == 0 && pv) {
#pragma unroll
      for (int o = 0; o < 3; o++) {
        if (o < Cout) {
          const float pr = keras_f16_point<T>(acc[o] + bv[o]);
          d[o] = pr - target[(size_t)p * Cout + o];
          if (pred_out) pred_out[(size_t)p * Cout + o] = pr;
          lacc = fmaf(d[o], d[o], lacc);
        }
      }
    }
    // the gradient entering the Dense output, dsc = d * gscale (an fp16 tensor under mixed_float16: keras_f16_point); the backward
    // MFMAs take it unscaled (dsc / gscale, = d itself when nothing is rounded) and multiply by gscale afterwards
    T dh[3], dl[3];
    float dsc[3];
#pragma unroll
    for (int o = 0; o < 3; o++) {
      dsc[o] = keras_f16_point<T>(d[o] * gscale);
      constexpr bool kF16 = sizeof(T) == 2 && !__is_same(T, __bf16);
      split(kF16 ? dsc[o] * inv_gscale : d[o], dh[o], dl[o]);
    }
    const u32x4_t bb = {pk(dh[0], dh[1]), pk(dh[2], dl[0]), pk(dl[1], dl[2]), 0u};   // zero in lanes g > 0 (d = 0 there)
    float dp[3];
#pragma unroll
    for (int o = 0; o < 3; o++) {
      dp[o] = __shfl(dsc[o], q, 64);                           // every lane group gets the gradient of its pixel
      if (g == 0) dbacc[o] += dp[o];
    }
    u32x4_t gout[2];
#pragma unroll
    for (int ct = 0; ct < 4; ct++) {
      f32x4_t gd = {0.f, 0.f, 0.f, 0.f};
      gd = mfma16<T>(a_b[ct][0], bb, gd);
      gd = mfma16<T>(a_b[ct][1], bb, gd);
      const int kk = ct >> 1, h = ct & 1;
      float gv[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const uint32_t xw = cur[kk][2 * h + (r >> 1)];
        const float xv = (r & 1) ? unpack_hi<T>(xw) : unpack_lo<T>(xw);
        gv[r] = xv > 0.f ? gd[r] * gscale : 0.f;
        bacc[kk][4 * h + r] += gv[r];
      }
      gout[kk][2 * h] = pack2<T>(gv[0], gv[1]);
      gout[kk][2 * h + 1] = pack2<T>(gv[2], gv[3]);
    }
    if (pv) {
      T* drow = dx + (size_t)p * lddx + 8 * g;
      *reinterpret_cast<u32x4_t*>(drow) = gout[0];
      *reinterpret_cast<u32x4_t*>(drow + 32) = gout[1];
    }
#pragma unroll
    for (int kk = 0; kk < 3; kk++)
#pragma unroll
      for (int j2 = 0; j2 < 4; j2++) {
        const float x0 = unpack_lo<T>(cur[kk][j2]), x1 = unpack_hi<T>(cur[kk][j2]);
#pragma unroll
        for (int o = 0; o < 3; o++) {
          wacc[kk][2 * j2][o] = fmaf(x0, dp[o], wacc[kk][2 * j2][o]);
          wacc[kk][2 * j2 + 1][o] = fmaf(x1, dp[o], wacc[kk][2 * j2 + 1][o]);
        }
      }
    cur[0] = nxt[0]; cur[1] = nxt[1]; cur[2] = nxt[2];
  }

  // ---- reductions: over the 16 pixel lanes (butterfly), then over the 4 waves (LDS), one partial row per work-group ----
  __shared__ float red[4][HEAD_ROW];
  for (int i = tid; i < 4 * HEAD_ROW; i += 256) (&red[0][0])[i] = 0.f;
  __syncthreads();
  auto bfly = [](float t) {
    t = row16_sum(t);
    return t;
  };
#pragma unroll
  for (int kk = 0; kk < 3; kk++)
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
      for (int o = 0; o < 3; o++) {
        const float t = bfly(wacc[kk][j][o]);
        const int c = 32 * kk + 8 * g + j;
        if (q == 0 && o < Cout && c < Cin && (kk < 2 || g == 0)) red[wave][c * Cout + o] = t;
      }
#pragma unroll
  for (int kk = 0; kk < 2; kk++)
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const float t = bfly(bacc[kk][j]);
      if (q == 0) red[wave][224 + 32 * kk + 8 * g + j] = t;
    }
#pragma unroll
  for (int o = 0; o < 3; o++) {
    const float t = bfly(dbacc[o]);
    if (lane == 0) red[wave][216 + o] = t;
  }
  {
    const float t = bfly(lacc);
    if (lane == 0) red[wave][219] = t;
  }
  __syncthreads();
  for (int i = tid; i < HEAD_ROW; i += 256) part[(size_t)blockIdx.x * HEAD_ROW + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
}

// sums the head's partial rows: column -> dW / db / loss / db of the layer below.  9 work-groups x (32 columns x 32 row lanes).
__global__ __launch_bounds__(1024) void dense_head_finish_kernel(const float* __restrict__ part, int rows, float* __restrict__ dw,
                                                                 float* __restrict__ db, float* __restrict__ loss,
                                                                 float* __restrict__ db_dx, int ndw, int Cout, float inv_n,
                                                                 int accumulate) {
  const int tid = threadIdx.x, cx = tid & 31, rl = tid >> 5;
  const int c = blockIdx.x * 32 + cx;
  double acc = 0.0;
  int r = rl;
  for (; r + 7 * 32 < rows; r += 8 * 32) {        // 8 independent loads in flight, added in row order (same sums as one at a time)
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; u++) t[u] = part[(size_t)(r + 32 * u) * HEAD_ROW + c];
#pragma unroll
    for (int u = 0; u < 8; u++) acc += (double)t[u];
  }
  for (; r < rows; r += 32) acc += (double)part[(size_t)r * HEAD_ROW + c];
  __shared__ double red[32][33];
  red[rl][cx] = acc;
  __syncthreads();
  if (rl == 0) {
    double t = 0.0;
    for (int k = 0; k < 32; k++) t += red[k][cx];
    const float tf = (float)t;
    if (c < ndw) dw[c] = accumulate ? dw[c] + tf : tf;
    else if (c >= 216 && c < 216 + Cout) { if (db) db[c - 216] = accumulate ? db[c - 216] + tf : tf; }
    else if (c == 219) *loss = (float)(t * (double)inv_n);
    else if (c >= 224 && db_dx) db_dx[c - 224] = accumulate ? db_dx[c - 224] + tf : tf;
  }
}

// ---- log_sample (train.py:323-496): the sampler's pointwise steps, fp32 state ------------------------------------------------
// fake = sqrt(a) x_theta + sqrt(1-a) eps_theta (train.py:372-375, 441-444); also stored in the compute dtype where the network
// reads its input (packed image and/or the image slice of R_0)
template <typename T>
__global__ void diffusion_mix_kernel(const float* __restrict__ x, const float* __restrict__ e, float sa, float sb,
                                     float* __restrict__ fake, T* __restrict__ out, int ldout, T* __restrict__ out2, int ldout2,
                                     size_t n, int C) {
  // plain IEEE multiplies and adds in the order of the formula, no FMA contraction: what an unfused TensorFlow op chain computes
  // (train.py:372-375), and independent of the code around it (as for adam_keras_update)
#pragma clang fp contract(off)
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float f = sa * x[i] + sb * e[i];
    fake[i] = f;
    const size_t pix = i / C;
    const int c = (int)(i - pix * C);
    out[pix * ldout + c] = from_f32<T>(f);
    if (out2) out2[pix * ldout2 + c] = from_f32<T>(f);
  }
}
// one sampler update from the prediction, by objective (train.py:338-355, 382-413, 452-479; modes of include/gct2.h):
//   X: x = pred, e = (fake - sa x) / sb;  EPS: e = pred, x = (fake - pred sb) / sa;  SCALED_EPS: e = pred / sb, x = (fake - pred) / sa;
//   ODE: x = (pred sb - fake sb1) / (sa1 sb - sa sb1), e untouched.  sa = sqrt(a_t), sb = sqrt(1 - a_t), sa1 / sb1 the same at t - 1.
template <int MODE>
__global__ void diffusion_update_kernel(const float* __restrict__ pred, const float* __restrict__ fake, float sa, float sb, float sb1, float den,
                                        float* __restrict__ x, float* __restrict__ e, size_t n) {
#pragma clang fp contract(off)          // (f - sa p) / sb etc. as separate roundings, like the reference's op chain (train.py:382-413)
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float p = pred[i], f = fake[i];
    if (MODE == GCT2_SAMPLE_X) {
      x[i] = p;
      e[i] = (f - sa * p) / sb;
    } else if (MODE == GCT2_SAMPLE_EPS) {
      e[i] = p;
      x[i] = (f - p * sb) / sa;
    } else if (MODE == GCT2_SAMPLE_SCALED_EPS) {
      e[i] = p / sb;
      x[i] = (f - p) / sa;
    } else {
      x[i] = (p * sb - f * sb1) / den;
    }
  }
}
// the four noise variants of train.py:416-431 from one eps image [H,W,C]: out[0] = eps, out[1] = nearest-upsample x4 of the
// 4x4 average pool, out[2] = rolled by one pixel along H and W, out[3] = per-pixel nearest entry of dictionary [H,W,K,C]
__global__ void noise_edits_kernel(const float* __restrict__ eps, const float* __restrict__ dict, int K, float* __restrict__ out,
                                   int H, int W, int C) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= H * W) return;
  const int h = idx / W, w = idx - h * W;
  const size_t img = (size_t)H * W * C;
  const float* px = eps + (size_t)idx * C;
  const int hs = (h + H - 1) % H, wsft = (w + W - 1) % W;     // tf.roll(x, 1, axis): out[i] = in[i - 1]
  const int h0 = h & ~3, w0 = w & ~3;
  for (int c = 0; c < C; c++) {
    out[(size_t)idx * C + c] = px[c];
    float a = 0.f;
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) a += eps[((size_t)(h0 + i) * W + (w0 + j)) * C + c];
    out[img + (size_t)idx * C + c] = a * (1.0f / 16.0f);
    out[2 * img + (size_t)idx * C + c] = eps[((size_t)hs * W + wsft) * C + c];
  }
  int best = 0;
  float bestd = 0.f;
  for (int k = 0; k < K; k++) {
    float d = 0.f;
    for (int c = 0; c < C; c++) {
      const float t = px[c] - dict[((size_t)idx * K + k) * C + c];
      d += t * t;
    }
    if (k == 0 || d < bestd) { bestd = d; best = k; }          // first minimum, like tf.argmin
  }
  for (int c = 0; c < C; c++) out[3 * img + (size_t)idx * C + c] = dict[((size_t)idx * K + best) * C + c];
}

// ---- input contract (train.py:285-293): random crop, random left-right flip, u8 -> value/128 - 1, one launch per batch ------
// src: concatenated decoded RGB images (u8, HWC); per image b: byte offset, original height/width, crop origin, flip flag
__global__ void image_prepare_kernel(const uint8_t* __restrict__ src, const int64_t* __restrict__ offsets, const int32_t* __restrict__ dims,
                                     float* __restrict__ dst, int B, int size) {
  const size_t per = (size_t)size * size * 3;
  const size_t total = per * B;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int b = (int)(i / per);
    const int r = (int)(i - (size_t)b * per);
    const int c = r % 3, xy = r / 3, x = xy % size, y = xy / size;
    const int W0 = dims[5 * b + 1], oy = dims[5 * b + 2], ox = dims[5 * b + 3], flip = dims[5 * b + 4];
    const int sx = ox + (flip ? size - 1 - x : x);
    const uint8_t v = src[offsets[b] + ((size_t)(oy + y) * W0 + sx) * 3 + c];
    dst[i] = (float)v * (1.0f / 128.0f) - 1.0f;                  // exact: v/128 has at most 8 significant bits
  }
}

// ---- small helpers of the off-by-default model variants (train.py:106-112 residual add, autodiff of the fused ReLUs,
// train.py:238-252 targets): elementwise, views with a pixel stride ----------------------------------------------------------
template <typename T>
__global__ void relu_mask_kernel(const T* __restrict__ act, int ldact, T* __restrict__ d, int ldd, size_t n, int C) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const size_t pix = i / C;
    const int c = (int)(i - pix * C);
    if (!(to_f32(act[pix * ldact + c]) > 0.f)) d[pix * ldd + c] = from_f32<T>(0.f);
  }
}
template <typename T>
__global__ void add_kernel(T* __restrict__ dst, int lddst, const T* __restrict__ src, int ldsrc, size_t n, int C) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const size_t pix = i / C;
    const int c = (int)(i - pix * C);
    dst[pix * lddst + c] = from_f32<T>(to_f32(dst[pix * lddst + c]) + to_f32(src[pix * ldsrc + c]));
  }
}
// out = a[b] * x + c[b] * eps (eps may be null: out = a[b] * x), fp32, per-image coefficients
__global__ void mix_per_image_kernel(const float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ a,
                                     const float* __restrict__ c, float* __restrict__ out, size_t n, size_t per_image) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const size_t b = i / per_image;
    out[i] = a[b] * x[i] + (eps ? c[b] * eps[i] : 0.f);
  }
}

// ---- bias gradient: db[c] += sum_m dz[m][c] --------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ dz, int ld, float* __restrict__ db, size_t M, int C,
                                                     int rows_per_block, float sign) {
  // thread = (channel within a 64-wide tile, one of 4 row lanes)
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const size_t r0 = (size_t)blockIdx.y * rows_per_block;
  const size_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  float acc = 0.f;
  if (c < C)
    for (size_t r = r0 + rl; r < r1; r += 4) acc += to_f32(dz[r * ld + c]);
  __shared__ float ws[256];
  ws[threadIdx.x] = acc;
  __syncthreads();
  if (rl == 0 && c < C) atomicAdd(db + c, sign * (ws[threadIdx.x] + ws[threadIdx.x + 64] + ws[threadIdx.x + 128] + ws[threadIdx.x + 192]));
}
// 16-bit types, 16-byte loads: thread = (8-channel chunk of a 64-wide tile, one of 32 row lanes)
template <typename T>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const T* __restrict__ dz, int ld, float* __restrict__ db, size_t M, int C,
                                                         int rows_per_block, float sign) {
  const int c8 = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c = blockIdx.x * 64 + c8 * 8;
  const size_t r0 = (size_t)blockIdx.y * rows_per_block;
  const size_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    for (size_t r = r0 + rl; r < r1; r += 32) {
      const u32x4_t v = gload128(dz + r * ld + c);
#pragma unroll
      for (int k = 0; k < 4; k++) { acc[2 * k] += unpack_lo<T>(v[k]); acc[2 * k + 1] += unpack_hi<T>(v[k]); }
    }
  }
  __shared__ float ws[32][65];
#pragma unroll
  for (int k = 0; k < 8; k++) ws[rl][c8 * 8 + k] = acc[k];
  __syncthreads();
  if (threadIdx.x < 64) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 32; r++) s += ws[r][threadIdx.x];
    const int cc = blockIdx.x * 64 + threadIdx.x;
    if (cc < C) atomicAdd(db + cc, sign * s);
  }
}

// ---- Keras Adam over a flat arena ------------------------------------------------------------------------
template <typename S, bool HAS_SHADOW>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                   float* __restrict__ g, S* __restrict__ shadow, size_t n, float alpha,
                                                   float b1, float b2, float eps, float grad_mul,
                                                   const gct2_loss_scale_state* __restrict__ ls, int zero_grad,
                                                   const float* __restrict__ slabs, int nslab, size_t slab_stride, size_t n_slab) {
  // slabs != NULL: the gradient of the first n_slab elements (a multiple of 4) is the ordered sum of `nslab` partial slabs
  // (slab s at slabs + s * slab_stride) left by a weight-gradient launch - it is never written to or read from g.
  // ls != NULL (LossScaleOptimizer): unscale, skip on non-finite gradients, alpha of the device-side step counter.
  const bool skip = ls && ls->found_inf != 0;
  const float inv_scale = (ls ? ls->inv_scale : 1.f) * grad_mul;
  if (ls) alpha = ls->alpha;
  const size_t n4 = n >> 2;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const float ob1 = 1.f - b1, ob2 = 1.f - b2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4_t gv;
    if (slabs && (i << 2) < n_slab) {
      const f32x4_t* src = reinterpret_cast<const f32x4_t*>(slabs) + i;
      const size_t st4 = slab_stride >> 2;
      gv = __builtin_nontemporal_load(src);
      int sidx = 1;
      for (; sidx + 8 <= nslab; sidx += 8) {     // 8 independent slab loads in flight, added in slab order
        f32x4_t t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) t[u] = __builtin_nontemporal_load(src + (size_t)(sidx + u) * st4);
#pragma unroll
        for (int u = 0; u < 8; u++) gv += t[u];
      }
      for (; sidx < nslab; sidx++) gv += __builtin_nontemporal_load(src + (size_t)sidx * st4);
    } else {
      gv = reinterpret_cast<f32x4_t*>(g)[i];
      if (zero_grad) reinterpret_cast<f32x4_t*>(g)[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    if (skip) continue;
    // everything here is touched once per step: streaming (nt) accesses keep it from displacing the GEMM operands in the L2s
    f32x4_t pv = __builtin_nontemporal_load(reinterpret_cast<f32x4_t*>(p) + i), mv = __builtin_nontemporal_load(reinterpret_cast<f32x4_t*>(m) + i),
            vv = __builtin_nontemporal_load(reinterpret_cast<f32x4_t*>(v) + i);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      float pp = pv[k], mm = mv[k], v1 = vv[k];
      adam_keras_update(pp, mm, v1, gv[k] * inv_scale, alpha, b1, ob1, b2, ob2, eps);
      pv[k] = pp; mv[k] = mm; vv[k] = v1;
    }
    __builtin_nontemporal_store(pv, reinterpret_cast<f32x4_t*>(p) + i);
    __builtin_nontemporal_store(mv, reinterpret_cast<f32x4_t*>(m) + i);
    __builtin_nontemporal_store(vv, reinterpret_cast<f32x4_t*>(v) + i);
    if (HAS_SHADOW) {
      if constexpr (sizeof(S) == 2) {
        u32x2_t o = {pack2<S>(pv[0], pv[1]), pack2<S>(pv[2], pv[3])};
        reinterpret_cast<u32x2_t*>(shadow)[i] = o;
      } else {
        reinterpret_cast<f32x4_t*>(shadow)[i] = pv;
      }
    }
  }
  // tail (n % 4 elements)
  const size_t i = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float gg = g[i] * inv_scale;
    if (zero_grad) g[i] = 0.f;
    if (!skip) {
      float mm = m[i], vv = v[i], pp = p[i];
      adam_keras_update(pp, mm, vv, gg, alpha, b1, ob1, b2, ob2, eps);
      m[i] = mm; v[i] = vv; p[i] = pp;
      if (HAS_SHADOW) shadow[i] = from_f32<S>(pp);
    }
  }
}

template <typename S>
__global__ void cast_kernel(const float* __restrict__ src, S* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = from_f32<S>(src[i]);
}

// ---- loss scale ------------------------------------------------------------------------------------------
__global__ void ls_init_kernel(gct2_loss_scale_state* s, float scale) {
  s->scale = scale; s->inv_scale = 1.f / scale; s->good_steps = 0; s->found_inf = 0;
  s->applied_steps = 0; s->alpha = 0.f; s->reserved[0] = s->reserved[1] = 0;
}
// found_inf = 0; alpha of THIS step: WarmUp (train.py:57-65, float32 like the reference) x Adam's bias correction [TF], with
// k = applied_steps = optimizer.iterations (a skipped step does not advance it)
__global__ void ls_begin_kernel(gct2_loss_scale_state* s, float base_lr, int warmup_steps, float b1, float b2) {
  s->found_inf = 0;
  const int k = s->applied_steps;
  const float lr = k < warmup_steps ? base_lr * (float)(k + 1) / (float)(warmup_steps + 1) : base_lr;
  const double t = (double)(k + 1);
  s->alpha = (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
}
__global__ void ls_check_kernel(const float* __restrict__ g, size_t n, gct2_loss_scale_state* s) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float x = g[i];
    bad |= !(fabsf(x) <= 3.402823466e38f);      // false for inf and nan
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(&s->found_inf, 1);
}
__global__ void ls_update_kernel(gct2_loss_scale_state* s, int growth_interval) {
  if (s->found_inf) {
    s->scale = fmaxf(s->scale * 0.5f, 1.f); s->good_steps = 0;
  } else {
    s->applied_steps += 1;
    s->good_steps += 1;
    if (s->good_steps >= growth_interval) { s->scale *= 2.f; s->good_steps = 0; }
  }
  s->inv_scale = 1.f / s->scale;
}

inline int blocks_for(size_t n, int per_block) {
  size_t b = (n + per_block - 1) / per_block;
  return (int)(b < 1 ? 1 : (b > (size_t)kMaxBlocks ? kMaxBlocks : b));
}

}  // namespace

// ---- host launchers (called from capi.hip) ------------------------------------------------------------------
int pw_rng_uniform_int(uint64_t seed, uint64_t stream_id, uint64_t offset, int32_t* out, size_t n, int lo, int hi, hipStream_t s) {
  if (n == 0) return GCT2_OK;
  hipLaunchKernelGGL(rng_uniform_int_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, seed, stream_id, offset, out, n, lo,
                     (uint32_t)(hi - lo + 1));
  return gct2_check_launch("rng_uniform_int");
}
int pw_rng_normal(uint64_t seed, uint64_t stream_id, uint64_t offset, float* out, size_t n, hipStream_t s) {
  if (n == 0) return GCT2_OK;
  hipLaunchKernelGGL(rng_normal_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, seed, stream_id, offset, out, n);
  return gct2_check_launch("rng_normal");
}
template <typename T>
static int noise_t(const float* x, const int32_t* t, const float* eps, void* out, int ldout, void* out2, int ldout2, int B, int HW, int C,
                   int steps, hipStream_t s) {
  const size_t npix = (size_t)B * HW;
  hipLaunchKernelGGL(noise_kernel<T>, dim3(blocks_for(npix * C, 256)), dim3(256), 0, s, x, t, eps, reinterpret_cast<T*>(out), ldout,
                     reinterpret_cast<T*>(out2), ldout2, npix, HW, C, steps + 1);
  return gct2_check_launch("noise_image");
}
int pw_noise(int dtype, const float* x, const int32_t* t, const float* eps, void* out, int ldout, void* out2, int ldout2, int B, int HW,
             int C, int steps, hipStream_t s) {
  if (dtype == GCT2_F32) return noise_t<float>(x, t, eps, out, ldout, out2, ldout2, B, HW, C, steps, s);
  if (dtype == GCT2_BF16) return noise_t<__bf16>(x, t, eps, out, ldout, out2, ldout2, B, HW, C, steps, s);
  return noise_t<_Float16>(x, t, eps, out, ldout, out2, ldout2, B, HW, C, steps, s);
}
template <typename T>
static int noise_rng_t(const float* x, const int32_t* t, uint64_t seed, uint64_t sid, uint64_t off, float* eps_out, void* out, int ldout,
                       void* out2, int ldout2, int B, int HW, int C, int steps, hipStream_t s) {
  const size_t npix = (size_t)B * HW;
  if constexpr (sizeof(T) == 2) {
    // the train step's call: 3 channels into the packed 4-slot image and nothing else, whole groups of four pixels / three counters
    if (C == 3 && ldout == 4 && !out2 && !eps_out && off % 12 == 0 && npix % 4 == 0 && HW % 4 == 0 && (uintptr_t)x % 16 == 0 &&
        (uintptr_t)out % 16 == 0) {
      hipLaunchKernelGGL(noise_rng_px4_kernel<T>, dim3(blocks_for(npix / 4, 256)), dim3(256), 0, s, x, t, seed, sid, off >> 2,
                         reinterpret_cast<T*>(out), npix / 4, HW, steps + 1);
      return gct2_check_launch("noise_image_rng");
    }
  }
  hipLaunchKernelGGL(noise_rng_kernel<T>, dim3(blocks_for(npix * C / 4 + 2, 256)), dim3(256), 0, s, x, t, seed, sid, off, eps_out,
                     reinterpret_cast<T*>(out), ldout, reinterpret_cast<T*>(out2), ldout2, npix * C, HW, C, steps + 1);
  return gct2_check_launch("noise_image_rng");
}
int pw_noise_rng(int dtype, const float* x, const int32_t* t, uint64_t seed, uint64_t sid, uint64_t off, float* eps_out, void* out,
                 int ldout, void* out2, int ldout2, int B, int HW, int C, int steps, hipStream_t s) {
  if (dtype == GCT2_F32) return noise_rng_t<float>(x, t, seed, sid, off, eps_out, out, ldout, out2, ldout2, B, HW, C, steps, s);
  if (dtype == GCT2_BF16) return noise_rng_t<__bf16>(x, t, seed, sid, off, eps_out, out, ldout, out2, ldout2, B, HW, C, steps, s);
  return noise_rng_t<_Float16>(x, t, seed, sid, off, eps_out, out, ldout, out2, ldout2, B, HW, C, steps, s);
}
template <typename T>
static int dense_fwd_t(const void* x, int ldx, const float* w, const float* b, float* y, int M, int Cin, int Cout, hipStream_t s) {
  hipLaunchKernelGGL(dense_fwd_kernel<T>, dim3(blocks_for(M, 256)), dim3(256), Cin * 4 * sizeof(float), s, reinterpret_cast<const T*>(x), ldx,
                     w, b, y, M, Cin, Cout);
  return gct2_check_launch("dense_fwd");
}
int pw_dense_fwd(int dtype, const void* x, int ldx, const float* w, const float* b, float* y, int M, int Cin, int Cout, hipStream_t s) {
  if (dtype == GCT2_F32) return dense_fwd_t<float>(x, ldx, w, b, y, M, Cin, Cout, s);
  if (dtype == GCT2_BF16) return dense_fwd_t<__bf16>(x, ldx, w, b, y, M, Cin, Cout, s);
  return dense_fwd_t<_Float16>(x, ldx, w, b, y, M, Cin, Cout, s);
}
template <typename T>
static int dense_bwd_t(const void* x, int ldx, const float* w, const float* dy, void* dx, int lddx, float* dw, float* db, int M, int Cin,
                       int Cout, int Cmask, int accumulate, hipStream_t s) {
  constexpr int PIX = 128;
  if (!accumulate) {            // the kernel adds its per-work-group sums with atomics
    (void)hipMemsetAsync(dw, 0, (size_t)Cin * Cout * sizeof(float), s);
    if (db) (void)hipMemsetAsync(db, 0, (size_t)Cout * sizeof(float), s);
  }
  const size_t lds = (size_t)Cin * 16 + PIX * 16 + (size_t)PIX * Cin * sizeof(T);
  const int ntiles = (M + PIX - 1) / PIX;
  const int grid = ntiles < 1024 ? ntiles : 1024;
  auto kern = dense_bwd_kernel<T, PIX>;
  if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, reinterpret_cast<const T*>(x), ldx, w, dy, reinterpret_cast<T*>(dx), lddx, dw, db,
                     M, Cin, Cout, Cmask);
  return gct2_check_launch("dense_bwd");
}
int pw_dense_bwd(int dtype, const void* x, int ldx, const float* w, const float* dy, void* dx, int lddx, float* dw, float* db, int M,
                 int Cin, int Cout, int Cmask, int accumulate, hipStream_t s) {
  if (dtype == GCT2_F32) return dense_bwd_t<float>(x, ldx, w, dy, dx, lddx, dw, db, M, Cin, Cout, Cmask, accumulate, s);
  if (dtype == GCT2_BF16) return dense_bwd_t<__bf16>(x, ldx, w, dy, dx, lddx, dw, db, M, Cin, Cout, Cmask, accumulate, s);
  return dense_bwd_t<_Float16>(x, ldx, w, dy, dx, lddx, dw, db, M, Cin, Cout, Cmask, accumulate, s);
}
// the ordered finish of the head's partial rows, also used by the UpShuffle_0 forward that carries the head in its epilogue
int pw_head_finish(const float* part, int rows, float* dw, float* db, float* loss, float* db_dx, int ndw, int Cout, float inv_n,
                   int accumulate, hipStream_t s) {
  hipLaunchKernelGGL(dense_head_finish_kernel, dim3(HEAD_ROW / 32), dim3(1024), 0, s, part, rows, dw, db, loss, db_dx, ndw, Cout, inv_n,
                     accumulate);
  return gct2_check_launch("dense_head_finish");
}

template <typename T>
static int dense_head_train_t(const gct2_ctx& c, const void* x, int ld, const float* w, const float* b, const float* target, float* pred,
                              void* dx, int lddx, float* dw, float* db, float* loss, float* partials, int M, int Cin, int Cout, int Cmask,
                              const float* ls, float* db_dx, const void* x2, int ldx2, int accumulate, hipStream_t s) {
  // matrix-core version: the reference head (64 masked U_0 channels + 3 image channels -> 3 outputs) with a workspace in the ctx
  const size_t ws_bytes = c.ws_bytes;
  float* ws = c.ws;
  if (Cmask == 64 && Cin >= 64 && Cin <= (x2 ? 68 : 72) && (x2 || ld >= 72) && Cout <= 3) {
    const int ngroups = (M + 15) / 16;
    const int grid = std::min(512, (ngroups + 3) / 4);
    if (ws && ws_bytes >= (size_t)grid * HEAD_ROW * sizeof(float)) {
      hipLaunchKernelGGL(dense_head_mfma_kernel<T>, dim3(grid), dim3(256), 0, s, reinterpret_cast<const T*>(x), ld, w, b, target, pred,
                         reinterpret_cast<T*>(dx), lddx, ws, M, Cin, Cout, ls, reinterpret_cast<const T*>(x2), ldx2);
      hipLaunchKernelGGL(dense_head_finish_kernel, dim3(HEAD_ROW / 32), dim3(1024), 0, s, ws, grid, dw, db, loss, db_dx, Cin * Cout, Cout,
                         1.0f / ((float)M * (float)Cout), accumulate);
      return gct2_check_launch("dense_head_train");
    }
  }
  if (x2) return gct2_fail(GCT2_EINVAL, "dense_head_train: a split input (x2) needs the matrix-core version: Cmask = 64, Cin <= 68, "
                                        "Cout <= 3 and a registered workspace");
  if (!accumulate) {            // the LDS-tile kernel adds its per-work-group sums with atomics
    (void)hipMemsetAsync(dw, 0, (size_t)Cin * Cout * sizeof(float), s);
    if (db) (void)hipMemsetAsync(db, 0, (size_t)Cout * sizeof(float), s);
    if (db_dx) (void)hipMemsetAsync(db_dx, 0, (size_t)Cmask * sizeof(float), s);
  }
  constexpr int PIX = 256;
  const size_t lds = (size_t)PIX * ld * 2 + (size_t)PIX * Cmask * 2 + PIX * 16 + (size_t)ld * 16;
  const int ntiles = (M + PIX - 1) / PIX;
  const int grid = ntiles < 1024 ? ntiles : 1024;
  auto kern = dense_head_train_kernel<T, PIX>;
  static size_t attr_lds = 0;
  if (lds > attr_lds) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_lds = lds;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, reinterpret_cast<const T*>(x), ld, w, b, target, pred, reinterpret_cast<T*>(dx),
                     lddx, dw, db, partials, M, Cin, Cout, Cmask, ls, db_dx);
  hipLaunchKernelGGL(mse_finish_kernel, dim3(1), dim3(256), 0, s, partials, grid, loss, 1.0f / ((float)M * (float)Cout));
  return gct2_check_launch("dense_head_train");
}
int pw_dense_head_train(const gct2_ctx& c, int dtype, const void* x, int ld, const float* w, const float* b, const float* target, float* pred,
                        void* dx, int lddx, float* dw, float* db, float* loss, float* partials, int M, int Cin, int Cout, int Cmask,
                        const float* ls, float* db_dx, const void* x2, int ldx2, int accumulate, hipStream_t s) {
  if (dtype == GCT2_BF16)
    return dense_head_train_t<__bf16>(c, x, ld, w, b, target, pred, dx, lddx, dw, db, loss, partials, M, Cin, Cout, Cmask, ls, db_dx, x2, ldx2,
                                      accumulate, s);
  return dense_head_train_t<_Float16>(c, x, ld, w, b, target, pred, dx, lddx, dw, db, loss, partials, M, Cin, Cout, Cmask, ls, db_dx, x2, ldx2,
                                      accumulate, s);
}
template <typename T>
static int diffusion_mix_t(const float* x, const float* e, float a, float* fake, void* out, int ldout, void* out2, int ldout2, size_t npix,
                           int C, hipStream_t s) {
  const size_t n = npix * C;
  hipLaunchKernelGGL(diffusion_mix_kernel<T>, dim3(blocks_for(n, 256)), dim3(256), 0, s, x, e, sqrtf(a), sqrtf(1.f - a), fake,
                     reinterpret_cast<T*>(out), ldout, reinterpret_cast<T*>(out2), ldout2, n, C);
  return gct2_check_launch("diffusion_mix");
}
int pw_diffusion_mix(int dtype, const float* x, const float* e, float a, float* fake, void* out, int ldout, void* out2, int ldout2,
                     size_t npix, int C, hipStream_t s) {
  if (dtype == GCT2_F32) return diffusion_mix_t<float>(x, e, a, fake, out, ldout, out2, ldout2, npix, C, s);
  if (dtype == GCT2_BF16) return diffusion_mix_t<__bf16>(x, e, a, fake, out, ldout, out2, ldout2, npix, C, s);
  return diffusion_mix_t<_Float16>(x, e, a, fake, out, ldout, out2, ldout2, npix, C, s);
}
int pw_diffusion_update(int mode, const float* pred, const float* fake, double a, double a1, float* x, float* e, size_t n, hipStream_t s) {
  const dim3 grid(blocks_for(n, 256)), block(256);
  // every scalar in double first, like the Python floats of train.py:382-391, then one cast where it meets an fp32 tensor
  const double dsa = sqrt(a), dsb = sqrt(1. - a), dsa1 = sqrt(a1), dsb1 = sqrt(1. - a1);
  const float sa = (float)dsa, sb = (float)dsb, sb1 = (float)dsb1, den = (float)(dsa1 * dsb - dsa * dsb1);
  if (mode == GCT2_SAMPLE_X) hipLaunchKernelGGL(diffusion_update_kernel<GCT2_SAMPLE_X>, grid, block, 0, s, pred, fake, sa, sb, sb1, den, x, e, n);
  else if (mode == GCT2_SAMPLE_EPS) hipLaunchKernelGGL(diffusion_update_kernel<GCT2_SAMPLE_EPS>, grid, block, 0, s, pred, fake, sa, sb, sb1, den, x, e, n);
  else if (mode == GCT2_SAMPLE_SCALED_EPS)
    hipLaunchKernelGGL(diffusion_update_kernel<GCT2_SAMPLE_SCALED_EPS>, grid, block, 0, s, pred, fake, sa, sb, sb1, den, x, e, n);
  else hipLaunchKernelGGL(diffusion_update_kernel<GCT2_SAMPLE_ODE>, grid, block, 0, s, pred, fake, sa, sb, sb1, den, x, e, n);
  return gct2_check_launch("diffusion_update");
}
int pw_noise_edits(const float* eps, const float* dict, int K, float* out, int H, int W, int C, hipStream_t s) {
  hipLaunchKernelGGL(noise_edits_kernel, dim3((H * W + 255) / 256), dim3(256), 0, s, eps, dict, K, out, H, W, C);
  return gct2_check_launch("noise_edits");
}
int pw_image_prepare(const uint8_t* src, const int64_t* offsets, const int32_t* dims, float* dst, int B, int size, hipStream_t s) {
  hipLaunchKernelGGL(image_prepare_kernel, dim3(blocks_for((size_t)B * size * size * 3, 256)), dim3(256), 0, s, src, offsets, dims, dst, B, size);
  return gct2_check_launch("image_prepare");
}
int pw_mse(const float* pred, const float* target, float* dpred, float* loss, float* partials, size_t n, const float* ls, hipStream_t s) {
  const int nb = blocks_for(n, 1024) > 1024 ? 1024 : blocks_for(n, 1024);
  hipLaunchKernelGGL(mse_kernel, dim3(nb), dim3(256), 0, s, pred, target, dpred, partials, n, ls);
  hipLaunchKernelGGL(mse_finish_kernel, dim3(1), dim3(256), 0, s, partials, nb, loss, 1.0f / (float)n);
  return gct2_check_launch("mse_fwd_bwd");
}
template <typename T>
static int relu_mask_t(const void* act, int ldact, void* d, int ldd, size_t npix, int C, hipStream_t s) {
  hipLaunchKernelGGL(relu_mask_kernel<T>, dim3(blocks_for(npix * C, 256)), dim3(256), 0, s, reinterpret_cast<const T*>(act), ldact,
                     reinterpret_cast<T*>(d), ldd, npix * C, C);
  return gct2_check_launch("relu_mask");
}
int pw_relu_mask(int dtype, const void* act, int ldact, void* d, int ldd, size_t npix, int C, hipStream_t s) {
  if (dtype == GCT2_F32) return relu_mask_t<float>(act, ldact, d, ldd, npix, C, s);
  if (dtype == GCT2_BF16) return relu_mask_t<__bf16>(act, ldact, d, ldd, npix, C, s);
  return relu_mask_t<_Float16>(act, ldact, d, ldd, npix, C, s);
}
template <typename T>
static int add_t(void* dst, int lddst, const void* src, int ldsrc, size_t npix, int C, hipStream_t s) {
  hipLaunchKernelGGL(add_kernel<T>, dim3(blocks_for(npix * C, 256)), dim3(256), 0, s, reinterpret_cast<T*>(dst), lddst,
                     reinterpret_cast<const T*>(src), ldsrc, npix * C, C);
  return gct2_check_launch("add");
}
int pw_add(int dtype, void* dst, int lddst, const void* src, int ldsrc, size_t npix, int C, hipStream_t s) {
  if (dtype == GCT2_F32) return add_t<float>(dst, lddst, src, ldsrc, npix, C, s);
  if (dtype == GCT2_BF16) return add_t<__bf16>(dst, lddst, src, ldsrc, npix, C, s);
  return add_t<_Float16>(dst, lddst, src, ldsrc, npix, C, s);
}
int pw_mix_per_image(const float* x, const float* eps, const float* a, const float* c, float* out, int B, size_t per_image, hipStream_t s) {
  const size_t n = (size_t)B * per_image;
  hipLaunchKernelGGL(mix_per_image_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, x, eps, a, c, out, n, per_image);
  return gct2_check_launch("mix_per_image");
}
template <typename T>
static int colsum_t(const void* dz, int ld, float* db, size_t M, int C, float sign, hipStream_t s) {
  const int ctiles = (C + 63) / 64;
  int rblocks = (int)((M + 511) / 512);
  const int cap = (1024 + ctiles - 1) / ctiles;
  if (rblocks > cap) rblocks = cap;
  if (rblocks < 1) rblocks = 1;
  const int rows_per_block = (int)((M + rblocks - 1) / rblocks);
  if constexpr (sizeof(T) == 2) {
    if (C % 8 == 0 && ld % 8 == 0 && (uintptr_t)dz % 16 == 0) {
      hipLaunchKernelGGL(colsum_vec_kernel<T>, dim3(ctiles, rblocks), dim3(256), 0, s, reinterpret_cast<const T*>(dz), ld, db, M, C,
                         rows_per_block, sign);
      return gct2_check_launch("colsum_vec");
    }
  }
  hipLaunchKernelGGL(colsum_kernel<T>, dim3(ctiles, rblocks), dim3(256), 0, s, reinterpret_cast<const T*>(dz), ld, db, M, C, rows_per_block, sign);
  return gct2_check_launch("colsum");
}
int pw_colsum(int dtype, const void* dz, int ld, float* db, size_t M, int C, float sign, hipStream_t s) {
  if (dtype == GCT2_F32) return colsum_t<float>(dz, ld, db, M, C, sign, s);
  if (dtype == GCT2_BF16) return colsum_t<__bf16>(dz, ld, db, M, C, sign, s);
  return colsum_t<_Float16>(dz, ld, db, M, C, sign, s);
}
int pw_adam(float* p, float* m, float* v, float* g, void* shadow, int sdt, size_t n, float alpha, float b1, float b2, float eps,
            float grad_mul, const gct2_loss_scale_state* ls, int zero_grad, hipStream_t s,
            const float* slabs, int nslab, size_t slab_stride, size_t n_slab) {
  if (n == 0) return GCT2_OK;
  const int nb = blocks_for(n / 4 + 4, 256);
#define GCT2_ADAM(S, HS) hipLaunchKernelGGL((adam_kernel<S, HS>), dim3(nb), dim3(256), 0, s, p, m, v, g, reinterpret_cast<S*>(shadow), n, alpha, b1, b2, eps, grad_mul, ls, zero_grad, slabs, nslab, slab_stride, n_slab)
  if (!shadow) GCT2_ADAM(float, false);
  else if (sdt == GCT2_BF16) GCT2_ADAM(__bf16, true);
  else if (sdt == GCT2_F16) GCT2_ADAM(_Float16, true);
  else GCT2_ADAM(float, true);
#undef GCT2_ADAM
  return gct2_check_launch("adam_keras_multi");
}
int pw_cast(int dtype, const float* src, void* dst, size_t n, hipStream_t s) {
  if (n == 0) return GCT2_OK;
  const int nb = blocks_for(n, 256);
  if (dtype == GCT2_BF16) hipLaunchKernelGGL(cast_kernel<__bf16>, dim3(nb), dim3(256), 0, s, src, reinterpret_cast<__bf16*>(dst), n);
  else if (dtype == GCT2_F16) hipLaunchKernelGGL(cast_kernel<_Float16>, dim3(nb), dim3(256), 0, s, src, reinterpret_cast<_Float16*>(dst), n);
  else hipLaunchKernelGGL(cast_kernel<float>, dim3(nb), dim3(256), 0, s, src, reinterpret_cast<float*>(dst), n);
  return gct2_check_launch("cast_from_f32");
}
int pw_ls_init(gct2_loss_scale_state* st, float scale, hipStream_t s) {
  hipLaunchKernelGGL(ls_init_kernel, dim3(1), dim3(1), 0, s, st, scale);
  return gct2_check_launch("loss_scale_init");
}
int pw_ls_begin(gct2_loss_scale_state* st, float base_lr, int warmup_steps, float b1, float b2, hipStream_t s) {
  hipLaunchKernelGGL(ls_begin_kernel, dim3(1), dim3(1), 0, s, st, base_lr, warmup_steps, b1, b2);
  return gct2_check_launch("loss_scale_begin");
}
int pw_ls_check(const float* g, size_t n, gct2_loss_scale_state* st, hipStream_t s) {
  if (n == 0) return GCT2_OK;
  hipLaunchKernelGGL(ls_check_kernel, dim3(blocks_for(n, 1024)), dim3(256), 0, s, g, n, st);
  return gct2_check_launch("scale_check_finite");
}
int pw_ls_update(gct2_loss_scale_state* st, int growth_interval, hipStream_t s) {
  hipLaunchKernelGGL(ls_update_kernel, dim3(1), dim3(1), 0, s, st, growth_interval);
  return gct2_check_launch("loss_scale_update");
}

// ---- ReLU bit plane derived from a stored activation tensor (fall-back of gct2_ctx_set_relu_bits: launches whose epilogue cannot
// write the plane - 8-byte epilogues, split-K finalize, the direct kernels): bit k of bits[pixel][c] = (y[pixel][8c + k] > 0)
namespace {
template <typename T>
__global__ void relu_bits_kernel(const T* __restrict__ y, int ldy, size_t pixels, int groups, unsigned char* __restrict__ bits, int ldbits) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pixels * (size_t)groups) return;
  const size_t pix = i / groups;
  const int gidx = (int)(i - pix * groups);
  const T* src = y + pix * ldy + 8 * gidx;
  unsigned m = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) m |= ((float)src[k] > 0.f ? 1u : 0u) << k;
  bits[pix * ldbits + gidx] = (unsigned char)m;
}
}  // namespace
int pw_relu_bits(int dtype, const void* y, int ldy, size_t pixels, int channels, unsigned char* bits, int ldbits, hipStream_t s) {
  const int groups = channels / 8;
  const size_t n = pixels * (size_t)groups;
  if (!n) return GCT2_OK;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == GCT2_F32) hipLaunchKernelGGL(relu_bits_kernel<float>, grid, dim3(256), 0, s, (const float*)y, ldy, pixels, groups, bits, ldbits);
  else if (dtype == GCT2_BF16) hipLaunchKernelGGL(relu_bits_kernel<__bf16>, grid, dim3(256), 0, s, (const __bf16*)y, ldy, pixels, groups, bits, ldbits);
  else hipLaunchKernelGGL(relu_bits_kernel<_Float16>, grid, dim3(256), 0, s, (const _Float16*)y, ldy, pixels, groups, bits, ldbits);
  return gct2_check_launch("relu_bits");
}

// gct2_stream_occupy: hold `workgroups` work-group slots for `ticks` of the constant 100-MHz counter.  Every wave watches its own
// clock, so every wave reaches the exit whatever the others do.
namespace {
__global__ __launch_bounds__(256) void occupy_kernel(unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
}  // namespace
int pw_occupy(int workgroups, unsigned long long ticks, hipStream_t s) {
  hipLaunchKernelGGL(occupy_kernel, dim3(workgroups), dim3(256), 0, s, ticks);
  return gct2_check_launch("stream_occupy");
}
