// Weight-gradient GEMM of the 4x4/stride-2 convolutions (autodiff of train.py:148-153,161-166) on gfx950.
//
//   dW[gc][cs] += sum_r big[pix_big(r, tap(gc))][cb(gc)] * small[r][cs],   gc = tap*Cb + cb  (16*Cb rows)
//
// r walks the SMALL grid [B,Hs,Ws]; pix_big(r,(kh,kw)) = (2sh+kh-1, 2sw+kw-1) on the BIG grid (zero outside).
//   Conv2D          : big = layer input x (Cb = Cin),  small = dz (Cs = Cout)  -> dW (4,4,Cin,Cout)
//   Conv2DTranspose : big = dz (Cb = Cout),            small = x  (Cs = Cin)   -> dW (4,4,Cout,Cin)
// Both operands have the REDUCTION index r as their slow (row) index in memory, so both LDS tiles are
// "T images" ([r][128 channels]) consumed through ds_read_tr16_b64.  LDS-DMA staging (buffer_load ... lds; swizzle on
// the source address, out-of-range offset = zero fill).  Three kernels:
//   wgrad256p_kernel : 256 x 256 tile, 8 waves of 128 x 64, four 32-row stages with a spanning pipeline (default where the
//                      tile count still gives about one work-group per CU)
//   wgrad256_kernel  : the same tile with two 64-row buffers (kept for A/B timing and as a parity cross-check)
//   wgrad_kernel     : 128 x 128 tile, 4 waves, one or two 64-row buffers, 2-4 work-groups per CU (small layers)
// The reduction over r is split across work-groups (XCD-aware 1-D grid); partial tiles go to workspace slabs summed in a
// fixed order by wgrad_reduce_kernel (reproducible), to their single owner (read-add-write), or to fp32 atomics.
#include "gct2_common.h"
#include <type_traits>
#include <algorithm>

namespace {

constexpr unsigned OOB = 0x80000000u;
typedef __attribute__((address_space(3))) void lds_void_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)OOB, 0x00020000);
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_piece, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)lds_piece, 16, (int)voff, 0, 0, 0);
}

// S1: the weight gradient of a 'same' stride-1 convolution with p.ks x p.ks taps (Block's 3x3, the 1x1 projection of residual=True:
// train.py:104-143): both tensors on ONE grid, tap (kh, kw) pairs pixel r with (h + kh - pad, w + kw - pad); everything else is shared.
template <typename T, int NBUF, bool S1 = false>
__global__ __launch_bounds__(256, NBUF == 1 ? 4 : 2) void wgrad_kernel(WgradParams p) {
  constexpr int IMG = 64 * 256;
  const int KS = S1 ? p.ks : 4, STRIDE = S1 ? 1 : 2, PAD = S1 ? (p.ks - 1) / 2 : 1;   // compile-time constants for the 4x4 layers
  __shared__ __attribute__((aligned(16))) char lds0[2 * IMG];     // [big image | small image]
  __shared__ __attribute__((aligned(16))) char lds1[NBUF == 2 ? 2 * IMG : 16];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wm = wave >> 1;
  const int Hs = p.Hs, Ws = p.Ws, Cb = p.Cb, Cs = p.Cs;
  const int Hb = STRIDE * Hs, Wb = STRIDE * Ws;
  const int R = p.B * Hs * Ws;
  const int GC = KS * KS * Cb;
  const int tiles_n = (Cs + 127) / 128;
  // XCD-aware 1-D grid: all output tiles of one r-split re-read the same rows of both operands, so a whole split
  // runs on ONE XCD (ids with equal id % 8 share an L2); with fewer than 8 splits the plain order is kept
  const int tiles = ((GC + 127) / 128) * tiles_n;
  int tile, split;
  if (p.rsplit >= 8) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    tile = j % tiles;
    split = (j / tiles) * 8 + xcd;
    if (split >= p.rsplit) return;
  } else {
    tile = blockIdx.x % tiles;
    split = blockIdx.x / tiles;
  }
  const int gc0 = (tile / tiles_n) * 128, cs0 = (tile % tiles_n) * 128;
  // r range of this split, in whole 64-row steps
  const int steps_total = (R + 63) / 64;
  const int steps_per = (steps_total + p.rsplit - 1) / p.rsplit;
  const int step_lo = split * steps_per;
  const int step_hi = min(steps_total, step_lo + steps_per);
  if (step_lo >= step_hi) return;

  const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(p.big), rs_s = make_rsrc(p.small);

  // piece q = wave + 4 i of a T image = rows 4q .. 4q+3; lane -> row 4q + (lane>>4), physical 16-byte chunk lane&15
  const int row0 = 4 * wave + (lane >> 4);                       // row of piece i is row0 + 16 i
  const int lc = ((((lane & 15) >> 1) ^ timg_swz(row0)) << 1) | (lane & 1);   // logical chunk (same for every i)
  const int gc = gc0 + lc * 8;
  const bool gc_ok = gc < GC;
  const int tap = gc_ok ? gc / Cb : 0, cb = gc_ok ? gc - tap * Cb : 0;
  const int kh = S1 ? tap / KS : tap >> 2, kw = S1 ? tap - kh * KS : tap & 3;
  const bool cs_ok = (cs0 + lc * 8) < Cs;
  const int ldb2 = p.ldbig * 2, lds2 = p.ldsmall * 2;

  // incremental (b, sh, sw) decode of r, advanced by 64 rows per step
  const int adv_w = 64 % Ws, q1 = 64 / Ws, adv_h = q1 % Hs, adv_b = q1 / Hs;
  int rb[4], rh[4], rw[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int r = step_lo * 64 + row0 + 16 * i;
    rw[i] = r % Ws; const int t = r / Ws; rh[i] = t % Hs; rb[i] = t / Hs;
  }

  auto issue = [&](int step, char* base) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int r = step * 64 + row0 + 16 * i;
      const int h = STRIDE * rh[i] + kh - PAD, w = STRIDE * rw[i] + kw - PAD;
      const bool okb = gc_ok && r < R && (unsigned)h < (unsigned)Hb && (unsigned)w < (unsigned)Wb;
      const unsigned offb = (unsigned)(((rb[i] * Hb + h) * Wb + w) * ldb2 + cb * 2);
      dma16(rs_b, base + (wave + 4 * i) * 1024, okb ? offb : OOB);
      const unsigned offs = (unsigned)(r * lds2 + (cs0 + lc * 8) * 2);
      dma16(rs_s, base + IMG + (wave + 4 * i) * 1024, (cs_ok && r < R) ? offs : OOB);
      // advance this row by 64 for the next step
      rw[i] += adv_w; rh[i] += adv_h; rb[i] += adv_b;
      if (rw[i] >= Ws) { rw[i] -= Ws; rh[i]++; }
      if (rh[i] >= Hs) { rh[i] -= Hs; rb[i]++; }
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](const char* base) {
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      u32x4_t bf[4], sf[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        bf[i] = timg_frag(base, wm * 64 + i * 16, kk, lane);
        sf[i] = timg_frag(base + IMG, wn * 64 + i * 16, kk, lane);
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = mfma16<T>(sf[j], bf[i], acc[i][j]);
    }
  };

  if constexpr (NBUF == 1) {
    // one 32-KiB buffer, 4 work-groups per CU cover each other's DMA latency
    for (int step = step_lo; step < step_hi; step++) {
      issue(step, lds0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      compute(lds0);
      __syncthreads();
    }
  } else {
    issue(step_lo, lds0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int step = step_lo; step < step_hi; step += 2) {        // two steps per trip: buffer roles are compile-time
      if (step + 1 < step_hi) issue(step + 1, lds1);
      compute(lds0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (step + 1 >= step_hi) break;
      if (step + 2 < step_hi) issue(step + 2, lds0);
      compute(lds1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  // MFMA operand order A = small (cs), B = big (gc): lane holds dW[gc = .. + (lane&15)][cs = .. + 4*(lane>>4) + r], i.e. four
  // consecutive columns of one row -> 16-byte stores (the other order costs four times the store instructions: -7..-24 %)
  //   rsplit == 1        : the tile has one owner -> plain read-add-write (no atomic unit, reproducible)
  //   slabs (p.ws)       : plain stores of the partial tile into slab[split]; wgrad_reduce_kernel adds the slabs in a fixed
  //                        order (reproducible, and plain stores run ~4x the chip-wide float-atomic rate)
  //   otherwise          : fp32 atomics
  float* __restrict__ out = p.ws ? p.ws + (size_t)split * GC * Cs : p.dw;
  const int mode = p.ws ? 2 : (p.rsplit == 1 ? 1 : 0);
  // an opaque copy of the lane id: keeps hipcc from hoisting the output addresses of all tiles above the reduction loop, where
  // they would occupy registers for the whole kernel
  int elane = lane;
  asm volatile("" : "+v"(elane));
  if constexpr (NBUF == 2) if (mode == 1 && p.adam.p) {   // (the single-buffer variant runs at 128 registers: not with this epilogue)
    // the tile has ONE owner and the caller wants the optimizer step: Keras Adam right here, on the accumulators - dW is neither
    // written nor re-read (8 B per parameter less), p / m / v / the operand copy are updated in place.  One accumulator row (four
    // 4-column groups) at a time: its twelve 16-byte loads are issued together, then the arithmetic, then the stores - element by
    // element the round trips serialise (measured: +28 us per step in that form).
    const float* __restrict__ ap = p.adam.p; const float* __restrict__ am = p.adam.m; const float* __restrict__ av = p.adam.v;
    const float ob1 = 1.f - p.adam.b1, ob2 = 1.f - p.adam.b2;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int row = gc0 + wm * 64 + i * 16 + (elane & 15);
      f32x4_t pv[4], mv[4], vv[4];
      bool ok[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int col = cs0 + wn * 64 + j * 16 + 4 * (elane >> 4);
        ok[j] = row < GC && col < Cs;
        const size_t e = ok[j] ? (size_t)row * Cs + col : 0;
        pv[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(ap + e));
        mv[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(am + e));
        vv[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(av + e));
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (!ok[j]) continue;
        const int col = cs0 + wn * 64 + j * 16 + 4 * (elane >> 4);
        const size_t e = (size_t)row * Cs + col;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          float pp = pv[j][r], mm = mv[j][r], v1 = vv[j][r];
          adam_keras_update(pp, mm, v1, acc[i][j][r] * p.adam.gmul, p.adam.alpha, p.adam.b1, ob1, p.adam.b2, ob2, p.adam.eps);
          pv[j][r] = pp; mv[j][r] = mm; vv[j][r] = v1;
        }
        __builtin_nontemporal_store(pv[j], reinterpret_cast<f32x4_t*>(p.adam.p + e));
        __builtin_nontemporal_store(mv[j], reinterpret_cast<f32x4_t*>(p.adam.m + e));
        __builtin_nontemporal_store(vv[j], reinterpret_cast<f32x4_t*>(p.adam.v + e));
        if (p.adam.shadow) {
          const u32x2_t o = {pack2<T>(pv[j][0], pv[j][1]), pack2<T>(pv[j][2], pv[j][3])};
          *reinterpret_cast<u32x2_t*>(reinterpret_cast<T*>(p.adam.shadow) + e) = o;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int row = gc0 + wm * 64 + i * 16 + (elane & 15);
    if (row >= GC) continue;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int col = cs0 + wn * 64 + j * 16 + 4 * (elane >> 4);
      if (col >= Cs) continue;                         // Cs is a multiple of 8: a 4-column group is inside or outside as a whole
      float* q = out + (size_t)row * Cs + col;
      if (mode == 2) *reinterpret_cast<f32x4_t*>(q) = acc[i][j];
      else if (mode == 1) { if (p.accumulate) *reinterpret_cast<f32x4_t*>(q) += acc[i][j]; else *reinterpret_cast<f32x4_t*>(q) = acc[i][j]; }
      else {
#pragma unroll
        for (int r = 0; r < 4; r++) atomicAdd(q + r, acc[i][j][r]);
      }
    }
  }
}

// ---- 256 x 256 output tile: 8 waves (2 along gc x 4 along cs), each 128 x 64 (8 x 4 MFMA tiles, 128 accumulator
// registers).  A 64-row step moves 64 KiB for 8.4 MFLOP = 128 FLOP per L2->LDS byte, twice the 128 x 128 tile: the
// measured bound of the small tile is the ~20 TB/s L2->LDS path.  Two 64-KiB LDS buffers, one
// work-group per CU.  Images per buffer: [big gc 0..127 | big gc 128..255 | small cs 0..127 | small cs 128..255].
template <typename T>
__global__ __launch_bounds__(512, 2) void wgrad256_kernel(WgradParams p) {
  constexpr int IMG = 64 * 256;
  __shared__ __attribute__((aligned(16))) char lds0[4 * IMG];
  __shared__ __attribute__((aligned(16))) char lds1[4 * IMG];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 3, wm = wave >> 2;
  const int Hs = p.Hs, Ws = p.Ws, Cb = p.Cb, Cs = p.Cs;
  const int Hb = 2 * Hs, Wb = 2 * Ws;
  const int R = p.B * Hs * Ws;
  const int GC = 16 * Cb;
  const int tiles_n = (Cs + 255) / 256;
  const int tiles = ((GC + 255) / 256) * tiles_n;
  int tile, split;
  if (p.rsplit >= 8) {            // a whole r-split on one XCD (ids with equal id % 8 share an L2)
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    tile = j % tiles;
    split = (j / tiles) * 8 + xcd;
    if (split >= p.rsplit) return;
  } else {
    tile = blockIdx.x % tiles;
    split = blockIdx.x / tiles;
  }
  const int gc0 = (tile / tiles_n) * 256, cs0 = (tile % tiles_n) * 256;
  const int steps_total = (R + 63) / 64;
  const int steps_per = (steps_total + p.rsplit - 1) / p.rsplit;
  const int step_lo = split * steps_per;
  const int step_hi = min(steps_total, step_lo + steps_per);
  if (step_lo >= step_hi) return;

  const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(p.big), rs_s = make_rsrc(p.small);
  // piece q = wave + 8 i (i < 2) of every image = rows 4q .. 4q+3; lane -> row 4q + (lane>>4), physical chunk lane&15
  const int row0 = 4 * wave + (lane >> 4);                       // rows row0 and row0 + 32
  const int lc = ((((lane & 15) >> 1) ^ timg_swz(row0)) << 1) | (lane & 1);   // same for row0 + 32
  int kh[2], kw[2], cb[2];
  bool gc_ok[2], cs_ok[2];
#pragma unroll
  for (int g = 0; g < 2; g++) {                                  // image g of each operand
    const int gc = gc0 + 128 * g + lc * 8;
    gc_ok[g] = gc < GC;
    const int tap = gc_ok[g] ? gc / Cb : 0;
    cb[g] = gc_ok[g] ? gc - tap * Cb : 0;
    kh[g] = tap >> 2; kw[g] = tap & 3;
    cs_ok[g] = (cs0 + 128 * g + lc * 8) < Cs;
  }
  const int ldb2 = p.ldbig * 2, lds2 = p.ldsmall * 2;
  const int adv_w = 64 % Ws, q1 = 64 / Ws, adv_h = q1 % Hs, adv_b = q1 / Hs;
  int rb[2], rh[2], rw[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int r = step_lo * 64 + row0 + 32 * i;
    rw[i] = r % Ws; const int t = r / Ws; rh[i] = t % Hs; rb[i] = t / Hs;
  }
  auto issue = [&](int step, char* base) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int r = step * 64 + row0 + 32 * i;
      const bool r_ok = r < R;
      char* piece = base + (wave + 8 * i) * 1024;
#pragma unroll
      for (int g = 0; g < 2; g++) {
        const int h = 2 * rh[i] + kh[g] - 1, w = 2 * rw[i] + kw[g] - 1;
        const bool okb = gc_ok[g] && r_ok && (unsigned)h < (unsigned)Hb && (unsigned)w < (unsigned)Wb;
        dma16(rs_b, piece + g * IMG, okb ? (unsigned)(((rb[i] * Hb + h) * Wb + w) * ldb2 + cb[g] * 2) : OOB);
        dma16(rs_s, piece + (2 + g) * IMG, (cs_ok[g] && r_ok) ? (unsigned)(r * lds2 + (cs0 + 128 * g + lc * 8) * 2) : OOB);
      }
      rw[i] += adv_w; rh[i] += adv_h; rb[i] += adv_b;
      if (rw[i] >= Ws) { rw[i] -= Ws; rh[i]++; }
      if (rh[i] >= Hs) { rh[i] -= Hs; rb[i]++; }
    }
  };

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](const char* base) {
    const char* bimg = base + wm * IMG;                           // this wave's 128 gc rows = one whole big image
    const char* simg = base + (2 + (wn >> 1)) * IMG;
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      u32x4_t sf[4];
#pragma unroll
      for (int j = 0; j < 4; j++) sf[j] = timg_frag(simg, (wn & 1) * 64 + j * 16, kk, lane);
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const u32x4_t bf = timg_frag(bimg, i * 16, kk, lane);
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = mfma16<T>(sf[j], bf, acc[i][j]);
      }
    }
  };

  issue(step_lo, lds0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int step = step_lo; step < step_hi; step += 2) {
    if (step + 1 < step_hi) issue(step + 1, lds1);
    compute(lds0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (step + 1 >= step_hi) break;
    if (step + 2 < step_hi) issue(step + 2, lds0);
    compute(lds1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  float* __restrict__ out = p.ws ? p.ws + (size_t)split * GC * Cs : p.dw;
  const int mode = p.ws ? 2 : (p.rsplit == 1 ? 1 : 0);
  // an opaque copy of the lane id: keeps hipcc from hoisting the 32 tiles' output addresses above the reduction loop (they would
  // occupy ~64 registers for the whole kernel: 440 spilled registers and a 10x slower kernel, measured r02)
  int elane = lane;
  asm volatile("" : "+v"(elane));
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int row = gc0 + wm * 128 + i * 16 + (elane & 15);
    if (row >= GC) continue;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int col = cs0 + wn * 64 + j * 16 + 4 * (elane >> 4);
      if (col >= Cs) continue;
      float* q = out + (size_t)row * Cs + col;
      if (mode == 2) *reinterpret_cast<f32x4_t*>(q) = acc[i][j];
      else if (mode == 1) { if (p.accumulate) *reinterpret_cast<f32x4_t*>(q) += acc[i][j]; else *reinterpret_cast<f32x4_t*>(q) = acc[i][j]; }
      else {
#pragma unroll
        for (int r = 0; r < 4; r++) atomicAdd(q + r, acc[i][j][r]);
      }
    }
  }
}

// ---- 256 (gc) x 128 (cs) tile, 8 waves of 64 x 64, ONE 48-KiB buffer, TWO work-groups per CU (r03) -------------------------------
// The structure that measures best for the forward / input-gradient GEMMs (tapgemm_kernel<..., 256, 128, ..., NBUF = 1>): no
// pipelining inside a work-group - issue, wait, multiply - and a second, independent work-group on the CU whose multiplies cover
// the first one's wait.  87 FLOP per staged byte (128 x 128: 65; 256 x 256: 131).  One-work-group-per-CU pipelines, however deep,
// measured 15-30 % slower than this arrangement on the tap GEMMs (profiles/r03_layer_variants.txt).
template <typename T>
__global__ __launch_bounds__(512, 4) void wgrad2x_kernel(WgradParams p) {
  constexpr int IMG = 64 * 256;                                   // [big gc 0..127 | big gc 128..255 | small cs 0..127]
  __shared__ __attribute__((aligned(16))) char lds0[3 * IMG];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wm = wave >> 1;                        // 4 (gc) x 2 (cs) waves of 64 x 64
  const int Hs = p.Hs, Ws = p.Ws, Cb = p.Cb, Cs = p.Cs;
  const int Hb = 2 * Hs, Wb = 2 * Ws;
  const int R = p.B * Hs * Ws;
  const int GC = 16 * Cb;
  const int tiles_n = (Cs + 127) / 128;
  const int tiles = ((GC + 255) / 256) * tiles_n;
  int tile, split;
  if (p.rsplit >= 8) {            // a whole r-split on one XCD (ids with equal id % 8 share an L2)
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    tile = j % tiles;
    split = (j / tiles) * 8 + xcd;
    if (split >= p.rsplit) return;
  } else {
    tile = blockIdx.x % tiles;
    split = blockIdx.x / tiles;
  }
  const int gc0 = (tile / tiles_n) * 256, cs0 = (tile % tiles_n) * 128;
  const int steps_total = (R + 63) / 64;
  const int steps_per = (steps_total + p.rsplit - 1) / p.rsplit;
  const int step_lo = split * steps_per;
  const int step_hi = min(steps_total, step_lo + steps_per);
  if (step_lo >= step_hi) return;

  const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(p.big), rs_s = make_rsrc(p.small);
  // piece q = wave + 8 i (i < 2) of every image = rows 4q .. 4q+3; lane -> row 4q + (lane>>4), physical chunk lane&15
  const int row0 = 4 * wave + (lane >> 4);                       // rows row0 and row0 + 32
  const int lc = ((((lane & 15) >> 1) ^ timg_swz(row0)) << 1) | (lane & 1);   // same for row0 + 32
  int kh[2], kw[2], cb[2];
  bool gc_ok[2];
#pragma unroll
  for (int g = 0; g < 2; g++) {                                  // the two big images
    const int gc = gc0 + 128 * g + lc * 8;
    gc_ok[g] = gc < GC;
    const int tap = gc_ok[g] ? gc / Cb : 0;
    cb[g] = gc_ok[g] ? gc - tap * Cb : 0;
    kh[g] = tap >> 2; kw[g] = tap & 3;
  }
  const bool cs_ok = (cs0 + lc * 8) < Cs;
  const int ldb2 = p.ldbig * 2, lds2 = p.ldsmall * 2;
  const int adv_w = 64 % Ws, q1 = 64 / Ws, adv_h = q1 % Hs, adv_b = q1 / Hs;
  int rb[2], rh[2], rw[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int r = step_lo * 64 + row0 + 32 * i;
    rw[i] = r % Ws; const int t = r / Ws; rh[i] = t % Hs; rb[i] = t / Hs;
  }
  auto issue = [&](int step) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int r = step * 64 + row0 + 32 * i;
      const bool r_ok = r < R;
      char* piece = lds0 + (wave + 8 * i) * 1024;
#pragma unroll
      for (int g = 0; g < 2; g++) {
        const int h = 2 * rh[i] + kh[g] - 1, w = 2 * rw[i] + kw[g] - 1;
        const bool okb = gc_ok[g] && r_ok && (unsigned)h < (unsigned)Hb && (unsigned)w < (unsigned)Wb;
        dma16(rs_b, piece + g * IMG, okb ? (unsigned)(((rb[i] * Hb + h) * Wb + w) * ldb2 + cb[g] * 2) : OOB);
      }
      dma16(rs_s, piece + 2 * IMG, (cs_ok && r_ok) ? (unsigned)(r * lds2 + (cs0 + lc * 8) * 2) : OOB);
      rw[i] += adv_w; rh[i] += adv_h; rb[i] += adv_b;
      if (rw[i] >= Ws) { rw[i] -= Ws; rh[i]++; }
      if (rh[i] >= Hs) { rh[i] -= Hs; rb[i]++; }
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto compute = [&]() {
    const char* bimg = lds0 + (wm >> 1) * IMG;
    const char* simg = lds0 + 2 * IMG;
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      u32x4_t bf[4], sf[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        bf[i] = timg_frag(bimg, (wm & 1) * 64 + i * 16, kk, lane);
        sf[i] = timg_frag(simg, wn * 64 + i * 16, kk, lane);
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = mfma16<T>(sf[j], bf[i], acc[i][j]);
    }
  };

  for (int step = step_lo; step < step_hi; step++) {
    issue(step);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    compute();
    __syncthreads();
  }

  float* __restrict__ out = p.ws ? p.ws + (size_t)split * GC * Cs : p.dw;
  const int mode = p.ws ? 2 : (p.rsplit == 1 ? 1 : 0);
  int elane = lane;
  asm volatile("" : "+v"(elane));
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int row = gc0 + wm * 64 + i * 16 + (elane & 15);
    if (row >= GC) continue;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int col = cs0 + wn * 64 + j * 16 + 4 * (elane >> 4);
      if (col >= Cs) continue;
      float* q = out + (size_t)row * Cs + col;
      if (mode == 2) *reinterpret_cast<f32x4_t*>(q) = acc[i][j];
      else if (mode == 1) { if (p.accumulate) *reinterpret_cast<f32x4_t*>(q) += acc[i][j]; else *reinterpret_cast<f32x4_t*>(q) = acc[i][j]; }
      else {
#pragma unroll
        for (int r = 0; r < 4; r++) atomicAdd(q + r, acc[i][j][r]);
      }
    }
  }
}

// ---- the same 256 x 256 tile with a spanning pipeline ----------------------------------------------------------------------
// One work-group per CU has no second work-group to cover its DMA latency, and a loop that drains vmcnt to 0 at every
// barrier exposes that latency every step (measured on wgrad256_kernel: MFMA-only 108 us, DMA-only 84 us, together 149 us).
// Here a stage is 32 rows of r (4 images x 8 KiB = 32 KiB), FOUR stage buffers, the DMA of stage s+3 is issued while stage s
// is multiplied and only stage s+1 is waited for (counted vmcnt, raw s_barrier): two stages stay in flight across every
// barrier.  The DMA is inline asm so that hipcc's own vmcnt bookkeeping does not see it (it would drain it at the loop
// head); every wait for it is written out below.
#define GCT2_VMCNT_ONLY(n) ((((n) & 0xF) | 0x70 | 0xF00 | ((((n) >> 4) & 3) << 14)))
__device__ __forceinline__ void dma16_hidden(__amdgpu_buffer_rsrc_t rsrc, char* lds_piece, unsigned voff) {
  const unsigned lds_addr = (unsigned)(uintptr_t)(lds_void_t*)lds_piece;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc)
               : "memory", "m0");
}

#ifdef GCT2_STAMP
__device__ __forceinline__ unsigned long long wg_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define WG_STAMP(k) st[k] = wg_stamp()
#else
#define WG_STAMP(k)
#endif
// NST = stage buffers of the ring (4: 128 KiB, three stages = 96 KiB in flight; 5: all 160 KiB of the CU's LDS, four stages = 128 KiB
// in flight).  The loop is bound by the latency of the staging requests that miss the XCD's L2 (a stage takes
// about (loaded miss latency) / (stages in flight)), so the deeper ring is the default.
template <typename T, int NST>
__global__ __launch_bounds__(512, 2) void wgrad256p_kernel(WgradParams p) {
#ifdef GCT2_STAMP
  unsigned long long st[4];
  WG_STAMP(0);
#endif
  static_assert(NST == 4 || NST == 5, "four or five stage buffers");
  constexpr int IMG = 32 * 256;                                   // one T image of a stage: 32 r-rows x 128 columns
  constexpr int NDMA = 4;                                         // DMA instructions per wave per stage (one piece of each image)
  __shared__ __attribute__((aligned(16))) char lds0[4 * IMG];
  __shared__ __attribute__((aligned(16))) char lds1[4 * IMG];
  __shared__ __attribute__((aligned(16))) char lds2[4 * IMG];
  __shared__ __attribute__((aligned(16))) char lds3[4 * IMG];
  __shared__ __attribute__((aligned(16))) char lds4[NST == 5 ? 4 * IMG : 16];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 3, wm = wave >> 2;
  const int Hs = p.Hs, Ws = p.Ws, Cb = p.Cb, Cs = p.Cs;
  const int Hb = 2 * Hs, Wb = 2 * Ws;
  const int R = p.B * Hs * Ws;
  const int GC = 16 * Cb;
  const int tiles_n = (Cs + 255) / 256;
  const int tiles = ((GC + 255) / 256) * tiles_n;
  int tile, split;
  if (p.rsplit >= 8) {            // a whole r-split on one XCD (ids with equal id % 8 share an L2)
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    tile = j % tiles;
    split = (j / tiles) * 8 + xcd;
    if (split >= p.rsplit) return;
  } else {
    tile = blockIdx.x % tiles;
    split = blockIdx.x / tiles;
  }
  const int gc0 = (tile / tiles_n) * 256, cs0 = (tile % tiles_n) * 256;
  const int steps_total = (R + 63) / 64;
  const int steps_per = (steps_total + p.rsplit - 1) / p.rsplit;
  const int step_lo = split * steps_per;
  const int step_hi = min(steps_total, step_lo + steps_per);
  if (step_lo >= step_hi) return;
  const int st_lo = 2 * step_lo, st_hi = 2 * step_hi;             // 32-row stages (rows >= R are zero-filled)

  const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(p.big), rs_s = make_rsrc(p.small);
  // piece q = wave of every image = rows 4q .. 4q+3; lane -> row 4q + (lane>>4), physical chunk lane&15
  const int row0 = 4 * wave + (lane >> 4);
  const int lc = ((((lane & 15) >> 1) ^ timg_swz(row0)) << 1) | (lane & 1);
  int kh[2], kw[2], cb[2];
  bool gc_ok[2], cs_ok[2];
#pragma unroll
  for (int g = 0; g < 2; g++) {
    const int gc = gc0 + 128 * g + lc * 8;
    gc_ok[g] = gc < GC;
    const int tap = gc_ok[g] ? gc / Cb : 0;
    cb[g] = gc_ok[g] ? gc - tap * Cb : 0;
    kh[g] = tap >> 2; kw[g] = tap & 3;
    cs_ok[g] = (cs0 + 128 * g + lc * 8) < Cs;
  }
  const int ldb2 = p.ldbig * 2, lds2b = p.ldsmall * 2;
  const int adv_w = 32 % Ws, q1 = 32 / Ws, adv_h = q1 % Hs, adv_b = q1 / Hs;
  int rb, rh, rw;
  {
    const int r = st_lo * 32 + row0;
    rw = r % Ws; const int t = r / Ws; rh = t % Hs; rb = t / Hs;
  }
  auto issue = [&](int st, char* base) {                          // stages are issued in increasing order: (rb, rh, rw) advance
    const int r = st * 32 + row0;
    const bool r_ok = r < R;
    char* piece = base + wave * 1024;
#pragma unroll
    for (int g = 0; g < 2; g++) {
      const int h = 2 * rh + kh[g] - 1, w = 2 * rw + kw[g] - 1;
      const bool okb = gc_ok[g] && r_ok && (unsigned)h < (unsigned)Hb && (unsigned)w < (unsigned)Wb;
      dma16_hidden(rs_b, piece + g * IMG, okb ? (unsigned)(((rb * Hb + h) * Wb + w) * ldb2 + cb[g] * 2) : OOB);
      dma16_hidden(rs_s, piece + (2 + g) * IMG, (cs_ok[g] && r_ok) ? (unsigned)(r * lds2b + (cs0 + 128 * g + lc * 8) * 2) : OOB);
    }
    rw += adv_w; rh += adv_h; rb += adv_b;
    if (rw >= Ws) { rw -= Ws; rh++; }
    if (rh >= Hs) { rh -= Hs; rb++; }
  };

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](const char* base) {
    const char* bimg = base + wm * IMG;
    const char* simg = base + (2 + (wn >> 1)) * IMG;
    // an opaque copy of the lane id: otherwise the fragment addresses of all four unrolled stages (4 buffers x 12 fragments) are
    // loop-invariant, get hoisted above the pipeline loop and spill (462 VGPRs of scratch, 10x slower: measured r02)
    int ql = lane;
    asm volatile("" : "+v"(ql));
    u32x4_t sf[4];
#pragma unroll
    for (int j = 0; j < 4; j++) sf[j] = timg_frag(simg, (wn & 1) * 64 + j * 16, 0, ql);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const u32x4_t bf = timg_frag(bimg, i * 16, 0, ql);
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = mfma16<T>(sf[j], bf, acc[i][j]);
    }
  };

  // stage s: issue s+NST-1, multiply s, wait until only the DMAs of the stages beyond s+1 are outstanding, barrier.
  // `live` is always true (rsplit >= 1) but opaque to hipcc: with the multiply unconditional the unrolled stages are merged
  // into one region whose live ranges no longer fit (256 VGPRs + 440 spilled, 10x slower; behind the guard: 176 VGPRs, no spill).
  const bool live = p.rsplit > 0;
  // wait until at most `ahead` whole stages (the youngest ones) are still in flight
  auto wait_ahead = [&](int ahead) {
    if (ahead >= 3) __builtin_amdgcn_s_waitcnt(GCT2_VMCNT_ONLY(3 * NDMA));
    else if (ahead == 2) __builtin_amdgcn_s_waitcnt(GCT2_VMCNT_ONLY(2 * NDMA));
    else if (ahead == 1) __builtin_amdgcn_s_waitcnt(GCT2_VMCNT_ONLY(NDMA));
    else __builtin_amdgcn_s_waitcnt(GCT2_VMCNT_ONLY(0));
  };
  auto stage = [&](int st, const char* cur, char* tgt) {
    if (st + NST - 1 < st_hi) issue(st + NST - 1, tgt);
    if (live) compute(cur);
    // stages issued so far: up to min(st + NST - 1, st_hi - 1); stage st + 1 must have landed: those beyond it may stay in flight
    wait_ahead(min(st + NST - 1, st_hi - 1) - (st + 1));
    __builtin_amdgcn_s_barrier();
  };
  WG_STAMP(1);
  issue(st_lo, lds0);
  if (st_lo + 1 < st_hi) issue(st_lo + 1, lds1);
  if (st_lo + 2 < st_hi) issue(st_lo + 2, lds2);
  if (NST == 5 && st_lo + 3 < st_hi) issue(st_lo + 3, lds3);
  wait_ahead(min(st_lo + NST - 2, st_hi - 1) - st_lo);             // stage st_lo has landed
  __builtin_amdgcn_s_barrier();
  if constexpr (NST == 4) {
    for (int st = st_lo; st < st_hi; st += 4) {                   // NST stages per trip: buffer roles are compile-time
      stage(st, lds0, lds3);
      if (st + 1 >= st_hi) break;
      stage(st + 1, lds1, lds0);
      if (st + 2 >= st_hi) break;
      stage(st + 2, lds2, lds1);
      if (st + 3 >= st_hi) break;
      stage(st + 3, lds3, lds2);
    }
  } else {
    for (int st = st_lo; st < st_hi; st += 5) {
      stage(st, lds0, lds4);
      if (st + 1 >= st_hi) break;
      stage(st + 1, lds1, lds0);
      if (st + 2 >= st_hi) break;
      stage(st + 2, lds2, lds1);
      if (st + 3 >= st_hi) break;
      stage(st + 3, lds3, lds2);
      if (st + 4 >= st_hi) break;
      stage(st + 4, lds4, lds3);
    }
  }
  WG_STAMP(2);
  float* __restrict__ out = p.ws ? p.ws + (size_t)split * GC * Cs : p.dw;
  const int mode = p.ws ? 2 : (p.rsplit == 1 ? 1 : 0);
  // an opaque copy of the lane id: keeps hipcc from hoisting the 32 tiles' output addresses above the reduction loop (they would
  // occupy ~64 registers for the whole kernel: 440 spilled registers and a 10x slower kernel, measured r02)
  int elane = lane;
  asm volatile("" : "+v"(elane));
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int row = gc0 + wm * 128 + i * 16 + (elane & 15);
    if (row >= GC) continue;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int col = cs0 + wn * 64 + j * 16 + 4 * (elane >> 4);
      if (col >= Cs) continue;
      float* q = out + (size_t)row * Cs + col;
      if (mode == 2) *reinterpret_cast<f32x4_t*>(q) = acc[i][j];
      else if (mode == 1) { if (p.accumulate) *reinterpret_cast<f32x4_t*>(q) += acc[i][j]; else *reinterpret_cast<f32x4_t*>(q) = acc[i][j]; }
      else {
#pragma unroll
        for (int r = 0; r < 4; r++) atomicAdd(q + r, acc[i][j][r]);
      }
    }
  }
#ifdef GCT2_STAMP
  WG_STAMP(3);
  if (p.stamps && lane == 0) {
    unsigned long long* o = p.stamps + ((size_t)blockIdx.x * 8 + wave) * 4;
    for (int q = 0; q < 4; q++) o[q] = st[q];
  }
#endif
}

// ---- the same pipeline with a lean stage (r03) ---------------------------------------------------------------------------------
// wgrad256p_kernel's stage carries ~80 vector instructions besides its 32 MFMAs and 24 transposed reads (the ISA shows exec-masked
// branches around three 64-bit multiply-adds per gathered piece, ~40 instructions that rebuild the 12 fragment addresses, a chain of
// scalar branches for the wait count): with two lock-stepped waves per SIMD that is more than the issue slots the MFMAs leave free.
// Here: (a) the gather addresses advance incrementally (adds and selects, no multiply, no branch: one stage = 32 rows further, with
// carries into the next image row / image), tap validity from four precomputed per-lane flags; (b) the fragment addresses are
// lane offsets computed ONCE, plus the stage's compile-time base; (c) ONE LDS array (the DMA is hidden inline asm, so hipcc has
// nothing to drain); (d) a constant vmcnt in the steady state.  Same arithmetic in the same order: bit-identical results.
template <typename T>
__global__ __launch_bounds__(512, 2) void wgrad256q_kernel(WgradParams p) {
  constexpr int NST = 5;
  constexpr int IMG = 32 * 256;                                   // one T image of a stage: 32 r-rows x 128 columns
  constexpr int STAGE = 4 * IMG;
  constexpr int NDMA = 4;
  __shared__ __attribute__((aligned(16))) char lds[NST * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 3, wm = wave >> 2;
  const int Hs = p.Hs, Ws = p.Ws, Cb = p.Cb, Cs = p.Cs;
  const int Hb = 2 * Hs, Wb = 2 * Ws;
  const int R = p.B * Hs * Ws;
  const int GC = 16 * Cb;
  const int tiles_n = (Cs + 255) / 256;
  const int tiles = ((GC + 255) / 256) * tiles_n;
  int tile, split;
  if (p.rsplit >= 8) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    tile = j % tiles;
    split = (j / tiles) * 8 + xcd;
    if (split >= p.rsplit) return;
  } else {
    tile = blockIdx.x % tiles;
    split = blockIdx.x / tiles;
  }
  const int gc0 = (tile / tiles_n) * 256, cs0 = (tile % tiles_n) * 256;
  const int steps_total = (R + 63) / 64;
  stagger_start(p.stagger);
  const int steps_per = (steps_total + p.rsplit - 1) / p.rsplit;
  const int step_lo = split * steps_per;
  const int step_hi = min(steps_total, step_lo + steps_per);
  if (step_lo >= step_hi) return;
  const int st_lo = 2 * step_lo, st_hi = 2 * step_hi;

  const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(p.big), rs_s = make_rsrc(p.small);
  const int row0 = 4 * wave + (lane >> 4);
  const int lc = ((((lane & 15) >> 1) ^ timg_swz(row0)) << 1) | (lane & 1);
  const int ldb2 = p.ldbig * 2, lds2b = p.ldsmall * 2;
  // per image g of the big operand: byte offset of tap (kh, kw) / channel cb relative to pixel (2 sh, 2 sw), and the four cases in
  // which the tap leaves the image: kh = 0 at the top row, kh = 3 at the bottom row, kw = 0 / 3 at the left / right column
  // (bit masks, not bools: chains of && on per-lane conditions compile to exec-masked branches - 20 scalar branches per stage)
  int dg[2];
  unsigned edge[2];                                                // bit 0: kh = 0, 1: kh = 3, 2: kw = 0, 3: kw = 3; bit 4: the image is out of range
  unsigned s_bad[2];                                               // small operand: column block out of range
#pragma unroll
  for (int g = 0; g < 2; g++) {
    const int gc = gc0 + 128 * g + lc * 8;
    const bool ok = gc < GC;
    const int tap = ok ? gc / Cb : 0;
    const int cb = ok ? gc - tap * Cb : 0;
    const int kh = tap >> 2, kw = tap & 3;
    dg[g] = ((kh - 1) * Wb + (kw - 1)) * ldb2 + cb * 2;
    edge[g] = (kh == 0 ? 1u : 0u) | (kh == 3 ? 2u : 0u) | (kw == 0 ? 4u : 0u) | (kw == 3 ? 8u : 0u) | (ok ? 0u : 16u);
    s_bad[g] = (cs0 + 128 * g + lc * 8) < Cs ? 0u : 16u;
  }
  // this lane's row of the current issue stage: r, its (image, row, column) on the small grid, the byte offsets of pixel (2 sh, 2 sw)
  // of the big tensor and of row r of the small one; advanced by 32 rows per issued stage
  const int adv_w = 32 % Ws, q1 = 32 / Ws, adv_h = q1 % Hs, adv_b = q1 / Hs;
  const int pixA = adv_w * 2 * ldb2, pixB = Wb * ldb2, pixCD = (adv_h * 2 * Wb + adv_b * Hb * Wb) * ldb2;
  int r = st_lo * 32 + row0;
  int rw = r % Ws, rh, rb;
  { const int t = r / Ws; rh = t % Hs; rb = t / Hs; }
  unsigned pix = (unsigned)(((rb * Hb + 2 * rh) * Wb + 2 * rw) * ldb2);
  unsigned soff = (unsigned)(r * lds2b + (cs0 + lc * 8) * 2);
  auto issue = [&](char* base) {                                  // stages are issued in increasing order
    // where this row sits: bit 0 top row, 1 bottom row, 2 left column, 3 right column; bit 4: beyond the last row (always "bad")
    const unsigned pos = (rh == 0 ? 1u : 0u) | (rh == Hs - 1 ? 2u : 0u) | (rw == 0 ? 4u : 0u) | (rw == Ws - 1 ? 8u : 0u) | (r < R ? 0u : 16u);
    char* piece = base + wave * 1024;
#pragma unroll
    for (int g = 0; g < 2; g++) {
      const unsigned badb = (edge[g] & pos & 15u) | ((edge[g] | pos) & 16u);
      const unsigned bads = (s_bad[g] | pos) & 16u;
      dma16_hidden(rs_b, piece + g * IMG, badb ? OOB : pix + (unsigned)dg[g]);
      dma16_hidden(rs_s, piece + (2 + g) * IMG, bads ? OOB : soff + (unsigned)(g * 256));
    }
    // 32 rows further
    r += 32; soff += (unsigned)(32 * lds2b);
    rw += adv_w; pix += (unsigned)pixA;
    const bool cw = rw >= Ws;
    rw -= cw ? Ws : 0; rh += cw ? 1 : 0; pix += cw ? (unsigned)pixB : 0u;
    rh += adv_h; rb += adv_b; pix += (unsigned)pixCD;
    const bool ch = rh >= Hs;                                     // (into the next image: the byte offset is already right, Hb = 2 Hs)
    rh -= ch ? Hs : 0; rb += ch ? 1 : 0;
  };

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment addresses: lane offsets inside a stage buffer, computed once (k0 = 8 (lane>>4) + ((lane>>2)&3), two transposed reads
  // 4 rows apart: the swizzle of row k0 + 4 equals that of row k0)
  int sf_off[4], bf_off[8];
  {
    const int g4 = lane >> 4, q = (lane >> 2) & 3, pq = lane & 3;
    const int k0 = 8 * g4 + q;
    const int swz = timg_swz(k0);
#pragma unroll
    for (int j = 0; j < 4; j++) sf_off[j] = (2 + (wn >> 1)) * IMG + k0 * 256 + (((((wn & 1) * 64 + j * 16) >> 4) ^ swz) << 5) + pq * 8;
#pragma unroll
    for (int i = 0; i < 8; i++) bf_off[i] = wm * IMG + k0 * 256 + ((i ^ swz) << 5) + pq * 8;
  }
  auto frag = [&](const char* base, int off) -> u32x4_t {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(base + off));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(base + off + 4 * 256));
    const u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
    return u32x4_t{l2[0], l2[1], h2[0], h2[1]};
  };
  auto compute = [&](const char* base) {
    u32x4_t sf[4];
#pragma unroll
    for (int j = 0; j < 4; j++) sf[j] = frag(base, sf_off[j]);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const u32x4_t bf = frag(base, bf_off[i]);
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = mfma16<T>(sf[j], bf, acc[i][j]);
    }
  };

  const bool live = p.rsplit > 0;                                  // opaque code-generation fence (see wgrad256p_kernel)
  auto wait_tail = [&](int ahead) {                               // the last stages: fewer than three stages are still in flight
    if (ahead >= 3) __builtin_amdgcn_s_waitcnt(GCT2_VMCNT_ONLY(3 * NDMA));
    else if (ahead == 2) __builtin_amdgcn_s_waitcnt(GCT2_VMCNT_ONLY(2 * NDMA));
    else if (ahead == 1) __builtin_amdgcn_s_waitcnt(GCT2_VMCNT_ONLY(NDMA));
    else __builtin_amdgcn_s_waitcnt(GCT2_VMCNT_ONLY(0));
  };
  auto stage = [&](int st, const char* cur, char* tgt) {
    const bool more = st + NST - 1 < st_hi;                        // block-uniform
    if (more) issue(tgt);
    if (live) compute(cur);
    if (more) __builtin_amdgcn_s_waitcnt(GCT2_VMCNT_ONLY(3 * NDMA));   // steady state: stages st+2 .. st+4 stay in flight
    else wait_tail(st_hi - 1 - (st + 1));
    __builtin_amdgcn_s_barrier();
  };
  issue(lds);
  if (st_lo + 1 < st_hi) issue(lds + STAGE);
  if (st_lo + 2 < st_hi) issue(lds + 2 * STAGE);
  if (st_lo + 3 < st_hi) issue(lds + 3 * STAGE);
  wait_tail(min(st_lo + 3, st_hi - 1) - st_lo);
  __builtin_amdgcn_s_barrier();
  // steady state: every stage of the trip still has a stage to issue (st + 4 + 4 < st_hi): no tail logic, one constant wait
  auto stage_fast = [&](const char* cur, char* tgt) {
    issue(tgt);
    if (live) compute(cur);
    __builtin_amdgcn_s_waitcnt(GCT2_VMCNT_ONLY(3 * NDMA));
    __builtin_amdgcn_s_barrier();
  };
  int st = st_lo;
  for (; st + 8 < st_hi; st += 5) {
    stage_fast(lds, lds + 4 * STAGE);
    stage_fast(lds + STAGE, lds);
    stage_fast(lds + 2 * STAGE, lds + STAGE);
    stage_fast(lds + 3 * STAGE, lds + 2 * STAGE);
    stage_fast(lds + 4 * STAGE, lds + 3 * STAGE);
  }
  for (; st < st_hi; st += 5) {                                   // the last trips (same buffer roles: st - st_lo is a multiple of 5)
    stage(st, lds, lds + 4 * STAGE);
    if (st + 1 >= st_hi) break;
    stage(st + 1, lds + STAGE, lds);
    if (st + 2 >= st_hi) break;
    stage(st + 2, lds + 2 * STAGE, lds + STAGE);
    if (st + 3 >= st_hi) break;
    stage(st + 3, lds + 3 * STAGE, lds + 2 * STAGE);
    if (st + 4 >= st_hi) break;
    stage(st + 4, lds + 4 * STAGE, lds + 3 * STAGE);
  }
  float* __restrict__ out = p.ws ? p.ws + (size_t)split * GC * Cs : p.dw;
  const int mode = p.ws ? 2 : (p.rsplit == 1 ? 1 : 0);
  int elane = lane;
  asm volatile("" : "+v"(elane));
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int row = gc0 + wm * 128 + i * 16 + (elane & 15);
    if (row >= GC) continue;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int col = cs0 + wn * 64 + j * 16 + 4 * (elane >> 4);
      if (col >= Cs) continue;
      float* q = out + (size_t)row * Cs + col;
      if (mode == 2) *reinterpret_cast<f32x4_t*>(q) = acc[i][j];
      else if (mode == 1) { if (p.accumulate) *reinterpret_cast<f32x4_t*>(q) += acc[i][j]; else *reinterpret_cast<f32x4_t*>(q) = acc[i][j]; }
      else {
#pragma unroll
        for (int rr = 0; rr < 4; rr++) atomicAdd(q + rr, acc[i][j][rr]);
      }
    }
  }
}

// ---- the lean pipeline with the fragments of the NEXT stage read during the MFMAs of the current one (r03) ---------------------
// wgrad256r_kernel's stage is [DMA issue + address updates] -> [24 transposed reads, drained] -> [32 MFMAs] -> [wait, barrier], in
// lock step on all 8 waves: the LDS pipe and the matrix cores take turns (tests/hw_probe/probe_wavetile.hip: that structure tops
// out at 1.51-1.59 PFLOP/s with no global traffic at all).  Here a wave keeps TWO fragment sets in registers: while the MFMAs of
// stage s run on one, the reads of stage s + 1 (already landed and barrier-visible) fill the other - in two halves so that the
// second half reuses the registers the first 16 MFMAs have freed.  The ring is the same five buffers: the buffer of stage s is free
// as soon as every wave holds its fragments (the barrier that ends stage s - 1), so stage s + 5 is issued into it during stage s:
// still four stages in flight.  Same multiplies in the same order per accumulator: bit-identical to wgrad256p / wgrad256q.
// (original comment of the lean stage follows)
// wgrad256p_kernel's stage carries ~80 vector instructions besides its 32 MFMAs and 24 transposed reads (the ISA shows exec-masked
// branches around three 64-bit multiply-adds per gathered piece, ~40 instructions that rebuild the 12 fragment addresses, a chain of
// scalar branches for the wait count): with two lock-stepped waves per SIMD that is more than the issue slots the MFMAs leave free.
// Here: (a) the gather addresses advance incrementally (adds and selects, no multiply, no branch: one stage = 32 rows further, with
// carries into the next image row / image), tap validity from four precomputed per-lane flags; (b) the fragment addresses are
// lane offsets computed ONCE, plus the stage's compile-time base; (c) ONE LDS array (the DMA is hidden inline asm, so hipcc has
// nothing to drain); (d) a constant vmcnt in the steady state.  Same arithmetic in the same order: bit-identical results.
template <typename T>
__global__ __launch_bounds__(512, 2) void wgrad256r_kernel(WgradParams p) {
  constexpr int NST = 5;
  constexpr int IMG = 32 * 256;                                   // one T image of a stage: 32 r-rows x 128 columns
  constexpr int STAGE = 4 * IMG;
  constexpr int NDMA = 4;
  __shared__ __attribute__((aligned(16))) char lds[NST * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 3, wm = wave >> 2;
  const int Hs = p.Hs, Ws = p.Ws, Cb = p.Cb, Cs = p.Cs;
  const int Hb = 2 * Hs, Wb = 2 * Ws;
  const int R = p.B * Hs * Ws;
  const int GC = 16 * Cb;
  const int tiles_n = (Cs + 255) / 256;
  const int tiles = ((GC + 255) / 256) * tiles_n;
  int tile, split;
  if (p.rsplit >= 8) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    tile = j % tiles;
    split = (j / tiles) * 8 + xcd;
    if (split >= p.rsplit) return;
  } else {
    tile = blockIdx.x % tiles;
    split = blockIdx.x / tiles;
  }
  const int gc0 = (tile / tiles_n) * 256, cs0 = (tile % tiles_n) * 256;
  const int steps_total = (R + 63) / 64;
  stagger_start(p.stagger);
  const int steps_per = (steps_total + p.rsplit - 1) / p.rsplit;
  const int step_lo = split * steps_per;
  const int step_hi = min(steps_total, step_lo + steps_per);
  if (step_lo >= step_hi) return;
  const int st_lo = 2 * step_lo, st_hi = 2 * step_hi;

  const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(p.big), rs_s = make_rsrc(p.small);
  const int row0 = 4 * wave + (lane >> 4);
  const int lc = ((((lane & 15) >> 1) ^ timg_swz(row0)) << 1) | (lane & 1);
  const int ldb2 = p.ldbig * 2, lds2b = p.ldsmall * 2;
  // per image g of the big operand: byte offset of tap (kh, kw) / channel cb relative to pixel (2 sh, 2 sw), and the four cases in
  // which the tap leaves the image: kh = 0 at the top row, kh = 3 at the bottom row, kw = 0 / 3 at the left / right column
  // (bit masks, not bools: chains of && on per-lane conditions compile to exec-masked branches - 20 scalar branches per stage)
  int dg[2];
  unsigned edge[2];                                                // bit 0: kh = 0, 1: kh = 3, 2: kw = 0, 3: kw = 3; bit 4: the image is out of range
  unsigned s_bad[2];                                               // small operand: column block out of range
#pragma unroll
  for (int g = 0; g < 2; g++) {
    const int gc = gc0 + 128 * g + lc * 8;
    const bool ok = gc < GC;
    const int tap = ok ? gc / Cb : 0;
    const int cb = ok ? gc - tap * Cb : 0;
    const int kh = tap >> 2, kw = tap & 3;
    dg[g] = ((kh - 1) * Wb + (kw - 1)) * ldb2 + cb * 2;
    edge[g] = (kh == 0 ? 1u : 0u) | (kh == 3 ? 2u : 0u) | (kw == 0 ? 4u : 0u) | (kw == 3 ? 8u : 0u) | (ok ? 0u : 16u);
    s_bad[g] = (cs0 + 128 * g + lc * 8) < Cs ? 0u : 16u;
  }
  // this lane's row of the current issue stage: r, its (image, row, column) on the small grid, the byte offsets of pixel (2 sh, 2 sw)
  // of the big tensor and of row r of the small one; advanced by 32 rows per issued stage
  const int adv_w = 32 % Ws, q1 = 32 / Ws, adv_h = q1 % Hs, adv_b = q1 / Hs;
  const int pixA = adv_w * 2 * ldb2, pixB = Wb * ldb2, pixCD = (adv_h * 2 * Wb + adv_b * Hb * Wb) * ldb2;
  int r = st_lo * 32 + row0;
  int rw = r % Ws, rh, rb;
  { const int t = r / Ws; rh = t % Hs; rb = t / Hs; }
  unsigned pix = (unsigned)(((rb * Hb + 2 * rh) * Wb + 2 * rw) * ldb2);
  unsigned soff = (unsigned)(r * lds2b + (cs0 + lc * 8) * 2);
  auto issue = [&](char* base) {                                  // stages are issued in increasing order
    // where this row sits: bit 0 top row, 1 bottom row, 2 left column, 3 right column; bit 4: beyond the last row (always "bad")
    const unsigned pos = (rh == 0 ? 1u : 0u) | (rh == Hs - 1 ? 2u : 0u) | (rw == 0 ? 4u : 0u) | (rw == Ws - 1 ? 8u : 0u) | (r < R ? 0u : 16u);
    char* piece = base + wave * 1024;
#pragma unroll
    for (int g = 0; g < 2; g++) {
      const unsigned badb = (edge[g] & pos & 15u) | ((edge[g] | pos) & 16u);
      const unsigned bads = (s_bad[g] | pos) & 16u;
      dma16_hidden(rs_b, piece + g * IMG, badb ? OOB : pix + (unsigned)dg[g]);
      dma16_hidden(rs_s, piece + (2 + g) * IMG, bads ? OOB : soff + (unsigned)(g * 256));
    }
    // 32 rows further
    r += 32; soff += (unsigned)(32 * lds2b);
    rw += adv_w; pix += (unsigned)pixA;
    const bool cw = rw >= Ws;
    rw -= cw ? Ws : 0; rh += cw ? 1 : 0; pix += cw ? (unsigned)pixB : 0u;
    rh += adv_h; rb += adv_b; pix += (unsigned)pixCD;
    const bool ch = rh >= Hs;                                     // (into the next image: the byte offset is already right, Hb = 2 Hs)
    rh -= ch ? Hs : 0; rb += ch ? 1 : 0;
  };

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment addresses: lane offsets inside a stage buffer, computed once (k0 = 8 (lane>>4) + ((lane>>2)&3), two transposed reads
  // 4 rows apart: the swizzle of row k0 + 4 equals that of row k0)
  // (fragment index i or j enters the offset as (idx ^ swz) << 5 = base ^ (idx << 5): ONE register per operand, one XOR per fragment)
  int sf_off0, bf_off0;
  {
    const int g4 = lane >> 4, q = (lane >> 2) & 3, pq = lane & 3;
    const int k0 = 8 * g4 + q;
    const int swz = timg_swz(k0);
    sf_off0 = (2 + (wn >> 1)) * IMG + k0 * 256 + (((4 * (wn & 1)) ^ swz) << 5) + pq * 8;
    bf_off0 = wm * IMG + k0 * 256 + (swz << 5) + pq * 8;
  }
  auto sf_off_ = [&](int j) { return sf_off0 ^ (j << 5); };
  auto bf_off_ = [&](int i) { return bf_off0 ^ (i << 5); };
  auto frag = [&](const char* base, int off) -> u32x4_t {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(base + off));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(base + off + 4 * 256));
    const u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
    return u32x4_t{l2[0], l2[1], h2[0], h2[1]};
  };
  // Registers: 128 accumulators + two small-operand sets (2 x 16) + a rolling window of FOUR big-operand fragments (16): the big
  // fragment of row i + 3 is read while row i is multiplied - across the stage boundary too (rows 5..7 fetch rows 0..2 of the next
  // stage, whose buffer is already visible), the next small set in the middle of the stage.  Two full sets (96) + 128 do not fit 256.
  u32x4_t sfr[2][4], bfw[4];
  auto wait_groups = [&](int g) {                                 // at most g DMA groups (stages) of this wave still in flight
    if (g >= 3) __builtin_amdgcn_s_waitcnt(GCT2_VMCNT_ONLY(3 * NDMA));
    else if (g == 2) __builtin_amdgcn_s_waitcnt(GCT2_VMCNT_ONLY(2 * NDMA));
    else if (g == 1) __builtin_amdgcn_s_waitcnt(GCT2_VMCNT_ONLY(NDMA));
    else __builtin_amdgcn_s_waitcnt(GCT2_VMCNT_ONLY(0));
  };
  // stage st: small set x and big rows 0..2 in registers; buffer `cur` holds stage st, `nxt` stage st + 1 (landed, visible), `tgt`
  // held stage st - 1 (every read of it was consumed before the last barrier): stage st + 4 goes there
  auto stage_r = [&](auto fast_c, int st, int x, const char* cur, const char* nxt, char* tgt) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(fast_c)::value;
    const bool more = FAST || st + NST - 1 < st_hi;                // a stage st + 4 to issue
    const bool next = FAST || st + 1 < st_hi;                      // a stage st + 1 to read
    if (more) issue(tgt);
    // opaque copies of the two lane offsets: without them the XOR-ed fragment addresses are loop-invariant, get hoisted out of the
    // trip and spilled - and every reload from scratch drains vmcnt, i.e. the whole DMA pipeline
    int so = sf_off0, bo = bf_off0;
    asm volatile("" : "+v"(so), "+v"(bo));
#pragma unroll
    for (int i = 0; i < 8; i++) {
      __builtin_amdgcn_sched_barrier(0);
      if (i + 3 < 8) bfw[(i + 3) & 3] = frag(cur, bo ^ ((i + 3) << 5));
      else if (next) bfw[(i + 3) & 3] = frag(nxt, bo ^ ((i + 3 - 8) << 5));
      if (i == 3 && next) {
#pragma unroll
        for (int j = 0; j < 4; j++) sfr[1 - x][j] = frag(nxt, so ^ (j << 5));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = mfma16<T>(sfr[x][j], bfw[i & 3], acc[i][j]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (FAST) __builtin_amdgcn_s_waitcnt(GCT2_VMCNT_ONLY(2 * NDMA));   // stage st + 2 has landed; st + 3, st + 4 stay in flight
    else wait_groups(min(max(st_hi - 1 - (st + 2), 0), 2));
    __builtin_amdgcn_s_barrier();
  };
  // prologue: four stages in flight, small set 0 and big rows 0..2 of stage st_lo in registers, stage st_lo + 1 visible
  issue(lds);
  if (st_lo + 1 < st_hi) issue(lds + STAGE);
  if (st_lo + 2 < st_hi) issue(lds + 2 * STAGE);
  if (st_lo + 3 < st_hi) issue(lds + 3 * STAGE);
  wait_groups(min(st_hi - 1 - st_lo, 3));
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int j = 0; j < 4; j++) sfr[0][j] = frag(lds, sf_off0 ^ (j << 5));
#pragma unroll
  for (int i = 0; i < 3; i++) bfw[i] = frag(lds, bf_off0 ^ (i << 5));
  wait_groups(min(max(st_hi - 1 - (st_lo + 1), 0), 2));
  __builtin_amdgcn_s_barrier();
#define GCT2_WR_STAGE(FAST, K)                                                                                        \
  {                                                                                                                   \
    if (!FAST && st + (K) >= st_hi) break;                                                                            \
    stage_r(std::integral_constant<bool, FAST>{}, st + (K), (K) & 1, lds + ((K) % 5) * STAGE, lds + (((K) + 1) % 5) * STAGE, \
            lds + (((K) + 4) % 5) * STAGE);                                                                            \
  }
#define GCT2_WR_TRIP(FAST)                                                                                            \
  GCT2_WR_STAGE(FAST, 0) GCT2_WR_STAGE(FAST, 1) GCT2_WR_STAGE(FAST, 2) GCT2_WR_STAGE(FAST, 3) GCT2_WR_STAGE(FAST, 4)     \
  GCT2_WR_STAGE(FAST, 5) GCT2_WR_STAGE(FAST, 6) GCT2_WR_STAGE(FAST, 7) GCT2_WR_STAGE(FAST, 8) GCT2_WR_STAGE(FAST, 9)
  int st = st_lo;
  for (; st + 9 + NST - 1 < st_hi; st += 10) { GCT2_WR_TRIP(true) }   // every stage of the trip still issues a stage (st + 9 + 4 < st_hi)
  for (; st < st_hi; st += 10) { GCT2_WR_TRIP(false) }           // the last trips (same roles: st - st_lo is a multiple of 10)
#undef GCT2_WR_TRIP
#undef GCT2_WR_STAGE
  float* __restrict__ out = p.ws ? p.ws + (size_t)split * GC * Cs : p.dw;
  const int mode = p.ws ? 2 : (p.rsplit == 1 ? 1 : 0);
  int elane = lane;
  asm volatile("" : "+v"(elane));
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int row = gc0 + wm * 128 + i * 16 + (elane & 15);
    if (row >= GC) continue;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int col = cs0 + wn * 64 + j * 16 + 4 * (elane >> 4);
      if (col >= Cs) continue;
      float* q = out + (size_t)row * Cs + col;
      if (mode == 2) *reinterpret_cast<f32x4_t*>(q) = acc[i][j];
      else if (mode == 1) { if (p.accumulate) *reinterpret_cast<f32x4_t*>(q) += acc[i][j]; else *reinterpret_cast<f32x4_t*>(q) = acc[i][j]; }
      else {
#pragma unroll
        for (int rr = 0; rr < 4; rr++) atomicAdd(q + rr, acc[i][j][rr]);
      }
    }
  }
}

// dw[e] += sum_s slab[s][e], 4 elements per thread, slabs added in index order
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, size_t n4, int nsplit,
                                                            int accumulate) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const f32x4_t* src = reinterpret_cast<const f32x4_t*>(ws) + i;
  f32x4_t a = src[0];
  int s = 1;
  for (; s + 8 <= nsplit; s += 8) {          // 8 independent slab loads in flight, added in slab order
    f32x4_t t[8];
#pragma unroll
    for (int u = 0; u < 8; u++) t[u] = src[(size_t)(s + u) * n4];
#pragma unroll
    for (int u = 0; u < 8; u++) a += t[u];
  }
  for (; s < nsplit; s++) a += src[(size_t)s * n4];
  if (accumulate) a += reinterpret_cast<const f32x4_t*>(dw)[i];   // without it the gradient is written, not read: 4 B/parameter less
  reinterpret_cast<f32x4_t*>(dw)[i] = a;
}

// the same sum for a SMALL tensor left as MANY slabs (the 3-channel layer: 6 K elements x 512 slabs): one thread per float4 would
// walk all slabs alone (latency-bound: 29 us on the critical tail of the step).  Work-group = 16 float4 columns x 16 slab groups;
// group g adds slabs [g*per, (g+1)*per) in index order, the 16 group sums are added in group order: a fixed order, bit-reproducible.
__global__ __launch_bounds__(256) void wgrad_reduce_wide_kernel(const float* __restrict__ ws, float* __restrict__ dw, size_t n4, int nsplit,
                                                                 int accumulate) {
  const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const size_t i = (size_t)blockIdx.x * 16 + col;
  const int per = (nsplit + 15) / 16;
  const int s_lo = grp * per, s_hi = min(nsplit, s_lo + per);
  f32x4_t a = {0.f, 0.f, 0.f, 0.f};
  if (i < n4) {
    const f32x4_t* src = reinterpret_cast<const f32x4_t*>(ws) + i;
    int s = s_lo;
    for (; s + 8 <= s_hi; s += 8) {
      f32x4_t t[8];
#pragma unroll
      for (int u = 0; u < 8; u++) t[u] = __builtin_nontemporal_load(src + (size_t)(s + u) * n4);
#pragma unroll
      for (int u = 0; u < 8; u++) a += t[u];
    }
    for (; s < s_hi; s++) a += __builtin_nontemporal_load(src + (size_t)s * n4);
  }
  __shared__ f32x4_t red[16][16];
  red[grp][col] = a;
  __syncthreads();
  if (grp == 0 && i < n4) {
    f32x4_t t = red[0][col];
#pragma unroll
    for (int g = 1; g < 16; g++) t += red[g][col];
    if (accumulate) t += reinterpret_cast<const f32x4_t*>(dw)[i];
    reinterpret_cast<f32x4_t*>(dw)[i] = t;
  }
}

}  // namespace

bool wgrad_mfma_supported(int dtype, const WgradParams& p) {
  if (dtype != GCT2_BF16 && dtype != GCT2_F16) return false;
  if (p.Cb % 8 || p.Cs % 8 || p.ldbig % 8 || p.ldsmall % 8) return false;
  if ((uintptr_t)p.big % 16 || (uintptr_t)p.small % 16 || (uintptr_t)p.dw % 16) return false;    // 16-byte loads and stores
  if (p.ks && (p.ks < 1 || p.ks > 7 || !(p.ks & 1))) return false;
  const size_t big_bytes = (size_t)p.B * p.Hs * p.Ws * (p.ks ? 1 : 4) * p.ldbig * 2, small_bytes = (size_t)p.B * p.Hs * p.Ws * p.ldsmall * 2;
  if (big_bytes >= 0x7ff00000u || small_bytes >= 0x7ff00000u) return false;     // 31-bit buffer offsets
  return true;
}

int wgrad_mfma(const gct2_ctx& c, int dtype, WgradParams p, hipStream_t s, WgradSlabs* defer) {
  const gct2_adam_args* want_adam = defer ? defer->want_adam : nullptr;
  if (defer) *defer = WgradSlabs{nullptr, 0, 0};
  const int g_wgrad_variant = c.wgrad_variant, g_wgrad_target = c.wgrad_target, g_wgrad_slab_max = c.wgrad_slab_max, g_wgrad_pipe = c.wgrad_pipe;
  const int R = p.B * p.Hs * p.Ws;
  const int taps = p.ks ? p.ks * p.ks : 16;          // p.ks != 0: stride-1 'same' convolution (128 x 128 tile kernels only)
  // 256 x 256 tile (one work-group per CU) whenever the 128 x 128 tiling would have to split the reduction anyway
  // (fewer than 512 tiles): measured -10..-22 % on U0/U1/U2/D1/D2 (profiles/r01_wgrad_variants.txt)
  const int tiles128 = ((taps * p.Cb + 127) / 128) * ((p.Cs + 127) / 128);
  const int tiles256 = ((taps * p.Cb + 255) / 256) * ((p.Cs + 255) / 256);
  // (the r03 band of 256 .. 511 small tiles asks for c.wgrad_big_minsteps = 8 steps per split; below 256 the r02 rule - 4 - stays:
  // a small-batch UpShuffle_0 would otherwise fall to 48 pixel splits of the small tile, i.e. to atomics)
  const int minsteps = tiles128 < 256 ? std::min(4, c.wgrad_big_minsteps) : c.wgrad_big_minsteps;
  const int blocks256 = tiles256 * std::max(1, std::min((g_wgrad_target + tiles256 - 1) / tiles256, ((R + 63) / 64) / minsteps));
  // ... and only if the big tiling still yields ~one work-group per CU (the 2x2 / 4x4 bottleneck levels have too few pixels)
  // ... and only below 256 small tiles: from there on the 128 x 128 tiling fills the chip with at most TWO pixel splits, and a
  // big-tile launch always leaves 256 work-groups x 256 KiB = 64 MiB of slabs (written here, read back by the optimizer), whatever
  // the size of the tensor.  Measured r02 (scripts/bench_wgrad.py, incl. the slab sum): DownShuffle_4 35 -> 18 us (one owner per
  // tile, no slabs), DownShuffle_3 57 -> 48, UpShuffle_2 153-160 -> 145 (two splits: 32 MiB of slabs instead of 64); in the step -5..-25 us (in-process A/B)
  // (going further down - the small tile with ~512 work-groups for DownShuffle_1/2 and UpShuffle_1, 32 MiB of slabs each - is faster
  // launch by launch (92 -> 80, 84 -> 80, 145 -> 143 us incl. the slab sum) and SLOWER in the step: +38 us in an in-process A/B, the
  // small work-groups interleave with the input-gradient chain's instead of alternating with them)
  // r03: with the lean stage the big tile wins UpShuffle_2 too (140 -> 130 us) and the limit moved to 512 small tiles: DownShuffle_3
  // comes along (47 -> 52 us alone) and the step is still 11 us shorter (profiles/r03_step_ab.txt); tuning bit 9 = the r02 limit.
  // DownShuffle_4 (16 steps of 64 rows) stays on the one-owner 128 x 128 tile: c.wgrad_big_minsteps = 8 steps per split
  const bool auto_tile = g_wgrad_variant == 0 || g_wgrad_variant == 6;
  const bool big_tile0 = !p.ks && (g_wgrad_variant == 2 || g_wgrad_variant == 4 || g_wgrad_variant == 5 || g_wgrad_variant == 8 || g_wgrad_variant == 9 || (auto_tile && tiles128 < c.wgrad_big_limit && blocks256 >= 192));
  // the five-stage pipeline runs the lean stage (wgrad256q_kernel, r03: -7..-16 % on the five big-tile layers) unless the tuning
  // word asks for the r02 stage code (variants 6 = automatic tile choice, 8 = 256 x 256 everywhere) or for four stages (bit 23)
  const bool lean_stage = g_wgrad_variant != 6 && g_wgrad_variant != 8 && c.wgrad_ring == 5;
  // variant 4 (r03): the 256 x 128 tile at two work-groups per CU in place of the 256 x 256 pipeline
  const bool tile2x = big_tile0 && g_wgrad_variant == 4;
  const bool big_tile = big_tile0 && !tile2x;
  const int tiles2x = ((taps * p.Cb + 255) / 256) * ((p.Cs + 127) / 128);
  const int tiles = tile2x ? tiles2x : (big_tile ? tiles256 : tiles128);
  const int steps_total = (R + 63) / 64;
  // aim at ~768 workgroups (3 per CU; 512 for the big tile) but keep >= 4 steps of 64 rows per split; one owner per tile
  // once the tiles alone give every CU a work-group
  int rsplit = tile2x ? (512 + tiles - 1) / tiles : big_tile ? (g_wgrad_target + tiles - 1) / tiles : (tiles >= 512 ? 1 : (768 + tiles - 1) / tiles);
  if (!big_tile && !tile2x && tiles >= 256 && tiles < 512) rsplit = steps_total >= 32 ? 2 : 1;    // two work-groups per CU once the reduction is long enough
  if (!big_tile && !tile2x && c.wgrad_split) rsplit = 1 << (c.wgrad_split - 1);
  rsplit = max(1, min(rsplit, steps_total / 4));
  const int per = (steps_total + rsplit - 1) / rsplit;
  rsplit = (steps_total + per - 1) / per;            // every split non-empty (each one owns a slab)
  p.rsplit = rsplit;
  p.ws = nullptr;
  p.stagger = c.stagger;
#ifdef GCT2_STAMP
  p.stamps = c.stamps;
#endif
  const size_t n = (size_t)taps * p.Cb * p.Cs;
  size_t ws_bytes = 0;
  float* ws = c.wgrad_scratch(&ws_bytes);
  // measured (scripts/bench_wgrad.py): slabs beat atomics up to ~24 splits; beyond that (a small-batch UpShuffle_0: 16 tiles x 48
  // splits) the many 1-MiB slabs cost a little more than the atomics they replace - and are taken all the same since r03: atomics
  // make the gradient depend on the arrival order (tuning bit 21 restores the r02 limit of 24)
  if (rsplit > 1 && rsplit <= ((big_tile || tile2x) ? 128 : g_wgrad_slab_max) && ws && n % 4 == 0 && (uintptr_t)p.dw % 16 == 0 && n * sizeof(float) * rsplit <= ws_bytes &&
      g_wgrad_variant != 7)
    p.ws = ws;
  // one owner per tile (no split), 4x4 layers on the 128 x 128 tile, operand copy in the compute dtype: the optimizer step the caller
  // asked for (gct2_adam_args) runs in the epilogue - the gradient never leaves the registers
  if (want_adam && c.wgrad_fuse_adam && rsplit == 1 && !big_tile && !tile2x && g_wgrad_variant != 1 && !p.ks && !p.accumulate && n % 4 == 0 &&
      (!want_adam->shadow || want_adam->shadow_dtype == dtype)) {
    p.adam.p = want_adam->p; p.adam.m = want_adam->m; p.adam.v = want_adam->v; p.adam.shadow = want_adam->shadow;
    p.adam.alpha = want_adam->alpha; p.adam.b1 = want_adam->beta1; p.adam.b2 = want_adam->beta2; p.adam.eps = want_adam->eps;
    p.adam.gmul = want_adam->grad_mul;
    defer->adam_done = true;
  }
  dim3 grid(rsplit >= 8 ? tiles * 8 * ((rsplit + 7) / 8) : tiles * rsplit);
  if (!p.ws && rsplit > 1 && !p.accumulate) {     // atomics add into the target: start it from zero
    (void)hipMemsetAsync(p.dw, 0, n * sizeof(float), s);
  }
  const bool one_buf = g_wgrad_variant == 1;
  if (p.ks) {
    if (dtype == GCT2_BF16) hipLaunchKernelGGL((wgrad_kernel<__bf16, 2, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((wgrad_kernel<_Float16, 2, true>), grid, dim3(256), 0, s, p);
  } else if (tile2x) {
    if (dtype == GCT2_BF16) hipLaunchKernelGGL(wgrad2x_kernel<__bf16>, grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL(wgrad2x_kernel<_Float16>, grid, dim3(512), 0, s, p);
  } else if (big_tile && g_wgrad_pipe && g_wgrad_variant == 9) {   // fragments of the next stage read during the MFMAs (A/B)
    if (dtype == GCT2_BF16) hipLaunchKernelGGL(wgrad256r_kernel<__bf16>, grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL(wgrad256r_kernel<_Float16>, grid, dim3(512), 0, s, p);
  } else if (big_tile && g_wgrad_pipe && lean_stage) {
    if (dtype == GCT2_BF16) hipLaunchKernelGGL(wgrad256q_kernel<__bf16>, grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL(wgrad256q_kernel<_Float16>, grid, dim3(512), 0, s, p);
  } else if (big_tile && g_wgrad_pipe) {
    if (c.wgrad_ring == 5) {
      if (dtype == GCT2_BF16) hipLaunchKernelGGL((wgrad256p_kernel<__bf16, 5>), grid, dim3(512), 0, s, p);
      else hipLaunchKernelGGL((wgrad256p_kernel<_Float16, 5>), grid, dim3(512), 0, s, p);
    } else {
      if (dtype == GCT2_BF16) hipLaunchKernelGGL((wgrad256p_kernel<__bf16, 4>), grid, dim3(512), 0, s, p);
      else hipLaunchKernelGGL((wgrad256p_kernel<_Float16, 4>), grid, dim3(512), 0, s, p);
    }
  } else if (big_tile) {
    if (dtype == GCT2_BF16) hipLaunchKernelGGL(wgrad256_kernel<__bf16>, grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL(wgrad256_kernel<_Float16>, grid, dim3(512), 0, s, p);
  } else if (dtype == GCT2_BF16) {
    if (one_buf) hipLaunchKernelGGL((wgrad_kernel<__bf16, 1>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((wgrad_kernel<__bf16, 2>), grid, dim3(256), 0, s, p);
  } else {
    if (one_buf) hipLaunchKernelGGL((wgrad_kernel<_Float16, 1>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((wgrad_kernel<_Float16, 2>), grid, dim3(256), 0, s, p);
  }
  if (p.ws && defer && !p.accumulate) *defer = WgradSlabs{p.ws, rsplit, n};     // the caller's optimizer kernel sums the slabs
  else if (p.ws) hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, p.ws, p.dw, n / 4, rsplit, p.accumulate);
  return gct2_check_launch("wgrad_mfma");
}

// dw (+)= sum of nsplit partial tensors of n elements each (n % 4 == 0), slab order: shared with the 3-channel layer (rgb_mfma.hip)
int wgrad_reduce(const float* ws, float* dw, size_t n, int nsplit, int accumulate, hipStream_t s) {
  if (n / 4 <= 16384 && nsplit >= 64) {        // small tensor, many slabs (the 3-channel layer)
    hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3((unsigned)((n / 4 + 15) / 16)), dim3(256), 0, s, ws, dw, n / 4, nsplit, accumulate);
    return gct2_check_launch("wgrad_reduce");
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, ws, dw, n / 4, nsplit, accumulate);
  return gct2_check_launch("wgrad_reduce");
}
