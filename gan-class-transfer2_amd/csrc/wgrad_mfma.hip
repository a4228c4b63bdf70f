// Weight-gradient GEMM of the 4x4/stride-2 convolutions (autodiff of train.py:148-153,161-166) on gfx950.
//
//   dW[gc][cs] += sum_r big[pix_big(r, tap(gc))][cb(gc)] * small[r][cs],   gc = tap*Cb + cb  (16*Cb rows)
//
// r walks the SMALL grid [B,Hs,Ws]; pix_big(r,(kh,kw)) = (2sh+kh-1, 2sw+kw-1) on the BIG grid (zero outside).
//   Conv2D          : big = layer input x (Cb = Cin),  small = dz (Cs = Cout)  -> dW (4,4,Cin,Cout)
//   Conv2DTranspose : big = dz (Cb = Cout),            small = x  (Cs = Cin)   -> dW (4,4,Cout,Cin)
// Both operands have the REDUCTION index r as their slow (row) index in memory, so both LDS tiles are
// "T images" ([r][128 channels]) consumed through ds_read_tr16_b64.  LDS-DMA staging (buffer_load ... lds; swizzle on
// the source address, out-of-range offset = zero fill).  Two kernels:
//   wgrad256q_kernel : 256 x 256 tile, 8 waves of 128 x 64, five 32-row stages in a ring (the DMA of stage s+4 is issued while stage s
//                      is multiplied, counted vmcnt across raw barriers), lean stage code; default where the tile count still gives
//                      about one work-group per CU
//   wgrad_kernel     : 128 x 128 tile, 4 waves, two 64-row buffers, 2 work-groups per CU (small layers, stride-1 convolutions)
// The reduction over r is split across work-groups (XCD-aware 1-D grid); partial tiles go to workspace slabs summed in a
// fixed order (by the fused optimizer launch or wgrad_reduce_kernel: reproducible), to their single owner, or to fp32 atomics.
// (r01-r03 also carried a two-buffer 256 x 256 tile, the r02 stage code of the ring, a rolling-window variant, a 256 x 128 tile at two
// work-groups per CU and Adam inside the epilogue: each measured equal or slower, DESIGN.md section 3, and removed in r04.)
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
// IMGAL (r05): a 64-row step covers WHOLE images of the small grid (64 % (Hs * Ws) == 0: every 128 x 128 launch of the reference topology -
// the 8 x 8 ... 1 x 1 grids of the bottleneck levels): a lane's position inside its image, its tap's validity and its byte offsets relative
// to the step's first image never change - the gather addresses are "constant + step x stride" (one add per piece) and the fragment
// addresses lane constants computed once, instead of ~90 vector instructions per step beside its 32 multiplies (decode of every row,
// three multiply-adds per gathered piece, the fragment addresses rebuilt per read).  Same sources, same zero fill, same multiplies in the
// same order: bit-identical to the general form.
template <typename T, bool S1 = false, bool IMGAL = false>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradParams p) {
  static_assert(!(S1 && IMGAL), "the image-aligned form exists for the 4x4 / stride-2 layers");
  constexpr int IMG = 64 * 256;
  const int KS = S1 ? p.ks : 4, STRIDE = S1 ? 1 : 2, PAD = S1 ? (p.ks - 1) / 2 : 1;   // compile-time constants for the 4x4 layers
  __shared__ __attribute__((aligned(16))) char lds0[2 * IMG];     // [big image | small image]
  __shared__ __attribute__((aligned(16))) char lds1[2 * IMG];

  GCT2_CLOCK_DECL;
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

  // ---- IMGAL: lane constants (row i of this lane sits at the same place of the same image-in-step in every step)
  unsigned cB[4], cS[4];
  bool vB[4];
  int rr4[4];
  const unsigned stepB = IMGAL ? (unsigned)((64 / (Hs * Ws)) * Hb * Wb * ldb2) : 0u, stepS = (unsigned)(64 * lds2);
  if constexpr (IMGAL) {
    const int hw = Hs * Ws;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int rr = row0 + 16 * i, img = rr / hw, pos = rr - img * hw, sh = pos / Ws, sw = pos - sh * Ws;
      const int h = 2 * sh + kh - 1, w = 2 * sw + kw - 1;
      rr4[i] = rr;
      vB[i] = gc_ok && (unsigned)h < (unsigned)Hb && (unsigned)w < (unsigned)Wb;
      cB[i] = (unsigned)(((img * Hb + h) * Wb + w) * ldb2 + cb * 2) + (unsigned)step_lo * stepB;
      cS[i] = (unsigned)(rr * lds2 + (cs0 + lc * 8) * 2) + (unsigned)step_lo * stepS;
    }
  }
  auto issue_al = [&](int step, char* base) {                     // steps are issued in increasing order: the offsets advance in place
    const int rem = R - step * 64;                                 // rows of this step that exist (uniform)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const bool in = rr4[i] < rem;
      dma16(rs_b, base + (wave + 4 * i) * 1024, (vB[i] && in) ? cB[i] : OOB);
      dma16(rs_s, base + IMG + (wave + 4 * i) * 1024, (cs_ok && in) ? cS[i] : OOB);
      cB[i] += stepB; cS[i] += stepS;
    }
  };
  auto issue_gen = [&](int step, char* base) {
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
  auto issue = [&](int step, char* base) {
    if constexpr (IMGAL) issue_al(step, base);
    else issue_gen(step, base);
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment addresses: lane offsets inside a T image, computed once (IMGAL); row k0 + 4 has the swizzle of row k0
  int bfo[2][4], sfo[2][4];
  if constexpr (IMGAL) {
    const int g4 = lane >> 4, q = (lane >> 2) & 3, pq = lane & 3;
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      const int k0 = kk * 32 + 8 * g4 + q, swz = timg_swz(k0);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        bfo[kk][i] = k0 * 256 + ((((wm * 64 + i * 16) >> 4) ^ swz) << 5) + pq * 8;
        sfo[kk][i] = IMG + k0 * 256 + ((((wn * 64 + i * 16) >> 4) ^ swz) << 5) + pq * 8;
      }
    }
  }
  auto frag_at = [&](const char* base, int off) -> u32x4_t {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(base + off));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(base + off + 4 * 256));
    const u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
    return u32x4_t{l2[0], l2[1], h2[0], h2[1]};
  };
  auto compute = [&](const char* base) {
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      u32x4_t bf[4], sf[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        if constexpr (IMGAL) {
          bf[i] = frag_at(base, bfo[kk][i]);
          sf[i] = frag_at(base, sfo[kk][i]);
        } else {
          bf[i] = timg_frag(base, wm * 64 + i * 16, kk, lane);
          sf[i] = timg_frag(base + IMG, wn * 64 + i * 16, kk, lane);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = mfma16<T>(sf[j], bf[i], acc[i][j]);
    }
  };

  GCT2_CLOCK_BEGIN;
  {
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

#ifdef GCT2_STAMP
  GCT2_CLOCK_END(p.clock ? p.stamps : nullptr, 4, wave, lane);
#endif
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
#ifdef GCT2_STAMP
  GCT2_CLOCK_EXIT(p.clock ? p.stamps : nullptr, 4, wave, lane);
#endif
}

#define GCT2_VMCNT_ONLY(n) ((((n) & 0xF) | 0x70 | 0xF00 | ((((n) >> 4) & 3) << 14)))
// the DMA as inline asm: invisible to hipcc's own vmcnt bookkeeping (which would drain it at the loop head); every wait is written out
__device__ __forceinline__ void dma16_hidden(__amdgpu_buffer_rsrc_t rsrc, char* lds_piece, unsigned voff) {
  const unsigned lds_addr = (unsigned)(uintptr_t)(lds_void_t*)lds_piece;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc)
               : "memory", "m0");
}

// ---- the same pipeline with a lean stage (r03) ---------------------------------------------------------------------------------
// wgrad256p_kernel's stage carries ~80 vector instructions besides its 32 MFMAs and 24 transposed reads (the ISA shows exec-masked
// branches around three 64-bit multiply-adds per gathered piece, ~40 instructions that rebuild the 12 fragment addresses, a chain of
// scalar branches for the wait count): with two lock-stepped waves per SIMD that is more than the issue slots the MFMAs leave free.
// Here: (a) the gather addresses advance incrementally (adds and selects, no multiply, no branch: one stage = 32 rows further, with
// carries into the next image row / image), tap validity from four precomputed per-lane flags; (b) the fragment addresses are
// lane offsets computed ONCE, plus the stage's compile-time base; (c) ONE LDS array (the DMA is hidden inline asm, so hipcc has
// nothing to drain); (d) a constant vmcnt in the steady state.  Same arithmetic in the same order: bit-identical results.
//
// r04, from the per-stage stamps of the diagnostic build (make stamp EXTRA=-DGCT2_PHASES, profiles/r04_wgrad_stage_phases.txt): a
// stage of the r03 order took ~1750 cycles for 1024 cycles of multiplies per SIMD - ~450 cycles of address code + DMA issue at the
// start in which NO wave of the CU multiplies (all 8 waves are in that phase together), then the older wave of every SIMD (waves
// 0-3: they win the arbitration for the matrix core) is through its multiplies 300 cycles before its SIMD-mate and idles ~600
// cycles at the barrier.  The memory system is not the limit (vmcnt wait: 8 cycles; every source byte from a 1-MiB window: -4 %),
// nor is where the DMA pieces sit between the multiplies (three placements: +-1 %).  Two changes:
//   TURNS    the two waves of a SIMD take turns.  Waves 0-3 read their 12 fragments, multiply, and issue their DMA pieces BEHIND their
//            multiplies; waves 4-7 issue first and multiply second - one wave's address code + DMA issue runs under its SIMD-mate's
//            multiplies.  All 24 fragment reads of a stage are issued up front (48 registers) and the 32 multiplies follow back to
//            back, so the wave that has the matrix core to itself never waits for LDS between rows.
//   ALIGNED  a 32-row stage that never leaves one image and starts at a multiple of its width or spans whole rows (Ws % 32 == 0, or
//            32 % Ws == 0 and Hs % (32 / Ws) == 0: every layer of the reference topology that takes this kernel) sits at the same
//            place for every lane: the position counters, the stage's byte offsets and its border bits are scalar, a lane adds a
//            constant offset and tests constant flags - 8 vector instructions per stage instead of ~45.
// Same sources, same zero fill, same multiplies in the same order: bit-identical to the r03 order (which other shapes keep).
//
// r05, PIPE: the fragments of stage s+1 are read UNDER the multiplies of stage s (behind every row of four multiplies the reads that
// refill that row's big-operand registers and one small-operand fragment), so a wave's turn is "multiply" / "issue DMA" without a read
// phase in front.  What it takes: stage s+1 must have landed, and every wave must know it, one barrier EARLIER than before (two stages
// stay in flight across a barrier instead of three; the ring keeps five buffers: s+1 is read, s+2 landed, s+3 / s+4 in flight, the
// buffer of s-1 is the DMA target); 16 more registers (the next stage's four small-operand fragments; a big-operand fragment's
// registers are free again as soon as its row is issued: 234 registers, no spill).  Same multiplies in the same order: bit-identical
// to the other orders.  Measured (profiles/r05_step_ab.txt): launch by launch within +-2 % of the r04 order (as were three more
// placements of the reads and the DMA pieces: one read behind every multiply, every wave issuing in front, one piece behind every
// second row) - a wave's multiplies take 820 cycles with the reads between them instead of 577 - but 17-30 us per STEP faster in
// interleaved in-process rounds on two boxes, which is why it is the default (tuning 16-23 = 5 keeps the r04 order).
template <typename T, bool TURNS, bool ALIGNED, bool PIPE = false>
__global__ __launch_bounds__(512, 2) void wgrad256q_kernel(WgradParams p) {
  static_assert(!PIPE || (TURNS && ALIGNED), "the pipelined order exists for the aligned, turn-taking stage only");
  constexpr int NST = 5;
  constexpr int IMG = 32 * 256;                                   // one T image of a stage: 32 r-rows x 128 columns
  constexpr int STAGE = 4 * IMG;
  constexpr int NDMA = 4;
  __shared__ __attribute__((aligned(16))) char lds[NST * STAGE];

  GCT2_CLOCK_DECL;
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

  // ---- ALIGNED: lane constants, scalar stage position
  const int span_w = min(32, Ws), span_h = max(1, 32 / Ws);
  const int l_rw = row0 % span_w, l_dh = row0 / span_w;
  unsigned cbig[2], csml[2], eflag[2];
  unsigned long long ms_never[2];
#pragma unroll
  for (int g = 0; g < 2; g++) {
    cbig[g] = (unsigned)((2 * l_dh * Wb + 2 * l_rw) * ldb2 + dg[g]);
    csml[g] = (unsigned)(row0 * lds2b + (cs0 + lc * 8) * 2 + g * 256);
    // bit 0: top tap on the stage's first image row, 1: bottom tap on its last, 2: left tap in its first column, 3: right tap in its
    // last, 4: tap / channel block out of range (always masked), 5: set in every lane (a stage beyond the last row masks all)
    eflag[g] = ((edge[g] & 1u) && l_dh == 0 ? 1u : 0u) | ((edge[g] & 2u) && l_dh == span_h - 1 ? 2u : 0u) | ((edge[g] & 4u) && l_rw == 0 ? 4u : 0u) |
               ((edge[g] & 8u) && l_rw == span_w - 1 ? 8u : 0u) | (edge[g] & 16u) | 32u;
    ms_never[g] = __builtin_amdgcn_ballot_w64(s_bad[g] != 0);
  }
  int u_rs = st_lo * 32;                                           // first row of the stage to issue next and where it sits: all uniform
  int u_w = u_rs % Ws, u_h, u_b;
  { const int t = u_rs / Ws; u_h = t % Hs; u_b = t / Hs; }
  unsigned u_pix = (unsigned)(((u_b * Hb + 2 * u_h) * Wb + 2 * u_w) * ldb2), u_soff = (unsigned)(u_rs * lds2b);
  auto issue_aligned = [&](char* base) {
    const bool beyond = u_rs >= R;
    const unsigned s_pos = (u_h == 0 ? 1u : 0u) | (u_h + span_h == Hs ? 2u : 0u) | (u_w == 0 ? 4u : 0u) | (u_w + span_w == Ws ? 8u : 0u) | 16u | (beyond ? 32u : 0u);
    const unsigned long long all = beyond ? ~0ull : 0ull;
    char* piece = base + wave * 1024;
#pragma unroll
    for (int g = 0; g < 2; g++) {
      const bool badb = (eflag[g] & s_pos) != 0u;
      const bool bads = __builtin_amdgcn_inverse_ballot_w64(all | ms_never[g]);
      dma16_hidden(rs_b, piece + g * IMG, badb ? OOB : u_pix + cbig[g]);
      dma16_hidden(rs_s, piece + (2 + g) * IMG, bads ? OOB : u_soff + csml[g]);
    }
    u_rs += 32; u_soff += (unsigned)(32 * lds2b);
    u_w += adv_w; u_pix += (unsigned)pixA;
    if (u_w >= Ws) { u_w -= Ws; u_h++; u_pix += (unsigned)pixB; }
    u_h += adv_h; u_b += adv_b; u_pix += (unsigned)pixCD;
    if (u_h >= Hs) { u_h -= Hs; u_b++; }
  };
  auto issue_x = [&](char* base) {
    if constexpr (ALIGNED) issue_aligned(base);
    else issue(base);
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
    if (more) issue_x(tgt);
    if (live) compute(cur);
    if (more) __builtin_amdgcn_s_waitcnt(GCT2_VMCNT_ONLY(3 * NDMA));   // steady state: stages st+2 .. st+4 stay in flight
    else wait_tail(st_hi - 1 - (st + 1));
    __builtin_amdgcn_s_barrier();
  };
  // steady state: every stage of the trip still has a stage to issue (st + 4 + 4 < st_hi): no tail logic, one constant wait
#if defined(GCT2_STAMP) && defined(GCT2_PHASES)
  // cycles per steady-state stage by phase (every stamp waits for its own value - the compiler may copy an asm output at once; in
  // the TURNS order no LDS read is outstanding at any stamp but [1]'s end, which waits for the reads anyway):
  // [0] issue block in front of the multiplies, [1] fragment reads until their data is there, [2] the 32 multiplies issued,
  // [3] issue block behind the multiplies, [4] vmcnt wait, [5] barrier, [7] stages
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, nstages = 0;
  auto now = [&]() -> unsigned long long {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
  };
#define GCT2_PH(x) x
#else
#define GCT2_PH(x)
#endif
  GCT2_CLOCK_BEGIN;
  issue_x(lds);
  if (st_lo + 1 < st_hi) issue_x(lds + STAGE);
  if (st_lo + 2 < st_hi) issue_x(lds + 2 * STAGE);
  if (st_lo + 3 < st_hi) issue_x(lds + 3 * STAGE);
  if constexpr (PIPE) {
    // stages st_lo and st_lo + 1 landed (stages come in pairs: a split has at least two), the next two may stay in flight
    if (st_lo + 3 < st_hi) __builtin_amdgcn_s_waitcnt(GCT2_VMCNT_ONLY(2 * NDMA));
    else if (st_lo + 2 < st_hi) __builtin_amdgcn_s_waitcnt(GCT2_VMCNT_ONLY(NDMA));
    else __builtin_amdgcn_s_waitcnt(GCT2_VMCNT_ONLY(0));
    __builtin_amdgcn_s_barrier();
    u32x4_t sf[4], bf[8];
#pragma unroll
    for (int j = 0; j < 4; j++) sf[j] = frag(lds, sf_off[j]);
#pragma unroll
    for (int i = 0; i < 8; i++) bf[i] = frag(lds, bf_off[i]);
    const bool front = wm != 0;
    // multiply the stage in registers; behind row i's four multiplies read row i of the next stage (and one small-operand fragment)
    auto rows = [&](const char* nxt, bool has_next) {
      u32x4_t sn[4];
#pragma unroll
      for (int i = 0; i < 8; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = mfma16<T>(sf[j], bf[i], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);                         // the reads stay behind the row whose registers they reuse
        if (has_next) {
          bf[i] = frag(nxt, bf_off[i]);
          if (i < 4) sn[i] = frag(nxt, sf_off[i]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (has_next) {
#pragma unroll
        for (int j = 0; j < 4; j++) sf[j] = sn[j];
      }
    };
    auto pstage_fast = [&](const char* nxt, char* tgt) {          // steady state: stages s+1 .. s+4 exist
      GCT2_PH(const unsigned long long t0 = now();)
      if (front) issue_aligned(tgt);
      GCT2_PH(const unsigned long long t1 = now();)
      __builtin_amdgcn_sched_barrier(0);
      if (live) rows(nxt, true);
      __builtin_amdgcn_sched_barrier(0);
      GCT2_PH(const unsigned long long t3 = now();)               // (waits for the next stage's fragment reads as well)
      if (!front) issue_aligned(tgt);
      GCT2_PH(const unsigned long long t4 = now();)
      __builtin_amdgcn_s_waitcnt(GCT2_VMCNT_ONLY(2 * NDMA));      // s+2 landed; s+3, s+4 stay in flight
      GCT2_PH(const unsigned long long t5 = now();)
      __builtin_amdgcn_s_barrier();
      GCT2_PH(const unsigned long long t6 = now();
              ph[0] += t1 - t0; ph[2] += t3 - t1; ph[3] += t4 - t3; ph[4] += t5 - t4; ph[5] += t6 - t5; nstages++;)
    };
    auto pstage = [&](int s, const char* nxt, char* tgt) {
      const bool more = s + NST - 1 < st_hi, has_next = s + 1 < st_hi;   // block-uniform
      if (more && front) issue_aligned(tgt);
      __builtin_amdgcn_sched_barrier(0);
      if (live) rows(nxt, has_next);
      __builtin_amdgcn_sched_barrier(0);
      if (more && !front) issue_aligned(tgt);
      const int ahead = st_hi - 1 - (s + 2);                       // stages behind s+2 that this wave has issued
      if (ahead >= 2) __builtin_amdgcn_s_waitcnt(GCT2_VMCNT_ONLY(2 * NDMA));
      else if (ahead == 1) __builtin_amdgcn_s_waitcnt(GCT2_VMCNT_ONLY(NDMA));
      else __builtin_amdgcn_s_waitcnt(GCT2_VMCNT_ONLY(0));
      __builtin_amdgcn_s_barrier();
    };
    int s = st_lo;
    for (; s + 8 < st_hi; s += 5) {
      pstage_fast(lds + STAGE, lds + 4 * STAGE);
      pstage_fast(lds + 2 * STAGE, lds);
      pstage_fast(lds + 3 * STAGE, lds + STAGE);
      pstage_fast(lds + 4 * STAGE, lds + 2 * STAGE);
      pstage_fast(lds, lds + 3 * STAGE);
    }
    for (; s < st_hi; s += 5) {
      pstage(s, lds + STAGE, lds + 4 * STAGE);
      if (s + 1 >= st_hi) break;
      pstage(s + 1, lds + 2 * STAGE, lds);
      if (s + 2 >= st_hi) break;
      pstage(s + 2, lds + 3 * STAGE, lds + STAGE);
      if (s + 3 >= st_hi) break;
      pstage(s + 3, lds + 4 * STAGE, lds + 2 * STAGE);
      if (s + 4 >= st_hi) break;
      pstage(s + 4, lds, lds + 3 * STAGE);
    }
  } else {
  wait_tail(min(st_lo + 3, st_hi - 1) - st_lo);
  __builtin_amdgcn_s_barrier();
  auto stage_fast = [&](const char* cur, char* tgt) {
    const bool front = TURNS ? wm != 0 : true;                     // issue block in front of the multiplies (else behind them)
    GCT2_PH(const unsigned long long t0 = now(); unsigned long long t2 = 0;)
    if (front) issue_x(tgt);
    GCT2_PH(const unsigned long long t1 = now();)
    __builtin_amdgcn_sched_barrier(0);
    if (live) {
      if constexpr (TURNS) {
        u32x4_t sf[4], bf[8];
#pragma unroll
        for (int j = 0; j < 4; j++) sf[j] = frag(cur, sf_off[j]);
#pragma unroll
        for (int i = 0; i < 8; i++) bf[i] = frag(cur, bf_off[i]);
        __builtin_amdgcn_sched_barrier(0);                         // all 24 reads in flight before the first multiply
        GCT2_PH(t2 = now();)
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[i][j] = mfma16<T>(sf[j], bf[i], acc[i][j]);
      } else {
        GCT2_PH(t2 = t1;)
        compute(cur);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    GCT2_PH(const unsigned long long t3 = now();)
    if (!front) issue_x(tgt);
    GCT2_PH(const unsigned long long t4 = now();)
    __builtin_amdgcn_s_waitcnt(GCT2_VMCNT_ONLY(3 * NDMA));
    GCT2_PH(const unsigned long long t5 = now();)
    __builtin_amdgcn_s_barrier();
    GCT2_PH(const unsigned long long t6 = now();
            ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3; ph[4] += t5 - t4; ph[5] += t6 - t5; nstages++;)
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
  }
#ifdef GCT2_STAMP
  GCT2_CLOCK_END(p.clock ? p.stamps : nullptr, 8, wave, lane);
#ifdef GCT2_PHASES
  if (p.clock && lane == 0 && (size_t)blockIdx.x * 8 + wave < ((size_t)1 << 15)) {       // phase region of the stamp buffer (scripts/stamp_clock.py)
    unsigned long long* o = p.stamps + ((size_t)1 << 18) + ((size_t)blockIdx.x * 8 + wave) * 8;
    for (int k = 0; k < 6; k++) o[k] = ph[k];
    o[6] = 0; o[7] = nstages;
  }
#endif
#endif
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
#ifdef GCT2_STAMP
  GCT2_CLOCK_EXIT(p.clock ? p.stamps : nullptr, 8, wave, lane);
#endif
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

int wgrad_mfma(gct2_ctx& c, int dtype, WgradParams p, hipStream_t s, WgradSlabs* defer) {
  if (defer) *defer = WgradSlabs{nullptr, 0, 0};
  const int variant = c.wgrad_variant;
  const int R = p.B * p.Hs * p.Ws;
  const int taps = p.ks ? p.ks * p.ks : 16;          // p.ks != 0: stride-1 'same' convolution (128 x 128 tile only)
  const int steps_total = (R + 63) / 64;
  // Tile choice.  The 256 x 256 pipeline (one work-group per CU, 128 FLOP per staged byte) takes every layer whose 128 x 128 tiling
  // would have to split the pixel range anyway (fewer than 512 small tiles) and that still gives >= 192 work-groups with >= 8 (below
  // 256 small tiles: >= 4) 64-row steps per pixel split.  Measured per layer and in the step: profiles/r01_wgrad_variants.txt,
  // r03_wgrad_lean_stage.txt, r03_step_ab.txt (U0/U1/U2/D1/D2/D3 take it at config 3; the 2x2 / 4x4 / 8x8 levels have too few pixels
  // and keep one-owner 128 x 128 tiles; going the other way - the small tile for D1/D2/U1 - is faster launch by launch and 38 us
  // slower in the step: its half-CU work-groups interleave with the input-gradient chain's instead of alternating with them).
  // A big-tile launch always leaves ~256 work-groups x 256 KiB = 64 MiB of slabs, whatever the size of the tensor.
  const int tiles128 = ((taps * p.Cb + 127) / 128) * ((p.Cs + 127) / 128);
  const int tiles256 = ((taps * p.Cb + 255) / 256) * ((p.Cs + 255) / 256);
  const int minsteps = tiles128 < 256 ? 4 : 8;
  const int blocks256 = tiles256 * std::max(1, std::min((256 + tiles256 - 1) / tiles256, steps_total / minsteps));
  const bool big_tile = !p.ks && (variant == 2 || variant == 4 || variant == 5 || (variant == 0 && tiles128 < 512 && blocks256 >= 192));
  const int tiles = big_tile ? tiles256 : tiles128;
  // pixel splits: ~256 work-groups for the big tile; ~768 (3 per CU) for the small one, one owner per tile once the tiles alone
  // give every CU two work-groups, two splits in between; always >= 4 steps of 64 rows per split
  int rsplit = big_tile ? (256 + tiles - 1) / tiles : (tiles >= 512 ? 1 : (768 + tiles - 1) / tiles);
  if (!big_tile && tiles >= 256 && tiles < 512) rsplit = steps_total >= 32 ? 2 : 1;
  if (!big_tile && c.wgrad_split) rsplit = 1 << (c.wgrad_split - 1);
  rsplit = max(1, min(rsplit, steps_total / 4));
  const int per = (steps_total + rsplit - 1) / rsplit;
  rsplit = (steps_total + per - 1) / per;            // every split non-empty (each one owns a slab)
  p.rsplit = rsplit;
  p.ws = nullptr;
#ifdef GCT2_STAMP
  p.stamps = c.stamps;
  p.clock = c.stamps_bytes >= GCT2_CLOCK_BYTES ? 1 : 0;
#endif
  const size_t n = (size_t)taps * p.Cb * p.Cs;
  size_t ws_bytes = 0;
  float* ws = c.wgrad_scratch(&ws_bytes);
  // split launches leave ordered slabs whenever the workspace holds them (atomics would make the gradient depend on the arrival
  // order; variant 7 keeps them for the comparison test)
  if (rsplit > 1 && rsplit <= 128 && ws && n % 4 == 0 && (uintptr_t)p.dw % 16 == 0 && n * sizeof(float) * rsplit <= ws_bytes && variant != 7)
    p.ws = ws;
  dim3 grid(rsplit >= 8 ? tiles * 8 * ((rsplit + 7) / 8) : tiles * rsplit);
  if (!p.ws && rsplit > 1 && !p.accumulate) {     // atomics add into the target: start it from zero
    (void)hipMemsetAsync(p.dw, 0, n * sizeof(float), s);
  }
  gct2_log(c, "wgrad:%s:rsplit=%d:%s", p.ks ? "s1" : (big_tile ? (variant == 4 ? "256q-r03" : (variant == 5 ? "256q-r04" : "256q")) : "128"), rsplit, p.ws ? "slabs" : (rsplit == 1 ? "owner" : "atomics"));
  if (p.ks) {
    if (dtype == GCT2_BF16) hipLaunchKernelGGL((wgrad_kernel<__bf16, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((wgrad_kernel<_Float16, true>), grid, dim3(256), 0, s, p);
  } else if (big_tile) {
    // stage-aligned geometry -> scalar address code and waves taking turns (see the kernel); other shapes keep the r03 stage order
    const bool aligned = (p.Ws % 32 == 0 || (32 % p.Ws == 0 && p.Hs % (32 / p.Ws) == 0)) && variant != 4;
    const bool pipe = aligned && variant != 5;       // r05: the next stage's fragments read under the multiplies (5 = the r04 order)
    if (dtype == GCT2_BF16) {
      if (pipe) hipLaunchKernelGGL((wgrad256q_kernel<__bf16, true, true, true>), grid, dim3(512), 0, s, p);
      else if (aligned) hipLaunchKernelGGL((wgrad256q_kernel<__bf16, true, true>), grid, dim3(512), 0, s, p);
      else hipLaunchKernelGGL((wgrad256q_kernel<__bf16, false, false>), grid, dim3(512), 0, s, p);
    } else {
      if (pipe) hipLaunchKernelGGL((wgrad256q_kernel<_Float16, true, true, true>), grid, dim3(512), 0, s, p);
      else if (aligned) hipLaunchKernelGGL((wgrad256q_kernel<_Float16, true, true>), grid, dim3(512), 0, s, p);
      else hipLaunchKernelGGL((wgrad256q_kernel<_Float16, false, false>), grid, dim3(512), 0, s, p);
    }
  } else {
    // whole images of the small grid per 64-row step (tuning 16-23 = 6 keeps the general form for the bit-identity test)
    const bool imgal = 64 % (p.Hs * p.Ws) == 0 && variant != 6;
    if (dtype == GCT2_BF16) {
      if (imgal) hipLaunchKernelGGL((wgrad_kernel<__bf16, false, true>), grid, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((wgrad_kernel<__bf16>), grid, dim3(256), 0, s, p);
    } else {
      if (imgal) hipLaunchKernelGGL((wgrad_kernel<_Float16, false, true>), grid, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((wgrad_kernel<_Float16>), grid, dim3(256), 0, s, p);
    }
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
