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
// channels of one tap per step.  Staging is LDS-DMA (buffer_load_dwordx4 ... lds): every wave-instruction
// deposits 1 KiB lane-linearly, so the XOR swizzle of the LDS images is applied to each lane's SOURCE address;
// zero padding (image border, ragged M/K/N) is an out-of-range buffer offset, which the hardware turns into
// zeros in LDS (probed: tests/hw_probe/probe_glds.hip).  Tile / buffering variants: see dispatch() below (default: one or
// two LDS buffers with the DMA of step t+1 issued before the MFMAs of step t, 2 work-groups per CU covering each other).
// MFMA orientation: A operand = weights (rows = n), B operand = activations (cols = m), so every lane ends
// with 4 consecutive output channels of one pixel -> 8-byte NHWC stores.
#include "gct2_common.h"
#include <algorithm>


namespace {

constexpr int BK = 64;
constexpr unsigned OOB = 0x80000000u;            // >= num_records of every descriptor below
// s_waitcnt immediate that waits for vmcnt <= n only (expcnt / lgkmcnt fields at their no-wait maxima), gfx9 encoding
#define VMCNT_ONLY(n) ((((n) & 0xF) | 0x70 | 0xF00 | ((((n) >> 4) & 3) << 14)))
typedef __attribute__((address_space(3))) void lds_void_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)OOB, 0x00020000);
}
// M0 = LDS address of the piece is written by the builtin in the same statement that uses it (cdna_hip_programming.md §5.7)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_piece, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)lds_piece, 16, (int)voff, 0, 0, 0);
}
// the same with a wave-uniform byte offset in an SGPR (tap / k-chunk part of the address; the bounds check sees the per-lane offset)
__device__ __forceinline__ void dma16s(__amdgpu_buffer_rsrc_t rsrc, char* lds_piece, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)lds_piece, 16, (int)voff, (int)soff, 0, 0);
}

// bias gradient = column sums of the masked gradient a dgrad launch produces: db[n] (+)= sum over pixels.  Channels
// [0, db_split) go to p.db, the rest to p.db2 (the output of an UpShuffle dgrad spans two layers' pre-activations).
__device__ __forceinline__ float* db_target(const TapGemmParams& p, int n) {
  return n < p.db_split ? (p.db ? p.db + n : nullptr) : (p.db2 ? p.db2 + (n - p.db_split) : nullptr);
}
// whether the target of channel n is added to (db_accumulate bits of include/gct2.h) or overwritten
__device__ __forceinline__ bool db_adds(const TapGemmParams& p, int n) { return (p.db_acc >> (n < p.db_split ? 0 : 1)) & 1; }

// NBUF = 2: two LDS buffers, the DMA of step t+1 is issued before the MFMAs of step t; NBUF = 1: issue, wait, multiply.  Either way
// vmcnt(0) + barrier per step, and 2 (or 4) independent work-groups per CU cover each other's waits - the arrangement that measured
// best on this chip (the three-buffer, 256 x 256 and five-stage-ring tiles of r01-r03 were 12-35 % slower and are gone).
// Every wave owns a 64 (m) x 64 (n) sub-tile.
#ifdef GCT2_STAMP
// diagnostic build (make stamp, scripts/stamp_clock.py): s_memrealtime at the phase boundaries of one wave per work-group,
// written to the buffer handed over with gct2_ctx_set_stamp_buffer (never part of the product build: gct2_build_flags())
__device__ __forceinline__ unsigned long long tg_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define TG_STAMP(k) st[k] = tg_stamp()
#else
#define TG_STAMP(k)
#endif
template <typename T, int FORM, int BM, int BN, int EPI, int NBUF>
__global__ __launch_bounds__((BM / 64) * (BN / 64) * 64, NBUF == 1 ? 4 : 2) void tapgemm_kernel(TapGemmParams p) {
  constexpr int WM = 64;
  constexpr int NWV = (BM / WM) * (BN / 64);       // waves, each a 64 x 64 sub-tile
  constexpr int MF = WM / 16;                      // 16-pixel fragments per wave
  static_assert(NBUF == 1 || NBUF == 2, "one or two LDS buffers");
  static_assert(NWV == 4 || NWV == 8, "4 or 8 waves");
  constexpr bool S1 = (FORM == FORM_S1 || FORM == FORM_S1T);   // stride-1 'same' convolution: p.ks x p.ks taps on the output's own grid
  constexpr bool WT = (FORM == FORM_CONV || FORM == FORM_S1);   // weights [tap][k][n] (T image); otherwise [tap][n][k] (N image)
  static_assert(!WT || BN % 128 == 0, "T images are 128 columns wide");
  constexpr int WAVES_N = BN / 64;
  constexpr int BKS = BK;                          // reduction elements per step
  constexpr int NA = BM / 8 / NWV;                 // 1-KiB pieces per wave, activation tile (8 rows of 128 B)
  constexpr int NW = (WT ? 16 * (BN / 128) : BN / 8) / NWV;
  constexpr int A_BYTES = BM * 128;
  constexpr int W_BYTES = WT ? 64 * 256 * (BN / 128) : BN * 128;   // T images: BN/128 of them side by side
  constexpr int NTAPS = (FORM == FORM_CONV) ? 16 : 4;              // (the stride-1 forms: p.ks * p.ks, run-time)

  // DISTINCT LDS objects: lets hipcc prove that the DMA into one buffer does not alias the ds_reads of another,
  // so it does not drain vmcnt before every read (cdna_hip_programming.md, "Three .s-level traps" (a))
  __shared__ __attribute__((aligned(16))) char lds0[A_BYTES + W_BYTES];
  __shared__ __attribute__((aligned(16))) char lds1[NBUF >= 2 ? A_BYTES + W_BYTES : 16];

#ifdef GCT2_STAMP
  unsigned long long st[5];
  unsigned long long* stamp_out = p.stamps;
  TG_STAMP(0);
#endif
  GCT2_CLOCK_DECL;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave % WAVES_N, wm = wave / WAVES_N;
  const int Hs = p.Hs, Ws = p.Ws, K = p.K, N = p.N;
  const int M = p.B * Hs * Ws;
  // 1-D grid: [k-slice][XCD-aware (m-tile, n-tile, phase)]
  constexpr int PH = (FORM == FORM_CONVT) ? 4 : 1;
  const int inner = p.n_tiles * PH, per_slice = 8 * p.xcd_chunk * inner;
  int kslice, m_tile, in;
  if (p.wstat) {
    // weight-stationary order (layers whose weights outweigh their activations: the U-Net's bottleneck): a weight slice =
    // (k-slice, n-tile, phase); the slices are dealt over the XCDs (ids with equal id % 8 share an L2) and an XCD runs ALL
    // m-tiles of a slice back to back, so every XCD streams 1/8 of the weight tensor instead of all of it (measured before: 8-10x
    // the algorithmic HBM traffic on these layers, profiles/r02_traffic_per_layer_before_wstat.json)
    const int xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
    const int slice = (j / p.m_tiles) * 8 + xcd;
    if (slice >= inner * p.ksplit) return;
    m_tile = j % p.m_tiles;
    kslice = slice / inner;
    in = slice - kslice * inner;
  } else {
    kslice = (int)blockIdx.x / per_slice;
    if (!xcd_tile((int)blockIdx.x - kslice * per_slice, p.m_tiles, inner, p.xcd_chunk, m_tile, in)) return;
  }
  const int phase = in % PH, n_tile = in / PH;
  const int m0 = m_tile * BM, n0 = n_tile * BN;
  const int ph = phase >> 1, pw = phase & 1;
  const int Hsrc = (FORM == FORM_CONV) ? 2 * Hs : Hs, Wsrc = (FORM == FORM_CONV) ? 2 * Ws : Ws;
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x), rs_w = make_rsrc(p.w);
  const int ldx2 = p.ldx * 2;                      // bytes per source pixel

  // ---- per-lane DMA descriptors (fixed over the whole K loop) ------------------------------------------
  // activation tile (N image): piece q = wave + 4 i holds rows 8q .. 8q+7; lane -> row 8q + (lane>>3),
  // physical chunk lane&7 = logical chunk ^ ((row>>1)&7)
  const int a_lchunk = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);   // 4*NWV*i is a multiple of 8
  unsigned a_off[NA];                              // byte offset of (row's tap-origin pixel, logical chunk)
  unsigned a_mask[NA];                             // bit t: tap t reads inside the image for this row
#pragma unroll
  for (int i = 0; i < NA; i++) {
    const int m = m0 + 8 * (wave + NWV * i) + (lane >> 3);
    a_off[i] = 0; a_mask[i] = 0;
    if (m < M) {
      int sw, sh, b;
      decode_pixel(m, Hs, Ws, p.hs_shift, p.ws_shift, sw, sh, b);
      // origin = the tap with the smallest row/column: (2sh-1, 2sw-1) for conv, (sh+ph-1, sw+pw-1) for convT
      const int h0 = (FORM == FORM_CONV) ? 2 * sh - 1 : (S1 ? sh - (p.ks - 1) / 2 : sh + ph - 1);
      const int w0 = (FORM == FORM_CONV) ? 2 * sw - 1 : (S1 ? sw - (p.ks - 1) / 2 : sw + pw - 1);
      a_off[i] = (unsigned)(((b * Hsrc + h0) * Wsrc + w0) * ldx2 + a_lchunk * 16);   // may wrap below 0: only used when valid
      if constexpr (S1) {
        for (int t2 = 0; t2 < p.ks * p.ks; t2++) {
          const int dh = t2 / p.ks, dw = t2 - dh * p.ks;
          if ((unsigned)(h0 + dh) < (unsigned)Hsrc && (unsigned)(w0 + dw) < (unsigned)Wsrc) a_mask[i] |= 1u << t2;
        }
      } else {
        // validity of tap (dh, dw) = (row dh inside) & (column dw inside): one TW-bit column mask, OR-ed in per valid row
        constexpr int TW = (FORM == FORM_CONV) ? 4 : 2;
        unsigned cm = 0;
#pragma unroll
        for (int dw = 0; dw < TW; dw++) cm |= ((unsigned)(w0 + dw) < (unsigned)Wsrc ? 1u : 0u) << dw;
#pragma unroll
        for (int dh = 0; dh < TW; dh++)
          if ((unsigned)(h0 + dh) < (unsigned)Hsrc) a_mask[i] |= cm << (dh * TW);
      }
    }
  }
  // weight tile
  unsigned w_off[NW];
  bool w_nok[NW];                                  // column (n) part of the validity
  int w_k[NW];                                     // FORM_CONV: k row inside the 64-step; FORM_CONVT: unused
#pragma unroll
  for (int i = 0; i < NW; i++) {
    if (WT) {                                      // T image: piece = 4 k-rows x 16 chunks
      const int q = wave + NWV * i;                // pieces 0..15 fill image 0 (columns n0..n0+127), 16..31 image 1
      const int k = 4 * (q & 15) + (lane >> 4);
      const int lc = ((((lane & 15) >> 1) ^ timg_swz(k)) << 1) | (lane & 1);
      const int nn = n0 + (q >> 4) * 128 + lc * 8;
      w_k[i] = k;
      w_nok[i] = nn < N;
      w_off[i] = (unsigned)((k * N + nn) * 2);
    } else {                                       // N image: piece = 8 n-rows x 8 chunks
      const int n = 8 * (wave + NWV * i) + (lane >> 3);
      w_k[i] = a_lchunk * 8;                       // first k of this lane's chunk
      w_nok[i] = (n0 + n) < N;
      w_off[i] = (unsigned)(((n0 + n) * K + a_lchunk * 8) * 2);
    }
  }

  const int nk = (K + BKS - 1) / BKS;
  const int niter = (S1 ? p.ks * p.ks : NTAPS) * nk;

  // ---- lean issue (r03; the 4x4 / stride-2 forms): the r02 code below spent ~45 vector and ~60 scalar instructions per step on
  // its 8 pieces (a scalar division for (tap, chunk), compare / select chains per piece).  Here: (tap, chunk) are counters (the steps
  // are issued in order), the source descriptor is based ONE row + ONE pixel in front of the tensor so that the per-lane origin
  // offsets are non-negative and the tap / chunk part of the address can go into the instruction's scalar offset, tap validity is
  // a shift of the inverted mask into bit 31 of the per-lane offset (out of range -> zeros), the weight offsets are loop-invariant.
  const unsigned shift_x = (unsigned)((Wsrc + 1) * ldx2);
  const __amdgpu_buffer_rsrc_t rs_xs = make_rsrc(reinterpret_cast<const char*>(p.x) - (S1 ? 0 : shift_x));
  unsigned a_voff[NA], a_nmask[NA], w_voff[NW];
#pragma unroll
  for (int i = 0; i < NA; i++) { a_voff[i] = a_mask[i] ? a_off[i] + shift_x : 0u; a_nmask[i] = ~a_mask[i]; }
#pragma unroll
  for (int i = 0; i < NW; i++) w_voff[i] = w_nok[i] ? w_off[i] : OOB;
  int c_tap = 0, c_kc = 0;                          // (tap, k-chunk) of the next step to issue: set below, once it_lo is known
  auto issue = [&](int it, char* abase) {
    if constexpr (!S1) {
      const int tap = c_tap, c0 = c_kc * BKS;
      int tap16, dh, dw;
      if (FORM == FORM_CONV) { tap16 = tap; dh = tap >> 2; dw = tap & 3; }
      else {
        const int a = tap >> 1, c = tap & 1;
        dh = 1 - a; dw = 1 - c;
        tap16 = (1 - ph + 2 * a) * 4 + (1 - pw + 2 * c);
      }
      const int abit = (FORM == FORM_CONVT) ? dh * 2 + dw : tap;
      const unsigned s_a = (unsigned)((dh * Wsrc + dw) * ldx2 + c0 * 2);
      const unsigned s_w = WT ? (unsigned)(((tap16 * K + c0) * N) * 2) : (unsigned)((tap16 * N * K + c0) * 2);
      char* wbase = abase + A_BYTES;
      if (c0 + BKS <= K) {                          // block-uniform: a full chunk needs no per-lane channel check
#pragma unroll
        for (int i = 0; i < NA; i++)
          dma16s(rs_xs, abase + (wave + NWV * i) * 1024, a_voff[i] | ((a_nmask[i] >> abit) << 31), s_a);
#pragma unroll
        for (int i = 0; i < NW; i++)
          dma16s(rs_w, wbase + (wave + NWV * i) * 1024, w_voff[i], s_w);
      } else {                                      // the ragged last chunk (K % 64 != 0)
        asm volatile("" ::: "memory");              // keeps hipcc from turning this branch into selects in the full-chunk path
#pragma unroll
        for (int i = 0; i < NA; i++) {
          const unsigned v = a_voff[i] | ((a_nmask[i] >> abit) << 31);
          dma16s(rs_xs, abase + (wave + NWV * i) * 1024, (c0 + a_lchunk * 8) < K ? v : OOB, s_a);
        }
#pragma unroll
        for (int i = 0; i < NW; i++)
          dma16s(rs_w, wbase + (wave + NWV * i) * 1024,
                            (c0 + w_k[i]) < K ? w_voff[i] : OOB, s_w);
      }
      c_kc++;
      if (c_kc == nk) { c_kc = 0; c_tap++; }
      return;
    }
    const int tap = it / nk, c0 = (it - tap * nk) * BKS;
    int tap16, dh, dw;
    if (FORM == FORM_CONV) { tap16 = tap; dh = tap >> 2; dw = tap & 3; }
    else if (S1) {                                 // the input gradient walks the same window with the kernel flipped
      dh = tap / p.ks; dw = tap - dh * p.ks;
      tap16 = (FORM == FORM_S1) ? tap : (p.ks - 1 - dh) * p.ks + (p.ks - 1 - dw);
    } else {
      // source row sh+ph-a  = origin + (1-a); kernel row kh = 1-ph+2a
      const int a = tap >> 1, c = tap & 1;
      dh = 1 - a; dw = 1 - c;
      tap16 = (1 - ph + 2 * a) * 4 + (1 - pw + 2 * c);
    }
    const int abit = (FORM == FORM_CONVT) ? dh * 2 + dw : tap;
    const unsigned tapoff = (unsigned)((dh * Wsrc + dw) * ldx2 + c0 * 2);
    const bool a_cok = (c0 + a_lchunk * 8) < K;
#pragma unroll
    for (int i = 0; i < NA; i++) {
      const bool ok = a_cok && ((a_mask[i] >> abit) & 1u);
      dma16(rs_x, abase + (wave + NWV * i) * 1024, ok ? a_off[i] + tapoff : OOB);
    }
    char* wbase = abase + A_BYTES;
    const unsigned wtap = WT ? (unsigned)(((tap16 * K + c0) * N) * 2) : (unsigned)((tap16 * N * K + c0) * 2);
#pragma unroll
    for (int i = 0; i < NW; i++) {
      const bool ok = w_nok[i] && (c0 + w_k[i]) < K;
      dma16(rs_w, wbase + (wave + NWV * i) * 1024, ok ? w_off[i] + wtap : OOB);
    }
  };

  f32x4_t acc[4][MF];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < MF; j++) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // split-K: this workgroup reduces iterations [it_lo, it_hi) only and leaves an fp32 partial slab
  const int it_per = (niter + p.ksplit - 1) / p.ksplit;
  const int it_lo = kslice * it_per, it_hi = min(niter, it_lo + it_per);
  c_tap = it_lo / nk;
  c_kc = it_lo - c_tap * nk;
  auto compute = [&](const char* a_img) {
    const char* w_img = a_img + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      u32x4_t wf[4], af[MF];
#pragma unroll
      for (int i = 0; i < 4; i++)
        wf[i] = WT ? timg_frag(w_img + (wn >> 1) * (64 * 256), (wn & 1) * 64 + i * 16, kk, lane)
                                    : nimg_frag(w_img, wn * 64 + i * 16, kk, lane);
#pragma unroll
      for (int j = 0; j < MF; j++) af[j] = nimg_frag(a_img, wm * WM + j * 16, kk, lane);
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < MF; j++) acc[i][j] = mfma16<T>(wf[i], af[j], acc[i][j]);
    }
  };
  // `live` is always true (ksplit >= 1) but opaque to hipcc: a code-generation fence.  With the multiplies unconditional the
  // unrolled steps of a trip are merged into one scheduling region and the register allocator spills (wgrad256p_kernel: 440
  // spilled registers, 10x slower; here: the 256 x 256 and three-buffer variants); behind the guard each step stays its own region.
  const bool live = p.ksplit > 0;
  TG_STAMP(1);
  GCT2_CLOCK_BEGIN;
  if constexpr (NBUF == 1) {
    // one LDS buffer (32 KiB): no overlap inside a work-group; 4 work-groups per CU cover each other instead
    for (int it = it_lo; it < it_hi; it++) {
      issue(it, lds0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (live) compute(lds0);
      __syncthreads();
    }
  } else if constexpr (NBUF == 2) {
    if (it_lo < it_hi) issue(it_lo, lds0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int it = it_lo; it < it_hi; it += 2) {    // two steps per trip: buffer roles are compile-time
      if (it + 1 < it_hi) issue(it + 1, lds1);
      if (live) compute(lds0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (it + 1 >= it_hi) break;
      if (it + 2 < it_hi) issue(it + 2, lds0);
      if (live) compute(lds1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

#ifdef GCT2_STAMP
  GCT2_CLOCK_END(p.clock ? p.stamps : nullptr, NWV, wave, lane);
#endif
  TG_STAMP(2);
  // ---- epilogue: lane holds out[m = .. + (lane&15)][n = .. + 4*(lane>>4) + r], r = 0..3 ----
  T* __restrict__ yout = reinterpret_cast<T*>(p.y);
  const T* __restrict__ actp = reinterpret_cast<const T*>(p.act);
  // an opaque copy of the lane id: keeps hipcc from hoisting the 16 tiles' output addresses above the K loop,
  // where they would occupy ~100 registers for the whole kernel (spills in the 8-wave variant)
  int elane = lane;
  asm volatile("" : "+v"(elane));
  f32x4_t bsum[4];                                  // bias-gradient partial sums: [n-fragment i][r], over this lane's pixels
#pragma unroll
  for (int i = 0; i < 4; i++) bsum[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const bool wide = p.wide && p.ksplit == 1;                 // block-uniform
  if (wide) {
    // 16-byte epilogue: v_permlane16_swap exchanges the odd 16-lane rows of fragment 2k with the even rows of fragment 2k+1, after
    // which a lane holds EIGHT consecutive channels of its pixel (rows g = 0, 2: fragment 2k, channels 4g .. 4g+7; rows g = 1, 3:
    // fragment 2k+1, channels 4(g-1) .. 4(g-1)+7): half the store / mask-load / accumulate-load instructions of the 8-byte form
    // (measured on the halo kernel, which gets the same layout from its weight image: -10..-15 %).
    const int eg = elane >> 4;
    const int nlane = wn * 64 + 16 * (eg & 1) + 4 * (eg & ~1);
    // the ReLU-mask words of pixel column j + 1 are loaded while column j is processed: one exposed load latency per tile instead
    // of one per column (in-kernel stamps, r02: the epilogue of UpShuffle_0's input gradient took 8.2 us of a
    // 41-us work-group life, most of it four serialized 16-byte-load round trips)
    auto out_pixel = [&](int j, size_t& opix) -> bool {
      const int m = m0 + wm * WM + j * 16 + (elane & 15);
      if (m >= M) return false;
      if (FORM != FORM_CONVT) opix = (size_t)m;
      else {
        int sw, sh, b;
        decode_pixel(m, Hs, Ws, p.hs_shift, p.ws_shift, sw, sh, b);
        opix = ((size_t)b * (2 * Hs) + 2 * sh + ph) * (2 * Ws) + 2 * sw + pw;
      }
      return true;
    };
    u32x4_t mk[2] = {u32x4_t{0u, 0u, 0u, 0u}, u32x4_t{0u, 0u, 0u, 0u}};
    auto load_masks = [&](int j, u32x4_t* dst) {
      size_t opix;
      if (EPI == EPI_MASK && actp && out_pixel(j, opix)) {
#pragma unroll
        for (int ip = 0; ip < 2; ip++) {
          const int n = n0 + nlane + 32 * ip;
          if (n >= N) continue;
          if (p.bits) dst[ip][0] = p.bits[opix * p.ldbits + (n >> 3)];          // one byte instead of 16 (block-uniform choice)
          else dst[ip] = *reinterpret_cast<const u32x4_t*>(actp + opix * p.ldact + n);
        }
      }
    };
    load_masks(0, mk);
#pragma unroll
    for (int j = 0; j < MF; j++) {
      u32x4_t mkn[2] = {u32x4_t{0u, 0u, 0u, 0u}, u32x4_t{0u, 0u, 0u, 0u}};
      if (j + 1 < MF) load_masks(j + 1, mkn);
      f32x4_t v0[2], v1[2];
#pragma unroll
      for (int ip = 0; ip < 2; ip++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {                           // all lanes take part in the exchange (no divergence before it)
          // inline asm: hipcc (ROCm 7.2) folds the four __builtin_amdgcn_permlane16_swap calls of this loop into one
          float xa = acc[2 * ip][j][r], xb = acc[2 * ip + 1][j][r];
          asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(xa), "+v"(xb));
          v0[ip][r] = xa;
          v1[ip][r] = xb;
        }
      }
      size_t opix;
      const bool row_ok = out_pixel(j, opix);
      unsigned wbits[2] = {0u, 0u};
#pragma unroll
      for (int ip = 0; ip < 2; ip++) {
        const int n = n0 + nlane + 32 * ip;
        if (!row_ok || n >= N) continue;                        // N is a multiple of 8
        f32x4_t a = v0[ip], c = v1[ip];
        if (EPI == EPI_BIAS_ACT) {
          if (p.bias) {
            a += *reinterpret_cast<const f32x4_t*>(p.bias + n);
            c += *reinterpret_cast<const f32x4_t*>(p.bias + n + 4);
          }
          if (p.relu) {
#pragma unroll
            for (int r = 0; r < 4; r++) { a[r] = fmaxf(a[r], 0.f); c[r] = fmaxf(c[r], 0.f); }
          }
        } else {
          if (actp && p.bits) apply_relu_bits8(mk[ip][0], a, c);
          else if (actp) {
            const u32x4_t a4 = mk[ip];
#pragma unroll
            for (int h = 0; h < 2; h++) {
              if (!(unpack_lo<T>(a4[h]) > 0.f)) a[2 * h] = 0.f;
              if (!(unpack_hi<T>(a4[h]) > 0.f)) a[2 * h + 1] = 0.f;
              if (!(unpack_lo<T>(a4[2 + h]) > 0.f)) c[2 * h] = 0.f;
              if (!(unpack_hi<T>(a4[2 + h]) > 0.f)) c[2 * h + 1] = 0.f;
            }
          }
          bsum[2 * ip] += a;
          bsum[2 * ip + 1] += c;
          if (p.accumulate) {
            const u32x4_t o4 = *reinterpret_cast<const u32x4_t*>(yout + opix * p.ldy + n);
#pragma unroll
            for (int h = 0; h < 2; h++) {
              a[2 * h] += unpack_lo<T>(o4[h]); a[2 * h + 1] += unpack_hi<T>(o4[h]);
              c[2 * h] += unpack_lo<T>(o4[2 + h]); c[2 * h + 1] += unpack_hi<T>(o4[2 + h]);
            }
          }
        }
        const u32x4_t o = {pack2<T>(a[0], a[1]), pack2<T>(a[2], a[3]), pack2<T>(c[0], c[1]), pack2<T>(c[2], c[3])};
        *reinterpret_cast<u32x4_t*>(yout + opix * p.ldy + n) = o;
        if (EPI == EPI_BIAS_ACT && p.bits && !p.bits_words) p.bits[opix * p.ldbits + (n >> 3)] = (unsigned char)relu_bits8<T>(o);
        if (EPI == EPI_BIAS_ACT && p.bits_words) wbits[ip] = relu_bits8<T>(o) << (((n >> 3) & 3) * 8);
      }
      if (EPI == EPI_BIAS_ACT && p.bits_words) {                 // block-uniform: every lane takes part in the row exchange
#pragma unroll
        for (int ip = 0; ip < 2; ip++) {
          const unsigned wd = rows4_or(wbits[ip]);               // the 32 channels n0 + 64 wn + 32 ip .. of this lane's pixel
          const int n32 = n0 + wn * 64 + 32 * ip;
          if (eg == 0 && row_ok && n32 < N) *reinterpret_cast<unsigned*>(p.bits + opix * p.ldbits + (n32 >> 3)) = wd;
        }
      }
      mk[0] = mkn[0]; mk[1] = mkn[1];
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
#pragma unroll
  for (int j = 0; j < MF; j++) {
    const int m = m0 + wm * WM + j * 16 + (elane & 15);
    if (m >= M) continue;
    size_t opix;
    if (FORM != FORM_CONVT) opix = (size_t)m;
    else {
      int sw, sh, b;
      decode_pixel(m, Hs, Ws, p.hs_shift, p.ws_shift, sw, sh, b);
      opix = ((size_t)b * (2 * Hs) + 2 * sh + ph) * (2 * Ws) + 2 * sw + pw;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int n = n0 + wn * 64 + i * 16 + 4 * (elane >> 4);
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
        bsum[i] += v;
        if (p.accumulate) {
          const u32x2_t o2 = *reinterpret_cast<const u32x2_t*>(yout + opix * p.ldy + n);
          v[0] += unpack_lo<T>(o2[0]); v[1] += unpack_hi<T>(o2[0]);
          v[2] += unpack_lo<T>(o2[1]); v[3] += unpack_hi<T>(o2[1]);
        }
      }
      u32x2_t o = {pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
      *reinterpret_cast<u32x2_t*>(yout + opix * p.ldy + n) = o;
    }
    __builtin_amdgcn_sched_barrier(0);   // one 16-pixel column of the tile at a time: bounds the epilogue's live registers
  }
  }
  TG_STAMP(3);
#ifdef GCT2_STAMP
  if (stamp_out && lane == 0) {
    TG_STAMP(4);
    unsigned long long* o = stamp_out + ((size_t)blockIdx.x * NWV + wave) * 8;
    for (int k = 0; k < 5; k++) o[k] = st[k];
  }
#endif
  if (EPI == EPI_MASK && (p.db || p.db2) && p.ksplit == 1) {
    // column sums over the wave's WM pixels: butterfly over the 16 lanes that share (lane>>4); then the waves of one
    // tile column meet in LDS (free after the K loop's last barrier) and the work-group stores ONE partial row,
    // dbws[m_tile * PH + phase][n0 .. n0+BN), summed in a fixed order by dbpart_reduce_kernel: no atomics, no zeroing,
    // bit-reproducible.  Without a workspace the sums go straight to db with atomics.
    float* red = reinterpret_cast<float*>(lds0);
    constexpr int WAVES_M = NWV / WAVES_N;
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float t = bsum[i][r];
        t = row16_sum(t);
        const int eg2 = elane >> 4;
        const int c = wide ? wn * 64 + 32 * (i >> 1) + 16 * (eg2 & 1) + 4 * (eg2 & ~1) + 4 * (i & 1) + r   // bsum[2k + h][r] after the swap
                           : wn * 64 + i * 16 + 4 * eg2 + r;
        if ((elane & 15) == 0) {
          if (p.dbws) red[wm * BN + c] = t;
          else if (n0 + c < N) {
            float* q = db_target(p, n0 + c);
            if (q) atomicAdd(q, t);
          }
        }
      }
    }
    if (p.dbws) {
      __syncthreads();
      if (tid < BN && n0 + tid < N) {
        float t = red[tid];
#pragma unroll
        for (int k = 1; k < WAVES_M; k++) t += red[k * BN + tid];
        p.dbws[(size_t)(m_tile * PH + phase) * N + n0 + tid] = t;
      }
    }
  }
#ifdef GCT2_STAMP
  GCT2_CLOCK_EXIT(p.clock ? p.stamps : nullptr, NWV, wave, lane);
#endif
}

// db[n] += sum over the partial rows part[rows][N] left by the GEMM epilogue / the split-K finalize (fixed order).
// Work-group = 32 columns x 128 row lanes; a wave holds 8 row lanes x 8 float4 column quads.
__global__ __launch_bounds__(1024) void dbpart_reduce_kernel(const float* __restrict__ part, int rows, TapGemmParams p) {
  const int N = p.N;
  const int tid = threadIdx.x, cq = tid & 7, rl = tid >> 3;
  const int n = blockIdx.x * 32 + cq * 4;
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  if (n < N)
    for (int r = rl; r < rows; r += 128) acc += *reinterpret_cast<const f32x4_t*>(part + (size_t)r * N + n);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    float t = acc[k];
    t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
    acc[k] = t;
  }
  __shared__ f32x4_t red[16][8];
  if ((tid & 63) < 8) red[tid >> 6][cq] = acc;
  __syncthreads();
  if (tid < 8 && n < N) {
    f32x4_t t = red[0][tid];
#pragma unroll
    for (int k = 1; k < 16; k++) t += red[k][tid];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float* q = db_target(p, n + r);
      if (q) *q = db_adds(p, n + r) ? *q + t[r] : t[r];
    }
  }
}

// the same reduction for every row set of a bias queue (gct2_ctx_set_bias_queue) in ONE launch: work-group = 32 columns of one job
// (blk0[k] = first work-group of job k), same row lanes, same order of additions as dbpart_reduce_kernel - the same bits.  A launch
// writes only the targets of its phase: 0 = the ones a job overwrites, 1 = the ones it adds to (a target gets its first writer's sums
// before its second writer's: two launches, in this order).
struct DbJobs { int njobs, phase; int blk0[17]; gct2_ctx::DbJob j[16]; };
__global__ __launch_bounds__(1024) void dbpart_reduce_multi_kernel(const DbJobs J) {
  int k = 0;
  while (k + 1 < J.njobs && (int)blockIdx.x >= J.blk0[k + 1]) k++;
  const gct2_ctx::DbJob job = J.j[k];
  const int N = job.N, rows = job.rows;
  const float* __restrict__ part = job.part;
  const int tid = threadIdx.x, cq = tid & 7, rl = tid >> 3;
  const int n = ((int)blockIdx.x - J.blk0[k]) * 32 + cq * 4;
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  if (n < N)
    for (int r = rl; r < rows; r += 128) acc += *reinterpret_cast<const f32x4_t*>(part + (size_t)r * N + n);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    float t = acc[i];
    t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
    acc[i] = t;
  }
  __shared__ f32x4_t red[16][8];
  if ((tid & 63) < 8) red[tid >> 6][cq] = acc;
  __syncthreads();
  if (tid < 8 && n < N) {
    f32x4_t t = red[0][tid];
#pragma unroll
    for (int i = 1; i < 16; i++) t += red[i][tid];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int col = n + r;
      float* q = col < job.db_split ? (job.db ? job.db + col : nullptr) : (job.db2 ? job.db2 + (col - job.db_split) : nullptr);
      const int adds = (job.db_acc >> (col < job.db_split ? 0 : 1)) & 1;
      if (q && adds == J.phase) *q = adds ? *q + t[r] : t[r];
    }
  }
}

// sums the split-K slabs and applies the epilogue the GEMM kernel skipped.  Work-group = 8 pixels x 128 channels,
// thread = 4 channels of one pixel (split-K layers have few pixels: keep the grid wide), and the bias-gradient
// column sums of the 8 pixels meet inside a wave (three xor-shuffles, no LDS) as one partial row for dbpart_reduce_kernel / the bias queue.
template <typename T, int EPI>
__global__ __launch_bounds__(256) void tapgemm_finalize_kernel(TapGemmParams p, size_t npix) {
  const int N = p.N;
  // r05: a wave = 8 pixels x 32 channels (lane = 8 * pixel + channel quad), so the eight pixels of a column meet INSIDE a wave (three
  // xor-shuffles) and the kernel needs no LDS: with 40 registers and no LDS its work-groups fit beside the weight-gradient tile that owns
  // every CU's LDS on the other stream, instead of waiting for one of those work-groups to retire (the 4 KiB of the r01-r04 form made
  // this 8-us kernel take 30-110 us in the step)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ty = lane >> 3;
  const int n = blockIdx.y * 128 + wave * 32 + (lane & 7) * 4;
  const size_t pix0 = (size_t)blockIdx.x * 8;
  T* __restrict__ yout = reinterpret_cast<T*>(p.y);
  f32x4_t bsum = {0.f, 0.f, 0.f, 0.f};
  if (n < N) {
    for (int k = 0; k < 1; k++) {
      const size_t opix = pix0 + ty;
      if (opix >= npix) break;
      f32x4_t v = {0.f, 0.f, 0.f, 0.f};
      {   // slab loads issued 8 at a time (independent), summed in slab order
        const float* base = p.ws + opix * N + n;
        const size_t sstride = npix * (size_t)N;
        int s = 0;
        // (the input-gradient form keeps TWO loads in flight and 30 registers: it has to fit into the 32 registers per lane that two
        // 240-register weight-gradient waves leave on a SIMD; the forward form runs beside nothing of that kind: eight loads)
        constexpr int NLD = EPI == EPI_MASK ? 2 : 8;
        for (; s + NLD <= p.ksplit; s += NLD) {
          f32x4_t t[NLD];
#pragma unroll
          for (int u = 0; u < NLD; u++) t[u] = *reinterpret_cast<const f32x4_t*>(base + (size_t)(s + u) * sstride);
#pragma unroll
          for (int u = 0; u < NLD; u++) v += t[u];
        }
        for (; s < p.ksplit; s++) v += *reinterpret_cast<const f32x4_t*>(base + (size_t)s * sstride);
      }
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
        bsum += v;
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
  if (EPI == EPI_MASK && (p.db || p.db2)) {
    // column sums of the work-group's 8 pixels: pixel pairs (1 apart), then 2 apart, then 4 apart - a fixed tree, every lane ends
    // with the total; the lanes of pixel 0 store the partial row (lanes outside N hold zeros and store nothing)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float t = bsum[r];
      t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
      bsum[r] = t;
    }
    if (ty == 0 && n < N) {
      if (p.dbws) *reinterpret_cast<f32x4_t*>(p.dbws + (size_t)blockIdx.x * N + n) = bsum;   // one partial row per work-group row
      else {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          float* q = db_target(p, n + r);
          if (q) atomicAdd(q, bsum[r]);
        }
      }
    }
  }
}

template <typename T, int FORM, int BM, int BN, int EPI, int NBUF>
int launch(gct2_ctx& c, TapGemmParams p, hipStream_t s) {
  const int M = p.B * p.Hs * p.Ws;
  constexpr int PH = FORM == FORM_CONVT ? 4 : 1;
  const int tiles = ((M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * PH;
  const int ntaps = FORM == FORM_CONV ? 16 : (FORM == FORM_CONVT ? 4 : p.ks * p.ks);
  constexpr int BKS = BK;
  const int niter = ntaps * ((p.K + BKS - 1) / BKS);
  const size_t npix = (size_t)M * PH;
  // small-M layers (bottleneck of the U-Net) cannot fill 256 CUs with output tiles: split the reduction
  p.ksplit = 1;
  p.ws = nullptr;
  p.wide = ((uintptr_t)p.y % 16 == 0 && p.ldy % 8 == 0 && (!p.act || ((uintptr_t)p.act % 16 == 0 && p.ldact % 8 == 0))) ? 1 : 0;
  const size_t ws_bytes = c.ws_bytes;
  float* ws = c.ws;
  p.m_tiles = (M + BM - 1) / BM;
  p.n_tiles = (p.N + BN - 1) / BN;
  p.ws_shift = pow2_shift(p.Ws);
  p.hs_shift = pow2_shift(p.Hs);
  const bool want_db = EPI == EPI_MASK && (p.db || p.db2);
#ifdef GCT2_STAMP
  p.stamps = c.stamps;
  p.clock = c.stamps_bytes >= GCT2_CLOCK_BYTES ? 1 : 0;
#endif
  // fused bias gradient: partial rows at the tail of the workspace (one per (m-tile, phase), or per finalize work-group)
  const size_t fin_rows = (npix + 7) / 8;
  const size_t dbws_bytes = want_db ? std::max((size_t)p.m_tiles * PH, fin_rows) * p.N * sizeof(float) : 0;
  const bool db_rows = want_db && ws && ws_bytes >= dbws_bytes + 16;
  const size_t slab_room = db_rows ? ws_bytes - dbws_bytes - 16 : ws_bytes;
  // (below ~1.2 work-groups per CU a second k-slice per tile beats the idle half of the chip: U_3 dgrad 102 -> 77 us)
  if (ws && !c.no_splitk && tiles < 300 && niter >= 4 * BK / BKS) {
    int want = (512 + tiles - 1) / tiles;
    const size_t slab = npix * p.N * sizeof(float);
    want = (int)std::min<size_t>((size_t)want, slab_room / slab);
    want = std::min(want, niter / (8 * BK / BKS));   // >= 8 K-steps per work-group: shorter slices are all prologue + slab traffic (2x2 levels: 43 -> 24 us)
    if (want >= 2) {
      const int per = (niter + want - 1) / want;
      p.ksplit = (niter + per - 1) / per;
      p.ws = ws;
    }
  }
  p.xcd_chunk = (p.m_tiles + 7) / 8;
  // tile -> XCD order: by default an XCD owns a band of m-tiles (the source pixels are re-read by every n-tile / phase / tap); when
  // the weight tensor is the bigger operand and there are at least 8 weight slices, an XCD owns weight slices instead (p.wstat)
  const size_t w_bytes = (size_t)(FORM == FORM_CONV || FORM == FORM_CONVT ? 16 : ntaps) * p.K * p.N * 2, src_bytes = (size_t)M * (FORM == FORM_CONV ? 4 : 1) * p.ldx * 2;
  const int slices = p.n_tiles * PH * p.ksplit;
  p.wstat = (c.xcd_order == 2 || (c.xcd_order == 0 && w_bytes >= 3 * src_bytes)) && slices >= 8 ? 1 : 0;   // measured per layer: profiles/r02_layers.txt
  dim3 grid(p.wstat ? 8 * ((slices + 7) / 8) * p.m_tiles : 8 * p.xcd_chunk * p.n_tiles * PH * p.ksplit);
  p.bits_words = (p.bits && (uintptr_t)p.bits % 4 == 0 && p.ldbits % 4 == 0 && p.N % 32 == 0) ? 1 : 0;
  auto kern = tapgemm_kernel<T, FORM, BM, BN, EPI, NBUF>;
  p.dbws = db_rows ? ws + (ws_bytes - dbws_bytes) / sizeof(float) / 4 * 4 : nullptr;
  // a registered bias queue takes the partial rows instead (same rows, same later reduction; the tile / split-K choice above does not
  // depend on it); no room in the queue: everything queued so far is reduced first, then this call reduces its own rows at once
  float* queued = (db_rows && c.dbq) ? c.dbq_alloc((p.ksplit > 1 ? fin_rows : (size_t)p.m_tiles * PH) * p.N) : nullptr;   // (the rows this launch really leaves)
  if (queued) p.dbws = queued;
  else if (db_rows && c.dbq) { if (int e = tapgemm_dbq_flush(c, s)) return e; }
  if (want_db && !p.dbws) {                  // the epilogue adds with atomics, at once: queued row sets of these targets go first
    if (int e = tapgemm_dbq_flush_for(c, p.db, p.db_split, p.db2, p.N - p.db_split, s)) return e;
    zero_overwritten_db(p, s);
  }
  gct2_log(c, "tap:%s:%dx%d:%s:ksplit=%d%s%s", FORM == FORM_CONV ? "conv" : FORM == FORM_CONVT ? "convT" : "s1", BM, BN,
           EPI == EPI_BIAS_ACT ? "bias_act" : "mask", p.ksplit, p.wstat ? ":wstat" : "", (p.bits && p.wide && p.ksplit == 1) ? ":bits" : "");
  hipLaunchKernelGGL(kern, grid, dim3((BM / 64) * (BN / 64) * 64), 0, s, p);
  if (p.ksplit > 1) {
    hipLaunchKernelGGL((tapgemm_finalize_kernel<T, EPI>), dim3((unsigned)fin_rows, (p.N + 127) / 128), dim3(256), 0, s, p, npix);
  }
  if (EPI == EPI_BIAS_ACT && p.bits && p.wide && p.ksplit == 1) c.relu_bits_done = 1;   // the 16-byte epilogue wrote the ReLU bit plane
  if (p.dbws) {
    const int rows = p.ksplit > 1 ? (int)fin_rows : p.m_tiles * PH;
    if (queued) { if (int e = tapgemm_dbq_push(c, queued, rows, p, s)) return e; }
    else hipLaunchKernelGGL(dbpart_reduce_kernel, dim3((p.N + 31) / 32), dim3(1024), 0, s, p.dbws, rows, p);
  }
  return gct2_check_launch("tapgemm_mfma");
}

template <typename T>
int dispatch(gct2_ctx& c, int form, int epi, const TapGemmParams& p, hipStream_t s) {
  // the stride-1 forms (off-by-default model variants): one tile shape; the epilogue is the forward's / the input gradient's
  // (tapgemm_mfma checks the pairing)
  if (form == FORM_S1) return launch<T, FORM_S1, 128, 128, EPI_BIAS_ACT, 2>(c, p, s);
  if (form == FORM_S1T) return launch<T, FORM_S1T, 128, 128, EPI_MASK, 2>(c, p, s);
  // automatic choice (per-layer A/B: profiles/r01_layer_variants_final.txt, r03_layer_variants.txt): the 256 x 128 single-buffer
  // tile (8 waves, 48 KiB, two work-groups per CU) moves 25 % fewer L2->LDS bytes per FLOP than the 128 x 128 two-buffer tile and
  // wins 5-14 % where it still yields >= 2 work-groups per CU; N <= 64 (UpShuffle_0 without the halo kernel) takes 256 x 64
  const int M = p.B * p.Hs * p.Ws;
  const int tiles256 = ((M + 255) / 256) * ((p.N + 127) / 128) * (form == FORM_CONVT ? 4 : 1);
  const bool big = c.tap_variant == 5 || (c.tap_variant == 0 && tiles256 >= 512);
  if (big && p.N > 64) {
    if (form == FORM_CONV) return epi == EPI_BIAS_ACT ? launch<T, FORM_CONV, 256, 128, EPI_BIAS_ACT, 1>(c, p, s)
                                                      : launch<T, FORM_CONV, 256, 128, EPI_MASK, 1>(c, p, s);
    return epi == EPI_BIAS_ACT ? launch<T, FORM_CONVT, 256, 128, EPI_BIAS_ACT, 1>(c, p, s)
                               : launch<T, FORM_CONVT, 256, 128, EPI_MASK, 1>(c, p, s);
  }
  if (form == FORM_CONV)
    return epi == EPI_BIAS_ACT ? launch<T, FORM_CONV, 128, 128, EPI_BIAS_ACT, 2>(c, p, s) : launch<T, FORM_CONV, 128, 128, EPI_MASK, 2>(c, p, s);
  if (p.N <= 64)
    return epi == EPI_BIAS_ACT ? launch<T, FORM_CONVT, 256, 64, EPI_BIAS_ACT, 2>(c, p, s) : launch<T, FORM_CONVT, 256, 64, EPI_MASK, 2>(c, p, s);
  return epi == EPI_BIAS_ACT ? launch<T, FORM_CONVT, 128, 128, EPI_BIAS_ACT, 2>(c, p, s) : launch<T, FORM_CONVT, 128, 128, EPI_MASK, 2>(c, p, s);
}

}  // namespace

// true when the MFMA path can take this problem (16-byte aligned rows, whole 8-channel chunks, 31-bit byte offsets)
bool tapgemm_mfma_supported(int dtype, const TapGemmParams& p) {
  if (dtype != GCT2_BF16 && dtype != GCT2_F16) return false;
  if (p.K % 8 || p.N % 8 || p.ldx % 8 || p.ldy % 4) return false;
  if (p.act && p.ldact % 4) return false;
  if ((uintptr_t)p.x % 16 || (uintptr_t)p.w % 16 || (uintptr_t)p.y % 8) return false;
  if (p.act && (uintptr_t)p.act % 8) return false;
  if (p.bias && (uintptr_t)p.bias % 16) return false;
  // buffer descriptors address 2 GiB: source tensor (BIG grid for conv-form) and the 16-tap weight tensor.  The lean issue bases the
  // source descriptor one row + one pixel IN FRONT of the tensor and adds that shift to every per-lane offset, whose bit 31 is the
  // "invalid tap" flag: the largest offset a valid lane can form (tensor + shift + tap / chunk part) must stay below 2^31
  const size_t src_bytes = (size_t)p.B * p.Hs * p.Ws * 4 * p.ldx * 2;
  const size_t shift = (size_t)(2 * p.Ws + 1) * p.ldx * 2, tap_max = (size_t)(3 * 2 * p.Ws + 3) * p.ldx * 2 + (size_t)p.K * 2;
  const size_t w_bytes = (size_t)16 * p.K * p.N * 2;
  if (src_bytes + shift + tap_max >= 0x7ff00000u || w_bytes >= 0x7ff00000u) return false;
  return true;
}

bool halo_convT_wanted(const gct2_ctx& c, int epi, const TapGemmParams& p);      // halo_mfma.hip
int halo_convT(gct2_ctx& c, int dtype, int epi, TapGemmParams p, hipStream_t s);

// the ordered row reduction of the fused bias gradients, for the other translation units that leave partial rows
// bias queue: record a row set / reduce everything recorded (two launches: overwriting targets, then adding ones)
int tapgemm_dbq_flush(gct2_ctx& c, hipStream_t s) {
  if (c.dbq_jobs.empty()) { c.dbq_used = 0; return GCT2_OK; }
  DbJobs J{};
  J.njobs = (int)c.dbq_jobs.size();
  int blocks = 0, any[2] = {0, 0};
  for (int k = 0; k < J.njobs; k++) {
    J.j[k] = c.dbq_jobs[k];
    J.blk0[k] = blocks;
    blocks += (J.j[k].N + 31) / 32;
    const gct2_ctx::DbJob& j = J.j[k];
    if (j.db && j.db_split > 0) any[j.db_acc & 1] = 1;
    if (j.db2 && j.db_split < j.N) any[(j.db_acc >> 1) & 1] = 1;
  }
  J.blk0[J.njobs] = blocks;
  gct2_log(c, "bias_queue:flush:sets=%d", J.njobs);
  c.dbq_jobs.clear();
  c.dbq_used = 0;
  for (int phase = 0; phase < 2; phase++) {
    if (!any[phase]) continue;
    J.phase = phase;
    hipLaunchKernelGGL(dbpart_reduce_multi_kernel, dim3(blocks), dim3(1024), 0, s, J);
  }
  return gct2_check_launch("bias_queue_flush");
}
namespace {
inline bool dbq_overlap(const float* x, int nx, const float* y, int ny) { return x && y && nx > 0 && ny > 0 && x < y + ny && y < x + nx; }
}
// An IMMEDIATE writer of bias gradients (direct kernels, the atomics epilogues taken without room for partial rows, a weight-gradient
// call's db) is about to touch [db, db + n0) / [db2, db2 + n1): row sets queued for the same targets are reduced first, in program
// order - otherwise a queued OVERWRITE would run behind this call's add and erase it.  No overlap: nothing happens.
int tapgemm_dbq_flush_for(gct2_ctx& c, const float* db, int n0, const float* db2, int n1, hipStream_t s) {
  for (const gct2_ctx::DbJob& j : c.dbq_jobs) {
    const float* jt[2] = {j.db_split > 0 ? j.db : nullptr, j.db_split < j.N ? j.db2 : nullptr};
    const int jn[2] = {j.db_split, j.N - j.db_split};
    for (int k = 0; k < 2; k++)
      if (dbq_overlap(db, n0, jt[k], jn[k]) || dbq_overlap(db2, n1, jt[k], jn[k])) return tapgemm_dbq_flush(c, s);
  }
  return GCT2_OK;
}
int tapgemm_dbq_push(gct2_ctx& c, const float* part, int rows, const TapGemmParams& p, hipStream_t s) {
  // The flush runs every overwriting target first and every adding target second, so per target only the order "one queued
  // OVERWRITE, then one ADD" survives being queued together.  Everything else that meets a queued job on the same target - a second
  // add (two adders would race inside the second launch), a second overwrite (a race inside the first), an overwrite behind a queued
  // add (would be reordered in front of it) - reduces what is queued before this job is recorded.
  const float* pt[2] = {p.db_split > 0 ? p.db : nullptr, p.db_split < p.N ? p.db2 : nullptr};
  const int pn[2] = {p.db_split, p.N - p.db_split};
  const bool padd[2] = {(p.db_acc & 1) != 0, (p.db_acc & 2) != 0};
  bool clash = false;
  for (const gct2_ctx::DbJob& j : c.dbq_jobs) {
    const float* jt[2] = {j.db_split > 0 ? j.db : nullptr, j.db_split < j.N ? j.db2 : nullptr};
    const int jn[2] = {j.db_split, j.N - j.db_split};
    const bool jadd[2] = {(j.db_acc & 1) != 0, (j.db_acc & 2) != 0};
    for (int a = 0; a < 2; a++)
      for (int b = 0; b < 2; b++)
        if (dbq_overlap(pt[a], pn[a], jt[b], jn[b]) && !(padd[a] && !jadd[b])) clash = true;
  }
  if (clash) {
    // (the rows of THIS job are already in the queue buffer: keep them valid across the flush by flushing the jobs only)
    const size_t used = c.dbq_used;
    if (int e = tapgemm_dbq_flush(c, s)) return e;
    c.dbq_used = used;
  }
  c.dbq_jobs.push_back(gct2_ctx::DbJob{part, rows, p.N, p.db, p.db_split, p.db2, p.db_acc});
  return GCT2_OK;
}

int tapgemm_dbpart_reduce(const float* part, int rows, const TapGemmParams& p, hipStream_t s) {
  hipLaunchKernelGGL(dbpart_reduce_kernel, dim3((p.N + 31) / 32), dim3(1024), 0, s, part, rows, p);
  return gct2_check_launch("dbpart_reduce");
}

int tapgemm_mfma(gct2_ctx& c, int dtype, int form, int epi, const TapGemmParams& p, hipStream_t s) {
  if (form == FORM_S1 || form == FORM_S1T) {
    if (epi != (form == FORM_S1 ? EPI_BIAS_ACT : EPI_MASK) || p.ks < 1 || p.ks > 5 || !(p.ks & 1))
      return gct2_fail(GCT2_EINVAL, "tapgemm_mfma: stride-1 form with kernel size %d / epilogue %d", p.ks, epi);
    if (dtype == GCT2_BF16) return dispatch<__bf16>(c, form, epi, p, s);
    return dispatch<_Float16>(c, form, epi, p, s);
  }
  // (a forced tile keeps the halo kernel out, unless the halo kernel is forced too: halo mode 2)
  if (form == FORM_CONVT && (c.tap_variant == 0 || c.halo_mode == 2) && halo_convT_wanted(c, epi, p)) return halo_convT(c, dtype, epi, p, s);
  if (dtype == GCT2_BF16) return dispatch<__bf16>(c, form, epi, p, s);
  return dispatch<_Float16>(c, form, epi, p, s);
}
