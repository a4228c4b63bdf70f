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
// HIDDEN = false: the builtin; hipcc counts the DMA in its own vmcnt bookkeeping (and drains it to 0 at the head of a
// loop that keeps more than one step in flight).  HIDDEN = true: the same instruction from inline asm, invisible to
// that bookkeeping; every wait for it is then placed by hand (the 3-buffer loop).  M0 = LDS address of the piece is
// written in the same statement that uses it (cdna_hip_programming.md §5.7).
template <bool HIDDEN>
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_piece, unsigned voff) {
  if constexpr (HIDDEN) {
    const unsigned lds_addr = (unsigned)(uintptr_t)(lds_void_t*)lds_piece;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc)
                 : "memory", "m0");
  } else {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)lds_piece, 16, (int)voff, 0, 0, 0);
  }
}

// the same with a wave-uniform byte offset in an SGPR (tap / k-chunk part of the address; the bounds check sees the per-lane offset)
template <bool HIDDEN>
__device__ __forceinline__ void dma16s(__amdgpu_buffer_rsrc_t rsrc, char* lds_piece, unsigned voff, unsigned soff) {
  if constexpr (HIDDEN) {
    const unsigned lds_addr = (unsigned)(uintptr_t)(lds_void_t*)lds_piece;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory", "m0");
  } else {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)lds_piece, 16, (int)voff, (int)soff, 0, 0);
  }
}

// bias gradient = column sums of the masked gradient a dgrad launch produces: db[n] (+)= sum over pixels.  Channels
// [0, db_split) go to p.db, the rest to p.db2 (the output of an UpShuffle dgrad spans two layers' pre-activations).
__device__ __forceinline__ float* db_target(const TapGemmParams& p, int n) {
  return n < p.db_split ? (p.db ? p.db + n : nullptr) : (p.db2 ? p.db2 + (n - p.db_split) : nullptr);
}
// whether the target of channel n is added to (db_accumulate bits of include/gct2.h) or overwritten
__device__ __forceinline__ bool db_adds(const TapGemmParams& p, int n) { return (p.db_acc >> (n < p.db_split ? 0 : 1)) & 1; }

// NBUF = 2: 4 waves (256 threads), 2 work-groups per CU cover each other's DMA latency, vmcnt(0) per step.
// NBUF = 3: 8 waves (512 threads, 256 x 128 tile), 1 work-group per CU, the DMA of step t+2 stays in flight
//           across the barrier that publishes step t+1 (counted vmcnt + raw s_barrier).
// WM = 64 : every wave owns a 64 (m) x 64 (n) sub-tile;  WM = 128: 128 (m) x 64 (n), used by the 256 x 256 tile
//           (8 waves; 96 LDS bytes per MFMA instead of 128, half the L2->LDS bytes per flop of the 128 x 128 tile).
#ifdef GCT2_STAMP
// diagnostic build (make EXTRA=-DGCT2_STAMP, scripts/stamp_layer.py): s_memrealtime at the phase boundaries of one wave per work-group,
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
template <typename T, int FORM, int BM, int BN, int EPI, int NBUF, int WM = 64>
__global__ __launch_bounds__((BM / WM) * (BN / 64) * 64, (NBUF == 1 && WM == 64) ? 4 : (WM == 64 ? 2 : 1)) void tapgemm_kernel(TapGemmParams p) {
  constexpr int NWV = (BM / WM) * (BN / 64);       // waves, each a WM x 64 sub-tile
  constexpr int MF = WM / 16;                      // 16-pixel fragments per wave
  static_assert(NWV == 4 || NWV == 8, "4 or 8 waves");
  constexpr bool S1 = (FORM == FORM_S1 || FORM == FORM_S1T);   // stride-1 'same' convolution: p.ks x p.ks taps on the output's own grid
  constexpr bool WT = (FORM == FORM_CONV || FORM == FORM_S1);   // weights [tap][k][n] (T image); otherwise [tap][n][k] (N image)
  static_assert(!WT || BN % 128 == 0, "T images are 128 columns wide");
  constexpr int WAVES_N = BN / 64;
  // RING (NBUF = 5): the deep pipeline of wgrad256p_kernel for the forward / input-gradient GEMMs.  A stage is HALF a 64-channel
  // step (32 channels of one tap: 16 KiB of pixels + 16 KiB of weights for the 256 x 256 tile), five stage buffers fill the CU's
  // 160 KiB of LDS, the DMA of stage s+4 is issued while stage s is multiplied and only stage s+1 is waited for (counted vmcnt,
  // raw s_barrier): FOUR stages = 128 KiB stay in flight across every barrier, against one step drained to zero at every barrier
  // in the other variants.  These loops are bound by the latency of the staging requests that miss the XCD's L2.
  constexpr bool RING = NBUF >= 5;             // NBUF = 6: the ring with the DMA pieces interleaved into the MFMA groups
  constexpr bool RING_IL = NBUF == 6;
  static_assert(!RING || (NWV == 8 && !S1), "the ring pipeline is built for the 8-wave 4x4 / stride-2 forms");
  constexpr int BKS = RING ? 32 : BK;              // reduction elements per stage
  constexpr int NA = RING ? BM / 16 / NWV : BM / 8 / NWV;   // 1-KiB pieces per wave, activation tile (8 rows of 128 B, or 16 rows of 64 B)
  constexpr int NW = RING ? (WT ? 8 * (BN / 128) : BN / 16) / NWV : (WT ? 16 * (BN / 128) : BN / 8) / NWV;
  constexpr int NDMA = NA + NW;                    // DMA instructions per wave per step
  constexpr int A_BYTES = RING ? BM * 64 : BM * 128;
  constexpr int W_BYTES = RING ? (WT ? 32 * 256 * (BN / 128) : BN * 64)
                               : (WT ? 64 * 256 * (BN / 128) : BN * 128);   // T images: BN/128 of them side by side
  constexpr int NTAPS = (FORM == FORM_CONV) ? 16 : 4;              // (the stride-1 forms: p.ks * p.ks, run-time)

  // DISTINCT LDS objects: lets hipcc prove that the DMA into one buffer does not alias the ds_reads of another,
  // so it does not drain vmcnt before every read (cdna_hip_programming.md, "Three .s-level traps" (a))
  __shared__ __attribute__((aligned(16))) char lds0[A_BYTES + W_BYTES];
  __shared__ __attribute__((aligned(16))) char lds1[NBUF >= 2 ? A_BYTES + W_BYTES : 16];
  __shared__ __attribute__((aligned(16))) char lds2[NBUF >= 3 ? A_BYTES + W_BYTES : 16];
  __shared__ __attribute__((aligned(16))) char lds3[RING ? A_BYTES + W_BYTES : 16];
  __shared__ __attribute__((aligned(16))) char lds4[RING ? A_BYTES + W_BYTES : 16];

#ifdef GCT2_STAMP
  unsigned long long st[5];
  unsigned long long* stamp_out = p.stamps;
  TG_STAMP(0);
#endif
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
  // RING: 64-byte rows (32 channels), piece q = wave + 8 i holds rows 16q .. 16q+15; lane -> row 16q + (lane>>2), physical chunk
  // lane&3 = logical chunk ^ ring_swz((row>>2)&3) (conflict-free for ds_read_b128 of 16 consecutive rows: checked exhaustively
  // against the lane groups of MI355X_MICROARCH.md, LDS table)
  const int a_lchunk = RING ? ((lane & 3) ^ ring_swz((lane >> 4) & 3))
                            : ((lane & 7) ^ ((4 * wave + (lane >> 4)) & 7));   // 4*NWV*i is a multiple of 8
  unsigned a_off[NA];                              // byte offset of (row's tap-origin pixel, logical chunk)
  unsigned a_mask[NA];                             // bit t: tap t reads inside the image for this row
#pragma unroll
  for (int i = 0; i < NA; i++) {
    const int m = RING ? m0 + 16 * (wave + NWV * i) + (lane >> 2) : m0 + 8 * (wave + NWV * i) + (lane >> 3);
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
    if (RING && WT) {                              // 32-row T images: piece `wave` of image i = k-rows 4 wave .. 4 wave + 3
      const int k = 4 * wave + (lane >> 4);
      const int lc = ((((lane & 15) >> 1) ^ timg_swz(k)) << 1) | (lane & 1);
      const int nn = n0 + i * 128 + lc * 8;
      w_k[i] = k;
      w_nok[i] = nn < N;
      w_off[i] = (unsigned)((k * N + nn) * 2);
    } else if (RING) {                             // 64-byte-row N image of the weights: piece = 16 n-rows x 4 chunks
      const int n = 16 * (wave + NWV * i) + (lane >> 2);
      w_k[i] = a_lchunk * 8;
      w_nok[i] = (n0 + n) < N;
      w_off[i] = (unsigned)(((n0 + n) * K + a_lchunk * 8) * 2);
    } else if (WT) {                               // T image: piece = 4 k-rows x 16 chunks
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
          dma16s<NBUF >= 3>(rs_xs, abase + (wave + NWV * i) * 1024, a_voff[i] | ((a_nmask[i] >> abit) << 31), s_a);
#pragma unroll
        for (int i = 0; i < NW; i++)
          dma16s<NBUF >= 3>(rs_w, wbase + ((RING && WT) ? i * (32 * 256) + wave * 1024 : (wave + NWV * i) * 1024), w_voff[i], s_w);
      } else {                                      // the ragged last chunk (K % 64 != 0)
        asm volatile("" ::: "memory");              // keeps hipcc from turning this branch into selects in the full-chunk path
#pragma unroll
        for (int i = 0; i < NA; i++) {
          const unsigned v = a_voff[i] | ((a_nmask[i] >> abit) << 31);
          dma16s<NBUF >= 3>(rs_xs, abase + (wave + NWV * i) * 1024, (c0 + a_lchunk * 8) < K ? v : OOB, s_a);
        }
#pragma unroll
        for (int i = 0; i < NW; i++)
          dma16s<NBUF >= 3>(rs_w, wbase + ((RING && WT) ? i * (32 * 256) + wave * 1024 : (wave + NWV * i) * 1024),
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
      dma16<NBUF >= 3>(rs_x, abase + (wave + NWV * i) * 1024, ok ? a_off[i] + tapoff : OOB);
    }
    char* wbase = abase + A_BYTES;
    const unsigned wtap = WT ? (unsigned)(((tap16 * K + c0) * N) * 2) : (unsigned)((tap16 * N * K + c0) * 2);
#pragma unroll
    for (int i = 0; i < NW; i++) {
      const bool ok = w_nok[i] && (c0 + w_k[i]) < K;
      // (RING + T image: piece `wave` of image i at i * 8 KiB; every other layout: consecutive pieces)
      dma16<NBUF >= 3>(rs_w, wbase + ((RING && WT) ? i * (32 * 256) + wave * 1024 : (wave + NWV * i) * 1024), ok ? w_off[i] + wtap : OOB);
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
  // RING: one 32-deep MFMA step per stage; fragments of the 64-byte-row images / the 32-row T images
  auto compute_ring = [&](const char* a_img) {
    const char* w_img = a_img + A_BYTES;
    int ql = lane;                                 // opaque copy: keeps the fragment addresses of the five unrolled stages out of
    asm volatile("" : "+v"(ql));                   // the loop preamble (they would be hoisted and spill, as in wgrad256p_kernel)
    u32x4_t wf[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
      wf[i] = WT ? timg_frag(w_img + (wn >> 1) * (32 * 256), (wn & 1) * 64 + i * 16, 0, ql) : ring_frag(w_img, wn * 64 + i * 16, ql);
#pragma unroll
    for (int j = 0; j < MF; j++) {
      const u32x4_t af = ring_frag(a_img, wm * WM + j * 16, ql);
#pragma unroll
      for (int i = 0; i < 4; i++) acc[i][j] = mfma16<T>(wf[i], af, acc[i][j]);
    }
  };
  // `live` is always true (ksplit >= 1) but opaque to hipcc: a code-generation fence.  With the multiplies unconditional the
  // unrolled steps of a trip are merged into one scheduling region and the register allocator spills (wgrad256p_kernel: 440
  // spilled registers, 10x slower; here: the 256 x 256 and three-buffer variants); behind the guard each step stays its own region.
  const bool live = p.ksplit > 0;
  TG_STAMP(1);
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
  } else if constexpr (RING) {
    // wait until at most `ahead` whole stages (the youngest ones) are still in flight
    auto wait_ahead = [&](int ahead) {
      if (ahead >= 3) __builtin_amdgcn_s_waitcnt(VMCNT_ONLY(3 * NDMA));
      else if (ahead == 2) __builtin_amdgcn_s_waitcnt(VMCNT_ONLY(2 * NDMA));
      else if (ahead == 1) __builtin_amdgcn_s_waitcnt(VMCNT_ONLY(NDMA));
      else __builtin_amdgcn_s_waitcnt(VMCNT_ONLY(0));
    };
    // stage s: issue s+4 into the buffer stage s-1 has just left (every wave is past the barrier that ended it), multiply s, wait
    // until stage s+1 has landed (everything older than the newest `ahead` stages), raw barrier: s+1 is read only after it
    // one DMA piece of a stage (q < NA: activation piece q, else weight piece q - NA): the interleaved form issues the four
    // pieces of stage s+4 BETWEEN the four MFMA groups of stage s, so that a piece's issue time (60-185 cycles each beside other
    // memory instructions, MI355X_MICROARCH.md) is covered by the eight MFMAs in front of it instead of idling the matrix pipe
    auto issue_piece = [&](int it, char* abase, int q) {
      const int tap = it / nk, c0 = (it - tap * nk) * BKS;
      int tap16, dh, dw;
      if (FORM == FORM_CONV) { tap16 = tap; dh = tap >> 2; dw = tap & 3; }
      else {
        const int a = tap >> 1, c = tap & 1;
        dh = 1 - a; dw = 1 - c;
        tap16 = (1 - ph + 2 * a) * 4 + (1 - pw + 2 * c);
      }
      if (q < NA) {
        const int abit = (FORM == FORM_CONVT) ? dh * 2 + dw : tap;
        const unsigned tapoff = (unsigned)((dh * Wsrc + dw) * ldx2 + c0 * 2);
        const bool ok = (c0 + a_lchunk * 8) < K && ((a_mask[q] >> abit) & 1u);
        dma16<true>(rs_x, abase + (wave + NWV * q) * 1024, ok ? a_off[q] + tapoff : OOB);
      } else {
        const int i = q - NA;
        const unsigned wtap = WT ? (unsigned)(((tap16 * K + c0) * N) * 2) : (unsigned)((tap16 * N * K + c0) * 2);
        const bool ok = w_nok[i] && (c0 + w_k[i]) < K;
        dma16<true>(rs_w, abase + A_BYTES + (WT ? i * (32 * 256) + wave * 1024 : (wave + NWV * i) * 1024), ok ? w_off[i] + wtap : OOB);
      }
    };
    auto stage_il = [&](int it, const char* cur, char* tgt) {
      const bool more = it + 4 < it_hi;
      const char* w_img = cur + A_BYTES;
      int ql = lane;
      asm volatile("" : "+v"(ql));
      u32x4_t wf[4];
#pragma unroll
      for (int i = 0; i < 4; i++)
        wf[i] = WT ? timg_frag(w_img + (wn >> 1) * (32 * 256), (wn & 1) * 64 + i * 16, 0, ql) : ring_frag(w_img, wn * 64 + i * 16, ql);
#pragma unroll
      for (int q = 0; q < MF / 2; q++) {
        const u32x4_t af0 = ring_frag(cur, wm * WM + (2 * q) * 16, ql), af1 = ring_frag(cur, wm * WM + (2 * q + 1) * 16, ql);
        if (more && q < NDMA) issue_piece(it + 4, tgt, q);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; i++) { acc[i][2 * q] = mfma16<T>(wf[i], af0, acc[i][2 * q]); acc[i][2 * q + 1] = mfma16<T>(wf[i], af1, acc[i][2 * q + 1]); }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
      }
      wait_ahead(min(it + 4, it_hi - 1) - (it + 1));
      __builtin_amdgcn_s_barrier();
    };
    auto stage = [&](int it, const char* cur, char* tgt) {
      if constexpr (RING_IL) {
        if (live) stage_il(it, cur, tgt);
        return;
      }
      if (it + 4 < it_hi) issue(it + 4, tgt);
      if (live) compute_ring(cur);
      wait_ahead(min(it + 4, it_hi - 1) - (it + 1));
      __builtin_amdgcn_s_barrier();
    };
    if (it_lo < it_hi) {
      issue(it_lo, lds0);
      if (it_lo + 1 < it_hi) issue(it_lo + 1, lds1);
      if (it_lo + 2 < it_hi) issue(it_lo + 2, lds2);
      if (it_lo + 3 < it_hi) issue(it_lo + 3, lds3);
      wait_ahead(min(it_lo + 3, it_hi - 1) - it_lo);
    }
    __builtin_amdgcn_s_barrier();
    for (int it = it_lo; it < it_hi; it += 5) {    // five stages per trip: buffer roles are compile-time
      stage(it, lds0, lds4);
      if (it + 1 >= it_hi) break;
      stage(it + 1, lds1, lds0);
      if (it + 2 >= it_hi) break;
      stage(it + 2, lds2, lds1);
      if (it + 3 >= it_hi) break;
      stage(it + 3, lds3, lds2);
      if (it + 4 >= it_hi) break;
      stage(it + 4, lds4, lds3);
    }
  } else {
    // step t: issue the DMA of step t+2, run the MFMAs of step t, then wait until only those NDMA newest DMAs are
    // outstanding (=> this wave's pieces of step t+1 have landed) and meet the other waves at a raw barrier
    // (a __syncthreads() here would drain vmcnt to 0).  Step t+1 is read only after that barrier.
    auto step = [&](int it, const char* cur, char* tgt) {
      const bool more = it + 2 < it_hi;
      if (more) issue(it + 2, tgt);
      if (live) compute(cur);
      if (more) __builtin_amdgcn_s_waitcnt(VMCNT_ONLY(NDMA));   // the builtin (not asm) so hipcc's own vmcnt bookkeeping sees it
      else __builtin_amdgcn_s_waitcnt(VMCNT_ONLY(0));
      __builtin_amdgcn_s_barrier();
    };
    if (it_lo < it_hi) issue(it_lo, lds0);
    if (it_lo + 1 < it_hi) {
      issue(it_lo + 1, lds1);
      __builtin_amdgcn_s_waitcnt(VMCNT_ONLY(NDMA));
    } else {
      __builtin_amdgcn_s_waitcnt(VMCNT_ONLY(0));
    }
    __builtin_amdgcn_s_barrier();
    for (int it = it_lo; it < it_hi; it += 3) {    // three steps per trip: buffer roles are compile-time
      step(it, lds0, lds2);
      if (it + 1 >= it_hi) break;
      step(it + 1, lds1, lds0);
      if (it + 2 >= it_hi) break;
      step(it + 2, lds2, lds1);
    }
  }

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
    // of one per column (in-kernel stamps, scripts/stamp_layer.py: the epilogue of UpShuffle_0's input gradient took 8.2 us of a
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

// ---- deferred row sums: ONE launch for every bias gradient of a reverse pass (gct2_rowsum_flush) -----------------------------------
// Block = 32 columns of ONE target x 128 row lanes (the geometry and the summation order of dbpart_reduce_kernel, so a deferred
// bias gradient has the bits of an immediate one); a target's sources are added in recording order: first writer + second writer
// of a concat slice, exactly as "overwrite, then add" does.  adam != null: Keras Adam on the bias right here (the per-layer
// optimizer launches of the fused step then cover the kernels only).
struct RowsumAdam { float* p; float* m; float* v; void* shadow; const float* g_base; int shadow_dtype; float alpha, b1, b2, eps, gmul; };
__global__ __launch_bounds__(1024) void rowsum_flush_kernel(RowsumTable tab, RowsumAdam ad) {
  int ti = 0;
  for (int k = 1; k < tab.ntargets; k++) if ((int)blockIdx.x >= tab.t[k].blk0) ti = k;     // block-uniform
  const RowsumTarget& T = tab.t[ti];
  const int tid = threadIdx.x, cq = tid & 7, rl = tid >> 3;
  const int n = ((int)blockIdx.x - T.blk0) * 32 + cq * 4;
  __shared__ f32x4_t red[16][8];
  f32x4_t total = {0.f, 0.f, 0.f, 0.f};
  for (int si = 0; si < T.nsrc; si++) {
    const RowsumSrc S = T.src[si];
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    if (n < T.ncols)
      for (int r = rl; r < S.rows; r += 128) acc += *reinterpret_cast<const f32x4_t*>(S.part + (size_t)r * S.ld + S.col0 + n);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      float t = acc[k];
      t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
      acc[k] = t;
    }
    __syncthreads();                                        // red is reused per source
    if ((tid & 63) < 8) red[tid >> 6][cq] = acc;
    __syncthreads();
    if (tid < 8) {
      f32x4_t t = red[0][tid];
#pragma unroll
      for (int k = 1; k < 16; k++) t += red[k][tid];
      total = si == 0 ? t : total + t;
    }
  }
  if (tid < 8 && n < T.ncols) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (n + r >= T.ncols) break;
      float* q = T.dst + n + r;
      // no recorded source: the launches reduced their rows themselves (no workspace rows, a full buffer, the direct kernels) and
      // the gradient is in place already
      const float gval = T.nsrc == 0 ? *q : (T.add ? *q + total[r] : total[r]);
      if (T.nsrc) *q = gval;
      if (ad.p && T.adam) {
        const size_t e = (size_t)(q - ad.g_base);
        float pp = ad.p[e], mm = ad.m[e], vv = ad.v[e];
        adam_keras_update(pp, mm, vv, gval * ad.gmul, ad.alpha, ad.b1, 1.f - ad.b1, ad.b2, 1.f - ad.b2, ad.eps);
        ad.p[e] = pp; ad.m[e] = mm; ad.v[e] = vv;
        if (ad.shadow) {
          if (ad.shadow_dtype == GCT2_BF16) reinterpret_cast<__bf16*>(ad.shadow)[e] = from_f32<__bf16>(pp);
          else if (ad.shadow_dtype == GCT2_F16) reinterpret_cast<_Float16*>(ad.shadow)[e] = from_f32<_Float16>(pp);
          else reinterpret_cast<float*>(ad.shadow)[e] = pp;
        }
      }
    }
  }
}

// sums the split-K slabs and applies the epilogue the GEMM kernel skipped.  Work-group = 8 pixels x 128 channels,
// thread = 4 channels of one pixel (split-K layers have few pixels: keep the grid wide), and the bias-gradient
// column sums of the 8 pixels are reduced in LDS into one partial row for dbpart_reduce_kernel.
template <typename T, int EPI>
__global__ __launch_bounds__(256) void tapgemm_finalize_kernel(TapGemmParams p, size_t npix) {
  const int N = p.N;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int n = blockIdx.y * 128 + tx * 4;
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
        for (; s + 8 <= p.ksplit; s += 8) {
          f32x4_t t[8];
#pragma unroll
          for (int u = 0; u < 8; u++) t[u] = *reinterpret_cast<const f32x4_t*>(base + (size_t)(s + u) * sstride);
#pragma unroll
          for (int u = 0; u < 8; u++) v += t[u];
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
    __shared__ f32x4_t red[8][32];
    red[ty][tx] = bsum;
    __syncthreads();
    if (ty == 0 && n < N) {
      f32x4_t t = red[0][tx];
#pragma unroll
      for (int k = 1; k < 8; k++) t += red[k][tx];
      if (p.dbws) *reinterpret_cast<f32x4_t*>(p.dbws + (size_t)blockIdx.x * N + n) = t;   // one partial row per work-group row
      else {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          float* q = db_target(p, n + r);
          if (q) atomicAdd(q, t[r]);
        }
      }
    }
  }
}

template <typename T, int FORM, int BM, int BN, int EPI, int NBUF, int WM = 64>
int launch(const gct2_ctx& c, TapGemmParams p, hipStream_t s) {
  const int M = p.B * p.Hs * p.Ws;
  constexpr int PH = FORM == FORM_CONVT ? 4 : 1;
  const int tiles = ((M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * PH;
  const int ntaps = FORM == FORM_CONV ? 16 : (FORM == FORM_CONVT ? 4 : p.ks * p.ks);
  constexpr int BKS = NBUF >= 5 ? 32 : BK;           // the ring pipeline walks half steps
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
#endif
  // fused bias gradient: partial rows at the tail of the workspace (one per (m-tile, phase), or per finalize work-group)
  const size_t fin_rows = (npix + 7) / 8;
  const size_t dbws_bytes = want_db ? std::max((size_t)p.m_tiles * PH, fin_rows) * p.N * sizeof(float) : 0;
  const bool db_rows = want_db && ws && ws_bytes >= dbws_bytes + 16;
  const size_t slab_room = db_rows ? ws_bytes - dbws_bytes - 16 : ws_bytes;
  // (below ~1.2 work-groups per CU a second k-slice per tile beats the idle half of the chip: U_3 dgrad 102 -> 77 us)
  if (ws && tiles < 300 && niter >= 4 * BK / BKS) {
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
  auto kern = tapgemm_kernel<T, FORM, BM, BN, EPI, NBUF, WM>;
  p.dbws = db_rows ? ws + (ws_bytes - dbws_bytes) / sizeof(float) / 4 * 4 : nullptr;
  // an open row-sum deferral (gct2_rowsum_begin): the partial rows go to the caller's row-sum buffer and stay there until the flush
  const int db_nrows = p.ksplit > 1 ? (int)fin_rows : p.m_tiles * PH;
  float* deferred = (want_db && p.dbws) ? rowsum_alloc(c, (size_t)db_nrows, p.N) : nullptr;
  if (deferred) p.dbws = deferred;
  if (want_db && !p.dbws) zero_overwritten_db(p, s);
  hipLaunchKernelGGL(kern, grid, dim3((BM / WM) * (BN / 64) * 64), 0, s, p);
  if (p.ksplit > 1) {
    hipLaunchKernelGGL((tapgemm_finalize_kernel<T, EPI>), dim3((unsigned)fin_rows, (p.N + 127) / 128), dim3(256), 0, s, p, npix);
  }
  if (EPI == EPI_BIAS_ACT && p.bits && p.wide && p.ksplit == 1) c.relu_bits_done = 1;   // the 16-byte epilogue wrote the ReLU bit plane
  if (deferred) rowsum_record(c, p, deferred, db_nrows);
  else if (p.dbws) {
    const int rows = p.ksplit > 1 ? (int)fin_rows : p.m_tiles * PH;
    hipLaunchKernelGGL(dbpart_reduce_kernel, dim3((p.N + 31) / 32), dim3(1024), 0, s, p.dbws, rows, p);
  }
  return gct2_check_launch("tapgemm_mfma");
}

template <typename T>
int dispatch(const gct2_ctx& c, int form, int epi, const TapGemmParams& p, hipStream_t s) {
  const int g_tapgemm_variant = c.tap_variant;
  // the stride-1 forms (off-by-default model variants): one tile shape; the epilogue is the forward's / the input gradient's
  // (tapgemm_mfma checks the pairing)
  if (form == FORM_S1) return launch<T, FORM_S1, 128, 128, EPI_BIAS_ACT, 2>(c, p, s);
  if (form == FORM_S1T) return launch<T, FORM_S1T, 128, 128, EPI_MASK, 2>(c, p, s);
  // big layers: 256 x 128 tile, 8 waves, 3 LDS buffers; layers with few output pixels keep the 128 x 128 tile
  // (more work-groups + split-K); N <= 64 (UpShuffle_0) uses the 256 x 64 tile
  const int M = p.B * p.Hs * p.Ws;
  const bool big = g_tapgemm_variant == 3;   // measured r01: the 4-wave tile at 2 work-groups per CU is faster (profiles/)
  // automatic choice (per-layer A/B in scripts/bench_layer.py, profiles/r01_layer_variants.txt): the 256 x 128 single-
  // buffer tile moves 25 % fewer L2->LDS bytes per FLOP and wins 5-14 % where it still yields >= 2 work-groups per CU.
  const int tiles256 = ((M + 255) / 256) * ((p.N + 127) / 128) * (form == FORM_CONVT ? 4 : 1);
  const bool auto5 = g_tapgemm_variant == 0 && tiles256 >= 512;
  if ((g_tapgemm_variant == 5 || auto5) && p.N > 64) {   // 256 x 128 tile, 8 waves, one LDS buffer (48 KiB), 2 work-groups per CU
    if (form == FORM_CONV) return epi == EPI_BIAS_ACT ? launch<T, FORM_CONV, 256, 128, EPI_BIAS_ACT, 1>(c, p, s)
                                                      : launch<T, FORM_CONV, 256, 128, EPI_MASK, 1>(c, p, s);
    return epi == EPI_BIAS_ACT ? launch<T, FORM_CONVT, 256, 128, EPI_BIAS_ACT, 1>(c, p, s)
                               : launch<T, FORM_CONVT, 256, 128, EPI_MASK, 1>(c, p, s);
  }
  // 256 x 256 tile, 8 waves of 128 x 64, two 64-KiB LDS buffers, one work-group per CU
  const int tiles6 = ((M + 255) / 256) * ((p.N + 255) / 256) * (form == FORM_CONVT ? 4 : 1);
  const bool auto6 = false && tiles6 >= 192;
  if ((g_tapgemm_variant == 6 || auto6) && p.N >= 256) {
    if (form == FORM_CONV) return epi == EPI_BIAS_ACT ? launch<T, FORM_CONV, 256, 256, EPI_BIAS_ACT, 2, 128>(c, p, s)
                                                      : launch<T, FORM_CONV, 256, 256, EPI_MASK, 2, 128>(c, p, s);
    return epi == EPI_BIAS_ACT ? launch<T, FORM_CONVT, 256, 256, EPI_BIAS_ACT, 2, 128>(c, p, s)
                               : launch<T, FORM_CONVT, 256, 256, EPI_MASK, 2, 128>(c, p, s);
  }
  // 256 x 256 tile with the five-stage ring (four stages in flight, counted vmcnt across raw barriers), one work-group per CU;
  // variant 8: the same with the DMA pieces of stage s+4 issued between the MFMA groups of stage s
  if (g_tapgemm_variant == 7 && p.N >= 256) {
    if (form == FORM_CONV) return epi == EPI_BIAS_ACT ? launch<T, FORM_CONV, 256, 256, EPI_BIAS_ACT, 5, 128>(c, p, s)
                                                      : launch<T, FORM_CONV, 256, 256, EPI_MASK, 5, 128>(c, p, s);
    return epi == EPI_BIAS_ACT ? launch<T, FORM_CONVT, 256, 256, EPI_BIAS_ACT, 5, 128>(c, p, s)
                               : launch<T, FORM_CONVT, 256, 256, EPI_MASK, 5, 128>(c, p, s);
  }
  if (g_tapgemm_variant == 8 && p.N >= 256) {
    if (form == FORM_CONV) return epi == EPI_BIAS_ACT ? launch<T, FORM_CONV, 256, 256, EPI_BIAS_ACT, 6, 128>(c, p, s)
                                                      : launch<T, FORM_CONV, 256, 256, EPI_MASK, 6, 128>(c, p, s);
    return epi == EPI_BIAS_ACT ? launch<T, FORM_CONVT, 256, 256, EPI_BIAS_ACT, 6, 128>(c, p, s)
                               : launch<T, FORM_CONVT, 256, 256, EPI_MASK, 6, 128>(c, p, s);
  }
  if (g_tapgemm_variant == 1) {
    if (form == FORM_CONV) return epi == EPI_BIAS_ACT ? launch<T, FORM_CONV, 128, 128, EPI_BIAS_ACT, 1>(c, p, s)
                                                      : launch<T, FORM_CONV, 128, 128, EPI_MASK, 1>(c, p, s);
    if (p.N > 64) return epi == EPI_BIAS_ACT ? launch<T, FORM_CONVT, 128, 128, EPI_BIAS_ACT, 1>(c, p, s)
                                             : launch<T, FORM_CONVT, 128, 128, EPI_MASK, 1>(c, p, s);
  }
  if (form == FORM_CONV) {
    if (big) return epi == EPI_BIAS_ACT ? launch<T, FORM_CONV, 256, 128, EPI_BIAS_ACT, 3>(c, p, s)
                                        : launch<T, FORM_CONV, 256, 128, EPI_MASK, 3>(c, p, s);
    return epi == EPI_BIAS_ACT ? launch<T, FORM_CONV, 128, 128, EPI_BIAS_ACT, 2>(c, p, s)
                               : launch<T, FORM_CONV, 128, 128, EPI_MASK, 2>(c, p, s);
  }
  const bool narrow = p.N <= 64;
  if (epi == EPI_BIAS_ACT) {
    if (narrow) return launch<T, FORM_CONVT, 256, 64, EPI_BIAS_ACT, 2>(c, p, s);
    return big ? launch<T, FORM_CONVT, 256, 128, EPI_BIAS_ACT, 3>(c, p, s) : launch<T, FORM_CONVT, 128, 128, EPI_BIAS_ACT, 2>(c, p, s);
  }
  if (narrow) return launch<T, FORM_CONVT, 256, 64, EPI_MASK, 2>(c, p, s);
  return big ? launch<T, FORM_CONVT, 256, 128, EPI_MASK, 3>(c, p, s) : launch<T, FORM_CONVT, 128, 128, EPI_MASK, 2>(c, p, s);
}

}  // namespace

float* rowsum_alloc(const gct2_ctx& c, size_t rows, int N) {
  RowsumState& rs = c.rowsum;
  if (!rs.open || !rs.buf) return nullptr;
  const size_t need = (rows * (size_t)N + 3) / 4 * 4;
  // two more targets must fit the table, the rows the buffer; otherwise this launch reduces its rows itself (same result)
  if (rs.used + need > rs.bytes / sizeof(float) || rs.table.ntargets + 2 > ROWSUM_MAX_TARGETS) { rs.overflow = true; return nullptr; }
  float* q = rs.buf + rs.used;
  rs.used += need;
  return q;
}
void rowsum_record(const gct2_ctx& c, const TapGemmParams& p, const float* part, int rows) {
  RowsumTable& tab = c.rowsum.table;
  auto add = [&](float* dst, int col0, int ncols, bool adds) {
    if (!dst || ncols <= 0) return;
    RowsumTarget* t = nullptr;
    for (int k = 0; k < tab.ntargets; k++) if (tab.t[k].dst == dst) t = &tab.t[k];
    if (t && (!adds || t->nsrc >= ROWSUM_MAX_SRC || t->ncols != ncols)) {
      // an overwriting launch supersedes what was recorded for this target (or the record is full: cannot happen for the U-Net,
      // every bias gradient has at most two writers) - start over
      if (!adds) { t->nsrc = 0; t->add = 0; }
      else return;                                           // unreachable by construction; drop rather than corrupt
    }
    if (!t) {
      t = &tab.t[tab.ntargets++];
      t->dst = dst; t->ncols = ncols; t->nsrc = 0; t->adam = 0;
      t->add = adds ? 1 : 0;                                 // first record of the pass adds: keep what dst holds
    }
    t->src[t->nsrc++] = RowsumSrc{part, rows, p.N, col0};
  };
  add(p.db, 0, p.db_split < p.N ? p.db_split : p.N, (p.db_acc & 1) != 0);
  add(p.db2, p.db_split, p.N - p.db_split, (p.db_acc & 2) != 0);
}
int rowsum_flush_launch(const gct2_ctx& c, const gct2_adam_args* adam, const float* g_base, const int64_t* bias_ranges, int nranges,
                        hipStream_t s) {
  RowsumState& rs = c.rowsum;
  RowsumTable& tab = rs.table;
  // the biases the caller wants the optimizer applied to: recorded targets get the flag, the others (their launches reduced their
  // rows themselves: the gradient is in the arena already) join the table without sources
  for (int r = 0; adam && r < nranges; r++) {
    float* dst = const_cast<float*>(g_base) + bias_ranges[2 * r];
    const int ncols = (int)bias_ranges[2 * r + 1];
    RowsumTarget* t = nullptr;
    for (int k = 0; k < tab.ntargets; k++) if (tab.t[k].dst == dst) t = &tab.t[k];
    if (!t) {
      if (tab.ntargets >= ROWSUM_MAX_TARGETS) return gct2_fail(GCT2_EINVAL, "rowsum_flush: more than %d bias targets", ROWSUM_MAX_TARGETS);
      t = &tab.t[tab.ntargets++];
      t->dst = dst; t->ncols = ncols; t->nsrc = 0; t->add = 1;
    } else if (t->ncols != ncols) return gct2_fail(GCT2_EINVAL, "rowsum_flush: bias range %d has %d elements, the recorded gradient %d", r, ncols, t->ncols);
    t->adam = 1;
  }
  int blocks = 0;
  for (int k = 0; k < tab.ntargets; k++) { tab.t[k].blk0 = blocks; blocks += (tab.t[k].ncols + 31) / 32; }
  tab.nblocks = blocks;
  RowsumAdam ad{};
  if (adam) ad = RowsumAdam{adam->p, adam->m, adam->v, adam->shadow, g_base, adam->shadow_dtype, adam->alpha, adam->beta1, adam->beta2, adam->eps, adam->grad_mul};
  if (blocks > 0) hipLaunchKernelGGL(rowsum_flush_kernel, dim3(blocks), dim3(1024), 0, s, tab, ad);
  rs.open = false; rs.used = 0; tab.ntargets = 0; tab.nblocks = 0;
  return gct2_check_launch("rowsum_flush");
}

// true when the MFMA path can take this problem (16-byte aligned rows, whole 8-channel chunks, 31-bit byte offsets)
bool tapgemm_mfma_supported(int dtype, const TapGemmParams& p) {
  if (dtype != GCT2_BF16 && dtype != GCT2_F16) return false;
  if (p.K % 8 || p.N % 8 || p.ldx % 8 || p.ldy % 4) return false;
  if (p.act && p.ldact % 4) return false;
  if ((uintptr_t)p.x % 16 || (uintptr_t)p.w % 16 || (uintptr_t)p.y % 8) return false;
  if (p.act && (uintptr_t)p.act % 8) return false;
  if (p.bias && (uintptr_t)p.bias % 16) return false;
  // buffer descriptors address 2 GiB: source tensor (BIG grid for conv-form) and the 16-tap weight tensor
  const size_t src_bytes = (size_t)p.B * p.Hs * p.Ws * 4 * p.ldx * 2;
  const size_t w_bytes = (size_t)16 * p.K * p.N * 2;
  if (src_bytes >= 0x7ff00000u || w_bytes >= 0x7ff00000u) return false;
  return true;
}

bool halo_convT_wanted(const gct2_ctx& c, int epi, const TapGemmParams& p);      // halo_mfma.hip
int halo_convT(const gct2_ctx& c, int dtype, int epi, TapGemmParams p, hipStream_t s);
bool halo_conv_wanted(const gct2_ctx& c, int epi, const TapGemmParams& p);       // halo_conv_mfma.hip
int halo_conv(const gct2_ctx& c, int dtype, int epi, TapGemmParams p, hipStream_t s);

// the ordered row reduction of the fused bias gradients, for the other translation units that leave partial rows
int tapgemm_dbpart_reduce(const float* part, int rows, const TapGemmParams& p, hipStream_t s) {
  hipLaunchKernelGGL(dbpart_reduce_kernel, dim3((p.N + 31) / 32), dim3(1024), 0, s, part, rows, p);
  return gct2_check_launch("dbpart_reduce");
}

int tapgemm_mfma(const gct2_ctx& c, int dtype, int form, int epi, const TapGemmParams& p, hipStream_t s) {
  if (form == FORM_S1 || form == FORM_S1T) {
    if (epi != (form == FORM_S1 ? EPI_BIAS_ACT : EPI_MASK) || p.ks < 1 || p.ks > 5 || !(p.ks & 1))
      return gct2_fail(GCT2_EINVAL, "tapgemm_mfma: stride-1 form with kernel size %d / epilogue %d", p.ks, epi);
    if (dtype == GCT2_BF16) return dispatch<__bf16>(c, form, epi, p, s);
    return dispatch<_Float16>(c, form, epi, p, s);
  }
  if (form == FORM_CONVT && c.tap_variant == 0 && halo_convT_wanted(c, epi, p)) return halo_convT(c, dtype, epi, p, s);
  if (form == FORM_CONV && c.tap_variant == 0 && halo_conv_wanted(c, epi, p)) return halo_conv(c, dtype, epi, p, s);
  if (dtype == GCT2_BF16) return dispatch<__bf16>(c, form, epi, p, s);
  return dispatch<_Float16>(c, form, epi, p, s);
}
