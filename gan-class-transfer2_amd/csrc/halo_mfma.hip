// Halo-tile variant of the FORM_CONVT tap GEMM (Conv2DTranspose forward, train.py:148-153, and the input gradient of
// Conv2D, train.py:161-166): ONE staged source patch serves
// all 4 output-parity phases and all 4 taps of each.
//
// tapgemm_kernel<FORM_CONVT> stages a 256-pixel x 64-channel source tile per (phase, tap, k-chunk): 16 tile loads per k-chunk
// for a 256-pixel patch, although the 16 (phase, tap) pairs only touch the 3 x 3 neighbourhood of every pixel.  For layers
// with few output channels (UpShuffle_0: N = 64) those loads are the bound (measured: full 204 us, MFMA-only 144 us,
// DMA-only 169 us).  Here a work-group owns a 16 x 16 patch of the SMALL grid: the 18 x 18 halo (324 pixel rows x 128 B) is
// staged ONCE per 64-channel k-chunk and every (phase, tap) reads its A fragments from it at a shifted row offset; only the
// weights (4 phases x 64 n x 64 k = 32 KiB) are staged per tap round.  L2->LDS traffic per k-chunk: 41 + 4 x 32 = 169 KiB
// instead of 640 KiB.
//
// 8 waves: wave = (phase, half of the patch) -> 128 pixels x 64 channels (8 x 4 MFMA tiles, 128 accumulator registers).
// A fragment = 16 pixels of one patch row = 16 CONSECUTIVE halo rows starting at an arbitrary row (the tap shift), so the halo
// image uses a swizzle that is conflict-free for ds_read_b128 at EVERY start row: chunk ^ (4*bit1(row) + 2*bit2(row))
// (exhaustive check over the lane groups of MI355X_MICROARCH.md, LDS table; the N-image swizzle needs aligned starts).
// Needs Hs, Ws multiples of 16.  Epilogues: bias + ReLU (forward) or mask / accumulate / bias-gradient rows (Conv2D dgrad).
#include "gct2_common.h"
#include <algorithm>
#include <type_traits>

namespace {

constexpr unsigned OOB = 0x80000000u;
typedef __attribute__((address_space(3))) void lds_void_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)OOB, 0x00020000);
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_piece, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)lds_piece, 16, (int)voff, 0, 0, 0);
}
// the same with a scalar byte offset (the k-chunk / tap part of the address: wave-uniform, outside the bounds check's per-lane offset)
__device__ __forceinline__ void dma16s(__amdgpu_buffer_rsrc_t rsrc, char* lds_piece, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)lds_piece, 16, (int)voff, (int)soff, 0, 0);
}
__device__ __forceinline__ int halo_swz(int row) { return (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1); }
// Weight image: fragment i of a wave holds the output channels 32 (i>>1) + 8 g + 4 (i&1) + r in its lane group g, so that the
// fragments 2k and 2k+1 together give every lane EIGHT consecutive channels of its pixel -> 16-byte stores / mask loads in the
// epilogue.  Row (= channel) of lane q in fragment i, and the swizzle that keeps those gathered rows conflict-free for
// ds_read_b128 (exhaustive search over XOR-linear swizzles against the lane groups of MI355X_MICROARCH.md, LDS table):
__device__ __forceinline__ int w_row(int i, int q) { return 32 * (i >> 1) + 8 * (q >> 2) + 4 * (i & 1) + (q & 3); }
__device__ __forceinline__ int w_swz(int n) { return (((n >> 3) & 1) << 1) | (((n >> 1) & 1) << 2); }

constexpr int HP = 18;                        // halo pitch (pixels per halo row)
constexpr int HPIECES = 41;                   // 1-KiB pieces (8 pixel rows each) covering the 324 halo pixels
constexpr int HALO_BYTES = HPIECES * 1024;
constexpr int WB_BYTES = 4 * 64 * 128;        // 4 phases x 64 n-rows x 128 B

#ifdef GCT2_STAMP
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define STAMP(k) st[k] = stamp()
#else
#define STAMP(k)
#endif
// The DMA pieces of the next round (and of the next halo) are issued BETWEEN the MFMA groups of the current round instead of in
// front of them: a piece costs 60-185 cycles of issue time (MI355X_MICROARCH.md), during which the wave issues nothing else, and the
// two waves of a SIMD reach that block together (r03: -3..-6 % against the r02 order, which is gone)
template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void halo_convT_kernel(TapGemmParams p) {
#ifdef GCT2_STAMP
  unsigned long long st[6];
  STAMP(0);
#endif
  // ONE array, the halo images first: every fragment address of the lean rounds is a per-lane register + a 16-bit immediate
  // (EPI_HEAD: 14 KiB more - the rest of the CU's 160 KiB - for the operand images of the Dense kernel, built while the first DMA is in flight)
  constexpr int HEAD_CT_OFF = 2 * HALO_BYTES + 2 * WB_BYTES, HEAD_CT_BYTES = 14 * 1024;
  __shared__ __attribute__((aligned(16))) char lds_all[2 * HALO_BYTES + 2 * WB_BYTES + (EPI == EPI_HEAD ? HEAD_CT_BYTES : 0)];
  char* const halo0 = lds_all;
  char* const halo1 = lds_all + HALO_BYTES;
  char* const wb0 = lds_all + 2 * HALO_BYTES;
  char* const wb1 = lds_all + 2 * HALO_BYTES + WB_BYTES;

  GCT2_CLOCK_DECL;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int phase = wave >> 1, mhalf = wave & 1;
  const int ph = phase >> 1, pw = phase & 1;
  const int g = lane >> 4, q = lane & 15;
  const int Hs = p.Hs, Ws = p.Ws, K = p.K, N = p.N;
  const int tx_n = Ws >> 4, ty_n = Hs >> 4;
  int m_tile, n_tile;
  if (!xcd_tile((int)blockIdx.x, p.m_tiles, p.n_tiles, p.xcd_chunk, m_tile, n_tile)) return;
  const int tx = m_tile % tx_n, tq = m_tile / tx_n, ty = tq % ty_n, b = tq / ty_n;
  const int sh0 = ty * 16, sw0 = tx * 16, n0 = n_tile * 64;
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x), rs_w = make_rsrc(p.w);
  const int ldx2 = p.ldx * 2;

  // ---- per-lane DMA descriptors ------------------------------------------------------------------------------------
  // halo: piece pi = wave + 8 i (i < 6, pi < 41) = halo rows 8 pi .. 8 pi + 7; lane -> row 8 pi + (lane>>3), physical chunk lane&7
  // (validity folded in: OOB lanes read zeros, and stay OOB whatever the scalar offset of the k-chunk adds)
  unsigned h_voff[6];
  int h_lchunk[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    const int pi = wave + 8 * i;
    const int row = 8 * pi + (lane >> 3);
    const int hy = row / HP, hx = row - hy * HP;
    const int lchunk = (lane & 7) ^ halo_swz(hx);
    h_lchunk[i] = lchunk;
    h_voff[i] = OOB;
    if (pi < HPIECES && row < HP * HP) {
      const int y = sh0 - 1 + hy, x = sw0 - 1 + hx;
      if ((unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws) h_voff[i] = (unsigned)(((b * Hs + y) * Ws + x) * ldx2 + lchunk * 16);
    }
  }
  // weights: piece (phase i, n-block wave): lane -> n = 8 wave + (lane>>3), physical chunk lane&7 (w_swz)
  const int w_n = 8 * wave + (lane >> 3);
  const int w_lchunk = (lane & 7) ^ w_swz(w_n);
  const unsigned w_voff = (n0 + w_n) < N ? (unsigned)(((n0 + w_n) * K + w_lchunk * 8) * 2) : OOB;
  const unsigned NK2 = (unsigned)(N * K * 2);

  const int nk = (K + 63) / 64;
  const int nround = 4 * nk;                                   // (k-chunk, tap round (a, c))

  auto issue_halo = [&](int kc, char* hbuf) {
    const int c0 = kc * 64;
#pragma unroll
    for (int i = 0; i < 6; i++) {
      const int pi = wave + 8 * i;
      if (pi < HPIECES)                                        // wave-uniform
        dma16s(rs_x, hbuf + pi * 1024, (c0 + h_lchunk[i] * 8) < K ? h_voff[i] : OOB, (unsigned)(c0 * 2));
    }
  };
  auto issue_w = [&](int round, char* wbuf) {
    const int kc = round >> 2, a = (round >> 1) & 1, c = round & 1;
    const int c0 = kc * 64;
    const unsigned voff = (c0 + w_lchunk * 8) < K ? w_voff : OOB;
#pragma unroll
    for (int i = 0; i < 4; i++) {                              // phase i = (i>>1, i&1): kernel tap (1 - ph + 2a, 1 - pw + 2c)
      const int tap16 = (1 - (i >> 1) + 2 * a) * 4 + (1 - (i & 1) + 2 * c);
      dma16s(rs_w, wbuf + i * 8192 + wave * 1024, voff, (unsigned)tap16 * NK2 + (unsigned)(c0 * 2));
    }
  };

  f32x4_t acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 8; j++) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // ---- the lean round (IL) --------------------------------------------------------------------------------------------
  // Fragment addresses: the halo swizzle is keyed on the halo COLUMN hx = pw - c + 1 + q, which neither the tap row a nor the
  // pixel row j changes, and the weight swizzle on bits of the channel row that the fragment index i does not touch.  So a lane
  // needs FOUR halo addresses (tap column c x k-half kk) and TWO weight addresses for the whole K loop, and buffer, tap row, pixel
  // row and fragment index are immediates of the ds_read (halo images first in LDS: at most 41984 + 8 * 2304 < 65536).  The r02
  // round recomputed every address (121 vector instructions per 64 MFMAs, PMC: profiles/r03_pmc_sq_counters.txt).
  // DMA offsets: per-lane offsets stay loop-invariant, the k-chunk and the tap go into the scalar offset of the instruction.
  int fa[2][2], fw[2];
#pragma unroll
  for (int c = 0; c < 2; c++) {
    const int hx = pw - c + 1 + q;
#pragma unroll
    for (int kk = 0; kk < 2; kk++) fa[c][kk] = (((mhalf * 8 + ph) * HP + hx) << 7) + (((4 * kk + g) ^ halo_swz(hx)) << 4);
  }
#pragma unroll
  for (int kk = 0; kk < 2; kk++) {
    const int n = w_row(0, q);
    fw[kk] = 2 * HALO_BYTES + phase * 8192 + n * 128 + (((4 * kk + g) ^ w_swz(n)) << 4);
  }
  const int kfull4 = 4 * (K / 64);                             // rounds of FULL 64-channel chunks
  // idx = round & 7 (literal at every call site: tap and buffer roles fold into immediates); hcur/hnext/wcur/wnext = byte offsets
  // of the buffers inside their group.  FAST: every piece of the next round exists and lies in a full chunk - no tail logic.
  auto lean_round = [&](auto fast_c, int round, int idx, int hcur, int hnext, int wcur, int wnext) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(fast_c)::value;
    const int a = (idx >> 1) & 1, c = idx & 1;
    const int idn = (idx + 1) & 7, an = (idn >> 1) & 1, cn = idn & 1;
    const bool more = FAST || round + 1 < nround;
    const bool halo_due = (idx & 3) == 3 && more;
    const int c0n = ((round + 1) >> 2) * 64;
    const unsigned s_k = (unsigned)(c0n * 2);
    const unsigned wv = FAST ? w_voff : ((c0n + w_lchunk * 8) < K ? w_voff : OOB);
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      u32x4_t wf[4];
#pragma unroll
      for (int i = 0; i < 4; i++) wf[i] = lds_read128(lds_all, fw[kk] + wcur + (32 * (i >> 1) + 4 * (i & 1)) * 128);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const u32x4_t af = lds_read128(lds_all, fa[c][kk] + hcur + (1 - a + j) * HP * 128);
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i][j] = mfma16<T>(wf[i], af, acc[i][j]);
        if (kk == 0 && (j & 1) && more) {                      // weight piece of phase i for the next round
          const int i = j >> 1;
          const int tap16 = (1 - (i >> 1) + 2 * an) * 4 + (1 - (i & 1) + 2 * cn);
          dma16s(rs_w, lds_all + 2 * HALO_BYTES + wnext + i * 8192 + wave * 1024, wv, (unsigned)tap16 * NK2 + s_k);
        }
        if (kk == 1 && j < 6 && halo_due) {                    // halo piece j of the next k-chunk
          const int pi = wave + 8 * j;
          if (pi < HPIECES)
            dma16s(rs_x, lds_all + hnext + pi * 1024, FAST ? h_voff[j] : ((c0n + h_lchunk[j] * 8) < K ? h_voff[j] : OOB), s_k);
        }
      }
    }
  };
  // ---- main loop: 8 rounds (two k-chunks) per trip so that every buffer role is a compile-time constant --------------
  // EPI_HEAD: the operand images of the Dense kernel for the epilogue's matrix-core contractions (see there), written behind the K loop's
  // buffers while the first halo / weight pieces are in flight: their global loads (67 x 3 floats, L2-resident) hide under that DMA
  auto head_term_bits = [](float v, int t) -> uint32_t {  // t-th term of v = hi + mid + lo in the storage type
    T a = from_f32<T>(v);
    if (t == 0) return (uint32_t)__builtin_bit_cast(uint16_t, a);
    float r = v - to_f32<T>(a);
    T b2 = from_f32<T>(r);
    if (t == 1) return (uint32_t)__builtin_bit_cast(uint16_t, b2);
    r = r - to_f32<T>(b2);
    return (uint32_t)__builtin_bit_cast(uint16_t, from_f32<T>(r));
  };
  GCT2_CLOCK_BEGIN;
  issue_halo(0, halo0);
  issue_w(0, wb0);
  if constexpr (EPI == EPI_HEAD) {
    const HeadFuse& hd = p.head;
    const int Cout = hd.Cout;
    for (int e = tid; e < 14 * 64; e += 512) {
      const int ent = e >> 6, L = e & 63, r16 = L & 15, kg = L >> 4;
      uint32_t v[8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        float w = 0.f;
        int t;
        if (ent < 6) {                                      // (F) entry (t, kk): A[row o = r16][k = channel 32 kk + 8 kg + i]
          t = ent >> 1;
          const int c = 32 * (ent & 1) + 8 * kg + i;
          if (r16 < Cout) w = hd.w[c * Cout + r16];
        } else {                                            // (G) entry (f, t): A[row r16 -> channel w_row(f, r16)][k = output i], lane group 0 only
          const int f = (ent - 6) >> 1;
          t = (ent - 6) & 1;
          if (kg == 0 && i < Cout) w = hd.w[w_row(f, r16) * Cout + i];
        }
        v[i] = head_term_bits(w, t);
      }
      *reinterpret_cast<u32x4_t*>(lds_all + HEAD_CT_OFF + e * 16) = u32x4_t{v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16)};
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#define GCT2_LEAN_ROUND(FAST, R, IDX, HCUR, HNEXT, WCUR, WNEXT)                              \
  {                                                                                          \
    lean_round(std::integral_constant<bool, FAST>{}, (R), IDX, HCUR, HNEXT, WCUR, WNEXT);    \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                         \
    __syncthreads();                                                                         \
    if (!FAST && (R) + 1 >= nround) break;                                                   \
  }
#define GCT2_LEAN_TRIP(FAST)                                                                 \
  GCT2_LEAN_ROUND(FAST, r + 0, 0, 0, HALO_BYTES, 0, WB_BYTES)                                \
  GCT2_LEAN_ROUND(FAST, r + 1, 1, 0, HALO_BYTES, WB_BYTES, 0)                                \
  GCT2_LEAN_ROUND(FAST, r + 2, 2, 0, HALO_BYTES, 0, WB_BYTES)                                \
  GCT2_LEAN_ROUND(FAST, r + 3, 3, 0, HALO_BYTES, WB_BYTES, 0)                                \
  GCT2_LEAN_ROUND(FAST, r + 4, 4, HALO_BYTES, 0, 0, WB_BYTES)                                \
  GCT2_LEAN_ROUND(FAST, r + 5, 5, HALO_BYTES, 0, WB_BYTES, 0)                                \
  GCT2_LEAN_ROUND(FAST, r + 6, 6, HALO_BYTES, 0, 0, WB_BYTES)                                \
  GCT2_LEAN_ROUND(FAST, r + 7, 7, HALO_BYTES, 0, WB_BYTES, 0)
  {
    int r = 0;
    for (; r + 8 < kfull4; r += 8) { GCT2_LEAN_TRIP(true) }   // every round of the trip issues a full next round
    for (; r < nround; r += 8) { GCT2_LEAN_TRIP(false) }      // the last trips: runtime tail logic, same buffer roles
  }
#undef GCT2_LEAN_TRIP
#undef GCT2_LEAN_ROUND
#ifdef GCT2_STAMP
  GCT2_CLOCK_END(p.clock ? p.stamps : nullptr, 8, wave, lane);
#endif

  // ---- epilogue: lane holds out[pixel (row mhalf*8 + j, col q)][n = n0 + 32 (i>>1) + 8 g + 4 (i&1) + r], phase (ph, pw) ----
  // EPI_BIAS_ACT: bias + ReLU (Conv2DTranspose forward).  EPI_MASK: ReLU mask of the tensor the gradient belongs to, optional
  // accumulation into the skip slice, column sums for the fused bias gradient (Conv2D input gradient), as in tapgemm_kernel.
  T* __restrict__ yout = reinterpret_cast<T*>(p.y);
  const T* __restrict__ actp = reinterpret_cast<const T*>(p.act);
  int elane = lane;
  asm volatile("" : "+v"(elane));                              // keeps the output addresses out of the K loop
  const int eq = elane & 15, eg = elane >> 4;
  if constexpr (EPI == EPI_HEAD) {
    STAMP(1);
    // ---- UpShuffle_0 forward + Dense(3) head + fp32 MSE + both gradients (train.py:188, 198-202, 262-272) ------------------
    // A wave holds ALL N = 64 channels of its 128 pixels (the 4 lane groups of a pixel column q carry 16 channels each), so the
    // head runs where the activations are produced: y = relu(acc + bias) rounded to the storage type (the value the unfused path
    // would have written to R_0), pred = [y, image] Wd + bd, d = pred - target, the loss, and dR_0 = (y > 0) * dpred Wd^T - the
    // only tensor this epilogue stores.  Dense kernel / bias gradients, the loss and UpShuffle_0's bias gradient leave as ONE
    // partial row per work-group (HEAD_ROW floats) for the ordered finish kernel of the head.
    //
    // r04: the three small contractions run on the MATRIX CORES (r03: ~290 vector instructions per patch row and wave, 16.3 us of a
    // 40.6-us work-group life at one work-group per CU - profiles/r04_kernel_clock.txt).  With the fp32 operands split into 2-3
    // terms of the storage type (hi + mid + lo: 24 bits) the results are fp32-accurate:
    //  (F) pred^T[o][px]  = Wd^T[o][c] . Y[c][px]     : B operand = the lane's OWN packed activations (k = channel 32 ip + 8 eg + i, column = pixel q)
    //                                                    -> rows o = 0..2 land in lane group 0 of pixel q, where d and dpred are formed
    //  (G) dy^T[c][px]    = Wd[c][o] . dpred^T[o][px]  : A rows in the permuted channel order w_row(), so that fragments 2 ip / 2 ip + 1 give every
    //                                                    lane the 8 channels it stores (the layout of the accumulators)
    //  (W) dW^T[o][c]    += dpred^T[o][px] . Y[px][c]  : contraction over PIXELS: Y is parked in LDS as a [pixel][64 channels] image (swizzled 32-byte
    //                                                    chunks) and read back TRANSPOSED (ds_read_b64_tr_b16), dpred^T from a small per-wave table
    // Everything is wave-private between the two barriers (a wave's DS instructions execute in order), no barrier inside the row loop.
    const HeadFuse& hd = p.head;
    const int Cout = hd.Cout;
    constexpr int YP_WAVE = 64 * 128;                       // park image of HALF a wave's pixels: 64 pixel rows x 128 B
    constexpr int CT_OFF = HEAD_CT_OFF;                     // operand images of the Dense kernel: 6 (F) + 8 (G) entries x 64 lanes x 16 B (built before the K loop)
    constexpr int DT_OFF = 8 * YP_WAVE;                     // dpred^T terms: [wave][2 terms][3 outputs][64 pixels] x 2 B
    constexpr int DT_WAVE = 2 * 3 * 64 * 2;
    constexpr int PC_OFF = DT_OFF + 8 * DT_WAVE;            // per-pixel constants of lane group 0: [3 image channels][4] Dense rows, then the Dense bias [4]
    constexpr int LB_OFF = PC_OFF + 64;                     // the layer's own bias [64] (LDS reads per row instead of global loads: those would sit in
    static_assert(LB_OFF + 256 <= HEAD_CT_OFF, "epilogue LDS map");                      // vmcnt behind the previous row's stores and wait for them)
    auto fsw = [](int k) { return ((k >> 1) & 1) | (((k >> 3) & 1) << 1); };      // chunk swizzle of the park image (conflict-free transposed reads)
    auto term_bits = head_term_bits;
    const T* __restrict__ x2 = reinterpret_cast<const T*>(hd.x2);
    const int nimg = (x2 != nullptr) ? min(hd.Cin - 64, 3) : 0;
    if (tid < 16) {                                          // lane group 0's per-pixel constants (read back per row: registers are scarce here)
      const int c = tid >> 2, o = tid & 3;
      float v = 0.f;
      if (o < Cout) v = c < 3 ? (c < nimg ? hd.w[(64 + c) * Cout + o] : 0.f) : (hd.b ? hd.b[o] : 0.f);
      reinterpret_cast<float*>(lds_all + PC_OFF)[tid] = v;
    }
    if (tid >= 64 && tid < 128) reinterpret_cast<float*>(lds_all + LB_OFF)[tid - 64] = p.bias ? p.bias[tid - 64] : 0.f;
    __syncthreads();
    STAMP(2);
    const float gscale = (hd.loss_scale ? *hd.loss_scale : 1.f) * 2.0f / hd.count;
    float bacc[16];                                        // column sums of dy (UpShuffle_0's bias gradient), this lane's 16 channels
    float wimg[3][3], dbd[3] = {0.f, 0.f, 0.f}, lacc = 0.f; // lane group 0: Dense kernel gradient of the image channels, Dense bias gradient, loss
#pragma unroll
    for (int c = 0; c < 16; c++) bacc[c] = 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) wimg[c][0] = wimg[c][1] = wimg[c][2] = 0.f;
    f32x4_t accW[4];                                       // (W): dW^T[o = 4 g + r][c = 16 f + (lane & 15)] over this wave's 128 pixels
#pragma unroll
    for (int f = 0; f < 4; f++) accW[f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    char* const ypark = lds_all + wave * YP_WAVE;
    char* const dtab = lds_all + DT_OFF + wave * DT_WAVE;
    const size_t pix0 = ((size_t)b * (2 * Hs) + 2 * (sh0 + mhalf * 8) + ph) * (2 * Ws) + 2 * (sw0 + eq) + pw;   // output pixel of patch row 0
    const size_t pix_step = (size_t)4 * Ws;                                                                   // ... two output rows further per patch row
    auto row_pix = [&](int j) { return pix0 + (size_t)j * pix_step; };
    // the per-pixel inputs of row j + 1 (target, packed image; lane group 0 only) are loaded while row j is processed
    float tg_n[3] = {0.f, 0.f, 0.f};
    u32x2_t im_n = {0u, 0u};
    auto prefetch = [&](int j) {                           // (every lane group loads: lane-divergent loads would hide the issue order from
      const size_t px = row_pix(j);                          // hipcc's vmcnt bookkeeping and every row would wait for the previous row's stores)
#pragma unroll
      for (int o = 0; o < 3; o++) tg_n[o] = o < Cout ? hd.target[px * Cout + o] : 0.f;
      if (nimg > 0) im_n = *reinterpret_cast<const u32x2_t*>(x2 + px * hd.ldx2);
    };
    prefetch(0);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      int ct = CT_OFF + elane * 16;                        // opaque per row: keeps the 14 operand fragments out of the registers between rows
      asm volatile("" : "+v"(ct));
      const size_t opix = row_pix(j);
      const float tg[3] = {tg_n[0], tg_n[1], tg_n[2]};
      const u32x2_t v2 = im_n;
      if (j < 7) prefetch(j + 1);
      const int krow = (j & 3) * 16 + eq;                  // this pixel's row in the park image of the current half
      // y = relu(acc + bias), rounded to the storage type: what R_0 would hold; parked for (W), used right away for (F)
      u32x4_t yv[2];
#pragma unroll
      for (int ip = 0; ip < 2; ip++) {
        const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(lds_all + LB_OFF + (32 * ip + 8 * eg) * 4);
        const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(lds_all + LB_OFF + (32 * ip + 8 * eg + 4) * 4);
        const f32x4_t v0 = acc[2 * ip][j] + b0, v1 = acc[2 * ip + 1][j] + b1;
        yv[ip] = u32x4_t{pack2<T>(fmaxf(v0[0], 0.f), fmaxf(v0[1], 0.f)), pack2<T>(fmaxf(v0[2], 0.f), fmaxf(v0[3], 0.f)),
                         pack2<T>(fmaxf(v1[0], 0.f), fmaxf(v1[1], 0.f)), pack2<T>(fmaxf(v1[2], 0.f), fmaxf(v1[3], 0.f))};
        *reinterpret_cast<u32x4_t*>(ypark + krow * 128 + (((2 * ip + (eg >> 1)) ^ fsw(krow)) << 5) + (eg & 1) * 16) = yv[ip];
      }
      // (F) pred^T = Wd^T . Y: three terms of the Dense kernel x two 32-channel blocks
      f32x4_t accP = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 3; t++)
#pragma unroll
        for (int kk = 0; kk < 2; kk++) accP = mfma16<T>(lds_read128(lds_all, ct + (t * 2 + kk) * 1024), yv[kk], accP);
      // lane group 0: prediction, loss, dpred of pixel q (rows o = 0..2 of the result)
      float dsc[3] = {0.f, 0.f, 0.f};
      if (eg == 0) {
        const float im[3] = {unpack_lo<T>(v2[0]), unpack_hi<T>(v2[0]), unpack_lo<T>(v2[1])};
        f32x4_t pc[4];                                       // rows 0..2: Dense rows of the image channels, row 3: Dense bias
#pragma unroll
        for (int c = 0; c < 4; c++) pc[c] = *reinterpret_cast<const f32x4_t*>(lds_all + PC_OFF + 16 * c);
#pragma unroll
        for (int o = 0; o < 3; o++) {
          float s = accP[o];
#pragma unroll
          for (int c = 0; c < 3; c++) s = fmaf(im[c], pc[c][o], s);
          const float pr = keras_f16_point<T>(s + pc[3][o]);
          const float d = o < Cout ? pr - tg[o] : 0.f;
          if (o < Cout) {
            if (hd.pred) hd.pred[opix * Cout + o] = pr;
            lacc = fmaf(d, d, lacc);
          }
          dsc[o] = keras_f16_point<T>(d * gscale);
          dbd[o] += dsc[o];
#pragma unroll
          for (int c = 0; c < 3; c++) wimg[c][o] = fmaf(im[c], dsc[o], wimg[c][o]);
        }
      }
      // dpred^T as two terms: the B operand of (G) (k = output, lane group 0) and the per-wave table for (W)
      u32x4_t dB[2];
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const uint32_t d0 = term_bits(dsc[0], t), d1 = term_bits(dsc[1], t), d2 = term_bits(dsc[2], t);
        dB[t] = u32x4_t{d0 | (d1 << 16), d2, 0u, 0u};       // (lane groups 1..3: dsc = 0 -> zero operand)
        if (eg == 0) {
          uint16_t* dt = reinterpret_cast<uint16_t*>(dtab) + (t * 3) * 64 + krow;
          dt[0] = (uint16_t)d0; dt[64] = (uint16_t)d1; dt[128] = (uint16_t)d2;
        }
      }
      // (G) dy^T = Wd . dpred^T (hi.hi + hi.lo + lo.hi), masked by y > 0, summed into the bias gradient, stored as dR_0
#pragma unroll
      for (int ip = 0; ip < 2; ip++) {
        f32x4_t gq[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int f = 2 * ip + h;
          const u32x4_t wh = lds_read128(lds_all, ct + (6 + f * 2) * 1024), wl = lds_read128(lds_all, ct + (7 + f * 2) * 1024);
          f32x4_t a = mfma16<T>(wh, dB[0], f32x4_t{0.f, 0.f, 0.f, 0.f});
          a = mfma16<T>(wh, dB[1], a);
          gq[h] = mfma16<T>(wl, dB[0], a);
        }
        apply_relu_bits8(relu_bits8<T>(yv[ip]), gq[0], gq[1]);
#pragma unroll
        for (int k = 0; k < 4; k++) { bacc[8 * ip + k] += gq[0][k]; bacc[8 * ip + 4 + k] += gq[1][k]; }
        const u32x4_t o = {pack2<T>(gq[0][0], gq[0][1]), pack2<T>(gq[0][2], gq[0][3]), pack2<T>(gq[1][0], gq[1][1]), pack2<T>(gq[1][2], gq[1][3])};
        *reinterpret_cast<u32x4_t*>(yout + opix * p.ldy + 32 * ip + 8 * eg) = o;
      }
      if ((j & 3) == 3) {
        // (W) for the four rows just parked: 64 pixels = two 32-pixel steps x two terms of dpred x four 16-channel fragments.
        // (compiler fences: the park image and the table are written and read through differently typed pointers; the hardware
        // executes a wave's DS instructions in order, so no wait is needed between the writes above and the reads below)
        asm volatile("" ::: "memory");
        const int wo = elane & 15, kg = elane >> 4;
        const int g4 = elane >> 4, q4 = (elane >> 2) & 3, pq = elane & 3;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
          u32x4_t da[2];
#pragma unroll
          for (int t = 0; t < 2; t++) {
            da[t] = lds_read128(dtab, ((t * 3 + (wo < 3 ? wo : 0)) * 64 + 32 * ks + 8 * kg) * 2);
            if (wo >= 3) da[t] = u32x4_t{0u, 0u, 0u, 0u};
          }
          const int k0 = 32 * ks + 8 * g4 + q4;
#pragma unroll
          for (int f = 0; f < 4; f++) {
            const int o0 = k0 * 128 + ((f ^ fsw(k0)) << 5) + pq * 8;
            const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(ypark + o0));
            const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(ypark + o0 + 4 * 128));
            const u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
            const u32x4_t yb = {l2[0], l2[1], h2[0], h2[1]};
            accW[f] = mfma16<T>(da[0], yb, accW[f]);
            accW[f] = mfma16<T>(da[1], yb, accW[f]);
          }
        }
        asm volatile("" ::: "memory");
      }
    }
    STAMP(3);
    __syncthreads();                                       // every wave is done with its park image and tables
    STAMP(4);
    float* red = reinterpret_cast<float*>(lds_all);        // [8][HEAD_ROW]
    for (int i = tid; i < 8 * HEAD_ROW; i += 512) red[i] = 0.f;
    __syncthreads();
    float* rw = red + wave * HEAD_ROW;
    if (eg == 0) {                                         // (W): rows o = r of lane group 0, column = channel 16 f + (lane & 15)
#pragma unroll
      for (int f = 0; f < 4; f++)
#pragma unroll
        for (int o = 0; o < 3; o++)
          if (o < Cout) rw[(16 * f + eq) * Cout + o] = accW[f][o];
    }
#pragma unroll
    for (int c = 0; c < 16; c++) {
      const float tb = row16_sum(bacc[c]);
      if (eq == 0) rw[224 + 32 * (c >> 3) + 8 * eg + (c & 7)] = tb;
    }
#pragma unroll
    for (int o = 0; o < 3; o++) {
      const float td = row16_sum(dbd[o]);
      if (elane == 0 && o < Cout) rw[216 + o] = td;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const float tw = row16_sum(wimg[c][o]);
        if (elane == 0 && o < Cout && c < nimg) rw[(64 + c) * Cout + o] = tw;
      }
    }
    {
      const float t = row16_sum(lacc);
      if (elane == 0) rw[219] = t;
    }
    __syncthreads();
    if (tid < HEAD_ROW) {
      float t = red[tid];
#pragma unroll
      for (int k = 1; k < 8; k++) t += red[k * HEAD_ROW + tid];
      hd.part[(size_t)m_tile * HEAD_ROW + tid] = t;
    }
#ifdef GCT2_STAMP
    STAMP(5);
    if (p.stamps && (tid & 63) == 0) {
      unsigned long long* o = p.stamps + ((size_t)m_tile * 8 + wave) * 8;
      for (int k = 0; k < 6; k++) o[k] = st[k];
    }
    GCT2_CLOCK_EXIT(p.clock ? p.stamps : nullptr, 8, wave, lane);
#endif
    return;
  }
  f32x4_t bsum[4];
#pragma unroll
  for (int i = 0; i < 4; i++) bsum[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // the ReLU-mask words of patch row j + 1 are loaded while row j is processed (as in tapgemm_kernel: one exposed load latency
  // per tile instead of one per row)
  auto out_pixel = [&](int j) {
    const int sh = sh0 + mhalf * 8 + j, sw = sw0 + eq;
    return ((size_t)b * (2 * Hs) + 2 * sh + ph) * (2 * Ws) + 2 * sw + pw;
  };
  u32x4_t mk[2] = {u32x4_t{0u, 0u, 0u, 0u}, u32x4_t{0u, 0u, 0u, 0u}};
  auto load_masks = [&](int j, u32x4_t* dst) {
    if (EPI == EPI_MASK && actp) {
      const size_t opix = out_pixel(j);
#pragma unroll
      for (int ip = 0; ip < 2; ip++) {
        const int n = n0 + 32 * ip + 8 * eg;
        if (n >= N) continue;
        if (p.bits) dst[ip][0] = p.bits[opix * p.ldbits + (n >> 3)];     // one byte instead of 16 (block-uniform choice)
        else dst[ip] = *reinterpret_cast<const u32x4_t*>(actp + opix * p.ldact + n);
      }
    }
  };
  load_masks(0, mk);
#pragma unroll
  for (int j = 0; j < 8; j++) {
    u32x4_t mkn[2] = {u32x4_t{0u, 0u, 0u, 0u}, u32x4_t{0u, 0u, 0u, 0u}};
    if (j + 1 < 8) load_masks(j + 1, mkn);
    const size_t opix = out_pixel(j);
    unsigned wbits[2] = {0u, 0u};
#pragma unroll
    for (int ip = 0; ip < 2; ip++) {                           // fragments 2 ip, 2 ip + 1: channels n .. n + 7 of this lane
      const int n = n0 + 32 * ip + 8 * eg;
      if (n >= N) continue;                                    // N is a multiple of 8
      f32x4_t v0 = acc[2 * ip][j], v1 = acc[2 * ip + 1][j];
      if (EPI == EPI_BIAS_ACT) {
        if (p.bias) {
          v0 += *reinterpret_cast<const f32x4_t*>(p.bias + n);
          v1 += *reinterpret_cast<const f32x4_t*>(p.bias + n + 4);
        }
        if (p.relu) {
#pragma unroll
          for (int r = 0; r < 4; r++) { v0[r] = fmaxf(v0[r], 0.f); v1[r] = fmaxf(v1[r], 0.f); }
        }
      } else {
        if (actp && p.bits) apply_relu_bits8(mk[ip][0], v0, v1);
        else if (actp) {
          const u32x4_t a4 = mk[ip];
#pragma unroll
          for (int h = 0; h < 2; h++) {
            if (!(unpack_lo<T>(a4[h]) > 0.f)) v0[2 * h] = 0.f;
            if (!(unpack_hi<T>(a4[h]) > 0.f)) v0[2 * h + 1] = 0.f;
            if (!(unpack_lo<T>(a4[2 + h]) > 0.f)) v1[2 * h] = 0.f;
            if (!(unpack_hi<T>(a4[2 + h]) > 0.f)) v1[2 * h + 1] = 0.f;
          }
        }
        bsum[2 * ip] += v0;
        bsum[2 * ip + 1] += v1;
        if (p.accumulate) {
          const u32x4_t o4 = *reinterpret_cast<const u32x4_t*>(yout + opix * p.ldy + n);
#pragma unroll
          for (int h = 0; h < 2; h++) {
            v0[2 * h] += unpack_lo<T>(o4[h]); v0[2 * h + 1] += unpack_hi<T>(o4[h]);
            v1[2 * h] += unpack_lo<T>(o4[2 + h]); v1[2 * h + 1] += unpack_hi<T>(o4[2 + h]);
          }
        }
      }
      const u32x4_t o = {pack2<T>(v0[0], v0[1]), pack2<T>(v0[2], v0[3]), pack2<T>(v1[0], v1[1]), pack2<T>(v1[2], v1[3])};
      *reinterpret_cast<u32x4_t*>(yout + opix * p.ldy + n) = o;
      if (EPI == EPI_BIAS_ACT && p.bits && !p.bits_words) p.bits[opix * p.ldbits + (n >> 3)] = (unsigned char)relu_bits8<T>(o);
      if (EPI == EPI_BIAS_ACT && p.bits_words) wbits[ip] = relu_bits8<T>(o) << (8 * eg);
    }
    if (EPI == EPI_BIAS_ACT && p.bits_words) {                   // block-uniform: every lane takes part in the row exchange
#pragma unroll
      for (int ip = 0; ip < 2; ip++) {
        const unsigned wd = rows4_or(wbits[ip]);                 // the 32 channels n0 + 32 ip .. of this lane's pixel
        if (eg == 0 && n0 + 32 * ip < N) *reinterpret_cast<unsigned*>(p.bits + opix * p.ldbits + ((n0 + 32 * ip) >> 3)) = wd;
      }
    }
    mk[0] = mkn[0]; mk[1] = mkn[1];
    __builtin_amdgcn_sched_barrier(0);
  }
  if (EPI == EPI_MASK && (p.db || p.db2)) {
    // all 8 waves cover the same 64 channels: butterfly over the 16 pixel lanes, meet in LDS (free after the last barrier of the
    // K loop), ONE partial row per work-group for the ordered row reduction (no atomics); without a workspace: atomics
    float* red = reinterpret_cast<float*>(wb0);
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float t = bsum[i][r];
        t = row16_sum(t);
        const int c = 32 * (i >> 1) + 8 * eg + 4 * (i & 1) + r;   // the channel this accumulator belongs to (w_row)
        if (eq == 0) {
          if (p.dbws) red[wave * 64 + c] = t;
          else if (n0 + c < N) {
            float* qd = (n0 + c) < p.db_split ? (p.db ? p.db + n0 + c : nullptr) : (p.db2 ? p.db2 + (n0 + c - p.db_split) : nullptr);
            if (qd) atomicAdd(qd, t);
          }
        }
      }
    }
    if (p.dbws) {
      __syncthreads();
      if (tid < 64 && n0 + tid < N) {
        float t = red[tid];
#pragma unroll
        for (int k = 1; k < 8; k++) t += red[k * 64 + tid];
        p.dbws[(size_t)m_tile * N + n0 + tid] = t;
      }
    }
  }
#ifdef GCT2_STAMP
  GCT2_CLOCK_EXIT(p.clock ? p.stamps : nullptr, 8, wave, lane);
#endif
}

}  // namespace

int tapgemm_dbpart_reduce(const float* part, int rows, const TapGemmParams& p, hipStream_t s);   // tapgemm_mfma.hip

// the halo kernel takes FORM_CONVT problems whose SMALL grid tiles into 16 x 16 patches: the Conv2DTranspose forward
// (bias + ReLU) and the Conv2D input gradient (mask / accumulate / fused bias gradient)
bool halo_convT_wanted(const gct2_ctx& c, int epi, const TapGemmParams& p) {
  const int g_halo_mode = c.halo_mode;
  if (g_halo_mode == 1) return false;
  if ((p.Hs & 15) || (p.Ws & 15)) return false;
  // 16-byte epilogue accesses: output (and mask) views aligned to 16 bytes with pixel strides that are multiples of 8 elements
  if ((uintptr_t)p.y % 16 || p.ldy % 8 || (p.act && ((uintptr_t)p.act % 16 || p.ldact % 8))) return false;
  if (g_halo_mode == 2) return true;
  // automatic: layers whose source-tile traffic dominates (few output channels per pixel) and that fill the chip
  const int tiles = p.B * (p.Hs >> 4) * (p.Ws >> 4) * ((p.N + 63) / 64);
  return p.N <= 256 && tiles >= 256;          // measured vs tapgemm: U0 fwd 177 -> 136 us, U1 fwd 126 -> 120, U2 fwd 124 -> 122, D1 dgrad 88 -> 81, D2 dgrad 70 -> 66
}

int pw_head_finish(const float* part, int rows, float* dw, float* db, float* loss, float* db_dx, int ndw, int Cout, float inv_n,
                   int accumulate, hipStream_t s);     // pointwise.hip

// UpShuffle_0 forward with the train-step head in its epilogue (EPI_HEAD): needs N = 64 (one n-tile: a wave owns every channel
// of its pixels), the shape constraints of the halo kernel and room for one partial row per work-group in the workspace.
bool halo_head_supported(const gct2_ctx& c, int dtype, const TapGemmParams& p) {
  if (dtype != GCT2_BF16 && dtype != GCT2_F16) return false;
  if (p.N != 64 || (p.Hs & 15) || (p.Ws & 15) || p.K % 8 || p.ldx % 8) return false;
  if ((uintptr_t)p.y % 16 || p.ldy % 8 || (uintptr_t)p.x % 16 || (uintptr_t)p.w % 16) return false;
  const size_t rows = (size_t)p.B * (p.Hs >> 4) * (p.Ws >> 4);
  return c.ws && c.ws_bytes >= rows * HEAD_ROW * sizeof(float);
}
int halo_head(gct2_ctx& c, int dtype, TapGemmParams p, float* dw, float* db, float* loss, float* db_up, int accumulate,
              hipStream_t s) {
  p.m_tiles = p.B * (p.Hs >> 4) * (p.Ws >> 4);
  p.n_tiles = 1;
  p.xcd_chunk = (p.m_tiles + 7) / 8;
  p.ksplit = 1;
  p.dbws = nullptr;
  p.head.part = c.ws;
#ifdef GCT2_STAMP
  p.stamps = c.stamps;
  p.clock = c.stamps_bytes >= GCT2_CLOCK_BYTES ? 1 : 0;
#endif
  dim3 grid(8 * p.xcd_chunk);
  gct2_log(c, "halo:convT:head");
  if (dtype == GCT2_BF16) hipLaunchKernelGGL((halo_convT_kernel<__bf16, EPI_HEAD>), grid, dim3(512), 0, s, p);
  else hipLaunchKernelGGL((halo_convT_kernel<_Float16, EPI_HEAD>), grid, dim3(512), 0, s, p);
  if (int e = gct2_check_launch("halo_head")) return e;
  return pw_head_finish(c.ws, p.m_tiles, dw, db, loss, db_up, p.head.Cin * p.head.Cout, p.head.Cout, 1.0f / p.head.count, accumulate, s);
}

int halo_convT(gct2_ctx& c, int dtype, int epi, TapGemmParams p, hipStream_t s) {
  p.m_tiles = p.B * (p.Hs >> 4) * (p.Ws >> 4);
  p.n_tiles = (p.N + 63) / 64;
  p.xcd_chunk = (p.m_tiles + 7) / 8;
  p.ksplit = 1;
  p.dbws = nullptr;
  p.bits_words = (p.bits && (uintptr_t)p.bits % 4 == 0 && p.ldbits % 4 == 0 && p.N % 32 == 0) ? 1 : 0;
  float* queued = nullptr;
  if (epi == EPI_MASK && (p.db || p.db2)) {      // partial bias-gradient rows at the tail of the workspace, one per work-group row
    const size_t need = (size_t)p.m_tiles * p.N * sizeof(float);
    if (c.ws && c.ws_bytes >= need + 16) p.dbws = c.ws + (c.ws_bytes - need) / sizeof(float) / 4 * 4;
    if (p.dbws && c.dbq) {                         // a registered bias queue takes the rows (tapgemm_mfma.hip)
      queued = c.dbq_alloc(need / sizeof(float));
      if (queued) p.dbws = queued;
      else if (int e = tapgemm_dbq_flush(c, s)) return e;
    }
    if (!p.dbws) {                                 // atomics, at once: queued row sets of these targets go first
      if (int e = tapgemm_dbq_flush_for(c, p.db, p.db_split, p.db2, p.N - p.db_split, s)) return e;
      zero_overwritten_db(p, s);
    }
  }
#ifdef GCT2_STAMP
  p.stamps = c.stamps;
  p.clock = c.stamps_bytes >= GCT2_CLOCK_BYTES ? 1 : 0;
#endif
  dim3 grid(8 * p.xcd_chunk * p.n_tiles);
  gct2_log(c, "halo:convT:%s%s", epi == EPI_BIAS_ACT ? "bias_act" : "mask", p.bits ? ":bits" : "");
  if (epi == EPI_BIAS_ACT) {
    if (dtype == GCT2_BF16) hipLaunchKernelGGL((halo_convT_kernel<__bf16, EPI_BIAS_ACT>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((halo_convT_kernel<_Float16, EPI_BIAS_ACT>), grid, dim3(512), 0, s, p);
  } else {
    if (dtype == GCT2_BF16) hipLaunchKernelGGL((halo_convT_kernel<__bf16, EPI_MASK>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((halo_convT_kernel<_Float16, EPI_MASK>), grid, dim3(512), 0, s, p);
  }
  if (epi == EPI_BIAS_ACT && p.bits) c.relu_bits_done = 1;      // the epilogue wrote the ReLU bit plane
  if (p.dbws) {
    if (queued) { if (int e = tapgemm_dbq_push(c, queued, p.m_tiles, p, s)) return e; }
    else if (int e = tapgemm_dbpart_reduce(p.dbws, p.m_tiles, p, s)) return e;
  }
  return gct2_check_launch("halo_convT");
}
