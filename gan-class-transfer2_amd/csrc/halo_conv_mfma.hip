// Halo-tile variant of the FORM_CONV tap GEMM (Conv2D forward, train.py:161-166, and the input gradient of Conv2DTranspose,
// train.py:148-153): the 4x4 / stride-2 window seen through a space-to-depth view of its source.
//
// tapgemm_kernel<FORM_CONV> stages a 256-pixel x 64-channel source tile per (tap, k-chunk): 16 gathers of 32 KiB per chunk for a
// 256-pixel patch (23 KiB through the CU's vector-memory path per million multiply-adds; r03 analysis in DESIGN.md: at full MFMA
// rate that is ~3/4 of the 64 B/clk the path delivers, and the launches sit at 32-45 % of the MFMA peak).  The 16 taps of an output
// pixel (oy, ox) read source rows 2oy-1 .. 2oy+2 and columns 2ox-1 .. 2ox+2.  With the source seen as a grid of 2 x 2 pixel blocks
// (block (sy, sx) = source pixels (2sy + phr, 2sx + phc), phr, phc in {0, 1}) tap (kh, kw) reads block (oy + dy, ox + dx), phase
// (phr, phc) with  dy = (kh+1)/2 - 1, phr = (kh+1) & 1  (and the same for kw): every output pixel touches the 3 x 3 blocks around
// its own, exactly the access pattern of the FORM_CONVT halo kernel (halo_mfma.hip) with the four phases as part of the REDUCTION
// instead of the output.  So: a work-group owns 16 x 16 output pixels x 256 output channels; per 16-channel k-chunk it stages the
// 18 x 18 block halo ONCE (324 rows x 128 B: 4 phases x 16 channels) and the weights of 4 taps per round (64 k-rows x 256 n as
// two T images); 10 KiB per million multiply-adds.
//
// 8 waves: wave = (channel quarter nq, half of the patch) -> 128 pixels x 64 channels (8 x 4 MFMA tiles, 128 accumulators).
// One MFMA reduction step = 2 taps x 16 channels: lane group g of a fragment carries tap (kh, kp + 2 (g>>1)), channels 8 (g&1) ..;
// the two taps of a step differ by one block column, which the per-lane address absorbs.  Round = tap row kh: steps kp = 0, 1.
// Fragment addresses are per-lane registers + immediates, the DMA offsets per-lane registers + a scalar (the lean round of
// halo_mfma.hip).  Needs Hs, Ws (output grid) multiples of 16 and K a multiple of 16.
#include "gct2_common.h"
#include <algorithm>
#include <type_traits>

namespace {

constexpr unsigned OOB = 0x80000000u;
typedef __attribute__((address_space(3))) void lds_void_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)OOB, 0x00020000);
}
// 64 lanes x 16 B -> 1 KiB of LDS at lds_piece; per-lane byte offset voff (OOB: zeros), wave-uniform byte offset soff
__device__ __forceinline__ void dma16s(__amdgpu_buffer_rsrc_t rsrc, char* lds_piece, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)lds_piece, 16, (int)voff, (int)soff, 0, 0);
}
// halo image swizzle, keyed on the halo COLUMN (see halo_mfma.hip): conflict-free for ds_read_b128 of 16 consecutive columns
__device__ __forceinline__ int halo_swz(int hx) { return (((hx >> 1) & 1) << 2) | (((hx >> 2) & 1) << 1); }

constexpr int HP = 18;                        // halo pitch (blocks per halo row)
constexpr int HPIECES = 41;                   // 1-KiB pieces (8 block rows each) covering the 324 halo blocks
constexpr int HALO_BYTES = HPIECES * 1024;
constexpr int WB_BYTES = 2 * 64 * 256;        // two T images (128 channels each) of 64 k-rows x 256 B

template <typename T, int EPI>
__global__ __launch_bounds__(512) void halo_conv_kernel(TapGemmParams p) {
  // ONE array, the halo images first: every fragment address is a per-lane register + a 16-bit immediate
  __shared__ __attribute__((aligned(16))) char lds_all[2 * HALO_BYTES + 2 * WB_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nq = wave >> 1, mhalf = wave & 1;
  const int g = lane >> 4, q = lane & 15;
  const int Hs = p.Hs, Ws = p.Ws, K = p.K, N = p.N;
  const int tx_n = Ws >> 4, ty_n = Hs >> 4;
  int m_tile, n_tile;
  if (!xcd_tile((int)blockIdx.x, p.m_tiles, p.n_tiles, p.xcd_chunk, m_tile, n_tile)) return;
  const int tx = m_tile % tx_n, tq = m_tile / tx_n, ty = tq % ty_n, b = tq / ty_n;
  const int sh0 = ty * 16, sw0 = tx * 16, n0 = n_tile * 256;
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x), rs_w = make_rsrc(p.w);
  const int ldx2 = p.ldx * 2;

  // ---- per-lane DMA descriptors (loop-invariant; the k-chunk and the tap row go into the scalar offset) ------------------
  // halo: piece pi = wave + 8 i (i < 6, pi < 41) = halo blocks 8 pi .. 8 pi + 7; lane -> block 8 pi + (lane>>3), physical chunk
  // lane&7; logical chunk = (phr, phc, channel half)
  unsigned h_voff[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    const int pi = wave + 8 * i;
    const int row = 8 * pi + (lane >> 3);
    const int hy = row / HP, hx = row - hy * HP;
    const int lc = (lane & 7) ^ halo_swz(hx);
    h_voff[i] = OOB;
    if (pi < HPIECES && row < HP * HP) {
      const int sy = sh0 - 1 + hy, sx = sw0 - 1 + hx;          // block coordinates: both phases of a block inside the grid are inside
      if ((unsigned)sy < (unsigned)Hs && (unsigned)sx < (unsigned)Ws)
        h_voff[i] = (unsigned)(((b * 2 * Hs + 2 * sy + (lc >> 2)) * 2 * Ws + 2 * sx + ((lc >> 1) & 1)) * ldx2 + (lc & 1) * 16);
    }
  }
  // weights: piece pc = wave + 8 i (i < 4): T image pc>>4, k-rows 4 (pc&15) ..+3; lane -> k-row + (lane>>4), physical 16-byte
  // chunk lane&15 (T image: 32-byte chunk c of row k at c ^ timg_swz(k)).  k-row = 32 kp + 16 tapsel + channel: tap (kh, kp + 2 tapsel)
  unsigned w_voff[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int pc = wave + 8 * i;
    const int krow = 4 * (pc & 15) + (lane >> 4);
    const int log32 = ((lane & 15) >> 1) ^ timg_swz(krow);
    const int n = n0 + 128 * (pc >> 4) + (log32 * 2 + (lane & 1)) * 8;
    const int kw = (krow >> 5) + 2 * ((krow >> 4) & 1);
    w_voff[i] = n < N ? (unsigned)(((kw * K + (krow & 15)) * N + n) * 2) : OOB;
  }
  const unsigned KN8 = (unsigned)(4 * K * N * 2);              // bytes per tap ROW of the weight tensor
  const unsigned N32 = (unsigned)(32 * N);                     // bytes per 16-channel chunk

  const int nround = 4 * (K >> 4);                             // (k-chunk, tap row kh)

  // ---- per-lane fragment addresses ------------------------------------------------------------------------------------------
  // source block column of lane (g, q) in step kp: hx = q + (g>>1) + kp (kw = kp + 2 (g>>1): dx = {-1, 0, 0, 1}[kw], + 1 halo);
  // block row of pixel row j in tap row kh: hy = mhalf*8 + j + 1 + dy(kh) -> immediate
  int fa[2][2];                                                // [kp][phr]
#pragma unroll
  for (int kp = 0; kp < 2; kp++) {
    const int hx = q + (g >> 1) + kp;
#pragma unroll
    for (int phr = 0; phr < 2; phr++) {
      const int lc = ((phr * 2 + (1 - kp)) * 2) + (g & 1);     // phc = (kw+1)&1 = 1 - kp
      fa[kp][phr] = ((mhalf * 8 * HP + hx) << 7) + ((lc ^ halo_swz(hx)) << 4);
    }
  }
  // weight fragment i of the wave: channels 64 nq + 16 i + (lane&15) -> T image nq>>1, 32-byte chunk 4 (nq&1) + i; the two
  // transposed reads of a fragment (k-rows +0, +4) and the step (k-rows +32) are immediates
  int fwv[4];
  {
    const int qq = (lane >> 2) & 3, pp = lane & 3;
    const int k0 = 8 * g + qq;
#pragma unroll
    for (int i = 0; i < 4; i++)
      fwv[i] = 2 * HALO_BYTES + (nq >> 1) * 16384 + k0 * 256 + (((4 * (nq & 1) + i) ^ timg_swz(k0)) << 5) + pp * 8;
  }
  auto wfrag = [&](int off) -> u32x4_t {
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(lds_all + off));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(lds_all + off + 1024));
    u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
    return u32x4_t{l2[0], l2[1], h2[0], h2[1]};
  };

  f32x4_t acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 8; j++) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // idx = round & 7 (a literal at every call site): halo buffer idx>>2, weight buffer idx&1, tap row kh = idx&3.
  // FAST: the next round exists (and, K being a multiple of 16, every chunk is full) - no tail logic.
  auto lean_round = [&](auto fast_c, int round, int idx, int hcur, int hnext, int wcur, int wnext) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(fast_c)::value;
    const int kh = idx & 3;
    const int dy1 = (kh + 1) >> 1, phr = (kh + 1) & 1;         // 1 + dy
    const int khn = (idx + 1) & 3;
    const bool more = FAST || round + 1 < nround;
    const bool halo_due = kh == 3 && more;
    const unsigned kcn = (unsigned)((round + 1) >> 2);
    const unsigned s_w = (unsigned)khn * KN8 + kcn * N32, s_h = kcn * 32u;
#pragma unroll
    for (int kp = 0; kp < 2; kp++) {
      u32x4_t wf[4];
#pragma unroll
      for (int i = 0; i < 4; i++) wf[i] = wfrag(fwv[i] + wcur + kp * 8192);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const u32x4_t af = lds_read128(lds_all, fa[kp][phr] + hcur + (j + dy1) * HP * 128);
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i][j] = mfma16<T>(wf[i], af, acc[i][j]);
        if (kp == 0 && (j & 1) && more) {                      // weight piece j>>1 of the next round
          const int i = j >> 1;
          dma16s(rs_w, lds_all + 2 * HALO_BYTES + wnext + (wave + 8 * i) * 1024, w_voff[i], s_w);
        }
        if (kp == 1 && j < 6 && halo_due) {                    // halo piece j of the next k-chunk
          const int pi = wave + 8 * j;
          if (pi < HPIECES) dma16s(rs_x, lds_all + hnext + pi * 1024, h_voff[j], s_h);
        }
      }
    }
  };

  // ---- main loop: 8 rounds (two k-chunks) per trip so that every buffer role is a compile-time constant ------------------
#pragma unroll
  for (int i = 0; i < 6; i++) {
    const int pi = wave + 8 * i;
    if (pi < HPIECES) dma16s(rs_x, lds_all + pi * 1024, h_voff[i], 0u);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) dma16s(rs_w, lds_all + 2 * HALO_BYTES + (wave + 8 * i) * 1024, w_voff[i], 0u);
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
    for (; r + 8 < nround; r += 8) { GCT2_LEAN_TRIP(true) }
    for (; r < nround; r += 8) { GCT2_LEAN_TRIP(false) }
  }
#undef GCT2_LEAN_TRIP
#undef GCT2_LEAN_ROUND

  // ---- epilogue: lane holds out[pixel (row mhalf*8 + j, col q)][n = n0 + 64 nq + 16 i + 4 g + r] ----------------------------
  // EPI_BIAS_ACT: bias + ReLU (Conv2D forward).  EPI_MASK: ReLU mask of the tensor the gradient belongs to, optional accumulation
  // into the output view, column sums for the fused bias gradient (Conv2DTranspose input gradient), as in tapgemm_kernel.
  T* __restrict__ yout = reinterpret_cast<T*>(p.y);
  const T* __restrict__ actp = reinterpret_cast<const T*>(p.act);
  int elane = lane;
  asm volatile("" : "+v"(elane));                              // keeps the output addresses out of the K loop
  const int eq = elane & 15, eg = elane >> 4;
  const int nw = n0 + 64 * nq + 4 * eg;                        // + 16 i
  f32x4_t bsum[4];
#pragma unroll
  for (int i = 0; i < 4; i++) bsum[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  auto out_pixel = [&](int j) { return (size_t)((b * Hs + sh0 + mhalf * 8 + j) * Ws + sw0 + eq); };
  u32x2_t mk[4];
  auto load_masks = [&](int j, u32x2_t* dst) {
    if (EPI == EPI_MASK && actp) {
      const size_t opix = out_pixel(j);
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (nw + 16 * i < N) dst[i] = *reinterpret_cast<const u32x2_t*>(actp + opix * p.ldact + nw + 16 * i);
    }
  };
#pragma unroll
  for (int i = 0; i < 4; i++) mk[i] = u32x2_t{0u, 0u};
  load_masks(0, mk);
#pragma unroll
  for (int j = 0; j < 8; j++) {
    u32x2_t mkn[4] = {u32x2_t{0u, 0u}, u32x2_t{0u, 0u}, u32x2_t{0u, 0u}, u32x2_t{0u, 0u}};
    if (j + 1 < 8) load_masks(j + 1, mkn);
    const size_t opix = out_pixel(j);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int n = nw + 16 * i;
      if (n >= N) continue;                                    // N is a multiple of 8
      f32x4_t v = acc[i][j];
      if (EPI == EPI_BIAS_ACT) {
        if (p.bias) v += *reinterpret_cast<const f32x4_t*>(p.bias + n);
        if (p.relu) {
#pragma unroll
          for (int r = 0; r < 4; r++) v[r] = fmaxf(v[r], 0.f);
        }
      } else {
        if (actp) {
          const u32x2_t a2 = mk[i];
#pragma unroll
          for (int h = 0; h < 2; h++) {
            if (!(unpack_lo<T>(a2[h]) > 0.f)) v[2 * h] = 0.f;
            if (!(unpack_hi<T>(a2[h]) > 0.f)) v[2 * h + 1] = 0.f;
          }
        }
        bsum[i] += v;
        if (p.accumulate) {
          const u32x2_t o2 = *reinterpret_cast<const u32x2_t*>(yout + opix * p.ldy + n);
#pragma unroll
          for (int h = 0; h < 2; h++) { v[2 * h] += unpack_lo<T>(o2[h]); v[2 * h + 1] += unpack_hi<T>(o2[h]); }
        }
      }
      const u32x2_t o = {pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
      *reinterpret_cast<u32x2_t*>(yout + opix * p.ldy + n) = o;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) mk[i] = mkn[i];
    __builtin_amdgcn_sched_barrier(0);
  }
  if (EPI == EPI_MASK && (p.db || p.db2)) {
    // the two waves of a channel quarter cover the same 64 channels: butterfly over the 16 pixel lanes, meet in LDS (free after the
    // last barrier of the K loop), ONE partial row per work-group for the ordered row reduction; without a workspace: atomics
    float* red = reinterpret_cast<float*>(lds_all);
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float t = bsum[i][r];
        t = row16_sum(t);
        const int c = 16 * i + 4 * eg + r;                     // channel inside the wave's quarter
        if (eq == 0) {
          if (p.dbws) red[wave * 64 + c] = t;
          else if (n0 + 64 * nq + c < N) {
            const int n = n0 + 64 * nq + c;
            float* qd = n < p.db_split ? (p.db ? p.db + n : nullptr) : (p.db2 ? p.db2 + (n - p.db_split) : nullptr);
            if (qd) atomicAdd(qd, t);
          }
        }
      }
    }
    if (p.dbws) {
      __syncthreads();
      if (tid < 256 && n0 + tid < N) {
        const int w0 = 2 * (tid >> 6), c = tid & 63;
        p.dbws[(size_t)m_tile * N + n0 + tid] = red[w0 * 64 + c] + red[(w0 + 1) * 64 + c];
      }
    }
  }
}

}  // namespace

int tapgemm_dbpart_reduce(const float* part, int rows, const TapGemmParams& p, hipStream_t s);   // tapgemm_mfma.hip

// the conv-form halo kernel takes FORM_CONV problems whose OUTPUT grid tiles into 16 x 16 patches: Conv2D forward (bias + ReLU)
// and the Conv2DTranspose input gradient (mask / accumulate / fused bias gradient)
bool halo_conv_wanted(const gct2_ctx& c, int epi, const TapGemmParams& p) {
  if (c.halo_mode == 1) return false;
  if ((p.Hs & 15) || (p.Ws & 15) || p.K % 16) return false;
  if (c.halo_mode == 2) return true;
  // NOT taken automatically: measured 0..12 % slower than the tap GEMM at two work-groups per CU on the four layers it fits
  // (profiles/r03_halo_conv.txt; DESIGN.md section 6: 32-byte source segments per 16-channel chunk, 8-byte epilogue accesses, and -
  // the larger part - one work-group per CU puts the store bursts of all CUs in lockstep).  Tuning bit 12 switches it on for
  // full 256-channel tiles that give every CU a work-group.
  if (!c.halo_conv_auto) return false;
  const int tiles = p.B * (p.Hs >> 4) * (p.Ws >> 4) * ((p.N + 255) / 256);
  return p.N % 256 == 0 && tiles >= 256;
}

int halo_conv(const gct2_ctx& c, int dtype, int epi, TapGemmParams p, hipStream_t s) {
  p.m_tiles = p.B * (p.Hs >> 4) * (p.Ws >> 4);
  p.n_tiles = (p.N + 255) / 256;
  p.xcd_chunk = (p.m_tiles + 7) / 8;
  p.ksplit = 1;
  p.dbws = nullptr;
  float* deferred = nullptr;
  if (epi == EPI_MASK && (p.db || p.db2)) {      // partial bias-gradient rows at the tail of the workspace, one per work-group row
    const size_t ws_bytes = c.ws_bytes;
    float* ws = c.ws;
    const size_t need = (size_t)p.m_tiles * p.N * sizeof(float);
    if (ws && ws_bytes >= need + 16) p.dbws = ws + (ws_bytes - need) / sizeof(float) / 4 * 4;
    if (p.dbws) {     // an open row-sum deferral: the rows stay in the caller's row-sum buffer until gct2_rowsum_flush
      deferred = rowsum_alloc(c, (size_t)p.m_tiles, p.N);
      if (deferred) p.dbws = deferred;
    }
    if (!p.dbws) zero_overwritten_db(p, s);
  }
  dim3 grid(8 * p.xcd_chunk * p.n_tiles);
  if (epi == EPI_BIAS_ACT) {
    if (dtype == GCT2_BF16) hipLaunchKernelGGL((halo_conv_kernel<__bf16, EPI_BIAS_ACT>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((halo_conv_kernel<_Float16, EPI_BIAS_ACT>), grid, dim3(512), 0, s, p);
  } else {
    if (dtype == GCT2_BF16) hipLaunchKernelGGL((halo_conv_kernel<__bf16, EPI_MASK>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((halo_conv_kernel<_Float16, EPI_MASK>), grid, dim3(512), 0, s, p);
  }
  if (deferred) rowsum_record(c, p, deferred, p.m_tiles);
  else if (p.dbws) {
    if (int e = tapgemm_dbpart_reduce(p.dbws, p.m_tiles, p, s)) return e;
  }
  return gct2_check_launch("halo_conv");
}
