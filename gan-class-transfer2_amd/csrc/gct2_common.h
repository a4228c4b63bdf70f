// Shared device-side helpers for the gfx950 kernels (wave = 64, MFMA 16x16x32, LDS tiles).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include "../../include/gct2.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

// host-side error plumbing (capi.hip)
int gct2_fail(int code, const char* fmt, ...);
int gct2_check_launch(const char* what);

// the call context of include/gct2.h: caller-owned scratch + tile-selection knobs.  Host memory, used by ONE host thread at a time
// (the one-shot ReLU plane and the launch log are written by the layer entry points).
struct gct2_ctx {
  float* ws = nullptr; size_t ws_bytes = 0;        // split-K slabs, partial rows (forward / input-gradient / head calls)
  float* wws = nullptr; size_t wws_bytes = 0;      // weight-gradient slabs (falls back to ws)
  int tap_variant = 0;                             // forward / input-gradient tile: 0 = automatic, 2 = 128 x 128, 5 = 256 x 128
  int wgrad_variant = 0;                           // weight-gradient tile: 0 = automatic, 2 = 256 x 256, 4 / 5 = 256 x 256 in the r03 / r04 stage order, 3 = 128 x 128, 6 = 128 x 128 with the general (r04) address code, 7 = 128 x 128 with atomics
  int halo_mode = 0;                               // 0 = automatic, 1 = never, 2 = wherever the shape allows
  int xcd_order = 0;                               // tile -> XCD order: 0 = automatic, 1 = m-tile bands, 2 = weight slices
  int wgrad_split = 0;                             // forced pixel split of the 128 x 128 weight-gradient tile: 0 = automatic, v: 2^(v-1)
  int no_splitk = 0;                               // 1: forward / input-gradient GEMMs never split their reduction (tuning bit 8)
  int force_direct = 0;
  unsigned long long* stamps = nullptr; size_t stamps_bytes = 0;   // diagnostic builds only (gct2_ctx_set_stamp_buffer)
  // ReLU bit plane for the NEXT layer call (gct2_ctx_set_relu_bits): every layer entry point takes it out of the ctx first thing
  // (consumed by the forward / input-gradient calls, an error on the others); relu_bits_done: the launch that just ran wrote the
  // plane in its epilogue (else the forward entry point derives it from y)
  unsigned char* relu_bits = nullptr; int relu_ldbits = 0; int relu_bits_done = 0;
  // launch log (gct2_ctx_log_launches): which kernel every layer call selected, as text tokens - for tests that must know
  bool log_on = false, log_full = false; std::string log;
  // deferred bias-gradient row sums (gct2_ctx_set_bias_queue, ABI v16): while a queue buffer is registered, the input-gradient calls
  // leave the partial rows of their fused bias gradients THERE (not at the tail of the workspace, which the next call reuses) and record
  // the job; gct2_bias_queue_flush sums every recorded row set - same geometry and order as the immediate reduction launch, hence the
  // same bits - in two launches (the targets a job overwrites first, the ones it adds to second) instead of one per call
  struct DbJob { const float* part; int rows, N; float* db; int db_split; float* db2; int db_acc; };
  float* dbq = nullptr; size_t dbq_floats = 0, dbq_used = 0;
  std::vector<DbJob> dbq_jobs;
  float* dbq_alloc(size_t floats) {
    floats = (floats + 3) / 4 * 4;
    if (!dbq || dbq_jobs.size() >= 16 || dbq_used + floats > dbq_floats) return nullptr;
    float* q = dbq + dbq_used;
    dbq_used += floats;
    return q;
  }
  float* wgrad_scratch(size_t* bytes) const {
    if (wws) { *bytes = wws_bytes; return wws; }
    *bytes = ws_bytes; return ws;
  }
};
// appends "token;" to the launch log of the ctx when it is enabled (capi.hip)
void gct2_log(gct2_ctx& c, const char* fmt, ...);

#ifdef GCT2_STAMP
// Diagnostic build only (make stamp / EXTRA=-DGCT2_STAMP; gct2_build_flags() says so and product hosts refuse the library).
// In-kernel clock of a K loop (MI355X_MICROARCH.md, "DVFS give-back" item 6): s_memtime (shader cycles) and s_memrealtime (100 MHz)
// stamped once in front of and once behind the loop; clock = d(memtime) / d(memrealtime) x 100 MHz.  Every wave writes its four
// values to the CLOCK REGION of the stamp buffer: entry (work-group * waves + wave) at u64 offset GCT2_CLOCK_OFF, if the buffer is
// at least GCT2_CLOCK_BYTES (8 MiB) long.  Nothing else reads that memory; no output depends on a stamp.
constexpr size_t GCT2_CLOCK_OFF = (size_t)1 << 19, GCT2_CLOCK_ENTRIES = (size_t)1 << 16, GCT2_CLOCK_BYTES = (GCT2_CLOCK_OFF + 8 * GCT2_CLOCK_ENTRIES) * 8;
// entry = 8 u64: [0] memtime, [1] memrealtime in front of the K loop; [2], [3] the same behind it; [4] memrealtime at kernel entry;
// [5] memrealtime at the end of the epilogue (phases of a work-group's life: setup = [1] - [4], loop = [3] - [1], epilogue = [5] - [3])
struct ClockStamp { unsigned long long t0 = 0, r0 = 0, t1 = 0, r1 = 0, rin = 0; };
__device__ __forceinline__ void clock_now(unsigned long long& t, unsigned long long& r) {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "=s"(r)::"memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ unsigned long long realtime_now() {
  unsigned long long r;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return r;
}
// the loop stamps are stored right behind the loop (as before); the exit stamp when the epilogue's stores have been ISSUED and
// drained (s_waitcnt vmcnt(0)): what a work-group's slot on the CU really costs
__device__ __forceinline__ void clock_store(unsigned long long* stamps, const ClockStamp& c, int waves_per_group, int wave, int lane) {
  const size_t e = (size_t)blockIdx.x * waves_per_group + wave;
  if (stamps && lane == 0 && e < GCT2_CLOCK_ENTRIES) {
    unsigned long long* o = stamps + GCT2_CLOCK_OFF + e * 8;
    o[0] = c.t0; o[1] = c.r0; o[2] = c.t1; o[3] = c.r1; o[4] = c.rin;
  }
}
__device__ __forceinline__ void clock_exit(unsigned long long* stamps, int waves_per_group, int wave, int lane) {
  const size_t e = (size_t)blockIdx.x * waves_per_group + wave;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long r = realtime_now();
  if (stamps && lane == 0 && e < GCT2_CLOCK_ENTRIES) stamps[GCT2_CLOCK_OFF + e * 8 + 5] = r;
}
#define GCT2_CLOCK_DECL ClockStamp clk_; clk_.rin = realtime_now()
#define GCT2_CLOCK_BEGIN clock_now(clk_.t0, clk_.r0)
#define GCT2_CLOCK_END(stamps, nwaves, wave, lane) do { clock_now(clk_.t1, clk_.r1); clock_store(stamps, clk_, nwaves, wave, lane); } while (0)
#define GCT2_CLOCK_EXIT(stamps, nwaves, wave, lane) clock_exit(stamps, nwaves, wave, lane)
#else
#define GCT2_CLOCK_DECL
#define GCT2_CLOCK_BEGIN
#define GCT2_CLOCK_END(stamps, nwaves, wave, lane)
#define GCT2_CLOCK_EXIT(stamps, nwaves, wave, lane)
#endif

template <typename T> struct is16 { static constexpr bool value = sizeof(T) == 2; };

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

// D(16x16,f32) += A(16x32) * B(32x16); operands as raw 128-bit fragments (8 x 16-bit).
// lane l holds A[row l&15][k = 8*(l>>4)+j], B[k = 8*(l>>4)+j][col l&15], D[row 4*(l>>4)+r][col l&15]
// (verified on hardware by tests/hw_probe/probe_mfma_tr.hip).
template <typename T> __device__ __forceinline__ f32x4_t mfma16(u32x4_t a, u32x4_t b, f32x4_t c);
template <> __device__ __forceinline__ f32x4_t mfma16<__bf16>(u32x4_t a, u32x4_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                 __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_t mfma16<_Float16>(u32x4_t a, u32x4_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a),
                                                __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// ---- LDS tile images -----------------------------------------------------------------------------
// "N image": [rows][64 x 16-bit] = 128-byte rows, 16-byte chunk c of row r stored at chunk
//  c ^ ((r>>1)&7): ds_read_b128 of (row = base+(lane&15), chunk = 4*kk+(lane>>4)) is conflict-free.
__device__ __forceinline__ int nimg_off(int row, int chunk) {
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}
// "T image": [64 k-rows][128 x 16-bit] = 256-byte rows (reduction index is the ROW); the 32-byte chunk
//  q of row k is stored at q ^ f(k), f(k) = (k&3) | ((k>>3)&1)<<2, so a ds_read_tr16_b64 (4 rows x 16
//  columns per 16-lane group) touches 64 distinct banks per 32-lane half.
__device__ __forceinline__ int timg_swz(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }
__device__ __forceinline__ int timg_off(int k, int chunk16) {
  return k * 256 + ((((chunk16 >> 1) ^ timg_swz(k))) << 5) + ((chunk16 & 1) << 4);
}

__device__ __forceinline__ u32x4_t lds_read128(const char* lds, int off) {
  return *reinterpret_cast<const u32x4_t*>(lds + off);
}
__device__ __forceinline__ void lds_write128(char* lds, int off, u32x4_t v) {
  *reinterpret_cast<u32x4_t*>(lds + off) = v;
}
// fragment (8 consecutive reduction elements for this lane's row/col) out of an N image
__device__ __forceinline__ u32x4_t nimg_frag(const char* img, int rowbase, int kk, int lane) {
  return lds_read128(img, nimg_off(rowbase + (lane & 15), 4 * kk + (lane >> 4)));
}
// same fragment out of a T image (column block `colbase`, multiple of 16) via two transposed reads
__device__ __forceinline__ u32x4_t timg_frag(const char* img, int colbase, int kk, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int k0 = kk * 32 + 8 * g + q;
  const int c32 = colbase >> 4;
  const int o0 = k0 * 256 + ((c32 ^ timg_swz(k0)) << 5) + p * 8;
  const int o1 = (k0 + 4) * 256 + ((c32 ^ timg_swz(k0 + 4)) << 5) + p * 8;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(img + o0));
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(img + o1));
  u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
  u32x4_t r = {l2[0], l2[1], h2[0], h2[1]};
  return r;
}

// sum over the 16 lanes of a row (lane & 15), result in every lane: four DPP adds (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror,
// row_mirror) - the same operand pairs, hence the same bits, as the butterfly t += __shfl_xor(t, 1 / 2 / 4 / 8), which hipcc lowers to
// four ds_bpermute round trips through the LDS crossbar
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float t) {
  t += dpp_f32<0xB1>(t);
  t += dpp_f32<0x4E>(t);
  t += dpp_f32<0x141>(t);
  t += dpp_f32<0x140>(t);
  return t;
}
// sum over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48), result in every lane: v_permlane16_swap / v_permlane32_swap
// of a value with itself leave (row, partner row) in the two operands - the same pairs as t += __shfl_xor(t, 16); t += __shfl_xor(t, 32)
// (the second operand goes through an opaque copy: with the SAME value tied to both read-write operands hipcc (ROCm 7.2) emitted a
// swap whose first operand kept its old upper half - caught by tests/hw_probe/probe_rowsum.hip)
__device__ __forceinline__ float rows4_sum(float t) {
  float a = t, b = t;
  asm volatile("" : "+v"(b));
  asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  t = a + b;
  a = t; b = t;
  asm volatile("" : "+v"(b));
  asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}

// OR over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48), result in every lane: the swaps of rows4_sum on integers
__device__ __forceinline__ unsigned rows4_or(unsigned t) {
  unsigned a = t, b = t;
  asm volatile("" : "+v"(b));
  asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  t = a | b;
  a = t; b = t;
  asm volatile("" : "+v"(b));
  asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a | b;
}

__device__ __forceinline__ u32x4_t gload128(const void* p) {
  return *reinterpret_cast<const u32x4_t*>(p);
}

// two floats -> one packed pair of the 16-bit storage type: ONE v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 (round to nearest even, the same
// bits as two scalar conversions; hipcc turns the scalar form into two conversions + shift + or: 4 instructions per pair in every epilogue)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
template <typename T> __device__ __forceinline__ uint32_t pack2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) T t2_t;
  static_assert(sizeof(T) == 2, "pack2 packs 16-bit storage types");
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, t2_t));
}
template <typename T> __device__ __forceinline__ float unpack_lo(uint32_t u) {
  return to_f32<T>(__builtin_bit_cast(T, (uint16_t)(u & 0xffffu)));
}
template <typename T> __device__ __forceinline__ float unpack_hi(uint32_t u) {
  return to_f32<T>(__builtin_bit_cast(T, (uint16_t)(u >> 16)));
}

// ReLU bit planes (r03): one byte per 8 consecutive channels of a pixel.  bit k = (element k of the 8 packed 16-bit values > 0) - the
// comparison the mask epilogues apply to the activation itself, so a plane written by a forward epilogue reproduces them exactly
template <typename T> __device__ __forceinline__ unsigned relu_bits8(u32x4_t o) {
  unsigned m = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    m |= (unpack_lo<T>(o[k]) > 0.f ? 1u : 0u) << (2 * k);
    m |= (unpack_hi<T>(o[k]) > 0.f ? 1u : 0u) << (2 * k + 1);
  }
  return m;
}
__device__ __forceinline__ void apply_relu_bits8(unsigned m, f32x4_t& a, f32x4_t& c) {
#pragma unroll
  for (int r = 0; r < 4; r++) {
    if (!((m >> r) & 1u)) a[r] = 0.f;
    if (!((m >> (4 + r)) & 1u)) c[r] = 0.f;
  }
}

// ---- geometry of the two 4x4/stride-2 "tap GEMMs" ------------------------------------------------
// FORM_CONV : out on the SMALL grid (Hs x Ws), source on the BIG grid (2Hs x 2Ws), 16 taps,
//             weights [tap][k][n]   (Conv2D forward, Conv2DTranspose input-gradient)
// FORM_CONVT: out on the BIG grid, one launch-z per output parity phase, source on the SMALL grid,
//             4 taps per phase, weights [tap][n][k]   (Conv2DTranspose forward, Conv2D input-gradient)
// FORM_S1   : 'same' stride-1 convolution with ks x ks taps (Block's 3x3, the 1x1 projection of residual=True; train.py:104-143):
//             out and source on ONE grid, tap (dh, dw) reads (h + dh - pad, w + dw - pad), weights [tap][k][n] like FORM_CONV
// FORM_S1T  : its input gradient: the same walk with the taps flipped, weights [tap][n][k] like FORM_CONVT
enum { FORM_CONV = 0, FORM_CONVT = 1, FORM_S1 = 2, FORM_S1T = 3 };
enum { EPI_BIAS_ACT = 0, EPI_MASK = 1, EPI_HEAD = 2 };

// Keras' mixed_float16 policy (train.py:43-45) makes the Dense output and the gradient entering it fp16 tensors; the loss is
// taken on the fp16 values cast to fp32 (train.py:262-263).  GCT2_F16 reproduces those two rounding points; fp32 / bf16 keep fp32.
template <typename T> __device__ __forceinline__ float keras_f16_point(float v) {
  if constexpr (sizeof(T) == 2 && !__is_same(T, __bf16)) return (float)(_Float16)v;
  else return v;
}

// EPI_HEAD (halo kernel, UpShuffle_0 forward of the train step): the Dense(3) head + fp32 MSE + both of their gradients are
// evaluated in the epilogue on the activations the work-group has just produced (train.py:198-202, 262-272), which are then
// never written to HBM; the output view receives the pre-activation gradient of the layer instead.
constexpr int HEAD_ROW = 288;       // partial row: [0,216) dW (c*Cout+o) | [216,219) db | 219 loss | [224,288) db of the layer below
struct HeadFuse {
  const float* w; const float* b;   // Dense kernel (Cin x Cout) and bias, fp32
  const float* target;              // fp32 [pixels][Cout]
  float* pred;                      // fp32 [pixels][Cout] or null
  const void* x2; int ldx2;         // input channels [N, Cin) of the head (the packed image), compute dtype
  float* part;                      // [work-groups][HEAD_ROW] partial rows
  const float* loss_scale;          // device scalar or null
  int Cin, Cout;                    // head input channels (N + image channels), outputs (<= 3)
  float count;                      // pixels * Cout (the mean of the loss)
};

struct TapGemmParams {
  const void* x; int ldx;        // source activations / gradients
  const void* w;                 // weights, Keras layout
  const float* bias;             // EPI_BIAS_ACT
  const void* act; int ldact;    // EPI_MASK: mask source on the output grid (may be null)
  void* y; int ldy;              // output view
  int B, Hs, Ws;                 // SMALL grid
  int K, N;                      // reduction channels per tap, output channels
  int relu, accumulate;
  float* ws; int ksplit;         // split-K: fp32 partial slabs [ksplit][out pixels][N] in the registered workspace
  int m_tiles, n_tiles, xcd_chunk;   // launch geometry (filled by the launcher): see xcd_tile()
  int wstat;                         // 1: weight-stationary tile -> XCD order (tapgemm_kernel)
  int wide;                          // output / mask views allow 16-byte accesses (filled by the launcher)
  float* db; int db_split; float* db2;   // EPI_MASK: bias-gradient targets (column sums of the masked result), may be null
  int db_acc;                            // bit 0: db is added to (else overwritten); bit 1: the same for db2
  float* dbws;                           // partial bias-gradient rows [m_tiles*phases | finalize rows][N] in the workspace, or null (atomics)
  HeadFuse head;                         // EPI_HEAD only
  int ks = 0;                            // FORM_S1 / FORM_S1T: kernel size (odd, <= 5)
  int bits_words = 0;                    // 1: plane and strides are 4-byte aligned and N % 32 == 0 -> the four lane rows of a pixel merge their bytes into ONE 32-bit store
  unsigned char* bits = nullptr; int ldbits = 0;   // ReLU bit plane [pixel][ldbits bytes], bit k of byte c = (channel 8c + k of the view > 0):
                                         // EPI_BIAS_ACT writes it beside y, EPI_MASK reads it instead of act (16-byte epilogues only)
  int ws_shift = -1, hs_shift = -1;      // log2 of Ws / Hs when they are powers of two (filled by the launcher), else -1: the per-lane
                                         // pixel decode then uses shifts instead of four integer divisions per row
#ifdef GCT2_STAMP
  unsigned long long* stamps = nullptr;  // diagnostic build: phase stamps of one wave per work-group (gct2_ctx_set_stamp_buffer)
  int clock = 0;                         // ... and the buffer is long enough for the clock region (GCT2_CLOCK_OFF)
#endif
};
inline int pow2_shift(int v) { return (v > 0 && !(v & (v - 1))) ? __builtin_ctz((unsigned)v) : -1; }
// m -> (sw, sh, b) on a [B][Hs][Ws] grid
__device__ __forceinline__ void decode_pixel(int m, int Hs, int Ws, int hs_shift, int ws_shift, int& sw, int& sh, int& b) {
  if (ws_shift >= 0 && hs_shift >= 0) {            // wave-uniform
    sw = m & (Ws - 1);
    const int t = m >> ws_shift;
    sh = t & (Hs - 1);
    b = t >> hs_shift;
  } else {
    sw = m % Ws;
    const int t = m / Ws;
    sh = t % Hs;
    b = t / Hs;
  }
}

// XCD-aware work-group -> tile map.  The dispatcher deals consecutive work-group ids round-robin over the 8
// XCDs (private L2 each; observed, speed only - MI355X_MICROARCH.md "Workgroup dispatch"), so ids with equal
// id % 8 share an L2.  Each XCD walks a CONTIGUOUS band of `chunk` m-tiles, and for every m-tile runs all of
// its `inner` siblings (n-tiles x output phases, which re-read the same source pixels) back to back.
// Returns false for the padding ids of a ragged last band.
__device__ __forceinline__ bool xcd_tile(int id, int m_tiles, int inner, int chunk, int& m_tile, int& in) {
  const int xcd = id & 7, j = id >> 3;
  in = j % inner;
  m_tile = xcd * chunk + j / inner;
  return (j / inner) < chunk && m_tile < m_tiles;
}

// the atomic fall-backs of the fused bias gradients add into their targets: overwritten targets start from zero
// bias queue of a call context (tapgemm_mfma.hip): record the partial rows a launch left in the queue buffer / reduce everything recorded
int pw_occupy(int workgroups, unsigned long long ticks, hipStream_t s);      // pointwise.hip: gct2_stream_occupy
int tapgemm_dbq_push(gct2_ctx& c, const float* part, int rows, const TapGemmParams& p, hipStream_t s);
int tapgemm_dbq_flush(gct2_ctx& c, hipStream_t s);
int tapgemm_dbq_flush_for(gct2_ctx& c, const float* db, int n0, const float* db2, int n1, hipStream_t s);   // in front of an immediate writer
inline void zero_overwritten_db(const TapGemmParams& p, hipStream_t s) {
  if (p.db && !(p.db_acc & 1) && p.db_split > 0) (void)hipMemsetAsync(p.db, 0, (size_t)p.db_split * sizeof(float), s);
  if (p.db2 && !(p.db_acc & 2) && p.N > p.db_split) (void)hipMemsetAsync(p.db2, 0, (size_t)(p.N - p.db_split) * sizeof(float), s);
}

// Keras ResourceApplyAdam on one element (epsilon added to sqrt(v)); ONE definition for every kernel that applies it
__device__ __forceinline__ void adam_keras_update(float& p, float& m, float& v, float g, float alpha, float b1, float ob1, float b2,
                                                  float ob2, float eps) {
  // no FMA contraction here: which products hipcc fuses depends on the code around the call (two kernels sharing this function once
  // differed in the last bit); plain IEEE multiplies and adds in this order are the same everywhere - and what an unfused
  // TensorFlow kernel computes
#pragma clang fp contract(off)
  const float m1 = b1 * m, m2 = ob1 * g;
  m = m1 + m2;
  const float v1 = b2 * v, v2 = (ob2 * g) * g;
  v = v1 + v2;
  const float num = alpha * m, den = sqrtf(v) + eps;
  p = p - num / den;
}
// wgrad: dw[tap][cb][cs] += sum_r big[pix_big(r,tap)][cb] * small[r][cs], r over the SMALL grid.
struct WgradParams {
  const void* big; int ldbig;     // tensor on the BIG grid (2Hs x 2Ws), Cb channels
  const void* small; int ldsmall; // tensor on the SMALL grid, Cs channels
  float* dw;                      // fp32 [16][Cb][Cs]
  int B, Hs, Ws, Cb, Cs;
  int rsplit;                     // number of r-range splits
  float* ws;                      // partial-tile slabs [rsplit][16*Cb][Cs] in the registered workspace, or null (atomics)
  int accumulate;                 // 1: dw += result (caller keeps a running / pre-zeroed gradient); 0: dw = result
  int ks = 0;                     // 0: the 4x4 / stride-2 layers; odd ks: 'same' stride-1 convolution (both tensors on one grid, ks*ks taps)
#ifdef GCT2_STAMP
  unsigned long long* stamps = nullptr;   // diagnostic build: phase stamps of one wave per work-group (gct2_ctx_set_stamp_buffer)
  int clock = 0;                          // ... and the buffer is long enough for the clock region (GCT2_CLOCK_OFF)
#endif
};
// wgrad_mfma(): when `defer` is non-null and the launch left its result as workspace slabs, the slab reduction is NOT launched and
// the slabs are described here (the caller folds them into the optimizer read); nslab = 0 means dw holds the gradient
struct WgradSlabs { const float* base; int nslab; size_t stride; };
