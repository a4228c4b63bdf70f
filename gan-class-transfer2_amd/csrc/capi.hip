// extern "C" boundary (include/gct2.h): argument validation, path selection (MFMA vs direct), launches.
#include "gct2_common.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>

// implemented in the kernel translation units
bool tapgemm_mfma_supported(int dtype, const TapGemmParams& p);
int tapgemm_mfma(gct2_ctx& c, int dtype, int form, int epi, const TapGemmParams& p, hipStream_t s);
int tapgemm_direct(int dtype, int form, int epi, const TapGemmParams& p, hipStream_t s);
bool wgrad_mfma_supported(int dtype, const WgradParams& p);
int wgrad_mfma(gct2_ctx& c, int dtype, WgradParams p, hipStream_t s, WgradSlabs* defer);
int wgrad_direct(int dtype, const WgradParams& p, hipStream_t s);
bool halo_head_supported(const gct2_ctx& c, int dtype, const TapGemmParams& p);
int halo_head(gct2_ctx& c, int dtype, TapGemmParams p, float* dw, float* db, float* loss, float* db_up, int accumulate, hipStream_t s);
bool rgb_fwd_supported(int dtype, const TapGemmParams& p);
int rgb_fwd(int dtype, const TapGemmParams& p, hipStream_t s);
bool rgb_fwd_writes_bits(const TapGemmParams& p);
int pw_relu_bits(int dtype, const void* y, int ldy, size_t pixels, int channels, unsigned char* bits, int ldbits, hipStream_t s);   // pointwise.hip
bool rgb_wgrad_supported(int dtype, const WgradParams& p);
int rgb_wgrad(const gct2_ctx& c, int dtype, WgradParams p, hipStream_t s, WgradSlabs* defer);
int pw_rng_uniform_int(uint64_t, uint64_t, uint64_t, int32_t*, size_t, int, int, hipStream_t);
int pw_rng_normal(uint64_t, uint64_t, uint64_t, float*, size_t, hipStream_t);
int pw_noise(int, const float*, const int32_t*, const float*, void*, int, void*, int, int, int, int, int, hipStream_t);
int pw_noise_rng(int, const float*, const int32_t*, uint64_t, uint64_t, uint64_t, float*, void*, int, void*, int, int, int, int, int, hipStream_t);
int pw_dense_fwd(int, const void*, int, const float*, const float*, float*, int, int, int, hipStream_t);
int pw_dense_bwd(int, const void*, int, const float*, const float*, void*, int, float*, float*, int, int, int, int, int, hipStream_t);
int pw_mse(const float*, const float*, float*, float*, float*, size_t, const float*, hipStream_t);
int pw_dense_head_train(const gct2_ctx&, int, const void*, int, const float*, const float*, const float*, float*, void*, int, float*, float*,
                        float*, float*, int, int, int, int, const float*, float*, const void*, int, int, hipStream_t);
int pw_colsum(int, const void*, int, float*, size_t, int, float, hipStream_t);
int pw_relu_mask(int, const void*, int, void*, int, size_t, int, hipStream_t);
int pw_add(int, void*, int, const void*, int, size_t, int, hipStream_t);
int pw_mix_per_image(const float*, const float*, const float*, const float*, float*, int, size_t, hipStream_t);
int conv_s1_direct(int dtype, bool dgrad, const void* x, int ldx, const void* w, const float* bias, const void* act, int ldact, void* y, int ldy,
                   int B, int H, int W, int K, int N, int KS, int relu, int accumulate, hipStream_t s);
int conv_s1_wgrad_direct(int dtype, const void* x, int ldx, const void* dz, int lddz, float* dw, int B, int H, int W, int Cin, int Cout, int KS,
                         int accumulate, hipStream_t s);
int pw_diffusion_mix(int, const float*, const float*, float, float*, void*, int, void*, int, size_t, int, hipStream_t);
int pw_diffusion_update(int, const float*, const float*, double, double, float*, float*, size_t, hipStream_t);
int pw_noise_edits(const float*, const float*, int, float*, int, int, int, hipStream_t);
int pw_image_prepare(const uint8_t*, const int64_t*, const int32_t*, float*, int, int, hipStream_t);
int pw_adam(float*, float*, float*, float*, void*, int, size_t, float, float, float, float, float, const gct2_loss_scale_state*, int, hipStream_t,
            const float* slabs = nullptr, int nslab = 0, size_t slab_stride = 0, size_t n_slab = 0);
int pw_cast(int, const float*, void*, size_t, hipStream_t);
int pw_ls_init(gct2_loss_scale_state*, float, hipStream_t);
int pw_ls_begin(gct2_loss_scale_state*, float, int, float, float, hipStream_t);
int pw_ls_check(const float*, size_t, gct2_loss_scale_state*, hipStream_t);
int pw_ls_update(gct2_loss_scale_state*, int, hipStream_t);

// the only static storage of the library, all of it per host thread: the text of the last error, and the context that stands in
// for ctx = NULL (no scratch, automatic tiles; reset at every use)
static thread_local char g_err[512] = "";
static thread_local gct2_ctx g_null_ctx{};

int gct2_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
void gct2_log(gct2_ctx& c, const char* fmt, ...) {
  if (!c.log_on) return;
  char buf[160];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  // bounded: a log left switched on for a whole run stops growing at 1 MiB and says so once
  constexpr size_t LOG_CAP = (size_t)1 << 20;
  if (c.log.size() + sizeof(buf) + 16 > LOG_CAP) {
    if (!c.log_full) { c.log += "log:truncated;"; c.log_full = true; }
    return;
  }
  c.log += buf;
  c.log += ';';
}
int gct2_check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return gct2_fail(GCT2_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return GCT2_OK;
}

namespace {
inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline bool dtype_ok(int d) { return d == GCT2_F32 || d == GCT2_BF16 || d == GCT2_F16; }
inline size_t esize(int d) { return d == GCT2_F32 ? 4 : 2; }

int check_conv_args(const char* fn, int dtype, const void* a, const void* b, const void* c, int B, int H, int W, int Cin, int Cout) {
  if (!dtype_ok(dtype)) return gct2_fail(GCT2_EINVAL, "%s: unknown dtype %d", fn, dtype);
  if (!a || !b || !c) return gct2_fail(GCT2_EINVAL, "%s: null pointer", fn);
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return gct2_fail(GCT2_EINVAL, "%s: non-positive dimension", fn);
  if ((size_t)B * H * W * 4 >= ((size_t)1 << 31)) return gct2_fail(GCT2_EINVAL, "%s: B*H*W too large for 32-bit pixel indices", fn);
  return GCT2_OK;
}
inline gct2_ctx& C(gct2_ctx* c) {
  if (c) return *c;
  g_null_ctx = gct2_ctx{};
  return g_null_ctx;
}
// The ReLU bit plane registered for THIS call (gct2_ctx_set_relu_bits is one-shot).  Every layer entry point constructs one of these
// FIRST THING - before any argument check - so that a plane never outlives the call it was registered for, whatever that call
// returns (r03 consumed it behind the checks: a rejected call leaked its plane to the next layer call).
struct PlaneTaken {
  unsigned char* bits; int ld;
  explicit PlaneTaken(gct2_ctx& c) : bits(c.relu_bits), ld(c.relu_ldbits) { c.relu_bits = nullptr; c.relu_ldbits = 0; c.relu_bits_done = 0; }
  // entry points that cannot use a plane: a pending one is a caller error
  int none(const char* fn) const {
    return bits ? gct2_fail(GCT2_EINVAL, "%s: a ReLU bit plane was registered (gct2_ctx_set_relu_bits), but this call neither writes nor reads one", fn) : GCT2_OK;
  }
  int into(const char* fn, int channels, TapGemmParams& p) const {
    if (!bits) return GCT2_OK;
    if (channels % 8 || ld < channels / 8) return gct2_fail(GCT2_EINVAL, "%s: ReLU bit plane needs channels %% 8 == 0 and ld_bytes >= channels / 8 (got %d, %d)", fn, channels, ld);
    p.bits = bits; p.ldbits = ld;
    return GCT2_OK;
  }
};
// forward calls: if the launch did not write the plane in its epilogue, derive it from the activation it stored
int finish_relu_bits(gct2_ctx& c, int dtype, const TapGemmParams& p, size_t pixels, void* stream) {
  if (!p.bits || c.relu_bits_done) return GCT2_OK;
  gct2_log(c, "relu_bits:derived");
  return pw_relu_bits(dtype, p.y, p.ldy, pixels, p.N, p.bits, p.ldbits, S(stream));
}
int run_tapgemm(gct2_ctx& c, int dtype, int form, int epi, const TapGemmParams& p, void* stream) {
  if (!c.force_direct && tapgemm_mfma_supported(dtype, p)) return tapgemm_mfma(c, dtype, form, epi, p, S(stream));
  gct2_log(c, "direct:tap");
  return tapgemm_direct(dtype, form, epi, p, S(stream));
}
// input-gradient launch with optional fused bias gradient: db (+)= column sums of the masked gradient THIS call produces
// (channels [0, split) -> db, the rest -> db2).  MFMA path: fused into the epilogue / split-K finalize.  Direct path:
// column sums of the output view after the launch, minus those before it when the launch accumulates.
int run_dgrad(gct2_ctx& c, int dtype, int form, TapGemmParams p, size_t out_pixels, float* db, int split, float* db2, int db_acc,
              void* stream) {
  if (split < 0 || split > p.N) return gct2_fail(GCT2_EINVAL, "dgrad: db_split out of range");
  p.db = db; p.db_split = split; p.db2 = db2; p.db_acc = db_acc;
  if (!c.force_direct && tapgemm_mfma_supported(dtype, p)) return tapgemm_mfma(c, dtype, form, EPI_MASK, p, S(stream));
  gct2_log(c, "direct:tap");
  // the column-sum kernels below add with atomics, at once: row sets queued for the same targets are reduced first (a queued
  // overwrite would otherwise run behind this call's add and erase it)
  if (int e = tapgemm_dbq_flush_for(c, db, split, db2, p.N - split, S(stream))) return e;
  zero_overwritten_db(p, S(stream));
  const size_t es = esize(dtype);
  auto sums = [&](float sign) -> int {
    if (db && split > 0) if (int e = pw_colsum(dtype, p.y, p.ldy, db, out_pixels, split, sign, S(stream))) return e;
    if (db2 && split < p.N)
      if (int e = pw_colsum(dtype, (const char*)p.y + (size_t)split * es, p.ldy, db2, out_pixels, p.N - split, sign, S(stream))) return e;
    return GCT2_OK;
  };
  if ((db || db2) && p.accumulate) if (int e = sums(-1.f)) return e;
  if (int e = tapgemm_direct(dtype, form, EPI_MASK, p, S(stream))) return e;
  return (db || db2) ? sums(1.f) : GCT2_OK;
}
int run_wgrad(gct2_ctx& c, int dtype, const WgradParams& p, void* stream, WgradSlabs* defer = nullptr) {
  if (!c.force_direct && wgrad_mfma_supported(dtype, p)) return wgrad_mfma(c, dtype, p, S(stream), defer);
  if (defer) *defer = WgradSlabs{nullptr, 0, 0};
  gct2_log(c, "direct:wgrad");
  return wgrad_direct(dtype, p, S(stream));
}
// bias gradient of a weight-gradient call: db (+)= column sums of dz (atomics: an overwritten target starts from zero)
int wgrad_db(gct2_ctx& c, int dtype, const void* dz, int lddz, float* db, size_t pixels, int Cout, int accumulate, void* stream) {
  if (int e = tapgemm_dbq_flush_for(c, db, Cout, nullptr, 0, S(stream))) return e;      // (an immediate writer: queued row sets of db go first)
  if (!accumulate) (void)hipMemsetAsync(db, 0, (size_t)Cout * sizeof(float), S(stream));
  return pw_colsum(dtype, dz, lddz, db, pixels, Cout, 1.f, S(stream));
}
// Keras Adam right behind a weight-gradient launch, on the same stream (gct2_adam_args): the range [p, p + n) of the caller's arenas
// starts with the layer's kernel (the engine's ranges hold the kernel only since r04: biases and Dense(3) live in one fp32 zone with
// a launch of its own); the kernel gradient comes from the slabs the launch left (never materialised) or from dw (written, not
// accumulated: no zeroing), whatever lies behind the kernel in the range from the gradient arena; nothing is zeroed
int check_adam_args(const gct2_adam_args* a, const float* dw, size_t nw) {
  if (!a->p || !a->m || !a->v) return gct2_fail(GCT2_EINVAL, "wgrad + adam: null arena pointers");
  if (a->n < nw || ((uintptr_t)a->p | (uintptr_t)a->m | (uintptr_t)a->v | (uintptr_t)dw) % 16)
    return gct2_fail(GCT2_EINVAL, "wgrad + adam: range shorter than the weight tensor or misaligned");
  return GCT2_OK;
}
int adam_after_wgrad(gct2_adam_args* a, float* dw, size_t nw, const WgradSlabs& sl, void* stream) {
  if (a->defer) {      // the caller runs the step later (gct2_adam_apply): tell it where the kernel gradient is
    a->slab_base = sl.base; a->nslab = sl.nslab; a->slab_stride = sl.stride;
    return GCT2_OK;
  }
  return pw_adam(a->p, a->m, a->v, dw, a->shadow, a->shadow_dtype, a->n, a->alpha, a->beta1, a->beta2, a->eps, a->grad_mul, nullptr, 0,
                 S(stream), sl.base, sl.nslab, sl.stride, sl.nslab ? nw : 0);
}
}  // namespace

extern "C" {

int gct2_abi_version(void) { return 17; }
int gct2_build_flags(void) {
#ifdef GCT2_STAMP
  return GCT2_BUILD_STAMP;
#else
  return 0;
#endif
}
const char* gct2_last_error(void) { return g_err; }

int gct2_ctx_create(gct2_ctx** ctx) {
  if (!ctx) return gct2_fail(GCT2_EINVAL, "ctx_create: null output pointer");
  *ctx = new (std::nothrow) gct2_ctx();
  return *ctx ? GCT2_OK : gct2_fail(GCT2_EINVAL, "ctx_create: out of host memory");
}
int gct2_ctx_destroy(gct2_ctx* ctx) {
  delete ctx;
  return GCT2_OK;
}
int gct2_ctx_set_workspace(gct2_ctx* ctx, void* ws, size_t bytes) {
  if (!ctx) return gct2_fail(GCT2_EINVAL, "ctx_set_workspace: null ctx");
  if (ws && ((uintptr_t)ws % 16)) return gct2_fail(GCT2_EINVAL, "ctx_set_workspace: pointer must be 16-byte aligned");
  ctx->ws = ws ? reinterpret_cast<float*>(ws) : nullptr;
  ctx->ws_bytes = ws ? bytes : 0;
  return GCT2_OK;
}
int gct2_ctx_set_wgrad_workspace(gct2_ctx* ctx, void* ws, size_t bytes) {
  if (!ctx) return gct2_fail(GCT2_EINVAL, "ctx_set_wgrad_workspace: null ctx");
  if (ws && ((uintptr_t)ws % 16)) return gct2_fail(GCT2_EINVAL, "ctx_set_wgrad_workspace: pointer must be 16-byte aligned");
  ctx->wws = ws ? reinterpret_cast<float*>(ws) : nullptr;
  ctx->wws_bytes = ws ? bytes : 0;
  return GCT2_OK;
}
int gct2_ctx_set_bias_queue(gct2_ctx* ctx, void* buf, size_t bytes) {
  if (!ctx) return gct2_fail(GCT2_EINVAL, "ctx_set_bias_queue: null ctx");
  if (buf && ((uintptr_t)buf % 16)) return gct2_fail(GCT2_EINVAL, "ctx_set_bias_queue: pointer must be 16-byte aligned");
  ctx->dbq_jobs.clear();                     // (row sets recorded and not flushed are dropped: flush before changing the buffer)
  ctx->dbq_used = 0;
  ctx->dbq = buf ? reinterpret_cast<float*>(buf) : nullptr;
  ctx->dbq_floats = buf ? bytes / sizeof(float) : 0;
  return GCT2_OK;
}
int gct2_bias_queue_flush(gct2_ctx* ctx, void* stream) {
  if (!ctx) return gct2_fail(GCT2_EINVAL, "bias_queue_flush: null ctx");
  return tapgemm_dbq_flush(*ctx, S(stream));
}
int gct2_ctx_set_tuning(gct2_ctx* ctx, int v) {
  if (!ctx) return gct2_fail(GCT2_EINVAL, "ctx_set_tuning: null ctx");
  const int tap = v & 0xff, wg = (v >> 16) & 0xff, known = 0x1ff | (0xff << 16) | (0x7f << 24);
  if ((v & ~known) || (tap != 0 && tap != 2 && tap != 5) || (wg != 0 && wg != 2 && wg != 3 && wg != 4 && wg != 5 && wg != 6 && wg != 7) || ((v >> 24) & 3) == 3 || ((v >> 26) & 3) == 3)
    return gct2_fail(GCT2_EINVAL, "ctx_set_tuning: unknown tuning word 0x%x (include/gct2.h)", v);
  ctx->tap_variant = tap;
  ctx->wgrad_variant = wg;
  ctx->no_splitk = (v >> 8) & 1;
  ctx->halo_mode = (v >> 24) & 3;
  ctx->xcd_order = (v >> 26) & 3;
  ctx->wgrad_split = (v >> 28) & 7;
  return GCT2_OK;
}
int gct2_ctx_set_relu_bits(gct2_ctx* ctx, void* bits, int ld_bytes) {
  if (!ctx) return gct2_fail(GCT2_EINVAL, "ctx_set_relu_bits: null ctx");
  if (bits && ld_bytes <= 0) return gct2_fail(GCT2_EINVAL, "ctx_set_relu_bits: ld_bytes must be positive");
  ctx->relu_bits = reinterpret_cast<unsigned char*>(bits);
  ctx->relu_ldbits = bits ? ld_bytes : 0;
  return GCT2_OK;
}
int gct2_ctx_set_stamp_buffer(gct2_ctx* ctx, void* stamps, size_t bytes) {
  if (!ctx) return gct2_fail(GCT2_EINVAL, "ctx_set_stamp_buffer: null ctx");
#ifdef GCT2_STAMP
  if (stamps && ((uintptr_t)stamps % 8)) return gct2_fail(GCT2_EINVAL, "ctx_set_stamp_buffer: pointer must be 8-byte aligned");
  ctx->stamps = reinterpret_cast<unsigned long long*>(stamps);
  ctx->stamps_bytes = stamps ? bytes : 0;
  return GCT2_OK;
#else
  (void)stamps; (void)bytes;
  return gct2_fail(GCT2_EINVAL, "ctx_set_stamp_buffer: this is the product build (rebuild with make EXTRA=-DGCT2_STAMP for in-kernel stamps)");
#endif
}
int gct2_ctx_log_launches(gct2_ctx* ctx, int on) {
  if (!ctx) return gct2_fail(GCT2_EINVAL, "ctx_log_launches: null ctx");
  ctx->log_on = on != 0;
  ctx->log.clear();
  ctx->log_full = false;
  return GCT2_OK;
}
int gct2_ctx_read_launch_log(gct2_ctx* ctx, char* buf, size_t bytes, size_t* needed) {
  if (!ctx) return gct2_fail(GCT2_EINVAL, "ctx_read_launch_log: null ctx");
  const size_t need = ctx->log.size() + 1;
  if (needed) *needed = need;
  if (!buf || bytes < need)       // nothing is copied and nothing is cleared: call again with `needed` bytes
    return gct2_fail(GCT2_EINVAL, "ctx_read_launch_log: the log holds %zu bytes, the buffer %zu", need, buf ? bytes : (size_t)0);
  memcpy(buf, ctx->log.data(), need - 1);
  buf[need - 1] = 0;
  ctx->log.clear();
  ctx->log_full = false;
  return GCT2_OK;
}
int gct2_ctx_force_direct(gct2_ctx* ctx, int on) {
  if (!ctx) return gct2_fail(GCT2_EINVAL, "ctx_force_direct: null ctx");
  ctx->force_direct = on ? 1 : 0;
  return GCT2_OK;
}

int gct2_device_check(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return gct2_fail(GCT2_ENODEV, "no HIP device");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return gct2_fail(GCT2_ENODEV, "hipGetDeviceProperties failed");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return gct2_fail(GCT2_ENODEV, "device is %s, this library is built for gfx950", prop.gcnArchName);
  return GCT2_OK;
}

int gct2_stream_occupy(void* stream, int workgroups, double microseconds) {
  if (workgroups < 1 || workgroups > 1024 || !(microseconds >= 0.0) || microseconds > 20000.0)
    return gct2_fail(GCT2_EINVAL, "stream_occupy: 1..1024 work-groups for 0..20000 us (got %d, %g)", workgroups, microseconds);
  return pw_occupy(workgroups, (unsigned long long)(microseconds * 100.0), S(stream));
}

int gct2_conv4s2_fwd(gct2_ctx* ctx, int dtype, const void* x, int ldx, const void* w, const float* bias, void* y, int ldy, int B, int H, int W,
                     int Cin, int Cout, int relu, void* stream) {
  gct2_ctx& c = C(ctx);
  const PlaneTaken plane(c);
  if (int e = check_conv_args("conv4s2_fwd", dtype, x, w, y, B, H, W, Cin, Cout)) return e;
  if ((H & 1) || (W & 1)) return gct2_fail(GCT2_EINVAL, "conv4s2_fwd: H=%d W=%d must be even (skip concat, train.py:114-119)", H, W);
  if (ldx < Cin || ldy < Cout) return gct2_fail(GCT2_EINVAL, "conv4s2_fwd: ld smaller than channel count");
  TapGemmParams p{x, ldx, w, bias, nullptr, 0, y, ldy, B, H / 2, W / 2, Cin, Cout, relu, 0};
  if (int e = plane.into("conv4s2_fwd", Cout, p)) return e;
  const size_t out_pixels = (size_t)B * (H / 2) * (W / 2);
  if (!c.force_direct && rgb_fwd_supported(dtype, p)) {                                       // image layer (Cin <= 4)
    gct2_log(c, "rgb:fwd");
    if (int e = rgb_fwd(dtype, p, S(stream))) return e;
    if (p.bits && rgb_fwd_writes_bits(p)) c.relu_bits_done = 1;
    return finish_relu_bits(c, dtype, p, out_pixels, stream);
  }
  if (int e = run_tapgemm(c, dtype, FORM_CONV, EPI_BIAS_ACT, p, stream)) return e;
  return finish_relu_bits(c, dtype, p, out_pixels, stream);
}

int gct2_conv4s2_dgrad(gct2_ctx* ctx, int dtype, const void* dz, int lddz, const void* w, const void* act, int ldact, void* dx, int lddx, int B,
                       int H, int W, int Cin, int Cout, int accumulate, float* db, int db_split, float* db2, int db_accumulate, void* stream) {
  gct2_ctx& c = C(ctx);
  const PlaneTaken plane(c);
  if (int e = check_conv_args("conv4s2_dgrad", dtype, dz, w, dx, B, H, W, Cin, Cout)) return e;
  if ((H & 1) || (W & 1)) return gct2_fail(GCT2_EINVAL, "conv4s2_dgrad: H=%d W=%d must be even", H, W);
  if (lddz < Cout || lddx < Cin || (act && ldact < Cin)) return gct2_fail(GCT2_EINVAL, "conv4s2_dgrad: ld smaller than channel count");
  TapGemmParams p{dz, lddz, w, nullptr, act, ldact, dx, lddx, B, H / 2, W / 2, Cout, Cin, 0, accumulate};
  if (int e = plane.into("conv4s2_dgrad", Cin, p)) return e;
  if (!act) p.bits = nullptr;                                   // the plane stands in for act: no mask asked for, none applied
  return run_dgrad(c, dtype, FORM_CONVT, p, (size_t)B * H * W, db, db_split, db2, db_accumulate, stream);
}

int gct2_conv4s2_wgrad(gct2_ctx* ctx, int dtype, const void* x, int ldx, const void* dz, int lddz, float* dw, float* db, int B, int H, int W,
                       int Cin, int Cout, int accumulate, gct2_adam_args* adam, void* stream) {
  gct2_ctx& c = C(ctx);
  if (int e = PlaneTaken(c).none("conv4s2_wgrad")) return e;
  if (int e = check_conv_args("conv4s2_wgrad", dtype, x, dz, dw, B, H, W, Cin, Cout)) return e;
  if ((H & 1) || (W & 1)) return gct2_fail(GCT2_EINVAL, "conv4s2_wgrad: H=%d W=%d must be even", H, W);
  if (ldx < Cin || lddz < Cout) return gct2_fail(GCT2_EINVAL, "conv4s2_wgrad: ld smaller than channel count");
  if (adam && accumulate) return gct2_fail(GCT2_EINVAL, "conv4s2_wgrad: the fused optimizer step needs accumulate = 0");
  WgradParams p{x, ldx, dz, lddz, dw, B, H / 2, W / 2, Cin, Cout, 1};
  p.accumulate = accumulate ? 1 : 0;
  WgradSlabs sl{nullptr, 0, 0};
  if (adam) if (int e = check_adam_args(adam, dw, (size_t)16 * Cin * Cout)) return e;
  if (!c.force_direct && rgb_wgrad_supported(dtype, p)) {
    gct2_log(c, "rgb:wgrad");
    if (int e = rgb_wgrad(c, dtype, p, S(stream), adam ? &sl : nullptr)) return e;
  } else if (int e = run_wgrad(c, dtype, p, stream, adam ? &sl : nullptr)) return e;
  if (db)
    if (int e = wgrad_db(c, dtype, dz, lddz, db, (size_t)B * (H / 2) * (W / 2), Cout, accumulate, stream)) return e;
  if (adam) return adam_after_wgrad(adam, dw, (size_t)16 * Cin * Cout, sl, stream);
  return GCT2_OK;
}

int gct2_convT4s2_fwd(gct2_ctx* ctx, int dtype, const void* x, int ldx, const void* w, const float* bias, void* y, int ldy, int B, int H, int W,
                      int Cin, int Cout, int relu, void* stream) {
  gct2_ctx& c = C(ctx);
  const PlaneTaken plane(c);
  if (int e = check_conv_args("convT4s2_fwd", dtype, x, w, y, B, 2 * H, 2 * W, Cin, Cout)) return e;
  if (ldx < Cin || ldy < Cout) return gct2_fail(GCT2_EINVAL, "convT4s2_fwd: ld smaller than channel count");
  TapGemmParams p{x, ldx, w, bias, nullptr, 0, y, ldy, B, H, W, Cin, Cout, relu, 0};
  if (int e = plane.into("convT4s2_fwd", Cout, p)) return e;
  if (int e = run_tapgemm(c, dtype, FORM_CONVT, EPI_BIAS_ACT, p, stream)) return e;
  return finish_relu_bits(c, dtype, p, (size_t)B * 4 * H * W, stream);
}

int gct2_convT4s2_fwd_head_train(gct2_ctx* ctx, int dtype, const void* x, int ldx, const void* w, const float* bias, const float* head_w,
                                 const float* head_b, const float* target, float* pred, void* dy, int lddy, float* head_dw, float* head_db,
                                 float* loss, int B, int H, int W, int Cin, int Cout, int head_Cin, int head_Cout,
                                 const float* loss_scale_ptr, float* db, const void* x2, int ldx2, int accumulate, void* stream) {
  gct2_ctx& c = C(ctx);
  if (int e = PlaneTaken(c).none("convT4s2_fwd_head_train")) return e;
  if (int e = check_conv_args("convT4s2_fwd_head_train", dtype, x, w, dy, B, 2 * H, 2 * W, Cin, Cout)) return e;
  if (!head_w || !target || !head_dw || !loss) return gct2_fail(GCT2_EINVAL, "convT4s2_fwd_head_train: null pointer");
  if (ldx < Cin || lddy < Cout) return gct2_fail(GCT2_EINVAL, "convT4s2_fwd_head_train: ld smaller than channel count");
  const int nimg = head_Cin - Cout;
  if (head_Cout < 1 || head_Cout > 3 || nimg < 0 || nimg > 3 || (nimg > 0 && (!x2 || ldx2 < 4 || ldx2 % 4 || (uintptr_t)x2 % 8)))
    return gct2_fail(GCT2_EINVAL, "convT4s2_fwd_head_train: head needs <= 3 outputs and <= 3 image channels in a packed x2 (8-byte rows)");
  TapGemmParams p{x, ldx, w, bias, nullptr, 0, dy, lddy, B, H, W, Cin, Cout, 1, 0};
  p.head = HeadFuse{head_w, head_b, target, pred, x2, ldx2, nullptr, loss_scale_ptr, head_Cin, head_Cout,
                    (float)((double)B * 2 * H * 2 * W * head_Cout)};
  if (c.force_direct || !halo_head_supported(c, dtype, p))
    return gct2_fail(GCT2_EINVAL, "convT4s2_fwd_head_train: needs a 16-bit dtype, Cout = 64, H and W multiples of 16, 16-byte aligned views "
                                  "and a ctx workspace of B*(H/16)*(W/16)*288 floats");
  return halo_head(c, dtype, p, head_dw, head_db, loss, db, accumulate, S(stream));
}

int gct2_convT4s2_dgrad(gct2_ctx* ctx, int dtype, const void* dz, int lddz, const void* w, const void* act, int ldact, void* dx, int lddx, int B,
                        int H, int W, int Cin, int Cout, int accumulate, float* db, int db_split, float* db2, int db_accumulate, void* stream) {
  gct2_ctx& c = C(ctx);
  const PlaneTaken plane(c);
  if (int e = check_conv_args("convT4s2_dgrad", dtype, dz, w, dx, B, 2 * H, 2 * W, Cin, Cout)) return e;
  if (lddz < Cout || lddx < Cin || (act && ldact < Cin)) return gct2_fail(GCT2_EINVAL, "convT4s2_dgrad: ld smaller than channel count");
  TapGemmParams p{dz, lddz, w, nullptr, act, ldact, dx, lddx, B, H, W, Cout, Cin, 0, accumulate};
  if (int e = plane.into("convT4s2_dgrad", Cin, p)) return e;
  if (!act) p.bits = nullptr;
  return run_dgrad(c, dtype, FORM_CONV, p, (size_t)B * H * W, db, db_split, db2, db_accumulate, stream);
}

int gct2_convT4s2_wgrad(gct2_ctx* ctx, int dtype, const void* x, int ldx, const void* dz, int lddz, float* dw, float* db, int B, int H, int W,
                        int Cin, int Cout, int accumulate, gct2_adam_args* adam, void* stream) {
  gct2_ctx& c = C(ctx);
  if (int e = PlaneTaken(c).none("convT4s2_wgrad")) return e;
  if (int e = check_conv_args("convT4s2_wgrad", dtype, x, dz, dw, B, 2 * H, 2 * W, Cin, Cout)) return e;
  if (ldx < Cin || lddz < Cout) return gct2_fail(GCT2_EINVAL, "convT4s2_wgrad: ld smaller than channel count");
  if (adam && accumulate) return gct2_fail(GCT2_EINVAL, "convT4s2_wgrad: the fused optimizer step needs accumulate = 0");
  WgradParams p{dz, lddz, x, ldx, dw, B, H, W, Cout, Cin, 1};
  p.accumulate = accumulate ? 1 : 0;
  WgradSlabs sl{nullptr, 0, 0};
  if (adam) if (int e = check_adam_args(adam, dw, (size_t)16 * Cin * Cout)) return e;
  if (int e = run_wgrad(c, dtype, p, stream, adam ? &sl : nullptr)) return e;
  if (db)
    if (int e = wgrad_db(c, dtype, dz, lddz, db, (size_t)B * (2 * H) * (2 * W), Cout, accumulate, stream)) return e;
  if (adam) return adam_after_wgrad(adam, dw, (size_t)16 * Cin * Cout, sl, stream);
  return GCT2_OK;
}

// ---- off-by-default model variants (train.py:20 block_depth, train.py:26 residual, train.py:29-32 targets) ----------------------
static int check_s1(const char* fn, int dtype, const void* a, const void* b, const void* c, int B, int H, int W, int Cin, int Cout, int KS) {
  if (int e = check_conv_args(fn, dtype, a, b, c, B, H, W, Cin, Cout)) return e;
  if (KS < 1 || KS > 7 || !(KS & 1)) return gct2_fail(GCT2_EINVAL, "%s: kernel size %d (odd, 1..7: 'same' padding is symmetric then)", fn, KS);
  return GCT2_OK;
}
int gct2_conv2d_s1_fwd(gct2_ctx* ctx, int dtype, const void* x, int ldx, const void* w, const float* bias, void* y, int ldy, int B, int H, int W,
                       int Cin, int Cout, int KS, int relu, void* stream) {
  gct2_ctx& c = C(ctx);
  if (int e = PlaneTaken(c).none("conv2d_s1_fwd")) return e;
  if (int e = check_s1("conv2d_s1_fwd", dtype, x, w, y, B, H, W, Cin, Cout, KS)) return e;
  if (ldx < Cin || ldy < Cout) return gct2_fail(GCT2_EINVAL, "conv2d_s1_fwd: ld smaller than channel count");
  {   // matrix-core form (third tap-GEMM form: ks x ks taps on one grid) where the 16-bit layouts allow it, else one thread per output
    TapGemmParams p{x, ldx, w, bias, nullptr, 0, y, ldy, B, H, W, Cin, Cout, relu, 0};
    p.ks = KS;
    if (!c.force_direct && KS <= 5 && tapgemm_mfma_supported(dtype, p)) return tapgemm_mfma(c, dtype, FORM_S1, EPI_BIAS_ACT, p, S(stream));
  }
  return conv_s1_direct(dtype, false, x, ldx, w, bias, nullptr, 0, y, ldy, B, H, W, Cin, Cout, KS, relu, 0, S(stream));
}
int gct2_conv2d_s1_dgrad(gct2_ctx* ctx, int dtype, const void* dz, int lddz, const void* w, const void* act, int ldact, void* dx, int lddx, int B,
                         int H, int W, int Cin, int Cout, int KS, int accumulate, void* stream) {
  gct2_ctx& c = C(ctx);
  if (int e = PlaneTaken(c).none("conv2d_s1_dgrad")) return e;
  if (int e = check_s1("conv2d_s1_dgrad", dtype, dz, w, dx, B, H, W, Cin, Cout, KS)) return e;
  if (lddz < Cout || lddx < Cin || (act && ldact < Cin)) return gct2_fail(GCT2_EINVAL, "conv2d_s1_dgrad: ld smaller than channel count");
  {
    TapGemmParams p{dz, lddz, w, nullptr, act, ldact, dx, lddx, B, H, W, Cout, Cin, 0, accumulate};
    p.ks = KS;
    if (!c.force_direct && KS <= 5 && tapgemm_mfma_supported(dtype, p)) return tapgemm_mfma(c, dtype, FORM_S1T, EPI_MASK, p, S(stream));
  }
  return conv_s1_direct(dtype, true, dz, lddz, w, nullptr, act, ldact, dx, lddx, B, H, W, Cout, Cin, KS, 0, accumulate, S(stream));
}
int gct2_conv2d_s1_wgrad(gct2_ctx* ctx, int dtype, const void* x, int ldx, const void* dz, int lddz, float* dw, float* db, int B, int H, int W,
                         int Cin, int Cout, int KS, int accumulate, void* stream) {
  gct2_ctx& c = C(ctx);
  if (int e = PlaneTaken(c).none("conv2d_s1_wgrad")) return e;
  if (int e = check_s1("conv2d_s1_wgrad", dtype, x, dz, dw, B, H, W, Cin, Cout, KS)) return e;
  if (ldx < Cin || lddz < Cout) return gct2_fail(GCT2_EINVAL, "conv2d_s1_wgrad: ld smaller than channel count");
  WgradParams p{x, ldx, dz, lddz, dw, B, H, W, Cin, Cout, 1};
  p.accumulate = accumulate ? 1 : 0;
  p.ks = KS;
  if (!c.force_direct && wgrad_mfma_supported(dtype, p)) {
    if (int e = wgrad_mfma(c, dtype, p, S(stream), nullptr)) return e;
  } else if (int e = conv_s1_wgrad_direct(dtype, x, ldx, dz, lddz, dw, B, H, W, Cin, Cout, KS, accumulate, S(stream))) return e;
  if (db) return wgrad_db(c, dtype, dz, lddz, db, (size_t)B * H * W, Cout, accumulate, stream);
  return GCT2_OK;
}
int gct2_relu_mask(int dtype, const void* act, int ldact, void* d, int ldd, size_t npix, int C, void* stream) {
  if (!dtype_ok(dtype) || !act || !d || C <= 0 || ldact < C || ldd < C) return gct2_fail(GCT2_EINVAL, "relu_mask: bad dtype, null pointer or ld < C");
  if (npix == 0) return GCT2_OK;
  return pw_relu_mask(dtype, act, ldact, d, ldd, npix, C, S(stream));
}
int gct2_add(int dtype, void* dst, int lddst, const void* src, int ldsrc, size_t npix, int C, void* stream) {
  if (!dtype_ok(dtype) || !dst || !src || C <= 0 || lddst < C || ldsrc < C) return gct2_fail(GCT2_EINVAL, "add: bad dtype, null pointer or ld < C");
  if (npix == 0) return GCT2_OK;
  return pw_add(dtype, dst, lddst, src, ldsrc, npix, C, S(stream));
}
int gct2_mix_per_image(const float* x, const float* eps, const float* a, const float* c, float* out, int B, size_t per_image, void* stream) {
  if (!x || !a || !out || (eps && !c) || B <= 0 || per_image == 0) return gct2_fail(GCT2_EINVAL, "mix_per_image: null pointer or empty batch");
  return pw_mix_per_image(x, eps, a, c, out, B, per_image, S(stream));
}

int gct2_dense_fwd(int dtype, const void* x, int ldx, const float* w, const float* b, float* y, int M, int Cin, int Cout, void* stream) {
  if (!dtype_ok(dtype) || !x || !w || !y) return gct2_fail(GCT2_EINVAL, "dense_fwd: bad dtype or null pointer");
  if (M <= 0 || Cin <= 0 || Cout <= 0 || Cout > 4 || ldx < Cin) return gct2_fail(GCT2_EINVAL, "dense_fwd: bad shape (Cout must be 1..4)");
  return pw_dense_fwd(dtype, x, ldx, w, b, y, M, Cin, Cout, S(stream));
}

int gct2_dense_bwd(int dtype, const void* x, int ldx, const float* w, const float* dy, void* dx, int lddx, float* dw, float* db, int M,
                   int Cin, int Cout, int Cmask, int accumulate, void* stream) {
  if (!dtype_ok(dtype) || !x || !w || !dy || !dx || !dw) return gct2_fail(GCT2_EINVAL, "dense_bwd: bad dtype or null pointer");
  if (M <= 0 || Cin <= 0 || Cout <= 0 || Cout > 4 || ldx < Cin || Cmask < 0 || Cmask > Cin || lddx < Cmask)
    return gct2_fail(GCT2_EINVAL, "dense_bwd: bad shape");
  if ((Cin + 1) * Cout > 2048) return gct2_fail(GCT2_EINVAL, "dense_bwd: (Cin+1)*Cout = %d exceeds 2048", (Cin + 1) * Cout);
  if ((size_t)Cin * 16 + 128 * 16 + (size_t)128 * Cin * esize(dtype) > 160 * 1024)
    return gct2_fail(GCT2_EINVAL, "dense_bwd: Cin=%d too large for the LDS tile", Cin);
  return pw_dense_bwd(dtype, x, ldx, w, dy, dx, lddx, dw, db, M, Cin, Cout, Cmask, accumulate, S(stream));
}

int gct2_dense_head_train(gct2_ctx* ctx, int dtype, const void* x, int ldx, const float* w, const float* b, const float* target, float* pred,
                          void* dx, int lddx, float* dw, float* db, float* loss, float* partials, int M, int Cin, int Cout, int Cmask,
                          const float* loss_scale_ptr, float* db_dx, const void* x2, int ldx2, int accumulate, void* stream) {
  gct2_ctx& c = C(ctx);
  if (int e = PlaneTaken(c).none("dense_head_train")) return e;
  if ((dtype != GCT2_BF16 && dtype != GCT2_F16) || !x || !w || !target || !dx || !dw || !loss || !partials)
    return gct2_fail(GCT2_EINVAL, "dense_head_train: 16-bit dtypes only / null pointer (use dense_fwd + mse_fwd_bwd + dense_bwd)");
  if (x2 && (ldx2 < Cin - Cmask || ldx2 % 4 || (uintptr_t)x2 % 8 || Cin - Cmask > 4))
    return gct2_fail(GCT2_EINVAL, "dense_head_train: x2 holds at most 4 channels per pixel in 8-byte aligned rows (ldx2 multiple of 4)");
  if (M <= 0 || Cin <= 0 || Cout <= 0 || Cout > 4 || ldx < (x2 ? Cmask : Cin) || ldx % 8 || lddx % 8 || Cmask % 8 || Cmask <= 0 || Cmask > Cin ||
      lddx < Cmask || (Cin + 1) * Cout > 256)
    return gct2_fail(GCT2_EINVAL, "dense_head_train: bad shape (need ldx, lddx, Cmask multiples of 8, (Cin+1)*Cout <= 256)");
  if ((uintptr_t)x % 16 || (uintptr_t)dx % 16) return gct2_fail(GCT2_EINVAL, "dense_head_train: views must be 16-byte aligned");
  if ((size_t)256 * ldx * 2 + (size_t)256 * Cmask * 2 + 256 * 16 + (size_t)ldx * 16 > 160 * 1024)
    return gct2_fail(GCT2_EINVAL, "dense_head_train: ldx=%d too large for the LDS tile", ldx);
  return pw_dense_head_train(c, dtype, x, ldx, w, b, target, pred, dx, lddx, dw, db, loss, partials, M, Cin, Cout, Cmask, loss_scale_ptr,
                             db_dx, x2, ldx2, accumulate, S(stream));
}

int gct2_rng_uniform_int(uint64_t seed, uint64_t stream_id, uint64_t offset, int32_t* out, size_t n, int lo, int hi, void* stream) {
  if (!out || hi < lo) return gct2_fail(GCT2_EINVAL, "rng_uniform_int: null output or empty range");
  return pw_rng_uniform_int(seed, stream_id, offset, out, n, lo, hi, S(stream));
}
int gct2_rng_normal(uint64_t seed, uint64_t stream_id, uint64_t offset, float* out, size_t n, void* stream) {
  if (!out) return gct2_fail(GCT2_EINVAL, "rng_normal: null output");
  return pw_rng_normal(seed, stream_id, offset, out, n, S(stream));
}

int gct2_noise_image(int dtype, const float* x, const int32_t* t_int, const float* eps, void* out, int ldout, void* out2, int ldout2,
                     int B, int HW, int C, int steps, void* stream) {
  if (!dtype_ok(dtype) || !x || !t_int || !eps || !out) return gct2_fail(GCT2_EINVAL, "noise_image: bad dtype or null pointer");
  if (B <= 0 || HW <= 0 || C <= 0 || ldout < C || (out2 && ldout2 < C) || steps <= 0) return gct2_fail(GCT2_EINVAL, "noise_image: bad shape");
  return pw_noise(dtype, x, t_int, eps, out, ldout, out2, ldout2, B, HW, C, steps, S(stream));
}

int gct2_noise_image_rng(int dtype, const float* x, const int32_t* t_int, uint64_t seed, uint64_t stream_id, uint64_t offset,
                         float* eps_out, void* out, int ldout, void* out2, int ldout2, int B, int HW, int C, int steps, void* stream) {
  if (!dtype_ok(dtype) || !x || !t_int || !out) return gct2_fail(GCT2_EINVAL, "noise_image_rng: bad dtype or null pointer");
  if (B <= 0 || HW <= 0 || C <= 0 || ldout < C || (out2 && ldout2 < C) || steps <= 0 || (size_t)B * HW * C >= ((size_t)1 << 31))
    return gct2_fail(GCT2_EINVAL, "noise_image_rng: bad shape");
  return pw_noise_rng(dtype, x, t_int, seed, stream_id, offset, eps_out, out, ldout, out2, ldout2, B, HW, C, steps, S(stream));
}

int gct2_diffusion_mix(int dtype, const float* x_theta, const float* eps_theta, float alpha, float* fake, void* out, int ldout, void* out2,
                       int ldout2, size_t npix, int C, void* stream) {
  if (!dtype_ok(dtype) || !x_theta || !eps_theta || !fake || !out) return gct2_fail(GCT2_EINVAL, "diffusion_mix: bad dtype or null pointer");
  if (npix == 0 || C <= 0 || ldout < C || (out2 && ldout2 < C) || !(alpha >= 0.f && alpha <= 1.f))
    return gct2_fail(GCT2_EINVAL, "diffusion_mix: bad shape or alpha outside [0, 1]");
  return pw_diffusion_mix(dtype, x_theta, eps_theta, alpha, fake, out, ldout, out2, ldout2, npix, C, S(stream));
}

int gct2_diffusion_update(int mode, const float* pred, const float* fake, double alpha, double alpha_prev, float* x_theta, float* eps_theta,
                          size_t n, void* stream) {
  if (mode < GCT2_SAMPLE_X || mode > GCT2_SAMPLE_ODE) return gct2_fail(GCT2_EINVAL, "diffusion_update: unknown mode %d", mode);
  if (!pred || !fake || !x_theta || (!eps_theta && mode != GCT2_SAMPLE_ODE) || n == 0)
    return gct2_fail(GCT2_EINVAL, "diffusion_update: null pointer or n == 0");
  if (!(alpha >= 0. && alpha < 1.)) return gct2_fail(GCT2_EINVAL, "diffusion_update: alpha must be in [0, 1)");
  if (mode != GCT2_SAMPLE_X && !(alpha > 0.)) return gct2_fail(GCT2_EINVAL, "diffusion_update: this mode divides by sqrt(alpha): alpha must be > 0");
  if (mode == GCT2_SAMPLE_ODE) {
    if (!(alpha_prev >= 0. && alpha_prev <= 1.)) return gct2_fail(GCT2_EINVAL, "diffusion_update: alpha_prev must be in [0, 1]");
    if (sqrt(alpha_prev) * sqrt(1. - alpha) - sqrt(alpha) * sqrt(1. - alpha_prev) == 0.)
      return gct2_fail(GCT2_EINVAL, "diffusion_update: alpha == alpha_prev makes the ODE step singular (train.py:386-391)");
  }
  return pw_diffusion_update(mode, pred, fake, alpha, alpha_prev, x_theta, eps_theta, n, S(stream));
}

int gct2_noise_edits(const float* eps, const float* dictionary, int K, float* out, int H, int W, int C, void* stream) {
  if (!eps || !dictionary || !out) return gct2_fail(GCT2_EINVAL, "noise_edits: null pointer");
  if (H <= 0 || W <= 0 || C <= 0 || K <= 0 || (H & 3) || (W & 3))
    return gct2_fail(GCT2_EINVAL, "noise_edits: H=%d W=%d must be positive multiples of 4 (avg_pool2d(4, 4, 'SAME') without padding)", H, W);
  return pw_noise_edits(eps, dictionary, K, out, H, W, C, S(stream));
}

int gct2_image_prepare(const uint8_t* src, const int64_t* offsets, const int32_t* dims, float* dst, int B, int size, void* stream) {
  if (!src || !offsets || !dims || !dst || B <= 0 || size <= 0) return gct2_fail(GCT2_EINVAL, "image_prepare: null pointer or empty batch");
  return pw_image_prepare(src, offsets, dims, dst, B, size, S(stream));
}

int gct2_mse_fwd_bwd(const float* pred, const float* target, float* dpred, float* loss, float* partials, size_t n,
                     const float* loss_scale_ptr, void* stream) {
  if (!pred || !target || !dpred || !loss || !partials || n == 0) return gct2_fail(GCT2_EINVAL, "mse_fwd_bwd: null pointer or n == 0");
  return pw_mse(pred, target, dpred, loss, partials, n, loss_scale_ptr, S(stream));
}

int gct2_adam_keras_multi(float* p, float* m, float* v, float* g, void* shadow, int shadow_dtype, size_t n, float alpha, float beta1,
                          float beta2, float eps, float grad_mul, const gct2_loss_scale_state* ls, int zero_grad, void* stream) {
  if (!p || !m || !v || !g) return gct2_fail(GCT2_EINVAL, "adam_keras_multi: null pointer");
  if (shadow && !dtype_ok(shadow_dtype)) return gct2_fail(GCT2_EINVAL, "adam_keras_multi: bad shadow dtype");
  if (((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (uintptr_t)g) % 16 || (shadow && (uintptr_t)shadow % 8))
    return gct2_fail(GCT2_EINVAL, "adam_keras_multi: arenas must be 16-byte aligned");
  return pw_adam(p, m, v, g, shadow, shadow_dtype, n, alpha, beta1, beta2, eps, grad_mul, ls, zero_grad, S(stream));
}

int gct2_adam_apply(const gct2_adam_args* a, float* dw, size_t nw, void* stream) {
  if (!a || !dw) return gct2_fail(GCT2_EINVAL, "adam_apply: null pointer");
  if (int e = check_adam_args(a, dw, nw)) return e;
  if (a->nslab < 0 || (a->nslab > 0 && (!a->slab_base || a->slab_stride < nw))) return gct2_fail(GCT2_EINVAL, "adam_apply: bad slab description");
  return pw_adam(a->p, a->m, a->v, dw, a->shadow, a->shadow_dtype, a->n, a->alpha, a->beta1, a->beta2, a->eps, a->grad_mul, nullptr, 0,
                 S(stream), a->slab_base, a->nslab, a->slab_stride, a->nslab ? nw : 0);
}

int gct2_cast_from_f32(int dtype, const float* src, void* dst, size_t n, void* stream) {
  if (!dtype_ok(dtype) || !src || !dst) return gct2_fail(GCT2_EINVAL, "cast_from_f32: bad dtype or null pointer");
  return pw_cast(dtype, src, dst, n, S(stream));
}

int gct2_loss_scale_init(gct2_loss_scale_state* st, float initial_scale, void* stream) {
  if (!st || !(initial_scale > 0.f)) return gct2_fail(GCT2_EINVAL, "loss_scale_init: null state or non-positive scale");
  return pw_ls_init(st, initial_scale, S(stream));
}
int gct2_loss_scale_begin(gct2_loss_scale_state* st, float base_lr, int warmup_steps, float beta1, float beta2, void* stream) {
  if (!st || warmup_steps < 0) return gct2_fail(GCT2_EINVAL, "loss_scale_begin: null state or negative warm-up");
  return pw_ls_begin(st, base_lr, warmup_steps, beta1, beta2, S(stream));
}
int gct2_scale_check_finite(const float* g, size_t n, gct2_loss_scale_state* st, void* stream) {
  if (!g || !st) return gct2_fail(GCT2_EINVAL, "scale_check_finite: null pointer");
  return pw_ls_check(g, n, st, S(stream));
}
int gct2_loss_scale_update(gct2_loss_scale_state* st, int growth_interval, void* stream) {
  if (!st || growth_interval <= 0) return gct2_fail(GCT2_EINVAL, "loss_scale_update: null state or bad interval");
  return pw_ls_update(st, growth_interval, S(stream));
}

}  // extern "C"
