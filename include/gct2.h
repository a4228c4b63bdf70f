/*
 * gct2.h - C ABI of the MI355X (gfx950) train-step kernels for relgukxilef/GAN-Class-Transfer2.
 *
 * The reference (/root/reference/train.py) has no native code and no FFI: every device op is a
 * TensorFlow/Keras call.  Each entry point below replaces one such call site (cited per function,
 * file:line into /root/reference/train.py) and is what a ctypes / cffi binding of the reference's
 * Python host code would bind.
 *
 * Conventions
 *  - plain pointers and sizes only; all pointers are DEVICE pointers unless stated otherwise.
 *  - activations are NHWC "views": pointer to channel 0 of pixel 0 + `ld` = distance between
 *    consecutive pixels in ELEMENTS (>= channels; lets a tensor live inside a channel slice of a
 *    wider concat buffer, which is how train.py:114-119's tf.concat becomes zero-copy).
 *  - `dtype` selects the storage/operand type of activations, activation gradients and the
 *    weight operand: GCT2_F32 (reference default, train.py:34,38), GCT2_BF16, GCT2_F16
 *    (train.py:43-45 mixed_float16).  Accumulation, biases, weight gradients, Adam state: fp32.
 *  - kernels are enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream).
 *    No allocation, no host synchronisation, no ownership transfer; re-entrant across streams.
 *  - return value: GCT2_OK or an error code; nothing is launched when arguments are rejected.
 */
#ifndef GCT2_H
#define GCT2_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { GCT2_F32 = 0, GCT2_BF16 = 1, GCT2_F16 = 2 };
enum {
  GCT2_OK = 0,
  GCT2_EINVAL = 1,   /* bad shape / stride / alignment / dtype */
  GCT2_ELAUNCH = 2,  /* HIP launch error (hipGetLastError != success) */
  GCT2_ENODEV = 3    /* no gfx950 device visible */
};

/* library / device identification ------------------------------------------------------------ */
int gct2_abi_version(void);                 /* bumps when a signature below changes (v13: ReLU bit planes; v14: pruned tuning word, launch log, no deferred row sums; v15: launch-log read reports the size it needs, step plans; v16: bias queue; v17: plan event kinds - system-scope and timed records, gct2_plan_elapsed) */
/* how the library was built: 0 for the product build; bit 0 (GCT2_BUILD_STAMP) = diagnostic build with in-kernel phase stamps
 * (make EXTRA=-DGCT2_STAMP).  Product hosts (the Python binding, bench.py, the tests) refuse a library whose flags are not 0. */
enum { GCT2_BUILD_STAMP = 1 };
int gct2_build_flags(void);
const char* gct2_last_error(void);          /* host string describing the last non-OK return */
int gct2_device_check(void);                /* GCT2_OK iff the current device is gfx950 */
/* ABI v17, host plumbing: `workgroups` 256-thread work-groups that do nothing but hold their slots on `stream` for `microseconds`
 * (every wave spins on the constant 100-MHz s_memrealtime counter and leaves by itself; at most 20 ms).  Two users: the host mirror
 * finds out which of its streams the runtime put on ONE hardware queue (a short occupy on stream A, a marker behind a trivial one on
 * stream B: B finishing only after A means they share a queue and would block each other for the whole run - engine.py
 * `distinct_stream`), and scripts/bench_dp_overhead.py stands a collective's wire time in for a collective on a one-GPU box. */
int gct2_stream_occupy(void* stream, int workgroups, double microseconds);
/* ---- call context: caller-owned scratch + tuning, one per engine / host thread ------------------------------------------
 * The library keeps NO process-wide mutable state (ABI v11).  A gct2_ctx is a small host object created by the caller; it
 * carries (a) the caller's device scratch for split reductions / partial rows and (b) the tile-selection knobs.  A ctx is used by
 * ONE host thread at a time (the layer entry points write its one-shot ReLU plane and its launch log).  Calls that
 * share one ctx must be enqueued on ONE stream at a time per scratch area (forward / input-gradient / head calls use the
 * main workspace, *_wgrad calls the weight-gradient workspace when one is set); calls with DIFFERENT ctx objects (two
 * engines, two host threads, two streams) are fully independent.  ctx = NULL is allowed everywhere: no scratch (split-K and
 * partial-row reductions are off: same results, slower small layers, fp32 atomics for the bias sums), automatic tiles. */
typedef struct gct2_ctx gct2_ctx;
int gct2_ctx_create(gct2_ctx** ctx);
int gct2_ctx_destroy(gct2_ctx* ctx);
/* device scratch (16-byte aligned) for the split-K partial sums of layers whose output is too small to fill the chip (the
 * U-Net's bottleneck levels), the partial rows of the fused bias gradients and the Dense head.  ws = NULL removes it. */
int gct2_ctx_set_workspace(gct2_ctx* ctx, void* ws, size_t bytes);
/* optional second scratch used by the weight-gradient entry points only (their partial-tile slabs): with it *_wgrad calls
 * may run on a second stream concurrently with the forward/dgrad calls of the same ctx (the engine's reverse pass does). */
int gct2_ctx_set_wgrad_workspace(gct2_ctx* ctx, void* ws, size_t bytes);
/* Bias queue (ABI v16).  The input-gradient entry points fuse the bias gradient of the tensor they write into their epilogue and leave
 * partial rows that ONE small launch per call sums in a fixed order.  Eleven such launches sit between the input-gradient launches of a
 * reverse pass; with a queue buffer registered (16-byte aligned device memory; a few MB: rows x channels x 4 bytes per call) the calls of
 * this ctx leave their partial rows in the buffer and record the row set instead, and gct2_bias_queue_flush sums every recorded set -
 * same geometry, same order of additions, same "first writer overwrites, second adds" as the immediate launches, hence the same bits -
 * in two launches on `stream`.  All calls that fill one queue and its flush must be enqueued on ONE stream; a call whose rows do not fit
 * (or a 17th row set) first flushes what is queued and then reduces its own rows at once; buf = NULL returns to the immediate form and
 * drops row sets that were recorded and not flushed.  Program order per target is kept whoever writes it (v17): a call that writes a
 * bias gradient AT ONCE - direct kernels, an atomics epilogue taken for lack of row space, the db of a weight-gradient call - first
 * flushes the queue if a queued row set names the same target, and of the combinations that can meet in the queue only "queued
 * overwrite, then ONE add" is recorded together; a second add, a second overwrite or an overwrite behind a queued add flush first. */
int gct2_ctx_set_bias_queue(gct2_ctx* ctx, void* buf, size_t bytes);
int gct2_bias_queue_flush(gct2_ctx* ctx, void* stream);
/* tuning: forces a tile / order instead of the automatic per-layer choice (same results for every value within the stated
 * tolerances; the parity tests and scripts/bench_layer.py use it).  Unknown words are rejected.
 * bits 0-7  : forward / input-gradient tile: 0 = automatic, 2 = 128x128 (two LDS buffers), 5 = 256x128 (one buffer, 8 waves);
 * bit 8     : the forward / input-gradient GEMMs never split their reduction over work-groups (parity tests at reduced batch: the
 *             kernels of the batch-64 dispatch then run the way they do at batch 64);
 * bits 16-23: weight-gradient tile: 0 = automatic, 2 = 256x256 five-stage ring, 4 / 5 = the same in the r03 / r04 stage order
 *             (bit-identity tests of the later orders), 3 = 128x128, 6 = 128x128 with the general address code (bit-identity test of the
 *             image-aligned form), 7 = 128x128 with fp32 atomics instead of ordered slabs (arrival-order
 *             dependent: comparison tests only);
 * bits 24-25: halo-tile kernel (Conv2DTranspose forward / Conv2D input gradient): 0 = automatic, 1 = never, 2 = wherever allowed;
 * bits 26-27: tile -> XCD order of the forward / input-gradient GEMMs: 0 = automatic, 1 = bands of output pixels per XCD,
 *             2 = weight slices per XCD;
 * bits 28-30: forced pixel split of the 128x128 weight-gradient tile: 0 = automatic, v = 1..7: 2^(v-1) splits. */
int gct2_ctx_set_tuning(gct2_ctx* ctx, int v);
/* test hook: non-zero routes every convolution of this ctx through the direct (non-MFMA) kernels */
int gct2_ctx_force_direct(gct2_ctx* ctx, int on);
/* ReLU bit plane for the NEXT layer call of this ctx (r03, ABI v13; one-shot: the call consumes and clears it).
 * bits: device bytes [pixels][ld_bytes], bit k of byte c <-> channel 8c + k of the call's output view.
 *  - before gct2_conv4s2_fwd / gct2_convT4s2_fwd: the call ALSO writes bits = (y > 0) for its Cout channels (in the epilogue of the
 *    16-byte-store kernels; derived from the stored y by one extra launch on the other paths) - beside y, which is unchanged;
 *  - before gct2_conv4s2_dgrad / gct2_convT4s2_dgrad with act != NULL: the call MAY read its ReLU mask from the plane instead of
 *    act (1 byte instead of 16 per 8 channels: the mask read is a third of the epilogue traffic of these calls).  The plane must
 *    equal (act > 0) over the Cin channels of the call - which of the two a given kernel reads is unspecified.
 * Channels must be a multiple of 8, ld_bytes >= channels / 8.  bits = NULL clears a pending plane.  EVERY layer entry point that
 * takes a ctx removes a pending plane first thing, whether it succeeds or not; the ones that cannot use a plane (weight gradients,
 * the head calls, the stride-1 convolutions) return GCT2_EINVAL when one was pending. */
int gct2_ctx_set_relu_bits(gct2_ctx* ctx, void* bits, int ld_bytes);
/* diagnostic builds only (gct2_build_flags() & GCT2_BUILD_STAMP): device buffer that receives the s_memrealtime phase stamps of
 * one wave per work-group of the next stamped launch of this ctx (layout: scripts/stamp_*.py).  GCT2_EINVAL in a product build. */
int gct2_ctx_set_stamp_buffer(gct2_ctx* ctx, void* stamps, size_t bytes);

/* launch log (tests / diagnostics): while switched on, every layer entry point of this ctx appends the kernel it selected as a
 * text token ("tap:conv:256x128:mask:ksplit=1:bits;", "halo:convT:bias_act;", "wgrad:256q:rsplit=8:slabs;", "rgb:fwd;",
 * "direct:tap;", "relu_bits:derived;" ...).  gct2_ctx_log_launches(ctx, on) clears the log and switches it; read copies the
 * NUL-terminated text and clears the log.  *needed (may be NULL) receives the bytes the text needs including the NUL; when the
 * buffer is smaller (or NULL) the call returns GCT2_EINVAL and NEITHER copies NOR clears anything (ABI v15; v14 truncated
 * silently).  The log stops growing at 1 MiB and then ends with "log:truncated;".  A parity test at reduced batch uses it to
 * prove that the kernels it forced are the ones that ran. */
int gct2_ctx_log_launches(gct2_ctx* ctx, int on);
int gct2_ctx_read_launch_log(gct2_ctx* ctx, char* buf, size_t bytes, size_t* needed);

/* ---- DownShuffle = Conv2D(f, 4, 2, 'same', relu)   train.py:158-169 ------------------------- */
/* y[b,oh,ow,o] = act(bias[o] + sum_{kh,kw,i} x[b,2oh+kh-1,2ow+kw-1,i] * w[kh,kw,i,o])
 * x: [B,H,W,Cin] view, H and W even;  w: Keras kernel (4,4,Cin,Cout) of `dtype`;
 * bias: fp32[Cout] or NULL;  y: [B,H/2,W/2,Cout] view;  relu != 0 applies max(.,0). */
int gct2_conv4s2_fwd(gct2_ctx* ctx, int dtype, const void* x, int ldx, const void* w, const float* bias,
                     void* y, int ldy, int B, int H, int W, int Cin, int Cout, int relu,
                     void* stream);

/* input gradient of the above (autodiff of train.py:161-166 inside Keras fit, train.py:516):
 * dx[b,ih,iw,i] (+)= mask * sum_{kh,kw,o} dz[b,oh,ow,o] * w[kh,kw,i,o],  ih = 2oh+kh-1.
 * dz: [B,H/2,W/2,Cout] PRE-activation gradient;  dx: [B,H,W,Cin] view.
 * act (may be NULL): [B,H,W,Cin] view of the tensor whose ReLU produced x; mask = act > 0, so dx is
 * again a pre-activation gradient.  accumulate != 0: dx += result (skip branch of the concat,
 * train.py:114-119), else dx = result.
 * Fused bias gradients (optional): since dx is the gradient w.r.t. the PRE-activation of the layer(s) that produced
 * `act`, its column sums are those layers' bias gradients.  db (+)= sums of THIS call's masked result over channels
 * [0, db_split), db2 (+)= over channels [db_split, Cin) (a concat buffer spans two layers); NULL = not wanted.
 * db_accumulate: bit 0 set -> db is added to, clear -> db is overwritten; bit 1 the same for db2.  (The skip slice of a
 * concat buffer receives two gradient contributions, train.py:114-119: the first launch overwrites, the second adds, so the
 * caller never has to zero a bias gradient.) */
int gct2_conv4s2_dgrad(gct2_ctx* ctx, int dtype, const void* dz, int lddz, const void* w, const void* act,
                       int ldact, void* dx, int lddx, int B, int H, int W, int Cin, int Cout,
                       int accumulate, float* db, int db_split, float* db2, int db_accumulate, void* stream);

/* optional optimizer step fused behind a weight-gradient call (single-replica training without loss scaling): Keras Adam
 * (as gct2_adam_keras_multi) over a contiguous parameter range of the caller's arenas that STARTS with the layer's kernel,
 * enqueued on the same stream right after the gradient.  p, m, v: fp32, start of the kernel's slice; n: elements of the
 * whole range (>= the kernel's 16*Cin*Cout; the engine's ranges are the kernel plus alignment padding - its biases live in a
 * separate fp32 zone with a launch of its own); dw (the call's gradient pointer) must be the matching start of the gradient
 * arena.  The kernel gradient is consumed straight from the launch's partial sums where it has them (it is then never
 * written to dw) or from dw; whatever lies behind the kernel in the range is read from the gradient arena.  Nothing is
 * zeroed.  Needs accumulate = 0. */
typedef struct gct2_adam_args {
  float* p; float* m; float* v;
  void* shadow; int shadow_dtype;      /* compute-dtype copy of p over the same range, or NULL */
  size_t n;
  float alpha, beta1, beta2, eps, grad_mul;
  /* defer != 0 (r04): the weight-gradient call launches NO optimizer step; it records where it left the kernel gradient - slab_base
   * / nslab / slab_stride: `nslab` partial tensors in the ctx's weight-gradient scratch (nslab = 0: the gradient is in dw) - and the
   * caller runs the step later with gct2_adam_apply(this struct, dw, 16*Cin*Cout, stream): the same launch, the same bits.  Until
   * then the scratch of that ctx must not be handed to another weight-gradient call (the engine gives such layers a ctx of their
   * own) and dw (plus whatever the range holds behind the kernel in the gradient arena) must stay untouched.  The engine uses it to run the optimizer step of the layers the
   * forward pass needs LAST inside the bottleneck window of the next forward pass. */
  int defer;
  const float* slab_base; int nslab; size_t slab_stride;     /* out (defer != 0) */
} gct2_adam_args;

/* weight + bias gradient: dw[kh,kw,i,o] += sum_{b,oh,ow} x[b,2oh+kh-1,2ow+kw-1,i]*dz[b,oh,ow,o],
 * db[o] += sum dz[..,o].  dw: fp32 (4,4,Cin,Cout), db: fp32[Cout] or NULL.  accumulate != 0: dw is a running (or
 * zeroed) buffer and is added to; accumulate == 0: dw is overwritten (saves reading it).  db follows the same flag. */
int gct2_conv4s2_wgrad(gct2_ctx* ctx, int dtype, const void* x, int ldx, const void* dz, int lddz, float* dw,
                       float* db, int B, int H, int W, int Cin, int Cout, int accumulate,
                       gct2_adam_args* adam /* or NULL */, void* stream);
/* the optimizer step a weight-gradient call with adam->defer != 0 left to the caller (see gct2_adam_args) */
int gct2_adam_apply(const gct2_adam_args* adam, float* dw, size_t nw, void* stream);

/* ---- UpShuffle = Conv2DTranspose(f, 4, 2, 'same', relu)   train.py:145-156 ------------------ */
/* y[b,2ih+kh-1,2iw+kw-1,o] += x[b,ih,iw,i] * w[kh,kw,o,i]; then bias, relu.
 * x: [B,H,W,Cin] view;  w: Keras kernel (4,4,Cout,Cin);  y: [B,2H,2W,Cout] view. */
int gct2_convT4s2_fwd(gct2_ctx* ctx, int dtype, const void* x, int ldx, const void* w, const float* bias,
                      void* y, int ldy, int B, int H, int W, int Cin, int Cout, int relu,
                      void* stream);

/* dx[b,ih,iw,i] (+)= mask * sum_{kh,kw,o} dz[b,2ih+kh-1,2iw+kw-1,o] * w[kh,kw,o,i]
 * dz: [B,2H,2W,Cout];  dx/act: [B,H,W,Cin] views;  mask/accumulate as for conv4s2_dgrad. */
int gct2_convT4s2_dgrad(gct2_ctx* ctx, int dtype, const void* dz, int lddz, const void* w, const void* act,
                        int ldact, void* dx, int lddx, int B, int H, int W, int Cin, int Cout,
                        int accumulate, float* db, int db_split, float* db2, int db_accumulate, void* stream);

/* dw[kh,kw,o,i] (+)= sum_{b,ih,iw} x[b,ih,iw,i] * dz[b,2ih+kh-1,2iw+kw-1,o]; db[o] += sum dz; accumulate as for
 * conv4s2_wgrad. */
int gct2_convT4s2_wgrad(gct2_ctx* ctx, int dtype, const void* x, int ldx, const void* dz, int lddz, float* dw,
                        float* db, int B, int H, int W, int Cin, int Cout, int accumulate,
                        gct2_adam_args* adam /* or NULL */, void* stream);

/* ---- the reference's off-by-default model variants ------------------------------------------------------------------------
 * Block = block_depth x Conv2D(filters, 3, 1, 'same', relu) (train.py:20, 123-143) and the bias-free projection
 * Dense(input_channels) of Residual's residual=True mode (train.py:26, 104-112; a Dense on a rank-4 tensor is a 1 x 1 convolution,
 * its (Cin, Cout) kernel the same memory as (1, 1, Cin, Cout)).  Stride-1 'same' convolution, KS odd (symmetric padding):
 *   y[b,h,w,o] = act(bias[o] + sum_{kh,kw,i} x[b,h+kh-p,w+kw-p,i] w[kh,kw,i,o]),  p = (KS-1)/2;  w: (KS,KS,Cin,Cout) of `dtype`.
 * dgrad: dx[b,h,w,i] (+)= mask * sum dz[b,h-kh+p,w-kw+p,o] w[kh,kw,i,o] (mask / accumulate as for conv4s2_dgrad);
 * wgrad: dw (+)= sum x dz (fp32), db (+)= column sums of dz.  16-bit dtypes with whole 8-channel chunks (Cin, Cout and the lds
 * multiples of 8, 16-byte aligned views, KS <= 5) run on the matrix cores as a third tap-GEMM form (KS x KS taps on one grid;
 * split-K / slabs through the ctx scratch like the 4x4 layers); fp32 and the remaining shapes (the 3-channel input of a Block in
 * front of level 0) on direct kernels, one thread per output. */
int gct2_conv2d_s1_fwd(gct2_ctx* ctx, int dtype, const void* x, int ldx, const void* w, const float* bias, void* y, int ldy,
                       int B, int H, int W, int Cin, int Cout, int KS, int relu, void* stream);
int gct2_conv2d_s1_dgrad(gct2_ctx* ctx, int dtype, const void* dz, int lddz, const void* w, const void* act, int ldact,
                         void* dx, int lddx, int B, int H, int W, int Cin, int Cout, int KS, int accumulate, void* stream);
int gct2_conv2d_s1_wgrad(gct2_ctx* ctx, int dtype, const void* x, int ldx, const void* dz, int lddz, float* dw, float* db,
                         int B, int H, int W, int Cin, int Cout, int KS, int accumulate, void* stream);
/* d[pix,c] = act[pix,c] > 0 ? d[pix,c] : 0 - the autodiff of a fused ReLU where no producing kernel applies the mask itself */
int gct2_relu_mask(int dtype, const void* act, int ldact, void* d, int ldd, size_t npix, int C, void* stream);
/* dst[pix,c] += src[pix,c] (train.py:112 `input + ...` and the gradient joins of the variants) */
int gct2_add(int dtype, void* dst, int lddst, const void* src, int ldsrc, size_t npix, int C, void* stream);
/* out = a[b] * x + c[b] * eps, fp32, per-image coefficients (eps = NULL: out = a[b] * x): the targets and prediction weights of
 * train.py:238-252 (ODE target, epsilon / scaled-epsilon prediction, prediction_weighting) */
int gct2_mix_per_image(const float* x, const float* eps, const float* a, const float* c, float* out, int B, size_t per_image,
                       void* stream);

/* ---- Dense(3) head on a rank-4 input   train.py:198-202 ------------------------------------- */
/* y[m,o] = b[o] + sum_i x[m,i] * w[i,o];  x: [M,Cin] view of `dtype`; w fp32 (Cin,Cout), Cout <= 4;
 * y: fp32 [M,Cout] contiguous (the loss is taken in fp32, train.py:262-263).  GCT2_F16: the values are rounded to fp16
 * first (mixed_float16 Dense output, cast to fp32 at train.py:263). */
int gct2_dense_fwd(int dtype, const void* x, int ldx, const float* w, const float* b, float* y,
                   int M, int Cin, int Cout, void* stream);

/* dx[m,i] = mask * sum_o dy[m,o] w[i,o] for i < Cmask (channels >= Cmask get no gradient written);
 * dw[i,o] (+)= sum_m x[m,i] dy[m,o];  db[o] (+)= sum_m dy[m,o]  (accumulate != 0 adds, else overwrites).   dy fp32 [M,Cout].
 * mask = x[m,i] > 0 (x is the ReLU output feeding the head; train.py:188,198).  GCT2_F16: dy is rounded to fp16 on read
 * (the gradient entering a mixed_float16 Dense output is fp16). */
int gct2_dense_bwd(int dtype, const void* x, int ldx, const float* w, const float* dy, void* dx,
                   int lddx, float* dw, float* db, int M, int Cin, int Cout, int Cmask,
                   int accumulate, void* stream);

/* fused train-step head (16-bit dtypes): dense_fwd + mse_fwd_bwd + dense_bwd in ONE pass over x (= R_0):
 *   pred = x w + b;  loss = mean((pred - target)^2);  dpred = loss_scale * 2 (pred - target) / (M Cout);
 *   dx[m, i < Cmask] = (x > 0) * dpred w^T;  dw (+)= x^T dpred;  db (+)= sum dpred.
 * GCT2_F16 follows Keras' mixed_float16 rounding points (train.py:43-45): the Dense output and the gradient entering it are
 * rounded to fp16 (pred is stored as fp32 of the fp16 value); GCT2_BF16 keeps both in fp32.
 * target: fp32 [M,Cout]; pred: fp32 [M,Cout] or NULL; loss: 1 float; partials: >= 1024 floats scratch.
 * x2 (may be NULL): the input channels [Cmask, Cin) come from this second view (ldx2 elements per pixel, at most 4
 * channels, 8-byte aligned rows) instead of x - the concat [UpShuffle_0 output, image] is then never assembled at all
 * (the image lives in its packed copy only).  Needs Cmask = 64, Cout <= 3 and a registered workspace.
 * Replaces train.py:198-202 + 262-272 and their autodiff inside Keras fit (train.py:516). */
int gct2_dense_head_train(gct2_ctx* ctx, int dtype, const void* x, int ldx, const float* w, const float* b,
                          const float* target, float* pred, void* dx, int lddx, float* dw, float* db,
                          float* loss, float* partials, int M, int Cin, int Cout, int Cmask,
                          const float* loss_scale_ptr, float* db_dx /* column sums of dx, or NULL */,
                          const void* x2, int ldx2, int accumulate /* dw, db, db_dx: 0 = overwrite, else add */,
                          void* stream);

/* UpShuffle_0's forward WITH the train-step head in its epilogue (16-bit dtypes): y = relu(convT(x) + bias) is consumed where
 * it is produced - Dense(3) + fp32 MSE + both of their gradients, exactly as gct2_dense_head_train computes them on the stored
 * activations (y is rounded to `dtype` first, like the tensor the separate call would have read) - and never written: dy
 * ([B,2H,2W,Cout] view) receives the gradient w.r.t. the layer's pre-activation, db its column sums (this layer's bias
 * gradient), head_dw / head_db the Dense gradients, loss the scalar.  Saves one write and one read of the largest activation
 * of the network plus the head launch (train.py:188, 198-202, 262-272 and their autodiff).
 * Needs Cout = 64, H and W multiples of 16, head_Cin - Cout <= 3 image channels in x2 (packed, ldx2 multiple of 4), head_Cout <= 3
 * and a ctx workspace of B*(H/16)*(W/16)*288 floats; GCT2_EINVAL otherwise (use convT4s2_fwd + dense_head_train). */
int gct2_convT4s2_fwd_head_train(gct2_ctx* ctx, int dtype, const void* x, int ldx, const void* w, const float* bias,
                                 const float* head_w, const float* head_b, const float* target, float* pred, void* dy,
                                 int lddy, float* head_dw, float* head_db, float* loss, int B, int H, int W, int Cin,
                                 int Cout, int head_Cin, int head_Cout, const float* loss_scale_ptr, float* db,
                                 const void* x2, int ldx2, int accumulate, void* stream);

/* ---- Trainer.call pieces   train.py:223-272 -------------------------------------------------- */
/* t_int[b] ~ U{1..steps} (train.py:224-226) and eps ~ N(0,1) (train.py:227) from a counter-based
 * Philox4x32-10 stream keyed by (seed, stream_id); `offset` = elements already drawn. */
int gct2_rng_uniform_int(uint64_t seed, uint64_t stream_id, uint64_t offset, int32_t* out, size_t n,
                         int lo, int hi_inclusive, void* stream);
int gct2_rng_normal(uint64_t seed, uint64_t stream_id, uint64_t offset, float* out, size_t n,
                    void* stream);

/* noised = x*sqrt(a_b) + eps*sqrt(1-a_b), a_b = 0.25*(1 - t_b/(steps+1))^2  (train.py:85-93,229-234)
 * x, eps: fp32 [B, HW, C] contiguous; t_int: int32[B]; out: view [B*HW, C] of `dtype` with ldout.
 * GCT2_F32 / GCT2_BF16: fp32 arithmetic, one rounding at the store.  GCT2_F16: every operation is rounded to fp16 as under
 * Keras' mixed_float16 policy, where x, eps and t are fp16 tensors (train.py:38, 227-234, 292).
 * out2 (may be NULL): a second view [B*HW, C] with ldout2 that receives the same values - the engine keeps a packed
 * copy of the image (ld 4) for DownShuffle_0 beside the slice of the concat buffer that Dense(3) reads. */
int gct2_noise_image(int dtype, const float* x, const int32_t* t_int, const float* eps, void* out,
                     int ldout, void* out2, int ldout2, int B, int HW, int C, int steps, void* stream);

/* the same with eps drawn inside the kernel from the positions [offset, offset + B*HW*C) of the gct2_rng_normal stream
 * (seed, stream_id): bit-identical to gct2_rng_normal followed by gct2_noise_image, without the eps round trip through HBM.
 * eps_out (may be NULL) receives the draws. */
int gct2_noise_image_rng(int dtype, const float* x, const int32_t* t_int, uint64_t seed, uint64_t stream_id,
                         uint64_t offset, float* eps_out, void* out, int ldout, void* out2, int ldout2, int B, int HW,
                         int C, int steps, void* stream);

/* ---- log_sample, the reference's sampler (train.py:323-496, every objective switch of train.py:29-32): pointwise steps, fp32 state ---- */
/* fake = sqrt(alpha) x_theta + sqrt(1-alpha) eps_theta  (train.py:372-375, 441-444), alpha = alpha_dash(t) from the caller.
 * x_theta, eps_theta, fake: fp32 [npix*C]; the network input is also stored in `dtype` into the view out [npix, C] (ldout) and,
 * if non-NULL, out2 (ldout2) - the packed image and the image slice of the concat buffer. */
int gct2_diffusion_mix(int dtype, const float* x_theta, const float* eps_theta, float alpha, float* fake,
                       void* out, int ldout, void* out2, int ldout2, size_t npix, int C, void* stream);
/* one sampler update from the network's prediction, by objective (train.py:338-355, 382-413, 452-479); fp32 [n] each,
 * a = alpha = alpha_dash(t), a1 = alpha_prev = alpha_dash(t - 1) (ODE mode only, ignored otherwise):
 *   GCT2_SAMPLE_X          (predict_x)            x_theta = pred;  eps_theta = (fake - sqrt(a) pred) / sqrt(1-a)
 *   GCT2_SAMPLE_EPS        (epsilon)              eps_theta = pred;  x_theta = (fake - pred sqrt(1-a)) / sqrt(a)
 *   GCT2_SAMPLE_SCALED_EPS (predict_scaled_epsilon) eps_theta = pred / sqrt(1-a);  x_theta = (fake - pred) / sqrt(a)
 *   GCT2_SAMPLE_ODE        (ordinary_differential_equation)
 *                          x_theta = (pred sqrt(1-a) - fake sqrt(1-a1)) / (sqrt(a1) sqrt(1-a) - sqrt(a) sqrt(1-a1));
 *                          eps_theta is NOT touched (the reference never updates it in this branch; may be NULL).
 */
enum { GCT2_SAMPLE_X = 0, GCT2_SAMPLE_EPS = 1, GCT2_SAMPLE_SCALED_EPS = 2, GCT2_SAMPLE_ODE = 3 };
/* alpha / alpha_prev are DOUBLES: the reference forms every coefficient (and the ODE denominator, a difference of nearly equal
 * products at steps = 200) from Python floats and only then multiplies fp32 tensors by it; the library does the same on the host. */
int gct2_diffusion_update(int mode, const float* pred, const float* fake, double alpha, double alpha_prev, float* x_theta,
                          float* eps_theta, size_t n, void* stream);
/* the four inputs of the reverse pass built from one inverted noise image eps [H,W,C] (train.py:416-431): out [4,H,W,C] =
 * eps | nearest-upsample x4 of avg_pool2d(eps, 4, 4) | tf.roll by 1 along H and W | per-pixel nearest of the K entries of
 * dictionary [H,W,K,C] (squared distance, first minimum).  H, W multiples of 4. */
int gct2_noise_edits(const float* eps, const float* dictionary, int K, float* out, int H, int W, int C,
                     void* stream);

/* ---- input contract, decode_file without the decoder (train.py:285-293) ------------------------------------------------ */
/* dst[b,y,x,c] = src_b[oy+y, ox + (flip ? size-1-x : x), c] / 128 - 1: random_crop + random_flip_left_right + the cast of
 * train.py:288-292 for a whole batch in one launch.  src: device buffer of decoded RGB images (u8, HWC) back to back;
 * offsets[b]: byte offset of image b; dims[b] = {H0, W0, oy, ox, flip} (int32, the caller draws the crop origin and the flip and
 * guarantees oy + size <= H0, ox + size <= W0); dst: fp32 [B, size, size, 3]. */
int gct2_image_prepare(const uint8_t* src, const int64_t* offsets, const int32_t* dims, float* dst, int B, int size,
                       void* stream);

/* loss = mean((target - pred)^2) in fp32 (train.py:272); dpred = loss_scale * 2 (pred-target)/n.
 * `loss` (1 float) is overwritten; `partials` is caller scratch of >= 1024 floats.
 * loss_scale_ptr: device pointer to the current loss scale (fp16 mode) or NULL for 1. */
int gct2_mse_fwd_bwd(const float* pred, const float* target, float* dpred, float* loss,
                     float* partials, size_t n, const float* loss_scale_ptr, void* stream);

/* ---- mixed precision (train.py:34,43-45,82-83) ------------------------------------------------ */
/* device-resident state of Keras' LossScaleOptimizer [TF] plus the optimizer step counter it gates: a skipped step (inf/nan
 * gradients) does not run the inner apply_gradients, so optimizer.iterations - and with it the WarmUp step and Adam's bias
 * correction - only advance on APPLIED steps.  32 bytes, 16-byte aligned. */
typedef struct {
  float scale; float inv_scale; int32_t good_steps; int32_t found_inf;
  int32_t applied_steps;   /* optimizer.iterations */
  float alpha;             /* lr(applied_steps) * sqrt(1-b2^t)/(1-b1^t), t = applied_steps+1: written by loss_scale_begin */
  int32_t reserved[2];
} gct2_loss_scale_state;
int gct2_loss_scale_init(gct2_loss_scale_state* state, float initial_scale, void* stream);
/* start of a step: found_inf = 0 and alpha for THIS step from the WarmUp schedule (train.py:57-65: float32
 * base * (step+1) / (warmup_steps+1) while step < warmup_steps, else base) and Adam's bias correction. */
int gct2_loss_scale_begin(gct2_loss_scale_state* state, float base_lr, int warmup_steps, float beta1, float beta2,
                          void* stream);
/* found_inf |= any(!isfinite(g)) */
int gct2_scale_check_finite(const float* g, size_t n, gct2_loss_scale_state* state, void* stream);
/* dynamic update [TF]: finite -> applied_steps++, good_steps++, scale x2 every growth_interval; non-finite -> scale /2, reset. */
int gct2_loss_scale_update(gct2_loss_scale_state* state, int growth_interval, void* stream);

/* ---- optimizer   train.py:50-65, 75 (Keras Adam + WarmUp) ------------------------------------ */
/* Keras ResourceApplyAdam over a flat fp32 arena of n parameters:
 *   g' = g * grad_mul * (ls ? ls->inv_scale : 1);  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;
 *   p -= alpha * m / (sqrt(v) + eps),  alpha = lr_k * sqrt(1-b2^t)/(1-b1^t) computed by the host (lr_k from the WarmUp
 *   schedule) - or, with ls != NULL, taken from ls->alpha (the device-side step counter decides) and the whole update is
 *   skipped when ls->found_inf != 0 (LossScaleOptimizer, train.py:82-83).
 *   grad_mul: host scalar (1/world_size turns the all-reduced SUM into the data-parallel mean).
 *   shadow (may be NULL): low-precision copy of p in `shadow_dtype` written in the same pass.
 *   zero_grad != 0: g is zeroed after use (for callers that accumulate gradients; the engine never needs it). */
int gct2_adam_keras_multi(float* p, float* m, float* v, float* g, void* shadow, int shadow_dtype,
                          size_t n, float alpha, float beta1, float beta2, float eps, float grad_mul,
                          const gct2_loss_scale_state* ls, int zero_grad, void* stream);

/* fp32 -> dtype cast of a flat array (initial weight shadows). */
int gct2_cast_from_f32(int dtype, const float* src, void* dst, size_t n, void* stream);

/* ---- step plans (ABI v15): Keras `fit` (train.py:516) runs one compiled train function per step; this is its counterpart ----
 * A plan is a host-side list of records - calls of the entry points above, event records, stream waits - built once and replayed
 * by ONE call per step (or per segment of a step, where the caller interleaves work of its own: the data-parallel exchange
 * hooks).  A replayed record is exactly the call it records: same entry point, same arguments, same stream - same kernels, same
 * bits.  Arguments are passed as one 64-bit slot each: pointers and 64-bit integers as they are, `int` sign-extended, `float` as
 * its bit pattern in the low 32 bits, `double` as its bit pattern.  Values that change from step to step (the input batch, RNG
 * offsets, the optimizer's alpha) are re-set with gct2_plan_set_arg before a run; structs passed by pointer (gct2_adam_args) are
 * read at run time, so the caller may update them in place.  A plan belongs to ONE host thread at a time, like the ctx objects its
 * records name; it creates one HIP event per event record (no timing, device-scope release) and destroys them with the plan.
 * Plannable entry points: every function of this header that takes a stream, plus gct2_ctx_set_relu_bits. */
typedef struct gct2_plan gct2_plan;
int gct2_plan_create(gct2_plan** plan);
int gct2_plan_destroy(gct2_plan* plan);
/* appends a call of entry point `name` ("gct2_conv4s2_fwd", ...); nargs must equal the entry point's parameter count; *index (may
 * be NULL) receives the record's position */
int gct2_plan_add_call(gct2_plan* plan, const char* name, const uint64_t* args, int nargs, int* index);
/* appends "record a new event on `stream`" / "make `stream` wait for event `event`" (an earlier add_record of this plan) */
int gct2_plan_add_record(gct2_plan* plan, void* stream, int* event);
int gct2_plan_add_wait(gct2_plan* plan, void* stream, int event);
/* ABI v17: the same record with a chosen event kind.  GCT2_EVENT_DEVICE = gct2_plan_add_record (no timing, device-scope release:
 * ordering between streams of one device); GCT2_EVENT_SYSTEM: no timing, system-scope release - for a record whose waiter hands
 * the data to ANOTHER device (the stream a collective of the data-parallel exchange is issued on); GCT2_EVENT_TIMED: a timing
 * event.  gct2_plan_elapsed: milliseconds between two TIMED records of the last run, both completed (the caller synchronised) -
 * how bench.py brackets every layer call of a replayed step without a host round trip between the event and the launch. */
enum { GCT2_EVENT_DEVICE = 0, GCT2_EVENT_SYSTEM = 1, GCT2_EVENT_TIMED = 2 };
int gct2_plan_add_record_kind(gct2_plan* plan, void* stream, int kind, int* event);
int gct2_plan_elapsed(gct2_plan* plan, int start_event, int end_event, float* ms);
int gct2_plan_size(const gct2_plan* plan, int* records);
int gct2_plan_set_arg(gct2_plan* plan, int index, int arg, uint64_t value);
/* executes records [first, first + count) in order; stops at the first record that fails, returns its status (message:
 * gct2_last_error) and, if `failed` is non-NULL, stores its index there (-1 when all succeeded) */
int gct2_plan_run(gct2_plan* plan, int first, int count, int* failed);

#ifdef __cplusplus
}
#endif
#endif /* GCT2_H */
