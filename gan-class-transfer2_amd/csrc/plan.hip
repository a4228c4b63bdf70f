// Step plans (include/gct2.h, ABI v15; v16 adds the bias-queue flush to the entry table): a pre-built list of launch records - entry-point calls of this library, event records and
// stream waits - that ONE C call walks.  The host mirror (engine.py) records the calls of a train step once and replays the list
// every step: the same entry points with the same arguments on the same streams, hence the same kernels and the same bits, without
// ~90 interpreter round trips per step (0.68 ms of Python enqueue per 2.5-ms step, 1.2-1.5 ms with the
// data-parallel hooks).  Host code only; nothing here launches a kernel of its own.
#include "gct2_common.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>

namespace {

constexpr int MAX_ARGS = 28;

// one 64-bit slot per argument: pointers and 64-bit integers as they are, ints sign-extended, floats as their bit pattern in the
// low half, doubles as their bit pattern
template <class T> T slot_as(uint64_t v) {
  if constexpr (std::is_pointer<T>::value) return reinterpret_cast<T>(static_cast<uintptr_t>(v));
  else if constexpr (std::is_same<T, float>::value) { const uint32_t u = (uint32_t)v; float f; memcpy(&f, &u, 4); return f; }
  else if constexpr (std::is_same<T, double>::value) { double d; memcpy(&d, &v, 8); return d; }
  else return static_cast<T>(v);
}
template <class... A, size_t... I> int invoke(int (*fn)(A...), const uint64_t* a, std::index_sequence<I...>) { return fn(slot_as<A>(a[I])...); }
template <class... A> int thunk(int (*fn)(A...), const uint64_t* a) { return invoke(fn, a, std::index_sequence_for<A...>{}); }
template <class... A> constexpr int arity(int (*)(A...)) { return (int)sizeof...(A); }

struct Entry { const char* name; int (*call)(const uint64_t*); int nargs; };
#define GCT2_ENTRY(f) {#f, [](const uint64_t* a) -> int { return thunk(&f, a); }, arity(&f)}
// every entry point that enqueues work on a stream (the context setters that a step uses are included: the one-shot ReLU plane)
const Entry ENTRIES[] = {
    GCT2_ENTRY(gct2_ctx_set_relu_bits), GCT2_ENTRY(gct2_bias_queue_flush),
    GCT2_ENTRY(gct2_conv4s2_fwd), GCT2_ENTRY(gct2_conv4s2_dgrad), GCT2_ENTRY(gct2_conv4s2_wgrad),
    GCT2_ENTRY(gct2_convT4s2_fwd), GCT2_ENTRY(gct2_convT4s2_fwd_head_train), GCT2_ENTRY(gct2_convT4s2_dgrad), GCT2_ENTRY(gct2_convT4s2_wgrad),
    GCT2_ENTRY(gct2_adam_apply), GCT2_ENTRY(gct2_adam_keras_multi),
    GCT2_ENTRY(gct2_conv2d_s1_fwd), GCT2_ENTRY(gct2_conv2d_s1_dgrad), GCT2_ENTRY(gct2_conv2d_s1_wgrad),
    GCT2_ENTRY(gct2_relu_mask), GCT2_ENTRY(gct2_add), GCT2_ENTRY(gct2_mix_per_image),
    GCT2_ENTRY(gct2_dense_fwd), GCT2_ENTRY(gct2_dense_bwd), GCT2_ENTRY(gct2_dense_head_train),
    GCT2_ENTRY(gct2_rng_uniform_int), GCT2_ENTRY(gct2_rng_normal), GCT2_ENTRY(gct2_noise_image), GCT2_ENTRY(gct2_noise_image_rng),
    GCT2_ENTRY(gct2_mse_fwd_bwd), GCT2_ENTRY(gct2_cast_from_f32),
    GCT2_ENTRY(gct2_loss_scale_begin), GCT2_ENTRY(gct2_scale_check_finite), GCT2_ENTRY(gct2_loss_scale_update),
    GCT2_ENTRY(gct2_diffusion_mix), GCT2_ENTRY(gct2_diffusion_update), GCT2_ENTRY(gct2_noise_edits), GCT2_ENTRY(gct2_image_prepare),
};
#undef GCT2_ENTRY

enum { OP_CALL = 0, OP_RECORD = 1, OP_WAIT = 2 };
struct Op {
  int kind, entry, nargs, event;
  hipStream_t stream;
  uint64_t a[MAX_ARGS];
};

}  // namespace

struct gct2_plan {
  std::vector<Op> ops;
  std::vector<hipEvent_t> events;
  std::vector<char> timed;       // per event: created with timing (GCT2_EVENT_TIMED)
};

extern "C" {

int gct2_plan_create(gct2_plan** plan) {
  if (!plan) return gct2_fail(GCT2_EINVAL, "plan_create: null output pointer");
  *plan = new (std::nothrow) gct2_plan();
  return *plan ? GCT2_OK : gct2_fail(GCT2_EINVAL, "plan_create: out of host memory");
}

int gct2_plan_destroy(gct2_plan* plan) {
  if (!plan) return GCT2_OK;
  for (hipEvent_t e : plan->events) (void)hipEventDestroy(e);
  delete plan;
  return GCT2_OK;
}

int gct2_plan_add_call(gct2_plan* plan, const char* name, const uint64_t* args, int nargs, int* index) {
  if (!plan || !name || (nargs > 0 && !args)) return gct2_fail(GCT2_EINVAL, "plan_add_call: null pointer");
  for (int e = 0; e < (int)(sizeof(ENTRIES) / sizeof(ENTRIES[0])); e++) {
    if (strcmp(ENTRIES[e].name, name)) continue;
    if (nargs != ENTRIES[e].nargs) return gct2_fail(GCT2_EINVAL, "plan_add_call: %s takes %d arguments, got %d", name, ENTRIES[e].nargs, nargs);
    static_assert(MAX_ARGS >= 28, "gct2_convT4s2_fwd_head_train has 28 arguments");
    if (nargs > MAX_ARGS) return gct2_fail(GCT2_EINVAL, "plan_add_call: too many arguments");
    Op op{};
    op.kind = OP_CALL; op.entry = e; op.nargs = nargs; op.event = -1; op.stream = nullptr;
    for (int i = 0; i < nargs; i++) op.a[i] = args[i];
    plan->ops.push_back(op);
    if (index) *index = (int)plan->ops.size() - 1;
    return GCT2_OK;
  }
  return gct2_fail(GCT2_EINVAL, "plan_add_call: %s is not an entry point a plan can hold", name);
}

int gct2_plan_add_record_kind(gct2_plan* plan, void* stream, int kind, int* event) {
  if (!plan || !event) return gct2_fail(GCT2_EINVAL, "plan_add_record: null pointer");
  if (kind < GCT2_EVENT_DEVICE || kind > GCT2_EVENT_TIMED) return gct2_fail(GCT2_EINVAL, "plan_add_record: unknown event kind %d", kind);
  hipEvent_t ev;
  // GCT2_EVENT_DEVICE: ordering inside one device only - no timing, device-scope release (the default is a system-scope fence per
  // record); GCT2_EVENT_SYSTEM: no timing, the runtime's default (system-scope) release - for a record whose waiter hands the data to
  // another device (the stream an RCCL collective is issued on); GCT2_EVENT_TIMED: a timing event (gct2_plan_elapsed)
  unsigned flags = kind == GCT2_EVENT_DEVICE ? (hipEventDisableTiming | hipEventReleaseToDevice) : kind == GCT2_EVENT_SYSTEM ? hipEventDisableTiming : hipEventDefault;
  if (kind == GCT2_EVENT_DEVICE)
    if (const char* e = getenv("GCT2_PLAN_EVENT_FLAGS")) flags = (unsigned)strtoul(e, nullptr, 0);      // diagnostics: A/B of the release scope
  if (hipEventCreateWithFlags(&ev, flags) != hipSuccess) {
    (void)hipGetLastError();
    return gct2_fail(GCT2_ELAUNCH, "plan_add_record: hipEventCreateWithFlags failed");
  }
  plan->events.push_back(ev);
  plan->timed.push_back(kind == GCT2_EVENT_TIMED);
  Op op{};
  op.kind = OP_RECORD; op.entry = -1; op.nargs = 0; op.event = (int)plan->events.size() - 1; op.stream = reinterpret_cast<hipStream_t>(stream);
  plan->ops.push_back(op);
  *event = op.event;
  return GCT2_OK;
}

int gct2_plan_add_record(gct2_plan* plan, void* stream, int* event) { return gct2_plan_add_record_kind(plan, stream, GCT2_EVENT_DEVICE, event); }

int gct2_plan_elapsed(gct2_plan* plan, int start, int end, float* ms) {
  if (!plan || !ms) return gct2_fail(GCT2_EINVAL, "plan_elapsed: null pointer");
  const int n = (int)plan->events.size();
  if (start < 0 || start >= n || end < 0 || end >= n || !plan->timed[start] || !plan->timed[end])
    return gct2_fail(GCT2_EINVAL, "plan_elapsed: events %d / %d are not timed records of this plan", start, end);
  // both events must have completed (the caller synchronised the device or the streams behind the run)
  if (hipEventElapsedTime(ms, plan->events[start], plan->events[end]) != hipSuccess) {
    (void)hipGetLastError();
    return gct2_fail(GCT2_ELAUNCH, "plan_elapsed: hipEventElapsedTime failed (events not recorded yet, or still pending)");
  }
  return GCT2_OK;
}

int gct2_plan_add_wait(gct2_plan* plan, void* stream, int event) {
  if (!plan || event < 0 || event >= (int)plan->events.size()) return gct2_fail(GCT2_EINVAL, "plan_add_wait: null plan or unknown event %d", event);
  Op op{};
  op.kind = OP_WAIT; op.entry = -1; op.nargs = 0; op.event = event; op.stream = reinterpret_cast<hipStream_t>(stream);
  plan->ops.push_back(op);
  return GCT2_OK;
}

int gct2_plan_size(const gct2_plan* plan, int* ops) {
  if (!plan || !ops) return gct2_fail(GCT2_EINVAL, "plan_size: null pointer");
  *ops = (int)plan->ops.size();
  return GCT2_OK;
}

int gct2_plan_set_arg(gct2_plan* plan, int index, int arg, uint64_t value) {
  if (!plan || index < 0 || index >= (int)plan->ops.size()) return gct2_fail(GCT2_EINVAL, "plan_set_arg: no record %d", index);
  Op& op = plan->ops[index];
  if (op.kind != OP_CALL || arg < 0 || arg >= op.nargs) return gct2_fail(GCT2_EINVAL, "plan_set_arg: record %d has no argument %d", index, arg);
  op.a[arg] = value;
  return GCT2_OK;
}

int gct2_plan_run(gct2_plan* plan, int first, int count, int* failed) {
  if (failed) *failed = -1;
  if (!plan || first < 0 || count < 0 || (size_t)first + (size_t)count > plan->ops.size())
    return gct2_fail(GCT2_EINVAL, "plan_run: records [%d, %d) outside the plan", first, first + count);
  for (int k = first; k < first + count; k++) {
    const Op& op = plan->ops[k];
    int rc = GCT2_OK;
    if (op.kind == OP_CALL) rc = ENTRIES[op.entry].call(op.a);
    else if (op.kind == OP_RECORD) {
      if (hipEventRecord(plan->events[op.event], op.stream) != hipSuccess) { (void)hipGetLastError(); rc = gct2_fail(GCT2_ELAUNCH, "plan_run: hipEventRecord failed"); }
    } else {
      if (hipStreamWaitEvent(op.stream, plan->events[op.event], 0) != hipSuccess) { (void)hipGetLastError(); rc = gct2_fail(GCT2_ELAUNCH, "plan_run: hipStreamWaitEvent failed"); }
    }
    if (rc != GCT2_OK) {          // the entry point's own message stays in gct2_last_error(); the caller learns which record it was
      if (failed) *failed = k;
      return rc;
    }
  }
  return GCT2_OK;
}

}  // extern "C"
