"""ctypes binding of csrc/libgct2.so (the C ABI declared in include/gct2.h).

There is NO fallback: if the shared library is missing or a call returns non-zero, this module
raises.  Nothing here imports `oracle/`.
"""
from __future__ import annotations

import ctypes as C
import os
import struct

F32, BF16, F16 = 0, 1, 2
DTYPE_NAMES = {F32: "f32", BF16: "bf16", F16: "f16"}
ABI_VERSION = 17
# gct2_diffusion_update modes (include/gct2.h; the sampler's objective switches, train.py:29-32)
SAMPLE_X, SAMPLE_EPS, SAMPLE_SCALED_EPS, SAMPLE_ODE = 0, 1, 2, 3
BUILD_STAMP = 1

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgct2.so")


class Gct2Error(RuntimeError):
    """non-zero status from libgct2 (mirrors TensorFlow raising on bad shapes)."""


class LossScaleState(C.Structure):
    """gct2_loss_scale_state (32 bytes): dynamic loss scale + the optimizer step counter it gates."""
    _fields_ = [("scale", C.c_float), ("inv_scale", C.c_float), ("good_steps", C.c_int32), ("found_inf", C.c_int32),
                ("applied_steps", C.c_int32), ("alpha", C.c_float), ("reserved", C.c_int32 * 2)]


class AdamArgs(C.Structure):
    """gct2_adam_args (include/gct2.h): the optimizer step fused behind a weight-gradient call."""
    _fields_ = [("p", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("shadow", C.c_void_p), ("shadow_dtype", C.c_int),
                ("n", C.c_size_t), ("alpha", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("grad_mul", C.c_float), ("defer", C.c_int), ("slab_base", C.c_void_p), ("nslab", C.c_int), ("slab_stride", C.c_size_t)]


_vp, _i, _f, _d, _u64, _sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_uint64, C.c_size_t

# name -> argtypes, exactly the prototypes of include/gct2.h
SIGNATURES = {
    "gct2_abi_version": [],
    "gct2_build_flags": [],
    "gct2_device_check": [],
    "gct2_stream_occupy": [_vp, _i, _d],
    "gct2_ctx_create": [C.POINTER(C.c_void_p)],
    "gct2_ctx_destroy": [_vp],
    "gct2_ctx_set_workspace": [_vp, _vp, _sz],
    "gct2_ctx_set_wgrad_workspace": [_vp, _vp, _sz],
    "gct2_ctx_set_bias_queue": [_vp, _vp, _sz],
    "gct2_bias_queue_flush": [_vp, _vp],
    "gct2_ctx_set_tuning": [_vp, _i],
    "gct2_ctx_force_direct": [_vp, _i],
    "gct2_ctx_set_stamp_buffer": [_vp, _vp, _sz],
    "gct2_ctx_set_relu_bits": [_vp, _vp, _i],
    "gct2_ctx_log_launches": [_vp, _i],
    "gct2_ctx_read_launch_log": [_vp, _vp, _sz, C.POINTER(C.c_size_t)],
    "gct2_diffusion_mix": [_i, _vp, _vp, _f, _vp, _vp, _i, _vp, _i, _sz, _i, _vp],
    "gct2_diffusion_update": [_i, _vp, _vp, _d, _d, _vp, _vp, _sz, _vp],
    "gct2_noise_edits": [_vp, _vp, _i, _vp, _i, _i, _i, _vp],
    "gct2_image_prepare": [_vp, _vp, _vp, _vp, _i, _i, _vp],
    "gct2_conv4s2_fwd": [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "gct2_conv4s2_dgrad": [_vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp],
    "gct2_conv4s2_wgrad": [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "gct2_adam_apply": [_vp, _vp, _sz, _vp],
    "gct2_convT4s2_fwd": [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "gct2_convT4s2_fwd_head_train": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i,
                                     _vp, _vp, _vp, _i, _i, _vp],
    "gct2_convT4s2_dgrad": [_vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp],
    "gct2_convT4s2_wgrad": [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "gct2_conv2d_s1_fwd": [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "gct2_conv2d_s1_dgrad": [_vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "gct2_conv2d_s1_wgrad": [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "gct2_relu_mask": [_i, _vp, _i, _vp, _i, _sz, _i, _vp],
    "gct2_add": [_i, _vp, _i, _vp, _i, _sz, _i, _vp],
    "gct2_mix_per_image": [_vp, _vp, _vp, _vp, _vp, _i, _sz, _vp],
    "gct2_dense_fwd": [_i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp],
    "gct2_dense_bwd": [_i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "gct2_dense_head_train": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp],
    "gct2_rng_uniform_int": [_u64, _u64, _u64, _vp, _sz, _i, _i, _vp],
    "gct2_rng_normal": [_u64, _u64, _u64, _vp, _sz, _vp],
    "gct2_noise_image": [_i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "gct2_noise_image_rng": [_i, _vp, _vp, _u64, _u64, _u64, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "gct2_mse_fwd_bwd": [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp],
    "gct2_adam_keras_multi": [_vp, _vp, _vp, _vp, _vp, _i, _sz, _f, _f, _f, _f, _f, _vp, _i, _vp],
    "gct2_cast_from_f32": [_i, _vp, _vp, _sz, _vp],
    "gct2_loss_scale_init": [_vp, _f, _vp],
    "gct2_loss_scale_begin": [_vp, _f, _i, _f, _f, _vp],
    "gct2_scale_check_finite": [_vp, _sz, _vp, _vp],
    "gct2_loss_scale_update": [_vp, _i, _vp],
    "gct2_plan_create": [C.POINTER(C.c_void_p)],
    "gct2_plan_destroy": [_vp],
    "gct2_plan_add_call": [_vp, C.c_char_p, C.POINTER(C.c_uint64), _i, C.POINTER(C.c_int)],
    "gct2_plan_add_record": [_vp, _vp, C.POINTER(C.c_int)],
    "gct2_plan_add_wait": [_vp, _vp, _i],
    "gct2_plan_add_record_kind": [_vp, _vp, _i, C.POINTER(C.c_int)],
    "gct2_plan_elapsed": [_vp, _i, _i, C.POINTER(C.c_float)],
    "gct2_plan_size": [_vp, C.POINTER(C.c_int)],
    "gct2_plan_set_arg": [_vp, _i, _i, _u64],
    "gct2_plan_run": [_vp, _i, _i, C.POINTER(C.c_int)],
}

_lib = None


def load() -> C.CDLL:
    """dlopen libgct2.so and attach prototypes; raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = LIB_PATH
    if os.environ.get("GCT2_ALLOW_DIAGNOSTIC_BUILD") == "1" and os.environ.get("GCT2_USE_STAMP_LIB") in ("1", "phases"):
        # `make -C csrc stamp` / `make -C csrc phases`: the diagnostic builds beside the product one (clock + work-group phases / the
        # same plus per-stage phase stamps inside wgrad256q_kernel's K loop, which slow that loop by a few percent)
        path = os.path.join(_HERE, "csrc", "libgct2_phases.so" if os.environ["GCT2_USE_STAMP_LIB"] == "phases" else "libgct2_stamp.so")
    if not os.path.exists(path):
        raise Gct2Error(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C gan-class-transfer2_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _i
    lib.gct2_last_error.argtypes = []
    lib.gct2_last_error.restype = C.c_char_p
    if lib.gct2_abi_version() != ABI_VERSION:
        raise Gct2Error(f"{LIB_PATH} has ABI version {lib.gct2_abi_version()}, this binding is written for {ABI_VERSION}: rebuild it "
                        "(`make -C gan-class-transfer2_amd/csrc`)")
    flags = lib.gct2_build_flags()
    if flags != 0 and os.environ.get("GCT2_ALLOW_DIAGNOSTIC_BUILD") != "1":
        raise Gct2Error(f"{LIB_PATH} is a DIAGNOSTIC build (gct2_build_flags() = {flags}: in-kernel stamps): results and timings of "
                        "such a library are never product numbers.  Rebuild with `make -C gan-class-transfer2_amd/csrc clean all`, or "
                        "set GCT2_ALLOW_DIAGNOSTIC_BUILD=1 for the scripts/stamp_*.py diagnostics")
    _lib = lib
    return lib


def build_flags() -> int:
    return int(load().gct2_build_flags())


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().gct2_last_error().decode("utf-8", "replace")
        raise Gct2Error(f"{what or 'gct2'} failed (status {rc}): {msg}")


class Slot:
    """an argument whose value changes from step to step (the input batch pointer, RNG offsets, the optimizer's alpha): `call`
    passes the value through; a Plan that records the call remembers (record, argument) under `key` and re-sets it before a run"""
    __slots__ = ("key", "value")

    def __init__(self, key: str, value):
        self.key, self.value = key, value


def call(name: str, *args) -> None:
    if _recording is not None:
        _recording.add_call(name, args)
        if not _recording.execute or name not in PLANNABLE:
            return
    check(getattr(load(), name)(*[a.value if type(a) is Slot else a for a in args]), name)


# the entry points a plan can hold (csrc/plan.hip ENTRIES): everything that enqueues work on a stream + the one-shot ReLU plane
PLANNABLE = frozenset(n for n, sig in SIGNATURES.items() if n == "gct2_ctx_set_relu_bits" or (
    not n.startswith(("gct2_ctx_", "gct2_plan_", "gct2_loss_scale_init")) and n not in ("gct2_abi_version", "gct2_build_flags", "gct2_device_check", "gct2_stream_occupy")))
_recording = None      # the Plan that is recording calls right now (one host thread drives an engine: _lib.call is not re-entrant)
_FLOAT_STRUCT = struct.Struct("<f")
_DOUBLE_STRUCT = struct.Struct("<d")


def _slot_bits(ctype, v) -> int:
    """one argument as the 64-bit slot of include/gct2.h's plan calls"""
    if ctype is _f:
        return int.from_bytes(_FLOAT_STRUCT.pack(float(v)), "little")
    if ctype is _d:
        return int.from_bytes(_DOUBLE_STRUCT.pack(float(v)), "little")
    if v is None:
        return 0
    return int(v) & 0xFFFFFFFFFFFFFFFF


EVENT_DEVICE, EVENT_SYSTEM, EVENT_TIMED = 0, 1, 2        # include/gct2.h GCT2_EVENT_*


class Plan:
    """gct2_plan (include/gct2.h): a recorded list of entry-point calls, event records and stream waits that one C call replays.

        plan = Plan(); plan.begin(execute=True)     # every _lib.call(...) from now on is appended (and, with execute, also run)
        ...                                          # plan.record(stream) / plan.wait(stream, ev) / plan.cut(tag) from the host code
        plan.end()
        plan.set("x", ptr); plan.run_segment(k)      # per step

    Segments: plan.cut(payload) closes the current segment; the caller runs segment k, does its own work for payload k (the
    data-parallel hooks), then segment k + 1."""

    def __init__(self):
        h = C.c_void_p()
        check(load().gct2_plan_create(C.byref(h)), "gct2_plan_create")
        self.handle = h.value
        self.execute = True
        self.slots = {}            # key -> [(record, argument, ctype)]
        self.values = {}           # key -> last value set
        self.cuts = []             # (first record of the NEXT segment, payload)
        self.keep = []             # objects the records point into (gct2_adam_args structs, tensors)
        self.n = 0

    # ---- recording ----------------------------------------------------------------------------------------------------------
    def begin(self, execute: bool = True) -> None:
        global _recording
        if _recording is not None:
            raise Gct2Error("a plan is already recording")
        self.execute = execute
        _recording = self

    def end(self) -> None:
        global _recording
        _recording = None
        self.cuts.append((self.n, None))

    def add_call(self, name: str, args) -> None:
        if name not in PLANNABLE:               # host-only state changes (context setters): done now, they persist until the replay
            check(getattr(load(), name)(*[a.value if type(a) is Slot else a for a in args]), name)
            return
        types = SIGNATURES[name]
        if len(types) != len(args):
            raise Gct2Error(f"{name}: {len(args)} arguments for {len(types)} parameters")
        arr = (C.c_uint64 * len(args))()
        for i, (t, a) in enumerate(zip(types, args)):
            if type(a) is Slot:
                self.slots.setdefault(a.key, []).append((self.n, i, t))
                self.values[a.key] = a.value
                a = a.value
            arr[i] = _slot_bits(t, a)
        idx = C.c_int(-1)
        check(load().gct2_plan_add_call(self.handle, name.encode(), arr, len(args), C.byref(idx)), "gct2_plan_add_call")
        self.n = idx.value + 1

    def record(self, stream: int, kind: int = EVENT_DEVICE) -> int:
        """appends "record a new event on `stream`"; kind: EVENT_DEVICE (ordering inside the device), EVENT_SYSTEM (the waiter hands the
        data to another device: the communication stream of the data-parallel exchange), EVENT_TIMED (elapsed_ms)"""
        ev = C.c_int(-1)
        check(load().gct2_plan_add_record_kind(self.handle, stream, kind, C.byref(ev)), "gct2_plan_add_record_kind")
        self.n += 1
        return ev.value

    def elapsed_ms(self, start: int, end: int) -> float:
        """milliseconds between two EVENT_TIMED records of the last run (both completed: synchronise first)"""
        ms = C.c_float(0.0)
        check(load().gct2_plan_elapsed(self.handle, start, end, C.byref(ms)), "gct2_plan_elapsed")
        return ms.value

    def wait(self, stream: int, event: int) -> None:
        check(load().gct2_plan_add_wait(self.handle, stream, event), "gct2_plan_add_wait")
        self.n += 1

    def cut(self, payload) -> None:
        self.cuts.append((self.n, payload))

    # ---- replay -------------------------------------------------------------------------------------------------------------
    def set(self, key: str, value) -> None:
        if self.values.get(key) == value:
            return
        self.values[key] = value
        L = load()
        for rec, arg, t in self.slots.get(key, ()):
            check(L.gct2_plan_set_arg(self.handle, rec, arg, _slot_bits(t, value)), "gct2_plan_set_arg")

    def segments(self) -> int:
        return len(self.cuts)

    def run_segment(self, k: int):
        """runs segment k and returns the payload of the cut that ends it (None for the last)"""
        first = self.cuts[k - 1][0] if k else 0
        last, payload = self.cuts[k]
        if last > first:
            failed = C.c_int(-1)
            rc = load().gct2_plan_run(self.handle, first, last - first, C.byref(failed))
            if rc != 0:
                check(rc, f"gct2_plan_run (record {failed.value})")
        return payload

    def __del__(self):
        try:
            if getattr(self, "handle", None) and _lib is not None:
                _lib.gct2_plan_destroy(self.handle)
                self.handle = None
        except Exception:       # interpreter shutdown
            pass


class Context:
    """gct2_ctx (include/gct2.h): the caller-owned scratch and tile-selection knobs that the convolution / head entry points
    take as their first argument.  The device tensors stay owned by whoever passes them in (kept alive here by reference);
    two Context objects never share anything, so two engines / threads / streams are independent."""

    def __init__(self):
        h = C.c_void_p()
        call("gct2_ctx_create", C.byref(h))
        self.handle = h.value
        self._keep = [None, None, None]
        # bumped by every setter: whoever caches something that bakes in this context's pointers or tile choices (the sampler's
        # HIP graphs of the forward pass) keys its cache on it
        self.version = 0
        self.tuning = 0
        self.direct = False
        self.logging = False

    def set_relu_bits(self, ptr, ld_bytes: int) -> None:
        """ReLU bit plane for the NEXT forward (written) / input-gradient (read instead of act) call of this context; one-shot.
        `ptr` is a device address inside a caller-owned uint8 tensor [pixels, ld_bytes] (None clears a pending plane)."""
        call("gct2_ctx_set_relu_bits", self.handle, ptr, int(ld_bytes) if ptr is not None else 0)

    def set_stamp_buffer(self, tensor) -> None:
        """diagnostic builds only (gct2_build_flags() & BUILD_STAMP): where the next stamped launch writes its phase stamps."""
        self._keep[2] = tensor
        self.version += 1
        call("gct2_ctx_set_stamp_buffer", self.handle, tensor.data_ptr() if tensor is not None else None,
             tensor.numel() * tensor.element_size() if tensor is not None else 0)

    def log_launches(self, on: bool = True) -> None:
        """launch log of this context (include/gct2.h): clears it and switches it on / off."""
        self.logging = bool(on)
        call("gct2_ctx_log_launches", self.handle, int(bool(on)))

    def read_launch_log(self) -> list:
        """the kernels the layer calls of this context selected since the last read, as text tokens"""
        need = C.c_size_t(0)
        buf = C.create_string_buffer(1 << 12)
        rc = load().gct2_ctx_read_launch_log(self.handle, C.cast(buf, C.c_void_p), len(buf), C.byref(need))
        if rc != 0 and need.value > len(buf):            # too small: nothing was cleared, the call said how much it needs
            buf = C.create_string_buffer(need.value)
            rc = load().gct2_ctx_read_launch_log(self.handle, C.cast(buf, C.c_void_p), len(buf), C.byref(need))
        check(rc, "gct2_ctx_read_launch_log")
        return [t for t in buf.value.decode().split(";") if t]

    def mirror(self, other: "Context") -> None:
        """take over the steering / observing state of another context (not its scratch): tile knobs, the direct-kernel switch, whether
        launches are logged, the diagnostic stamp buffer"""
        if self.tuning != other.tuning:
            self.set_tuning(other.tuning)
        if self.direct != other.direct:
            self.force_direct(other.direct)
        if self.logging != other.logging:
            self.log_launches(other.logging)
        if self._keep[2] is not other._keep[2]:
            self.set_stamp_buffer(other._keep[2])

    def set_workspace(self, tensor) -> None:
        self._keep[0] = tensor
        self.version += 1
        call("gct2_ctx_set_workspace", self.handle, tensor.data_ptr() if tensor is not None else None,
             tensor.numel() * tensor.element_size() if tensor is not None else 0)

    def set_wgrad_workspace(self, tensor) -> None:
        self._keep[1] = tensor
        self.version += 1
        call("gct2_ctx_set_wgrad_workspace", self.handle, tensor.data_ptr() if tensor is not None else None,
             tensor.numel() * tensor.element_size() if tensor is not None else 0)

    def set_bias_queue(self, tensor) -> None:
        """queue buffer for the deferred bias-gradient row sums of this context's input-gradient calls (None: immediate reductions)"""
        self._keep.append(tensor)
        self.version += 1
        call("gct2_ctx_set_bias_queue", self.handle, tensor.data_ptr() if tensor is not None else None,
             tensor.numel() * tensor.element_size() if tensor is not None else 0)

    def set_tuning(self, v: int) -> None:
        self.version += 1
        self.tuning = int(v)
        call("gct2_ctx_set_tuning", self.handle, int(v))

    def force_direct(self, on: bool) -> None:
        self.version += 1
        self.direct = bool(on)
        call("gct2_ctx_force_direct", self.handle, int(bool(on)))

    def __del__(self):
        try:
            if getattr(self, "handle", None) and _lib is not None:
                _lib.gct2_ctx_destroy(self.handle)
                self.handle = None
        except Exception:       # interpreter shutdown
            pass
