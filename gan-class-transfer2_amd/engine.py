"""Execution plan of one `Trainer` train step (train.py:223-272 + Keras fit/Adam, train.py:511-517)
over the C ABI of include/gct2.h.

HBM layout (all NHWC, `ld` = pixel stride in elements):
  R_i   [B, H/2^i, W/2^i, ld_i]   the output of Residual_i = concat([U_i(..), x_i], -1) (train.py:113-119):
        channels [0, Fu_i) are written by UpShuffle_i, channels [Fu_i, Fu_i+C_i) by the producer of x_i
        (DownShuffle_{i-1}, or the noising kernel for i = 0)  ->  the concat is never materialised.
  dR_i  same shape: gradient w.r.t. the PRE-activation of every channel of R_i (ReLU masks are applied by
        the kernel that writes the gradient), so consumers (wgrad, dgrad) read it unmasked.
  parameters: one fp32 arena (p, m, v, g) + a compute-dtype shadow arena, tensors ordered by backward
        completion (dense, U_0..U_{n-1}, D_{n-1}..D_0) so gradient buckets are contiguous ranges.
"""
from __future__ import annotations

import ctypes
import math
import os
import threading
from collections import OrderedDict
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import BF16, F16, F32, Slot, call

TORCH_DTYPE = {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}


def _round_up(a: int, b: int) -> int:
    return (a + b - 1) // b * b


@dataclass
class Topology:
    """channel rule of Denoiser.__init__ (train.py:179-190)."""
    pixel_size: int = 128
    max_size: int = 512
    octaves: int = 6

    def fd(self, i: int) -> int:
        return min(self.pixel_size * 2 ** i, self.max_size)

    def fu(self, i: int) -> int:
        return min(self.pixel_size * 2 ** i // 2, self.max_size)

    def cx(self, i: int) -> int:
        return 3 if i == 0 else self.fd(i - 1)

    def up_in(self, i: int) -> int:
        return self.fd(i) if i == self.octaves - 1 else self.fu(i + 1) + self.fd(i)

    def param_shapes(self) -> Dict[str, Tuple[int, ...]]:
        s: Dict[str, Tuple[int, ...]] = {}
        for i in range(self.octaves):
            s[f"D{i}.w"] = (4, 4, self.cx(i), self.fd(i))
            s[f"D{i}.b"] = (self.fd(i),)
            s[f"U{i}.w"] = (4, 4, self.fu(i), self.up_in(i))
            s[f"U{i}.b"] = (self.fu(i),)
        s["dense.w"] = (self.fu(0) + 3, 3)
        s["dense.b"] = (3,)
        return s

    def backward_order(self) -> List[str]:
        names = ["dense.w", "dense.b"]
        for i in range(self.octaves):
            names += [f"U{i}.w", f"U{i}.b"]
        for i in reversed(range(self.octaves)):
            names += [f"D{i}.w", f"D{i}.b"]
        return names

    def layer_order(self) -> List[str]:
        """layers in backward completion order; one gradient bucket per layer."""
        return ["dense"] + [f"U{i}" for i in range(self.octaves)] + [f"D{i}" for i in reversed(range(self.octaves))]


class ParamArena:
    ALIGN = 64  # elements; keeps every tensor 16-byte aligned in the 16-bit shadow too

    def __init__(self, topo: Topology, dtype: int, device: torch.device):
        self.topo, self.dtype, self.device = topo, dtype, device
        self.shapes = topo.param_shapes()
        self.offsets: Dict[str, int] = {}
        off = 0
        # Layout (r04): the convolution kernels first, in backward completion order (U_0 .. U_{n-1}, D_{n-1} .. D_0: a gradient bucket
        # is a contiguous range), then ONE zone with every parameter the kernels read in FP32 straight from the master arena - the
        # Dense(3) kernel and bias and all convolution biases (a few thousand floats).  A layer's fused optimizer launch covers its
        # kernel; the zone gets one small launch once the last input gradient (the last writer of a bias gradient) is enqueued.  For
        # the sharded data-parallel step this puts the fp32-read parameters into the LAST bucket, which is all-reduced and updated on
        # every rank (replicated): no exchange of fp32 values, no second collective behind the last bucket (r03 needed both).
        self.layer_ranges: Dict[str, Tuple[int, int]] = {}
        convs = [l for l in topo.layer_order() if l != "dense"]
        for layer in convs:
            lo = off
            self.offsets[layer + ".w"] = off
            off = _round_up(off + int(np.prod(self.shapes[layer + ".w"])), self.ALIGN)
            self.layer_ranges[layer] = (lo, off)
        zone_lo = off
        for name in ["dense.w", "dense.b"] + [l + ".b" for l in convs]:
            self.offsets[name] = off
            off = _round_up(off + int(np.prod(self.shapes[name])), self.ALIGN)
            if name == "dense.b":
                self.layer_ranges["dense"] = (zone_lo, off)
        off = _round_up(off, 64 * self.ALIGN)       # the arena splits into equal 16-byte aligned shards for up to 64 ranks
        self.layer_ranges["fp32"] = (zone_lo, off)  # (contains the "dense" range)
        self.total = off
        z = lambda dt: torch.zeros(self.total, dtype=dt, device=device)
        self._p, self._m, self._v, self.g = z(torch.float32), z(torch.float32), z(torch.float32), z(torch.float32)
        self._shadow = z(TORCH_DTYPE[dtype]) if dtype != F32 else None
        # the engine may hold optimizer launches back (UNetEngine.defer_adam: the Adam step of the layers the forward pass needs last
        # runs inside the NEXT forward pass): whoever looks at p / m / v / shadow through these properties gets them flushed first
        self.before_read: Optional[Callable[[], None]] = None

    def _sync(self) -> None:
        if self.before_read is not None:
            self.before_read()

    @property
    def p(self) -> torch.Tensor:
        self._sync(); return self._p

    @property
    def m(self) -> torch.Tensor:
        self._sync(); return self._m

    @property
    def v(self) -> torch.Tensor:
        self._sync(); return self._v

    @property
    def shadow(self) -> Optional[torch.Tensor]:
        self._sync(); return self._shadow

    LAYOUT = 2        # bumped when the order of tensors inside the arenas changes (state_dict carries it)

    @staticmethod
    def legacy_offsets(topo: "Topology") -> Tuple[Dict[str, int], int]:
        """tensor offsets of the r01-r03 arena layout ([kernel | bias] per layer, backward completion order, dense first) and its
        length - what checkpoints written before ParamArena.LAYOUT existed hold (UNetEngine.load_state_dict converts them)"""
        shapes, offs, off = topo.param_shapes(), {}, 0
        for layer in topo.layer_order():
            for suffix in (".w", ".b"):
                offs[layer + suffix] = off
                off = _round_up(off + int(np.prod(shapes[layer + suffix])), ParamArena.ALIGN)
        return offs, _round_up(off, 64 * ParamArena.ALIGN)

    def ready_order(self) -> List[str]:
        """arena ranges in the order their gradients complete during the reverse pass: the convolution layers (each ready hook
        fires when the layer's weight gradient is enqueued), then the fp32 zone (complete with the last input gradient)."""
        return [l for l in self.topo.layer_order() if l != "dense"] + ["fp32"]

    def numel(self, name: str) -> int:
        return int(np.prod(self.shapes[name]))

    def _view(self, arena: torch.Tensor, name: str) -> torch.Tensor:
        o = self.offsets[name]
        return arena[o:o + self.numel(name)].view(self.shapes[name])

    def param(self, name): return self._view(self.p, name)
    def grad(self, name): return self._view(self.g, name)
    def slot_m(self, name): return self._view(self.m, name)
    def slot_v(self, name): return self._view(self.v, name)

    def wptr(self, name: str) -> int:
        """device pointer of the compute-dtype operand copy of a weight."""
        o = self.offsets[name]                    # (raw storage: the engine orders its own launches against pending updates)
        if self._shadow is None:
            return self._p.data_ptr() + 4 * o
        return self._shadow.data_ptr() + 2 * o

    def pptr(self, name: str) -> int:
        return self._p.data_ptr() + 4 * self.offsets[name]

    def gptr(self, name: str) -> int:
        return self.g.data_ptr() + 4 * self.offsets[name]

    def refresh_shadow(self, stream: int) -> None:
        if self.shadow is not None:
            call("gct2_cast_from_f32", self.dtype, self._p.data_ptr(), self._shadow.data_ptr(), self.total, stream)

    def glorot_init(self, seed: int = 1234) -> None:
        """Keras glorot_uniform kernels, zero biases (train.py:134,149,162)."""
        gen = torch.Generator(device="cpu").manual_seed(seed)
        for name in sorted(self.shapes):
            shp = self.shapes[name]
            if name.endswith(".b"):
                self.param(name).zero_()
                continue
            if len(shp) == 2:
                fan_in, fan_out = shp
            else:
                rf = shp[0] * shp[1]
                fan_in, fan_out = rf * shp[2], rf * shp[3]
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            w = (torch.rand(shp, generator=gen, dtype=torch.float32) * 2 - 1) * lim
            self.param(name).copy_(w.to(self.device))


class _Buffers:
    pass


class _StepPlan:
    """one recorded train step (UNetEngine._planned_step): the gct2_plan, the buffer set it was recorded on, the host-side state the
    step leaves behind"""

    def __init__(self, b: _Buffers):
        self.plan = _lib.Plan()
        self.b = b                     # (holds the buffer set: its id() is part of the plan's key)
        self.loss = None
        self.after: dict = {}
        self.d_off_t = self.d_off_eps = self.d_its = 0
        self.adam_inputs: dict = {}
        self.x_ref = None


# ---- stream placement -------------------------------------------------------------------------------------------------------------
# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES (4) hardware queues in the order the streams are first
# used, and two streams on one queue block each other for the whole run.  Which queue a new stream lands on depends on how many
# streams the process used before - the collective library's included - so "the next stream of the pool" made the same engine run
# 2.49 or 2.78 ms per step depending on its position in the process (scripts/probe_stream_queues.py, profiles/r06_stream_queues.txt:
# pool streams 0 and 4 share the caller's queue, 1 and 5 each other's, ...).  The streams an engine and its data-parallel wrapper
# use are therefore (a) PROBED: a candidate is taken only if a marker on it does not wait for a 300-us occupy on any stream it has
# to run beside (gct2_stream_occupy), and (b) REGISTERED per (device, caller stream, role): the third engine of a process runs on
# the same streams as the first.
_STREAMS: Dict[tuple, "torch.cuda.Stream"] = {}
_STREAMS_LOCK = threading.Lock()    # (engines may be built from several host threads: one probe at a time, one registry)
_STREAM_LOG: list = []              # (role, candidate index, microseconds the marker waited per avoided stream) - scripts print it


def _marker_delay_us(busy: "torch.cuda.Stream", cand: "torch.cuda.Stream", device: torch.device, occupy_us: float = 300.0) -> float:
    """microseconds between the start of an occupy on `busy` and the end of a trivial launch enqueued on `cand` right behind it:
    a few us when the two streams sit on different hardware queues, >= occupy_us when they share one"""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(device)
    e0.record(busy)
    call("gct2_stream_occupy", busy.cuda_stream, 1, float(occupy_us))
    call("gct2_stream_occupy", cand.cuda_stream, 1, 1.0)
    e1.record(cand)
    cand.synchronize()
    busy.synchronize()
    return e0.elapsed_time(e1) * 1e3


def distinct_stream(device: torch.device, role: str, caller: "torch.cuda.Stream", priority: int = 0, tries: int = 8) -> "torch.cuda.Stream":
    """the stream of `role` ("side", "chain", "comm") for work launched beside `caller` on `device`: registered once per (device,
    caller, role); a new one is the first of up to `tries` pool streams that shares a hardware queue neither with the caller nor
    with the streams already registered for this caller (GCT2_STREAM_PROBE=0: the first candidate, unprobed)."""
    key = (device.index, caller.cuda_stream, role)
    st = _STREAMS.get(key)
    if st is not None:
        return st
    with _STREAMS_LOCK:
        return _pick_stream(key, device, role, caller, priority, tries)


def _pick_stream(key: tuple, device: torch.device, role: str, caller: "torch.cuda.Stream", priority: int, tries: int) -> "torch.cuda.Stream":
    st = _STREAMS.get(key)              # (another thread may have registered it while this one waited for the lock)
    if st is not None:
        return st
    avoid = [caller] + [v for (d, c, _), v in _STREAMS.items() if d == device.index and c == caller.cuda_stream]
    probe = os.environ.get("GCT2_STREAM_PROBE", "1") != "0"
    best, best_wait = None, None
    for k in range(tries if probe else 1):
        cand = torch.cuda.Stream(device=device, priority=priority)
        if any(cand.cuda_stream == a.cuda_stream for a in avoid):
            continue
        waits = [_marker_delay_us(a, cand, device) for a in avoid] if probe else []
        _STREAM_LOG.append((role, k, [round(w, 1) for w in waits]))
        worst = max(waits) if waits else 0.0
        if best is None or worst < best_wait:
            best, best_wait = cand, worst
        if worst < 150.0:                                  # (half of the occupy: a shared queue shows >= 300)
            break
    if best is None:
        best = torch.cuda.Stream(device=device, priority=priority)
    _STREAMS[key] = best
    return best


_ADAM_INPUT_FIELDS = ("p", "m", "v", "shadow", "shadow_dtype", "n", "beta1", "beta2", "eps", "grad_mul", "defer")


class UNetEngine:
    """the planned (zero-copy concat) forward / backward / optimizer step of the Denoiser U-Net."""

    def __init__(self, topo: Topology, dtype: int = BF16, device: Optional[torch.device] = None, steps: int = 200,
                 base_lr: float = 2e-5, warm_up: int = 2000, beta_1: float = 0.9, beta_2: float = 0.999,
                 epsilon: float = 1e-7, loss_scaling: bool = False, seed: int = 1234, rng_seed: int = 0,
                 workspace_mb: int = 64, predict_x: bool = True, predict_scaled_epsilon: bool = False,
                 prediction_weighting: bool = False, ordinary_differential_equation: bool = False):
        self.lib = _lib.load()
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        if self.device.type != "cuda":
            raise _lib.Gct2Error("UNetEngine needs a HIP device (torch device 'cuda'); there is no CPU path")
        if self.device.index is None:                       # ("cuda" without an index: the current device - the stream registry keys on it)
            self.device = torch.device("cuda", torch.cuda.current_device())
        call("gct2_device_check")
        self.topo, self.dtype, self.steps = topo, dtype, steps
        self.base_lr, self.warm_up = base_lr, warm_up
        self.beta_1, self.beta_2, self.epsilon = beta_1, beta_2, epsilon
        self.loss_scaling = loss_scaling
        # the objective switches of train.py:29-32 (defaults = the reference's: the network predicts the clean image)
        self.predict_x, self.predict_scaled_epsilon = predict_x, predict_scaled_epsilon
        self.prediction_weighting, self.ordinary_differential_equation = prediction_weighting, ordinary_differential_equation
        self.use_fused_head = True     # False: dense_fwd + mse_fwd_bwd + dense_bwd as three kernels
        self.fuse_u0_head = True       # train step: the head runs in UpShuffle_0's forward epilogue (R_0 is never written)
        self.keep_pred = False         # train step: also store the prediction (buffers().pred); the loss does not need it
        # train step: the forward epilogues of the large levels also write ReLU bit planes (1 bit per activation) and the input-gradient
        # epilogues read their masks from them instead of re-reading the activations (gct2_ctx_set_relu_bits; -16 bits per element read)
        self.relu_bits = True
        self.relu_bits_min_bytes = 32 << 20     # planes only for activation tensors of at least this size (see buffers())
        self.arena = ParamArena(topo, dtype, self.device)
        self.arena.glorot_init(seed)
        self.arena.refresh_shadow(self._stream())
        self._iterations = 0           # optimizer.iterations [TF] (with loss scaling the counter lives on the device)
        self.rng_seed, self.rng_offset_t, self.rng_offset_eps = rng_seed, 0, 0
        # activation / gradient buffer sets by (B, H, W), least recently used first; at most `max_buffer_sets` stay allocated (the
        # train step, the sampler's batch 1 and batch 6 sets and one spare: a set at config 3 is ~0.8 GB, r02 kept every shape ever seen)
        self._bufs: "OrderedDict[Tuple[int, int, int], _Buffers]" = OrderedDict()
        self.max_buffer_sets = 4
        # the call context (include/gct2.h gct2_ctx): this engine's scratch tensors and tile knobs; nothing is process-wide
        self.ctx = _lib.Context()
        # split-K scratch for the bottleneck layers, partial rows of the fused bias gradients / the head; caller-owned = this tensor
        self.workspace = torch.empty(workspace_mb << 18, dtype=torch.float32, device=self.device) if workspace_mb else None
        # second stream + its own scratch: the weight-gradient kernels (and, single-GPU, the per-layer Adam launches) run
        # beside the dgrad chain instead of between its links (backward())
        self.overlap = True
        self.fuse_adam = True          # per-layer Adam fused behind the weight-gradient calls (single replica, no loss scaling)
        # (a stream that shares no hardware queue with the caller's; the same one for every engine of this caller: distinct_stream)
        self._caller0 = torch.cuda.current_stream(self.device)
        self._side = distinct_stream(self.device, "side", self._caller0)
        # chain_priority: run the dgrad chain (the only true dependency chain of the reverse pass, with its short split-K finalize /
        # row-sum launches) on a HIGH-priority stream of its own, so that its work-groups are dispatched ahead of the side stream's
        # queued weight-gradient work-groups.  Worth -2 % when it was introduced (r02.c); since the image layer's weight gradient
        # left the tail and the bottleneck weight gradients shrank it measures +0.8 % in an in-process A/B (scripts/bench_phases.py,
        # GCT2_AB=chain_priority: 2.743 vs 2.765 ms), and every additional stream is a hazard once more than GPU_MAX_HW_QUEUES = 4
        # are in use (distributed._one_stream_less): off by default, the stream is created on first use
        self.chain_priority = False
        self._chain_stream = None
        self.wgrad_workspace = torch.empty(workspace_mb << 18, dtype=torch.float32, device=self.device) if workspace_mb else None
        self.ctx.set_workspace(self.workspace)
        self.ctx.set_wgrad_workspace(self.wgrad_workspace)
        # the image layer's weight gradient (HBM-bound, no LDS) is enqueued on the dgrad chain's stream, which has nothing left to
        # do by then, so that it runs BESIDE DownShuffle_1's MFMA-bound weight gradient instead of behind it at the very end of the
        # step; its slabs go to the chain's own scratch (free once the last input gradient is done): a second context
        self.tail_on_chain = True
        self.ctx_tail = _lib.Context()
        self.ctx_tail.set_workspace(self.workspace)
        # deferred optimizer steps (r04): in the fused single-replica step the Adam launches of the layers the forward pass needs LAST
        # (the head and the outer UpShuffle levels: 5.5 M of the 41.7 M parameters, but three of the six 64-MiB slab sets) are not run
        # behind their weight gradients - beside other full-chip GEMMs, where HBM-bound work costs its full duration - but inside the
        # NEXT forward pass's bottleneck window (DownShuffle_3 .. UpShuffle_3: latency-bound launches that leave the memory system
        # idle), on the side stream, finished before UpShuffle_2 reads its weights.  Same launches, same bits; everything that looks at
        # the parameters (arena properties, state_dict, predict, the sampler) flushes them first.  The slabs of a deferred layer must
        # outlive the step: such layers get a call context of their own (_defer_ctx) whose weight-gradient scratch nobody else uses.
        self.defer_adam = True
        self.defer_layers = ("U0", "U1", "U2")
        self.defer_window_at = 3          # the held-back launches start once DownShuffle_<this> of the next forward pass is enqueued
        # r05: the eleven small launches that sum the partial rows of the fused bias gradients leave the input-gradient chain - the calls
        # queue their rows (gct2_ctx_set_bias_queue) and ONE flush (two launches) behind the last input gradient sums them: same bits
        self.defer_rowsums = True
        # EXPERIMENT (r06; off): the outer `split_forward_levels` levels of the forward pass of a train step run as
        # TWO half-batch chains on two streams (convolutions are image-local, train.py:148-166), so that the tail of one chain's launch
        # overlaps the head of the other's - prices the launch ramp / drain of those levels with the kernels that exist
        self.split_forward_levels = 0
        self._split_ctxs: Dict[int, "_lib.Context"] = {}
        self._bias_queue = None
        self._bias_queue_on = False
        self._pending: list = []
        self._pending_event = None
        self._flush_event = None
        self._pending_names: set = set()
        self._defer_ctxs: Dict[str, "_lib.Context"] = {}
        self._defer_ws: Dict[str, torch.Tensor] = {}
        self.arena.before_read = self.flush_deferred
        # one gct2_adam_args per layer for the engine's lifetime (filled in place every step): a step plan bakes their addresses
        self._adam_args: Dict[str, "_lib.AdamArgs"] = {}
        # step plans (r05, include/gct2.h gct2_plan): the C-ABI calls, event records and stream waits of a train step are recorded ONCE
        # per step shape and replayed by one C call per step (per segment when a gradient-ready hook is installed) instead of ~90
        # interpreter round trips; same calls, same arguments, same streams - same bits (tests/test_step_gpu.py).  Steps with injected
        # t_int / eps or a non-default objective, and anything outside train_step, stay eager.
        self.use_plan = True
        self._plans: Dict[tuple, "_StepPlan"] = {}
        self._plan_seen: Dict[tuple, int] = {}
        self.plan_after = 2              # a step shape is recorded once it has been seen this many times (buffers and deferrals warmed)
        self.ls_state = None
        if loss_scaling:
            self.enable_loss_scaling()
        # called with the layer name as soon as that layer's gradient kernels are enqueued (DP all-reduce)
        self.grad_ready_hook: Optional[Callable[[str], None]] = None
        # plan-aware hooks (the data-parallel wrappers): the hook does its stream plumbing through _wait_stream / apply_adam(stream=...)
        # and issues its collectives through host_call(), so it can be RECORDED with the step - a replay then runs only the collectives
        # from the interpreter.  A plain hook (False) is called between two plan segments at every step instead.
        self.hook_plan_aware = False
        # called once per train_step(apply=False) right after the reverse pass, inside the (recorded) step body: the wrappers' tail
        # (remaining bucket updates, joining the communication stream, finish_step)
        self.post_backward: Optional[Callable[[], None]] = None
        self.post_backward_ran = False
        # called at the end of every REPLAYED step: a plan-aware wrapper's Python state (bucket counters) is advanced by its hooks, which
        # only run while a step is being recorded - it puts that state where a full step leaves it (distributed.py)
        self.post_replay: Optional[Callable[[], None]] = None
        self._grads_in_arena = True    # False after a step whose fused optimizer consumed weight-gradient slabs in place

    # ------------------------------------------------------------------------------------------
    def enable_loss_scaling(self, initial_scale: float = 2.0 ** 15) -> None:
        """tf.keras.mixed_precision.LossScaleOptimizer (train.py:82-83): allocate the device-side state (32-byte
        gct2_loss_scale_state).  Allowed until the first optimizer step, so `trainer(example)` may come before `compile`
        exactly as in train.py:505-514."""
        if self.ls_state is not None:
            return
        if self._iterations != 0:
            raise _lib.Gct2Error("loss scaling cannot be switched on after optimizer steps have been applied")
        self.loss_scaling = True
        self.ls_state = torch.zeros(8, dtype=torch.int32, device=self.device)
        call("gct2_loss_scale_init", self.ls_state.data_ptr(), float(initial_scale), self._stream())

    @property
    def iterations(self) -> int:
        """optimizer.iterations [TF].  Under LossScaleOptimizer a skipped step does not advance it, and whether a step was
        skipped is only known on the device: the counter lives there (gct2_loss_scale_state.applied_steps) and reading it
        synchronises."""
        if self.ls_state is not None:
            return int(self.ls_state[4].item())
        return self._iterations

    @iterations.setter
    def iterations(self, k: int) -> None:
        self._iterations = int(k)
        if self.ls_state is not None:
            self.ls_state[4] = int(k)

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def stream_for(self, role: str) -> "torch.cuda.Stream":
        """the registered stream of `role` beside this engine's caller stream (the data-parallel wrappers ask for "comm")"""
        return distinct_stream(self.device, role, self._caller0)

    # ---- stream plumbing: torch events when the step runs eagerly, plan records while a step plan is being recorded ---------------
    def _mark(self, stream: "torch.cuda.Stream", system_scope: bool = False):
        """a point on `stream` that another stream can wait for.  system_scope: the waiter hands the data to another device (the
        communication stream in front of a collective) - a recorded event then releases at system scope (torch's events always do);
        the waits between the step's own streams keep the cheaper device-scope release they were measured with"""
        P = _lib._recording
        if P is not None:
            return P.record(stream.cuda_stream, _lib.EVENT_SYSTEM if system_scope else _lib.EVENT_DEVICE)
        ev = torch.cuda.Event()
        ev.record(stream)
        return ev

    def _wait(self, stream: "torch.cuda.Stream", token) -> None:
        P = _lib._recording
        if P is not None:
            P.wait(stream.cuda_stream, token)
        else:
            stream.wait_event(token)

    def _wait_stream(self, waiter: "torch.cuda.Stream", other: "torch.cuda.Stream", system_scope: bool = False) -> None:
        self._wait(waiter, self._mark(other, system_scope))

    # ---- deferred optimizer steps ------------------------------------------------------------------------------------------
    def _defer_ctx(self, layer: str) -> "_lib.Context":
        """the call context of a deferred layer's weight gradient: its own slab scratch, everything else mirrored from self.ctx
        (tile knobs, the direct-kernel switch, the launch log, the diagnostic stamp buffer) so that whoever steers or observes the
        step through self.ctx steers and observes these launches too"""
        c = self._defer_ctxs.get(layer)
        if c is None:
            c = self._defer_ctxs[layer] = _lib.Context()
            self._defer_ws[layer] = torch.empty_like(self.wgrad_workspace)
            c.set_wgrad_workspace(self._defer_ws[layer])
        c.mirror(self.ctx)
        return c

    def read_launch_log(self) -> list:
        """launch tokens of every context this engine drives (the main one, the chain-tail one, the deferred layers'), main first"""
        out = self.ctx.read_launch_log() + self.ctx_tail.read_launch_log()
        for c in self._defer_ctxs.values():
            out += c.read_launch_log()
        return out

    def _launch_pending(self, stream: int) -> None:
        """the optimizer launches held back by the last step, in the order the fused step would have run them"""
        A = self.arena
        for _, layer, args in self._pending:
            call("gct2_adam_apply", ctypes.addressof(args), A.gptr(layer + ".w"), A.numel(layer + ".w"), stream)
        self._pending = []

    def flush_deferred(self) -> None:
        """run whatever optimizer launches the last train step held back, on the current stream (no-op when nothing is pending).
        Called by everything that reads or overwrites parameters outside the train step itself."""
        cur = torch.cuda.current_stream(self.device)
        if self._pending:
            self._launch_pending(cur.cuda_stream)
            # the launches write p / m / v / shadow of the deferred layers and read their slabs; the caller may be on ANY stream
            # (predict / state_dict / the sampler on an evaluation stream): whatever uses the parameters or the slabs next - the next
            # forward pass, the next reverse pass, on whichever stream - waits for this event first
            if _lib._recording is None:                          # (inside a recorded step the plan's own records order everything)
                self._flush_event = torch.cuda.Event()
                self._flush_event.record(cur)
        if self._pending_event is not None:
            self._wait(cur, self._pending_event)
            self._pending_event = None
        self._pending_names = set()

    def _after_flush(self, stream: "torch.cuda.Stream") -> None:
        """orders `stream` behind the last flush_deferred() that ran on another stream (no-op otherwise)"""
        if self._flush_event is not None and _lib._recording is None:      # (train_step has waited before it records or replays)
            stream.wait_event(self._flush_event)

    def check_input_shape(self, H: int, W: int) -> None:
        n = self.topo.octaves
        if H % (2 ** n) or W % (2 ** n):
            raise ValueError(
                f"spatial size {H}x{W} is not divisible by 2**octaves = {2 ** n}: the channel concat of "
                "train.py:114-119 would see mismatched shapes (TensorFlow raises InvalidArgumentError here)")

    def buffers(self, B: int, H: int, W: int) -> _Buffers:
        key = (B, H, W)
        if key in self._bufs:
            self._bufs.move_to_end(key)
            return self._bufs[key]
        self.check_input_shape(H, W)
        while len(self._bufs) >= max(1, self.max_buffer_sets):
            _, old = self._bufs.popitem(last=False)             # least recently used; graphs captured on it go with it
            graphs = getattr(self, "_forward_graphs", None)
            if graphs:
                for gk in [k for k in graphs if k[0] == id(old)]:
                    del graphs[gk]
        t, n, dt = self.topo, self.topo.octaves, TORCH_DTYPE[self.dtype]
        b = _Buffers()
        b.B, b.H, b.W = B, H, W
        b.hw = [(H >> i, W >> i) for i in range(n + 1)]
        b.ld = [_round_up(t.fu(i) + t.cx(i), 8) for i in range(n)]
        z = lambda *shape, dtype=dt: torch.zeros(*shape, dtype=dtype, device=self.device)
        b.R = [z(B, b.hw[i][0], b.hw[i][1], b.ld[i]) for i in range(n)]
        # gradient buffers: same shapes, except level 0 - the image gets no gradient (train.py:224-236: x and eps are data), so dR_0
        # only holds UpShuffle_0's Fu_0 channels: 128-byte pixels at the reference width instead of R_0's 144-byte rows, which
        # straddle two 128-byte lines per pixel (r04; U0's input / weight gradients read it, the fused head writes it)
        b.ldd = [_round_up(t.fu(0), 8) if i == 0 else b.ld[i] for i in range(n)]
        b.dR = [z(B, b.hw[i][0], b.hw[i][1], b.ldd[i]) for i in range(n)]
        # packed copy of the network input (3 channels + a zero slot): DownShuffle_0 and its weight gradient gather 4x4
        # windows from it instead of striding through R_0's 144-byte rows
        b.img = z(B, H, W, 4)
        # ReLU bit planes of R_1 .. R_{n-1}, for tensors of >= relu_bits_min_bytes (32 MiB: the three big levels of config 3 / 5;
        # below that the mask reads are noise and the small layers' split-K launches need an extra launch to derive the plane -
        # measured at 3x64x64, batch 32: +9 us per step with planes on every level, -45 us at config 3)
        b.bits = [None] * n
        b.bits_valid = False
        for i in range(1, n):
            px = B * b.hw[i][0] * b.hw[i][1]
            if self.dtype != F32 and px * b.ld[i] * 2 >= self.relu_bits_min_bytes and b.ld[i] % 8 == 0 and t.fu(i) % 8 == 0:
                b.bits[i] = torch.zeros(px, b.ld[i] // 8, dtype=torch.uint8, device=self.device)
        b.Dlast = z(B, b.hw[n][0], b.hw[n][1], t.fd(n - 1))
        b.dDlast = z(B, b.hw[n][0], b.hw[n][1], t.fd(n - 1))
        b.pred = z(B, H, W, 3, dtype=torch.float32)
        b.dpred = z(B, H, W, 3, dtype=torch.float32)
        b.eps = z(B, H, W, 3, dtype=torch.float32)
        b.target = None                 # fp32 [B,H,W,3], allocated by the non-default objectives (train.py:238-252)
        b.t_int = z(B, dtype=torch.int32)
        b.loss = z(1, dtype=torch.float32)
        b.partials = z(1024, dtype=torch.float32)
        self._bufs[key] = b
        return b

    def _esize(self) -> int:
        return 4 if self.dtype == F32 else 2

    def _slice_ptr(self, buf: torch.Tensor, ch: int) -> int:
        return buf.data_ptr() + ch * self._esize()

    # ---- parameters ---------------------------------------------------------------------------
    def set_params(self, params: Dict[str, "np.ndarray | torch.Tensor"]) -> None:
        for name, val in params.items():
            tv = torch.as_tensor(np.asarray(val, dtype=np.float32) if not torch.is_tensor(val) else val)
            self.arena.param(name).copy_(tv.to(self.device, torch.float32))
        self.arena.refresh_shadow(self._stream())

    def get_params(self) -> Dict[str, np.ndarray]:
        return {k: self.arena.param(k).detach().cpu().numpy().copy() for k in self.arena.shapes}

    def get_grads(self) -> Dict[str, np.ndarray]:
        """the gradient arena of the LAST backward pass (every step overwrites it; nothing accumulates across steps).  Not
        available after a step whose fused optimizer consumed the weight-gradient partial sums in place (train_step with
        apply=True on one replica without loss scaling): run that step with apply=False, or set fuse_adam = False."""
        if not self._grads_in_arena:
            raise _lib.Gct2Error("the last step applied its weight gradients straight from the kernels' partial sums: the "
                                 "gradient arena does not hold them (use train_step(..., apply=False) or fuse_adam = False)")
        return {k: self.arena.grad(k).detach().cpu().numpy().copy() for k in self.arena.shapes}

    # ---- Trainer.call pieces --------------------------------------------------------------------
    def sample_noise(self, b: _Buffers) -> None:
        """t_int ~ U{1..steps}, eps ~ N(0,1)  (train.py:224-227) from the library's Philox streams."""
        s = self._stream()
        call("gct2_rng_uniform_int", self.rng_seed, 1, self.rng_offset_t, b.t_int.data_ptr(), b.B, 1, self.steps, s)
        call("gct2_rng_normal", self.rng_seed, 2, self.rng_offset_eps, b.eps.data_ptr(), b.eps.numel(), s)
        self.rng_offset_t += b.B
        self.rng_offset_eps += b.eps.numel()

    def sample_and_noise_into_r0(self, b: _Buffers, x: torch.Tensor, keep_eps: bool = False) -> None:
        """train.py:224-234 in two launches: t_int, then noising with eps drawn inside the kernel (same stream positions
        as sample_noise + noise_into_r0, bit-identical result, no eps round trip through HBM)."""
        s = self._stream()
        out, ldout, out2, ldout2 = self._noise_targets(b)
        call("gct2_rng_uniform_int", self.rng_seed, 1, Slot("off_t", self.rng_offset_t), b.t_int.data_ptr(), b.B, 1, self.steps, s)
        call("gct2_noise_image_rng", self.dtype, Slot("x", x.data_ptr()), b.t_int.data_ptr(), self.rng_seed, 2, Slot("off_eps", self.rng_offset_eps),
             b.eps.data_ptr() if keep_eps else None, out, ldout, out2, ldout2, b.B, b.H * b.W, 3, self.steps, s)
        self.rng_offset_t += b.B
        self.rng_offset_eps += b.eps.numel()

    def _noise_targets(self, b: _Buffers, unfused_head: bool = False) -> Tuple[int, int, Optional[int], int]:
        """where the noised image goes: always the packed copy `img` (DownShuffle_0 reads it, and so does the fused head
        through its x2 argument); the image slice of R_0 only when the unfused Dense kernels will read R_0 as one
        67-channel view (its 6-byte writes into 144-byte rows cost more than the rest of the noising).  unfused_head: the
        caller runs forward(head=True) afterwards (Trainer.call, no gradients), which always reads the 67-channel view."""
        if self.fused_head_ok() and not unfused_head:
            return b.img.data_ptr(), 4, None, 0
        return self._slice_ptr(b.R[0], self.topo.fu(0)), b.ld[0], b.img.data_ptr(), 4

    def noise_into_r0(self, b: _Buffers, x: torch.Tensor, unfused_head: bool = False) -> None:
        out, ldout, out2, ldout2 = self._noise_targets(b, unfused_head)
        call("gct2_noise_image", self.dtype, x.data_ptr(), b.t_int.data_ptr(), b.eps.data_ptr(), out, ldout, out2, ldout2,
             b.B, b.H * b.W, 3, self.steps, self._stream())

    def load_input_into_r0(self, b: _Buffers, noised: torch.Tensor) -> None:
        """Denoiser.call on an externally prepared image (sampler / inference path)."""
        t = self.topo
        b.R[0][..., t.fu(0):t.fu(0) + 3].copy_(noised.to(TORCH_DTYPE[self.dtype]))
        b.img[..., :3].copy_(noised.to(TORCH_DTYPE[self.dtype]))

    def forward(self, b: _Buffers, head: bool = True, stop_before_u0: bool = False, planes: bool = False, in_step: bool = False) -> torch.Tensor:
        """Denoiser.call (train.py:206-215): R_0's image slice must already hold the network input.
        head=False stops before Dense(3) (the train step runs the fused head kernel instead); stop_before_u0 also leaves
        UpShuffle_0 to the caller (u0_head_train: its forward carries the head in its epilogue); planes: also write the ReLU bit
        planes of the large levels (the train step: backward() then reads its masks from them)."""
        t, n, s, dt, A, cx = self.topo, self.topo.octaves, self._stream(), self.dtype, self.arena, self.ctx.handle
        planes = planes and self.relu_bits
        b.bits_valid = planes
        if not in_step:
            self.flush_deferred()                              # (a captured graph, the sampler, predict: no side-stream work in here)
        cur = torch.cuda.current_stream(self.device)
        self._after_flush(cur)
        window_at = min(self.defer_window_at, n - 1)            # deferred Adam starts once DownShuffle_{window_at} is enqueued

        def join_pending() -> None:                             # the deferred updates are done before their weights are read
            if self._pending_event is not None:
                self._wait(cur, self._pending_event)
                self._pending_event = None
                self._pending_names = set()

        def plane(level: int, ch: int) -> None:                 # one-shot: applies to the layer call that follows
            if planes and b.bits[level] is not None:
                self.ctx.set_relu_bits(b.bits[level].data_ptr() + ch // 8, b.ld[level] // 8)
        # two half-batch chains for the outer levels (experiment, see __init__): chain 0 on the current stream, chain 1 on the side stream
        L = min(int(self.split_forward_levels), n - 1) if (in_step and self.overlap and b.B % 2 == 0 and dt != F32) else 0
        es, Bh = self._esize(), b.B // 2
        side = self._side

        def split_ctx(tuning: int) -> "_lib.Context":
            # the tile the FULL batch would take, forced (a half batch falls below the automatic thresholds), never split-K: no scratch,
            # so both chains may share the context; one context per tuning word because setters are not part of a step plan
            c = self._split_ctxs.get(tuning)
            if c is None:
                c = self._split_ctxs[tuning] = _lib.Context()
                c.set_tuning(tuning)
            return c

        def halves():
            return ((0, cur), (1, side))
        if L:
            self._wait_stream(side, cur)                       # the noised image is there
        for i in range(n):                                      # DownShuffle_i  (train.py:184)
            H, W = b.hw[i]
            if i < n - 1:
                y, ldy = self._slice_ptr(b.R[i + 1], t.fu(i + 1)), b.ld[i + 1]
            else:
                y, ldy = b.Dlast.data_ptr(), t.fd(i)
            x, ldx = (b.img.data_ptr(), 4) if i == 0 else (self._slice_ptr(b.R[i], t.fu(i)), b.ld[i])
            if i < L:
                tiles256 = ((b.B * (H // 2) * (W // 2) + 255) // 256) * ((t.fd(i) + 127) // 128)
                c = split_ctx((5 if tiles256 >= 512 else 2) | (1 << 8))
                for h, st in halves():
                    if planes and b.bits[i + 1] is not None:
                        c.set_relu_bits(b.bits[i + 1].data_ptr() + t.fu(i + 1) // 8 + h * Bh * (H // 2) * (W // 2) * (b.ld[i + 1] // 8), b.ld[i + 1] // 8)
                    call("gct2_conv4s2_fwd", c.handle, dt, x + h * Bh * H * W * ldx * es, ldx, A.wptr(f"D{i}.w"), A.pptr(f"D{i}.b"),
                         y + h * Bh * (H // 2) * (W // 2) * ldy * es, ldy, Bh, H, W, t.cx(i), t.fd(i), 1, st.cuda_stream)
                if i == L - 1:
                    self._wait_stream(cur, side)
                continue
            if i < n - 1:
                plane(i + 1, t.fu(i + 1))
            call("gct2_conv4s2_fwd", cx, dt, x, ldx, A.wptr(f"D{i}.w"), A.pptr(f"D{i}.b"), y, ldy, b.B, H, W, t.cx(i), t.fd(i), 1, s)
            if i == window_at and self._pending:
                self._wait_stream(self._side, cur)
                self._launch_pending(self._side.cuda_stream)
                self._pending_event = self._mark(self._side)
        for i in reversed(range(n)):                            # UpShuffle_i    (train.py:188)
            Hi, Wi = b.hw[i + 1]
            if i < n - 1:
                x, ldx = b.R[i + 1].data_ptr(), b.ld[i + 1]
            else:
                x, ldx = b.Dlast.data_ptr(), t.fd(i)
            if f"U{i}" in self._pending_names or i == 0:
                join_pending()
            if i == 0 and L > 1:
                self._wait_stream(cur, side)                   # both chains of the levels above are done
            if i == 0 and stop_before_u0:
                return b.pred
            if 1 <= i < L:
                if i == L - 1:
                    self._wait_stream(side, cur)               # UpShuffle_L's output (and the deferred updates joined above)
                c = split_ctx(2 << 24)                         # the halo kernel wherever it is allowed, as at the full batch
                for h, st in halves():
                    if planes and b.bits[i] is not None:
                        c.set_relu_bits(b.bits[i].data_ptr() + h * Bh * 4 * Hi * Wi * (b.ld[i] // 8), b.ld[i] // 8)
                    call("gct2_convT4s2_fwd", c.handle, dt, x + h * Bh * Hi * Wi * ldx * es, ldx, A.wptr(f"U{i}.w"), A.pptr(f"U{i}.b"),
                         b.R[i].data_ptr() + h * Bh * 4 * Hi * Wi * b.ld[i] * es, b.ld[i], Bh, Hi, Wi, t.up_in(i), t.fu(i), 1, st.cuda_stream)
                continue
            if i >= 1:
                plane(i, 0)
            call("gct2_convT4s2_fwd", cx, dt, x, ldx, A.wptr(f"U{i}.w"), A.pptr(f"U{i}.b"), b.R[i].data_ptr(), b.ld[i],
                 b.B, Hi, Wi, t.up_in(i), t.fu(i), 1, s)
        join_pending()
        if not head:
            return b.pred
        M = b.B * b.H * b.W                                     # Dense(3)       (train.py:198-202)
        call("gct2_dense_fwd", dt, b.R[0].data_ptr(), b.ld[0], A.pptr("dense.w"), A.pptr("dense.b"), b.pred.data_ptr(),
             M, t.fu(0) + 3, 3, s)
        return b.pred

    def loss_and_dpred(self, b: _Buffers, target: torch.Tensor, target_is_x: bool = False) -> torch.Tensor:
        """fp32 MSE against the clean image (predict_x, train.py:243-244,262-272) and its gradient."""
        ls_ptr = self.ls_state.data_ptr() if self.ls_state is not None else None
        call("gct2_mse_fwd_bwd", b.pred.data_ptr(), Slot("x", target.data_ptr()) if target_is_x else target.data_ptr(), b.dpred.data_ptr(), b.loss.data_ptr(),
             b.partials.data_ptr(), b.pred.numel(), ls_ptr, self._stream())
        return b.loss

    # ---- objectives other than the default (train.py:238-252) ----------------------------------------------------------
    def default_objective(self) -> bool:
        return self.predict_x and not self.ordinary_differential_equation

    def objective_weighted(self) -> bool:
        """train.py:250-252: prediction and target both scaled by sqrt(1 - alpha_dash(t)) (epsilon branch only)."""
        return (not self.default_objective()) and (not self.ordinary_differential_equation) and self.prediction_weighting

    def objective_coefficients(self, b: _Buffers) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """per-image (a, c, w): target = a x + c eps, prediction weight w - B-element vectors from t_int."""
        t = b.t_int.to(torch.float32)
        ad = lambda u: 0.25 * (1.0 - u / (self.steps + 1)) ** 2          # alpha_dash, train.py:85-93
        one, zero = torch.ones_like(t), torch.zeros_like(t)
        if self.ordinary_differential_equation:                           # train.py:238-242
            a1 = ad(t - 1)
            return a1.sqrt().contiguous(), (1 - a1).sqrt().contiguous(), one
        if self.predict_x:                                                # train.py:243-244
            return one, zero, one
        s = (1 - ad(t)).sqrt()
        c = s if self.predict_scaled_epsilon else one                     # train.py:245-248
        if self.prediction_weighting:                                     # train.py:250-252
            return zero, (c * s).contiguous(), s.contiguous()
        return zero, c.contiguous(), one

    def make_target(self, b: _Buffers, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """the regression target of train.py:238-252 as an fp32 [B,H,W,3] tensor (needs b.eps) and the prediction weights."""
        if b.target is None:
            b.target = torch.zeros_like(b.eps)
        a, c, w = self.objective_coefficients(b)
        b._coef = (a, c, w)                                               # keep the vectors alive until the kernels ran
        call("gct2_mix_per_image", x.data_ptr(), b.eps.data_ptr(), a.data_ptr(), c.data_ptr(), b.target.data_ptr(), b.B,
             b.H * b.W * 3, self._stream())
        return b.target, w

    def weighted_loss_and_dpred(self, b: _Buffers, target: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        """prediction_weighting (train.py:250-252): loss = mean((w_b pred - target)^2); dpred = w_b * dloss/d(w_b pred)."""
        n = b.H * b.W * 3
        call("gct2_mix_per_image", b.pred.data_ptr(), None, w.data_ptr(), None, b.pred.data_ptr(), b.B, n, self._stream())
        loss = self.loss_and_dpred(b, target)
        call("gct2_mix_per_image", b.dpred.data_ptr(), None, w.data_ptr(), None, b.dpred.data_ptr(), b.B, n, self._stream())
        return loss

    def fused_head_ok(self) -> bool:
        """the matrix-core head with the split input (R_0's UpShuffle_0 channels + the packed image): the reference topology
        (Fu_0 = 64) in a 16-bit mode with a workspace; anything else runs dense_fwd + mse_fwd_bwd + dense_bwd."""
        return (self.use_fused_head and self.dtype != F32 and self.topo.fu(0) == 64 and self.workspace is not None
                and not self.objective_weighted())

    def head_train(self, b: _Buffers, target: torch.Tensor, target_is_x: bool = False) -> torch.Tensor:
        """Dense(3) + fp32 MSE + both of their gradients in one pass over R_0 (gct2_dense_head_train)."""
        t, A = self.topo, self.arena
        ls_ptr = self.ls_state.data_ptr() if self.ls_state is not None else None
        call("gct2_dense_head_train", self.ctx.handle, self.dtype, b.R[0].data_ptr(), b.ld[0], A.pptr("dense.w"), A.pptr("dense.b"),
             Slot("x", target.data_ptr()) if target_is_x else target.data_ptr(), b.pred.data_ptr() if self.keep_pred else None, b.dR[0].data_ptr(), b.ldd[0], A.gptr("dense.w"), A.gptr("dense.b"),
             b.loss.data_ptr(), b.partials.data_ptr(), b.B * b.H * b.W, t.fu(0) + 3, 3, t.fu(0), ls_ptr, A.gptr("U0.b"),
             b.img.data_ptr(), 4, 0, self._stream())
        return b.loss

    def fused_u0_head_ok(self, b: _Buffers) -> bool:
        """UpShuffle_0's forward with the head in its epilogue (gct2_convT4s2_fwd_head_train): the fused head's conditions plus
        UpShuffle_0's source grid tiling into 16 x 16 patches and one partial row per patch in the workspace."""
        if not (self.fuse_u0_head and self.fused_head_ok() and self.topo.octaves >= 1):
            return False
        Hs, Ws = b.hw[1]
        return Hs % 16 == 0 and Ws % 16 == 0 and self.workspace.numel() >= b.B * (Hs // 16) * (Ws // 16) * 288

    def u0_head_train(self, b: _Buffers, target: torch.Tensor, target_is_x: bool = False) -> torch.Tensor:
        """UpShuffle_0 forward + Dense(3) + fp32 MSE + both gradients in one launch: R_0 is never written, dR_0 receives the
        gradient w.r.t. UpShuffle_0's pre-activation (train.py:188, 198-202, 262-272)."""
        t, A, n = self.topo, self.arena, self.topo.octaves
        Hi, Wi = b.hw[1]
        x, ldx = (b.R[1].data_ptr(), b.ld[1]) if n > 1 else (b.Dlast.data_ptr(), t.fd(0))
        ls_ptr = self.ls_state.data_ptr() if self.ls_state is not None else None
        call("gct2_convT4s2_fwd_head_train", self.ctx.handle, self.dtype, x, ldx, A.wptr("U0.w"), A.pptr("U0.b"),
             A.pptr("dense.w"), A.pptr("dense.b"), Slot("x", target.data_ptr()) if target_is_x else target.data_ptr(),
             b.pred.data_ptr() if self.keep_pred else None, b.dR[0].data_ptr(), b.ldd[0],
             A.gptr("dense.w"), A.gptr("dense.b"), b.loss.data_ptr(), b.B, Hi, Wi, t.up_in(0), t.fu(0), t.fu(0) + 3, 3, ls_ptr,
             A.gptr("U0.b"), b.img.data_ptr(), 4, 0, self._stream())
        return b.loss

    def host_call(self, fn: Callable[[], None]) -> None:
        """a Python callable that must run at THIS point of the step (a collective of the data-parallel exchange): run now - or, while a
        step plan is being recorded, noted as the end of a plan segment; every replay calls it between that segment and the next"""
        P = _lib._recording
        if P is not None:
            P.cut(("__call__", fn))
        else:
            fn()

    def _ready(self, layer: str, stream: Optional["torch.cuda.Stream"] = None) -> None:
        """gradient-ready hook of `layer` (data-parallel wrappers), called with `stream` - the stream whose kernels produced the
        gradients - current.  While a step plan is being recorded a plain hook is NOT called: the plan is cut here and the replay calls
        it between two segments; a plan-aware hook (hook_plan_aware) runs now and is recorded like the rest of the step."""
        if self.grad_ready_hook is None:
            return
        P = _lib._recording
        if P is not None and not self.hook_plan_aware:
            P.cut((layer, stream))
        elif stream is None:
            self.grad_ready_hook(layer)
        else:
            with torch.cuda.stream(stream):
                self.grad_ready_hook(layer)

    def backward(self, b: _Buffers, head_done: bool = False, adam_inline: bool = False) -> None:
        """reverse pass (what tape.gradient does inside Keras fit, train.py:516); fills the g arena.
        head_done: dR_0 and the Dense gradients were already produced by head_train.

        Two streams: the dgrad chain (the only true dependency chain of the reverse pass) stays on the current stream;
        every weight gradient is enqueued on the side stream behind an event of the dgrad launch that produced its dz, so
        its work-groups fill the tails and the small bottleneck launches of the chain.  adam_inline (single replica, no loss
        scaling): each layer's Adam step is fused behind its weight-gradient call (gct2_adam_args) once the layer's dgrad -
        the last reader of its weights - is done.  The current stream joins the side stream before returning."""
        t, n, dt, A, cx = self.topo, self.topo.octaves, self.dtype, self.arena, self.ctx.handle
        want_queue = bool(self.defer_rowsums) and self.workspace is not None
        if want_queue != self._bias_queue_on:
            if want_queue and self._bias_queue is None:
                self._bias_queue = torch.empty(4 << 20, dtype=torch.float32, device=self.device)       # 16 MiB: ~6 MB of rows per pass at config 3
            self.ctx.set_bias_queue(self._bias_queue if want_queue else None)
            self._bias_queue_on = want_queue
        caller = torch.cuda.current_stream(self.device)
        if self.overlap and self.chain_priority and self._chain_stream is None:
            self._chain_stream = distinct_stream(self.device, "chain", self._caller0, priority=-1)
        main = self._chain_stream if (self.overlap and self.chain_priority) else caller
        if main is not caller:
            self._wait_stream(main, caller)
        self._after_flush(main)
        side = self._side if self.overlap else main
        s, sw = main.cuda_stream, side.cuda_stream
        M = b.B * b.H * b.W
        if not head_done:
            call("gct2_dense_bwd", dt, b.R[0].data_ptr(), b.ld[0], A.pptr("dense.w"), b.dpred.data_ptr(), b.dR[0].data_ptr(),
                 b.ldd[0], A.gptr("dense.w"), A.gptr("dense.b"), M, t.fu(0) + 3, 3, t.fu(0), 0, s)
        self._ready("dense", main)                              # hooks record their events on the stream the gradients come from

        def side_waits_main() -> None:                          # side stream: everything enqueued on main so far is visible
            if side is not main:
                self._wait_stream(side, main)

        # adam_inline: every layer's Keras-Adam step is fused behind its weight-gradient call (gct2_adam_args): the gradient of
        # the kernel is consumed from the launch's partial sums or from the arena without ever being zeroed.  The update writes
        # the layer's operand copy, so it must follow the layer's dgrad (the last reader): in that mode the dgrad is enqueued
        # first and the side stream waits for it - the weight gradient of layer L then runs beside the dgrad of layer L+1.
        # the optimizer steps of defer_layers are held back until the next forward pass (see __init__)
        deferred = set(self.defer_layers) if (adam_inline and self.defer_adam and side is not main and self.wgrad_workspace is not None) else set()
        if self._pending:                                       # (a reverse pass without a forward pass in front of it: scripts, tests)
            self.flush_deferred()

        alpha = self.adam_alpha() if adam_inline else 0.0

        def fused(layer: str):
            if not adam_inline:
                return None
            lo, hi = A.layer_ranges[layer]
            args = self._adam_args.get(layer)                   # one struct per layer for the engine's lifetime: stable addresses
            if args is None:
                args = self._adam_args[layer] = _lib.AdamArgs()
            args.p, args.m, args.v = A._p.data_ptr() + 4 * lo, A._m.data_ptr() + 4 * lo, A._v.data_ptr() + 4 * lo
            args.shadow = (A._shadow.data_ptr() + 2 * lo) if A._shadow is not None else None
            args.shadow_dtype, args.n = self.dtype, hi - lo
            args.alpha, args.beta1, args.beta2, args.eps, args.grad_mul = alpha, self.beta_1, self.beta_2, self.epsilon, 1.0
            args.defer = 1 if layer in deferred else 0
            if layer in deferred:
                self._pending.append(("layer", layer, args))
                self._pending_names.add(layer)
            used.append(layer)
            return ctypes.addressof(args)

        def wctx(layer: str) -> int:                            # a deferred layer's slabs must outlive the step: its own scratch
            if layer not in deferred:
                return cx
            return self._defer_ctx(layer).handle

        used: list = []                                         # layers whose optimizer step rides on their weight-gradient call
        self._fused_layers = used
        for i in range(n):                                      # UpShuffle_i backward, outermost first
            Hi, Wi = b.hw[i + 1]
            if i < n - 1:
                x, ldx, dx, lddx = b.R[i + 1].data_ptr(), b.ld[i + 1], b.dR[i + 1].data_ptr(), b.ld[i + 1]
            else:
                x, ldx, dx, lddx = b.Dlast.data_ptr(), t.fd(i), b.dDlast.data_ptr(), t.fd(i)
            dz, lddz = b.dR[i].data_ptr(), b.ldd[i]
            # bias gradients are column sums of pre-activation gradients: each dgrad launch produces the sums of the tensor it
            # writes (fused into its epilogue), so only U_0's bias needs the wgrad entry point's db when the head is unfused.
            # Nothing accumulates across steps: the first writer of a bias gradient overwrites it (db_accumulate bit clear),
            # the second one - DownShuffle_{i+1}'s dgrad adding the skip part of the concat - adds (bit set).
            db_u = A.gptr(f"U{i}.b") if (i == 0 and not head_done) else None
            if i < n - 1:       # dx = dR_{i+1}: channels [0, Fu_{i+1}) belong to U_{i+1}, the rest to D_i
                db, split, db2 = A.gptr(f"U{i + 1}.b"), t.fu(i + 1), A.gptr(f"D{i}.b")
            else:               # dx = gradient of D_{n-1}'s output
                db, split, db2 = A.gptr(f"D{i}.b"), t.fd(i), None

            def dgrad_u():
                if i < n - 1 and b.bits_valid and b.bits[i + 1] is not None:        # mask of dR_{i+1} = (R_{i+1} > 0): the plane
                    self.ctx.set_relu_bits(b.bits[i + 1].data_ptr(), b.ld[i + 1] // 8)
                call("gct2_convT4s2_dgrad", cx, dt, dz, lddz, A.wptr(f"U{i}.w"), x, ldx, dx, lddx, b.B, Hi, Wi, t.up_in(i),
                     t.fu(i), 0, db, split, db2, 0, s)

            if adam_inline:
                dgrad_u()
            side_waits_main()                                   # dz (and this layer's bias gradient) are complete
            call("gct2_convT4s2_wgrad", wctx(f"U{i}"), dt, x, ldx, dz, lddz, A.gptr(f"U{i}.w"), db_u, b.B, Hi, Wi, t.up_in(i), t.fu(i), 0,
                 fused(f"U{i}"), sw)
            self._ready(f"U{i}", side)
            if not adam_inline:
                dgrad_u()
        for i in reversed(range(n)):                            # DownShuffle_i backward, innermost first
            H, W = b.hw[i]
            if i < n - 1:
                dz, lddz = self._slice_ptr(b.dR[i + 1], t.fu(i + 1)), b.ld[i + 1]
            else:
                dz, lddz = b.dDlast.data_ptr(), t.fd(i)
            x, ldx = self._slice_ptr(b.R[i], t.fu(i)), b.ld[i]
            xw, ldxw = (b.img.data_ptr(), 4) if i == 0 else (x, ldx)

            def dgrad_d():
                if i > 0:                                       # the image itself needs no gradient
                    if b.bits_valid and b.bits[i] is not None:
                        self.ctx.set_relu_bits(b.bits[i].data_ptr() + t.fu(i) // 8, b.ld[i] // 8)
                    call("gct2_conv4s2_dgrad", cx, dt, dz, lddz, A.wptr(f"D{i}.w"), x, ldx, self._slice_ptr(b.dR[i], t.fu(i)),
                         b.ld[i], b.B, H, W, t.cx(i), t.fd(i), 1, A.gptr(f"D{i - 1}.b"), t.cx(i), None, 1, s)

            if adam_inline:
                dgrad_d()
            if i == 0 and self._bias_queue_on:                  # every input-gradient launch is enqueued: sum the queued bias rows
                call("gct2_bias_queue_flush", cx, s)
            if i == 0:
                # every input-gradient launch (the writers of the bias gradients) and the head are enqueued on the chain's stream: the
                # fp32 zone - Dense(3) and all biases - is complete.  Its optimizer step is one small launch right here (nothing reads
                # these parameters before the next forward pass); the data-parallel wrappers get their hook at the end instead.
                if adam_inline:
                    if not head_done and side is not main:      # UpShuffle_0's bias gradient came from its weight-gradient call (side stream)
                        self._wait_stream(main, side)
                    self._apply_adam(*A.layer_ranges["fp32"], stream=s)
            if i == 0 and adam_inline and side is not main and self.tail_on_chain and self.workspace is not None:
                call("gct2_conv4s2_wgrad", self.ctx_tail.handle, dt, xw, ldxw, dz, lddz, A.gptr("D0.w"), None, b.B, H, W, t.cx(0),
                     t.fd(0), 0, fused("D0"), s)
                self._ready("D0", main)
                continue
            side_waits_main()
            call("gct2_conv4s2_wgrad", cx, dt, xw, ldxw, dz, lddz, A.gptr(f"D{i}.w"), None, b.B, H, W, t.cx(i), t.fd(i), 0,
                 fused(f"D{i}"), sw)
            self._ready(f"D{i}", side)
            if not adam_inline:
                dgrad_d()
        if side is not main:
            self._wait_stream(main, side)
        self._ready("fp32", main)                               # last hook: both streams' gradient writers are in front of it
        if main is not caller:
            self._wait_stream(caller, main)

    # ---- optimizer (train.py:50-65,75) -----------------------------------------------------------
    def learning_rate(self, k: Optional[int] = None) -> float:
        """WarmUp.__call__ (train.py:57-65), float32 arithmetic like the reference."""
        k = self.iterations if k is None else k
        if k < self.warm_up:
            return float(np.float32(self.base_lr) * np.float32(k + 1) / np.float32(self.warm_up + 1))
        return float(np.float32(self.base_lr))

    def adam_alpha(self, k: Optional[int] = None) -> float:
        """lr_k * sqrt(1 - b2^t) / (1 - b1^t), t = k + 1, with the betas as the float32 hyper-parameters Keras holds them as
        [TF]; gct2_loss_scale_begin computes the same on the device when the step counter lives there."""
        k = self.iterations if k is None else k
        tt = k + 1
        b1, b2 = float(np.float32(self.beta_1)), float(np.float32(self.beta_2))
        return self.learning_rate(k) * math.sqrt(1.0 - b2 ** tt) / (1.0 - b1 ** tt)

    def apply_adam(self, lo: int = 0, hi: Optional[int] = None, grad_div: float = 1.0, stream: Optional[int] = None) -> None:
        """Keras Adam on arena range [lo, hi); does not advance `iterations` (see finish_step).
        grad_div > 1 folds the data-parallel mean (sum over ranks / world size) into the gradient read."""
        self.flush_deferred()
        self._apply_adam(lo, hi, grad_div, stream)

    def _apply_adam(self, lo: int = 0, hi: Optional[int] = None, grad_div: float = 1.0, stream: Optional[int] = None) -> None:
        A, s = self.arena, (self._stream() if stream is None else stream)
        hi = A.total if hi is None else hi
        # with loss scaling the kernel takes inv_scale / found_inf / alpha from the device-resident state (the step counter that
        # alpha depends on only advances on applied steps); otherwise alpha comes from the host's counter
        ls_ptr = self.ls_state.data_ptr() if self.ls_state is not None else None
        alpha = 0.0 if self.ls_state is not None else self.adam_alpha()
        shadow = None if A._shadow is None else A._shadow.data_ptr() + 2 * lo
        call("gct2_adam_keras_multi", A._p.data_ptr() + 4 * lo, A._m.data_ptr() + 4 * lo, A._v.data_ptr() + 4 * lo,
             A.g.data_ptr() + 4 * lo, shadow, self.dtype, hi - lo, Slot("alpha", alpha), self.beta_1, self.beta_2,
             self.epsilon, 1.0 / grad_div, ls_ptr, 0, s)

    def finish_step(self) -> None:
        if self.ls_state is not None:      # applied_steps (= optimizer.iterations) advances on the device, only if finite
            call("gct2_loss_scale_update", self.ls_state.data_ptr(), 2000, self._stream())
        else:
            self._iterations += 1

    def begin_step(self) -> None:
        if self.ls_state is not None:
            call("gct2_loss_scale_begin", self.ls_state.data_ptr(), float(self.base_lr), int(self.warm_up), float(self.beta_1),
                 float(self.beta_2), self._stream())

    def check_finite(self, lo: int = 0, hi: Optional[int] = None, stream: Optional[int] = None) -> None:
        if self.ls_state is not None:
            hi = self.arena.total if hi is None else hi
            call("gct2_scale_check_finite", self.arena.g.data_ptr() + 4 * lo, hi - lo, self.ls_state.data_ptr(),
                 self._stream() if stream is None else stream)

    # ---- the whole step ---------------------------------------------------------------------------
    def train_step(self, x: torch.Tensor, t_int: Optional[torch.Tensor] = None, eps: Optional[torch.Tensor] = None,
                   apply: bool = True) -> torch.Tensor:
        """x: fp32 [B,H,W,3] on the device.  Returns the fp32 loss as a 1-element device tensor
        (no host synchronisation).  t_int / eps may be injected for parity runs (SURVEY.md §8c RNG)."""
        if x.dim() != 4 or x.shape[-1] != 3:
            raise ValueError(f"expected an NHWC batch [B,H,W,3], got {tuple(x.shape)}")
        if x.dtype != torch.float32 or not x.is_contiguous() or x.device != self.device:
            x = x.to(self.device, torch.float32).contiguous()
        B, H, W, _ = x.shape
        b = self.buffers(B, H, W)
        inline = apply and self.fuse_adam and self.ls_state is None
        self.post_backward_ran = self.post_backward is not None and not apply
        cur = torch.cuda.current_stream(self.device)
        if self._flush_event is not None:                      # a flush on another stream (predict / state_dict there): the step waits once
            cur.wait_event(self._flush_event)
            self._flush_event = None
        if self.use_plan and t_int is None and eps is None and self.default_objective() and _lib._recording is None:
            return self._planned_step(b, x, apply, inline, cur)
        return self._step_body(b, x, t_int, eps, apply, inline)

    def _step_body(self, b: _Buffers, x: torch.Tensor, t_int, eps, apply: bool, inline: bool) -> torch.Tensor:
        """the C-ABI calls of one train step in order (run eagerly, or recorded into a step plan by _planned_step)"""
        if not inline:
            self.flush_deferred()                              # (a fused step schedules them inside its forward pass instead)
        self.begin_step()
        default_obj = self.default_objective()
        if t_int is None and eps is None:
            # the normal training path: eps never touches HBM (unless the target is built from it, train.py:238-252)
            self.sample_and_noise_into_r0(b, x, keep_eps=not default_obj)
        else:
            if t_int is None or eps is None:
                self.sample_noise(b)
            if t_int is not None:
                b.t_int.copy_(t_int.to(self.device, torch.int32))
            if eps is not None:
                b.eps.copy_(eps.to(self.device, torch.float32))
            self.noise_into_r0(b, x)
        target, w = (x, None) if default_obj else self.make_target(b, x)
        weighted = self.objective_weighted()
        fused = self.fused_head_ok()
        if fused and self.fused_u0_head_ok(b):
            self.forward(b, head=False, stop_before_u0=True, planes=True, in_step=True)
            loss = self.u0_head_train(b, target, default_obj)
        else:
            self.forward(b, head=not fused, planes=True, in_step=True)
            if weighted:
                loss = self.weighted_loss_and_dpred(b, target, w)
            else:
                loss = self.head_train(b, target, default_obj) if fused else self.loss_and_dpred(b, target, default_obj)
        # single GPU without loss scaling: Adam rides the side stream inside backward(); the loss-scaled step has to see
        # every gradient (finite check) before any update
        self.flush_deferred()                                  # (no-op after a forward pass that scheduled them; covers octaves < 4 corner cases)
        if _lib._recording is not None:
            # the per-layer gct2_adam_args structs are read when a call is MADE: the deferred launches of the previous step (enqueued by
            # the forward pass above) must see the previous step's alpha, the weight-gradient calls below this step's - the replay
            # refreshes the structs between the two segments
            _lib._recording.cut(("__alpha__", None))
        self.backward(b, head_done=fused, adam_inline=inline)
        self._grads_in_arena = not inline
        if self.post_backward is not None and not apply:
            self.post_backward()
        if apply:
            if not inline:
                self.check_finite()
                self.apply_adam()
            self.finish_step()
        return loss

    # ---- step plans ------------------------------------------------------------------------------------------------------------
    def _plan_key(self, b: _Buffers, apply: bool, inline: bool, cur: "torch.cuda.Stream") -> tuple:
        """everything the call list of a step depends on besides the per-step slots"""
        return (id(b), apply, inline, cur.cuda_stream, self.ctx.version, self.ctx_tail.version,
                tuple(sorted((k, c.version) for k, c in self._defer_ctxs.items())), tuple(l for _, l, _ in self._pending),
                id(self.grad_ready_hook), self.hook_plan_aware, id(self.post_backward), self.overlap, self.chain_priority, self.fuse_adam, self.defer_adam, tuple(self.defer_layers),
                self.defer_window_at, self.defer_rowsums, self.tail_on_chain, self.use_fused_head, self.fuse_u0_head, self.keep_pred, self.relu_bits,
                self.ls_state is not None, self.workspace is not None, self.wgrad_workspace is not None, self.steps, self.rng_seed,
                # baked into recorded arguments (gct2_adam_keras_multi, gct2_loss_scale_begin) or restored into the per-layer
                # gct2_adam_args by every replay: Trainer.compile() may legally rewrite them between steps
                self.dtype, float(self.beta_1), float(self.beta_2), float(self.epsilon), float(self.base_lr), int(self.warm_up),
                id(self.post_replay), int(self.split_forward_levels), self._side.cuda_stream,
                self._chain_stream.cuda_stream if self._chain_stream is not None else 0)

    def _planned_step(self, b: _Buffers, x: torch.Tensor, apply: bool, inline: bool, cur: "torch.cuda.Stream") -> torch.Tensor:
        key = self._plan_key(b, apply, inline, cur)
        sp = self._plans.get(key)
        if sp is None:
            seen = self._plan_seen[key] = self._plan_seen.get(key, 0) + 1
            if seen < self.plan_after:
                return self._step_body(b, x, None, None, apply, inline)
            if len(self._plans) >= 8:                          # bounded: the oldest plan goes (and with it its events)
                self._plans.pop(next(iter(self._plans)))
            # record WITHOUT executing (host state - RNG offsets, deferrals, counters - advances exactly as in the eager step), then
            # replay what was recorded: the recording step itself already runs through the plan
            off_t, off_eps, its = self.rng_offset_t, self.rng_offset_eps, self._iterations
            # recording fills the per-layer gct2_adam_args structs for THIS step, but the deferred launches of the previous step (which
            # the replay below enqueues first) read theirs when they are made: put those back before replaying
            held = {l: bytes(a) for _, l, a in self._pending}
            # recording runs the step body with nothing enqueued: if it (or a hook) raises, every piece of host state it advanced goes
            # back to where it was, the half-built plan is dropped, and the caller sees the exception with the engine as before the
            # call (offsets and counters stayed advanced, _pending named launches whose slabs were never produced)
            before = dict(pending=list(self._pending), pending_names=set(self._pending_names), pending_event=self._pending_event,
                          bits_valid=b.bits_valid, grads_in_arena=self._grads_in_arena, post_backward_ran=self.post_backward_ran,
                          structs={l: bytes(a) for l, a in self._adam_args.items()})
            sp = _StepPlan(b)
            sp.plan.begin(execute=False)
            try:
                sp.loss = self._step_body(b, x, None, None, apply, inline)
            except BaseException:
                sp.plan.end()
                self.rng_offset_t, self.rng_offset_eps, self._iterations = off_t, off_eps, its
                self._pending, self._pending_names = before["pending"], before["pending_names"]
                self._pending_event, b.bits_valid = before["pending_event"], before["bits_valid"]
                self._grads_in_arena, self.post_backward_ran = before["grads_in_arena"], before["post_backward_ran"]
                for l, raw in before["structs"].items():
                    ctypes.memmove(ctypes.addressof(self._adam_args[l]), raw, len(raw))
                self._plan_seen[key] = self.plan_after - 1     # (the next step of this shape tries again)
                raise
            sp.plan.end()
            sp.after = dict(pending=list(self._pending), pending_names=set(self._pending_names), bits_valid=b.bits_valid,
                            grads_in_arena=self._grads_in_arena, fused_layers=list(getattr(self, "_fused_layers", [])))
            # the per-layer gct2_adam_args structs are shared with eager steps (which may fill them differently: another deferral
            # set, another range): a replay puts the recorded inputs back first
            sp.adam_inputs = {l: {f: getattr(self._adam_args[l], f) for f in _ADAM_INPUT_FIELDS} for l in sp.after["fused_layers"]}
            sp.d_off_t, sp.d_off_eps, sp.d_its = self.rng_offset_t - off_t, self.rng_offset_eps - off_eps, self._iterations - its
            self.rng_offset_t, self.rng_offset_eps, self._iterations = off_t, off_eps, its      # (the replay below advances them)
            for l, raw in held.items():
                ctypes.memmove(ctypes.addressof(self._adam_args[l]), raw, len(raw))
            # the post-step state (deferrals pending for the NEXT forward pass) must be the pre-step state of a steady-state replay:
            # a plan recorded on a step that started without deferrals is used for exactly such steps (its key says so)
            self._plans[key] = sp
        return self._replay(sp, x, inline)

    def _replay(self, sp: "_StepPlan", x: torch.Tensor, inline: bool) -> torch.Tensor:
        P = sp.plan
        P.set("x", x.data_ptr())
        P.set("off_t", self.rng_offset_t)
        P.set("off_eps", self.rng_offset_eps)
        alpha = self.adam_alpha() if self.ls_state is None else 0.0
        P.set("alpha", alpha)
        sp.x_ref = x                                            # keeps the batch alive until the next step replaces it
        hook = self.grad_ready_hook
        for k in range(P.segments()):
            payload = P.run_segment(k)
            if payload is None:
                continue
            layer, stream = payload
            if layer == "__call__":                             # a collective of a recorded data-parallel hook (host_call)
                stream()
            elif layer == "__alpha__":                          # between the forward and the reverse pass (see _step_body): until here the
                for l, fields in sp.adam_inputs.items():        # structs of the deferred layers belonged to the PREVIOUS step's launches
                    a = self._adam_args[l]
                    for f, v in fields.items():
                        setattr(a, f, v)
                    a.alpha = alpha
            elif hook is not None:
                if stream is None:
                    hook(layer)
                else:
                    with torch.cuda.stream(stream):
                        hook(layer)
        a = sp.after
        self.rng_offset_t += sp.d_off_t
        self.rng_offset_eps += sp.d_off_eps
        self._iterations += sp.d_its
        self._pending, self._pending_names, self._pending_event = list(a["pending"]), set(a["pending_names"]), None
        sp.b.bits_valid, self._grads_in_arena = a["bits_valid"], a["grads_in_arena"]
        if self.post_replay is not None:
            self.post_replay()
        return sp.loss

    def predict(self, noised: torch.Tensor) -> torch.Tensor:
        B, H, W, _ = noised.shape
        b = self.buffers(B, H, W)
        self.load_input_into_r0(b, noised)
        return self.forward(b)

    # ---- state serialisation (the reference has none, train.py:499-503 only logs) ----------------
    def state_dict(self) -> Dict[str, torch.Tensor]:
        """everything a run needs to continue bit-identically: fp32 parameters and Adam slots (one arena each, layout =
        ParamArena.offsets), optimizer iterations, RNG stream positions, loss-scale state.  CPU tensors, safetensors-ready."""
        if getattr(self, "_masters_sharded", False):
            raise _lib.Gct2Error("the fp32 parameters and Adam slots of this engine are sharded over the data-parallel ranks "
                                 "(ShardedDataParallelStep): use its state_dict() / save_checkpoint() (they gather first), or call "
                                 "gather_master() on every rank")
        torch.cuda.synchronize(self.device)
        A = self.arena
        sd = {"arena.p": A.p.cpu(), "arena.m": A.m.cpu(), "arena.v": A.v.cpu(),
              "counters": torch.tensor([self.iterations, self.rng_seed, self.rng_offset_t, self.rng_offset_eps], dtype=torch.int64),
              "topology": torch.tensor([self.topo.pixel_size, self.topo.max_size, self.topo.octaves, A.total, A.LAYOUT], dtype=torch.int64)}
        if self.ls_state is not None:
            sd["loss_scale_state"] = self.ls_state.cpu()
        return sd

    def named_state_dict(self) -> Dict[str, torch.Tensor]:
        """the same state keyed BY PARAMETER NAME - "p/U0.w", "m/U0.w", "v/U0.w", ... in the tensors' own (Keras) shapes plus the
        counters - independent of how the arenas are laid out: the exchange format between arena layouts, engines and other
        frameworks.  CPU tensors, safetensors-ready; load_named_state_dict reads it back."""
        sd = self.state_dict()
        A, out = self.arena, {k: v for k, v in sd.items() if not k.startswith("arena.") and k != "topology"}
        out["topology"] = sd["topology"][:3].clone()
        for slot in ("p", "m", "v"):
            flat = sd["arena." + slot]
            for name, o in A.offsets.items():
                out[f"{slot}/{name}"] = flat[o:o + A.numel(name)].view(A.shapes[name]).clone()
        return out

    def load_named_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        A = self.arena
        if [int(v) for v in sd["topology"][:3]] != [self.topo.pixel_size, self.topo.max_size, self.topo.octaves]:
            raise ValueError(f"checkpoint topology {[int(v) for v in sd['topology'][:3]]} != engine")
        flat = {slot: torch.zeros(A.total, dtype=torch.float32) for slot in ("p", "m", "v")}
        for slot in flat:
            for name, o in A.offsets.items():
                t = sd[f"{slot}/{name}"]
                if tuple(t.shape) != tuple(A.shapes[name]):
                    raise ValueError(f"{slot}/{name}: shape {tuple(t.shape)} != {tuple(A.shapes[name])}")
                flat[slot][o:o + A.numel(name)] = t.reshape(-1).to(torch.float32)
        arena_sd = {k: v for k, v in sd.items() if "/" not in k}
        arena_sd.update({"arena." + slot: flat[slot] for slot in flat})
        arena_sd["topology"] = torch.tensor([self.topo.pixel_size, self.topo.max_size, self.topo.octaves, A.total, A.LAYOUT], dtype=torch.int64)
        self.load_state_dict(arena_sd)

    def _convert_legacy_layout(self, sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """a checkpoint of the r01-r03 layout ([kernel | bias] per layer, 4-entry topology record) re-laid into this arena"""
        A = self.arena
        old, old_total = ParamArena.legacy_offsets(self.topo)
        if int(sd["topology"][3]) != old_total:
            raise ValueError(f"legacy checkpoint holds {int(sd['topology'][3])} elements per arena, this topology's r01-r03 layout {old_total}")
        out = dict(sd)
        for slot in ("p", "m", "v"):
            src, dst = sd["arena." + slot], torch.zeros(A.total, dtype=torch.float32)
            for name, o in A.offsets.items():
                dst[o:o + A.numel(name)] = src[old[name]:old[name] + A.numel(name)]
            out["arena." + slot] = dst
        out["topology"] = torch.tensor([self.topo.pixel_size, self.topo.max_size, self.topo.octaves, A.total, A.LAYOUT], dtype=torch.int64)
        return out

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        A = self.arena
        want = [self.topo.pixel_size, self.topo.max_size, self.topo.octaves, A.total, A.LAYOUT]
        have = [int(v) for v in sd["topology"]]
        if len(have) == 4:             # r01-r03 checkpoints carry no layout tag: [kernel | bias] per layer - converted on the way in
            if have[:3] != want[:3]:
                raise ValueError(f"checkpoint topology {have[:3]} != engine {want[:3]}")
            sd = self._convert_legacy_layout(sd)
            have = [int(v) for v in sd["topology"]]
        if have != want:
            raise ValueError(f"checkpoint topology / layout {have} != engine {want}")
        if ("loss_scale_state" in sd) != (self.ls_state is not None):
            raise ValueError("checkpoint and engine disagree on dynamic loss scaling (mixed_precision, train.py:34)")
        for name in ("p", "m", "v"):
            getattr(A, name).copy_(sd["arena." + name].to(self.device, torch.float32))
        A.g.zero_()
        its, self.rng_seed, self.rng_offset_t, self.rng_offset_eps = (int(v) for v in sd["counters"])
        if self.ls_state is not None:
            self.ls_state.copy_(sd["loss_scale_state"].to(self.device))
        self.iterations = its
        A.refresh_shadow(self._stream())
        self._masters_sharded = False                          # the arenas are whole again (a sharded wrapper sets it at its next step)

    def save_checkpoint(self, path: str) -> None:
        from safetensors.torch import save_file
        save_file(self.state_dict(), path)

    def load_checkpoint(self, path: str) -> None:
        from safetensors.torch import load_file        # safetensors: loading executes nothing from the file
        self.load_state_dict(load_file(path))

    def loss_scale(self) -> Tuple[float, int]:
        if self.ls_state is None:
            return 1.0, 0
        raw = self.ls_state.cpu()
        return float(raw[:1].view(torch.float32)[0]), int(raw[2])
