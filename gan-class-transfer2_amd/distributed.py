"""Data-parallel replicas: one process per GPU, gradients summed with RCCL all-reduce over xGMI.

The reference is single-GPU (train.py:40) - this is the build-side addition of SURVEY.md §8(e).  The path shards by
batch only: every rank holds a full replica and `bs/rank` images; the ONE exchange step is the all-reduce of the
gradient arena, issued bucket by bucket in backward-completion order on a side stream while the rest of the
backward pass is still running; the 1/world_size of the mean is folded into Adam's gradient read
(gct2_adam_keras_multi's inv_scale), so replicas stay bit-identical.

The reducer only needs a flat gradient tensor and the contiguous [lo, hi) range of every layer, so it runs
unchanged on CPU tensors with the gloo backend (tests/test_distributed_cpu.py).

Two exchange schemes (SURVEY.md App. D prices them on 7 xGMI links x ~153 GB/s per GPU):
  DataParallelStep         all-reduce of every bucket, every rank runs the whole Adam step (2 (N-1)/N x 4 B per parameter on
                           the wire, 30-38 B per parameter of optimizer traffic on every GPU);
  ShardedDataParallelStep  reduce-scatter of every bucket, Adam on the rank's OWN shard only (optimizer state is sharded), then an
                           all-gather of the updated compute-dtype weights: (N-1)/N x (4 + 2) B per parameter on the wire for
                           bf16/fp16 and 1/N of the optimizer traffic per GPU.  Replicas end with bit-identical operand copies.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


class _Plumbing:
    """stream ordering and host calls of a data-parallel wrapper.  Through the engine when it has the hooks for it (UNetEngine: while
    a step plan is being recorded a stream wait becomes a plan record and a collective the end of a plan segment, so a replayed
    step runs nothing of the wrapper from the interpreter but its collectives); directly otherwise (the CPU stand-in of the tests,
    a stand-alone reducer)."""

    def __init__(self, engine=None):
        self.engine = engine

    def stream_waits(self, waiter, other, to_comm: bool = False) -> None:
        """to_comm: `waiter` is the communication stream - what it waits for is read by a collective and leaves the device, so a
        recorded event releases at system scope (the step's own stream-to-stream waits keep device scope)"""
        f = getattr(self.engine, "_wait_stream", None)
        if f is not None:
            f(waiter, other, to_comm)
        else:
            ev = torch.cuda.Event()
            ev.record(other)
            waiter.wait_event(ev)

    def host_call(self, fn) -> None:
        f = getattr(self.engine, "host_call", None)
        if f is not None:
            f(fn)
        else:
            fn()


def _comm_stream(engine, device):
    """the communication stream of a wrapper: the engine's registered "comm" stream - one that shares a hardware queue neither with
    the caller's stream nor with the weight-gradient side stream, and the same one for every wrapper of this caller (engine.py
    distinct_stream) - or a plain new stream for a stand-alone reducer"""
    f = getattr(engine, "stream_for", None)
    return f("comm") if f is not None else torch.cuda.Stream(device=device)


def _warm_up_collective(group, stream, device) -> None:
    """ONE tiny collective on the communication stream at construction: the collective library creates its communicator and takes its
    internal streams at a DEFINED point of the process (its first call), not in the middle of the first timed step"""
    if not dist.is_initialized():
        return
    with torch.cuda.stream(stream):
        dist.all_reduce(torch.zeros(64, dtype=torch.float32, device=device), op=dist.ReduceOp.SUM, group=group)
    stream.synchronize()


class BucketedAllReducer:
    """sums `flat[lo:hi]` across ranks, one collective per bucket of consecutive layers.

    layer_order / layer_ranges: layers in the order their gradients complete during backward, each a
    contiguous range of `flat` (ParamArena orders the arena that way).  xGMI is point-to-point, so few large
    collectives beat many small ones: layers are merged until a bucket holds >= bucket_elems elements."""

    def __init__(self, flat: torch.Tensor, layer_order: Sequence[str], layer_ranges: Dict[str, Tuple[int, int]],
                 bucket_elems: int = 4 << 20, group=None, force_exchange: bool = False, engine=None):
        self.flat, self.group = flat, group
        self.pl = _Plumbing(engine)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # force_exchange: issue the collectives even at world size 1 (tests drive the stream plumbing on one GPU)
        self.exchange = self.world > 1 or (force_exchange and dist.is_initialized())
        self.buckets: List[Tuple[int, int]] = []
        self.flush_after: Dict[str, int] = {}          # layer name -> bucket index closed by that layer
        lo = None
        for k, name in enumerate(layer_order):
            l, h = layer_ranges[name]
            lo = l if lo is None else lo
            if h - lo >= bucket_elems or k == len(layer_order) - 1:
                self.flush_after[name] = len(self.buckets)
                self.buckets.append((lo, h))
                lo = None
        self.on_cuda = flat.is_cuda
        self.comm_stream = _comm_stream(engine, flat.device) if self.on_cuda else None
        if self.on_cuda and self.exchange:
            _warm_up_collective(group, self.comm_stream, flat.device)
        self.works: List[Optional[object]] = [None] * len(self.buckets)
        self.launched = 0

    def begin(self) -> None:
        self.works = [None] * len(self.buckets)
        self.launched = 0

    def grad_ready(self, layer: str) -> None:
        """call when every kernel writing `layer`'s gradients has been enqueued on the current stream."""
        if not self.exchange or layer not in self.flush_after:
            return
        idx = self.flush_after[layer]
        lo, hi = self.buckets[idx]
        view = self.flat[lo:hi]
        if self.on_cuda:
            # the communication stream waits for the gradients, then carries the collective: a stream-synchronous call (the
            # communication stream is blocked until the sum is there, the host is not), so whatever is enqueued on that stream behind
            # it - the bucket's optimizer step - is ordered without a work handle
            self.pl.stream_waits(self.comm_stream, torch.cuda.current_stream(self.flat.device), to_comm=True)
            self.pl.host_call(lambda idx=idx: self._issue(idx))
        else:
            self.works[idx] = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.launched += 1

    def _issue(self, idx: int) -> None:
        """(runs at every step, recorded or replayed: `launched` counts the collectives really issued)"""
        lo, hi = self.buckets[idx]
        with torch.cuda.stream(self.comm_stream):
            dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
        self.launched += 1

    def wait_bucket(self, idx: int) -> Tuple[int, int]:
        """block the CURRENT stream (not the host, on a GPU) until bucket idx holds the global sum."""
        if self.on_cuda:
            if self.exchange:
                cur = torch.cuda.current_stream(self.flat.device)
                if cur != self.comm_stream:
                    self.pl.stream_waits(cur, self.comm_stream)
            return self.buckets[idx]
        w = self.works[idx]
        if w is not None:
            w.wait()
        return self.buckets[idx]


def _one_stream_less(engine, exchange: bool) -> None:
    """The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES = 4 hardware queues, and streams that share a queue
    block each other (a stream waiting for an event stalls its queue-mate).  With engine.chain_priority the single-GPU step uses three
    (caller, the high-priority input-gradient chain, the weight-gradient side stream); an exchange adds the communication stream and the
    collective library's own, which pushed the chain onto a shared queue: measured on one rank with the exchange forced
    (scripts/bench_dp_overhead.py) 3.99 ms (all-reduce) / 4.24 ms (sharded) per step against 2.76 ms without exchange.  With the
    chain back on the caller's stream: 2.80 / 3.00 ms (the latter with the whole optimizer on one rank; it is 1/N per rank)."""
    if exchange and hasattr(engine, "chain_priority"):
        engine.chain_priority = False


class DataParallelStep:
    """drives UNetEngine.train_step on every rank with overlapped gradient all-reduce.

    Without loss scaling the optimizer is overlapped too: when bucket k's all-reduce has been enqueued, the Adam update of
    bucket k-1 is enqueued behind its (already running) all-reduce on the communication stream, so it executes while the
    backward pass is still producing later buckets.  That is safe because a bucket closes only after the side stream of
    UNetEngine.backward has waited for the dgrad launches of every earlier layer - the last readers of their weights."""

    def __init__(self, engine, bucket_elems: int = 4 << 20, group=None, force_exchange: bool = False):
        self.engine = engine
        A = engine.arena
        self.reducer = BucketedAllReducer(A.g, A.ready_order(), A.layer_ranges, bucket_elems, group, force_exchange, engine=engine)
        self.pl = self.reducer.pl
        engine.grad_ready_hook = self._grad_ready
        engine.hook_plan_aware = True          # the hook below does its plumbing through the engine: recorded with the step
        engine.post_backward = self._tail
        engine.post_replay = self._after_replay
        self.world = self.reducer.world
        self._adam_next = 0            # first bucket whose update has not been enqueued in this step
        _one_stream_less(engine, self.reducer.exchange)

    def broadcast_parameters(self, src: int = 0) -> None:
        if self.world > 1:
            dist.broadcast(self.engine.arena.p, src)
            self.engine.arena.refresh_shadow(self.engine._stream())

    def _adam_bucket(self, idx: int) -> None:
        red, eng = self.reducer, self.engine
        lo, hi = red.buckets[idx]
        # on the communication stream, behind the bucket's (stream-synchronous) all-reduce; mean over ranks folded into the gradient read
        eng.apply_adam(lo, hi, grad_div=float(self.world), stream=red.comm_stream.cuda_stream)
        self._adam_next = idx + 1

    def _grad_ready(self, layer: str) -> None:
        red = self.reducer
        red.grad_ready(layer)
        idx = red.flush_after.get(layer)
        if idx is None or not red.exchange or not red.on_cuda or self.engine.ls_state is not None:
            return
        while self._adam_next < idx:                            # buckets before the one just closed
            self._adam_bucket(self._adam_next)

    def train_step(self, x, t_int=None, eps=None):
        eng, red = self.engine, self.reducer
        if not red.exchange:       # nothing to exchange: the engine's own step (Adam inline on its side stream)
            return eng.train_step(x, t_int, eps, apply=True)
        red.begin()
        self._adam_next = 0
        self._tail_done = False
        loss = eng.train_step(x, t_int, eps, apply=False)      # backward fires _grad_ready per layer, then post_backward = _tail
        if not (self._tail_done or getattr(eng, "post_backward_ran", False)):
            self._tail()                                         # (an engine without the post_backward hook: the tests' CPU stand-in)
        return loss

    def _after_replay(self) -> None:
        """UNetEngine.post_replay: the hooks and _tail only run while a step is RECORDED; a replayed step ran the same launches, so
        the wrapper's counters go where a full step leaves them (reducer.launched is counted by _issue itself)"""
        self._adam_next = len(self.reducer.buckets) if self.engine.ls_state is None and self.reducer.on_cuda else 0
        self._tail_done = True

    def _tail(self) -> None:
        """what follows the reverse pass: the updates of the buckets still open, the caller's stream joins the communication stream,
        the step counter - part of the step body (UNetEngine.post_backward), so a step plan records it"""
        eng, red = self.engine, self.reducer
        self._tail_done = True
        assert not red.exchange or not red.on_cuda or _recording_plan() or red.launched == len(red.buckets), (red.launched, len(red.buckets))
        if eng.ls_state is not None:
            # fp16 + dynamic loss scale (train.py:82-83): an inf/nan on ANY rank survives the SUM all-reduce, so the
            # finite check of the reduced arena gives every rank the same skip decision without a second collective
            for idx in range(len(red.buckets)):
                red.wait_bucket(idx)
            eng.check_finite()
            eng.apply_adam(grad_div=float(self.world))
        elif red.on_cuda:
            while self._adam_next < len(red.buckets):
                self._adam_bucket(self._adam_next)
            self.pl.stream_waits(torch.cuda.current_stream(eng.device), red.comm_stream)
        else:
            for idx in range(len(red.buckets)):
                lo, hi = red.wait_bucket(idx)
                eng.apply_adam(lo, hi, grad_div=float(self.world))
        eng.finish_step()


class ShardedDataParallelStep:
    """reduce-scatter -> Adam on the own shard -> all-gather of the updated weights (module docstring).

    Buckets are fixed-size pieces of the gradient arena in backward-completion order (multiples of world x 64 elements, so every
    rank's shard of a bucket is a 16-byte aligned equal part; a bucket is independent of layer boundaries).  A bucket's
    reduce-scatter is enqueued on the communication stream as soon as the layer holding its last element is ready.  Its Adam
    step and the all-gather overwrite weights, so they wait until a LATER layer is ready - UNetEngine.backward enqueues a
    layer's input-gradient launch (the last reader of its weights) after the layer's ready hook and makes the side stream wait
    for it before the next layer's hook - or until the end of the reverse pass.

    The fp32 master parameters and Adam slots are only maintained for the rank's own shards (`gather_master()` assembles the full
    arenas on every rank: checkpoints, tests).  With loss scaling every rank checks its shards and the found_inf flags are
    combined with one 4-byte MAX all-reduce.

    The LAST bucket is REPLICATED (r04): it holds the last convolution layers of the reverse pass and the arena's fp32 zone - the
    Dense(3) layer and every bias, which the kernels read in fp32 from the master arena - and is all-reduced and updated on every
    rank.  It closes only when the reverse pass is over, so it is kept small (0.54 M of the 41.7 M parameters at the reference
    topology) and its exchange is ONE collective; r03 exchanged the fp32-read parameters with a second collective and three indexing
    kernels behind the last bucket's reduce-scatter / all-gather."""

    def __init__(self, engine, bucket_elems: int = 4 << 20, group=None, force_exchange: bool = False, tail_layers: int = 3):
        self.engine, self.group = engine, group
        A = engine.arena
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.exchange = self.world > 1 or (force_exchange and dist.is_initialized())
        unit = self.world * 64
        if A.total % unit:
            raise ValueError(f"arena of {A.total} elements is not a multiple of world x 64 = {unit}")
        bsz = max(unit, (bucket_elems + unit - 1) // unit * unit)
        order = A.ready_order()                 # convolution layers in backward completion order, then the fp32 zone
        # The LAST bucket closes only when the reverse pass is over (DownShuffle_0's weight gradient, the fp32 zone), so its exchange
        # is always exposed: keep it small.  It starts at the first of the `tail_layers` last ranges (DownShuffle_1, DownShuffle_0 and
        # the fp32 zone of the reference topology: 0.54 M of the 41.7 M parameters), rounded down to a shard unit;
        # every bucket before it is a fixed-size piece that closes with an earlier layer.
        # tail_layers is clamped into [1, len(order) - 1] and the tail never starts behind the fp32 zone (0, too large or
        # a tail that rounds to nothing used to leave fixed-size pieces of which the last was marked replicated, with the zone possibly
        # straddling two of them), so the zone - which every rank reads in fp32 and nobody all-gathers - always sits in ONE replicated bucket
        tail_layers = min(max(int(tail_layers), 1), len(order) - 1)
        tail = min(A.layer_ranges[order[-tail_layers]][0], A.layer_ranges["fp32"][0]) // unit * unit
        if tail <= 0:
            tail = 0                                           # (a tiny arena: one replicated bucket - a plain all-reduce step)
        self.buckets: List[Tuple[int, int]] = [(lo, min(lo + bsz, tail)) for lo in range(0, tail, bsz)]
        self.buckets.append((tail, A.total))
        self.layer_index = {name: i for i, name in enumerate(order)}
        ends = [A.layer_ranges[name][1] for name in order]
        # last_layer[k]: index of the layer that holds the last element of bucket k (the bucket is complete when it is ready)
        self.last_layer = [next(i for i, e in enumerate(ends) if e >= hi) for _, hi in self.buckets]
        self.on_cuda = A.g.is_cuda
        self.comm_stream = _comm_stream(engine, A.g.device) if self.on_cuda else None
        if self.on_cuda and self.exchange:
            _warm_up_collective(group, self.comm_stream, A.g.device)
        self.pl = _Plumbing(engine)
        engine.grad_ready_hook = self._grad_ready
        engine.hook_plan_aware = True          # the hook does its plumbing through the engine: recorded with the step
        engine.post_backward = self._tail
        engine.post_replay = self._after_replay
        _one_stream_less(engine, self.exchange)
        self.events: List[Tuple[int, object, object]] = []      # (bucket, start, end) of the collectives when timing is on
        self.time_collectives = False
        # Parameters the kernels read in FP32 straight from the master arena (every bias, the Dense(3) kernel and bias): the sharded
        # buckets only all-gather the compute-dtype shadow, so these must live in the replicated last bucket (ParamArena puts them
        # in one zone at the end of the arena for exactly this)
        if self.buckets[-1][0] > A.layer_ranges["fp32"][0]:
            raise ValueError("internal: the replicated last bucket does not contain the arena's fp32 zone")
        engine._masters_sharded = False     # True after a sharded step, until gather_master(): UNetEngine.state_dict refuses
        self._begin()

    # ---- helpers ----------------------------------------------------------------------------------------------------------
    def _begin(self) -> None:
        self.next_rs = 0            # first bucket whose reduce-scatter has not been enqueued
        self.next_opt = 0           # first bucket whose Adam + all-gather has not been enqueued
        self.launched = 0

    def replicated(self, k: int) -> bool:
        return k == len(self.buckets) - 1

    def shard(self, k: int) -> Tuple[int, int]:
        lo, hi = self.buckets[k]
        if self.replicated(k):                  # every rank owns (and updates) the whole last bucket
            return lo, hi
        n = (hi - lo) // self.world
        return lo + self.rank * n, lo + (self.rank + 1) * n

    def _on_comm(self):
        return torch.cuda.stream(self.comm_stream) if self.on_cuda else _NullCtx()

    def _comm_waits_current(self) -> None:
        if self.on_cuda:
            self.pl.stream_waits(self.comm_stream, torch.cuda.current_stream(self.engine.device), to_comm=True)

    def _timed(self, k: int, fn) -> None:
        if self.time_collectives and self.on_cuda:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(self.comm_stream)
            fn()
            e.record(self.comm_stream)
            self.events.append((k, s, e))
        else:
            fn()

    def _reduce_scatter(self, k: int) -> None:
        self.pl.host_call(lambda k=k: self._issue_reduce(k))

    def _issue_reduce(self, k: int) -> None:
        """the bucket's gradient exchange on the communication stream (stream-synchronous: what is enqueued there behind it is ordered).
        Runs at every step, recorded or replayed: `launched` counts the exchanges really issued."""
        self.launched += 1
        g = self.engine.arena.g
        lo, hi = self.buckets[k]
        slo, shi = self.shard(k)
        with self._on_comm():
            if self.replicated(k):
                self._timed(k, lambda: dist.all_reduce(g[lo:hi], op=dist.ReduceOp.SUM, group=self.group))
            else:
                self._timed(k, lambda: dist.reduce_scatter_tensor(g[slo:shi], g[lo:hi], op=dist.ReduceOp.SUM, group=self.group))

    def _optimize_and_gather(self, k: int) -> None:
        eng = self.engine
        slo, shi = self.shard(k)
        stream = self.comm_stream.cuda_stream if self.on_cuda else None
        eng.apply_adam(slo, shi, grad_div=float(self.world), stream=stream)       # on the communication stream, behind the exchange
        if self.replicated(k):                  # updated identically everywhere: nothing to gather
            return
        self.pl.host_call(lambda k=k: self._issue_gather(k))

    def _issue_gather(self, k: int) -> None:
        A = self.engine.arena
        lo, hi = self.buckets[k]
        slo, shi = self.shard(k)
        w = A._shadow if A._shadow is not None else A._p
        with self._on_comm():
            self._timed(k, lambda: dist.all_gather_into_tensor(w[lo:hi], w[slo:shi], group=self.group))

    def _grad_ready(self, layer: str) -> None:
        if not self.exchange:
            return
        if layer not in self.layer_index:       # ("dense": its gradients live in the fp32 zone, which has its own hook)
            return
        q = self.layer_index[layer]
        rs_due = self.next_rs < len(self.buckets) and self.last_layer[self.next_rs] <= q
        opt_due = self.engine.ls_state is None and self.next_opt < len(self.buckets) and self.last_layer[self.next_opt] < q
        if rs_due or opt_due:
            self._comm_waits_current()          # the hook runs in the stream context that produced the layer's gradients
        while self.next_rs < len(self.buckets) and self.last_layer[self.next_rs] <= q:
            self._reduce_scatter(self.next_rs)
            self.next_rs += 1
        if self.engine.ls_state is None:
            # buckets that end in an EARLIER layer: every reader of their weights has been enqueued and waited for by now
            while self.next_opt < self.next_rs and self.last_layer[self.next_opt] < q:
                self._optimize_and_gather(self.next_opt)
                self.next_opt += 1

    # ---- public -----------------------------------------------------------------------------------------------------------
    def broadcast_parameters(self, src: int = 0) -> None:
        if self.world > 1:
            A = self.engine.arena
            for t in (A.p, A.m, A.v):
                dist.broadcast(t, src, group=self.group)
            A.refresh_shadow(self.engine._stream())

    def train_step(self, x, t_int=None, eps=None):
        eng = self.engine
        if not self.exchange:
            return eng.train_step(x, t_int, eps, apply=True)
        self._begin()
        self._tail_done = False
        loss = eng.train_step(x, t_int, eps, apply=False)       # backward fires _grad_ready per layer, then post_backward = _tail
        if not (self._tail_done or getattr(eng, "post_backward_ran", False)):
            self._tail()                                         # (an engine without the post_backward hook: the tests' CPU stand-in)
        eng._masters_sharded = True
        return loss

    def _after_replay(self) -> None:
        """UNetEngine.post_replay: the hooks and _tail only run while a step is RECORDED; a replayed step issued the same exchanges
        (counted by _issue_reduce), so the bucket cursors go where a full step leaves them"""
        self.next_rs = self.next_opt = len(self.buckets)
        self._tail_done = True
        assert self.launched == len(self.buckets), (self.launched, len(self.buckets))

    def _tail(self) -> None:
        """what follows the reverse pass - part of the step body (UNetEngine.post_backward), so a step plan records it"""
        eng = self.engine
        self._tail_done = True
        assert self.next_rs == len(self.buckets)
        self._comm_waits_current()                               # every input-gradient launch has been enqueued by now
        if eng.ls_state is not None:
            # fp16 + dynamic loss scale (train.py:82-83): each rank checks the shards it owns; one 4-byte MAX all-reduce makes
            # the skip decision global before any update
            stream = self.comm_stream.cuda_stream if self.on_cuda else None
            for k in range(len(self.buckets)):
                eng.check_finite(*self.shard(k), stream=stream)
            self.pl.host_call(self._issue_found_inf)
        while self.next_opt < len(self.buckets):
            self._optimize_and_gather(self.next_opt)
            self.next_opt += 1
        if self.on_cuda:
            self.pl.stream_waits(torch.cuda.current_stream(eng.device), self.comm_stream)
        eng.finish_step()

    def _issue_found_inf(self) -> None:
        with self._on_comm():
            dist.all_reduce(self.engine.ls_state[3:4], op=dist.ReduceOp.MAX, group=self.group)

    def gather_master(self) -> None:
        """assemble the full fp32 parameter and Adam-slot arenas on every rank from the per-rank shards (a collective: every rank
        calls it)."""
        if not self.exchange:
            return
        A = self.engine.arena
        for t in (A.p, A.m, A.v):
            for k, (lo, hi) in enumerate(self.buckets):
                if self.replicated(k):
                    continue
                slo, shi = self.shard(k)
                dist.all_gather_into_tensor(t[lo:hi], t[slo:shi].clone(), group=self.group)
        self.engine._masters_sharded = False

    def state_dict(self):
        """UNetEngine.state_dict of the WHOLE model: the sharded masters and Adam slots are gathered first (a collective: every
        rank calls it; each gets the same dictionary).  The engine's own state_dict refuses while the masters are sharded - a
        checkpoint written from one rank's arenas would resume from stale parameters and zero Adam slots outside its shards."""
        self.gather_master()
        return self.engine.state_dict()

    def save_checkpoint(self, path: str) -> None:
        """every rank calls it (gather), rank 0 writes."""
        sd = self.state_dict()
        if self.rank == 0:
            from safetensors.torch import save_file
            save_file(sd, path)

    def collective_times_ms(self) -> List[Tuple[int, float]]:
        return [(k, s.elapsed_time(e)) for k, s, e in self.events]


def _recording_plan() -> bool:
    """True while a step plan is being recorded (the collectives of that step are issued by the replay that follows)"""
    try:
        from . import _lib
    except ImportError:            # (the CPU tests import this module on its own)
        return False
    return _lib._recording is not None


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
