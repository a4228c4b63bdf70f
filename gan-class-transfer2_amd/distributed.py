"""Data-parallel replicas: one process per GPU, gradients summed with RCCL all-reduce over xGMI.

The reference is single-GPU (train.py:40) - this is the build-side addition of SURVEY.md §8(e).  The path shards by
batch only: every rank holds a full replica and `bs/rank` images; the ONE exchange step is the all-reduce of the
gradient arena, issued bucket by bucket in backward-completion order on a side stream while the rest of the
backward pass is still running; the 1/world_size of the mean is folded into Adam's gradient read
(gct2_adam_keras_multi's inv_scale), so replicas stay bit-identical.

The reducer only needs a flat gradient tensor and the contiguous [lo, hi) range of every layer, so it runs
unchanged on CPU tensors with the gloo backend (tests/test_distributed_cpu.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


class BucketedAllReducer:
    """sums `flat[lo:hi]` across ranks, one collective per bucket of consecutive layers.

    layer_order / layer_ranges: layers in the order their gradients complete during backward, each a
    contiguous range of `flat` (ParamArena orders the arena that way).  xGMI is point-to-point, so few large
    collectives beat many small ones: layers are merged until a bucket holds >= bucket_elems elements."""

    def __init__(self, flat: torch.Tensor, layer_order: Sequence[str], layer_ranges: Dict[str, Tuple[int, int]],
                 bucket_elems: int = 4 << 20, group=None, force_exchange: bool = False):
        self.flat, self.group = flat, group
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
        self.comm_stream = torch.cuda.Stream(device=flat.device) if self.on_cuda else None
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
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.flat.device))
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                self.works[idx] = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            self.works[idx] = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.launched += 1

    def wait_bucket(self, idx: int) -> Tuple[int, int]:
        """block the CURRENT stream (not the host, on a GPU) until bucket idx holds the global sum."""
        w = self.works[idx]
        if w is not None:
            w.wait()
            if self.on_cuda:
                torch.cuda.current_stream(self.flat.device).wait_stream(self.comm_stream)
        return self.buckets[idx]


class DataParallelStep:
    """drives UNetEngine.train_step on every rank with overlapped gradient all-reduce.

    Without loss scaling the optimizer is overlapped too: when bucket k's all-reduce has been enqueued, the Adam update of
    bucket k-1 is enqueued behind its (already running) all-reduce on the communication stream, so it executes while the
    backward pass is still producing later buckets.  That is safe because a bucket closes only after the side stream of
    UNetEngine.backward has waited for the dgrad launches of every earlier layer - the last readers of their weights."""

    def __init__(self, engine, bucket_elems: int = 4 << 20, group=None, force_exchange: bool = False):
        self.engine = engine
        A = engine.arena
        self.reducer = BucketedAllReducer(A.g, engine.topo.layer_order(), A.layer_ranges, bucket_elems, group, force_exchange)
        engine.grad_ready_hook = self._grad_ready
        self.world = self.reducer.world
        self._adam_next = 0            # first bucket whose update has not been enqueued in this step

    def broadcast_parameters(self, src: int = 0) -> None:
        if self.world > 1:
            dist.broadcast(self.engine.arena.p, src)
            self.engine.arena.refresh_shadow(self.engine._stream())

    def _adam_bucket(self, idx: int) -> None:
        red, eng = self.reducer, self.engine
        with torch.cuda.stream(red.comm_stream):
            lo, hi = red.wait_bucket(idx)                       # the comm stream waits for the collective, not the host
            eng.apply_adam(lo, hi, grad_div=float(self.world))  # mean over ranks folded into the gradient read
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
        loss = eng.train_step(x, t_int, eps, apply=False)      # backward fires _grad_ready per layer
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
            torch.cuda.current_stream(eng.device).wait_stream(red.comm_stream)
        else:
            for idx in range(len(red.buckets)):
                lo, hi = red.wait_bucket(idx)
                eng.apply_adam(lo, hi, grad_div=float(self.world))
        eng.finish_step()
        return loss
