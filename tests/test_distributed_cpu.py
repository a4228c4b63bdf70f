"""world_size-2 gloo run (CPU) of the data-parallel exchange step: bucketed all-reduce of the gradient
arena in backward-completion order, mean folded into the optimizer read.  Property proven here:
  N replicas on bs/N images each, gradients summed and divided by N  ==  one process on the global batch
(the loss is a mean over the batch, train.py:272), and replicas end bit-identical."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import denoiser_oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gan_class_transfer2_amd as g
        from gan_class_transfer2_amd.distributed import BucketedAllReducer
        from gan_class_transfer2_amd.engine import ParamArena

        cfg = O.OracleConfig(size=16, pixel_size=8, max_size=16, octaves=2, batch_size=4)
        topo = g.Topology(cfg.pixel_size, cfg.max_size, cfg.octaves)
        params = O.init_params(cfg, seed=21)
        x, t_int, eps = O.synthetic_batch(cfg, seed=5)
        per = cfg.batch_size // world
        sl = slice(rank * per, (rank + 1) * per)
        _, _, g_local, _ = O.trainer_step(params, x[sl], t_int[sl], eps[sl], cfg)
        _, _, g_global, _ = O.trainer_step(params, x, t_int, eps, cfg)

        A = ParamArena(topo, g.F32, torch.device("cpu"))
        for k, v in g_local.items():
            A.grad(k).copy_(torch.tensor(v, dtype=torch.float32))
        red = BucketedAllReducer(A.g, A.ready_order(), A.layer_ranges, bucket_elems=1500)
        assert red.world == world and len(red.buckets) >= 2
        assert red.buckets[0][0] == 0 and red.buckets[-1][1] == A.total
        assert all(red.buckets[i][1] == red.buckets[i + 1][0] for i in range(len(red.buckets) - 1))
        red.begin()
        for layer in ["dense"] + A.ready_order():  # the order UNetEngine.backward fires its hook in
            red.grad_ready(layer)
        assert red.launched == len(red.buckets)
        for i in range(len(red.buckets)):
            red.wait_bucket(i)
        for k, v in g_global.items():
            got = A.grad(k).numpy().astype(np.float64) / world
            err = np.linalg.norm(got - v) / (np.linalg.norm(v) + 1e-30)
            assert err < 1e-6, (k, err)
        # replicas bit-identical after the exchange
        gathered = [torch.zeros_like(A.g) for _ in range(world)]
        dist.all_gather(gathered, A.g)
        assert all(torch.equal(gathered[0], t) for t in gathered)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_bucketed_allreduce_equals_global_batch(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_single_process_reducer_is_a_noop():
    import gan_class_transfer2_amd as g
    from gan_class_transfer2_amd.distributed import BucketedAllReducer
    from gan_class_transfer2_amd.engine import ParamArena
    topo = g.Topology(8, 16, 2)
    A = ParamArena(topo, g.F32, torch.device("cpu"))
    A.g.fill_(3.0)
    red = BucketedAllReducer(A.g, A.ready_order(), A.layer_ranges)
    red.begin()
    for layer in ["dense"] + A.ready_order():
        red.grad_ready(layer)
    assert red.world == 1 and red.launched == 0 and red.wait_bucket(0) == red.buckets[0] and float(A.g.min()) == 3.0


class _CpuEngine:
    """the slice of UNetEngine that ShardedDataParallelStep drives, on CPU tensors: gradients from the oracle, ready hooks in
    backward order, Keras Adam (oracle arithmetic) on arena ranges."""

    def __init__(self, cfg, topo, params, dtype=0, loss_scaling=False):
        from gan_class_transfer2_amd.engine import ParamArena
        self.cfg, self.topo, self.device = cfg, topo, torch.device("cpu")
        self.arena = ParamArena(topo, dtype, self.device)        # 16-bit dtype: a compute-dtype shadow arena like the GPU engine's
        for k, v in params.items():
            self.arena.param(k).copy_(torch.tensor(v, dtype=torch.float32))
        self._cast_shadow(0, self.arena.total)
        # gct2_loss_scale_state as the GPU engine keeps it: [scale bits, inv_scale bits, good_steps, found_inf, applied_steps, ...]
        self.ls_state = torch.zeros(8, dtype=torch.int32) if loss_scaling else None
        self.scale = 2.0 ** 15
        self.grad_ready_hook, self.iterations = None, 0
        self.inject = None          # (flat arena position, value) written into the LOCAL gradient before the hooks fire

    def _stream(self):
        return None

    def _cast_shadow(self, lo, hi):
        A = self.arena
        if A.shadow is not None:
            A.shadow[lo:hi] = A.p[lo:hi].to(A.shadow.dtype)

    def operand_params(self):
        """what a replica computes with: 16-bit kernels from the shadow, biases and the Dense(3) layer in fp32 from the master
        arena (engine.forward hands A.wptr(kernel) and A.pptr(bias / dense) to the library)."""
        A = self.arena
        out = {}
        for k in A.shapes:
            fp32_read = k.endswith(".b") or k.startswith("dense.") or A.shadow is None
            src = A.p if fp32_read else A.shadow
            out[k] = A._view(src, k).to(torch.float64).numpy().copy()
        return out

    def train_step(self, x, t_int, eps, apply=False):
        if self.ls_state is not None:
            self.ls_state[3] = 0                                   # gct2_loss_scale_begin
        loss, _, grads, _ = O.trainer_step(self.operand_params(), x, t_int, eps, self.cfg,
                                           loss_scale=self.scale if self.ls_state is not None else 1.0)
        for k, v in grads.items():
            self.arena.grad(k).copy_(torch.tensor(v, dtype=torch.float32))
        if self.inject is not None:
            self.arena.g[self.inject[0]] = self.inject[1]
        for layer in ["dense"] + self.arena.ready_order():   # UNetEngine.backward: head, the layers, the fp32 zone last
            self.grad_ready_hook(layer)
        return loss

    def check_finite(self, lo=0, hi=None, stream=None):
        hi = self.arena.total if hi is None else hi
        if not bool(torch.isfinite(self.arena.g[lo:hi]).all()):
            self.ls_state[3] = 1

    def apply_adam(self, lo=0, hi=None, grad_div=1.0, stream=None):
        A = self.arena
        hi = A.total if hi is None else hi
        if self.ls_state is not None and int(self.ls_state[3]) != 0:
            return                                                 # LossScaleOptimizer: the whole update is skipped
        inv = np.float32(1.0 / self.scale) if self.ls_state is not None else np.float32(1.0)
        p, m, v = O.keras_adam_step(A.p[lo:hi].numpy(), A.g[lo:hi].numpy() * inv / np.float32(grad_div), A.m[lo:hi].numpy(),
                                    A.v[lo:hi].numpy(), self.iterations, self.cfg)
        A.p[lo:hi] = torch.tensor(p); A.m[lo:hi] = torch.tensor(m); A.v[lo:hi] = torch.tensor(v)
        self._cast_shadow(lo, hi)

    def finish_step(self):
        if self.ls_state is not None and int(self.ls_state[3]) != 0:
            self.scale /= 2                                        # skipped step: iterations do not advance
            return
        self.iterations += 1


def _sharded_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gan_class_transfer2_amd as g
        from gan_class_transfer2_amd.distributed import ShardedDataParallelStep
        cfg = O.OracleConfig(size=16, pixel_size=8, max_size=16, octaves=2, batch_size=4)
        topo = g.Topology(cfg.pixel_size, cfg.max_size, cfg.octaves)
        params = O.init_params(cfg, seed=21)
        eng = _CpuEngine(cfg, topo, params)
        dp = ShardedDataParallelStep(eng, bucket_elems=1000)
        assert dp.world == world and dp.exchange and len(dp.buckets) >= 4
        assert all((hi - lo) % (world * 64) == 0 for lo, hi in dp.buckets) and dp.buckets[-1][1] == eng.arena.total
        per = cfg.batch_size // world
        for step in range(2):
            x, t_int, eps = O.synthetic_batch(cfg, seed=30 + step)
            sl = slice(rank * per, (rank + 1) * per)
            dp.train_step(x[sl], t_int[sl], eps[sl])
        assert eng.iterations == 2 and dp.launched == len(dp.buckets)
        # the weights every rank computes with (here fp32: the all-gathered parameter arena) are bit-identical ...
        gathered = [torch.zeros_like(eng.arena.p) for _ in range(world)]
        dist.all_gather(gathered, eng.arena.p)
        assert all(torch.equal(gathered[0], t) for t in gathered)
        # ... while Adam slots only live on the owning rank until gather_master()
        own = torch.zeros(eng.arena.total, dtype=torch.bool)
        for k in range(len(dp.buckets)):
            lo, hi = dp.shard(k)
            own[lo:hi] = True
        assert float(eng.arena.m[~own].abs().max()) == 0 and float(eng.arena.m[own].abs().max()) > 0
        dp.gather_master()
        if rank == 0:
            np.savez(os.path.join(out_dir, "sharded.npz"), p=eng.arena.p.numpy(), m=eng.arena.m.numpy(), v=eng.arena.v.numpy())
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharded_step_equals_global_batch(tmp_path):
    """ShardedDataParallelStep (reduce-scatter -> Adam on the own shard -> all-gather) on 2 gloo ranks with half batches == one
    process on the full batch: parameters and Adam slots after two steps (fp32 noise of the summation order only)."""
    world, port = 2, _free_port()
    mp.spawn(_sharded_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
    import gan_class_transfer2_amd as g
    cfg = O.OracleConfig(size=16, pixel_size=8, max_size=16, octaves=2, batch_size=4)
    eng = _CpuEngine(cfg, g.Topology(cfg.pixel_size, cfg.max_size, cfg.octaves), O.init_params(cfg, seed=21))
    eng.grad_ready_hook = lambda layer: None
    for step in range(2):
        x, t_int, eps = O.synthetic_batch(cfg, seed=30 + step)
        eng.train_step(x, t_int, eps)
        eng.apply_adam()
        eng.finish_step()
    z = np.load(tmp_path / "sharded.npz")
    for name in ("p", "m", "v"):
        ref = getattr(eng.arena, name).numpy().astype(np.float64)
        err = np.linalg.norm(z[name] - ref) / np.linalg.norm(ref)
        assert err <= 2e-6, (name, err)


# ---- world size 4: interior shards, buckets that do not end on layer boundaries, 16-bit shadow + fp32-read parameters, loss scaling
W4_CFG = dict(size=16, pixel_size=8, max_size=16, octaves=2, batch_size=4)
W4_STEPS, W4_INF_STEP = 3, 1


def _w4_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gan_class_transfer2_amd as g
        from gan_class_transfer2_amd.distributed import ShardedDataParallelStep
        cfg = O.OracleConfig(**W4_CFG)
        topo = g.Topology(cfg.pixel_size, cfg.max_size, cfg.octaves)
        eng = _CpuEngine(cfg, topo, O.init_params(cfg, seed=21), dtype=g.BF16, loss_scaling=True)
        dp = ShardedDataParallelStep(eng, bucket_elems=1000, tail_layers=2)      # (tiny topology: DownShuffle_0 + the fp32 zone)
        A = eng.arena
        assert dp.world == 4 and dp.exchange and len(dp.buckets) >= 5
        layer_ends = {hi for _, hi in A.layer_ranges.values()}
        assert any(hi not in layer_ends for _, hi in dp.buckets[:-1])                  # buckets cut through layers
        # the last bucket holds (about) the last layer and the fp32 zone only: its exchange is the exposed tail of the step.
        # It is REPLICATED: all-reduced and updated on every rank, so the fp32-read parameters never need an exchange of their own
        tail_lo = A.layer_ranges[A.ready_order()[-2]][0]
        assert dp.buckets[-1][1] == A.total and tail_lo - 4 * 64 < dp.buckets[-1][0] <= tail_lo
        assert dp.replicated(len(dp.buckets) - 1) and dp.shard(len(dp.buckets) - 1) == dp.buckets[-1]
        assert dp.buckets[-1][0] <= A.layer_ranges["fp32"][0] and not hasattr(dp, "_exchange_fp32_read_parameters")
        # an inf lands, at one step, in rank 2's LOCAL gradient at a position whose reduced value belongs to rank 1 (an interior
        # shard of a middle bucket): the owner's finite check + the 4-byte MAX all-reduce must make every rank skip
        k_mid = len(dp.buckets) // 2
        lo, hi = dp.buckets[k_mid]
        pos = lo + (hi - lo) // 4 + 3
        for step in range(W4_STEPS):
            x, t_int, eps = O.synthetic_batch(cfg, seed=40 + step)
            eng.inject = (pos, float("inf")) if (step == W4_INF_STEP and rank == 2) else None
            dp.train_step(x[rank:rank + 1], t_int[rank:rank + 1], eps[rank:rank + 1])
        assert eng.iterations == W4_STEPS - 1 and eng.scale == 2.0 ** 14
        # WITHOUT gather_master(): every rank computes with the same operands - 16-bit kernels AND the fp32-read biases / Dense(3)
        names = [n for n in A.shapes if n.endswith(".b") or n.startswith("dense.")]
        small = torch.cat([A.param(n).reshape(-1) for n in names])
        both = [torch.zeros_like(small) for _ in range(world)]
        dist.all_gather(both, small)
        assert all(torch.equal(both[0], t) for t in both), "fp32-read parameters differ between replicas"
        sh = [torch.zeros_like(A.shadow) for _ in range(world)]
        dist.all_gather(sh, A.shadow)
        assert all(torch.equal(sh[0], t) for t in sh)
        # the masters are sharded now: UNetEngine.state_dict() refuses until gather_master() (checked on the GPU engine)
        assert eng._masters_sharded
        dp.gather_master()
        assert not eng._masters_sharded
        if rank == 3:
            np.savez(os.path.join(out_dir, "w4.npz"), p=A.p.numpy(), m=A.m.numpy(), v=A.v.numpy(), small=small.numpy())
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_four_rank_gloo_sharded_step_interior_shards_loss_scaling(tmp_path):
    """ShardedDataParallelStep at world size 4: interior ranks' shards, fixed-size buckets that cut through
    layers, a small last bucket, the 16-bit shadow all-gather PLUS the fp32-read parameters (biases and Dense(3)
    were never exchanged), dynamic loss scaling with one rank producing an inf at one step.  Must equal one process on the global
    batch that skips the same step."""
    world, port = 4, _free_port()
    mp.spawn(_w4_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
    import gan_class_transfer2_amd as g
    cfg = O.OracleConfig(**W4_CFG)
    eng = _CpuEngine(cfg, g.Topology(cfg.pixel_size, cfg.max_size, cfg.octaves), O.init_params(cfg, seed=21), dtype=g.BF16, loss_scaling=True)
    eng.grad_ready_hook = lambda layer: None
    for step in range(W4_STEPS):
        x, t_int, eps = O.synthetic_batch(cfg, seed=40 + step)
        eng.inject = (0, float("inf")) if step == W4_INF_STEP else None
        eng.train_step(x, t_int, eps)
        eng.check_finite()
        eng.apply_adam()
        eng.finish_step()
    assert eng.iterations == W4_STEPS - 1
    z = np.load(tmp_path / "w4.npz")
    A = eng.arena
    for name in ("p", "m", "v"):
        ref = getattr(A, name).numpy().astype(np.float64)
        err = np.linalg.norm(z[name] - ref) / np.linalg.norm(ref)
        assert err <= 3e-6, (name, err)
    names = [n for n in A.shapes if n.endswith(".b") or n.startswith("dense.")]
    small_ref = torch.cat([A.param(n).reshape(-1) for n in names]).numpy()
    assert np.abs(small_ref).max() > 0                      # the biases did move (they start at zero)
    assert np.linalg.norm(z["small"] - small_ref) <= 3e-6 * np.linalg.norm(small_ref)


@pytest.mark.parametrize("tail_layers", [0, 1, 3, 99])
def test_sharded_tail_layers_is_clamped(tail_layers):
    """whatever tail_layers is given (0, larger than the number of ranges, a tail that rounds down to nothing), the bucket
    list ends with ONE replicated bucket that contains the whole fp32 zone, and the buckets tile the arena without gaps."""
    import gan_class_transfer2_amd as g
    from gan_class_transfer2_amd.distributed import ShardedDataParallelStep
    cfg = O.OracleConfig(size=16, pixel_size=8, max_size=16, octaves=2, batch_size=4)
    eng = _CpuEngine(cfg, g.Topology(cfg.pixel_size, cfg.max_size, cfg.octaves), O.init_params(cfg, seed=21))
    dp = ShardedDataParallelStep(eng, bucket_elems=1000, tail_layers=tail_layers)
    A = eng.arena
    assert dp.buckets[0][0] == 0 and dp.buckets[-1][1] == A.total
    assert all(dp.buckets[i][1] == dp.buckets[i + 1][0] for i in range(len(dp.buckets) - 1))
    assert dp.replicated(len(dp.buckets) - 1) and dp.buckets[-1][0] <= A.layer_ranges["fp32"][0]
    assert not any(dp.replicated(k) for k in range(len(dp.buckets) - 1))
