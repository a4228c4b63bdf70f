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
        red = BucketedAllReducer(A.g, topo.layer_order(), A.layer_ranges, bucket_elems=1500)
        assert red.world == world and len(red.buckets) >= 2
        assert red.buckets[0][0] == 0 and red.buckets[-1][1] == A.total
        assert all(red.buckets[i][1] == red.buckets[i + 1][0] for i in range(len(red.buckets) - 1))
        red.begin()
        for layer in topo.layer_order():           # the order UNetEngine.backward fires its hook in
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
    red = BucketedAllReducer(A.g, topo.layer_order(), A.layer_ranges)
    red.begin()
    for layer in topo.layer_order():
        red.grad_ready(layer)
    assert red.world == 1 and red.launched == 0 and red.wait_bucket(0) == red.buckets[0] and float(A.g.min()) == 3.0


class _CpuEngine:
    """the slice of UNetEngine that ShardedDataParallelStep drives, on CPU tensors: gradients from the oracle, ready hooks in
    backward order, Keras Adam (oracle arithmetic) on arena ranges."""

    def __init__(self, cfg, topo, params):
        import gan_class_transfer2_amd as g
        from gan_class_transfer2_amd.engine import ParamArena
        self.cfg, self.topo, self.device = cfg, topo, torch.device("cpu")
        self.arena = ParamArena(topo, g.F32, self.device)
        for k, v in params.items():
            self.arena.param(k).copy_(torch.tensor(v, dtype=torch.float32))
        self.ls_state, self.grad_ready_hook, self.iterations = None, None, 0

    def _stream(self):
        return None

    def train_step(self, x, t_int, eps, apply=False):
        params = {k: self.arena.param(k).numpy().astype(np.float64) for k in self.arena.shapes}
        loss, _, grads, _ = O.trainer_step(params, x, t_int, eps, self.cfg)
        for k, v in grads.items():
            self.arena.grad(k).copy_(torch.tensor(v, dtype=torch.float32))
        for layer in self.topo.layer_order():
            self.grad_ready_hook(layer)
        return loss

    def apply_adam(self, lo=0, hi=None, grad_div=1.0, stream=None):
        A = self.arena
        hi = A.total if hi is None else hi
        p, m, v = O.keras_adam_step(A.p[lo:hi].numpy(), A.g[lo:hi].numpy() / np.float32(grad_div), A.m[lo:hi].numpy(), A.v[lo:hi].numpy(),
                                    self.iterations, self.cfg)
        A.p[lo:hi] = torch.tensor(p); A.m[lo:hi] = torch.tensor(m); A.v[lo:hi] = torch.tensor(v)

    def finish_step(self):
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
