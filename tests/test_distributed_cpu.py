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
