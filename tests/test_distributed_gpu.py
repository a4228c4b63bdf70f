"""The data-parallel step END TO END on real kernels: two processes share the one GPU of the test box and exchange
gradients through gloo (RCCL refuses two ranks on one device), which drives exactly the code the N-GPU run uses - hooks from the
two-stream reverse pass, bucketed all-reduce on the communication stream, per-bucket Adam behind it - and must reproduce the
single-process step on the global batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import denoiser_oracle as O

pytestmark = pytest.mark.gpu
CASES = {   # name: (dtype code, topology, steps, tolerance on the final parameters, bucket size in elements)
    "f32": (0, dict(size=16, pixel_size=8, max_size=16, octaves=2), 2, 2e-6, 1500),
    "bf16": (1, dict(size=16, pixel_size=128, max_size=256, octaves=2), 2, 2e-3, 200_000),
}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _batches(cfg, nsteps):
    return [O.synthetic_batch(cfg, seed=10 + k) for k in range(nsteps)]


def _run(eng_or_dp, cfg, nsteps, sl, dev):
    for x, t_int, eps in _batches(cfg, nsteps):
        eng_or_dp.train_step(torch.tensor(x[sl], dtype=torch.float32, device=dev), torch.tensor(t_int[sl]),
                             torch.tensor(eps[sl], dtype=torch.float32))
    torch.cuda.synchronize()


def _worker(rank, world, port, case, out_dir, mode="allreduce"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gan_class_transfer2_amd as g
        from gan_class_transfer2_amd.distributed import DataParallelStep, ShardedDataParallelStep
        dt, topo_kw, nsteps, _, bucket = CASES[case]
        cfg = O.OracleConfig(batch_size=4, **topo_kw)
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        eng = g.UNetEngine(g.Topology(cfg.pixel_size, cfg.max_size, cfg.octaves), dt, dev)
        eng.set_params(O.init_params(cfg, seed=21))
        per = cfg.batch_size // world
        if mode == "sharded":
            dp = ShardedDataParallelStep(eng, bucket_elems=bucket)
            assert dp.world == world and dp.exchange and len(dp.buckets) >= 2
            _run(dp, cfg, nsteps, slice(rank * per, (rank + 1) * per), dev)
            assert eng.iterations == nsteps and dp.launched == len(dp.buckets)
            if eng.arena.shadow is not None:       # what the replicas compute with: bit-identical operand copies
                np.save(os.path.join(out_dir, f"shadow{rank}.npy"), eng.arena.shadow.float().cpu().numpy())
            dp.gather_master()                     # the fp32 masters are sharded: assemble them for the comparison
        else:
            dp = DataParallelStep(eng, bucket_elems=bucket)
            assert dp.world == world and dp.reducer.exchange and len(dp.reducer.buckets) >= 2
            _run(dp, cfg, nsteps, slice(rank * per, (rank + 1) * per), dev)
            assert eng.iterations == nsteps and dp.reducer.launched == len(dp.reducer.buckets)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **eng.get_params())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["allreduce", "sharded"])
@pytest.mark.parametrize("case", ["f32", "bf16"])
def test_two_ranks_on_one_gpu_equal_global_batch(gpu, tmp_path, case, mode):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, case, str(tmp_path), mode), nprocs=world, join=True)
    if mode == "sharded" and case == "bf16":
        assert np.array_equal(np.load(tmp_path / "shadow0.npy"), np.load(tmp_path / "shadow1.npy"))
    import gan_class_transfer2_amd as g
    dt, topo_kw, nsteps, tol, _ = CASES[case]
    cfg = O.OracleConfig(batch_size=4, **topo_kw)
    eng = g.UNetEngine(g.Topology(cfg.pixel_size, cfg.max_size, cfg.octaves), dt, gpu)
    eng.set_params(O.init_params(cfg, seed=21))
    _run(eng, cfg, nsteps, slice(0, cfg.batch_size), gpu)
    ref = eng.get_params()
    r0, r1 = (np.load(tmp_path / f"rank{r}.npz") for r in range(world))
    for k, v in ref.items():
        assert np.array_equal(r0[k], r1[k]), k                                   # replicas stay bit-identical
        err = np.linalg.norm(r0[k].astype(np.float64) - v) / (np.linalg.norm(v) + 1e-30)
        assert err <= tol, (k, err)
