#!/usr/bin/env python3
"""Headline benchmark: images/sec of the full `Trainer` train step (RNG + noise -> U-Net forward -> fp32 MSE ->
backward -> [RCCL all-reduce] -> Keras Adam) on synthetic 3x128x128 batches, bs 64 per GPU, bf16 operands with
fp32 accumulation (BASELINE.json config 3; config 4 when launched on N > 1 GPUs).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel family (HIP events around every launch of
the three MFMA kernel families inside the timed region); `step_roofline_frac` is the whole step against the
2.5 PFLOP/s dense bf16 MFMA peak (F_train = 32.1314 GFLOP/image at 128^2).
`cpu_baseline` times oracle/torch_cross.py (a CPU restatement of train.py - TensorFlow is not installable here) on
the host cores, rank 0, N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK = 2.5e15          # dense bf16, MI355X_MICROARCH.md chip table


def layer_flops(topo, B, H, W):
    """algorithmic FLOPs (2*M*N*K, zero-padding taps counted) per C-ABI call family for one step."""
    n = topo.octaves
    f = {"conv_form": 0.0, "convT_form": 0.0, "wgrad": 0.0, "other": 0.0}
    for i in range(n):
        h, w = H >> i, W >> i
        d = 2.0 * B * (h // 2) * (w // 2) * topo.fd(i) * 16 * topo.cx(i)          # D_i forward
        u = 2.0 * B * h * w * topo.fu(i) * 4 * topo.up_in(i)                      # U_i forward
        f["conv_form"] += u                      # U_i dgrad runs the conv-form kernel
        f["convT_form"] += u                     # U_i forward
        f["wgrad"] += u + (d if i > 0 else 0.0)
        if i == 0:
            f["other"] += 2 * d                  # 3-channel image layer: direct kernels (fwd + wgrad), no dgrad
        else:
            f["conv_form"] += d                  # D_i forward
            f["convT_form"] += d                 # D_i dgrad
    dense = 2.0 * B * H * W * 3 * (topo.fu(0) + 3)
    f["other"] += 3 * dense
    return f


def f_train_per_image(topo, H, W):
    f = layer_flops(topo, 1, H, W)
    return sum(f.values())


class KernelTimer:
    """HIP events on the stream each kernel is launched on, per kernel family."""

    FAMILY = {
        "gct2_conv4s2_fwd": "conv_form", "gct2_convT4s2_dgrad": "conv_form",
        "gct2_convT4s2_fwd": "convT_form", "gct2_conv4s2_dgrad": "convT_form",
        "gct2_conv4s2_wgrad": "wgrad", "gct2_convT4s2_wgrad": "wgrad",
    }

    def __init__(self):
        self.events = []      # (family, start, end) of the steps inside the timed region (two streams: in-situ durations)
        self.isolated = []    # the same from the serial-stream steps run after the timed region (one kernel at a time)
        self.sink = self.events
        self.streams = {}     # raw stream handle -> torch stream object
        self.enabled = False

    def install(self, engine_module, lib_module):
        orig = lib_module.call
        timer = self

        def timed_call(name, *args):
            fam = timer.FAMILY.get(name) if timer.enabled else None
            # D0 (Cin = 3) runs the direct kernel: keep it out of the MFMA families
            if fam is not None and ((name == "gct2_conv4s2_fwd" and args[-4] == 3) or
                                    (name == "gct2_conv4s2_wgrad" and args[-5] == 3)):
                fam = None
            if fam is None:
                return orig(name, *args)
            # events go on the stream the kernel is launched on (the last argument): the weight gradients run on the engine's
            # side stream, concurrently with the dgrad chain - their durations are the in-situ ones, like rocprofv3's
            h = int(args[-1] or 0)
            st = timer.streams.get(h)
            if st is None:
                st = timer.streams[h] = torch.cuda.ExternalStream(h) if h else torch.cuda.default_stream()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(st)
            orig(name, *args)
            e.record(st)
            timer.sink.append((fam, s, e))

        engine_module.call = timed_call

    def summary(self, events):
        tot, cnt = {}, {}
        for fam, s, e in events:
            tot[fam] = tot.get(fam, 0.0) + s.elapsed_time(e) * 1e-3
            cnt[fam] = cnt.get(fam, 0) + 1
        return tot, cnt


def host_cores() -> int:
    """cores this process may really use: affinity mask and cgroup CPU quota (the GPU box gives a share of a big
    host; sizing the thread pool to os.cpu_count() there oversubscribes it badly)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


def cpu_baseline(topo_kw, size, batch, steps, warmup, budget_s=25.0):
    """timed CPU restatement; stops early once `budget_s` seconds of timed work have been spent."""
    from oracle import denoiser_oracle as O
    from oracle import torch_cross as T
    torch.set_num_threads(host_cores())
    cfg = O.OracleConfig(size=size, batch_size=batch, **topo_kw)
    tr = T.TorchCpuTrainer(cfg, O.init_params(cfg, 1234, dtype="float32"))
    gen = torch.Generator().manual_seed(0)
    x = torch.randint(0, 256, (batch, size, size, 3), generator=gen).float() / 128 - 1
    for _ in range(warmup):
        tr.train_step(x)
    t0 = time.perf_counter()
    done = 0
    for _ in range(steps):
        tr.train_step(x)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return batch * done / dt, dt / done, done


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--variant", type=int, default=0, help="tapgemm tile variant hook (0 auto, 2, 3): A/B timing only")
    ap.add_argument("--serial-streams", action="store_true",
                    help="run the weight gradients and Adam on the main stream (no overlap): per-kernel profiling runs")
    ap.add_argument("--dp-mode", default="sharded", choices=["sharded", "allreduce"],
                    help="N > 1: reduce-scatter + sharded Adam + all-gather of the weights (default), or bucketed all-reduce")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch with python -m torch.distributed.run --nproc-per-node N for --gpus N > 1")
    # rehearsal of the N > 1 code path on a one-GPU box (never a measurement): GCT2_BENCH_REHEARSAL=1 puts every rank on device 0
    # and exchanges through gloo (RCCL refuses two ranks on one device)
    rehearsal = os.environ.get("GCT2_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import gan_class_transfer2_amd as g
    from gan_class_transfer2_amd import _lib, engine as engine_mod
    from gan_class_transfer2_amd.distributed import DataParallelStep, ShardedDataParallelStep

    if _lib.build_flags() != 0:       # (_lib.load() refuses a stamped library already unless the diagnostic override is set)
        raise SystemExit(f"bench.py refuses a diagnostic build of libgct2.so (gct2_build_flags() = {_lib.build_flags()}): rebuild it")
    dtype = {"bf16": g.BF16, "f16": g.F16, "f32": g.F32}[args.dtype]
    topo = g.Topology(128, 512, 6)                      # reference defaults, train.py:18-21
    eng = g.UNetEngine(topo, dtype, dev, rng_seed=rank, loss_scaling=(args.dtype == "f16"))
    if args.variant:
        eng.ctx.set_tuning(args.variant)
    eng.overlap = not args.serial_streams
    sharded = world > 1 and args.dp_mode == "sharded"
    dp = ShardedDataParallelStep(eng) if sharded else DataParallelStep(eng)
    dp.broadcast_parameters(0)

    B, S = args.batch, args.size
    gen = torch.Generator().manual_seed(rank)          # loader contract: u8/128 - 1 (train.py:292)
    x = (torch.randint(0, 256, (B, S, S, 3), generator=gen).float() / 128 - 1).to(dev)

    timer = KernelTimer()
    if not args.no_kernel_events:
        timer.install(engine_mod, _lib)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        dp.train_step(x)
    barrier()
    # the timed region carries no probes: HIP events around every MFMA launch cost a barrier packet each (~8 % of a step), so the
    # per-kernel legs below run on extra steps AFTER it
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = dp.train_step(x)
    eng.flush_deferred()       # the optimizer launches the last step held back for the next forward pass: inside the timed region
    barrier()
    dt = time.perf_counter() - t0
    # in-situ leg: the step as timed (two streams), with events: what a launch takes while it shares the chip with the other stream
    ev_steps = 0
    if not args.no_kernel_events:
        timer.enabled = True
        for _ in range(4):
            dp.train_step(x)
            ev_steps += 1
        barrier()
        timer.enabled = False
    # roofline leg: the same step with ONE stream, so that every MFMA launch has the chip to itself and its HIP-event duration
    # is the kernel's own (with two streams a launch shares the CUs with whatever the other stream is running)
    iso_steps = 0
    if not args.no_kernel_events:
        overlap0, eng.overlap = eng.overlap, False
        fuse0, eng.fuse_adam = eng.fuse_adam, False              # the fused optimizer step would be timed as part of the wgrad calls
        timer.sink, timer.enabled = timer.isolated, True
        for _ in range(4):
            dp.train_step(x)
            iso_steps += 1
        barrier()
        timer.enabled, eng.overlap, eng.fuse_adam = False, overlap0, fuse0
    comm = None
    if world > 1:
        # evidence of what RCCL ran (rank 0): ranks it saw, the buckets of one extra step and the HIP-event time of every collective
        buckets = dp.buckets if sharded else dp.reducer.buckets
        esz = 4
        # streams this process drives: the caller's, the weight-gradient side stream, the communication stream (+ the chain's
        # own when chain_priority is on) - RCCL adds its internal one; more than GPU_MAX_HW_QUEUES = 4 share hardware queues
        streams = 2 + (1 if eng.chain_priority else 0) + 1
        comm = {"backend": dist.get_backend(), "rccl_ranks": dist.get_world_size(), "mode": args.dp_mode, "buckets": len(buckets),
                "bytes_per_bucket": [int((hi - lo) * esz) for lo, hi in buckets], "process_streams": streams,
                "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))}
        if sharded:
            dp.time_collectives, dp.events = True, []
            dp.train_step(x)
            barrier()
            dp.time_collectives = False
            comm["collective_ms"] = [[k, round(ms, 4)] for k, ms in dp.collective_times_ms()]
            comm["collectives"] = "per bucket: reduce_scatter (fp32 gradients), all_gather (compute-dtype weights)"
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax[0])
    loss_val = float(loss[0])

    if rank == 0:
        imgs = world * B * args.steps / dt
        f_img = f_train_per_image(topo, S, S)
        out = {
            "metric": f"images/sec (train step) 3x{S}x{S} bs={B}/GPU", "value": round(imgs, 2), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: all ranks on one GPU over gloo - not a measurement)",
            "config": {"workload": f"BASELINE config {(5 if args.dtype == 'f16' else 3 if world == 1 else 4)}: Trainer step, 3x{S}x{S}, bs {B}/GPU, octaves 6, "
                                   f"pixel_size 128, max_size 512, {args.dtype} operands / fp32 accumulate, Keras Adam + WarmUp",
                       "global_batch": world * B, "parallelism": f"dp{world}"},
            "loss": loss_val,
            "library": {"abi": int(_lib.load().gct2_abi_version()), "build_flags": int(_lib.build_flags())},
            "comm": comm,
            "flops_per_image": f_img,
            "step_roofline_frac": round(imgs / world * f_img / MFMA_PEAK, 5),
        }
        if timer.isolated:
            fl = layer_flops(topo, B, S, S)

            def families(events, nsteps):
                tot, cnt = timer.summary(events)
                return tot, cnt, {fam: {"launches_per_step": cnt[fam] // nsteps, "ms_per_step": round(tot[fam] / nsteps * 1e3, 4),
                                        "tflops": round(fl[fam] * nsteps / tot[fam] / 1e12, 2)} for fam in tot}

            tot, cnt, fams = families(timer.isolated, iso_steps)
            dom = max(tot, key=tot.get)
            achieved = fl[dom] * iso_steps / tot[dom] / 1e12
            # achieved = algorithmic FLOPs of ALL launches of the dominant family in the serial-stream steps / their summed
            # HIP-event time (= average FLOPs per launch / average launch duration).  traffic / mfma_util are NOT measured in this
            # run: they come from the committed rocprofv3 --pmc passes over the ENGINE'S OWN launches (scripts/engine_layers.py --pmc,
            # scripts/collect_engine_pmc.py, scripts/profile_round.sh), only for the default config; the *_source fields say so.
            traffic, traffic_source, mfma_util, mfma_source = None, None, None, None
            fam_of = lambda lay, d: ("wgrad" if d == "wgrad" else "conv_form" if (lay[0] == "D") == (d == "fwd") else "convT_form")
            tname = "r04_traffic_per_layer.json"
            tpath = os.path.join(ROOT, "profiles", tname)
            if os.path.exists(tpath) and (S, B, args.dtype, world) == (128, 64, "bf16", 1):
                with open(tpath) as f:
                    rows = [r for r in json.load(f)["layers"] if r["form"] != "rgb" and fam_of(r["layer"], r["dir"].split("+")[0]) == dom]
                if rows:
                    traffic = sum(r["read_MB"] + r["write_MB"] for r in rows) * 1e6 / len(rows)
                    traffic_source = (f"profiles/{tname}: mean of the {len(rows)} launches of this family in the ENGINE'S OWN step (its strides, bit planes, "
                                      "dispatch; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections; committed file - not this run)")
            uname = "r04_mfma_util.txt"
            upath = os.path.join(ROOT, "profiles", uname)
            if os.path.exists(upath) and (S, B, args.dtype, world) == (128, 64, "bf16", 1):
                busy = cyc = 0.0
                with open(upath) as f:
                    for line in f:
                        c = line.split()
                        if line.startswith(("#", "lay")) or len(c) < 8 or c[0] == "D0":
                            continue
                        if fam_of(c[0], c[1].split("+")[0]) == dom:
                            busy += float(c[-5]); cyc += float(c[-3])
                if cyc:
                    mfma_util = busy / (1024 * cyc)
                    mfma_source = (f"profiles/{uname}: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) summed over this family's launches in the "
                                   "engine's own step (rocprofv3 --pmc; committed file - not this run); in-kernel clocks: profiles/r04_kernel_clock.txt")
            out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": MFMA_PEAK / 1e12,
                               "unit": "TFLOP/s", "frac": round(achieved * 1e12 / MFMA_PEAK, 5),
                               "traffic": None if traffic is None else round(traffic), "traffic_source": traffic_source,
                               "mfma_util": None if mfma_util is None else round(mfma_util, 4), "mfma_util_source": mfma_source,
                               "flops_per_launch": fl[dom] / (cnt[dom] // iso_steps), "event_steps": iso_steps,
                               "avg_launch_us": round(tot[dom] / cnt[dom] * 1e6, 2),
                               "mode": "one stream (4 extra steps after the timed region): isolated launch durations"}
            out["kernels"] = fams
            if timer.events:   # the same calls inside the timed region: two streams share the chip, and each wgrad call also
                out["kernels_two_streams"] = families(timer.events, ev_steps)[2]      # carries its layer's fused Adam launch
        if world == 1 and not args.no_cpu_baseline:
            kw = dict(pixel_size=128, max_size=512)
            v3, s3, n3 = cpu_baseline(dict(octaves=6, **kw), S, 4, 100, 1, budget_s=12.0)   # ~12 s of CPU work
            v1, s1, n1 = cpu_baseline(dict(octaves=5, **kw), 32, 8, 10, 2)
            out["cpu_baseline"] = {"value": round(v3, 3), "unit": "images/sec", "cores": host_cores(), "kind": "port",
                                   "sample": f"oracle/torch_cross.py (PyTorch CPU fp32 restatement of train.py; TensorFlow unavailable), "
                                             f"same 3x{S}x{S} workload at batch 4, 1 warm-up + {n3} timed steps ({s3 * 1e3:.0f} ms/step)",
                                   "config1_images_per_sec": round(v1, 2), "config1_ms_per_step": round(s1 * 1e3, 2),
                                   "config1": f"BASELINE config 1: 3x32x32, bs 8, octaves 5, 2 warm-up + {n1} timed steps"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
