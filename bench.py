#!/usr/bin/env python3
"""Headline benchmark: images/sec of the full `Trainer` train step (RNG + noise -> U-Net forward -> fp32 MSE ->
backward -> [RCCL all-reduce] -> Keras Adam) on synthetic 3x128x128 batches, bs 64 per GPU, bf16 operands with
fp32 accumulation (BASELINE.json config 3; config 4 when launched on N > 1 GPUs).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...          # starts its N ranks itself (one process per GPU, before this process touches a GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `roofline` is for the dominant KERNEL SYMBOL (form, tile AND epilogue: 1:1 with a rocprofv3 row) by
the sum of its calls' MEDIAN durations over >= 8 extra single-stream steps.  Those steps are REPLAYED from a recorded step plan whose
layer calls are bracketed by timed event records on the launching stream (gct2_plan_add_record_kind / gct2_plan_elapsed), each measured
step enqueued right behind an unmeasured one: no interpreter code runs between an event and its launch and the GPU never idles in
front of one (r05's eager leg booked host stalls as kernel time: DESIGN.md section 6).  The kernel each call selected comes from the
library's launch log; `roofline.achieved` = the symbol's algorithmic FLOPs / that time, `min / median / max` per symbol are in
`kernel_symbols`, `roofline.suspect` is set when the figure contradicts the step time; `kernels` keeps the per-family table;
`roofline.step_frac` (= `step_roofline_frac`) is the whole step against the 2.5 PFLOP/s dense bf16 MFMA peak (SURVEY.md §8d:
F_train = 32.1314 GFLOP/image at 128^2).
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


def call_flops(name, a):
    """algorithmic FLOPs (2*M*N*K, padding taps counted) of one layer call; argument positions as in include/gct2.h"""
    if name == "gct2_convT4s2_fwd_head_train":
        Bn, H, W, Cin, Cout = a[15:20]
        return 2.0 * Bn * 4 * H * W * Cout * 4 * Cin
    off = 9 if name.endswith("dgrad") else 8
    Bn, H, W, Cin, Cout = a[off:off + 5]
    # conv4s2_*: H, W = the big grid, (H/2)(W/2) * 16 Cin = H W 4 Cin; convT4s2_*: H, W = the small grid, 4 H W * 4 Cin
    return 2.0 * Bn * H * W * Cout * (16 * Cin if name.startswith("gct2_convT") else 4 * Cin)


def call_label(name, a):
    """("U2", "dgrad") ... of a layer call: level from the channel counts of the reference topology is ambiguous, so from the grid: the
    level of a layer is log2(image height / its big-grid height), the image height is the largest big grid seen (set by bench main)"""
    kind = "U" if name.startswith("gct2_convT") else "D"
    what = "fwd" if "fwd" in name else ("dgrad" if name.endswith("dgrad") else "wgrad")
    h = a[16] if name == "gct2_convT4s2_fwd_head_train" else a[10 if what == "dgrad" else 9]
    big = 2 * h if kind == "U" else h
    return (f"{kind}{(call_label.size // big).bit_length() - 1}", what)


call_label.size = 128


_FORM = {"conv": 0, "convT": 1, "s1": 2}


def kernel_symbol(token):
    """launch-log token (include/gct2.h gct2_ctx_log_launches) -> the kernel symbol, one per rocprofv3 row: form, tile and EPILOGUE
    (r05 merged `tap:conv:256x128:bias_act` - DownShuffle_1's forward, 67 us - with `...:mask` - the UpShuffle input gradients,
    141 us - into one "symbol" that no profiler row corresponds to)"""
    t = token.split(":")
    if t[0] == "wgrad":
        return "wgrad256q_kernel" if t[1].startswith("256") else "wgrad_kernel"
    if t[0] == "tap":
        return f"tapgemm_kernel<{t[1]},{t[2]},{t[3]}>"
    if t[0] == "halo":
        return f"halo_convT_kernel<{'head' if 'head' in t else t[2]}>"
    if t[0] == "rgb":
        return f"rgb_{t[1]}_kernel"
    return token


def rocprof_pattern(symbol):
    """substring of the MANGLED kernel name (rocprofv3 --mangled-kernels) that this symbol's launches carry, dtype left out:
    scripts/compare_bench_rocprof.py joins the bench line with a kernel_stats.csv through it"""
    if symbol.startswith("tapgemm_kernel<"):
        form, tile, epi = symbol[len("tapgemm_kernel<"):-1].split(",")
        bm, bn = tile.split("x")
        return f"tapgemm_kernelI*Li{_FORM[form]}ELi{bm}ELi{bn}ELi{0 if epi == 'bias_act' else 1}E"
    if symbol.startswith("halo_convT_kernel<"):
        return f"halo_convT_kernelI*Li{('bias_act', 'mask', 'head').index(symbol[len('halo_convT_kernel<'):-1])}E"
    return symbol.split("<")[0] + "I"


class KernelTimer:
    """Per layer call of the step: family, kernel symbol (launch log), FLOPs and - through a recorded step plan - timed event records
    on the stream the call is launched on.  Two phases per leg, driven by the engine's own plan machinery (a step shape is run
    eagerly once, then recorded and replayed):
      * the EAGER step of the leg: every layer call runs with the launch log on -> which kernel it selected (no events);
      * the RECORDING step: `timed record, call, timed record` are appended to the plan instead of a host-side event pair, so a replay
        is one C call per segment with nothing of the interpreter between an event and its launch."""

    FAMILY = {
        "gct2_conv4s2_fwd": "conv_form", "gct2_convT4s2_dgrad": "conv_form",
        "gct2_convT4s2_fwd": "convT_form", "gct2_conv4s2_dgrad": "convT_form", "gct2_convT4s2_fwd_head_train": "convT_form",
        "gct2_conv4s2_wgrad": "wgrad", "gct2_convT4s2_wgrad": "wgrad",
    }

    def __init__(self):
        self.enabled = False
        self.split_adam = False   # isolated leg: the fused optimizer launch of a weight-gradient call is issued BEHIND the end record
        self.ctxs = {}            # ctx handle -> _lib.Context (launch logs)
        self.names = []           # eager step: (family, symbol, flops, (layer, direction)) per layer call, in call order
        self.pairs = []           # recording step: (start record, end record, (layer, direction)) per layer call, in call order
        self.plan = None          # the _lib.Plan the pairs belong to
        self.lib = None
        self.context_source = None

    def reset(self):
        self.names, self.pairs, self.plan = [], [], None

    def install(self, engine_module, lib_module, contexts):
        orig = lib_module.call
        timer = self
        timer.lib = lib_module
        for c in contexts:
            self.ctxs[c.handle] = c

        def timed_call(name, *args):
            fam = timer.FAMILY.get(name) if timer.enabled else None
            if fam is None:
                return orig(name, *args)
            label = call_label(name, args)
            P = lib_module._recording
            if P is None:
                # eager step: the kernel this call selects, from its context's launch log
                ctx = timer.ctxs.get(args[0])
                if ctx is None and timer.context_source is not None:      # (contexts the engine created since: the deferred layers')
                    timer.ctxs.update({c.handle: c for c in timer.context_source()})
                    ctx = timer.ctxs.get(args[0])
                if ctx is None:
                    raise RuntimeError(f"bench.py: {name} was called with a context the timer does not know")
                ctx.log_launches(True)
                orig(name, *args)
                toks = [t for t in ctx.read_launch_log() if not t.startswith(("relu_bits", "bias_queue"))]
                ctx.log_launches(False)
                sym = kernel_symbol(toks[0]) if toks else name
                timer.names.append(("other" if sym.startswith("rgb") else fam, sym, call_flops(name, args), label))
                return None
            # recording step: events go on the stream the kernel is launched on (the last argument) - the weight gradients run on the
            # engine's side stream in the two-stream leg, where their durations are the in-situ ones, like rocprofv3's
            if timer.plan is None:
                timer.plan = P
            h = int(args[-1] or 0)
            adam = None
            if timer.split_adam and fam == "wgrad" and args[-2]:
                # the engine's deferral mechanism (gct2_adam_args.defer + gct2_adam_apply: the same launch, the same bits): the struct is
                # read when the call is MADE, i.e. at every replay - `defer` stays set (the engine records the struct's inputs after the
                # recording step and restores them in front of every replayed reverse pass), the optimizer launch follows the end record
                adam = lib_module.AdamArgs.from_address(int(args[-2]))
                if adam.defer:
                    adam = None                      # (already deferred by the engine)
                else:
                    adam.defer = 1
            e0 = P.record(h, lib_module.EVENT_TIMED)
            orig(name, *args)
            e1 = P.record(h, lib_module.EVENT_TIMED)
            if adam is not None:
                orig("gct2_adam_apply", int(args[-2]), args[6], 16 * args[11] * args[12], args[-1])
            timer.pairs.append((e0, e1, label))
            return None

        engine_module.call = timed_call

    def read(self):
        """milliseconds of every bracketed call of the LAST replay (the caller synchronised)"""
        return [self.plan.elapsed_ms(e0, e1) for e0, e1, _ in self.pairs]


def median(v):
    v = sorted(v)
    n = len(v)
    return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])


def symbol_table(names, samples, key):
    """names: (family, symbol, flops, label) per call; samples[i] = the call's durations in ms over the measured steps.
    Per key (0 family, 1 symbol): launches per step, the SUM OF THE CALLS' MEDIANS (robust against a stalled sample), FLOPs,
    and min / median / max over every sample of every call of the key"""
    rows = {}
    for (nm, ms) in zip(names, samples):
        r = rows.setdefault(nm[key], {"n": 0, "sum_median_ms": 0.0, "flops": 0.0, "all": [], "layers": []})
        r["n"] += 1
        r["sum_median_ms"] += median(ms)
        r["flops"] += nm[2]
        r["all"] += list(ms)
        r["layers"].append(f"{nm[3][0]}.{nm[3][1]}")
    out = {}
    for k, r in rows.items():
        t = r["sum_median_ms"] * 1e-3
        out[k] = {"launches_per_step": r["n"], "ms_per_step": round(r["sum_median_ms"], 4), "gflop_per_launch": round(r["flops"] / r["n"] / 1e9, 3),
                  "avg_launch_us": round(r["sum_median_ms"] / r["n"] * 1e3, 2), "min_us": round(min(r["all"]) * 1e3, 2),
                  "median_us": round(median(r["all"]) * 1e3, 2), "max_us": round(max(r["all"]) * 1e3, 2),
                  "tflops": round(r["flops"] / t / 1e12, 2) if t > 0 else None}
        if key == 1:
            out[k]["layers"] = sorted(r["layers"])
            out[k]["rocprof_mangled"] = rocprof_pattern(k)
    return rows, out


def pmc_evidence(layers, default_config):
    """(traffic bytes per launch, source, mfma utilisation, source) of the launches `layers` = {("U0", "wgrad"), ...} (the calls of this
    run that selected the dominant kernel) from the newest committed PMC files over the engine's own launches"""
    import glob
    if not default_config or not layers:
        return None, None, None, None
    traffic = tsrc = util = usrc = None
    tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_per_layer.json")))
    tfiles = [f for f in tfiles if "before" not in f]
    if tfiles:
        with open(tfiles[-1]) as f:
            rows = json.load(f)["layers"]
        sel = [r for r in rows if (r["layer"], r["dir"].split("+")[0]) in layers]
        if sel:
            traffic = sum(r["read_MB"] + r["write_MB"] for r in sel) * 1e6 / len(sel)
            tsrc = (f"profiles/{os.path.basename(tfiles[-1])}: mean over the {len(sel)} launches of this kernel in the ENGINE'S OWN step (its strides, "
                    "bit planes, dispatch; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections; committed file - not this run)")
    ufiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_mfma_util.txt")))
    if ufiles:
        busy = cyc = 0.0
        with open(ufiles[-1]) as f:
            for line in f:
                c = line.split()
                if line.startswith(("#", "lay")) or len(c) < 8:
                    continue
                if (c[0], c[1].split("+")[0]) in layers:
                    busy += float(c[-5]); cyc += float(c[-3])
        if cyc:
            util = busy / (1024 * cyc)
            usrc = (f"profiles/{os.path.basename(ufiles[-1])}: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) summed over this kernel's "
                    "launches in the engine's own step (rocprofv3 --pmc; committed file - not this run)")
    return traffic, tsrc, util, usrc


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


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher in front of it: start the N ranks (one process per GPU) as ONE child -
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port> bench.py <same
    arguments>` - BEFORE this process has made any GPU call (nothing here initialises HIP; the parent never does), relay rank 0's
    JSON line and return the child's exit code.  A failed rank makes the child, hence this process, exit non-zero."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for out in proc.stdout:
        if out.startswith('{"metric"'):
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks exited 0 without a result line\n")
        rc = 1
    return rc


def launcher_selftest(args, world: int, rank: int) -> int:
    """GCT2_BENCH_LAUNCH_TEST=1 (tests/test_host_cpu.py, no GPU): the rank side of the launch contract without the workload - gloo
    rendezvous from the launcher's environment, barrier, MAX over ranks of a per-rank time, ONE JSON line from rank 0.  Never a
    measurement: `value` is 0 and `data` says so."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if os.environ.get("GCT2_BENCH_LAUNCH_TEST_FAIL_RANK") == str(rank):
        return 3                                            # a failing rank: the launcher must report it (test)
    dist.barrier()
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "launcher self-test", "value": 0.0, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": args.dtype, "data": "none (GCT2_BENCH_LAUNCH_TEST: launch contract only, no GPU work)",
                          "max_over_ranks": float(t[0])}), flush=True)
    dist.destroy_process_group()
    return 0


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
    ap.add_argument("--kernel-events-multi", action="store_true", help="run the per-kernel legs at N > 1 too (default: N = 1 only)")
    ap.add_argument("--event-steps", type=int, default=8, help="measured steps of the per-kernel (roofline) leg; per call the median is reported")
    ap.add_argument("--variant", type=int, default=0, help="tapgemm tile variant hook (0 auto, 2, 3): A/B timing only")
    ap.add_argument("--no-plan", action="store_true", help="run every step through the interpreter (no recorded step plan): A/B of the host side")
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
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))        # `python bench.py --gpus N`: this process only starts and relays the ranks
    if os.environ.get("GCT2_BENCH_LAUNCH_TEST") == "1":
        raise SystemExit(launcher_selftest(args, world, rank))
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
    eng.use_plan = not args.no_plan
    sharded = world > 1 and args.dp_mode == "sharded"
    dp = ShardedDataParallelStep(eng) if sharded else DataParallelStep(eng)
    dp.broadcast_parameters(0)

    B, S = args.batch, args.size
    call_label.size = S
    gen = torch.Generator().manual_seed(rank)          # loader contract: u8/128 - 1 (train.py:292)
    x = (torch.randint(0, 256, (B, S, S, 3), generator=gen).float() / 128 - 1).to(dev)

    timer = KernelTimer()
    if not args.no_kernel_events:
        timer.install(engine_mod, _lib, [eng.ctx, eng.ctx_tail])
        timer.context_source = lambda: [eng.ctx, eng.ctx_tail, *eng._defer_ctxs.values()]

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
    host_dt = time.perf_counter() - t0                 # host time to enqueue the K steps (the GPU is still running)
    barrier()
    dt = time.perf_counter() - t0
    # ---- per-kernel legs: extra steps AFTER the timed region, replayed from step plans that carry timed event records ---------------
    def event_leg(overlap, split_adam, nsamples):
        """one eager step (which kernel every layer call selects), one recording step, then `nsamples` x (an unmeasured replay + a
        measured replay enqueued right behind it, one synchronisation, the measured replay's event pairs read back).  Returns
        (names, samples) with samples[i] = the durations of call i in ms."""
        import gc
        overlap0, plan0 = eng.overlap, eng.use_plan
        barrier()
        eng._plans.clear(); eng._plan_seen.clear()           # the timed region's plan has no timed records: record this leg's own
        eng.overlap, eng.use_plan = overlap, True
        timer.reset()
        timer.enabled, timer.split_adam = True, split_adam
        gc.collect(); gc.disable()
        try:
            # eager steps (each names its layer calls afresh) until the engine records one: a step shape is recorded the
            # `plan_after`-th time it is seen, and the two-stream step changes shape once (its first step starts without deferred
            # optimizer launches); the recording step appends the timed records and is replayed at once
            for _ in range(2 * eng.plan_after + 2):
                if timer.plan is not None:
                    break
                keep = timer.names
                timer.names = []
                dp.train_step(x)
                if timer.plan is not None:
                    timer.names = keep                        # (the recording step names nothing)
            if timer.plan is None or [n[3] for n in timer.names] != [p_[2] for p_ in timer.pairs]:
                raise RuntimeError("bench.py: the recorded step's layer calls do not match the eager step's "
                                   f"({len(timer.names)} named, {len(timer.pairs)} bracketed)")
            barrier()
            samples = [[] for _ in timer.pairs]
            for _ in range(nsamples):
                dp.train_step(x)                              # keeps the GPU busy while the host enqueues the measured step
                dp.train_step(x)
                barrier()
                for i, ms in enumerate(timer.read()):
                    samples[i].append(ms)
        finally:
            gc.enable()
            timer.enabled, timer.split_adam = False, False
            eng.flush_deferred()
            barrier()
            eng._plans.clear(); eng._plan_seen.clear()       # (these plans bake the leg's deferral of the optimizer launches)
            eng.overlap, eng.use_plan = overlap0, plan0
        return list(timer.names), samples

    situ = iso = leg_error = None
    # N > 1: the per-kernel legs are off unless asked for (--kernel-events-multi): a rank-local failure inside a leg would leave the other
    # ranks waiting in a collective, and the scaling runs only need `value`; the roofline block of a multi-GPU line carries step_frac
    if not args.no_kernel_events and (world == 1 or args.kernel_events_multi):
        # (the headline number is already measured: a failure in a per-kernel leg is reported in the line, it never costs the line)
        try:
            # in-situ leg: the step as timed (two streams): what a call takes while it shares the chip with the other stream's kernels
            # (a weight-gradient call includes its fused optimizer launch)
            if eng.overlap:
                situ = event_leg(True, False, 4)
            # roofline leg: the same step on ONE stream, so that every launch has the chip to itself and its event-pair duration is
            # the kernel's own (+ the helper launches of calls that have any: split-K finalize); the fused optimizer launch of every
            # weight-gradient call is issued behind the call's end record (KernelTimer.split_adam)
            iso = event_leg(False, True, max(1, args.event_steps))
        except Exception as e:           # noqa: BLE001 - reported, see above
            leg_error = f"{type(e).__name__}: {e}"
            iso = None
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
            # stream_picks: (role, pool candidate, us a marker on it waited behind an occupy on each stream it runs beside) - a candidate
            # that shares a hardware queue with one of them (>= 300 us) is passed over (engine.distinct_stream)
            "host": {"step_plan": bool(eng.use_plan), "enqueue_ms_per_step": round(host_dt / args.steps * 1e3, 4),
                     "stream_picks": [list(p_) for p_ in engine_mod._STREAM_LOG]},
            "comm": comm,
            "flops_per_image": f_img,
            "step_roofline_frac": round(imgs / world * f_img / MFMA_PEAK, 5),
        }
        out["roofline"] = {"bound": "mfma", "step_frac": out["step_roofline_frac"], "peak": MFMA_PEAK / 1e12, "unit": "TFLOP/s"}
        if leg_error is not None:
            out["roofline"]["leg_error"] = leg_error
        if iso is not None:
            names, samples = iso
            _, fams = symbol_table(names, samples, 0)
            rows, syms = symbol_table(names, samples, 1)
            dom = max(rows, key=lambda k: rows[k]["sum_median_ms"])        # the kernel symbol with the largest sum of per-call medians
            r = rows[dom]
            achieved = r["flops"] / (r["sum_median_ms"] * 1e-3) / 1e12        # its algorithmic FLOPs / its time, per step
            # traffic / mfma_util are NOT measured in this run: they come from the committed rocprofv3 --pmc passes over the ENGINE'S
            # OWN launches (scripts/engine_layers.py --pmc, scripts/collect_engine_pmc.py, scripts/profile_round.sh), default config
            # only, summed over the layers whose calls selected the dominant kernel; the *_source fields say so.
            dom_layers = sorted({nm[3] for nm in names if nm[1] == dom})
            traffic, traffic_source, mfma_util, mfma_source = pmc_evidence(set(dom_layers), (S, B, args.dtype, world) == (128, 64, "bf16", 1))
            # cross-checks: no kernel can take more of the one-stream step than the step (the timed two-stream step is at most ~10 %
            # shorter than the one-stream one); a sample far from its call's median is reported, never averaged in
            step_ms = dt / args.steps * 1e3
            outliers = [{"call": f"{nm[3][0]}.{nm[3][1]}", "symbol": nm[1], "median_us": round(median(ms) * 1e3, 2), "sample_us": round(max(ms) * 1e3, 2)}
                        for nm, ms in zip(names, samples) if max(ms) > 1.5 * median(ms) + 0.01]
            suspect = r["sum_median_ms"] > 1.2 * step_ms or sum(v["sum_median_ms"] for v in rows.values()) > 1.5 * step_ms
            out["roofline"].update({
                "kernel": dom, "kernel_layers": [f"{l}.{d}" for l, d in dom_layers], "achieved": round(achieved, 2), "frac": round(achieved * 1e12 / MFMA_PEAK, 5),
                "traffic": None if traffic is None else round(traffic), "traffic_source": traffic_source,
                "mfma_util": None if mfma_util is None else round(mfma_util, 4), "mfma_util_source": mfma_source,
                "flops_per_launch": r["flops"] / r["n"], "launches_per_step": r["n"], "event_steps": len(samples[0]),
                "avg_launch_us": round(r["sum_median_ms"] / r["n"] * 1e3, 2), "min_us": syms[dom]["min_us"], "median_us": syms[dom]["median_us"],
                "max_us": syms[dom]["max_us"], "kernel_ms_per_step": round(r["sum_median_ms"], 4), "rocprof_mangled": rocprof_pattern(dom),
                "suspect": bool(suspect), "outlier_samples": outliers[:8],
                "mode": f"one stream, {len(samples[0])} extra steps after the timed region, each REPLAYED from a recorded step plan right behind an unmeasured "
                        "replay (the GPU never waits for the host); timed event records on the launching stream around every layer call; per call the "
                        "MEDIAN over the steps, per symbol the sum of its calls' medians; the kernel of every call from the library's launch log; "
                        "weight-gradient calls without their optimizer launch"})
            out["kernel_symbols"] = syms
            out["kernels"] = fams
            if situ is not None:   # the same calls inside the two-stream step: a launch shares the chip with the other stream's, and each
                out["kernels_two_streams"] = symbol_table(situ[0], situ[1], 0)[1]      # wgrad call carries its fused Adam launch
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
