"""per-GPU cost of the data-parallel step variants without a second GPU (never a scaling measurement).  ONE rank, RCCL group of size
1, exchange forced.  Rows:
  * the plain step (fused per-layer Adam), first and last in the process: equal now that every engine of a caller runs on the same
    probed streams (engine.distinct_stream; r05: 2641 vs 2439 us);
  * the two exchange schemes with the size-1 collectives really issued (launch cost, no wire time);
  * MODELLED rows: every collective replaced by gct2_stream_occupy on the communication stream - a kernel that
    holds as many work-group slots as the collective library's kernel would (GCT2_DP_WGS, default 32 and 64 are both run) for
    base latency + bytes on the wire / (7 links x 153 GB/s x 0.8), SURVEY.md App. D: all-reduce 2 (N-1)/N S, reduce-scatter and
    all-gather (N-1)/N S each, N = 8 - and, for the sharded scheme, the optimizer on 1/8 of every sharded bucket (a rank's share).
    What is real in those rows: the dependency structure (what waits for the exchange and what the exchange waits for), the CUs the
    exchange takes from the GEMMs, the per-bucket optimizer launches.  What is a model: the duration of each exchange.
usage: python scripts/bench_dp_overhead.py [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import gan_class_transfer2_amd as g
from gan_class_transfer2_amd import engine as E
from gan_class_transfer2_amd.distributed import DataParallelStep, ShardedDataParallelStep

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
N, LINKS, LINK_BPS, EFF, BASE_US = 8, 7, 153e9, 0.8, 10.0
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.rand(64, 128, 128, 3, device=dev) * 2 - 1
topo = g.Topology(128, 512, 6)


def timed(fn, eng, n):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    eng.flush_deferred()           # optimizer launches a fused step held back belong to the timed work
    e1.record()
    torch.cuda.synchronize()
    t_gpu = e0.elapsed_time(e1) * 1e3 / n
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_host = (time.perf_counter() - t0) * 1e6 / n            # enqueue only (no synchronisation inside the loop)
    torch.cuda.synchronize()
    return t_gpu, t_host


def row(name, make):
    eng = g.UNetEngine(topo, g.BF16, dev)
    step = make(eng)
    t_gpu, t_host = timed(lambda: step.train_step(x), eng, iters)
    print("%-74s %8.1f us per step   host enqueue %7.1f us" % (name, t_gpu, t_host), flush=True)
    return t_gpu


def rank_of_8(e):
    # the per-rank OPTIMIZER work of an 8-GPU job on this one GPU: every sharded bucket is updated on 1/8 of its range (rank 0's shard),
    # the replicated last bucket in full
    st = ShardedDataParallelStep(e, force_exchange=True)
    st.world, st.rank = N, 0
    return st


plain0 = row("plain (fused Adam), first engine of the process", lambda e: e)
print("# streams picked (role, candidate, us a marker waited behind each stream it must run beside):", E._STREAM_LOG)
row("bucketed all-reduce, size-1 collectives issued (no wire time)", lambda e: DataParallelStep(e, force_exchange=True))
row("sharded, size-1 collectives issued (no wire time), whole optimizer here", lambda e: ShardedDataParallelStep(e, force_exchange=True))

# ---- modelled wire time ---------------------------------------------------------------------------------------------------------
_ar, _rs, _ag = dist.all_reduce, dist.reduce_scatter_tensor, dist.all_gather_into_tensor
wgs = 32
wire_total = [0.0]


def occupy(nbytes_on_wire):
    us = BASE_US + nbytes_on_wire / (LINKS * LINK_BPS * EFF) * 1e6
    wire_total[0] += us
    g._lib.call("gct2_stream_occupy", torch.cuda.current_stream(dev).cuda_stream, wgs, us)


def m_all_reduce(t, *a, **k):
    occupy(2.0 * (N - 1) / N * t.numel() * t.element_size())


def m_reduce_scatter(out, inp, *a, **k):
    occupy((N - 1) / N * inp.numel() * inp.element_size())


def m_all_gather(out, inp, *a, **k):
    occupy((N - 1) / N * out.numel() * out.element_size())


dist.all_reduce, dist.reduce_scatter_tensor, dist.all_gather_into_tensor = m_all_reduce, m_reduce_scatter, m_all_gather
res = {}
try:
    for wgs in (32, 64):
        for name, make in (("all-reduce", lambda e: DataParallelStep(e, force_exchange=True)), ("sharded, optimizer share of a rank of 8", rank_of_8)):
            wire_total[0] = 0.0
            t = row(f"MODELLED N={N}: {name}; exchange holds {wgs} work-group slots", make)
            per_step = wire_total[0] / (2 * iters + 20)
            print(f"    modelled exchange time enqueued per step: {per_step:7.1f} us (serial sum over the buckets; base {BASE_US} us + bytes / {LINKS * LINK_BPS * EFF / 1e9:.0f} GB/s each)")
            res[(name, wgs)] = t
finally:
    dist.all_reduce, dist.reduce_scatter_tensor, dist.all_gather_into_tensor = _ar, _rs, _ag
plain1 = row("plain (fused Adam), last engine of the process", lambda e: e)
print(f"# order independence: first {plain0:.1f} vs last {plain1:.1f} us ({(plain1 / plain0 - 1) * 100:+.1f} %)")
for (name, w), t in res.items():
    print(f"# MODELLED weak-scaling efficiency at N={N} ({name}, {w} slots): plain / modelled step = {plain1 / t:.3f}")
dist.destroy_process_group()
