"""per-GPU cost of the data-parallel step variants without a second GPU: ONE rank, RCCL group of size 1, exchange forced
(collectives are issued and cost their launch, not their wire time).  Compares the plain step (fused per-layer Adam), the
bucketed all-reduce step and the sharded (reduce-scatter / Adam on the shard / all-gather) step.  A rehearsal of the code
path and its fixed overhead, never a scaling measurement.  usage: python scripts/bench_dp_overhead.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import gan_class_transfer2_amd as g
from gan_class_transfer2_amd.distributed import DataParallelStep, ShardedDataParallelStep

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
_engines = []
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.rand(64, 128, 128, 3, device=dev) * 2 - 1


def timed(fn, n):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    for e in list(_engines):
        e.flush_deferred()           # optimizer launches a fused step held back belong to the timed work
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for name, make in (("plain (fused Adam)", lambda e: e),
                   ("bucketed all-reduce", lambda e: DataParallelStep(e, force_exchange=True)),
                   ("sharded", lambda e: ShardedDataParallelStep(e, force_exchange=True))):
    eng = g.UNetEngine(g.Topology(128, 512, 6), g.BF16, dev)
    _engines[:] = [eng]
    step = make(eng)
    t_gpu = timed(lambda: step.train_step(x), iters)
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step.train_step(x)
    t_host = (time.perf_counter() - t0) * 1e6 / iters            # enqueue only (no synchronisation inside the loop)
    torch.cuda.synchronize()
    print("%-26s %8.1f us per step   host enqueue %8.1f us" % (name, t_gpu, t_host))
    del step, eng
# ... with the input-gradient chain on a high-priority stream of its own, as in the single-GPU step (one HIP stream more: the runtime
# multiplexes streams onto GPU_MAX_HW_QUEUES = 4 hardware queues, and streams that share one block each other; the data-parallel
# wrappers therefore switch it off - distributed._one_stream_less)
for name, make in (("all-reduce + chain stream", lambda e: DataParallelStep(e, force_exchange=True)),
                   ("sharded + chain stream", lambda e: ShardedDataParallelStep(e, force_exchange=True))):
    eng = g.UNetEngine(g.Topology(128, 512, 6), g.BF16, dev)
    step = make(eng)
    eng.chain_priority = True
    print("%-26s %8.1f us per step" % (name, timed(lambda: step.train_step(x), iters)))
    del step, eng

# the same two data-parallel steps with the collectives stubbed out (nothing enqueued): what is left is the stream / event plumbing
# and the per-bucket optimizer launches on the communication stream
class _Done:
    def wait(self):
        return True
_ar, _rs, _ag = dist.all_reduce, dist.reduce_scatter_tensor, dist.all_gather_into_tensor
dist.all_reduce = lambda *a, **k: _Done()
dist.reduce_scatter_tensor = lambda *a, **k: _Done()
dist.all_gather_into_tensor = lambda *a, **k: _Done()
def _as_rank_of_8(e):
    # the per-rank OPTIMIZER work of an 8-GPU job on this one GPU: every sharded bucket is updated on 1/8 of its range (rank 0's shard),
    # the replicated last bucket in full; collectives stubbed, so what is measured is the step + plumbing + 1/8 of the Adam traffic
    st = ShardedDataParallelStep(e, force_exchange=True)
    st.world, st.rank = 8, 0
    return st


for name, make in (("all-reduce, stubbed", lambda e: DataParallelStep(e, force_exchange=True)),
                   ("sharded, stubbed", lambda e: ShardedDataParallelStep(e, force_exchange=True)),
                   ("sharded, stubbed, Adam on 1/8 of every bucket (rank of 8)", _as_rank_of_8)):
    eng = g.UNetEngine(g.Topology(128, 512, 6), g.BF16, dev)
    _engines[:] = [eng]
    step = make(eng)
    print("%-22s %8.1f us per step" % (name, timed(lambda: step.train_step(x), iters)))
    del step, eng
# the plain step once more at the end of the process (same clocks / allocator state as the rows above)
eng = g.UNetEngine(g.Topology(128, 512, 6), g.BF16, dev)
_engines[:] = [eng]
print("%-22s %8.1f us per step" % ("plain (fused Adam), again", timed(lambda: eng.train_step(x), iters)))
del eng
# ... and with every optimizer launch deferred to the end of the reverse pass (none beside the GEMMs)
eng = g.UNetEngine(g.Topology(128, 512, 6), g.BF16, dev)
step = DataParallelStep(eng, force_exchange=True)
eng.grad_ready_hook = step.reducer.grad_ready
print("%-22s %8.1f us per step" % ("all-reduce, stubbed, Adam at the end", timed(lambda: step.train_step(x), iters)))
eng.grad_ready_hook = None
print("%-22s %8.1f us per step" % ("  ... and no hooks at all", timed(lambda: step.train_step(x), iters)))
del step, eng
dist.all_reduce, dist.reduce_scatter_tensor, dist.all_gather_into_tensor = _ar, _rs, _ag
dist.destroy_process_group()
