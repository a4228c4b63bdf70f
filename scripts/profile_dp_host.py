"""host-side cost of the data-parallel step by function (cProfile; ONE rank, RCCL group of size 1, exchange forced): where the interpreter
time of DataParallelStep / ShardedDataParallelStep goes once the engine's own calls are replayed from a step plan.
usage: python scripts/profile_dp_host.py [allreduce|sharded] [iters]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import gan_class_transfer2_amd as g
from gan_class_transfer2_amd.distributed import DataParallelStep, ShardedDataParallelStep

mode = sys.argv[1] if len(sys.argv) > 1 else "allreduce"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.rand(64, 128, 128, 3, device=dev) * 2 - 1
eng = g.UNetEngine(g.Topology(128, 512, 6), g.BF16, dev)
step = (DataParallelStep if mode == "allreduce" else ShardedDataParallelStep)(eng, force_exchange=True)
for _ in range(10):
    step.train_step(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    step.train_step(x)
host = (time.perf_counter() - t0) / iters * 1e6
torch.cuda.synchronize()
print(f"{mode}: host enqueue {host:.0f} us per step, {len(step.buckets if mode != 'allreduce' else step.reducer.buckets)} buckets")
pr = cProfile.Profile()
pr.enable()
for _ in range(iters):
    step.train_step(x)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
dist.destroy_process_group()
