"""wall time of the phases of one train step, no profiler attached (diagnostic): forward (noise .. fused head), reverse pass
with the fused optimizer, whole step; compare with the per-kernel sums of profiles/rNN_kernel_stats*.csv to see what the
launch boundaries cost.  usage: python scripts/bench_phases.py [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gan_class_transfer2_amd as g
from gan_class_transfer2_amd.engine import Topology, UNetEngine, BF16

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
eng = UNetEngine(Topology(128, 512, 6), BF16, dev)
BATCH, SIZE = int(os.environ.get("AB_BATCH", "64")), int(os.environ.get("AB_SIZE", "128"))     # config 3 unless told otherwise
x = torch.rand(BATCH, SIZE, SIZE, 3, device=dev) * 2 - 1
b = eng.buffers(BATCH, SIZE, SIZE)


def fwd():
    eng.begin_step()
    eng.sample_and_noise_into_r0(b, x, keep_eps=False)
    eng.forward(b, head=False, stop_before_u0=True, planes=True)
    return eng.u0_head_train(b, x)


def bwd():
    eng.backward(b, head_done=True, adam_inline=True)
    eng.finish_step()


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    eng.flush_deferred()          # optimizer launches the last step held back: they belong to the timed work
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for _ in range(5):
    eng.train_step(x)
print("step      %8.1f us" % timed(lambda: eng.train_step(x), iters))
print("forward   %8.1f us  (back to back: host-bound if above the step's share)" % timed(fwd, iters))
fwd()
print("reverse   %8.1f us" % timed(bwd, iters))
# the forward again with a device-side idle gap in front of every repetition, so the host is ahead of the GPU as in the step
def fwd_ahead():
    torch.cuda._sleep(4_000_000)
    fwd()
t_sleep = timed(lambda: torch.cuda._sleep(4_000_000), iters)
print("forward   %8.1f us  (host ahead: enqueued behind a device-side sleep)" % (timed(fwd_ahead, iters) - t_sleep))
# in-process A/B of engine switches (boxes differ by +-3 %, so only same-process comparisons count): GCT2_AB=attr[,attr...]
for attr in [a for a in os.environ.get("GCT2_AB", "").split(",") if a]:
    for rnd in range(3):
        res = []
        for val in (False, True):
            setattr(eng, attr, val)
            for _ in range(3):
                eng.train_step(x)
            res.append(timed(lambda: eng.train_step(x), iters))
        print("A/B %-16s off %8.1f us   on %8.1f us" % (attr, res[0], res[1]))

if os.environ.get("GCT2_AB_ENV"):
    name = os.environ["GCT2_AB_ENV"]
    for rnd in range(4):
        res = []
        for val in ("1", "0"):
            os.environ[name] = val
            for _ in range(3):
                eng.train_step(x)
            res.append(timed(lambda: eng.train_step(x), iters))
        print("A/B env %-20s =1 %8.1f us   =0 %8.1f us" % (name, res[0], res[1]))
# in-process comparison of ctx tuning words (include/gct2.h gct2_ctx_set_tuning) on the whole step: GCT2_AB_TUNE=0,2,5,...
if os.environ.get("GCT2_AB_TUNE"):
    words = [int(v, 0) for v in os.environ["GCT2_AB_TUNE"].split(",")]
    for rnd in range(3):
        row = []
        for wd in words:
            eng.ctx.set_tuning(wd)
            for _ in range(3):
                eng.train_step(x)
            row.append("%#x: %7.1f" % (wd, timed(lambda: eng.train_step(x), iters)))
        print("tuning  " + "   ".join(row))
    eng.ctx.set_tuning(0)
# tuning word for the reverse pass only (the forward keeps the automatic choices): GCT2_AB_BWD_TUNE=1,2,...
if os.environ.get("GCT2_AB_BWD_TUNE"):
    words = [0] + [int(v, 0) for v in os.environ["GCT2_AB_BWD_TUNE"].split(",")]
    orig = eng.backward
    state = {"w": 0}
    def wrapped(*a, **k):
        eng.ctx.set_tuning(state["w"])
        try:
            return orig(*a, **k)
        finally:
            eng.ctx.set_tuning(0)
    eng.backward = wrapped
    for rnd in range(3):
        row = []
        for wd in words:
            state["w"] = wd
            for _ in range(3):
                eng.train_step(x)
            row.append("%#x: %7.1f" % (wd, timed(lambda: eng.train_step(x), iters)))
        print("reverse-pass tuning  " + "   ".join(row))
    eng.backward = orig
