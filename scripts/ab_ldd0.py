"""in-process A/B of the dR_0 layout on the whole train step (config 3): ld 64 (128-byte pixels, r04) against ld 72 (R_0's stride,
r03), interleaved rounds in ONE process.  usage: python scripts/ab_ldd0.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gan_class_transfer2_amd as g
from gan_class_transfer2_amd.engine import Topology, UNetEngine, BF16

rounds, iters = int(os.environ.get("AB_ROUNDS", "5")), int(os.environ.get("AB_ITERS", "20"))
dev = torch.device("cuda", 0)
eng = UNetEngine(Topology(128, 512, 6), BF16, dev)
x = torch.rand(64, 128, 128, 3, device=dev) * 2 - 1
b = eng.buffers(64, 128, 128)
bufs = {ld: torch.zeros(64, 128, 128, ld, dtype=torch.bfloat16, device=dev) for ld in (64, 72)}


def timed(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        eng.train_step(x)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


res = {64: [], 72: []}
for r in range(rounds):
    for ld in (64, 72):
        b.ldd[0], b.dR[0] = ld, bufs[ld]
        for _ in range(3):
            eng.train_step(x)
        res[ld].append(timed(iters))
for ld in (64, 72):
    a = np.array(res[ld])
    print("dR_0 ld %d: step median %7.1f us   min %7.1f   (rounds: %s)" % (ld, np.median(a), a.min(), " ".join("%.0f" % t for t in a)))
