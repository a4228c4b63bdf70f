"""in-process A/B of gct2_ctx_set_tuning words on the whole train step (config 3): interleaved rounds in ONE process, median and
minimum per value (boxes differ by +-3 %, separate invocations are not comparable: cdna_hip_programming.md rule 24).
usage: python scripts/ab_tuning.py <tuning> [<tuning> ...]     (integers; 0 = the defaults)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gan_class_transfer2_amd as g
from gan_class_transfer2_amd.engine import Topology, UNetEngine, BF16

vals = [int(v, 0) for v in sys.argv[1:]] or [0]
rounds, iters = int(os.environ.get("AB_ROUNDS", "5")), int(os.environ.get("AB_ITERS", "20"))
dev = torch.device("cuda", 0)
eng = UNetEngine(Topology(128, 512, 6), BF16, dev)
x = torch.rand(64, 128, 128, 3, device=dev) * 2 - 1
for _ in range(5):
    eng.train_step(x)


def timed(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        eng.train_step(x)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


res = {v: [] for v in vals}
for r in range(rounds):
    for v in vals:
        eng.ctx.set_tuning(v if v < (1 << 31) else v - (1 << 32))
        for _ in range(3):
            eng.train_step(x)
        res[v].append(timed(iters))
eng.ctx.set_tuning(0)
for v in vals:
    a = np.array(res[v])
    print("tuning 0x%08x: step median %7.1f us   min %7.1f   (rounds: %s)" % (v & 0xffffffff, np.median(a), a.min(), " ".join("%.0f" % t for t in a)))
