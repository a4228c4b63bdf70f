"""in-process A/B of engine attributes on the whole train step (config 3): interleaved rounds in ONE process (boxes differ by +-3 %).
usage: python scripts/ab_engine.py "<assignments>" "<assignments>" ...      each argument = python statements run with `eng` in scope,
e.g.   python scripts/ab_engine.py "eng.defer_adam=False" "eng.defer_adam=True" "eng.defer_adam=True; eng.defer_layers=('dense','U0','U1')" """
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gan_class_transfer2_amd as g
from gan_class_transfer2_amd.engine import Topology, UNetEngine, BF16

variants = sys.argv[1:] or ["pass"]
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
    eng.flush_deferred()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


res = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        eng.flush_deferred()
        exec(v, {"eng": eng, "g": g, "torch": torch})
        for _ in range(3):
            eng.train_step(x)
        res[v].append(timed(iters))
for v in variants:
    a = np.array(res[v])
    print("%-70s step median %7.1f us   min %7.1f   (rounds: %s)" % (v[:70], np.median(a), a.min(), " ".join("%.0f" % t for t in a)))
