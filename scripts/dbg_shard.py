import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gan_class_transfer2_amd as g
dev = torch.device("cuda", 0)
def rel(a, b): return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30))
size, B = 128, 64
topo = g.Topology(128, 512, 6)
gen = torch.Generator().manual_seed(3)
x = (torch.randint(0, 256, (B, size, size, 3), generator=gen).float() / 128 - 1).to(dev)
t_int = torch.randint(1, 201, (B,), generator=gen, dtype=torch.int32)
eps = torch.randn(B, size, size, 3, generator=gen)
def run(dt, sl, variant=0):
    g._lib.load().gct2_debug_tapgemm_variant(variant)
    eng = g.UNetEngine(topo, dt, dev)
    eng.arena.g.zero_()
    loss = eng.train_step(x[sl].contiguous(), t_int[sl].contiguous(), eps[sl].contiguous(), apply=False)
    torch.cuda.synchronize()
    b = eng.buffers(x[sl].shape[0], size, size)
    g._lib.load().gct2_debug_tapgemm_variant(0)
    return dict(loss=float(loss[0]), g=eng.arena.g.clone(), R=[r.float().clone() for r in b.R], dR=[r.float().clone() for r in b.dR], pred=b.pred.clone(), ranges=eng.arena.layer_ranges)
h = B // 2
ref = run(g.F32, slice(0, h))            # direct fp32 kernels on the first half: the reference
full = run(g.BF16, slice(0, B))
half = run(g.BF16, slice(0, h))
full2 = run(g.BF16, slice(0, B), variant=2)     # forced 128x128 tiles
fu = [64, 128, 256, 512, 512, 512]
print("loss ref/full/half/full(v2):", ref["loss"], full["loss"], half["loss"], full2["loss"])
for name, r in (("full[:h]", full), ("half", half), ("full v2[:h]", full2)):
    print(name, "R_i U-part vs fp32:", ["%.1e" % rel(r["R"][i][:h, ..., :fu[i]], ref["R"][i][..., :fu[i]]) for i in range(6)], "pred %.1e" % rel(r["pred"][:h], ref["pred"]))
print("full vs half R_i:", ["%.1e" % rel(full["R"][i][:h], half["R"][i]) for i in range(6)], "pred %.1e" % rel(full["pred"][:h], half["pred"]))
print("full vs full(v2) R_i:", ["%.1e" % rel(full["R"][i], full2["R"][i]) for i in range(6)])
sc = {"full[:h]": 2.0, "half": 1.0}
for name, r in (("full[:h]", full), ("half", half)):
    print(name, "dR_i vs fp32:", ["%.1e" % rel(sc[name] * r["dR"][i][:h], ref["dR"][i]) for i in range(6)])
