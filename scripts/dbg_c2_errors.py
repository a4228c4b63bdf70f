"""per-tensor gradient error of the bf16 step at BASELINE config 2 vs the fp32 CPU oracle (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import denoiser_oracle as O, torch_cross as T
import gan_class_transfer2_amd as g
dev = torch.device("cuda", 0)
cfg = O.OracleConfig(size=64, batch_size=32, octaves=6)
params = O.init_params(cfg, seed=1234, dtype=np.float32)
x, t_int, eps = O.synthetic_batch(cfg, seed=0, dtype=np.float32)
loss_ref, pred_ref, grads_ref = T.trainer_step(params, x, t_int, eps, cfg, dtype=torch.float32)
for dt, name in ((g.F32, "f32"), (g.BF16, "bf16")):
    eng = g.UNetEngine(g.Topology(128, 512, 6), dt, dev)
    eng.set_params(params)
    loss = eng.train_step(torch.tensor(x, device=dev), torch.tensor(t_int), torch.tensor(eps), apply=False)
    torch.cuda.synchronize()
    gr = eng.get_grads()
    print(name, "loss", float(loss[0]), loss_ref)
    for k in eng.topo.backward_order():
        e = np.linalg.norm(gr[k] - grads_ref[k]) / (np.linalg.norm(grads_ref[k]) + 1e-30)
        print(f"  {k:8s} rel_l2 {e:.3e}  |g| {np.linalg.norm(grads_ref[k]):.3e}")
