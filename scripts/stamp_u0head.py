"""phases of the fused UpShuffle_0 + head launch from in-kernel s_memrealtime stamps (diagnostic build only):
    make -C gan-class-transfer2_amd/csrc clean all EXTRA=-DGCT2_STAMP && python scripts/stamp_u0head.py && make -C gan-class-transfer2_amd/csrc clean all
The stamped build writes its stamps where the prediction would go (keep_pred) and no prediction: never ship it."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import gan_class_transfer2_amd as g
dev = torch.device("cuda", 0)
eng = g.UNetEngine(g.Topology(128, 512, 6), g.BF16, dev)
eng.keep_pred = True
B, S = 64, 128
x = (torch.randint(0, 256, (B, S, S, 3)).float() / 128 - 1).to(dev)
b = eng.buffers(B, S, S)
eng.sample_and_noise_into_r0(b, x)
eng.forward(b, head=False, stop_before_u0=True)
for _ in range(5):
    eng.u0_head_train(b, x)
torch.cuda.synchronize()
st = b.pred.reshape(-1)[: 1024 * 8 * 8 * 2].view(torch.int64).reshape(1024, 8, 8).cpu().numpy()[:, :, :6].astype(np.int64)
d = np.diff(st, axis=2) / 100.0          # s_memrealtime: 100 MHz -> us
names = ["K loop", "park acts + head weights", "row loop (8 rows)", "barrier", "reductions + partial row"]
for k, n in enumerate(names):
    print("%-28s median %6.2f us   p10 %6.2f   p90 %6.2f" % (n, np.median(d[:, :, k]), np.percentile(d[:, :, k], 10), np.percentile(d[:, :, k], 90)))
tot = (st[:, :, 5] - st[:, :, 0]) / 100.0
print("work-group life             median %6.2f us" % np.median(tot))
print("kernel span (first start .. last end) %6.2f us" % ((st[:, :, 5].max() - st[:, :, 0].min()) / 100.0))
