"""phases of the fused UpShuffle_0 + head launch from in-kernel s_memrealtime stamps (diagnostic build only):
    make -C gan-class-transfer2_amd/csrc clean all EXTRA=-DGCT2_STAMP && python scripts/stamp_u0head.py && make -C gan-class-transfer2_amd/csrc clean all
The stamps go to the buffer handed over with gct2_ctx_set_stamp_buffer; the product build has no stamps (gct2_build_flags() == 0)."""
import sys, os
os.environ["GCT2_ALLOW_DIAGNOSTIC_BUILD"] = "1"       # the binding refuses a stamped library otherwise
os.environ["GCT2_USE_STAMP_LIB"] = "1"               # libgct2_stamp.so (make -C gan-class-transfer2_amd/csrc stamp)
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import gan_class_transfer2_amd as g
dev = torch.device("cuda", 0)
eng = g.UNetEngine(g.Topology(128, 512, 6), g.BF16, dev)
assert g._lib.build_flags() & g._lib.BUILD_STAMP, "build with: make -C gan-class-transfer2_amd/csrc clean all EXTRA=-DGCT2_STAMP"
stamps = torch.zeros(1024 * 8 * 8, dtype=torch.int64, device=dev)
eng.ctx.set_stamp_buffer(stamps)
B, S = 64, 128
x = (torch.randint(0, 256, (B, S, S, 3)).float() / 128 - 1).to(dev)
b = eng.buffers(B, S, S)
eng.sample_and_noise_into_r0(b, x)
eng.forward(b, head=False, stop_before_u0=True)
for _ in range(5):
    eng.u0_head_train(b, x)
torch.cuda.synchronize()
st = stamps.reshape(1024, 8, 8).cpu().numpy()[:, :, :6].astype(np.int64)
d = np.diff(st, axis=2) / 100.0          # s_memrealtime: 100 MHz -> us
names = ["K loop", "operand images of the Dense kernel + barrier", "row loop (8 rows)", "barrier", "reductions + partial row"]
for k, n in enumerate(names):
    print("%-28s median %6.2f us   p10 %6.2f   p90 %6.2f" % (n, np.median(d[:, :, k]), np.percentile(d[:, :, k], 10), np.percentile(d[:, :, k], 90)))
tot = (st[:, :, 5] - st[:, :, 0]) / 100.0
print("work-group life             median %6.2f us" % np.median(tot))
print("kernel span (first start .. last end) %6.2f us" % ((st[:, :, 5].max() - st[:, :, 0].min()) / 100.0))
