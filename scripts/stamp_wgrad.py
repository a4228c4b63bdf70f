"""phases of the 256 x 256 weight-gradient launch from in-kernel s_memrealtime stamps (diagnostic build only):
    make -C gan-class-transfer2_amd/csrc clean all EXTRA=-DGCT2_STAMP && python scripts/stamp_wgrad.py [U0|U1] && make -C gan-class-transfer2_amd/csrc clean all"""
import sys, os
os.environ["GCT2_ALLOW_DIAGNOSTIC_BUILD"] = "1"       # the binding refuses a stamped library otherwise (gct2_build_flags)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import gan_class_transfer2_amd as g
L = g._lib
dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "U1"
H, W, Cin, Cout = {"U0": (64, 64, 256, 64), "U1": (32, 32, 512, 128)}[name]
B, bf = 64, torch.bfloat16
ws = torch.empty(64 << 18, dtype=torch.float32, device=dev)
ctx = L.Context(); ctx.set_workspace(ws)
x = torch.randn(B, H, W, Cin, device=dev).to(bf); dz = torch.randn(B, 2 * H, 2 * W, Cout, device=dev).to(bf)
dw = torch.zeros(4, 4, Cout, Cin, device=dev)
stamps = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
assert L.build_flags() & L.BUILD_STAMP, "build with: make -C gan-class-transfer2_amd/csrc clean all EXTRA=-DGCT2_STAMP"
ctx.set_stamp_buffer(stamps)
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    L.call("gct2_convT4s2_wgrad", ctx.handle, 1, x.data_ptr(), Cin, dz.data_ptr(), Cout, dw.data_ptr(), None, B, H, W, Cin, Cout, 0, None, s)
torch.cuda.synchronize()
st = stamps.cpu().numpy().reshape(-1, 4)
st = st[st[:, 0] != 0]
d = np.diff(st, axis=1) / 100.0
for k, n in enumerate(["setup", "reduction loop", "epilogue (slab store)"]):
    print("%-24s median %7.2f us   p10 %7.2f   p90 %7.2f" % (n, np.median(d[:, k]), np.percentile(d[:, k], 10), np.percentile(d[:, k], 90)))
print("waves %d; life median %.2f us; span %.1f us" % (len(st), np.median(st[:, 3] - st[:, 0]) / 100.0, (st[:, 3].max() - st[:, 0].min()) / 100.0))
