"""phases of a tap-GEMM input-gradient launch from in-kernel s_memrealtime stamps (diagnostic build only):
    make -C gan-class-transfer2_amd/csrc clean all EXTRA=-DGCT2_STAMP && python scripts/stamp_layer.py [layer] && make -C gan-class-transfer2_amd/csrc clean all
layer: U0 (default), U1, U2 = Conv2DTranspose input gradients of config 3 (conv-form 256 x 128 tiles)."""
import sys, os
os.environ["GCT2_ALLOW_DIAGNOSTIC_BUILD"] = "1"       # the binding refuses a stamped library otherwise (gct2_build_flags)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import gan_class_transfer2_amd as g
L = g._lib
dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "U0"
H, W, Cin, Cout = {"U0": (64, 64, 256, 64), "U1": (32, 32, 512, 128), "U2": (16, 16, 1024, 256)}[name]
B = 64
bf = torch.bfloat16
ws = torch.empty(64 << 18, dtype=torch.float32, device=dev)
ctx = L.Context(); ctx.set_workspace(ws)
dz = torch.randn(B, 2 * H, 2 * W, Cout, device=dev).to(bf); w = (torch.randn(4, 4, Cout, Cin, device=dev) * .05).to(bf)
act = torch.randn(B, H, W, Cin, device=dev).to(bf); dx = torch.empty_like(act)
stamps = torch.zeros(1 << 22, dtype=torch.int64, device=dev)
assert L.build_flags() & L.BUILD_STAMP, "build with: make -C gan-class-transfer2_amd/csrc clean all EXTRA=-DGCT2_STAMP"
ctx.set_stamp_buffer(stamps)
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    L.call("gct2_convT4s2_dgrad", ctx.handle, 1, dz.data_ptr(), Cout, w.data_ptr(), act.data_ptr(), Cin, dx.data_ptr(), Cin, B, H, W, Cin, Cout,
           0, None, 0, None, 0, s)
torch.cuda.synchronize()
st = stamps.cpu().numpy().reshape(-1, 8)
st = st[st[:, 0] != 0][:, :5]
d = np.diff(st, axis=1) / 100.0
for k, n in enumerate(["setup (descriptors)", "K loop", "epilogue (mask, store)", "stamp write"]):
    print("%-26s median %6.2f us   p10 %6.2f   p90 %6.2f" % (n, np.median(d[:, k]), np.percentile(d[:, k], 10), np.percentile(d[:, k], 90)))
print("waves stamped %d; wave life median %.2f us; span %.1f us" % (len(st), np.median(st[:, 4] - st[:, 0]) / 100.0, (st[:, 4].max() - st[:, 0].min()) / 100.0))
