"""time single conv layers through the C ABI (diagnostic): python scripts/bench_layer.py [variant ...]
variant = the value handed to gct2_ctx_set_tuning (tile variant in bits 0-7, halo mode in bits 24-25)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gan_class_transfer2_amd as g
L = g._lib
dev = torch.device("cuda", 0)
ws = torch.empty(64 << 18, dtype=torch.float32, device=dev)
CTX = L.Context(); CTX.set_workspace(ws)
B = 64
SMALL = {
    "D3.fwd": ("conv_fwd", 16, 16, 512, 512), "D4.fwd": ("conv_fwd", 8, 8, 512, 512), "D5.fwd": ("conv_fwd", 4, 4, 512, 512),
    "U5.fwd": ("convT_fwd", 2, 2, 512, 512), "U4.fwd": ("convT_fwd", 4, 4, 1024, 512), "U3.fwd": ("convT_fwd", 8, 8, 1024, 512),
    "U3.dgrad": ("convT_dgrad", 8, 8, 1024, 512), "U4.dgrad": ("convT_dgrad", 4, 4, 1024, 512), "U5.dgrad": ("convT_dgrad", 2, 2, 512, 512),
    "D5.dgrad": ("conv_dgrad", 4, 4, 512, 512), "D4.dgrad": ("conv_dgrad", 8, 8, 512, 512), "D3.dgrad": ("conv_dgrad", 16, 16, 512, 512),
}
LAYERS = {  # name: (kind, H, W, Cin, Cout)  kind: conv fwd / convT fwd / convT dgrad (conv-form) / conv dgrad (convT-form)
    "D1.fwd": ("conv_fwd", 64, 64, 128, 256), "D2.fwd": ("conv_fwd", 32, 32, 256, 512),
    "U2.fwd": ("convT_fwd", 16, 16, 1024, 256), "U1.fwd": ("convT_fwd", 32, 32, 512, 128), "U0.fwd": ("convT_fwd", 64, 64, 256, 64),
    "D3.fwd": ("conv_fwd", 16, 16, 512, 512), "U3.fwd": ("convT_fwd", 8, 8, 1024, 512),
    "U3.dgrad": ("convT_dgrad", 8, 8, 1024, 512), "D3.dgrad": ("conv_dgrad", 16, 16, 512, 512),
    "U2.dgrad": ("convT_dgrad", 16, 16, 1024, 256), "D2.dgrad": ("conv_dgrad", 32, 32, 256, 512),
    "U1.dgrad": ("convT_dgrad", 32, 32, 512, 128), "U0.dgrad": ("convT_dgrad", 64, 64, 256, 64), "D1.dgrad": ("conv_dgrad", 64, 64, 128, 256),
}
def run(name, variant, iters=20):
    kind, H, W, Cin, Cout = LAYERS[name]
    CTX.set_tuning(variant)
    bf = torch.bfloat16
    s = torch.cuda.current_stream().cuda_stream
    if kind == "conv_fwd":
        x = torch.randn(B, H, W, Cin, device=dev).to(bf); w = (torch.randn(4, 4, Cin, Cout, device=dev) * .05).to(bf)
        y = torch.empty(B, H // 2, W // 2, Cout, device=dev, dtype=bf); b = torch.zeros(Cout, device=dev)
        f = lambda: L.call("gct2_conv4s2_fwd", CTX.handle, 1, x.data_ptr(), Cin, w.data_ptr(), b.data_ptr(), y.data_ptr(), Cout, B, H, W, Cin, Cout, 1, s)
        flops = 2.0 * B * (H // 2) * (W // 2) * Cout * 16 * Cin
    elif kind == "convT_fwd":
        x = torch.randn(B, H, W, Cin, device=dev).to(bf); w = (torch.randn(4, 4, Cout, Cin, device=dev) * .05).to(bf)
        y = torch.empty(B, 2 * H, 2 * W, Cout, device=dev, dtype=bf); b = torch.zeros(Cout, device=dev)
        f = lambda: L.call("gct2_convT4s2_fwd", CTX.handle, 1, x.data_ptr(), Cin, w.data_ptr(), b.data_ptr(), y.data_ptr(), Cout, B, H, W, Cin, Cout, 1, s)
        flops = 2.0 * B * 4 * H * W * Cout * 4 * Cin
    elif kind == "convT_dgrad":
        dz = torch.randn(B, 2 * H, 2 * W, Cout, device=dev).to(bf); w = (torch.randn(4, 4, Cout, Cin, device=dev) * .05).to(bf)
        act = torch.randn(B, H, W, Cin, device=dev).to(bf); dx = torch.empty_like(act)
        actp = None if os.environ.get("NOMASK") else act.data_ptr()     # NOMASK=1: what the ReLU-mask read costs
        f = lambda: L.call("gct2_convT4s2_dgrad", CTX.handle, 1, dz.data_ptr(), Cout, w.data_ptr(), actp, Cin, dx.data_ptr(), Cin, B, H, W, Cin, Cout, 0, None, 0, None, 0, s)
        flops = 2.0 * B * H * W * Cin * 16 * Cout
    else:
        dz = torch.randn(B, H // 2, W // 2, Cout, device=dev).to(bf); w = (torch.randn(4, 4, Cin, Cout, device=dev) * .05).to(bf)
        act = torch.randn(B, H, W, Cin, device=dev).to(bf); dx = torch.zeros_like(act)
        actp = None if os.environ.get("NOMASK") else act.data_ptr()
        f = lambda: L.call("gct2_conv4s2_dgrad", CTX.handle, 1, dz.data_ptr(), Cout, w.data_ptr(), actp, Cin, dx.data_ptr(), Cin, B, H, W, Cin, Cout, 1, None, 0, None, 0, s)
        flops = 2.0 * B * H * W * Cin * 4 * Cout
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    return us, flops / us / 1e6
if os.environ.get("LAYERSET") == "small":
    LAYERS = SMALL
if __name__ == "__main__":
    # variant bits 24-25: halo kernel mode (1 = never, 2 = wherever the shape allows), e.g. 33554432 = halo forced
    variants = [int(v) for v in sys.argv[1:]] or [0, 2, 5]
    print("layer      " + "".join(f"{'v%d/h%d/x%d' % (v & 255, (v >> 24) & 3, v >> 26):>16s}" for v in variants))
    for name in LAYERS:
        print(f"{name:10s} " + "".join("%8.1fus %4.0fTF" % run(name, v) for v in variants))
    CTX.set_tuning(0)
