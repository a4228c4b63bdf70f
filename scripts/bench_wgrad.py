"""time single weight-gradient layers through the C ABI (diagnostic).
usage: python scripts/bench_wgrad.py [code ...]   code = weight-gradient tile variant (bits 16-23 of gct2_ctx_set_tuning)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gan_class_transfer2_amd as g
L = g._lib
dev = torch.device("cuda", 0)
B = 64
ws = torch.empty(64 << 18, dtype=torch.float32, device=dev)
CTX = L.Context(); CTX.set_workspace(ws)   # code 7 = keep atomics although a workspace is registered
LAYERS = {  # name: (kind, H, W, Cin, Cout) with H, W = spatial size of the layer INPUT
    "U0.wgrad": ("convT", 64, 64, 256, 64), "U1.wgrad": ("convT", 32, 32, 512, 128), "U2.wgrad": ("convT", 16, 16, 1024, 256),
    "U3.wgrad": ("convT", 8, 8, 1024, 512), "D1.wgrad": ("conv", 64, 64, 128, 256), "D2.wgrad": ("conv", 32, 32, 256, 512),
    "D3.wgrad": ("conv", 16, 16, 512, 512), "D4.wgrad": ("conv", 8, 8, 512, 512), "D5.wgrad": ("conv", 4, 4, 512, 512),
    "U5.wgrad": ("convT", 2, 2, 512, 512), "U4.wgrad": ("convT", 4, 4, 1024, 512),
}
def run(name, code, iters=20):
    kind, H, W, Cin, Cout = LAYERS[name]
    CTX.set_tuning((code << 16) | int(os.environ.get("TUNE_OR", "0"), 0))   # TUNE_OR: bits outside 16-23 (e.g. the stagger units)
    bf = torch.bfloat16
    s = torch.cuda.current_stream().cuda_stream
    if kind == "conv":
        x = torch.randn(B, H, W, Cin, device=dev).to(bf); dz = torch.randn(B, H // 2, W // 2, Cout, device=dev).to(bf)
        dw = torch.zeros(4, 4, Cin, Cout, device=dev)
        f = lambda: L.call("gct2_conv4s2_wgrad", CTX.handle, 1, x.data_ptr(), Cin, dz.data_ptr(), Cout, dw.data_ptr(), None, B, H, W, Cin, Cout, 0, None, s)
        flops = 2.0 * B * (H // 2) * (W // 2) * Cout * 16 * Cin
    else:
        x = torch.randn(B, H, W, Cin, device=dev).to(bf); dz = torch.randn(B, 2 * H, 2 * W, Cout, device=dev).to(bf)
        dw = torch.zeros(4, 4, Cout, Cin, device=dev)
        f = lambda: L.call("gct2_convT4s2_wgrad", CTX.handle, 1, x.data_ptr(), Cin, dz.data_ptr(), Cout, dw.data_ptr(), None, B, H, W, Cin, Cout, 0, None, s)
        flops = 2.0 * B * H * W * Cout * 16 * Cin
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    return us, flops / us / 1e6
if __name__ == "__main__":
    codes = [int(v) for v in sys.argv[1:]] or [0, 7]
    print("layer      " + "".join(f"{'v%d' % v:>16s}" for v in codes))
    for name in LAYERS:
        print(f"{name:10s} " + "".join("%8.1fus %4.0fTF" % run(name, v) for v in codes))
    CTX.set_tuning(0)
