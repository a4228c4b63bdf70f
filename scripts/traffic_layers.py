"""run every conv layer of config 3 (3x128x128, batch 64, bf16) standalone, K launches each in a fixed order, so that a
`rocprofv3 --pmc` pass over this script can be attributed launch by launch (scripts/collect_traffic_layers.py).  Writes the launch
plan (layer, direction, algorithmic HBM bytes: every operand touched once) to gpurun_out/traffic_plan.json.
usage: python scripts/traffic_layers.py [K]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gan_class_transfer2_amd as g
L = g._lib
dev = torch.device("cuda", 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = 64
ws = torch.empty(64 << 18, dtype=torch.float32, device=dev); wws = torch.empty(64 << 18, dtype=torch.float32, device=dev)
CTX = L.Context(); CTX.set_workspace(ws); CTX.set_wgrad_workspace(wws)
bf = torch.bfloat16
s = torch.cuda.current_stream().cuda_stream
# (name, H, W of the layer INPUT, Cin, Cout) for DownShuffle (conv) and UpShuffle (convT) layers of the reference topology
DOWN = [("D1", 64, 64, 128, 256), ("D2", 32, 32, 256, 512), ("D3", 16, 16, 512, 512), ("D4", 8, 8, 512, 512), ("D5", 4, 4, 512, 512)]
UP = [("U5", 2, 2, 512, 512), ("U4", 4, 4, 1024, 512), ("U3", 8, 8, 1024, 512), ("U2", 16, 16, 1024, 256), ("U1", 32, 32, 512, 128), ("U0", 64, 64, 256, 64)]
plan = []
def rnd(*shape): return torch.randn(*shape, device=dev).to(bf)
for name, H, W, Cin, Cout in DOWN:
    x, w, b = rnd(B, H, W, Cin).clamp_min(0), (rnd(4, 4, Cin, Cout) * .05), torch.zeros(Cout, device=dev)
    y = torch.empty(B, H // 2, W // 2, Cout, device=dev, dtype=bf); dz = rnd(B, H // 2, W // 2, Cout); dx = torch.zeros_like(x)
    dw = torch.zeros(4, 4, Cin, Cout, device=dev)
    nin, nout, nw = x.numel() * 2, y.numel() * 2, w.numel() * 2
    for _ in range(K): L.call("gct2_conv4s2_fwd", CTX.handle, 1, x.data_ptr(), Cin, w.data_ptr(), b.data_ptr(), y.data_ptr(), Cout, B, H, W, Cin, Cout, 1, s)
    plan.append(dict(layer=name, dir="fwd", form="conv", alg_bytes=nin + nw + nout, flops=2.0 * B * (H // 2) * (W // 2) * Cout * 16 * Cin))
    for _ in range(K): L.call("gct2_conv4s2_dgrad", CTX.handle, 1, dz.data_ptr(), Cout, w.data_ptr(), x.data_ptr(), Cin, dx.data_ptr(), Cin, B, H, W, Cin, Cout, 1, None, 0, None, 0, s)
    plan.append(dict(layer=name, dir="dgrad", form="convT", alg_bytes=nout + nw + 3 * nin, flops=2.0 * B * (H // 2) * (W // 2) * Cout * 16 * Cin))   # dz, w, mask + read-modify-write of dx
    for _ in range(K): L.call("gct2_conv4s2_wgrad", CTX.handle, 1, x.data_ptr(), Cin, dz.data_ptr(), Cout, dw.data_ptr(), None, B, H, W, Cin, Cout, 0, None, s)
    plan.append(dict(layer=name, dir="wgrad", form="wgrad", alg_bytes=nin + nout + dw.numel() * 4, flops=2.0 * B * (H // 2) * (W // 2) * Cout * 16 * Cin))
    torch.cuda.synchronize()
for name, H, W, Cin, Cout in UP:
    x, w, b = rnd(B, H, W, Cin).clamp_min(0), (rnd(4, 4, Cout, Cin) * .05), torch.zeros(Cout, device=dev)
    y = torch.empty(B, 2 * H, 2 * W, Cout, device=dev, dtype=bf); dz = rnd(B, 2 * H, 2 * W, Cout); dx = torch.empty_like(x)
    dw = torch.zeros(4, 4, Cout, Cin, device=dev)
    nin, nout, nw = x.numel() * 2, y.numel() * 2, w.numel() * 2
    for _ in range(K): L.call("gct2_convT4s2_fwd", CTX.handle, 1, x.data_ptr(), Cin, w.data_ptr(), b.data_ptr(), y.data_ptr(), Cout, B, H, W, Cin, Cout, 1, s)
    plan.append(dict(layer=name, dir="fwd", form="convT", alg_bytes=nin + nw + nout, flops=2.0 * B * 4 * H * W * Cout * 4 * Cin))
    for _ in range(K): L.call("gct2_convT4s2_dgrad", CTX.handle, 1, dz.data_ptr(), Cout, w.data_ptr(), x.data_ptr(), Cin, dx.data_ptr(), Cin, B, H, W, Cin, Cout, 0, None, 0, None, 0, s)
    plan.append(dict(layer=name, dir="dgrad", form="conv", alg_bytes=nout + nw + 2 * nin, flops=2.0 * B * 4 * H * W * Cout * 4 * Cin))          # dz, w, mask, dx
    for _ in range(K): L.call("gct2_convT4s2_wgrad", CTX.handle, 1, x.data_ptr(), Cin, dz.data_ptr(), Cout, dw.data_ptr(), None, B, H, W, Cin, Cout, 0, None, s)
    plan.append(dict(layer=name, dir="wgrad", form="wgrad", alg_bytes=nin + nout + dw.numel() * 4, flops=2.0 * B * 4 * H * W * Cout * 4 * Cin))
    torch.cuda.synchronize()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(dict(K=K, plan=plan), open("gpurun_out/traffic_plan.json", "w"), indent=1)
print("launched", len(plan), "x", K)
