"""time the stride-1 'same' convolution (Block's 3x3, train.py:123-143) through the C ABI: matrix-core forms vs the direct kernels
(diagnostic).  usage: python scripts/bench_s1.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gan_class_transfer2_amd as g
L = g._lib
dev = torch.device("cuda", 0)
ws = torch.empty(64 << 18, dtype=torch.float32, device=dev)
bf = torch.bfloat16
s = torch.cuda.current_stream().cuda_stream


def timed(f, iters):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for (B, H, W, Cin, Cout, KS) in [(64, 64, 64, 128, 128, 3), (64, 16, 16, 512, 512, 3), (64, 64, 64, 128, 64, 1)]:
    x = torch.randn(B, H, W, Cin, device=dev).to(bf); w = (torch.randn(KS, KS, Cin, Cout, device=dev) * .05).to(bf)
    y = torch.empty(B, H, W, Cout, device=dev, dtype=bf); b = torch.zeros(Cout, device=dev)
    dz = torch.randn(B, H, W, Cout, device=dev).to(bf); dx = torch.empty_like(x); dw = torch.empty(KS, KS, Cin, Cout, device=dev)
    flops = 2.0 * B * H * W * Cin * Cout * KS * KS
    row = f"B{B} {H}x{W} {Cin}->{Cout} k{KS}: "
    for direct in (False, True):
        ctx = L.Context(); ctx.set_workspace(ws); ctx.force_direct(direct)
        it = 3 if direct else 20
        t_f = timed(lambda: L.call("gct2_conv2d_s1_fwd", ctx.handle, 1, x.data_ptr(), Cin, w.data_ptr(), b.data_ptr(), y.data_ptr(), Cout, B, H, W, Cin, Cout, KS, 1, s), it)
        t_d = timed(lambda: L.call("gct2_conv2d_s1_dgrad", ctx.handle, 1, dz.data_ptr(), Cout, w.data_ptr(), x.data_ptr(), Cin, dx.data_ptr(), Cin, B, H, W, Cin, Cout, KS, 0, s), it)
        t_w = timed(lambda: L.call("gct2_conv2d_s1_wgrad", ctx.handle, 1, x.data_ptr(), Cin, dz.data_ptr(), Cout, dw.data_ptr(), None, B, H, W, Cin, Cout, KS, 0, s), it)
        row += ("direct " if direct else "mfma ") + "fwd %.0f us (%.0f TF) dgrad %.0f (%.0f) wgrad %.0f (%.0f)   " % (
            t_f, flops / t_f / 1e6, t_d, flops / t_d / 1e6, t_w, flops / t_w / 1e6)
    print(row)
