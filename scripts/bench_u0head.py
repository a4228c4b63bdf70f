"""time UpShuffle_0's forward + head as one launch (gct2_convT4s2_fwd_head_train) and as two (diagnostic, config-3 shapes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gan_class_transfer2_amd as g
dev = torch.device("cuda", 0)
eng = g.UNetEngine(g.Topology(128, 512, 6), g.BF16, dev)
B, S = 64, 128
x = (torch.randint(0, 256, (B, S, S, 3)).float() / 128 - 1).to(dev)
b = eng.buffers(B, S, S)
eng.sample_and_noise_into_r0(b, x)
eng.forward(b, head=False)
def timeit(f, iters=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
t, A, s = eng.topo, eng.arena, eng._stream()
def u0_fwd():
    g._lib.call("gct2_convT4s2_fwd", eng.ctx.handle, eng.dtype, b.R[1].data_ptr(), b.ld[1], A.wptr("U0.w"), A.pptr("U0.b"), b.R[0].data_ptr(), b.ld[0],
                B, S // 2, S // 2, t.up_in(0), t.fu(0), 1, s)
print("U0 forward                 %7.1f us" % timeit(u0_fwd))
print("head (separate)            %7.1f us" % timeit(lambda: eng.head_train(b, x)))
print("U0 forward + head, fused   %7.1f us" % timeit(lambda: eng.u0_head_train(b, x)))
