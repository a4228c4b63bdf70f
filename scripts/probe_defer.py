"""does the optimizer hide under the forward pass?  (diagnostic; the idea: defer the Adam step of dense + UpShuffle_0..5 - 26.5 M of
the 41.7 M parameters - from the reverse pass into the first third of the NEXT forward pass, which runs on one stream with the
HBM mostly idle.)  Measured r02: forward alone 857 us, that Adam alone 146 us, both on two streams 993 us - no overlap at all,
also with the Adam grid capped at 128 / 256 / 512 work-groups: the GEMM loops are bound by memory latency, and whatever loads the
memory system slows them by about its own time.  The deferral was not built."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import gan_class_transfer2_amd as g
from gan_class_transfer2_amd.engine import Topology, UNetEngine, BF16
dev = torch.device("cuda", 0)
eng = UNetEngine(Topology(128, 512, 6), BF16, dev)
x = torch.rand(64, 128, 128, 3, device=dev) * 2 - 1
b = eng.buffers(64, 128, 128)
for _ in range(3): eng.train_step(x)
A = eng.arena
lo, hi = A.layer_ranges["dense"][0], A.layer_ranges["U5"][1]
print("deferred range: %.1f M of %.1f M parameters" % ((hi - lo) / 1e6, A.total / 1e6))
side = torch.cuda.Stream()
def fwd():
    eng.begin_step()
    eng.sample_and_noise_into_r0(b, x, keep_eps=False)
    eng.forward(b, head=False, stop_before_u0=True)
    return eng.u0_head_train(b, x)
def adam_side():
    eng.apply_adam(lo, hi, stream=side.cuda_stream)
def both():
    side.wait_stream(torch.cuda.current_stream())
    adam_side()
    fwd()
    torch.cuda.current_stream().wait_stream(side)
def timed(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
def adam_only():
    side.wait_stream(torch.cuda.current_stream()); adam_side(); torch.cuda.current_stream().wait_stream(side)
print("forward alone          %7.1f us" % timed(fwd))
print("Adam(dense..U5) alone  %7.1f us" % timed(adam_only))
print("forward || Adam        %7.1f us" % timed(both))
