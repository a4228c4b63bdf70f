"""does the optimizer hide under the BOTTLENECK window of the forward pass?  (diagnostic, r03)
r02's probe_defer.py started the deferred Adam step at the beginning of the forward pass, beside the big MFMA-bound layers: no
overlap (857 + 146 -> 993 us).  Here it starts when DownShuffle_3's forward is enqueued - the 4x4 ... 16x16 levels (D3 .. U3)
are latency-bound launches with few work-groups - for the layers whose weights the forward pass needs last (dense, UpShuffle_0..3)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import gan_class_transfer2_amd as g
from gan_class_transfer2_amd import engine as E
from gan_class_transfer2_amd.engine import Topology, UNetEngine, BF16
dev = torch.device("cuda", 0)
eng = UNetEngine(Topology(128, 512, 6), BF16, dev)
x = torch.rand(64, 128, 128, 3, device=dev) * 2 - 1
b = eng.buffers(64, 128, 128)
for _ in range(3): eng.train_step(x)
A = eng.arena
side = torch.cuda.Stream()
orig_call = E.call
state = {"n": 0, "at": None, "range": None}
def hooked(name, *args):
    orig_call(name, *args)
    if state["at"] is not None and name == "gct2_conv4s2_fwd":
        state["n"] += 1
        if state["n"] == state["at"]:
            lo, hi = state["range"]
            side.wait_stream(torch.cuda.current_stream())
            eng.apply_adam(lo, hi, stream=side.cuda_stream)
E.call = hooked
def fwd():
    state["n"] = 0
    eng.begin_step()
    eng.sample_and_noise_into_r0(b, x, keep_eps=False)
    eng.forward(b, head=False, stop_before_u0=True)
    r = eng.u0_head_train(b, x)
    torch.cuda.current_stream().wait_stream(side)
    return r
def timed(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
base = timed(fwd)
print("forward alone                      %7.1f us" % base)
for last in ("U1", "U2", "U3"):
    lo, hi = A.layer_ranges["dense"][0], A.layer_ranges[last][1]
    state["range"] = (lo, hi)
    state["at"] = None
    def adam_only():
        side.wait_stream(torch.cuda.current_stream()); eng.apply_adam(lo, hi, stream=side.cuda_stream); torch.cuda.current_stream().wait_stream(side)
    t_adam = timed(adam_only)
    for at in (1, 4, 5, 6):
        state["at"] = at
        t = timed(fwd)
        print("Adam(dense..%s, %4.1f M, alone %6.1f us) started behind conv fwd #%d: forward %7.1f us (+%5.1f)" %
              (last, (hi - lo) / 1e6, t_adam, at, t, t - base))
    state["at"] = None
