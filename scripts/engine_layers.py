"""the C-ABI calls of ONE train step of the planned engine, replayed call by call on the ENGINE'S OWN buffers: its pixel strides
(channel slices of the concat buffers, the dense dR_0), its ReLU bit planes, its fused bias-gradient targets, its automatic dispatch.
"Standalone" numbers from this script are therefore the step's launches, not dense look-alikes (scripts/bench_layer.py /
bench_wgrad.py time dense views: fine for kernel A/Bs, not for the step).

    python scripts/engine_layers.py                    # time every layer call: us and TFLOP/s (20 replays each)
    python scripts/engine_layers.py --pmc 5            # replay every call 5x in a fixed order, no timing: for `rocprofv3 --pmc ...`
                                                       # passes; writes the launch plan to gpurun_out/engine_plan.json
    options: --size 128 --batch 64 --dtype bf16 --tuning 0 --ldd0 72 (the r03 layout of dR_0) --only U0,D1 --iters 20
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gan_class_transfer2_amd as g
from gan_class_transfer2_amd import _lib, engine as engine_mod

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=128)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--tuning", type=lambda v: int(v, 0), default=0)
ap.add_argument("--ldd0", type=int, default=0)
ap.add_argument("--pmc", type=int, default=0)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--only", default="")
ap.add_argument("--no-planes", action="store_true")
ap.add_argument("--cu-mask", default="", help="replay on a stream created with hipExtStreamCreateWithCUMask: 'N' = the first N mask bits, "
                "'N/8' = N bits spread over every group of 8 (bit i set iff i %% 8 < N/32) - what a launch costs on a share of the chip")
ap.add_argument("--zeros", action="store_true", help="replay on all-zero activations, gradients and weights (the DVFS test of MI355X_MICROARCH.md: same instruction stream, no operand toggling)")
args = ap.parse_args()

dev = torch.device("cuda", 0)
dt = {"bf16": g.BF16, "f16": g.F16}[args.dtype]
eng = g.UNetEngine(g.Topology(128, 512, 6), dt, dev, loss_scaling=(args.dtype == "f16"))
eng.overlap = False
eng.use_plan = False                      # every call goes through the interpreter: this script records / replays them                       # one stream: every call is replayed alone
if args.no_planes:
    eng.relu_bits = False
if args.tuning:
    eng.ctx.set_tuning(args.tuning)
B, S = args.batch, args.size
x = (torch.randint(0, 256, (B, S, S, 3)).float() / 128 - 1).to(dev)
b = eng.buffers(B, S, S)
if args.ldd0:                             # A/B of the dR_0 layout
    b.ldd[0] = args.ldd0
    b.dR[0] = torch.zeros(B, S, S, args.ldd0, dtype=b.dR[0].dtype, device=dev)
for _ in range(2):
    eng.train_step(x, apply=False)
torch.cuda.synchronize()

# ---- record one step ----------------------------------------------------------------------------------------------------
rec = []
orig_call = _lib.call


def recorder(name, *a):
    rec.append((name, a))
    return orig_call(name, *a)


_lib.call = engine_mod.call = recorder
eng.train_step(x, apply=False)
_lib.call = engine_mod.call = orig_call
torch.cuda.synchronize()

LAYER = {"gct2_conv4s2_fwd", "gct2_convT4s2_fwd", "gct2_conv4s2_dgrad", "gct2_convT4s2_dgrad", "gct2_conv4s2_wgrad", "gct2_convT4s2_wgrad",
         "gct2_convT4s2_fwd_head_train"}
es = 2


def describe(name, a):
    """(label, form, flops, algorithmic HBM bytes) of one layer call; argument positions as in include/gct2.h"""
    lvl = lambda h_big: (S // h_big).bit_length() - 1
    if name == "gct2_conv4s2_fwd":
        _, _, _, ldx, _, _, _, ldy, Bn, H, W, Cin, Cout = a[:13]
        px = Bn * (H // 2) * (W // 2)
        return (f"D{lvl(H)}.fwd", "conv" if Cin > 4 else "rgb", 2.0 * px * Cout * 16 * Cin,
                Bn * H * W * Cin * es + 16 * Cin * Cout * es + px * Cout * es)
    if name == "gct2_convT4s2_fwd":
        Bn, H, W, Cin, Cout = a[8:13]
        return (f"U{lvl(2 * H)}.fwd", "convT", 2.0 * Bn * 4 * H * W * Cout * 4 * Cin, Bn * H * W * Cin * es + 16 * Cin * Cout * es + Bn * 4 * H * W * Cout * es)
    if name == "gct2_convT4s2_fwd_head_train":
        Bn, H, W, Cin, Cout = a[15:20]
        # x, w, target fp32, packed image, dR_0 written
        return ("U0.fwd+head", "convT", 2.0 * Bn * 4 * H * W * Cout * 4 * Cin + 3 * 2.0 * Bn * 4 * H * W * 3 * 67,
                Bn * H * W * Cin * es + 16 * Cin * Cout * es + Bn * 4 * H * W * (12 + 8 + Cout * es))
    if name == "gct2_conv4s2_dgrad":
        Bn, H, W, Cin, Cout, acc = a[9:15]
        px = Bn * H * W
        return (f"D{lvl(H)}.dgrad", "convT", 2.0 * Bn * (H // 2) * (W // 2) * Cout * 16 * Cin,
                Bn * (H // 2) * (W // 2) * Cout * es + 16 * Cin * Cout * es + px * Cin // 8 + px * Cin * es * (2 if acc else 1))
    if name == "gct2_convT4s2_dgrad":
        Bn, H, W, Cin, Cout, acc = a[9:15]
        px = Bn * H * W
        return (f"U{lvl(2 * H)}.dgrad", "conv", 2.0 * Bn * 4 * H * W * Cout * 4 * Cin,
                Bn * 4 * H * W * Cout * es + 16 * Cin * Cout * es + px * Cin // 8 + px * Cin * es * (2 if acc else 1))
    if name == "gct2_conv4s2_wgrad":
        Bn, H, W, Cin, Cout = a[8:13]
        px = Bn * (H // 2) * (W // 2)
        return (f"D{lvl(H)}.wgrad", "wgrad" if Cin > 4 else "rgb", 2.0 * px * Cout * 16 * Cin, Bn * H * W * Cin * es + px * Cout * es + 16 * Cin * Cout * 4)
    if name == "gct2_convT4s2_wgrad":
        Bn, H, W, Cin, Cout = a[8:13]
        return (f"U{lvl(2 * H)}.wgrad", "wgrad", 2.0 * Bn * 4 * H * W * Cout * 4 * Cin, Bn * H * W * Cin * es + Bn * 4 * H * W * Cout * es + 16 * Cin * Cout * 4)
    raise KeyError(name)


# a layer call + the one-shot plane registered in front of it
calls, pending = [], None
for name, a in rec:
    if name == "gct2_ctx_set_relu_bits":
        pending = a
    elif name in LAYER:
        calls.append((name, a, pending))
        pending = None
only = [s for s in args.only.split(",") if s]
if args.zeros:
    for v in vars(b).values():
        for t in (v if isinstance(v, (list, tuple)) else [v]):
            if torch.is_tensor(t) and t.is_floating_point():
                t.zero_()
    eng.arena._shadow.zero_()
    torch.cuda.synchronize()


masked = None
if args.cu_mask:
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    n = int(args.cu_mask.split("/")[0])
    spread = "/" in args.cu_mask
    words = (ctypes.c_uint32 * 8)()
    for i in range(256):
        if (i % 8 < n // 32) if spread else (i < n):
            words[i // 32] |= 1 << (i % 32)
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask failed ({rc})"
    masked = torch.cuda.ExternalStream(h.value, device=dev)
    print(f"# replayed on a CU-masked stream: {sum(bin(w).count('1') for w in words)} of 256 mask bits set ({'spread over every group of 8' if spread else 'the first bits'})")


def replay(name, a, plane):
    if plane is not None:
        orig_call("gct2_ctx_set_relu_bits", *plane)
    if masked is not None:
        a = tuple(a[:-1]) + (masked.cuda_stream,)
    orig_call(name, *a)


plan = []
if not args.pmc:
    print(f"# engine launches, {S}x{S} batch {B} {args.dtype}, tuning {args.tuning:#x}, ldd0 {b.ldd[0]}, planes {'off' if args.no_planes else 'on'}")
tot_us = 0.0
for name, a, plane in calls:
    label, form, flops, alg = describe(name, a)
    if only and not any(label.startswith(o) for o in only):
        continue
    if args.pmc:
        for _ in range(args.pmc):
            replay(name, a, plane)
        torch.cuda.synchronize()
        plan.append(dict(layer=label.split(".")[0], dir=label.split(".")[1], form=form, alg_bytes=alg, flops=flops, plane=plane is not None))
        continue
    for _ in range(3):
        replay(name, a, plane)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(masked) if masked is not None else e0.record()
    for _ in range(args.iters):
        replay(name, a, plane)
    e1.record(masked) if masked is not None else e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / args.iters
    tot_us += us
    print(f"{label:14s} {us:8.1f} us  {flops / us / 1e6:6.0f} TF/s  alg {alg / 1e6:7.1f} MB  {'plane' if plane is not None else ''}")
if args.pmc:
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(dict(K=args.pmc, plan=plan, size=S, batch=B, dtype=args.dtype), open("gpurun_out/engine_plan.json", "w"), indent=1)
    print("replayed", len(plan), "calls x", args.pmc)
else:
    print(f"sum {tot_us:8.1f} us")
