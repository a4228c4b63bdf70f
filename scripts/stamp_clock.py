"""in-kernel clock of the K loops of the kernels that ship (diagnostic build only: `make -C gan-class-transfer2_amd/csrc stamp`).

MI355X_MICROARCH.md, "DVFS give-back" item 6: clock = d(s_memtime) / d(s_memrealtime) x 100 MHz, stamped once in front of and once
behind the loop after >= 2 s of back-to-back launches on random data, median over the waves.  The launches are the ENGINE'S OWN
(config 3: its strides, planes, fused bias sums, automatic dispatch), recorded from one train step and replayed - as in
scripts/engine_layers.py.

    python scripts/stamp_clock.py [--seconds 2.0] [--layers U0.wgrad,U1.fwd,...]
prints per layer: kernel (launch log), in-kernel clock (median, p10, p90 over waves), K-loop time per work-group, MFMA rate inside
the loop, and the whole launch for comparison.
"""
import argparse, os, sys
os.environ["GCT2_ALLOW_DIAGNOSTIC_BUILD"] = "1"
os.environ["GCT2_USE_STAMP_LIB"] = "phases" if "--phases" in sys.argv else "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gan_class_transfer2_amd as g
from gan_class_transfer2_amd import _lib, engine as engine_mod

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=2.0)
ap.add_argument("--tuning", type=lambda v: int(v, 0), default=0)
ap.add_argument("--phases", action="store_true", help="load libgct2_phases.so (make phases): per-stage phase stamps of wgrad256q_kernel")
ap.add_argument("--layers", default="U0.wgrad,U1.wgrad,D1.wgrad,U1.fwd,U2.fwd,D1.dgrad,D2.dgrad,U0.dgrad,U1.dgrad,U2.dgrad,D1.fwd,D2.fwd,U3.fwd,U3.dgrad,D3.dgrad,U3.wgrad")
args = ap.parse_args()
assert _lib.build_flags() & _lib.BUILD_STAMP, "needs the diagnostic build: make -C gan-class-transfer2_amd/csrc stamp"

dev = torch.device("cuda", 0)
B, S = 64, 128
eng = g.UNetEngine(g.Topology(128, 512, 6), g.BF16, dev)
eng.overlap = False
eng.use_plan = False                      # every call goes through the interpreter: this script records / replays them
if args.tuning:
    eng.ctx.set_tuning(args.tuning)
x = (torch.randint(0, 256, (B, S, S, 3)).float() / 128 - 1).to(dev)
for _ in range(2):
    eng.train_step(x, apply=False)                 # random data everywhere: activations and gradients of a real step
torch.cuda.synchronize()
rec, orig = [], _lib.call


def recorder(name, *a):
    rec.append((name, a))
    return orig(name, *a)


_lib.call = engine_mod.call = recorder
eng.train_step(x, apply=False)
_lib.call = engine_mod.call = orig
torch.cuda.synchronize()

LAYER = {"gct2_conv4s2_fwd": ("D", "fwd", 9, 0), "gct2_convT4s2_fwd": ("U", "fwd", 9, 1), "gct2_conv4s2_dgrad": ("D", "dgrad", 10, 0),
         "gct2_convT4s2_dgrad": ("U", "dgrad", 10, 1), "gct2_conv4s2_wgrad": ("D", "wgrad", 9, 0), "gct2_convT4s2_wgrad": ("U", "wgrad", 9, 1),
         "gct2_convT4s2_fwd_head_train": ("U", "fwd", 16, 1)}


def label(name, a):
    kind, what, hpos, small = LAYER[name]
    H = a[hpos] * (2 if small else 1)              # the layer's big-grid height
    return f"{kind}{(S // H).bit_length() - 1}.{what}"


def flops(name, a):
    if name == "gct2_convT4s2_fwd_head_train":
        Bn, H, W, Cin, Cout = a[15:20]
        return 2.0 * Bn * 4 * H * W * Cout * 4 * Cin
    off = 9 if "dgrad" in name else 8
    Bn, H, W, Cin, Cout = a[off:off + 5]
    return 2.0 * Bn * H * W * Cout * (16 * Cin if name.startswith("gct2_convT") else 4 * Cin)   # conv: (H/2)(W/2) * 16 Cin = H W 4 Cin; convT: H, W = the small grid


calls, pending = {}, None
for name, a in rec:
    if name == "gct2_ctx_set_relu_bits":
        pending = a
    elif name in LAYER:
        calls[label(name, a)] = (name, a, pending)
        pending = None

stamps = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
OFF = 1 << 19
print(f"# in-kernel clock of the K loops, config 3 (3x128x128, batch 64, bf16), >= {args.seconds} s of back-to-back launches before the stamped one")
print("# phases of a work-group's life from s_memrealtime: setup = entry -> K loop, loop, epilogue = K loop end -> stores drained; span = first entry -> last exit")
print("# idle CU-us (r06) = span x 256 CUs - (sum of every wave's life, entry -> drained stores) / (waves resident per CU): the CU-time of the launch in which a wave slot")
print("#   was empty - start skew, drain, rounds that do not fill the chip; 'idle %' = that / (span x 256).  The K-loop / setup / epilogue split of the busy part is in the phase columns.")
print("# layer       kernel                                      clock GHz (median p10 p90)   setup / K loop / epilogue us (median per wave)   waves   PFLOP/s in loop   span us   launch us   idle CU-us  idle %")
for lab in [s for s in args.layers.split(",") if s]:
    if lab not in calls:
        print(f"{lab:12s} (no such call)")
        continue
    name, a, plane = calls[lab]

    def launch():
        if plane is not None:
            orig("gct2_ctx_set_relu_bits", *plane)
        orig(name, *a)

    eng.ctx.set_stamp_buffer(None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n, t = 0, 0.0
    while t < args.seconds:                        # back-to-back: the chip settles at the clock this kernel sustains
        e0.record()
        for _ in range(200):
            launch()
        e1.record()
        torch.cuda.synchronize()
        t += e0.elapsed_time(e1) * 1e-3
        n += 200
    us_launch = t / n * 1e6
    stamps.zero_()
    eng.ctx.set_stamp_buffer(stamps)
    eng.ctx.log_launches(True)
    for _ in range(8):                             # no gap in front of the stamped launch either: the last of these is read
        launch()
    torch.cuda.synchronize()
    kern = [k for k in eng.ctx.read_launch_log() if not k.startswith(("relu_bits", "bias_queue"))][-1]
    eng.ctx.log_launches(False)
    eng.ctx.set_stamp_buffer(None)
    c = stamps[OFF:].cpu().numpy().reshape(-1, 8)
    c = c[(c[:, 1] != 0) & (c[:, 3] > c[:, 1]) & (c[:, 5] != 0)]
    if not len(c):
        print(f"{lab:12s} {kern:42s}  (this kernel carries no stamps)   launch {us_launch:7.1f} us")
        continue
    clk = (c[:, 2] - c[:, 0]) / (c[:, 3] - c[:, 1]) * 0.1          # cycles per 10 ns -> GHz
    setup_us, loop_us, epi_us = (c[:, 1] - c[:, 4]) / 100.0, (c[:, 3] - c[:, 1]) / 100.0, (c[:, 5] - c[:, 3]) / 100.0
    span = (c[:, 5].max() - c[:, 4].min()) / 100.0
    nwaves = len(c)
    # MFMA rate inside the loop: the launch's FLOPs, spread over the waves that ran, per median loop time, times the waves resident at once
    fl = flops(name, a)
    per_wave = fl / nwaves
    resident = min(nwaves, 256 * 16 if "256x128" in kern else 256 * 8)     # waves resident at once: two 8-wave groups per CU, else 8 waves per CU
    rate = per_wave / (np.median(loop_us) * 1e-6) * resident / 1e15
    life_us = float(((c[:, 5] - c[:, 4]) / 100.0).sum())
    idle_cu_us = span * 256.0 - life_us / (resident / 256.0 if resident >= 256 else 1.0)
    ph = stamps[1 << 18:(1 << 18) + 8 * (1 << 15)].cpu().numpy().reshape(-1, 8)
    grp = (np.arange(len(ph)) % 8) // 4            # wave group of the entry (8 waves per work-group: waves 0-3 / 4-7 share SIMDs pairwise)
    for gsel in (0, 1):                            # --phases (make phases): cycles per steady-state stage of wgrad256q_kernel, by phase (mean over the waves of a group;
                                                   # r03 order: reads and multiplies interleave, both are in the '32 MFMA' column)
        q = ph[(ph[:, 7] > 0) & (grp == gsel)]
        if len(q):
            per = q[:, :6] / q[:, 7:8]
            m = per.mean(axis=0)
            print(f"{lab:12s} waves {4 * gsel}-{4 * gsel + 3}: issue (front) {m[0]:5.0f} | reads {m[1]:5.0f} | 32 MFMA {m[2]:5.0f} | issue (behind) {m[3]:5.0f} | "
                  f"vmcnt wait {m[4]:4.0f} | barrier {m[5]:5.0f} | sum {m.sum():6.0f} cycles per stage ({int(q[:, 7].mean())} stages per wave)")
    print(f"{lab:12s} {kern:42s}  {np.median(clk):5.3f} {np.percentile(clk, 10):5.3f} {np.percentile(clk, 90):5.3f}      "
          f"{np.median(setup_us):6.2f} / {np.median(loop_us):7.2f} / {np.median(epi_us):6.2f}   {nwaves:6d}   {rate:6.3f}   {span:7.1f}   {us_launch:7.1f}   {idle_cu_us:9.0f}   {100.0 * idle_cu_us / (span * 256.0):5.1f}")
