"""time gan_class_transfer2_amd.log_sample at the reference's settings (steps 200: 401 network evaluations) - diagnostic."""
import sys, os, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gan_class_transfer2_amd as g
dev = torch.device("cuda", 0)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dt = {"bf16": g.BF16, "f32": g.F32, "f16": g.F16}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
eng = g.UNetEngine(g.Topology(128, 512, 6), dt, dev)
den = types.SimpleNamespace(ensure_engine=lambda: eng)
gen = torch.Generator().manual_seed(0)
img = (torch.randint(0, 256, (1, size, size, 3), generator=gen).float() / 128 - 1).to(dev)
ex = torch.randn(1, 2, size, size, 3, generator=gen).to(dev)
dic = torch.randn(size, size, 8, 3, generator=gen).to(dev)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = g.log_sample(den, img, ex, dic)
    torch.cuda.synchronize(); dt_s = time.perf_counter() - t0
    print(f"log_sample size {size}: {dt_s * 1e3:.1f} ms for 401 network evaluations ({dt_s / 401 * 1e6:.0f} us each); "
          f"example_loss {float(res['example_loss']):.4f}, fake finite: {bool(torch.isfinite(res['fake']).all())}")
