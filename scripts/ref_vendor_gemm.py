"""reference point, not product code: what the vendor GEMM (torch.mm -> hipBLASLt / rocBLAS) reaches on this box for plain bf16 GEMMs of
the shapes the big layers reduce to (same FLOPs, no gather, no epilogue).  usage: python scripts/ref_vendor_gemm.py"""
import torch
dev = torch.device("cuda", 0)
def timeit(f, iters=20):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
g = torch.Generator(device=dev); g.manual_seed(1)
def rnd(*s): return (torch.rand(*s, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
cases = [
    ("UpShuffle_2 forward as one GEMM     M 65536 N 1024 K 4096 (NN)", lambda: (rnd(65536, 4096), rnd(4096, 1024)), lambda a, b: torch.mm(a, b), 2 * 65536 * 1024 * 4096),
    ("UpShuffle_2 weight gradient         M 4096 N 1024 K 65536 (TN: both operands K-major)", lambda: (rnd(65536, 4096), rnd(65536, 1024)), lambda a, b: torch.mm(a.t(), b), 2 * 65536 * 1024 * 4096),
    ("UpShuffle_0 weight gradient         M 1024 N 256 K 262144 (TN)", lambda: (rnd(262144, 1024), rnd(262144, 256)), lambda a, b: torch.mm(a.t(), b), 2 * 262144 * 1024 * 256),
    ("UpShuffle_0 input gradient          M 262144 N 256 K 1024 (NN)", lambda: (rnd(262144, 1024), rnd(1024, 256)), lambda a, b: torch.mm(a, b), 2 * 262144 * 1024 * 256),
    ("DownShuffle_1 forward               M 65536 N 256 K 2048 (NN)", lambda: (rnd(65536, 2048), rnd(2048, 256)), lambda a, b: torch.mm(a, b), 2 * 65536 * 2048 * 256),
    ("square 8192^3 (NN)", lambda: (rnd(8192, 8192), rnd(8192, 8192)), lambda a, b: torch.mm(a, b), 2 * 8192 ** 3),
    ("square 8192^3 (NT)", lambda: (rnd(8192, 8192), rnd(8192, 8192)), lambda a, b: torch.mm(a, b.t()), 2 * 8192 ** 3),
]
for name, make, f, flops in cases:
    a, b = make()
    t = timeit(lambda: f(a, b))
    print("%-86s %8.1f us   %7.1f TFLOP/s" % (name, t, flops / t / 1e6), flush=True)
    del a, b
