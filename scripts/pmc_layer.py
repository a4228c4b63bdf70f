"""a few launches of ONE layer through the C ABI, for rocprofv3 --pmc passes (diagnostic):
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE ... --output-format csv -d out -- python3 scripts/pmc_layer.py U1.wgrad 0 [iters]
layer names: those of scripts/bench_wgrad.py (X.wgrad) and scripts/bench_layer.py (X.fwd / X.dgrad); second argument: tuning word."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv, args = sys.argv[:1], sys.argv[1:]
name, tuning, iters = args[0], int(args[1], 0), int(args[2]) if len(args) > 2 else 5
if name.endswith(".wgrad"):
    import bench_wgrad as B
    B.run(name, tuning >> 16 if tuning >= (1 << 16) else tuning, iters)
else:
    import bench_layer as B
    B.run(name, tuning, iters)
import torch
torch.cuda.synchronize()
print("done", name, tuning, iters)
