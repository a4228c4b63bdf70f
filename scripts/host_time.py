import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import gan_class_transfer2_amd as g
dev = torch.device("cuda", 0)
eng = g.UNetEngine(g.Topology(128, 512, 6), g.BF16, dev)
x = (torch.randint(0, 256, (64, 128, 128, 3)).float() / 128 - 1).to(dev)
for _ in range(5): eng.train_step(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): eng.train_step(x)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/20:.3f} ms/step, total {1e3*(t2-t0)/20:.3f} ms/step")
