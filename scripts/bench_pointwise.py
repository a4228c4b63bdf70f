"""time the HBM-bound kernels of the step through the C ABI (diagnostic): noise, Dense head."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gan_class_transfer2_amd as g
L = g._lib
dev = torch.device("cuda", 0)
ws = torch.empty(64 << 18, dtype=torch.float32, device=dev)
CTX = L.Context(); CTX.set_workspace(ws)
B, H, W = 64, 128, 128
M = B * H * W
bf = torch.bfloat16
s = torch.cuda.current_stream().cuda_stream
def timeit(f, iters=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
x = torch.rand(B, H, W, 3, device=dev) * 2 - 1
t = torch.randint(1, 201, (B,), dtype=torch.int32, device=dev)
r0 = torch.zeros(M, 72, dtype=bf, device=dev); img = torch.zeros(M, 4, dtype=bf, device=dev)
print("noise_image_rng (R0 slice + packed) %7.1f us" % timeit(lambda: L.call("gct2_noise_image_rng", 1, x.data_ptr(), t.data_ptr(), 1, 2, 12345, None,
      r0.data_ptr() + 2 * 64, 72, img.data_ptr(), 4, B, H * W, 3, 200, s)))
print("noise_image_rng (packed only)      %7.1f us" % timeit(lambda: L.call("gct2_noise_image_rng", 1, x.data_ptr(), t.data_ptr(), 1, 2, 12345, None,
      img.data_ptr(), 4, None, 0, B, H * W, 3, 200, s)))
r0[:, :64] = torch.randn(M, 64, device=dev).clamp_min(0).to(bf)
w = (torch.randn(67, 3, device=dev) * 0.3); b = torch.zeros(3, device=dev)
pred = torch.zeros(M, 3, device=dev); dx = torch.zeros(M, 72, dtype=bf, device=dev)
dw = torch.zeros(67, 3, device=dev); db = torch.zeros(3, device=dev); loss = torch.zeros(1, device=dev); part = torch.zeros(1024, device=dev)
dbx = torch.zeros(64, device=dev)
us = timeit(lambda: L.call("gct2_dense_head_train", CTX.handle, 1, r0.data_ptr(), 72, w.data_ptr(), b.data_ptr(), x.data_ptr(), pred.data_ptr(),
      dx.data_ptr(), 72, dw.data_ptr(), db.data_ptr(), loss.data_ptr(), part.data_ptr(), M, 67, 3, 64, None, dbx.data_ptr(), img.data_ptr(), 4, 0, s))
print("dense_head_train  %7.1f us  (%.2f TB/s of 151+134+12.6+12.6 MB)" % (us, (M * (144 + 128 + 12 + 12)) / us / 1e6))

# the 3-channel layer (DownShuffle_0) on the packed image: forward and weight gradient
img.copy_(torch.randn(M, 4, device=dev).to(bf)); img[:, 3] = 0
w0 = (torch.randn(4, 4, 3, 128, device=dev) * 0.1).to(bf); b0 = torch.zeros(128, device=dev)
y1 = torch.zeros(B, H // 2, W // 2, 256, dtype=bf, device=dev)
print("D0 forward        %7.1f us" % timeit(lambda: L.call("gct2_conv4s2_fwd", CTX.handle, 1, img.data_ptr(), 4, w0.data_ptr(), b0.data_ptr(),
      y1.data_ptr() + 2 * 128, 256, B, H, W, 3, 128, 1, s)))
dz1 = torch.randn(B, H // 2, W // 2, 256, device=dev).to(bf); dw0 = torch.zeros(4, 4, 3, 128, device=dev)
print("D0 weight grad    %7.1f us" % timeit(lambda: L.call("gct2_conv4s2_wgrad", CTX.handle, 1, img.data_ptr(), 4, dz1.data_ptr() + 2 * 128, 256,
      dw0.data_ptr(), None, B, H, W, 3, 128, 1, None, s)))
