import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import gan_class_transfer2_amd as gct
from gan_class_transfer2_amd import model as M

M.configure(size=128, batch_size=8, compute_dtype="bfloat16")
optimizer = gct.Adam(gct.WarmUp(2e-5, M.warm_up))
denoiser = gct.Denoiser()
trainer = gct.Trainer(denoiser)
rng = np.random.default_rng(0)
imgs = [rng.integers(0, 256, (160, 140, 3), dtype=np.uint8) for _ in range(20)]
def dataset():
    return iter(gct.ImageDataset(imgs, M.size, M.batch_size))
ds = dataset()
example = next(ds)[0]
loss = gct.identity(example, trainer(example))
trainer.compile(optimizer, gct.identity)
seen = {}
example_image = example[:1].clone()
ex = torch.randn(1, 2, M.size, M.size, 3, device=example.device)
dic = torch.randn(M.size, M.size, 8, 3, device=example.device)
M.configure(steps=4)     # keep the sampler short in this check (the reference uses 200)
cb = gct.make_log_sample(denoiser, example_image, ex, dic, lambda epoch, images: seen.update({epoch: sorted(images)}))
M.configure(steps=200)
trainer.fit(ds, steps_per_epoch=3, epochs=2, callbacks=[gct.LambdaCallback(on_epoch_begin=lambda e, l: None)], verbose=0)
out = gct.log_sample(denoiser, example_image, ex, dic, steps=4, test_step=2)
torch.cuda.synchronize()
print("loss", float(loss), "iterations", denoiser.engine.iterations, "sampler keys", sorted(out))
print("pred shape", tuple(denoiser((example, None)).shape))
