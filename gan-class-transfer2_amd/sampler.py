"""`log_sample` - the reference's sampler (train.py:323-496), default flags (predict_x, no ODE): one-shot denoising at
`test_step`, `steps` network evaluations that invert the example image into noise, the four edits of that noise
(avg-pool/upsample, roll, per-pixel codebook), and `steps` evaluations of reverse sampling on a batch of six.

Everything stays on the device: the state (x_theta, epsilon_theta, fake) is fp32 like the reference's, the pointwise steps
are library kernels (gct2_diffusion_mix / _update / gct2_noise_edits, include/gct2.h), the network evaluations reuse the
planned forward pass of UNetEngine at batch 1 and 6.  Returns the tensors the reference hands to tf.summary."""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch

from . import model as M
from ._lib import call
from .engine import UNetEngine


def _alpha(t: int, steps: int) -> float:
    tt = t / (steps + 1)                        # train.py:85-93, python floats as in the reference
    return (1 - tt) ** 2 * 0.25


class _Sampler:
    def __init__(self, eng: UNetEngine, steps: int, use_graph: bool = True):
        self.eng, self.steps = eng, steps
        # a network evaluation at batch 1 / 6 is ~20 short launches, and log_sample runs 401 of them back to back: launch-bound.
        # The forward pass of a buffer set is captured ONCE into a HIP graph (every pointer and shape is fixed per buffer set) and
        # replayed; the pointwise steps around it carry per-step scalars and stay ordinary launches.
        self.use_graph = use_graph
        # kept on the engine: the reference calls log_sample once per epoch, the buffer sets and arenas (hence the graphs) live on
        if not hasattr(eng, "_forward_graphs"):
            eng._forward_graphs = {}
        self._graphs: Dict[int, "torch.cuda.CUDAGraph"] = eng._forward_graphs

    def forward(self, b) -> None:
        g = self._graphs.get(id(b))
        if g is None and self.use_graph:
            self.eng.forward(b)                         # (also the first evaluation: whatever lazy set-up there is happens here)
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self.eng.forward(b)
                self._graphs[id(b)] = g
            except Exception:                           # capture refused (e.g. a profiler holding the stream): plain launches
                self.use_graph = False
            return
        if g is not None:
            g.replay()
        else:
            self.eng.forward(b)

    def _stream(self) -> int:
        return self.eng._stream()

    def mix(self, b, x: torch.Tensor, e: torch.Tensor, t: int, fake: torch.Tensor) -> None:
        """fake = sqrt(a_t) x + sqrt(1 - a_t) e, also written where the network reads its input (packed image + R_0 slice)."""
        eng = self.eng
        call("gct2_diffusion_mix", eng.dtype, x.data_ptr(), e.data_ptr(), _alpha(t, self.steps), fake.data_ptr(),
             b.img.data_ptr(), 4, eng._slice_ptr(b.R[0], eng.topo.fu(0)), b.ld[0], b.B * b.H * b.W, 3, self._stream())

    def update(self, b, fake: torch.Tensor, t: int, x: torch.Tensor, e: torch.Tensor) -> None:
        """x_theta = prediction; eps_theta = (fake - sqrt(a_t) x_theta) / sqrt(1 - a_t)   (train.py:394-398)."""
        call("gct2_diffusion_update", b.pred.data_ptr(), fake.data_ptr(), _alpha(t, self.steps), x.data_ptr(), e.data_ptr(),
             b.pred.numel(), self._stream())

    def step(self, b, x, e, fake, t) -> None:
        self.mix(b, x, e, t, fake)
        self.forward(b)                         # denoiser((fake, t)): t is ignored (train.py:208-210)
        self.update(b, fake, t, x, e)


def log_sample(denoiser: "M.Denoiser", example_image: torch.Tensor, example: torch.Tensor, dictionary: torch.Tensor,
               steps: Optional[int] = None, test_step: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """example_image [1,H,W,3] fp32 in [-1,1) (train.py:305); example [1,2,H,W,3] ~ N(0,1) (train.py:306);
    dictionary [H,W,2**bits_per_pixel,3] ~ N(0,1) (train.py:308-311); all on the HIP device."""
    steps = M.steps if steps is None else steps
    test_step = M.test_step if test_step is None else test_step
    eng = denoiser.ensure_engine()
    dev = eng.device
    f32 = lambda t: t.to(dev, torch.float32).contiguous()
    example_image, example, dictionary = f32(example_image), f32(example), f32(dictionary)
    _, H, W, _ = example_image.shape
    if example.shape != (1, 2, H, W, 3) or dictionary.shape[:2] != (H, W) or dictionary.shape[3] != 3:
        raise ValueError("log_sample: example must be [1,2,H,W,3] and dictionary [H,W,K,3] for an example image [1,H,W,3]")
    S = _Sampler(eng, steps)
    out: Dict[str, torch.Tensor] = {}
    image = example_image[0][None].contiguous()

    # ---- single-shot denoising at test_step (train.py:325-361)
    b1 = eng.buffers(1, H, W)
    fake = torch.empty_like(image)
    S.mix(b1, image, example[0, :1].contiguous(), test_step, fake)
    eng.forward(b1)
    out["denoised"] = b1.pred.clone()
    loss, dpred = torch.zeros(1, device=dev), torch.empty_like(b1.pred)
    call("gct2_mse_fwd_bwd", b1.pred.data_ptr(), image.data_ptr(), dpred.data_ptr(), loss.data_ptr(), b1.partials.data_ptr(),
         b1.pred.numel(), None, eng._stream())
    out["example_loss"] = loss.sqrt()           # reduce_mean((image - denoised)**2)**0.5   (train.py:357-361)

    # ---- forward diffusion: invert the example image (train.py:364-411)
    x_theta, eps_theta = image.clone(), image.clone()
    for t in range(1, steps + 1):
        S.step(b1, x_theta, eps_theta, fake, t)
    out["epsilon_theta"] = eps_theta.clone()

    # ---- the four edits + the two random noises -> batch of six (train.py:413-437)
    edits = torch.empty(4, H, W, 3, device=dev)
    call("gct2_noise_edits", eps_theta.data_ptr(), dictionary.data_ptr(), dictionary.shape[2], edits.data_ptr(), H, W, 3, eng._stream())
    start = torch.cat([example[0], edits], 0).contiguous()
    b6 = eng.buffers(6, H, W)
    x_theta, eps_theta, fake = start.clone(), start.clone(), torch.empty_like(start)

    # ---- backward diffusion (train.py:439-495)
    marks = ((steps, "step_1"), (steps // 4, "step_0.25"), (2 * steps // 4, "step_0.5"), (3 * steps // 4, "step_0.75"))
    for t in range(steps, 0, -1):
        S.step(b6, x_theta, eps_theta, fake, t)
        for tm, name in marks:                  # the reference's four `if`s (train.py:481-488): distinct tags, may share a t
            if t == tm:
                out[name] = x_theta.clone()
    out["fake"] = x_theta.clone()
    return out


def make_log_sample(denoiser: "M.Denoiser", example_image, example, dictionary, sink: Callable[[int, Dict[str, torch.Tensor]], None]):
    """callback with the reference's signature `log_sample(epochs, logs)` (train.py:323, 519-521); `sink(epoch, images)` stands
    for the TensorBoard summary writer."""
    def cb(epochs, logs):
        sink(epochs, log_sample(denoiser, example_image, example, dictionary))
    return cb
