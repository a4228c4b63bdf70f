"""`log_sample` - the reference's sampler (train.py:323-496) with every objective switch it reads (train.py:29-32:
`predict_x`, `predict_scaled_epsilon`, `ordinary_differential_equation`; `prediction_weighting` is a training-only switch,
log_sample never reads it): one-shot denoising at `test_step` (at `steps / 2` in ODE mode, train.py:326-328), `steps` network
evaluations that invert the example image into noise, the four edits of that noise (avg-pool/upsample, roll, per-pixel
codebook), and `steps` evaluations of reverse sampling on a batch of six.

Everything stays on the device: the state (x_theta, epsilon_theta, fake) is fp32 like the reference's, the pointwise steps
are library kernels (gct2_diffusion_mix / _update / gct2_noise_edits, include/gct2.h).  The network evaluations reuse the
planned forward pass of UNetEngine at batch 1 and 6 (captured once into a HIP graph per buffer set), or - for the topology
variants of train.py:20, 26, 27 - VariantEngine.predict, layer by layer.  Returns the tensors the reference hands to tf.summary."""
from __future__ import annotations

import warnings
from typing import Callable, Dict, Optional, Tuple

import torch

from . import _lib
from . import model as M
from ._lib import SAMPLE_EPS, SAMPLE_ODE, SAMPLE_SCALED_EPS, SAMPLE_X, call
from .engine import TORCH_DTYPE, UNetEngine


def _alpha(t: float, steps: int) -> float:
    tt = t / (steps + 1)                        # train.py:85-93, python floats as in the reference
    return (1 - tt) ** 2 * 0.25


def sample_mode(predict_x: bool, predict_scaled_epsilon: bool, ordinary_differential_equation: bool) -> int:
    """the branch order of train.py:338-355, 382-413, 452-479: ODE first, then predict_x, then the two epsilon forms."""
    if ordinary_differential_equation:
        return SAMPLE_ODE
    if predict_x:
        return SAMPLE_X
    return SAMPLE_SCALED_EPS if predict_scaled_epsilon else SAMPLE_EPS


class _Sampler:
    """one network + the pointwise steps around it.  `net(B)` hands out the evaluation of a batch size: (input views for
    gct2_diffusion_mix, a function that runs the network and returns the fp32 prediction tensor)."""

    def __init__(self, eng, steps: int, mode: int, H: int, W: int, use_graph: bool = True):
        self.eng, self.steps, self.mode, self.H, self.W = eng, steps, mode, H, W
        self.planned = isinstance(eng, UNetEngine)
        # a network evaluation at batch 1 / 6 is ~20 short launches, and log_sample runs 401 of them back to back: launch-bound.
        # The planned forward pass of a buffer set is captured ONCE into a HIP graph (every pointer and shape is fixed per buffer
        # set) and replayed; the pointwise steps around it carry per-step scalars and stay ordinary launches.
        self.use_graph = use_graph and self.planned
        # kept on the engine: the reference calls log_sample once per epoch, the buffer sets and arenas (hence the graphs) live on.
        # A graph bakes in the workspace pointers and the tile choices of the engine's call context at capture time: the cache is
        # keyed on the context's version (bumped by every set_workspace / set_tuning / force_direct) and holds the buffer set
        # itself, so neither a re-tuned context nor a recycled id() can replay a stale graph.
        if self.planned and not hasattr(eng, "_forward_graphs"):
            eng._forward_graphs = {}
        self._inputs: Dict[int, torch.Tensor] = {}

    # ---- network evaluation ---------------------------------------------------------------------------------------------
    def _graph_for(self, b):
        eng = self.eng
        cache: Dict[Tuple[int, int], Tuple[object, object]] = eng._forward_graphs
        version = eng.ctx.version
        for key in [k for k in cache if k[1] != version]:       # the context changed: every captured graph is stale
            del cache[key]
        hit = cache.get((id(b), version))
        if hit is not None and hit[1] is b:
            return hit[0]
        eng.forward(b)                          # (also the first evaluation: whatever lazy set-up there is happens here)
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                eng.forward(b)
        except _lib.Gct2Error:                  # a launch was rejected: a real error, not a refused capture
            raise
        except RuntimeError as e:               # capture refused (e.g. a profiler holding the stream): plain launches from now on
            warnings.warn(f"log_sample: HIP graph capture of the forward pass was refused ({e}); using plain launches")
            self.use_graph = False
            return None
        cache[(id(b), version)] = (g, b)
        return None                             # this evaluation has already run (the warm-up call above)

    def evaluate(self, B: int) -> torch.Tensor:
        """denoiser((fake, t)) on the image gct2_diffusion_mix has just stored: t is ignored (train.py:208-210).  Returns the fp32
        prediction [B,H,W,3] (valid until the next evaluation)."""
        eng = self.eng
        if not self.planned:
            return eng.predict(self._inputs[B])
        b = eng.buffers(B, self.H, self.W)
        eng.flush_deferred()                     # optimizer launches the last train step held back: never inside a captured graph
        if not self.use_graph:
            eng.forward(b)
            return b.pred
        cache = eng._forward_graphs
        hit = cache.get((id(b), eng.ctx.version))
        if hit is not None and hit[1] is b:
            hit[0].replay()
        else:
            self._graph_for(b)
        return b.pred

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.eng.device).cuda_stream

    def mix(self, B: int, x: torch.Tensor, e: torch.Tensor, alpha: float, fake: torch.Tensor) -> None:
        """fake = sqrt(alpha) x + sqrt(1 - alpha) e, also written where the network reads its input."""
        eng = self.eng
        if self.planned:
            b = eng.buffers(B, self.H, self.W)
            call("gct2_diffusion_mix", eng.dtype, x.data_ptr(), e.data_ptr(), alpha, fake.data_ptr(), b.img.data_ptr(), 4,
                 eng._slice_ptr(b.R[0], eng.topo.fu(0)), b.ld[0], B * self.H * self.W, 3, self._stream())
        else:
            inp = self._inputs.get(B)
            if inp is None:
                inp = self._inputs[B] = torch.empty(B, self.H, self.W, 3, dtype=TORCH_DTYPE[eng.dtype], device=eng.device)
            call("gct2_diffusion_mix", eng.dtype, x.data_ptr(), e.data_ptr(), alpha, fake.data_ptr(), inp.data_ptr(), 3, None, 0,
                 B * self.H * self.W, 3, self._stream())

    def update(self, pred: torch.Tensor, fake: torch.Tensor, alpha: float, alpha_prev: float, x: torch.Tensor, e: Optional[torch.Tensor]) -> None:
        """train.py:382-413 / 452-479 by mode (gct2_diffusion_update)."""
        call("gct2_diffusion_update", self.mode, pred.data_ptr(), fake.data_ptr(), alpha, alpha_prev, x.data_ptr(),
             e.data_ptr() if e is not None else None, pred.numel(), self._stream())

    def step(self, B: int, x, e, fake, t: int) -> None:
        a = _alpha(t, self.steps)
        self.mix(B, x, e, a, fake)
        pred = self.evaluate(B)
        # ODE mode leaves epsilon_theta alone (train.py:382-392: only x_theta is assigned; `fake = ...` there is dead, the next
        # iteration recomputes it)
        self.update(pred, fake, a, _alpha(t - 1, self.steps), x, None if self.mode == SAMPLE_ODE else e)


def _resolve_switches(eng, predict_x, predict_scaled_epsilon, ordinary_differential_equation) -> int:
    """the sampler's objective switches: explicit arguments, else the module-level globals (train.py reads its globals when
    log_sample runs).  A network trained for another objective would sample garbage without any error: refuse a mismatch
    with the engine's own switches."""
    want = dict(predict_x=M.predict_x if predict_x is None else predict_x,
                predict_scaled_epsilon=M.predict_scaled_epsilon if predict_scaled_epsilon is None else predict_scaled_epsilon,
                ordinary_differential_equation=(M.ordinary_differential_equation if ordinary_differential_equation is None
                                                else ordinary_differential_equation))
    mode = sample_mode(**want)
    if all(hasattr(eng, k) for k in want):
        have = sample_mode(eng.predict_x, eng.predict_scaled_epsilon, eng.ordinary_differential_equation)
        if have != mode:
            raise ValueError(f"log_sample: objective switches {want} select sampler mode {mode}, but the engine was set up for "
                             f"mode {have} (predict_x={eng.predict_x}, predict_scaled_epsilon={eng.predict_scaled_epsilon}, "
                             f"ordinary_differential_equation={eng.ordinary_differential_equation}); train.py:29-32 are read by "
                             "Trainer.call and log_sample alike")
    return mode


def log_sample(denoiser: "M.Denoiser", example_image: torch.Tensor, example: torch.Tensor, dictionary: torch.Tensor,
               steps: Optional[int] = None, test_step: Optional[int] = None, predict_x: Optional[bool] = None,
               predict_scaled_epsilon: Optional[bool] = None, ordinary_differential_equation: Optional[bool] = None,
               use_graph: bool = True) -> Dict[str, torch.Tensor]:
    """example_image [1,H,W,3] fp32 in [-1,1) (train.py:305); example [1,2,H,W,3] ~ N(0,1) (train.py:306);
    dictionary [H,W,2**bits_per_pixel,3] ~ N(0,1) (train.py:308-311); all on the HIP device."""
    steps = M.steps if steps is None else steps
    test_step = M.test_step if test_step is None else test_step
    eng = denoiser.ensure_engine()
    mode = _resolve_switches(eng, predict_x, predict_scaled_epsilon, ordinary_differential_equation)
    dev = eng.device
    f32 = lambda t: t.to(dev, torch.float32).contiguous()
    example_image, example, dictionary = f32(example_image), f32(example), f32(dictionary)
    _, H, W, _ = example_image.shape
    if example.shape != (1, 2, H, W, 3) or dictionary.shape[:2] != (H, W) or dictionary.shape[3] != 3:
        raise ValueError("log_sample: example must be [1,2,H,W,3] and dictionary [H,W,K,3] for an example image [1,H,W,3]")
    S = _Sampler(eng, steps, mode, H, W, use_graph)
    out: Dict[str, torch.Tensor] = {}
    image = example_image[0][None].contiguous()

    # ---- single-shot denoising (train.py:325-361): at test_step; ODE mode mixes with alpha_dash(steps / 2) ** 0.5
    if mode == SAMPLE_ODE:
        factor = _alpha(steps / 2, steps) ** 0.5                  # train.py:326-328
        a_t, a_prev = _alpha(steps / 2, steps), _alpha(steps / 2 - 1, steps)
    else:
        factor = _alpha(test_step, steps)
        a_t, a_prev = factor, 0.0
    fake = torch.empty_like(image)
    S.mix(1, image, example[0, :1].contiguous(), factor, fake)
    # (never through the graph: the first evaluation of a buffer set is the capture's warm-up anyway)
    pred = eng.forward(eng.buffers(1, H, W)) if S.planned else S.evaluate(1)
    denoised, scratch = torch.empty_like(image), torch.empty_like(image)
    # denoised of train.py:338-355 is x_theta of the same update the loops use, at alpha = image_factor
    S.update(pred, fake, a_t, a_prev, denoised, scratch)
    out["denoised"] = denoised
    loss, dpred, partials = torch.zeros(1, device=dev), torch.empty_like(denoised), torch.zeros(1024, device=dev)
    call("gct2_mse_fwd_bwd", denoised.data_ptr(), image.data_ptr(), dpred.data_ptr(), loss.data_ptr(), partials.data_ptr(),
         denoised.numel(), None, S._stream())
    out["example_loss"] = loss.sqrt()           # reduce_mean((image - denoised)**2)**0.5   (train.py:357-361)

    # ---- forward diffusion: invert the example image (train.py:364-413)
    x_theta, eps_theta = image.clone(), image.clone()
    for t in range(1, steps + 1):
        S.step(1, x_theta, eps_theta, fake, t)
    out["epsilon_theta"] = eps_theta.clone()

    # ---- the four edits + the two random noises -> batch of six (train.py:415-437)
    edits = torch.empty(4, H, W, 3, device=dev)
    call("gct2_noise_edits", eps_theta.data_ptr(), dictionary.data_ptr(), dictionary.shape[2], edits.data_ptr(), H, W, 3, S._stream())
    start = torch.cat([example[0], edits], 0).contiguous()
    x_theta, eps_theta, fake = start.clone(), start.clone(), torch.empty_like(start)

    # ---- backward diffusion (train.py:439-496)
    marks = ((steps, "step_1"), (steps // 4, "step_0.25"), (2 * steps // 4, "step_0.5"), (3 * steps // 4, "step_0.75"))
    for t in range(steps, 0, -1):
        S.step(6, x_theta, eps_theta, fake, t)
        for tm, name in marks:                  # the reference's four `if`s (train.py:488-495): distinct tags, may share a t
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
