"""Host-side mirror of the reference's Python model/train-loop surface (train.py:17-283, 498-523).

Same names, constructor arguments and defaults as /root/reference/train.py so a user of that script finds
`WarmUp`, `alpha_dash`, `Residual`, `Block`, `UpShuffle`, `DownShuffle`, `identity`, `Denoiser`, `Trainer`,
`compile`/`fit` and the module-level hyper-parameters here; tensors are NHWC torch tensors on the HIP device
and all arithmetic runs in libgct2.so (include/gct2.h).  There is no TensorFlow and no CPU fallback.

Module-level globals below ARE the config API, like the reference's (train.py:17-36); `configure(...)` sets
several at once.  `compute_dtype` is the one MI355X-specific knob.
"""
from __future__ import annotations

import sys
from typing import Callable, Dict, Iterable, List, Optional, Sequence as Seq

import torch

from . import _lib
from ._lib import BF16, F16, F32, call
from .engine import TORCH_DTYPE, Topology, UNetEngine

# ---- train.py:17-36 ---------------------------------------------------------------------------------
size = 256
pixel_size = 128 * 1
max_size = 512 * 1
block_depth = 0
octaves = 6  # bottleneck = 4x4

batch_size = 1
steps = 200

residual = False
concat = True

predict_x = True  # as opposed to epsilon
predict_scaled_epsilon = False
prediction_weighting = False
ordinary_differential_equation = False

mixed_precision = False

warm_up = 2_000

# MI355X knob: None -> float32, or float16 when mixed_precision (train.py:38); "bfloat16" selects the
# bf16-operand / fp32-accumulate MFMA path that BASELINE.json's metric is quoted on.
compute_dtype: Optional[str] = None

_DTYPES = {"float32": F32, "bfloat16": BF16, "float16": F16}


def configure(**kw) -> None:
    """set module-level hyper-parameters (the reference edits them in source, train.py:5-36)."""
    mod = sys.modules[__name__]
    for k, v in kw.items():
        if not hasattr(mod, k):
            raise AttributeError(f"unknown hyper-parameter {k!r}")
        setattr(mod, k, v)


def objective_switches() -> Dict[str, bool]:
    """the four objective globals of train.py:29-32 as they stand NOW (Trainer.call and log_sample read them at call time)."""
    return dict(predict_x=bool(predict_x), predict_scaled_epsilon=bool(predict_scaled_epsilon),
                prediction_weighting=bool(prediction_weighting),
                ordinary_differential_equation=bool(ordinary_differential_equation))


def preferred_dtype_code() -> int:
    """train.py:38: preferred_type = float16 if mixed_precision else float32 (+ the bf16 knob)."""
    if compute_dtype is not None:
        return _DTYPES[compute_dtype]
    return F16 if mixed_precision else F32


# ---- optimizer pieces (train.py:47-83) -----------------------------------------------------------------
class WarmUp:
    """train.py:50-65: lr(step) = base*(step+1)/(warmup_steps+1) while step < warmup_steps, else base."""

    def __init__(self, base, warmup_steps):
        self.base = base
        self.warmup_steps = warmup_steps

    def __call__(self, step):
        import numpy as np
        if step < self.warmup_steps:
            return float(np.float32(self.base) * np.float32(step + 1) / np.float32(self.warmup_steps + 1))
        return float(np.float32(self.base))


class Adam:
    """tf.keras.optimizers.Adam hyper-parameters (train.py:75); the update itself is
    gct2_adam_keras_multi (epsilon added to sqrt(v))."""

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = learning_rate, beta_1, beta_2, epsilon
        self.loss_scaling = False
        self._engine = None            # bound by Trainer.compile / train_step: the step counter lives with the engine

    @property
    def iterations(self) -> int:
        """optimizer.iterations [TF]: applied steps (a step skipped by the loss-scale logic does not count)."""
        return 0 if self._engine is None else self._engine.iterations

    def lr(self, step: int) -> float:
        return self.learning_rate(step) if callable(self.learning_rate) else float(self.learning_rate)


class LossScaleOptimizer:
    """tf.keras.mixed_precision.LossScaleOptimizer (train.py:82-83): dynamic loss scaling."""

    def __init__(self, inner_optimizer: Adam):
        self.inner = inner_optimizer
        self.inner.loss_scaling = True

    def __getattr__(self, k):
        return getattr(self.inner, k)


def default_optimizer():
    """train.py:75,82-83"""
    opt = Adam(WarmUp(2e-5, warm_up))
    return LossScaleOptimizer(opt) if mixed_precision else opt


def alpha_dash(t):
    """train.py:85-93"""
    t = t / (steps + 1)
    return (1 - t) ** 2 * 0.25


test_step = 25  # train.py:95


def identity(y_true, y_pred):
    """train.py:171-173: reduce_mean(y_pred), y_true ignored."""
    return torch.mean(y_pred)


# ---- eager layers (train.py:97-169) ---------------------------------------------------------------------
def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _code_of(t: torch.Tensor) -> int:
    return {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}[t.dtype]


def _as_compute(x: torch.Tensor, code: int) -> torch.Tensor:
    if not x.is_cuda:
        raise _lib.Gct2Error("layers run on the HIP device only (there is no CPU path)")
    return x.to(TORCH_DTYPE[code]).contiguous()


class Layer:
    def build(self, input_shape):
        pass

    def call(self, input):
        raise NotImplementedError

    def __call__(self, input):
        if not getattr(self, "_built", False):
            self.build(tuple(input[0].shape) if isinstance(input, (tuple, list)) else tuple(input.shape))
            self._built = True
        return self.call(input)


class Sequential(Layer):
    """the subset of tf.keras.Sequential the reference uses (train.py:130,183,191)."""

    def __init__(self, layers: Seq[Layer]):
        self.layers = list(layers)

    def call(self, input):
        for layer in self.layers:
            input = layer(input)
        return input


class Residual(Layer):
    """train.py:97-121: residual (input + Dense(module(input))), concat (the default) or plain module."""

    def __init__(self, module, highway=lambda x: x):
        self.module = module
        self.highway = highway
        self.dense = None

    def build(self, input_shape):
        if residual:                                             # train.py:104-108
            self.dense = Dense(input_shape[-1], use_bias=False)

    def call(self, input):
        if residual:                                             # train.py:111-112
            out = input.contiguous().clone()
            proj = self.dense(self.module(input)).to(out.dtype)
            C = out.shape[-1]
            call("gct2_add", _code_of(out), out.data_ptr(), C, proj.data_ptr(), C, out.numel() // C, C, _stream(out))
            return out
        if concat:
            return torch.cat([self.module(input).to(input.dtype), self.highway(input)], -1)
        return self.module(input)


class Block(Layer):
    """train.py:123-143: block_depth x [Conv2D(filters, 3, 1, 'same', relu)]; identity at block_depth = 0."""

    def __init__(self, filters):
        self.filters = filters
        self.convs = [Conv3x3(filters) for _ in range(block_depth)]

    def call(self, input):
        for conv in self.convs:
            input = conv(input)
        return input


class _ConvLayer(Layer):
    """shared storage for DownShuffle / UpShuffle: kernel/bias are fp32 master views (possibly into a
    Denoiser's parameter arena) plus a compute-dtype operand copy."""

    def __init__(self, filters):
        self.filters = filters
        self.kernel: Optional[torch.Tensor] = None      # fp32, Keras layout
        self.bias: Optional[torch.Tensor] = None
        self._operand = None                             # callable -> device pointer of compute-dtype kernel
        self.dtype_code = preferred_dtype_code()

    def _kernel_shape(self, cin):
        raise NotImplementedError

    def build(self, input_shape):
        if self.kernel is not None:
            return
        import math
        cin = input_shape[-1]
        shp = self._kernel_shape(cin)
        lim = math.sqrt(6.0 / (16 * shp[2] + 16 * shp[3]))
        dev = torch.device("cuda", torch.cuda.current_device())
        self.kernel = ((torch.rand(shp) * 2 - 1) * lim).to(dev)
        self.bias = torch.zeros(self.filters, device=dev)

    def _operand_tensor(self) -> torch.Tensor:
        return self.kernel if self.dtype_code == F32 else self.kernel.to(TORCH_DTYPE[self.dtype_code])


class Conv3x3(_ConvLayer):
    """the Conv2D(filters, 3, 1, 'same', relu) of Block (train.py:131-139)."""

    def _kernel_shape(self, cin):
        return (3, 3, cin, self.filters)

    def build(self, input_shape):
        if self.kernel is not None:
            return
        import math
        cin = input_shape[-1]
        lim = math.sqrt(6.0 / (9 * cin + 9 * self.filters))
        dev = torch.device("cuda", torch.cuda.current_device())
        self.kernel = ((torch.rand(3, 3, cin, self.filters) * 2 - 1) * lim).to(dev)
        self.bias = torch.zeros(self.filters, device=dev)

    def call(self, input):
        x = _as_compute(input, self.dtype_code)
        B, H, W, C = x.shape
        y = torch.empty(B, H, W, self.filters, dtype=x.dtype, device=x.device)
        w = self._operand_tensor()
        call("gct2_conv2d_s1_fwd", None, self.dtype_code, x.data_ptr(), C, w.data_ptr(), self.bias.data_ptr(), y.data_ptr(), self.filters,
             B, H, W, C, self.filters, 3, 1, _stream(x))
        return y


class UpShuffle(_ConvLayer):
    """train.py:145-156: Conv2DTranspose(filters, 4, 2, 'same', relu)."""

    def _kernel_shape(self, cin):
        return (4, 4, self.filters, cin)

    def call(self, input):
        x = _as_compute(input, self.dtype_code)
        B, H, W, C = x.shape
        y = torch.empty(B, 2 * H, 2 * W, self.filters, dtype=x.dtype, device=x.device)
        w = self._operand_tensor()
        call("gct2_convT4s2_fwd", None, self.dtype_code, x.data_ptr(), C, w.data_ptr(), self.bias.data_ptr(), y.data_ptr(),
             self.filters, B, H, W, C, self.filters, 1, _stream(x))
        return y


class DownShuffle(_ConvLayer):
    """train.py:158-169: Conv2D(filters, 4, 2, 'same', relu)."""

    def _kernel_shape(self, cin):
        return (4, 4, cin, self.filters)

    def call(self, input):
        x = _as_compute(input, self.dtype_code)
        B, H, W, C = x.shape
        if H % 2 or W % 2:
            raise ValueError(f"DownShuffle needs even spatial dims, got {H}x{W}")
        y = torch.empty(B, H // 2, W // 2, self.filters, dtype=x.dtype, device=x.device)
        w = self._operand_tensor()
        call("gct2_conv4s2_fwd", None, self.dtype_code, x.data_ptr(), C, w.data_ptr(), self.bias.data_ptr(), y.data_ptr(),
             self.filters, B, H, W, C, self.filters, 1, _stream(x))
        return y


class Dense(Layer):
    """tf.keras.layers.Dense(units) on a rank-4 input: the Dense(3) head (train.py:198-202; fp32 output for the fp32 loss) and,
    with use_bias=False, the projection of Residual's residual=True mode (train.py:106; a 1 x 1 convolution in the compute dtype)."""

    def __init__(self, units, use_bias=True):
        self.units = units
        self.use_bias = use_bias
        self.kernel = None
        self.bias = None
        self.dtype_code = preferred_dtype_code()

    def build(self, input_shape):
        if self.kernel is not None:
            return
        import math
        cin = input_shape[-1]
        lim = math.sqrt(6.0 / (cin + self.units))
        dev = torch.device("cuda", torch.cuda.current_device())
        self.kernel = ((torch.rand(cin, self.units) * 2 - 1) * lim).to(dev)
        self.bias = torch.zeros(self.units, device=dev) if self.use_bias else None

    def call(self, input):
        x = _as_compute(input, self.dtype_code)
        C = x.shape[-1]
        M = x.numel() // C
        if self.use_bias and self.units <= 4:
            y = torch.empty(*x.shape[:-1], self.units, dtype=torch.float32, device=x.device)
            call("gct2_dense_fwd", self.dtype_code, x.data_ptr(), C, self.kernel.data_ptr(), self.bias.data_ptr(), y.data_ptr(),
                 M, C, self.units, _stream(x))
            return y
        w = self.kernel if self.dtype_code == F32 else self.kernel.to(TORCH_DTYPE[self.dtype_code])
        y = torch.empty(*x.shape[:-1], self.units, dtype=x.dtype, device=x.device)
        call("gct2_conv2d_s1_fwd", None, self.dtype_code, x.data_ptr(), C, w.data_ptr(), self.bias.data_ptr() if self.bias is not None else None,
             y.data_ptr(), self.units, M, 1, 1, C, self.units, 1, 0, _stream(x))
        return y


# ---- the model (train.py:175-283) -------------------------------------------------------------------------
class Denoiser(Layer):
    """train.py:175-215.  `self.middle` has the reference's nested structure (and is eagerly callable, one
    kernel launch + one concat copy per layer); `call` runs the planned zero-copy engine instead.
    Both read the SAME parameters: the layers' kernel/bias are views into the engine's arena."""

    def __init__(self, seed: int = 1234, device: Optional[torch.device] = None):
        self.topology = Topology(pixel_size, max_size, octaves)
        self.dtype_code = preferred_dtype_code()
        self._device = device
        self._seed = seed
        self.downs: List[DownShuffle] = [None] * octaves
        self.ups: List[UpShuffle] = [None] * octaves
        self.middle = Block(min(pixel_size * 2 ** octaves, max_size))
        for i in reversed(range(octaves)):
            filters = min(pixel_size * 2 ** i, max_size)
            self.downs[i] = DownShuffle(filters)
            self.ups[i] = UpShuffle(min(pixel_size * 2 ** i // 2, max_size))
            self.middle = Residual(
                Sequential([
                    self.downs[i],
                    Block(filters),
                    self.middle,
                    Block(filters),
                    self.ups[i],
                ])
            )
        self.head = Dense(3)
        self.middle = Sequential([
            Block(pixel_size),
            self.middle,
            Block(pixel_size),
            self.head,
        ])
        self.engine: Optional[UNetEngine] = None

    def variant(self) -> bool:
        """any switch that leaves the default topology (train.py:20, 26, 27): those run on variants.VariantEngine."""
        return block_depth != 0 or residual or not concat

    def ensure_engine(self, **engine_kw):
        """build the engine on first use.  Whoever comes first - `denoiser(...)`, `trainable_variables`, the sampler callback at
        on_epoch_begin, or `trainer(...)` - the engine gets the objective switches of train.py:29-32 from the module-level
        globals (train.py reads them as module constants at call time; r02 took them from Trainer's constructor only, so an
        engine built by anything else silently trained the default objective)."""
        if self.engine is None and self.variant():
            from .variants import VariantEngine
            kw = dict(steps=steps, warm_up=warm_up, seed=self._seed, loss_scaling=bool(mixed_precision), **objective_switches())
            kw.update(engine_kw)
            self.engine = VariantEngine(pixel_size, max_size, octaves, block_depth, residual, concat, self.dtype_code, self._device, **kw)
            self._bind_variant_parameters()
        if self.engine is None:
            # no optimizer known yet (train.py:505-509 calls the model before compile): the module-level mixed_precision
            # decides about loss scaling, as it decides about the LossScaleOptimizer wrapper in train.py:82-83
            kw = dict(steps=steps, warm_up=warm_up, seed=self._seed, loss_scaling=bool(mixed_precision), **objective_switches())
            kw.update(engine_kw)
            self.engine = UNetEngine(self.topology, self.dtype_code, self._device, **kw)
            A = self.engine.arena
            for i in range(octaves):
                for layer, tag in ((self.downs[i], f"D{i}"), (self.ups[i], f"U{i}")):
                    layer.kernel, layer.bias = A.param(tag + ".w"), A.param(tag + ".b")
                    layer.dtype_code = self.dtype_code
                    layer._built = True
            self.head.kernel, self.head.bias = A.param("dense.w"), A.param("dense.b")
            self.head.dtype_code = self.dtype_code
            self.head._built = True
        return self.engine

    def _bind_variant_parameters(self) -> None:
        """the nested eager layers of self.middle share the variant engine's parameters: both enumerate the layers in forward
        order (train.py:183-204), so the k-th layer with a kernel is the k-th (kernel[, bias]) group of the engine."""
        net = self.engine.net
        groups: Dict[str, Dict[str, torch.Tensor]] = {}
        for name, _ in net.specs:
            groups.setdefault(name.rsplit(".", 1)[0], {})[name.rsplit(".", 1)[1]] = net.view(net.p, name)
        order = list(groups)

        def walk(layer):
            if isinstance(layer, Sequential):
                for sub in layer.layers:
                    yield from walk(sub)
            elif isinstance(layer, Residual):
                yield from walk(layer.module)
                if residual:
                    layer.dense = Dense(0, use_bias=False)
                    layer._built = True                          # build() would replace the bound projection
                    yield layer.dense
            elif isinstance(layer, Block):
                yield from layer.convs
            else:
                yield layer

        layers = list(walk(self.middle))
        assert len(layers) == len(order), (len(layers), len(order))
        for layer, key in zip(layers, order):
            layer.kernel, layer.bias = groups[key]["w"], groups[key].get("b")
            if isinstance(layer, Dense):
                layer.units = layer.kernel.shape[-1]
            layer.dtype_code = self.dtype_code
            layer._built = True

    @property
    def trainable_variables(self) -> Dict[str, torch.Tensor]:
        eng = self.ensure_engine()
        if self.variant():
            return {k: eng.net.view(eng.net.p, k) for k in eng.net.shapes}
        A = eng.arena
        return {k: A.param(k) for k in A.shapes}

    def call(self, input):
        x, t = input            # t is ignored by the reference as well (train.py:208-210)
        eng = self.ensure_engine()
        return eng.predict(x).clone()

    def call_eager(self, input):
        """the reference's literal layer-by-layer evaluation of self.middle (train.py:210)."""
        x, t = input
        self.ensure_engine()
        return self.middle(x)


class LambdaCallback:
    """tf.keras.callbacks.LambdaCallback(on_epoch_begin=...) (train.py:519-521)."""

    def __init__(self, on_epoch_begin: Optional[Callable] = None, on_epoch_end: Optional[Callable] = None):
        self.on_epoch_begin = on_epoch_begin
        self.on_epoch_end = on_epoch_end


class Trainer(Layer):
    """train.py:217-283 + the Keras compile/fit driver (train.py:511-523)."""

    def __init__(self, denoiser: Denoiser):
        self.denoiser = denoiser
        self.optimizer = None
        self.loss_fn = None

    def _engine(self) -> UNetEngine:
        opt = self.optimizer
        kw = {}
        if opt is not None and self.denoiser.engine is None:
            inner = getattr(opt, "inner", opt)
            kw.update(beta_1=inner.beta_1, beta_2=inner.beta_2, epsilon=inner.epsilon,
                      loss_scaling=bool(inner.loss_scaling))
            lr = inner.learning_rate
            if isinstance(lr, WarmUp):
                kw.update(base_lr=lr.base, warm_up=lr.warmup_steps)
            elif not callable(lr):
                kw.update(base_lr=float(lr), warm_up=0)
            else:
                raise NotImplementedError("only WarmUp or constant learning rates are supported")
        eng = self.denoiser.ensure_engine(**kw)
        # train.py:238-252 reads the objective globals every time Trainer.call runs: an engine built earlier (by denoiser(...),
        # trainable_variables, the log_sample callback) follows the switches as they stand now
        for k, v in objective_switches().items():
            setattr(eng, k, v)
        return eng

    def call(self, x):
        """returns the scalar fp32 loss for a freshly noised batch (train.py:223-272); no gradients."""
        eng = self._engine()
        x = x.to(eng.device, torch.float32).contiguous()
        if self.denoiser.variant():
            return eng.train_step(x, backward=False).clone()[0]
        b = eng.buffers(*x.shape[:3])
        eng.sample_noise(b)
        eng.noise_into_r0(b, x, unfused_head=True)     # forward(head=True) reads the image channels from R_0 itself
        eng.forward(b)
        if eng.default_objective():
            return eng.loss_and_dpred(b, x).clone()[0]
        target, w = eng.make_target(b, x)              # train.py:238-252
        if eng.objective_weighted():
            return eng.weighted_loss_and_dpred(b, target, w).clone()[0]
        return eng.loss_and_dpred(b, target).clone()[0]

    def compile(self, optimizer, loss):
        """train.py:511-514"""
        if self.denoiser.engine is not None and optimizer is not None:
            eng, inner = self.denoiser.engine, getattr(optimizer, "inner", optimizer)
            eng.beta_1, eng.beta_2, eng.epsilon = inner.beta_1, inner.beta_2, inner.epsilon
            lr = inner.learning_rate
            if isinstance(lr, WarmUp):
                eng.base_lr, eng.warm_up = lr.base, lr.warmup_steps
            elif not callable(lr):
                eng.base_lr, eng.warm_up = float(lr), 0
            if inner.loss_scaling and eng.ls_state is None:
                eng.enable_loss_scaling()                 # train.py:505-514: the model is called before compile
            elif not inner.loss_scaling and eng.ls_state is not None:
                if eng.iterations != 0:
                    raise _lib.Gct2Error("the engine has already stepped with dynamic loss scaling; it cannot be dropped now")
                eng.ls_state, eng.loss_scaling = None, False
        self.optimizer, self.loss_fn = optimizer, loss
        inner = getattr(optimizer, "inner", optimizer)
        if inner is not None:
            inner._engine = self.denoiser.engine

    def train_step(self, data):
        """one Keras train_step on a (x, y) batch with y == x (train.py:293): returns {'loss': tensor}."""
        x = data[0] if isinstance(data, (tuple, list)) else data
        eng = self._engine()
        loss = eng.train_step(x)
        inner = getattr(self.optimizer, "inner", self.optimizer)
        if inner is not None:
            inner._engine = eng
        return {"loss": loss}

    def fit(self, dataset: Iterable, steps_per_epoch: int = 1000, epochs: int = 1, callbacks: Seq = (), verbose: int = 1):
        """train.py:516-523.  `dataset` yields (image, image) batches, NHWC in [-1, 1)."""
        if self.optimizer is None:
            raise RuntimeError("call compile(optimizer, loss) before fit (train.py:511)")
        it = iter(dataset)
        history = {"loss": []}
        for epoch in range(epochs):
            logs: Dict[str, float] = {}
            for cb in callbacks:
                if getattr(cb, "on_epoch_begin", None):
                    cb.on_epoch_begin(epoch, logs)
            running = None
            for _ in range(steps_per_epoch):
                out = self.train_step(next(it))
                running = out["loss"] if running is None else running + out["loss"]
            logs["loss"] = float(running[0]) / steps_per_epoch     # one host sync per epoch
            history["loss"].append(logs["loss"])
            if verbose:
                print(f"Epoch {epoch + 1}/{epochs} - loss: {logs['loss']:.6f}", flush=True)
            for cb in callbacks:
                if getattr(cb, "on_epoch_end", None):
                    cb.on_epoch_end(epoch, logs)
        return history
