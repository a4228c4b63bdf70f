"""CPU ORACLE (test infrastructure, NOT product code) for one `Trainer` train step.

Restates, in numpy, the arithmetic of /root/reference/train.py's hot path exactly as
SURVEY.md Appendix A spells it out.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this package; the product package
(`gan_class_transfer2_amd`) never does and fails loudly when its HIP library is missing.

PARITY UNPINNED: the reference is a TensorFlow/Keras script with no tests, no golden
vectors and no fixtures, and TensorFlow is not installable in the build container
(`import tensorflow` -> ModuleNotFoundError).  This restatement is therefore
pinned only by (i) the definitions read off train.py (cited per function below),
(ii) a pure-Python-loop statement of the same index formulas at tiny shapes
(`naive_*` below) and (iii) an independent torch.nn.functional + autograd formulation
(`oracle/torch_cross.py`).

Layouts are the reference's (Keras channels_last):
  activations  [B, H, W, C]
  Conv2D kernel          (kh, kw, Cin,  Cout)   train.py:161-166
  Conv2DTranspose kernel (kh, kw, Cout, Cin)    train.py:148-153
  Dense kernel           (Cin, Cout)            train.py:198-202
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np


# --------------------------------------------------------------------------------------
# configuration = the reference's module-level globals (train.py:17-36)
# --------------------------------------------------------------------------------------
@dataclass
class OracleConfig:
    size: int = 256            # train.py:17
    pixel_size: int = 128      # train.py:18
    max_size: int = 512        # train.py:19
    block_depth: int = 0       # train.py:20 (Block is identity at 0)
    octaves: int = 6           # train.py:21
    batch_size: int = 1        # train.py:23
    steps: int = 200           # train.py:24
    warm_up: int = 2000        # train.py:36
    base_lr: float = 2e-5      # train.py:75
    beta_1: float = 0.9        # Keras Adam defaults [TF]
    beta_2: float = 0.999
    epsilon: float = 1e-7

    def down_filters(self, i: int) -> int:      # train.py:181
        return min(self.pixel_size * 2 ** i, self.max_size)

    def up_filters(self, i: int) -> int:        # train.py:188
        return min(self.pixel_size * 2 ** i // 2, self.max_size)

    def level_in_channels(self, i: int) -> int:
        """channels of x_i, the input of level i (x_0 = noised image, 3 channels)."""
        return 3 if i == 0 else self.down_filters(i - 1)

    def up_in_channels(self, i: int) -> int:
        """channels fed to UpShuffle_i: Residual_{i+1} output, or D_{octaves-1} output."""
        if i == self.octaves - 1:
            return self.down_filters(i)
        return self.up_filters(i + 1) + self.down_filters(i)

    def check_legal(self) -> None:
        # every D_i input must be even  <=>  size % 2**octaves == 0
        if self.size % (2 ** self.octaves) != 0:
            raise ValueError(
                f"size={self.size} is not divisible by 2**octaves={2 ** self.octaves}: "
                "the skip concat (train.py:114-119) would see mismatched spatial sizes")


# --------------------------------------------------------------------------------------
# schedule pieces
# --------------------------------------------------------------------------------------
def alpha_dash(t, steps: int = 200):
    """train.py:85-93: (1 - t/(steps+1))**2 * 0.25"""
    t = np.asarray(t, dtype=np.float64) / (steps + 1)
    return (1.0 - t) ** 2 * 0.25


def warmup_lr(step: int, base: float = 2e-5, warmup_steps: int = 2000) -> float:
    """train.py:57-65.  `step` = optimizer.iterations BEFORE the increment [TF].
    The reference computes base * float32(step+1) / (warmup+1) in float32."""
    if step < warmup_steps:
        return float(np.float32(base) * np.float32(step + 1) / np.float32(warmup_steps + 1))
    return float(np.float32(base))


# --------------------------------------------------------------------------------------
# parameter construction (Glorot uniform, zero biases)
# --------------------------------------------------------------------------------------
def param_names(cfg: OracleConfig) -> List[str]:
    names = []
    for i in range(cfg.octaves):
        names += [f"D{i}.w", f"D{i}.b"]
    for i in reversed(range(cfg.octaves)):
        names += [f"U{i}.w", f"U{i}.b"]
    names += ["dense.w", "dense.b"]
    return names


def param_shapes(cfg: OracleConfig) -> Dict[str, Tuple[int, ...]]:
    shapes: Dict[str, Tuple[int, ...]] = {}
    for i in range(cfg.octaves):
        shapes[f"D{i}.w"] = (4, 4, cfg.level_in_channels(i), cfg.down_filters(i))
        shapes[f"D{i}.b"] = (cfg.down_filters(i),)
        shapes[f"U{i}.w"] = (4, 4, cfg.up_filters(i), cfg.up_in_channels(i))
        shapes[f"U{i}.b"] = (cfg.up_filters(i),)
    cin = cfg.up_filters(0) + 3
    shapes["dense.w"] = (cin, 3)
    shapes["dense.b"] = (3,)
    return shapes


def glorot_limit(shape: Tuple[int, ...]) -> float:
    """Keras glorot_uniform [TF]: limit = sqrt(6 / (fan_in + fan_out)); for a 4-D kernel
    fan_in = prod(shape[:-2])*shape[-2], fan_out = prod(shape[:-2])*shape[-1]."""
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = rf * shape[-2], rf * shape[-1]
    return math.sqrt(6.0 / (fan_in + fan_out))


def init_params(cfg: OracleConfig, seed: int = 1234, dtype=np.float64) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    shapes = param_shapes(cfg)
    params = {}
    for name in param_names(cfg):
        shp = shapes[name]
        if name.endswith(".b"):
            params[name] = np.zeros(shp, dtype=dtype)
        else:
            lim = glorot_limit(shp)
            params[name] = rng.uniform(-lim, lim, size=shp).astype(np.float32).astype(dtype)
    return params


# --------------------------------------------------------------------------------------
# operand rounding model (what the bf16 / fp16 HIP path stores in HBM)
# --------------------------------------------------------------------------------------
def round_bf16(a: np.ndarray) -> np.ndarray:
    """round-to-nearest-even f32 -> bf16 -> back, value-preserving dtype of `a`."""
    f = np.ascontiguousarray(a, dtype=np.float32)
    u = f.view(np.uint32)
    r = ((u.astype(np.uint64) + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) << 16
    out = r.view(np.float32).astype(a.dtype if a.dtype.kind == "f" else np.float32)
    return out.reshape(a.shape)


def round_f16(a: np.ndarray) -> np.ndarray:
    return np.asarray(a, dtype=np.float32).astype(np.float16).astype(a.dtype)


def _rounder(mode: Optional[str]):
    if mode is None or mode == "f32":
        return lambda a: a
    if mode == "bf16":
        return round_bf16
    if mode == "f16":
        return round_f16
    raise ValueError(mode)


# --------------------------------------------------------------------------------------
# layer arithmetic, vectorised
# --------------------------------------------------------------------------------------
def conv4s2_fwd(x: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    """pre-activation of Conv2D(f, 4, 2, 'same')  (train.py:161-166, A.2), even H/W:
    z[b,oh,ow,o] = b[o] + sum_{kh,kw,i} x[b, 2oh+kh-1, 2ow+kw-1, i] * W[kh,kw,i,o]"""
    B, H, W, C = x.shape
    assert H % 2 == 0 and W % 2 == 0 and w.shape[:3] == (4, 4, C)
    Ho, Wo = H // 2, W // 2
    xp = np.zeros((B, H + 2, W + 2, C), dtype=x.dtype)
    xp[:, 1:-1, 1:-1, :] = x
    z = np.zeros((B, Ho, Wo, w.shape[3]), dtype=x.dtype)
    for kh in range(4):
        for kw in range(4):
            z += xp[:, kh:kh + 2 * Ho:2, kw:kw + 2 * Wo:2, :] @ w[kh, kw]
    return z + b


def conv4s2_bwd(x, w, dz):
    """gradients of conv4s2_fwd wrt (x, w, b) given dz = dL/dz."""
    B, H, W, C = x.shape
    Ho, Wo = H // 2, W // 2
    xp = np.zeros((B, H + 2, W + 2, C), dtype=x.dtype)
    xp[:, 1:-1, 1:-1, :] = x
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(w)
    dz2 = dz.reshape(-1, dz.shape[-1])
    for kh in range(4):
        for kw in range(4):
            patch = xp[:, kh:kh + 2 * Ho:2, kw:kw + 2 * Wo:2, :]
            dw[kh, kw] = patch.reshape(-1, C).T @ dz2
            dxp[:, kh:kh + 2 * Ho:2, kw:kw + 2 * Wo:2, :] += dz @ w[kh, kw].T
    return dxp[:, 1:-1, 1:-1, :], dw, dz2.sum(0)


def convT4s2_fwd(x: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    """pre-activation of Conv2DTranspose(f, 4, 2, 'same') (train.py:148-153, A.3):
    z[b, 2ih+kh-1, 2iw+kw-1, o] += x[b,ih,iw,i] * W[kh,kw,o,i];  out = 2*in."""
    B, H, W, C = x.shape
    assert w.shape[0] == 4 and w.shape[1] == 4 and w.shape[3] == C
    zp = np.zeros((B, 2 * H + 2, 2 * W + 2, w.shape[2]), dtype=x.dtype)
    for kh in range(4):
        for kw in range(4):
            zp[:, kh:kh + 2 * H:2, kw:kw + 2 * W:2, :] += x @ w[kh, kw].T
    return zp[:, 1:-1, 1:-1, :] + b


def convT4s2_bwd(x, w, dz):
    B, H, W, C = x.shape
    dzp = np.zeros((B, 2 * H + 2, 2 * W + 2, dz.shape[-1]), dtype=dz.dtype)
    dzp[:, 1:-1, 1:-1, :] = dz
    dx = np.zeros_like(x)
    dw = np.zeros_like(w)
    x2 = x.reshape(-1, C)
    for kh in range(4):
        for kw in range(4):
            patch = dzp[:, kh:kh + 2 * H:2, kw:kw + 2 * W:2, :]
            dx += patch @ w[kh, kw]
            dw[kh, kw] = patch.reshape(-1, dz.shape[-1]).T @ x2
    return dx, dw, dz.reshape(-1, dz.shape[-1]).sum(0)


# --------------------------------------------------------------------------------------
# definition-level (pure Python loops) statements of the same formulas, tiny shapes only
# --------------------------------------------------------------------------------------
def naive_conv4s2_fwd(x, w, b):
    B, H, W, C = x.shape
    Ho, Wo, O = H // 2, W // 2, w.shape[3]
    z = np.zeros((B, Ho, Wo, O), dtype=np.float64)
    for n in range(B):
        for oh in range(Ho):
            for ow in range(Wo):
                for o in range(O):
                    acc = float(b[o])
                    for kh in range(4):
                        ih = 2 * oh + kh - 1
                        if ih < 0 or ih >= H:
                            continue
                        for kw in range(4):
                            iw = 2 * ow + kw - 1
                            if iw < 0 or iw >= W:
                                continue
                            for i in range(C):
                                acc += float(x[n, ih, iw, i]) * float(w[kh, kw, i, o])
                    z[n, oh, ow, o] = acc
    return z


def naive_convT4s2_fwd(x, w, b):
    B, H, W, C = x.shape
    O = w.shape[2]
    z = np.zeros((B, 2 * H, 2 * W, O), dtype=np.float64)
    for n in range(B):
        for ih in range(H):
            for iw in range(W):
                for kh in range(4):
                    oh = 2 * ih + kh - 1
                    if oh < 0 or oh >= 2 * H:
                        continue
                    for kw in range(4):
                        ow = 2 * iw + kw - 1
                        if ow < 0 or ow >= 2 * W:
                            continue
                        for o in range(O):
                            for i in range(C):
                                z[n, oh, ow, o] += float(x[n, ih, iw, i]) * float(w[kh, kw, o, i])
    return z + np.asarray(b, dtype=np.float64)


# --------------------------------------------------------------------------------------
# the network (train.py:175-215) and the objective (train.py:223-272)
# --------------------------------------------------------------------------------------
def noise_image(x, t_int, eps, steps=200):
    """train.py:229-234; t_int has shape [B] (the reference keeps it as [B,1,1,1])."""
    a = alpha_dash(np.asarray(t_int, dtype=np.float64), steps).reshape(-1, 1, 1, 1).astype(x.dtype)
    return x * np.sqrt(a) + eps * np.sqrt(1.0 - a)


def noise_image_f16(x, t_int, eps, steps=200):
    """train.py:229-234 under `mixed_precision = True` (train.py:34,38,43-45): x (train.py:292), epsilon (train.py:227) and
    t = cast(t_int, x.dtype) (train.py:229) are float16 tensors, so alpha_dash and the mix run in float16 - every operation
    rounds to fp16 [TF]: t /= steps+1; 1 - t; (.)**2; * 0.25; **0.5; 1 - alpha; **0.5; the two products; the sum.
    numpy's float16 arithmetic rounds each operation the same way.  Returns a float16 array."""
    h = np.float16
    t = np.asarray(t_int).astype(h) / h(steps + 1)
    om = h(1.0) - t
    a = (om * om).astype(h) * h(0.25)
    sa = np.sqrt(a.astype(np.float32)).astype(h).reshape(-1, 1, 1, 1)
    sb = np.sqrt((h(1.0) - a).astype(np.float32)).astype(h).reshape(-1, 1, 1, 1)
    x16, e16 = np.asarray(x).astype(h), np.asarray(eps).astype(h)
    return (x16 * sa).astype(h) + (e16 * sb).astype(h)


def unet_forward(params, x0, cfg: OracleConfig, operand_round: Optional[str] = None, keras_strict: bool = False):
    """Denoiser.call (train.py:206-215): `t` is ignored; returns (prediction, cache).

    operand_round in {None,'bf16','f16'} models the low-precision HIP path: weights and every
    stored activation are rounded to that type, accumulation stays in the array dtype.
    The Dense head output is kept unrounded (the HIP path writes it as fp32).

    keras_strict (with operand_round, r06): the two rounding points of Keras' mixed-precision policy (train.py:43-45) that the HIP
    path - and therefore the default rounding model - leaves out, both in the direction of MORE precision: under `mixed_float16` a
    layer's variables are cast to the compute dtype, its convolution / matmul returns a compute-dtype tensor, and the bias is
    added to THAT in the compute dtype [TF] - z = rnd(rnd(conv) + rnd(b)) - where the HIP kernels add the fp32 bias to the fp32
    accumulator and round once; the Dense(3) kernel and bias are cast as well.  (The second point - variable gradients are
    compute-dtype tensors - is in unet_backward.)  Used to MEASURE the documented deviation (tests log it), never as the parity
    target of the HIP path."""
    rnd = _rounder(operand_round)
    n = cfg.octaves
    # conv / transposed-conv kernels are consumed as rounded operands; the 67x3 Dense kernel stays fp32
    wq = {k: (rnd(v) if k.endswith(".w") and (k != "dense.w" or keras_strict) else v) for k, v in params.items()}

    def with_bias(z_nobias, b):         # the layer's pre-activation
        return rnd(rnd(z_nobias) + rnd(b)) if keras_strict else z_nobias + b
    zero = lambda b: np.zeros_like(b)
    xs = [rnd(x0)]                      # x_i : input of level i
    for i in range(n):
        z = with_bias(conv4s2_fwd(xs[i], wq[f"D{i}.w"], zero(params[f"D{i}.b"])), params[f"D{i}.b"])
        xs.append(rnd(np.maximum(z, 0)))
    # inner_{n-1} is Block(...) = identity (train.py:179, block_depth = 0)
    r = xs[n]                           # what UpShuffle_{n-1} consumes
    rs = [None] * n                     # R_i = concat([U_i(...), x_i], -1)  (train.py:113-119)
    uin = [None] * n
    for i in reversed(range(n)):
        uin[i] = r
        z = with_bias(convT4s2_fwd(r, wq[f"U{i}.w"], zero(params[f"U{i}.b"])), params[f"U{i}.b"])
        u = rnd(np.maximum(z, 0))
        r = np.concatenate([u, xs[i]], axis=-1)
        rs[i] = r
    pred = with_bias(rs[0] @ wq["dense.w"], params["dense.b"])       # Dense(3), linear (train.py:198-202)
    return pred, dict(xs=xs, rs=rs, uin=uin, wq=wq)


def unet_backward(params, cache, dpred, cfg: OracleConfig, operand_round: Optional[str] = None, keras_strict: bool = False):
    """hand-derived reverse pass; returns gradients for every parameter.
    Stored activation gradients are rounded like the HIP path stores them.

    keras_strict: every variable gradient is a COMPUTE-DTYPE tensor, as under Keras' mixed-precision policy [TF] (the gradient of
    the variable's fp32 -> fp16 cast is the cast back of an fp16 tensor): rounded to `operand_round` on the way out, overflowing to
    inf beyond the type's range (fp16: 65504 - which makes the reference's LossScaleOptimizer skip steps that the HIP path, with
    its fp32 weight and bias gradients, applies)."""
    rnd = _rounder(operand_round)
    n = cfg.octaves
    xs, rs, uin, wq = cache["xs"], cache["rs"], cache["uin"], cache["wq"]
    g: Dict[str, np.ndarray] = {}
    r0 = rs[0].reshape(-1, rs[0].shape[-1])
    dp = dpred.reshape(-1, 3)
    g["dense.w"] = r0.T @ dp
    g["dense.b"] = dp.sum(0)
    dR = dpred @ wq["dense.w"].T                            # gradient wrt R_0 (all channels)

    def down_pass(i, dR_i):
        """backward of level i given dL/dR_i; returns dL/dx_i (None for i == 0)."""
        fu = cfg.up_filters(i)
        # U_i : ReLU mask, then transposed-conv backward
        du = rnd(dR_i[..., :fu] * (rs[i][..., :fu] > 0))
        dxin, g[f"U{i}.w"], g[f"U{i}.b"] = convT4s2_bwd(uin[i], wq[f"U{i}.w"], du)
        if i == n - 1:
            d_out_D = dxin                                   # straight into D_{n-1}'s output
        else:
            d_out_D = down_pass(i + 1, dxin)                 # gradient wrt x_{i+1} = D_i output
        dz = rnd(d_out_D * (xs[i + 1] > 0))
        dx, g[f"D{i}.w"], g[f"D{i}.b"] = conv4s2_bwd(xs[i], wq[f"D{i}.w"], dz)
        # skip branch of the concat adds to the conv's input gradient
        return dx + dR_i[..., fu:]

    down_pass(0, dR)
    if keras_strict:
        with np.errstate(over="ignore"):
            g = {k: rnd(v) for k, v in g.items()}
    return g


def objective_terms(x, t_int, eps, steps: int = 200, predict_x: bool = True, predict_scaled_epsilon: bool = False,
                    prediction_weighting: bool = False, ordinary_differential_equation: bool = False):
    """target and prediction weight of train.py:238-252 for the four mode switches of train.py:29-32:
        ODE (train.py:238-242):   target = x sqrt(a(t-1)) + eps sqrt(1 - a(t-1))
        predict_x (243-244):      target = x
        else (245-252):           target = eps, * sqrt(1 - a(t)) if predict_scaled_epsilon;
                                  if prediction_weighting: target and prediction both * sqrt(1 - a(t))
    returns (target, w) with w the per-image factor on the prediction ([B,1,1,1], ones unless prediction_weighting)."""
    t = np.asarray(t_int, dtype=np.float64).reshape(-1, 1, 1, 1)
    w = np.ones_like(t)
    if ordinary_differential_equation:
        a1 = alpha_dash(t - 1, steps)
        return x * np.sqrt(a1) + eps * np.sqrt(1.0 - a1), w
    if predict_x:
        return x, w
    target = eps
    s = np.sqrt(1.0 - alpha_dash(t, steps))
    if predict_scaled_epsilon:
        target = target * s
    if prediction_weighting:
        target = target * s
        w = s
    return target, w


def trainer_step(params, x, t_int, eps, cfg: OracleConfig, operand_round: Optional[str] = None, loss_scale: float = 1.0,
                 objective: Optional[dict] = None, keras_strict: bool = False):
    """Trainer.call (default branch predict_x=True; `objective` = keyword arguments of objective_terms selects the other
    branches of train.py:238-252): returns (loss, pred, grads, noised).
    loss = mean((x - pred)^2) in the working dtype (train.py:262-272);
    identity(...) then takes reduce_mean of that scalar (train.py:171-173) = same scalar.

    operand_round == "f16" is the reference's `mixed_precision = True` mode (Keras mixed_float16, train.py:43-45) with its
    rounding points [TF]: the noising runs in fp16 arithmetic (noise_image_f16), the Dense(3) output is an fp16 tensor that
    train.py:263 casts to fp32 for the loss, and the gradient that enters it is fp16 as well - which is why the
    LossScaleOptimizer (train.py:82-83) multiplies the loss by `loss_scale` first; the returned gradients are the SCALED ones.
    (Known deviation kept out of this model and of the HIP path, DESIGN.md section 4: Keras also rounds every variable gradient
    and the conv output before the bias add to fp16.  keras_strict=True ADDS those two rounding points - unet_forward /
    unet_backward - so that the size of the deviation can be measured: profiles/r06_parity.json.)"""
    f16 = operand_round == "f16"
    noised = noise_image_f16(x, t_int, eps, cfg.steps).astype(x.dtype) if f16 else noise_image(x, t_int, eps, cfg.steps)
    pred, cache = unet_forward(params, noised, cfg, operand_round, keras_strict)
    if f16:
        pred = round_f16(pred)
    target, w = objective_terms(x, t_int, eps, cfg.steps, **(objective or {}))
    diff = pred * w - target
    nel = diff.size
    loss = float(np.sum(diff.astype(np.float64) ** 2) / nel)
    dpred = (2.0 * loss_scale / nel) * diff * w
    if f16:
        dpred = round_f16(dpred)
    grads = unet_backward(params, cache, dpred, cfg, operand_round, keras_strict)
    return loss, pred, grads, noised


# --------------------------------------------------------------------------------------
# Keras Adam + WarmUp; fp32 arithmetic like ResourceApplyAdam [TF]
# --------------------------------------------------------------------------------------
def keras_adam_step(p, g, m, v, k: int, cfg: OracleConfig, dtype=np.float32):
    """one apply_gradients for a single tensor.  `k` = iterations before the step.
    m <- b1 m + (1-b1) g ; v <- b2 v + (1-b2) g^2 ;
    p <- p - lr_k * sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps),  t = k+1,
    eps = 1e-7 added to sqrt(v), NOT to sqrt(v_hat)."""
    t = k + 1
    lr = warmup_lr(k, cfg.base_lr, cfg.warm_up)
    b1, b2 = float(np.float32(cfg.beta_1)), float(np.float32(cfg.beta_2))   # Keras holds the hyper-parameters as float32 [TF]
    alpha = dtype(lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t))
    p = p.astype(dtype); g = g.astype(dtype); m = m.astype(dtype); v = v.astype(dtype)
    m = dtype(b1) * m + dtype(1.0 - b1) * g
    v = dtype(b2) * v + dtype(1.0 - b2) * g * g
    p = p - alpha * m / (np.sqrt(v) + dtype(cfg.epsilon))
    return p, m, v


@dataclass
class LossScaleState:
    """Keras LossScaleOptimizer dynamic loss scaling [TF] (train.py:82-83):
    start 2**15; x2 after 2000 consecutive finite steps; /2 and skip the update on inf/nan."""
    scale: float = 2.0 ** 15
    good_steps: int = 0
    growth_interval: int = 2000

    def update(self, grads_finite: bool) -> bool:
        """returns True when the optimizer step must be applied."""
        if grads_finite:
            self.good_steps += 1
            if self.good_steps >= self.growth_interval:
                self.scale *= 2.0
                self.good_steps = 0
            return True
        self.scale = max(self.scale / 2.0, 1.0)
        self.good_steps = 0
        return False


@dataclass
class OracleTrainer:
    """stateful wrapper: the reference's `trainer.fit` driver reduced to its arithmetic."""
    cfg: OracleConfig
    params: Dict[str, np.ndarray]
    operand_round: Optional[str] = None
    iterations: int = 0
    m: Dict[str, np.ndarray] = field(default_factory=dict)
    v: Dict[str, np.ndarray] = field(default_factory=dict)
    loss_scale: Optional[LossScaleState] = None     # LossScaleOptimizer (train.py:82-83); None = plain Adam

    def __post_init__(self):
        for k_, p in self.params.items():
            self.m.setdefault(k_, np.zeros_like(p, dtype=np.float32))
            self.v.setdefault(k_, np.zeros_like(p, dtype=np.float32))

    def train_step(self, x, t_int, eps):
        scale = self.loss_scale.scale if self.loss_scale is not None else 1.0
        loss, pred, grads, _ = trainer_step(self.params, x, t_int, eps, self.cfg, self.operand_round, loss_scale=scale)
        if self.loss_scale is not None:
            # LossScaleOptimizer [TF]: unscale; inf/nan anywhere -> halve the scale and skip the inner apply_gradients, so
            # optimizer.iterations (and with it the WarmUp step and Adam's bias correction) does not advance
            with np.errstate(invalid="ignore", over="ignore"):
                grads = {k_: g / scale for k_, g in grads.items()}
            finite = all(bool(np.isfinite(g).all()) for g in grads.values())
            if not self.loss_scale.update(finite):
                return loss, pred, grads
        for name in self.params:
            p, m, v = keras_adam_step(self.params[name], grads[name], self.m[name], self.v[name],
                                      self.iterations, self.cfg)
            self.params[name] = p.astype(self.params[name].dtype)
            self.m[name], self.v[name] = m, v
        self.iterations += 1
        return loss, pred, grads


# --------------------------------------------------------------------------------------
# synthetic inputs with the loader's value contract (train.py:292: u8/128 - 1)
# --------------------------------------------------------------------------------------
def synthetic_batch(cfg: OracleConfig, seed: int = 0, batch: Optional[int] = None, dtype=np.float64):
    rng = np.random.default_rng(seed)
    B = batch or cfg.batch_size
    u8 = rng.integers(0, 256, size=(B, cfg.size, cfg.size, 3), dtype=np.int64)
    x = (u8.astype(np.float64) / 128.0 - 1.0).astype(dtype)
    t_int = rng.integers(1, cfg.steps + 1, size=(B,), dtype=np.int64)
    eps = rng.standard_normal(size=x.shape).astype(np.float32).astype(dtype)
    return x, t_int, eps
