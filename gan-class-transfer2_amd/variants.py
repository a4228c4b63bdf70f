"""The reference's off-by-default model variants, executed layer by layer through the C ABI.

    block_depth > 0   Block = block_depth x Conv2D(filters, 3, 1, 'same', relu)                        train.py:20, 123-143
    residual = True   Residual.call = input + Dense(input_channels, use_bias=False)(module(input))     train.py:26, 104-112
    concat = False    Residual.call = module(input)                                                    train.py:27, 120-121

They change the channel plan of the whole network (a Block in front of level 0 turns the 3-channel image into pixel_size channels,
the residual form keeps every level's width), so the zero-copy plan of engine.py (built for the default topology, the metric's hot
path) does not apply: `VariantEngine` walks the nested structure of train.py:175-204 with a forward and a hand-written reverse pass
per node.  Same kernels as the hot path for DownShuffle / UpShuffle (gct2_conv4s2_*, gct2_convT4s2_*), the stride-1 convolution
entry points for Block and the 1 x 1 projection, the Dense(3) head, MSE, Keras Adam, the loss-scale state machine.  PyTorch only
moves memory here (clone, cat, contiguous slices).  Built for results (parity-tested against oracle/variants_oracle.py), not for the
roofline: these branches are unreachable at the reference's defaults.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import BF16, F16, F32, call

TORCH_DTYPE = {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}


class _Net:
    """parameter bookkeeping shared by the nodes: one flat fp32 arena each for p, m, v, g and a compute-dtype operand copy."""

    def __init__(self, dtype: int, device: torch.device):
        self.dtype, self.device = dtype, device
        self.specs: List[Tuple[str, Tuple[int, ...]]] = []
        self.offsets: Dict[str, int] = {}
        self.total = 0
        self.ctx = _lib.Context()

    def declare(self, name: str, shape: Tuple[int, ...]) -> str:
        self.specs.append((name, shape))
        self.offsets[name] = self.total
        self.total += (int(np.prod(shape)) + 63) // 64 * 64
        return name

    def allocate(self) -> None:
        z = lambda dt: torch.zeros(max(self.total, 64), dtype=dt, device=self.device)
        self.p, self.m, self.v, self.g = z(torch.float32), z(torch.float32), z(torch.float32), z(torch.float32)
        self.op = z(TORCH_DTYPE[self.dtype]) if self.dtype != F32 else self.p
        self.shapes = dict(self.specs)

    def view(self, arena: torch.Tensor, name: str) -> torch.Tensor:
        o, shp = self.offsets[name], self.shapes[name]
        return arena[o:o + int(np.prod(shp))].view(shp)

    def optr(self, name: str) -> int:       # operand (compute dtype) pointer
        return self.op.data_ptr() + self.offsets[name] * self.op.element_size()

    def pptr(self, name: str) -> int:
        return self.p.data_ptr() + 4 * self.offsets[name]

    def gptr(self, name: str) -> int:
        return self.g.data_ptr() + 4 * self.offsets[name]

    def stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream


class _Conv:
    """y = relu(conv(x) + b) for the three convolution kinds; reverse pass: ReLU mask, weight / bias gradient, input gradient."""

    def __init__(self, net: _Net, name: str, kind: str, cin: int, cout: int):
        self.net, self.kind, self.cin, self.cout = net, kind, cin, cout
        shape = {"down": (4, 4, cin, cout), "up": (4, 4, cout, cin), "c3": (3, 3, cin, cout)}[kind]
        self.w, self.b = net.declare(name + ".w", shape), net.declare(name + ".b", (cout,))

    def fwd(self, x: torch.Tensor) -> torch.Tensor:
        n, (B, H, W, C) = self.net, x.shape
        assert C == self.cin, (C, self.cin)
        Ho, Wo = {"down": (H // 2, W // 2), "up": (2 * H, 2 * W), "c3": (H, W)}[self.kind]
        y = torch.empty(B, Ho, Wo, self.cout, dtype=x.dtype, device=x.device)
        cx, dt, s = n.ctx.handle, n.dtype, n.stream()
        if self.kind == "down":
            if H % 2 or W % 2:
                raise ValueError(f"DownShuffle needs even spatial dims, got {H}x{W} (train.py:114-119)")
            call("gct2_conv4s2_fwd", cx, dt, x.data_ptr(), C, n.optr(self.w), n.pptr(self.b), y.data_ptr(), self.cout, B, H, W, C, self.cout, 1, s)
        elif self.kind == "up":
            call("gct2_convT4s2_fwd", cx, dt, x.data_ptr(), C, n.optr(self.w), n.pptr(self.b), y.data_ptr(), self.cout, B, H, W, C, self.cout, 1, s)
        else:
            call("gct2_conv2d_s1_fwd", cx, dt, x.data_ptr(), C, n.optr(self.w), n.pptr(self.b), y.data_ptr(), self.cout, B, H, W, C, self.cout, 3, 1, s)
        self.x, self.y = x, y
        return y

    def bwd(self, dy: torch.Tensor) -> torch.Tensor:
        n, x, y = self.net, self.x, self.y
        B, H, W, C = x.shape
        cx, dt, s = n.ctx.handle, n.dtype, n.stream()
        dz = dy.contiguous().clone()
        call("gct2_relu_mask", dt, y.data_ptr(), self.cout, dz.data_ptr(), self.cout, dz.numel() // self.cout, self.cout, s)
        dx = torch.empty_like(x)
        if self.kind == "down":
            call("gct2_conv4s2_wgrad", cx, dt, x.data_ptr(), C, dz.data_ptr(), self.cout, n.gptr(self.w), n.gptr(self.b), B, H, W, C, self.cout, 0, None, s)
            call("gct2_conv4s2_dgrad", cx, dt, dz.data_ptr(), self.cout, n.optr(self.w), None, 0, dx.data_ptr(), C, B, H, W, C, self.cout, 0,
                 None, 0, None, 0, s)
        elif self.kind == "up":
            call("gct2_convT4s2_wgrad", cx, dt, x.data_ptr(), C, dz.data_ptr(), self.cout, n.gptr(self.w), n.gptr(self.b), B, H, W, C, self.cout, 0, None, s)
            call("gct2_convT4s2_dgrad", cx, dt, dz.data_ptr(), self.cout, n.optr(self.w), None, 0, dx.data_ptr(), C, B, H, W, C, self.cout, 0,
                 None, 0, None, 0, s)
        else:
            call("gct2_conv2d_s1_wgrad", cx, dt, x.data_ptr(), C, dz.data_ptr(), self.cout, n.gptr(self.w), n.gptr(self.b), B, H, W, C, self.cout, 3, 0, s)
            call("gct2_conv2d_s1_dgrad", cx, dt, dz.data_ptr(), self.cout, n.optr(self.w), None, 0, dx.data_ptr(), C, B, H, W, C, self.cout, 3, 0, s)
        self.x = self.y = None
        return dx


class _Proj:
    """Dense(cout, use_bias=False) on a rank-4 tensor (train.py:106): a 1 x 1 convolution without bias or activation."""

    def __init__(self, net: _Net, name: str, cin: int, cout: int):
        self.net, self.cin, self.cout = net, cin, cout
        self.w = net.declare(name + ".w", (cin, cout))

    def fwd(self, x):
        n, (B, H, W, C) = self.net, x.shape
        y = torch.empty(B, H, W, self.cout, dtype=x.dtype, device=x.device)
        call("gct2_conv2d_s1_fwd", n.ctx.handle, n.dtype, x.data_ptr(), C, n.optr(self.w), None, y.data_ptr(), self.cout, B, H, W, C, self.cout,
             1, 0, n.stream())
        self.x = x
        return y

    def bwd(self, dy):
        n, x = self.net, self.x
        B, H, W, C = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        call("gct2_conv2d_s1_wgrad", n.ctx.handle, n.dtype, x.data_ptr(), C, dy.data_ptr(), self.cout, n.gptr(self.w), None, B, H, W, C, self.cout,
             1, 0, n.stream())
        call("gct2_conv2d_s1_dgrad", n.ctx.handle, n.dtype, dy.data_ptr(), self.cout, n.optr(self.w), None, 0, dx.data_ptr(), C, B, H, W, C,
             self.cout, 1, 0, n.stream())
        self.x = None
        return dx


class _Seq:
    def __init__(self, nodes):
        self.nodes = [n for n in nodes if n is not None]

    def fwd(self, x):
        for n in self.nodes:
            x = n.fwd(x)
        return x

    def bwd(self, dy):
        for n in reversed(self.nodes):
            dy = n.bwd(dy)
        return dy


class _Residual:
    """train.py:97-121 in all three modes."""

    def __init__(self, net: _Net, name: str, module: _Seq, cin: int, cmod: int, residual: bool, concat: bool):
        self.net, self.module, self.cmod = net, module, cmod
        self.mode = "residual" if residual else ("concat" if concat else "module")
        self.proj = _Proj(net, name + ".dense", cmod, cin) if residual else None          # train.py:106: Dense(input_shape[-1])
        self.cout = {"residual": cin, "concat": cmod + cin, "module": cmod}[self.mode]

    def _add(self, dst, src):
        C = dst.shape[-1]
        call("gct2_add", self.net.dtype, dst.data_ptr(), C, src.data_ptr(), C, dst.numel() // C, C, self.net.stream())

    def fwd(self, x):
        m = self.module.fwd(x)
        if self.mode == "residual":                              # input + self.dense(self.module(input))
            y = x.clone()
            self._add(y, self.proj.fwd(m))
            return y
        if self.mode == "concat":                                # tf.concat([module(input), highway(input)], -1)
            return torch.cat([m, x], -1)
        return m

    def bwd(self, dy):
        if self.mode == "residual":
            dx = dy.contiguous().clone()
            self._add(dx, self.module.bwd(self.proj.bwd(dy)))
            return dx
        if self.mode == "concat":
            dx = dy[..., self.cmod:].contiguous()
            self._add(dx, self.module.bwd(dy[..., :self.cmod].contiguous()))
            return dx
        return self.module.bwd(dy)


class _Head:
    """Dense(3) (train.py:198-202): fp32 output for the fp32 loss (train.py:262-263)."""

    def __init__(self, net: _Net, cin: int):
        self.net, self.cin = net, cin
        self.w, self.b = net.declare("dense.w", (cin, 3)), net.declare("dense.b", (3,))

    def fwd(self, x):
        n = self.net
        M = x.numel() // self.cin
        y = torch.empty(*x.shape[:-1], 3, dtype=torch.float32, device=x.device)
        call("gct2_dense_fwd", n.dtype, x.data_ptr(), self.cin, n.pptr(self.w), n.pptr(self.b), y.data_ptr(), M, self.cin, 3, n.stream())
        self.x = x
        return y

    def bwd(self, dpred):
        """dpred: fp32 [B,H,W,3].  Kernel / bias gradients in fp32 from the fp32 gradient (gct2_dense_bwd with no masked input
        gradient); the input gradient through the 1 x 1 convolution entry point, since the head's input is not a ReLU output in
        every variant."""
        n, x = self.net, self.x
        B, H, W, C = x.shape
        M = B * H * W
        dummy = torch.empty(8, dtype=x.dtype, device=x.device)
        call("gct2_dense_bwd", n.dtype, x.data_ptr(), C, n.pptr(self.w), dpred.data_ptr(), dummy.data_ptr(), 0, n.gptr(self.w), n.gptr(self.b),
             M, C, 3, 0, 0, n.stream())
        dz = dpred.to(x.dtype)
        dx = torch.empty_like(x)
        call("gct2_conv2d_s1_dgrad", n.ctx.handle, n.dtype, dz.data_ptr(), 3, n.optr(self.w), None, 0, dx.data_ptr(), C, B, H, W, C, 3, 1, 0,
             n.stream())
        self.x = None
        return dx


def build_structure(net: _Net, pixel_size: int, max_size: int, octaves: int, block_depth: int, residual: bool, concat: bool):
    """Denoiser.__init__ (train.py:175-204) with every switch honoured; returns (top sequential, channel count fed to Dense(3))."""

    def block(name: str, cin: int, filters: int):
        """Block(filters) (train.py:123-143): block_depth x [Conv2D(filters, 3, 1, 'same', relu)]; identity at depth 0."""
        nodes, c = [], cin
        for d in range(block_depth):
            nodes.append(_Conv(net, f"{name}.{d}", "c3", c, filters))
            c = filters
        return (_Seq(nodes) if nodes else None), c

    def level(i: int, cin: int):
        """the Residual of level i (train.py:180-190) and its output channel count."""
        filters = min(pixel_size * 2 ** i, max_size)
        down = _Conv(net, f"D{i}", "down", cin, filters)
        ba, c = block(f"blkA{i}", filters, filters)
        if i + 1 < octaves:
            inner, c = level(i + 1, c)
        else:
            inner, c = block("blkMid", c, min(pixel_size * 2 ** octaves, max_size))          # train.py:179
        bb, c = block(f"blkB{i}", c, filters)
        fu = min(pixel_size * 2 ** i // 2, max_size)
        up = _Conv(net, f"U{i}", "up", c, fu)
        res = _Residual(net, f"res{i}", _Seq([down, ba, inner, bb, up]), cin, fu, residual, concat)
        return res, res.cout

    b0, c = block("blkTopA", 3, pixel_size)                                                   # train.py:192
    if octaves > 0:
        mid, c = level(0, c)
    else:
        mid, c = block("blkMid", c, min(pixel_size, max_size))
    b1, c = block("blkTopB", c, pixel_size)                                                   # train.py:194
    head = _Head(net, c)
    return _Seq([b0, mid, b1, head]), c


class VariantEngine:
    """train step of a Denoiser built with block_depth > 0 / residual / concat=False (same surface as UNetEngine where it matters:
    train_step, predict, get/set_params, get_grads, iterations, loss_scale, the objective switches)."""

    def __init__(self, pixel_size: int, max_size: int, octaves: int, block_depth: int, residual: bool, concat: bool, dtype: int = F32,
                 device: Optional[torch.device] = None, steps: int = 200, base_lr: float = 2e-5, warm_up: int = 2000, beta_1: float = 0.9,
                 beta_2: float = 0.999, epsilon: float = 1e-7, loss_scaling: bool = False, seed: int = 1234, rng_seed: int = 0,
                 predict_x: bool = True, predict_scaled_epsilon: bool = False, prediction_weighting: bool = False,
                 ordinary_differential_equation: bool = False):
        _lib.load()
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        if self.device.type != "cuda":
            raise _lib.Gct2Error("VariantEngine needs a HIP device (torch device 'cuda'); there is no CPU path")
        call("gct2_device_check")
        self.dtype, self.steps, self.octaves = dtype, steps, octaves
        self.base_lr, self.warm_up, self.beta_1, self.beta_2, self.epsilon = base_lr, warm_up, beta_1, beta_2, epsilon
        self.predict_x, self.predict_scaled_epsilon = predict_x, predict_scaled_epsilon
        self.prediction_weighting, self.ordinary_differential_equation = prediction_weighting, ordinary_differential_equation
        self.net = _Net(dtype, self.device)
        self.workspace = torch.empty(16 << 18, dtype=torch.float32, device=self.device)
        self.net.ctx.set_workspace(self.workspace)
        self.top, self.head_cin = build_structure(self.net, pixel_size, max_size, octaves, block_depth, residual, concat)
        self.net.allocate()
        self.glorot_init(seed)
        self._iterations = 0
        self.ls_state = None
        if loss_scaling:
            self.ls_state = torch.zeros(8, dtype=torch.int32, device=self.device)
            call("gct2_loss_scale_init", self.ls_state.data_ptr(), float(2 ** 15), self.net.stream())
        self.rng_seed, self.rng_offset_t, self.rng_offset_eps = rng_seed, 0, 0
        self.partials = torch.zeros(1024, dtype=torch.float32, device=self.device)
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.last = {}

    # ---- parameters ---------------------------------------------------------------------------------------------------------
    @property
    def shapes(self) -> Dict[str, Tuple[int, ...]]:
        return self.net.shapes

    def glorot_init(self, seed: int) -> None:
        """Keras glorot_uniform kernels, zero biases (train.py:134,149,162; Dense default [TF])."""
        gen = torch.Generator(device="cpu").manual_seed(seed)
        for name, shp in self.net.specs:
            if name.endswith(".b"):
                continue
            rf = int(np.prod(shp[:-2])) if len(shp) > 2 else 1
            lim = math.sqrt(6.0 / (rf * shp[-2] + rf * shp[-1]))
            self.net.view(self.net.p, name).copy_(((torch.rand(shp, generator=gen) * 2 - 1) * lim).to(self.device))
        self.refresh_operands()

    def refresh_operands(self) -> None:
        if self.dtype != F32:
            call("gct2_cast_from_f32", self.dtype, self.net.p.data_ptr(), self.net.op.data_ptr(), self.net.p.numel(), self.net.stream())

    def set_params(self, params) -> None:
        for k, v in params.items():
            self.net.view(self.net.p, k).copy_(torch.as_tensor(np.asarray(v, dtype=np.float32)).to(self.device))
        self.refresh_operands()

    def get_params(self):
        return {k: self.net.view(self.net.p, k).cpu().numpy().copy() for k in self.net.shapes}

    def get_grads(self):
        return {k: self.net.view(self.net.g, k).cpu().numpy().copy() for k in self.net.shapes}

    @property
    def iterations(self) -> int:
        return int(self.ls_state[4].item()) if self.ls_state is not None else self._iterations

    def loss_scale(self):
        if self.ls_state is None:
            return 1.0, 0
        raw = self.ls_state.cpu()
        return float(raw[:1].view(torch.float32)[0]), int(raw[2])

    # ---- the step -----------------------------------------------------------------------------------------------------------
    def _alpha(self) -> float:
        k = self._iterations
        lr = float(np.float32(self.base_lr) * np.float32(k + 1) / np.float32(self.warm_up + 1)) if k < self.warm_up else float(np.float32(self.base_lr))
        b1, b2 = float(np.float32(self.beta_1)), float(np.float32(self.beta_2))
        return lr * math.sqrt(1.0 - b2 ** (k + 1)) / (1.0 - b1 ** (k + 1))

    def _noised(self, x, t_int, eps):
        B, H, W, _ = x.shape
        s = self.net.stream()
        if t_int is None:
            t_int = torch.zeros(B, dtype=torch.int32, device=self.device)
            call("gct2_rng_uniform_int", self.rng_seed, 1, self.rng_offset_t, t_int.data_ptr(), B, 1, self.steps, s)
            self.rng_offset_t += B
        else:
            t_int = t_int.to(self.device, torch.int32).contiguous()
        if eps is None:
            eps = torch.zeros_like(x)
            call("gct2_rng_normal", self.rng_seed, 2, self.rng_offset_eps, eps.data_ptr(), eps.numel(), s)
            self.rng_offset_eps += eps.numel()
        else:
            eps = eps.to(self.device, torch.float32).contiguous()
        out = torch.empty(B, H, W, 3, dtype=TORCH_DTYPE[self.dtype], device=self.device)
        call("gct2_noise_image", self.dtype, x.data_ptr(), t_int.data_ptr(), eps.data_ptr(), out.data_ptr(), 3, None, 0, B, H * W, 3,
             self.steps, s)
        return out, t_int, eps

    def _objective(self, x, t_int, eps):
        """(target fp32, prediction weights or None) of train.py:238-252."""
        ode, px = self.ordinary_differential_equation, self.predict_x
        if px and not ode:
            return x, None
        t = t_int.to(torch.float32)
        ad = lambda u: 0.25 * (1.0 - u / (self.steps + 1)) ** 2
        one, zero = torch.ones_like(t), torch.zeros_like(t)
        w = None
        if ode:
            a1 = ad(t - 1)
            a, c = a1.sqrt(), (1 - a1).sqrt()
        else:
            s = (1 - ad(t)).sqrt()
            a, c = zero, (s if self.predict_scaled_epsilon else one)
            if self.prediction_weighting:
                c, w = c * s, s.contiguous()
        a, c = a.contiguous(), c.contiguous()
        target = torch.empty_like(x)
        call("gct2_mix_per_image", x.data_ptr(), eps.data_ptr(), a.data_ptr(), c.data_ptr(), target.data_ptr(), x.shape[0],
             x.numel() // x.shape[0], self.net.stream())
        self.last["coef"] = (a, c, w)
        return target, w

    def enable_loss_scaling(self, initial_scale: float = 2.0 ** 15) -> None:
        if self.ls_state is None:
            if self._iterations:
                raise _lib.Gct2Error("loss scaling cannot be switched on after optimizer steps have been applied")
            self.ls_state = torch.zeros(8, dtype=torch.int32, device=self.device)
            call("gct2_loss_scale_init", self.ls_state.data_ptr(), float(initial_scale), self.net.stream())

    def train_step(self, x, t_int=None, eps=None, apply: bool = True, backward: bool = True):
        """backward=False: Trainer.call (train.py:223-272), the loss of a freshly noised batch without gradients."""
        if x.dim() != 4 or x.shape[-1] != 3:
            raise ValueError(f"expected an NHWC batch [B,H,W,3], got {tuple(x.shape)}")
        x = x.to(self.device, torch.float32).contiguous()
        B, H, W, _ = x.shape
        if H % (2 ** self.octaves) or W % (2 ** self.octaves):
            raise ValueError(f"spatial size {H}x{W} is not divisible by 2**octaves (train.py:114-119)")
        s = self.net.stream()
        n = B * H * W * 3
        if self.ls_state is not None:
            call("gct2_loss_scale_begin", self.ls_state.data_ptr(), float(self.base_lr), int(self.warm_up), float(self.beta_1), float(self.beta_2), s)
        noised, t_int, eps = self._noised(x, t_int, eps)
        pred = self.top.fwd(noised)
        target, w = self._objective(x, t_int, eps)
        if w is not None:
            call("gct2_mix_per_image", pred.data_ptr(), None, w.data_ptr(), None, pred.data_ptr(), B, H * W * 3, s)
        dpred = torch.empty_like(pred)
        ls_ptr = self.ls_state.data_ptr() if self.ls_state is not None else None
        call("gct2_mse_fwd_bwd", pred.data_ptr(), target.data_ptr(), dpred.data_ptr(), self.loss.data_ptr(), self.partials.data_ptr(), n, ls_ptr, s)
        if w is not None:
            call("gct2_mix_per_image", dpred.data_ptr(), None, w.data_ptr(), None, dpred.data_ptr(), B, H * W * 3, s)
        self.last.update(pred=pred, noised=noised)
        if not backward:
            return self.loss
        self.top.bwd(dpred)
        if apply:
            self.apply_adam()
        return self.loss

    def apply_adam(self) -> None:
        N, s = self.net, self.net.stream()
        if self.ls_state is not None:
            call("gct2_scale_check_finite", N.g.data_ptr(), N.g.numel(), self.ls_state.data_ptr(), s)
        shadow = N.op.data_ptr() if self.dtype != F32 else None
        call("gct2_adam_keras_multi", N.p.data_ptr(), N.m.data_ptr(), N.v.data_ptr(), N.g.data_ptr(), shadow, self.dtype, N.p.numel(),
             0.0 if self.ls_state is not None else self._alpha(), self.beta_1, self.beta_2, self.epsilon, 1.0,
             self.ls_state.data_ptr() if self.ls_state is not None else None, 0, s)
        if self.ls_state is not None:
            call("gct2_loss_scale_update", self.ls_state.data_ptr(), 2000, s)
        else:
            self._iterations += 1

    def predict(self, noised: torch.Tensor) -> torch.Tensor:
        return self.top.fwd(noised.to(self.device, TORCH_DTYPE[self.dtype]).contiguous())
