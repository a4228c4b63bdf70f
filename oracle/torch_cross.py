"""CPU ORACLE, second derivation (test infrastructure, NOT product code).

The same `Trainer` step as `denoiser_oracle.py`, but written with torch.nn.functional and
autograd on CPU tensors: an independent formulation (library convolutions + automatic
differentiation) used to cross-check the hand-derived numpy restatement, and - being
multi-threaded - the timed `cpu_baseline` ("port") in bench.py.

PARITY UNPINNED (no TensorFlow here, the reference holds no fixtures): see denoiser_oracle.py.

Keras <-> torch weight layouts:
  Conv2D          (kh,kw,Cin,Cout)  -> conv2d weight            (Cout,Cin,kh,kw), padding=1
  Conv2DTranspose (kh,kw,Cout,Cin)  -> conv_transpose2d weight  (Cin,Cout,kh,kw), padding=1
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from .denoiser_oracle import OracleConfig, alpha_dash, param_names, warmup_lr


def to_torch_params(params: Dict[str, np.ndarray], dtype=torch.float64, requires_grad=True):
    out = {}
    for k, v in params.items():
        t = torch.tensor(np.asarray(v), dtype=dtype)
        out[k] = t.requires_grad_(requires_grad)
    return out


def forward(tp: Dict[str, torch.Tensor], noised: torch.Tensor, cfg: OracleConfig) -> torch.Tensor:
    """noised: [B,H,W,3] (NHWC like the reference); returns pred [B,H,W,3]."""
    n = cfg.octaves
    x = noised.permute(0, 3, 1, 2)                      # NCHW for torch
    xs = [x]
    for i in range(n):
        w = tp[f"D{i}.w"].permute(3, 2, 0, 1)
        xs.append(F.relu(F.conv2d(xs[i], w, tp[f"D{i}.b"], stride=2, padding=1)))
    r = xs[n]
    for i in reversed(range(n)):
        w = tp[f"U{i}.w"].permute(3, 2, 0, 1)           # (Cin,Cout,kh,kw)
        u = F.relu(F.conv_transpose2d(r, w, tp[f"U{i}.b"], stride=2, padding=1))
        r = torch.cat([u, xs[i]], dim=1)                # module output FIRST (train.py:114-119)
    r = r.permute(0, 2, 3, 1)
    return r @ tp["dense.w"] + tp["dense.b"]


def trainer_step(params, x, t_int, eps, cfg: OracleConfig, dtype=torch.float64):
    tp = to_torch_params(params, dtype)
    xt = torch.tensor(np.asarray(x), dtype=dtype)
    et = torch.tensor(np.asarray(eps), dtype=dtype)
    a = torch.tensor(alpha_dash(np.asarray(t_int), cfg.steps), dtype=dtype).reshape(-1, 1, 1, 1)
    noised = xt * a.sqrt() + et * (1 - a).sqrt()
    pred = forward(tp, noised, cfg)
    loss = torch.mean((xt.float().to(dtype) - pred) ** 2)
    loss.backward()
    grads = {k: v.grad.detach().numpy() for k, v in tp.items()}
    return float(loss.detach()), pred.detach().numpy(), grads


class TorchCpuTrainer:
    """fp32 multi-threaded CPU train step (forward, MSE, backward, Keras Adam + WarmUp):
    the timed CPU baseline.  Same arithmetic as OracleTrainer.train_step."""

    def __init__(self, cfg: OracleConfig, params: Dict[str, np.ndarray]):
        self.cfg = cfg
        self.tp = to_torch_params(params, torch.float32)
        self.m = {k: torch.zeros_like(v) for k, v in self.tp.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.tp.items()}
        self.iterations = 0
        self.gen = torch.Generator().manual_seed(0)

    def train_step(self, x: torch.Tensor, t_int=None, eps=None) -> float:
        cfg = self.cfg
        B = x.shape[0]
        if t_int is None:
            t_int = torch.randint(1, cfg.steps + 1, (B,), generator=self.gen)
        if eps is None:
            eps = torch.randn(x.shape, generator=self.gen)
        a = (0.25 * (1 - t_int.float() / (cfg.steps + 1)) ** 2).reshape(-1, 1, 1, 1)
        noised = x * a.sqrt() + eps * (1 - a).sqrt()
        for p in self.tp.values():
            p.grad = None
        pred = forward(self.tp, noised, cfg)
        loss = torch.mean((x - pred) ** 2)
        loss.backward()
        k = self.iterations
        t = k + 1
        lr = warmup_lr(k, cfg.base_lr, cfg.warm_up)
        alpha = lr * math.sqrt(1 - cfg.beta_2 ** t) / (1 - cfg.beta_1 ** t)
        with torch.no_grad():
            for name in param_names(cfg):
                p, g = self.tp[name], self.tp[name].grad
                m, v = self.m[name], self.v[name]
                m.mul_(cfg.beta_1).add_(g, alpha=1 - cfg.beta_1)
                v.mul_(cfg.beta_2).addcmul_(g, g, value=1 - cfg.beta_2)
                p.addcdiv_(m, v.sqrt().add_(cfg.epsilon), value=-alpha)
        self.iterations += 1
        return float(loss.detach())
