"""CPU ORACLE (test infrastructure, NOT product code) for the reference's off-by-default model variants.

Restates /root/reference/train.py with its switches honoured:
    block_depth > 0   Block = block_depth x Conv2D(filters, 3, 1, 'same', relu)                     train.py:20, 123-143
    residual = True   Residual.call = input + Dense(input_channels, use_bias=False)(module(input))  train.py:26, 104-112
    concat = False    Residual.call = module(input)                                                 train.py:27, 120-121
and the objective switches of train.py:29-32, 238-252 (through denoiser_oracle.objective_terms).

numpy, fp64; the reverse pass is a small tape of hand-written vector-Jacobian products (one closure per operation), the
convolutions reuse denoiser_oracle's 4x4/stride-2 forms and add the 'same' stride-1 form below.  PARITY UNPINNED like the rest
(no TensorFlow, no fixtures in the reference): pinned by the definition-level loop `naive_conv_s1` and by an independent
torch.nn.functional + autograd evaluation in tests/test_oracle_cpu.py.  Only tests/ may import this module.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import numpy as np

from . import denoiser_oracle as O


# ---- stride-1 'same' convolution, odd kernel size (train.py:131-139; [TF] 'same': symmetric padding (KS-1)/2) -------------------
def conv_s1_fwd(x, w, b=None):
    B, H, W, C = x.shape
    KS = w.shape[0]
    p = (KS - 1) // 2
    xp = np.zeros((B, H + 2 * p, W + 2 * p, C), dtype=x.dtype)
    xp[:, p:p + H, p:p + W, :] = x
    z = np.zeros((B, H, W, w.shape[3]), dtype=x.dtype)
    for kh in range(KS):
        for kw in range(KS):
            z += xp[:, kh:kh + H, kw:kw + W, :] @ w[kh, kw]
    return z if b is None else z + b


def conv_s1_bwd(x, w, dz):
    B, H, W, C = x.shape
    KS = w.shape[0]
    p = (KS - 1) // 2
    xp = np.zeros((B, H + 2 * p, W + 2 * p, C), dtype=x.dtype)
    xp[:, p:p + H, p:p + W, :] = x
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(w)
    dz2 = dz.reshape(-1, dz.shape[-1])
    for kh in range(KS):
        for kw in range(KS):
            dw[kh, kw] = xp[:, kh:kh + H, kw:kw + W, :].reshape(-1, C).T @ dz2
            dxp[:, kh:kh + H, kw:kw + W, :] += dz @ w[kh, kw].T
    return dxp[:, p:p + H, p:p + W, :], dw, dz2.sum(0)


def naive_conv_s1(x, w, b):
    """definition-level loops (tiny shapes only)."""
    B, H, W, C = x.shape
    KS, Oc = w.shape[0], w.shape[3]
    p = (KS - 1) // 2
    z = np.zeros((B, H, W, Oc))
    for n in range(B):
        for h in range(H):
            for ww in range(W):
                for o in range(Oc):
                    acc = float(b[o])
                    for kh in range(KS):
                        for kw in range(KS):
                            hh, wq = h + kh - p, ww + kw - p
                            if 0 <= hh < H and 0 <= wq < W:
                                for i in range(C):
                                    acc += float(x[n, hh, wq, i]) * float(w[kh, kw, i, o])
                    z[n, h, ww, o] = acc
    return z


# ---- the network with every switch (train.py:175-204) ------------------------------------------------------------------------
def variant_param_shapes(cfg: O.OracleConfig, block_depth: int, residual: bool, concat: bool) -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) in forward order; the names are those of gan_class_transfer2_amd.variants.build_structure."""
    specs: List[Tuple[str, Tuple[int, ...]]] = []

    def block(name, cin, filters):
        c = cin
        for d in range(block_depth):
            specs.append((f"{name}.{d}.w", (3, 3, c, filters)))
            specs.append((f"{name}.{d}.b", (filters,)))
            c = filters
        return c

    def level(i, cin):
        f = cfg.down_filters(i)
        specs.append((f"D{i}.w", (4, 4, cin, f))); specs.append((f"D{i}.b", (f,)))
        c = block(f"blkA{i}", f, f)
        if i + 1 < cfg.octaves:
            c = level(i + 1, c)
        else:
            c = block("blkMid", c, min(cfg.pixel_size * 2 ** cfg.octaves, cfg.max_size))
        c = block(f"blkB{i}", c, f)
        fu = cfg.up_filters(i)
        specs.append((f"U{i}.w", (4, 4, fu, c))); specs.append((f"U{i}.b", (fu,)))
        if residual:
            specs.append((f"res{i}.dense.w", (fu, cin)))
            return cin
        return fu + cin if concat else fu

    c = block("blkTopA", 3, cfg.pixel_size)
    c = level(0, c)
    c = block("blkTopB", c, cfg.pixel_size)
    specs.append(("dense.w", (c, 3))); specs.append(("dense.b", (3,)))
    return specs


def init_variant_params(cfg, block_depth, residual, concat, seed=1234):
    rng = np.random.default_rng(seed)
    params = {}
    for name, shp in variant_param_shapes(cfg, block_depth, residual, concat):
        if name.endswith(".b"):
            params[name] = (rng.standard_normal(shp) * 0.05).astype(np.float32).astype(np.float64)   # non-zero: exercises the bias path
        else:
            lim = O.glorot_limit(shp)
            params[name] = rng.uniform(-lim, lim, size=shp).astype(np.float32).astype(np.float64)
    return params


def variant_forward_backward(params, x0, cfg, block_depth, residual, concat, dpred_fn: Callable, operand_round: Optional[str] = None,
                             round_head: bool = False):
    """forward through the nested structure recording a tape; dpred_fn(pred) -> (loss, dpred); returns (loss, pred, grads).

    operand_round in {None, 'bf16', 'f16'} models what variants.VariantEngine keeps in HBM in a 16-bit mode (the same model as
    denoiser_oracle.unet_forward / unet_backward): convolution / projection kernels are consumed as rounded operands (the Dense(3)
    kernel stays fp32 in the forward pass), every stored activation is rounded once, every stored activation gradient is rounded
    once, and a gradient that is the SUM of two stored tensors (skip + module path of Residual, gct2_add) is rounded again;
    accumulation stays in the array dtype.  round_head (the f16 policy): the Dense(3) output and the gradient entering it are fp16."""
    rnd = O._rounder(operand_round)
    tape: List[Callable[[], None]] = []
    grads: Dict[str, np.ndarray] = {}
    wq = {k: (rnd(v) if k.endswith(".w") and k != "dense.w" else v) for k, v in params.items()}

    class V:   # a value with its gradient slot
        def __init__(self, val):
            self.val, self.grad = val, np.zeros_like(val)

        def add_grad(self, c):
            """one more stored gradient tensor flows in: stored rounded, and the running sum is what the add kernel stores"""
            self.grad = rnd(self.grad + rnd(c))

    def conv(kind, name, v: V) -> V:
        w, b = wq[name + ".w"], params[name + ".b"]
        fwd = {"down": O.conv4s2_fwd, "up": O.convT4s2_fwd, "c3": conv_s1_fwd}[kind]
        bwd = {"down": O.conv4s2_bwd, "up": O.convT4s2_bwd, "c3": conv_s1_bwd}[kind]
        out = V(rnd(np.maximum(fwd(v.val, w, b), 0)))                         # activation='relu' (train.py:134,150,163)

        def back():
            dz = rnd(out.grad * (out.val > 0))
            dx, grads[name + ".w"], grads[name + ".b"] = bwd(v.val, w, dz)
            v.add_grad(dx)
        tape.append(back)
        return out

    def block(name, v: V) -> V:
        for d in range(block_depth):
            v = conv("c3", f"{name}.{d}", v)
        return v

    def level(i, v: V) -> V:
        h = conv("down", f"D{i}", v)
        h = block(f"blkA{i}", h)
        h = level(i + 1, h) if i + 1 < cfg.octaves else block("blkMid", h)
        h = block(f"blkB{i}", h)
        h = conv("up", f"U{i}", h)
        if residual:                                                           # train.py:111-112
            wd = wq[f"res{i}.dense.w"]
            out = V(rnd(v.val + rnd(h.val @ wd)))

            def back():
                grads[f"res{i}.dense.w"] = h.val.reshape(-1, h.val.shape[-1]).T @ out.grad.reshape(-1, out.grad.shape[-1])
                h.add_grad(out.grad @ wd.T)
                v.add_grad(out.grad)
            tape.append(back)
            return out
        if concat:                                                             # train.py:113-119
            out = V(np.concatenate([h.val, v.val], -1))
            cm = h.val.shape[-1]

            def back():
                h.add_grad(out.grad[..., :cm])
                v.add_grad(out.grad[..., cm:])
            tape.append(back)
            return out
        return h                                                               # train.py:120-121

    vin = V(rnd(x0))
    h = block("blkTopA", vin)
    h = level(0, h)
    h = block("blkTopB", h)
    pred = h.val @ params["dense.w"] + params["dense.b"]                      # Dense(3), linear (train.py:198-202)
    if round_head:
        pred = O.round_f16(pred)
    loss, dpred = dpred_fn(pred)
    if round_head:
        dpred = O.round_f16(dpred)
    grads["dense.w"] = h.val.reshape(-1, h.val.shape[-1]).T @ dpred.reshape(-1, 3)
    grads["dense.b"] = dpred.reshape(-1, 3).sum(0)
    h.add_grad(rnd(dpred) @ rnd(params["dense.w"]).T)                        # the input gradient runs through the 1 x 1 convolution entry
    for back in reversed(tape):
        back()
    return loss, pred, grads


def variant_trainer_step(params, x, t_int, eps, cfg, block_depth=0, residual=False, concat=True, objective: Optional[dict] = None,
                         operand_round: Optional[str] = None, loss_scale: float = 1.0):
    """Trainer.call (train.py:223-272) on the variant network: (loss, pred, grads).  operand_round: see variant_forward_backward;
    'f16' also takes the fp16 noising arithmetic and the two fp16 points of the Dense head (denoiser_oracle.trainer_step), and
    the returned gradients are the SCALED ones (loss_scale, train.py:82-83)."""
    f16 = operand_round == "f16"
    noised = O.noise_image_f16(x, t_int, eps, cfg.steps).astype(x.dtype) if f16 else O.noise_image(x, t_int, eps, cfg.steps)
    target, w = O.objective_terms(x, t_int, eps, cfg.steps, **(objective or {}))

    def dpred_fn(pred):
        diff = pred * w - target
        return float(np.mean(diff ** 2)), (2.0 * loss_scale / diff.size) * diff * w
    return variant_forward_backward(params, noised, cfg, block_depth, residual, concat, dpred_fn, operand_round, round_head=f16)
