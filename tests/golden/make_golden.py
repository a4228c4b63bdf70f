"""Generates tests/golden/tiny_step.npz from the CPU oracle (run from the repo root:
`python tests/golden/make_golden.py`).

The reference holds no fixtures and cannot run here (no TensorFlow), so these vectors come from
this repo's own restatement (oracle/denoiser_oracle.py, fp64) after it was cross-checked against the independent
torch.nn.functional + autograd formulation (oracle/torch_cross.py): PARITY UNPINNED w.r.t. TensorFlow itself.
Contents: seeded weights / image / t_int / eps  ->  loss, prediction, every gradient, and the parameters +
Adam slots after two Keras-Adam steps, for a tiny topology (size 16, octaves 2, pixel_size 8, max_size 16, B 2).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import denoiser_oracle as O  # noqa: E402
from oracle import torch_cross as T  # noqa: E402

TINY = dict(size=16, pixel_size=8, max_size=16, octaves=2, batch_size=2)


def main():
    cfg = O.OracleConfig(**TINY)
    params = O.init_params(cfg, seed=1234)
    # non-zero biases so the bias path is exercised (the reference starts them at zero)
    rng = np.random.default_rng(99)
    for k in params:
        if k.endswith(".b"):
            params[k] = (rng.standard_normal(params[k].shape) * 0.05).astype(np.float32).astype(np.float64)
    x, t_int, eps = O.synthetic_batch(cfg, seed=0)
    loss, pred, grads, noised = O.trainer_step(params, x, t_int, eps, cfg)
    l2, p2, g2 = T.trainer_step(params, x, t_int, eps, cfg)
    assert abs(loss - l2) < 1e-12 and max(np.abs(grads[k] - g2[k]).max() for k in grads) < 1e-12
    tr = O.OracleTrainer(cfg, {k: v.copy() for k, v in params.items()})
    losses = []
    for step in range(2):
        xs, ts, es = O.synthetic_batch(cfg, seed=step)
        losses.append(tr.train_step(xs, ts, es)[0])
    out = {"x": x, "t_int": t_int, "eps": eps, "loss": np.float64(loss), "pred": pred, "noised": noised,
           "losses2": np.asarray(losses)}
    for k, v in params.items():
        out["param/" + k] = v
        out["grad/" + k] = grads[k]
        out["param2/" + k] = tr.params[k].copy()
        out["m2/" + k] = tr.m[k].copy()
        out["v2/" + k] = tr.v[k].copy()
    # a third step from the state after two: what a run resumed from (param2, m2, v2, iterations = 2) must reproduce
    xs, ts, es = O.synthetic_batch(cfg, seed=2)
    out["loss3"] = np.float64(tr.train_step(xs, ts, es)[0])
    for k in params:
        out["param3/" + k] = tr.params[k].copy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tiny_step.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
