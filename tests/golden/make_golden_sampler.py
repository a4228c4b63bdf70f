"""Generates tests/golden/tiny_sampler.npz from the CPU oracle of the reference's sampler (run from the repo root:
`python tests/golden/make_golden_sampler.py`).  PARITY UNPINNED w.r.t. TensorFlow (see make_golden.py): the vectors come
from oracle/sampler_oracle.py (fp64) with the denoiser evaluated by the independent torch formulation as a cross-check.
Contents: seeded weights / example image / example noises / dictionary -> every tensor log_sample hands to tf.summary
(train.py:323-496) for a tiny topology (size 16, octaves 2, pixel_size 8, max_size 16) with steps = 6, test_step = 2:
`out/<tensor>` for the reference's default switches, `mode/<eps|scaled_eps|ode>/<tensor>` for the other branches of
train.py:338-355, 382-413, 452-479 (predict_x = False, + predict_scaled_epsilon, ordinary_differential_equation)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import denoiser_oracle as O  # noqa: E402
from oracle import sampler_oracle as S  # noqa: E402

TINY = dict(size=16, pixel_size=8, max_size=16, octaves=2, batch_size=1)
STEPS, TEST_STEP, BITS = 6, 2, 3
MODES = {"eps": dict(predict_x=False), "scaled_eps": dict(predict_x=False, predict_scaled_epsilon=True),
         "ode": dict(ordinary_differential_equation=True)}


def inputs():
    cfg = O.OracleConfig(**TINY)
    params = O.init_params(cfg, seed=4321)
    rng = np.random.default_rng(7)
    for k in params:
        if k.endswith(".b"):
            params[k] = (rng.standard_normal(params[k].shape) * 0.05).astype(np.float32).astype(np.float64)
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    image = f32(np.floor(rng.uniform(0, 256, (1, cfg.size, cfg.size, 3))) / 128 - 1)      # loader contract, train.py:292
    example = f32(rng.standard_normal((1, 2, cfg.size, cfg.size, 3)))                     # train.py:306
    dictionary = f32(rng.standard_normal((cfg.size, cfg.size, 2 ** BITS, 3)))             # train.py:308-311
    return cfg, params, image, example, dictionary


def main():
    cfg, params, image, example, dictionary = inputs()
    res = S.log_sample(S.unet_denoiser(params, cfg), image, example, dictionary, STEPS, TEST_STEP)
    out = {"example_image": image, "example": example, "dictionary": dictionary}
    for k, v in params.items():
        out["param/" + k] = v
    for k, v in res.items():
        out["out/" + k] = np.asarray(v)
    for name, kw in MODES.items():
        for k, v in S.log_sample(S.unet_denoiser(params, cfg), image, example, dictionary, STEPS, TEST_STEP, **kw).items():
            out[f"mode/{name}/{k}"] = np.asarray(v)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tiny_sampler.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
