"""TEST INFRASTRUCTURE ONLY - CPU restatement (numpy, fp64) of the reference's sampler `log_sample`
(/root/reference/train.py:323-496) with every objective switch it reads (predict_x, predict_scaled_epsilon,
ordinary_differential_equation; train.py:29-32).
Only tests/ and __graft_entry__.smoke() may import this; the product path never does.

PARITY UNPINNED: TensorFlow is absent and the reference holds no fixtures; this follows the text of
train.py line by line and is cross-checked by the identities in tests/test_oracle_cpu.py.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import numpy as np

from . import denoiser_oracle as O


def noise_edits(eps: np.ndarray, dictionary: np.ndarray) -> np.ndarray:
    """train.py:416-431.  eps [1,H,W,C], dictionary [H,W,K,C] -> [4,H,W,C] = eps | pixelated | shifted | quantised."""
    _, H, W, C = eps.shape
    # tf.nn.avg_pool2d(eps, 4, 4, 'SAME') then UpSampling2D(4, 'nearest')   (train.py:416-418); H, W multiples of 4
    pooled = eps.reshape(1, H // 4, 4, W // 4, 4, C).mean(axis=(2, 4))
    pixelated = np.repeat(np.repeat(pooled, 4, axis=1), 4, axis=2)
    shifted = np.roll(np.roll(eps, 1, 1), 1, 2)                                      # train.py:420
    err = ((eps[..., None, :] - dictionary[None]) ** 2).sum(-1)                      # train.py:422-424  [1,H,W,K]
    idx = err.argmin(-1)                                                             # first minimum, like tf.argmin
    quantised = np.take_along_axis(dictionary[None], idx[..., None, None], axis=3)[..., 0, :]   # train.py:425-428
    return np.concatenate([eps, pixelated, shifted, quantised], 0)                   # train.py:430


def log_sample(denoise: Callable[[np.ndarray], np.ndarray], example_image: np.ndarray, example: np.ndarray,
               dictionary: np.ndarray, steps: int = 200, test_step: int = 25, predict_x: bool = True,
               predict_scaled_epsilon: bool = False, ordinary_differential_equation: bool = False) -> Dict[str, np.ndarray]:
    """denoise(x [B,H,W,3]) -> prediction, standing for denoiser((x, t)) (t is ignored, train.py:208-210).
    example_image [1,H,W,3]; example [1,2,H,W,3] (train.py:305-306); dictionary [H,W,K,3] (train.py:309-311).
    The three switches are the module globals of train.py:29-32 that log_sample reads (prediction_weighting is not among them)."""
    a = lambda t: float(O.alpha_dash(t, steps))
    ode, px, pse = ordinary_differential_equation, predict_x, predict_scaled_epsilon
    out: Dict[str, np.ndarray] = {}
    image = example_image[0][None]
    # single-shot denoising (train.py:325-361)
    f = a(test_step)
    if ode:
        f = a(steps / 2) ** 0.5                                                       # train.py:326-328
    noised = image * f ** 0.5 + example[0, :1] * (1 - f) ** 0.5
    prediction = denoise(noised)
    if ode:                                                                           # train.py:338-347
        denoised = (prediction * (1 - a(steps / 2)) ** 0.5 - noised * (1 - a(steps / 2 - 1)) ** 0.5) / (
            a(steps / 2 - 1) ** 0.5 * (1 - a(steps / 2)) ** 0.5 - a(steps / 2) ** 0.5 * (1 - a(steps / 2 - 1)) ** 0.5)
    elif px:                                                                          # train.py:348-349
        denoised = prediction
    else:                                                                             # train.py:350-355
        if not pse:
            prediction = prediction * (1 - f) ** 0.5
        denoised = (noised - prediction) / f ** 0.5
    out["denoised"] = denoised
    out["example_loss"] = np.sqrt(np.mean((image - denoised) ** 2))

    def update(prediction, fake, x_theta, eps_theta, t):
        """train.py:382-413 == 452-479 (the two loops carry the same update; their dead `fake = ...` lines differ only)."""
        if ode:
            x_theta = (prediction * (1 - a(t)) ** 0.5 - fake * (1 - a(t - 1)) ** 0.5) / (
                a(t - 1) ** 0.5 * (1 - a(t)) ** 0.5 - a(t) ** 0.5 * (1 - a(t - 1)) ** 0.5)
            return x_theta, eps_theta                                                 # epsilon_theta is never reassigned
        if px:
            x_theta = prediction
            eps_theta = (fake - a(t) ** 0.5 * x_theta) / (1 - a(t)) ** 0.5
            return x_theta, eps_theta
        if pse:
            eps_theta = prediction / (1 - a(t)) ** 0.5
            scaled = prediction
        else:
            eps_theta = prediction
            scaled = prediction * (1 - a(t)) ** 0.5
        return (fake - scaled) / a(t) ** 0.5, eps_theta

    # forward diffusion: invert the example image into noise (train.py:364-413)
    x_theta = image
    eps_theta = x_theta
    for t in reversed(range(steps, 0, -1)):          # 1 .. steps
        fake = a(t) ** 0.5 * x_theta + (1 - a(t)) ** 0.5 * eps_theta
        x_theta, eps_theta = update(denoise(fake), fake, x_theta, eps_theta, t)
    out["epsilon_theta"] = eps_theta
    # backward diffusion from the two random noises and the four edits of the inverted one (train.py:415-496)
    fake = np.concatenate([example[0], noise_edits(eps_theta, dictionary)], 0)
    x_theta = fake
    eps_theta = fake
    for t in range(steps, 0, -1):
        fake = a(t) ** 0.5 * x_theta + (1 - a(t)) ** 0.5 * eps_theta
        x_theta, eps_theta = update(denoise(fake), fake, x_theta, eps_theta, t)
        if t == steps:
            out["step_1"] = x_theta
        if t == steps // 4:
            out["step_0.25"] = x_theta
        if t == 2 * steps // 4:
            out["step_0.5"] = x_theta
        if t == 3 * steps // 4:
            out["step_0.75"] = x_theta
    out["fake"] = x_theta
    return out


def unet_denoiser(params, cfg: O.OracleConfig, operand_round: Optional[str] = None) -> Callable[[np.ndarray], np.ndarray]:
    return lambda x: O.unet_forward(params, x, cfg, operand_round)[0]


def decode_contract(image_u8: np.ndarray, oy: int, ox: int, flip: bool, size: int) -> np.ndarray:
    """train.py:288-292 after the decoder: crop [size, size, 3] at (oy, ox), optional left-right flip, value / 128 - 1."""
    out = image_u8[oy:oy + size, ox:ox + size, :]
    if flip:
        out = out[:, ::-1, :]
    return out.astype(np.float64) / 128 - 1
