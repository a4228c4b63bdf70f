"""GPU parity of the sampler (train.py:323-496): the three pointwise kernels through the C ABI against
numpy, and gan_class_transfer2_amd.sampler.log_sample against oracle/sampler_oracle.py on the committed fixture."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import denoiser_oracle as O
from oracle import sampler_oracle as S

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_sampler.npz")
TDT = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}


def lib():
    import gan_class_transfer2_amd as g
    return g._lib


def stream():
    return torch.cuda.current_stream().cuda_stream


def rel_l2(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("dt", [0, 1, 2])
def test_diffusion_mix_and_update(gpu, dt):
    rng = np.random.default_rng(21)
    npix, C, ld = 1000, 3, 8
    x, e = rng.standard_normal((npix, C)).astype(np.float32), rng.standard_normal((npix, C)).astype(np.float32)
    a = float(O.alpha_dash(37, 200))
    xd, ed = torch.tensor(x, device=gpu), torch.tensor(e, device=gpu)
    fake = torch.zeros(npix, C, device=gpu)
    out = torch.full((npix, ld), 5.0, dtype=TDT[dt], device=gpu); out2 = torch.zeros(npix, 4, dtype=TDT[dt], device=gpu)
    lib().call("gct2_diffusion_mix", dt, xd.data_ptr(), ed.data_ptr(), a, fake.data_ptr(), out.data_ptr() + 2 * out.element_size(), ld,
               out2.data_ptr(), 4, npix, C, stream())
    torch.cuda.synchronize()
    ref = a ** 0.5 * x.astype(np.float64) + (1 - a) ** 0.5 * e
    assert rel_l2(fake.cpu().numpy(), ref) <= 2e-7
    assert torch.equal(out[:, 2:5], fake.to(TDT[dt])) and torch.equal(out2[:, :3], out[:, 2:5])
    assert float((out[:, :2].float() - 5).abs().max()) == 0 and float((out[:, 5:].float() - 5).abs().max()) == 0
    pred = rng.standard_normal((npix, C)).astype(np.float32)
    pd_ = torch.tensor(pred, device=gpu)
    xt, et = torch.zeros_like(pd_), torch.zeros_like(pd_)
    lib().call("gct2_diffusion_update", pd_.data_ptr(), fake.data_ptr(), a, xt.data_ptr(), et.data_ptr(), pd_.numel(), stream())
    torch.cuda.synchronize()
    assert torch.equal(xt, pd_)
    assert rel_l2(et.cpu().numpy(), (fake.cpu().numpy().astype(np.float64) - a ** 0.5 * pred) / (1 - a) ** 0.5) <= 2e-7
    # the predict_x invariant: sqrt(a) x_theta + sqrt(1-a) eps_theta reproduces fake
    back = a ** 0.5 * xt.double() + (1 - a) ** 0.5 * et.double()
    assert rel_l2(back.cpu().numpy(), fake.cpu().numpy()) <= 5e-7


@pytest.mark.parametrize("shape", [(8, 12, 8), (16, 16, 8), (4, 4, 3)])
def test_noise_edits(gpu, shape):
    H, W, K = shape
    rng = np.random.default_rng(22)
    eps = rng.standard_normal((1, H, W, 3)).astype(np.float32)
    dic = rng.standard_normal((H, W, K, 3)).astype(np.float32)
    out = torch.zeros(4, H, W, 3, device=gpu)
    ed, dd = torch.tensor(eps, device=gpu), torch.tensor(dic, device=gpu)
    lib().call("gct2_noise_edits", ed.data_ptr(), dd.data_ptr(), K, out.data_ptr(), H, W, 3, stream())
    torch.cuda.synchronize()
    ref = S.noise_edits(eps.astype(np.float64), dic.astype(np.float64))
    got = out.cpu().numpy()
    assert np.array_equal(got[0], eps[0]) and np.array_equal(got[2], ref[2].astype(np.float32))
    assert np.abs(got[1] - ref[1]).max() <= 1e-6
    assert np.array_equal(got[3], ref[3].astype(np.float32))         # same argmin (no near-ties in random data), entries copied
    with pytest.raises(lib().Gct2Error):
        lib().call("gct2_noise_edits", ed.data_ptr(), dd.data_ptr(), K, out.data_ptr(), H + 1, W, 3, stream())


def _engine_for_fixture(gpu, dtype, z):
    import gan_class_transfer2_amd as g
    cfg = O.OracleConfig(size=16, pixel_size=8, max_size=16, octaves=2, batch_size=1)
    eng = g.UNetEngine(g.Topology(cfg.pixel_size, cfg.max_size, cfg.octaves), dtype, gpu, steps=6)
    eng.set_params({k[6:]: z[k] for k in z.files if k.startswith("param/")})
    return cfg, eng


def test_log_sample_fp32_against_golden(gpu):
    """fp32 mode (the reference's default arithmetic) against the fp64 oracle fixture: 1 + 6 + 6 network evaluations."""
    import gan_class_transfer2_amd as g
    z = np.load(GOLDEN)
    cfg, eng = _engine_for_fixture(gpu, 0, z)
    den = types.SimpleNamespace(ensure_engine=lambda: eng)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=gpu)
    res = g.log_sample(den, t(z["example_image"]), t(z["example"]), t(z["dictionary"]), steps=6, test_step=2)
    torch.cuda.synchronize()
    assert set(res) == {k[4:] for k in z.files if k.startswith("out/")}
    for k, v in res.items():
        ref = z["out/" + k]
        assert tuple(v.shape) == (ref.shape if ref.shape else (1,)), k
        assert rel_l2(v.cpu().numpy().reshape(ref.shape), ref) <= 2e-5, k


def test_log_sample_bf16_against_rounded_oracle(gpu):
    """bf16 operands: compared with the oracle evaluating the denoiser under the same rounding model; the sampler state is
    fp32 on both sides, so only the accumulation order of each network evaluation differs, amplified by the 12 steps."""
    import gan_class_transfer2_amd as g
    z = np.load(GOLDEN)
    cfg, eng = _engine_for_fixture(gpu, 1, z)
    params = {k[6:]: z[k] for k in z.files if k.startswith("param/")}
    ref = S.log_sample(S.unet_denoiser(params, cfg, "bf16"), z["example_image"], z["example"], z["dictionary"], 6, 2)
    den = types.SimpleNamespace(ensure_engine=lambda: eng)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=gpu)
    res = g.log_sample(den, t(z["example_image"]), t(z["example"]), t(z["dictionary"]), steps=6, test_step=2)
    torch.cuda.synchronize()
    for k in ("denoised", "epsilon_theta", "step_1", "fake"):
        assert rel_l2(res[k].cpu().numpy(), ref[k]) <= 3e-2, k
