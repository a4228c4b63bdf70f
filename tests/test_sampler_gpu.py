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
    lib().call("gct2_diffusion_update", lib().SAMPLE_X, pd_.data_ptr(), fake.data_ptr(), a, 0.0, xt.data_ptr(), et.data_ptr(), pd_.numel(), stream())
    torch.cuda.synchronize()
    assert torch.equal(xt, pd_)
    assert rel_l2(et.cpu().numpy(), (fake.cpu().numpy().astype(np.float64) - a ** 0.5 * pred) / (1 - a) ** 0.5) <= 2e-7
    # the predict_x invariant: sqrt(a) x_theta + sqrt(1-a) eps_theta reproduces fake
    back = a ** 0.5 * xt.double() + (1 - a) ** 0.5 * et.double()
    assert rel_l2(back.cpu().numpy(), fake.cpu().numpy()) <= 5e-7
    # the other objectives of train.py:382-413 (epsilon, scaled epsilon, ODE), each against the reference's formulas in fp64
    f64, p64 = fake.cpu().numpy().astype(np.float64), pred.astype(np.float64)
    a1 = float(O.alpha_dash(36, 200))
    L = lib()
    for mode, x_ref, e_ref in (
            (L.SAMPLE_EPS, (f64 - p64 * (1 - a) ** 0.5) / a ** 0.5, p64),
            (L.SAMPLE_SCALED_EPS, (f64 - p64) / a ** 0.5, p64 / (1 - a) ** 0.5),
            (L.SAMPLE_ODE, (p64 * (1 - a) ** 0.5 - f64 * (1 - a1) ** 0.5) / (a1 ** 0.5 * (1 - a) ** 0.5 - a ** 0.5 * (1 - a1) ** 0.5), None)):
        xt.fill_(7.0); et.fill_(7.0)
        L.call("gct2_diffusion_update", mode, pd_.data_ptr(), fake.data_ptr(), a, a1, xt.data_ptr(), et.data_ptr() if e_ref is not None else None,
               pd_.numel(), stream())
        torch.cuda.synchronize()
        assert rel_l2(xt.cpu().numpy(), x_ref) <= 5e-6, mode         # the ODE quotient subtracts nearly equal products
        if e_ref is None:
            assert float((et - 7.0).abs().max()) == 0                 # epsilon_theta is not touched in ODE mode
        else:
            assert rel_l2(et.cpu().numpy(), e_ref) <= 2e-7, mode
    with pytest.raises(L.Gct2Error):
        L.call("gct2_diffusion_update", 4, pd_.data_ptr(), fake.data_ptr(), a, a1, xt.data_ptr(), et.data_ptr(), pd_.numel(), stream())
    with pytest.raises(L.Gct2Error):                                   # alpha == alpha_prev: singular ODE step
        L.call("gct2_diffusion_update", L.SAMPLE_ODE, pd_.data_ptr(), fake.data_ptr(), a, a, xt.data_ptr(), None, pd_.numel(), stream())


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


SWITCHES = {"default": {}, "eps": dict(predict_x=False), "scaled_eps": dict(predict_x=False, predict_scaled_epsilon=True),
            "ode": dict(ordinary_differential_equation=True)}


def _engine_for_fixture(gpu, dtype, z, **switches):
    import gan_class_transfer2_amd as g
    cfg = O.OracleConfig(size=16, pixel_size=8, max_size=16, octaves=2, batch_size=1)
    eng = g.UNetEngine(g.Topology(cfg.pixel_size, cfg.max_size, cfg.octaves), dtype, gpu, steps=6, **switches)
    eng.set_params({k[6:]: z[k] for k in z.files if k.startswith("param/")})
    return cfg, eng


def _fixture_outputs(z, mode):
    prefix = "out/" if mode == "default" else f"mode/{mode}/"
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}


@pytest.mark.parametrize("mode", list(SWITCHES))
def test_log_sample_fp32_against_golden(gpu, mode, parity_log):
    """fp32 mode (the reference's default arithmetic) against the fp64 oracle fixture, for every objective branch of
    train.py:338-355, 382-413, 452-479: 1 + 6 + 6 network evaluations."""
    import gan_class_transfer2_amd as g
    z = np.load(GOLDEN)
    cfg, eng = _engine_for_fixture(gpu, 0, z, **SWITCHES[mode])
    den = types.SimpleNamespace(ensure_engine=lambda: eng)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=gpu)
    res = g.log_sample(den, t(z["example_image"]), t(z["example"]), t(z["dictionary"]), steps=6, test_step=2, **SWITCHES[mode])
    torch.cuda.synchronize()
    want = _fixture_outputs(z, mode)
    assert set(res) == set(want)
    tol = 2e-5        # measured (profiles/r03_parity.json): 1e-7 .. 3e-7 in every mode
    errs = {}
    for k, v in res.items():
        ref = want[k]
        assert tuple(v.shape) == (ref.shape if ref.shape else (1,)), k
        errs[k] = rel_l2(v.cpu().numpy().reshape(ref.shape), ref)
    parity_log(f"log_sample_fp32_{mode}", **errs)
    for k, e in errs.items():
        assert e <= tol, (mode, k, e)


def test_log_sample_refuses_mismatched_switches(gpu):
    """a network trained for one objective sampled with another would give wrong images without an error (r02 did): refused."""
    import gan_class_transfer2_amd as g
    z = np.load(GOLDEN)
    cfg, eng = _engine_for_fixture(gpu, 0, z, predict_x=False)
    den = types.SimpleNamespace(ensure_engine=lambda: eng)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=gpu)
    with pytest.raises(ValueError):
        g.log_sample(den, t(z["example_image"]), t(z["example"]), t(z["dictionary"]), steps=6, test_step=2, predict_x=True)


@pytest.mark.parametrize("mode", ["default", "eps"])
def test_log_sample_bf16_against_rounded_oracle(gpu, mode):
    """bf16 operands: compared with the oracle evaluating the denoiser under the same rounding model; the sampler state is
    fp32 on both sides, so only the accumulation order of each network evaluation differs, amplified by the 12 steps."""
    import gan_class_transfer2_amd as g
    z = np.load(GOLDEN)
    cfg, eng = _engine_for_fixture(gpu, 1, z, **SWITCHES[mode])
    params = {k[6:]: z[k] for k in z.files if k.startswith("param/")}
    ref = S.log_sample(S.unet_denoiser(params, cfg, "bf16"), z["example_image"], z["example"], z["dictionary"], 6, 2, **SWITCHES[mode])
    den = types.SimpleNamespace(ensure_engine=lambda: eng)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=gpu)
    res = g.log_sample(den, t(z["example_image"]), t(z["example"]), t(z["dictionary"]), steps=6, test_step=2, **SWITCHES[mode])
    torch.cuda.synchronize()
    for k in ("denoised", "epsilon_theta", "step_1", "fake"):
        assert rel_l2(res[k].cpu().numpy(), ref[k]) <= (3e-2 if mode == "default" else 6e-2), (mode, k)


def test_log_sample_graph_replay_equals_plain_launches(gpu):
    """the HIP-graph replay of the forward pass against plain launches: bit-identical, also with a train step between two calls
    (new weights, same graphs) and after the call context was re-tuned (the cached graphs bake in its tile choices: they must be
    dropped, not replayed)."""
    import gan_class_transfer2_amd as g
    z = np.load(GOLDEN)
    cfg, eng = _engine_for_fixture(gpu, 1, z)
    den = types.SimpleNamespace(ensure_engine=lambda: eng)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=gpu)
    args = (t(z["example_image"]), t(z["example"]), t(z["dictionary"]))

    def both():
        a = g.log_sample(den, *args, steps=6, test_step=2, use_graph=True)
        b = g.log_sample(den, *args, steps=6, test_step=2, use_graph=False)
        torch.cuda.synchronize()
        for k in a:
            assert torch.equal(a[k], b[k]), k
        return a

    first = both()
    assert len(eng._forward_graphs) == 2                         # batch 1 and batch 6
    x = torch.tensor(np.floor(np.random.default_rng(3).uniform(0, 256, (2, 16, 16, 3))) / 128 - 1, dtype=torch.float32, device=gpu)
    eng.base_lr, eng.warm_up = 1e-2, 0                           # a visible update
    eng.train_step(x)
    second = both()                                              # replays the graphs captured before the step
    assert not torch.equal(first["fake"], second["fake"])
    version = eng.ctx.version
    eng.ctx.set_tuning(2 | (1 << 24))                            # 128 x 128 tiles for every layer, no halo kernel
    assert eng.ctx.version != version
    third = both()
    assert all(k[1] == eng.ctx.version for k in eng._forward_graphs)
    for k in second:                                             # same arithmetic per output element whatever the tile: same bits? not
        assert rel_l2(third[k].cpu().numpy(), second[k].cpu().numpy()) <= 2e-2, k   # promised; only graph == plain is (checked in both())


@pytest.mark.parametrize("variant", [dict(block_depth=1, residual=False, concat=True), dict(block_depth=0, residual=True, concat=False)])
def test_log_sample_on_a_variant_network(gpu, variant):
    """block_depth > 0 / residual=True networks run on VariantEngine, which has no planned buffers: r02's sampler died with an
    AttributeError there.  fp32 against the variant oracle's forward pass inside the sampler oracle."""
    import gan_class_transfer2_amd as g
    from gan_class_transfer2_amd.variants import VariantEngine
    from oracle import variants_oracle as V
    cfg = O.OracleConfig(size=16, pixel_size=8, max_size=16, octaves=2, batch_size=1)
    params = V.init_variant_params(cfg, variant["block_depth"], variant["residual"], variant["concat"], seed=77)
    eng = VariantEngine(cfg.pixel_size, cfg.max_size, cfg.octaves, variant["block_depth"], variant["residual"], variant["concat"], 0, gpu,
                        steps=6, predict_x=False)
    eng.set_params(params)
    z = np.load(GOLDEN)
    zero = lambda pred: (0.0, np.zeros_like(pred))
    denoise = lambda x: V.variant_forward_backward(params, x, cfg, variant["block_depth"], variant["residual"], variant["concat"], zero)[1]
    ref = S.log_sample(denoise, z["example_image"], z["example"], z["dictionary"], 6, 2, predict_x=False)
    den = types.SimpleNamespace(ensure_engine=lambda: eng)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=gpu)
    res = g.log_sample(den, t(z["example_image"]), t(z["example"]), t(z["dictionary"]), steps=6, test_step=2, predict_x=False)
    torch.cuda.synchronize()
    for k, v in res.items():
        assert rel_l2(v.cpu().numpy().reshape(np.asarray(ref[k]).shape), ref[k]) <= 2e-4, k
