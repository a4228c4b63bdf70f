"""CPU tier: pins the oracle itself (it is the parity anchor for every GPU test).

The reference holds no golden vectors, so the oracle is pinned by
  (1) definition-level pure-Python loops of SURVEY.md A.2/A.3 at tiny shapes,
  (2) an independent torch.nn.functional + autograd derivation of the whole step,
  (3) analytic identities (convT is the vjp of conv; linearity),
  (4) the committed golden fixture (regression of the restatement itself).
"""
import os

import numpy as np
import pytest
import torch

from oracle import denoiser_oracle as O
from oracle import torch_cross as T

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_step.npz")


def test_vectorised_matches_definition_loops():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 4, 6, 3)); b = rng.standard_normal(5)
    w = rng.standard_normal((4, 4, 3, 5))
    assert np.abs(O.naive_conv4s2_fwd(x, w, b) - O.conv4s2_fwd(x, w, b)).max() < 1e-12
    wt = rng.standard_normal((4, 4, 5, 3))
    assert np.abs(O.naive_convT4s2_fwd(x, wt, b) - O.convT4s2_fwd(x, wt, b)).max() < 1e-12


def test_convT_is_vjp_of_conv_and_interior_tap_count():
    """A.3: Conv2DTranspose == gradient of Conv2D w.r.t. its input, no spatial flip; interior pixels get 2x2 taps."""
    rng = np.random.default_rng(1)
    x = rng.standard_normal((1, 6, 6, 4)); w = rng.standard_normal((4, 4, 4, 7)); dz = rng.standard_normal((1, 3, 3, 7))
    dx, _, _ = O.conv4s2_bwd(x, w, dz)
    assert np.abs(dx - O.convT4s2_fwd(dz, w, np.zeros(4))).max() < 1e-12
    ones = O.convT4s2_fwd(np.ones((1, 3, 3, 1)), np.ones((4, 4, 1, 1)), np.zeros(1))[0, :, :, 0]
    assert ones[2, 2] == 4 and ones[0, 0] == 1 and ones[0, 2] == 2 and ones.shape == (6, 6)


def test_same_padding_rule_even_input():
    """A.2: 'same' with k=4, s=2, even input -> pad 1/1, out = in/2; the first tap row reads x[-1] = 0."""
    x = np.zeros((1, 4, 4, 1)); x[0, 0, 0, 0] = 1.0
    w = np.zeros((4, 4, 1, 1)); w[1, 1, 0, 0] = 1.0          # tap (kh,kw)=(1,1) reads x[2oh, 2ow]
    z = O.conv4s2_fwd(x, w, np.zeros(1))
    assert z.shape == (1, 2, 2, 1) and z[0, 0, 0, 0] == 1.0 and z.sum() == 1.0


@pytest.mark.parametrize("cfgkw", [dict(size=16, pixel_size=8, max_size=16, octaves=2, batch_size=2),
                                   dict(size=16, pixel_size=4, max_size=8, octaves=3, batch_size=1)])
def test_full_step_numpy_vs_torch_autograd(cfgkw):
    cfg = O.OracleConfig(**cfgkw)
    params = O.init_params(cfg, seed=7)
    for k in params:
        if k.endswith(".b"):
            params[k] = np.random.default_rng(3).standard_normal(params[k].shape) * 0.1
    x, t_int, eps = O.synthetic_batch(cfg, seed=2)
    l1, p1, g1, _ = O.trainer_step(params, x, t_int, eps, cfg)
    l2, p2, g2 = T.trainer_step(params, x, t_int, eps, cfg)
    assert abs(l1 - l2) < 1e-12 and np.abs(p1 - p2).max() < 1e-12
    for k in g1:
        assert np.abs(g1[k] - g2[k]).max() <= 1e-12 * max(1.0, np.abs(g2[k]).max()), k


def test_golden_fixture_regression():
    z = np.load(GOLDEN)
    cfg = O.OracleConfig(size=16, pixel_size=8, max_size=16, octaves=2, batch_size=2)
    params = {k[6:]: z[k] for k in z.files if k.startswith("param/")}
    loss, pred, grads, noised = O.trainer_step(params, z["x"], z["t_int"], z["eps"], cfg)
    assert abs(loss - float(z["loss"])) < 1e-13 and np.abs(pred - z["pred"]).max() < 1e-12
    for k in grads:
        assert np.abs(grads[k] - z["grad/" + k]).max() < 1e-12


def test_topology_channel_rule_and_parameter_count():
    """41,691,660 parameters at the reference defaults, 29,107,724 at octaves 5."""
    cfg = O.OracleConfig(size=128)
    assert [cfg.down_filters(i) for i in range(6)] == [128, 256, 512, 512, 512, 512]
    assert [cfg.up_filters(i) for i in range(6)] == [64, 128, 256, 512, 512, 512]
    assert [cfg.up_in_channels(i) for i in range(6)] == [256, 512, 1024, 1024, 1024, 512]
    assert sum(int(np.prod(s)) for s in O.param_shapes(cfg).values()) == 41_691_660
    assert sum(int(np.prod(s)) for s in O.param_shapes(O.OracleConfig(size=32, octaves=5)).values()) == 29_107_724
    with pytest.raises(ValueError):
        O.OracleConfig(size=32, octaves=6).check_legal()       # App. B: size % 2**octaves != 0


def test_alpha_dash_and_warmup():
    assert abs(float(O.alpha_dash(0)) - 0.25) < 1e-15                       # train.py:93
    assert abs(float(O.alpha_dash(200)) - 0.25 * (1 - 200 / 201) ** 2) < 1e-18
    assert abs(O.warmup_lr(0) - 2e-5 / 2001) < 1e-12                        # train.py:61-63
    assert abs(O.warmup_lr(1999) - 2e-5 * 2000 / 2001) < 1e-11
    assert abs(O.warmup_lr(2000) - 2e-5) < 1e-12 and abs(O.warmup_lr(10 ** 6) - 2e-5) < 1e-12


def test_keras_adam_definition_and_epsilon_placement():
    """A.6: eps is added to sqrt(v) (not sqrt(v_hat)); differs from torch.optim.Adam even with eps=1e-7."""
    cfg = O.OracleConfig()
    g = np.full(4, 1e-5, dtype=np.float32); p = np.zeros(4, dtype=np.float32)     # p = 0: the update is not absorbed
    p1, m1, v1 = O.keras_adam_step(p, g, np.zeros(4, np.float32), np.zeros(4, np.float32), 0, cfg)
    lr = 2e-5 / 2001
    m, v = 0.1 * 1e-5, 0.001 * 1e-10
    expect = -lr * np.sqrt(1 - 0.999) / (1 - 0.9) * m / (np.sqrt(v) + 1e-7)
    assert np.allclose(p1, expect, rtol=1e-5, atol=0) and np.allclose(m1, m) and np.allclose(v1, v)
    tp = torch.zeros(4, requires_grad=True); opt = torch.optim.Adam([tp], lr=lr, eps=1e-7)
    tp.grad = torch.full((4,), 1e-5); opt.step()
    upd_keras, upd_torch = -float(p1[0]), -float(tp[0])
    assert abs(upd_keras - upd_torch) / upd_torch > 0.2


def test_loss_scale_state_machine():
    s = O.LossScaleState(growth_interval=3)
    assert [s.update(True), s.update(True)] == [True, True] and s.scale == 2.0 ** 15
    assert s.update(True) and s.scale == 2.0 ** 16 and s.good_steps == 0
    assert not s.update(False) and s.scale == 2.0 ** 15 and s.good_steps == 0


def test_operand_rounding_model_bf16():
    a = np.array([1.0, 1.00390625, 1.005859375, -3.14159, 1e-40], dtype=np.float64)
    r = O.round_bf16(a)
    ref = torch.tensor(a, dtype=torch.float32).to(torch.bfloat16).to(torch.float64).numpy()
    assert np.array_equal(r, ref)
    cfg = O.OracleConfig(size=16, pixel_size=8, max_size=16, octaves=2, batch_size=2)
    params = O.init_params(cfg, 1); x, t, e = O.synthetic_batch(cfg, 1)
    l0 = O.trainer_step(params, x, t, e, cfg)[0]
    l1 = O.trainer_step(params, x, t, e, cfg, operand_round="bf16")[0]
    assert 0 < abs(l0 - l1) / l0 < 2e-2


def test_oracle_trainer_two_steps_decrease_nothing_weird():
    cfg = O.OracleConfig(size=16, pixel_size=8, max_size=16, octaves=2, batch_size=2)
    tr = O.OracleTrainer(cfg, O.init_params(cfg, 3))
    x, t, e = O.synthetic_batch(cfg, 4)
    l0 = tr.train_step(x, t, e)[0]; l1 = tr.train_step(x, t, e)[0]
    assert tr.iterations == 2 and np.isfinite(l0) and np.isfinite(l1) and l1 < l0     # same batch twice: loss must drop


# ---- the sampler restatement (oracle/sampler_oracle.py; train.py:323-496) ------------------------------------------------
def test_noise_edits_against_definition_loops():
    from oracle import sampler_oracle as S
    rng = np.random.default_rng(3)
    H, W, K = 8, 12, 8
    eps, dic = rng.standard_normal((1, H, W, 3)), rng.standard_normal((H, W, K, 3))
    out = S.noise_edits(eps, dic)
    assert out.shape == (4, H, W, 3) and np.array_equal(out[0], eps[0])
    for h in range(H):
        for w in range(W):
            blk = eps[0, h // 4 * 4:h // 4 * 4 + 4, w // 4 * 4:w // 4 * 4 + 4]
            assert np.allclose(out[1, h, w], blk.mean((0, 1)), atol=1e-14)                 # avg_pool 4x4 + nearest upsample
            assert np.array_equal(out[2, h, w], eps[0, (h - 1) % H, (w - 1) % W])           # tf.roll by 1 on both axes
            d = ((eps[0, h, w][None] - dic[h, w]) ** 2).sum(-1)
            assert np.array_equal(out[3, h, w], dic[h, w, int(np.argmin(d))])               # per-pixel codebook


def test_sampler_identities_and_golden():
    """(1) call pattern and recurrences of train.py:323-496 checked on an instrumented stand-in denoiser; (2) the committed
    fixture is reproduced bit for bit."""
    import importlib.util
    from oracle import sampler_oracle as S
    rng = np.random.default_rng(5)
    img, ex = rng.uniform(-1, 1, (1, 8, 8, 3)), rng.standard_normal((1, 2, 8, 8, 3))
    dic = rng.standard_normal((8, 8, 8, 3))
    calls = []

    def den(x):
        calls.append(x.copy())
        return 0.5 * x + 0.1

    res = S.log_sample(den, img, ex, dic, steps=5, test_step=2)
    a = lambda t: float(O.alpha_dash(t, 5))
    assert len(calls) == 1 + 5 + 5 and [c.shape[0] for c in calls] == [1] * 6 + [6] * 5   # train.py:332, 377, 446
    assert np.allclose(calls[0], img * a(2) ** 0.5 + ex[0, :1] * (1 - a(2)) ** 0.5, atol=1e-15)
    assert np.allclose(calls[1], (a(1) ** 0.5 + (1 - a(1)) ** 0.5) * img, atol=1e-15)      # eps_theta starts as the image (train.py:367)
    # the predict_x update keeps sqrt(a) x_theta + sqrt(1-a) eps_theta == fake: the next input follows from the previous pair
    x1 = den(calls[1]); calls.pop()
    e1 = (calls[1] - a(1) ** 0.5 * x1) / (1 - a(1)) ** 0.5
    assert np.allclose(calls[2], a(2) ** 0.5 * x1 + (1 - a(2)) ** 0.5 * e1, atol=1e-14)
    assert np.allclose(calls[6][:2], (a(5) ** 0.5 + (1 - a(5)) ** 0.5) * ex[0], atol=1e-14)        # the two random noises lead the batch of six
    assert res["fake"].shape == (6, 8, 8, 3) and set(res) >= {"denoised", "example_loss", "step_1", "step_0.25", "step_0.5", "step_0.75"}
    spec = importlib.util.spec_from_file_location("mgs", os.path.join(os.path.dirname(GOLDEN), "make_golden_sampler.py"))
    mgs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mgs)
    z = np.load(os.path.join(os.path.dirname(GOLDEN), "tiny_sampler.npz"))
    cfg, params, image, example, dictionary = mgs.inputs()
    assert all(np.array_equal(params[k], z["param/" + k]) for k in params) and np.array_equal(image, z["example_image"])
    out = S.log_sample(S.unet_denoiser(params, cfg), image, example, dictionary, mgs.STEPS, mgs.TEST_STEP)
    for k, v in out.items():
        assert np.abs(np.asarray(v) - z["out/" + k]).max() < 1e-12, k
    # the denoiser evaluations agree with the independent torch formulation
    import torch as _t
    pred_o = O.unet_forward(params, image, cfg)[0]
    pred_t = T.unet_forward(params, image, cfg) if hasattr(T, "unet_forward") else None
    if pred_t is not None:
        assert np.abs(pred_o - np.asarray(pred_t)).max() < 1e-10


def test_variants_oracle_against_loops_and_torch_autograd():
    """oracle/variants_oracle.py (block_depth, residual, concat switches of train.py:20-27 and the objectives of train.py:238-252):
    the stride-1 'same' convolution against definition-level loops, and loss / every gradient of each variant against an
    independent torch.nn.functional + autograd evaluation of the same structure."""
    import torch.nn.functional as F
    from oracle import variants_oracle as V
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 5, 4, 3)); w = rng.standard_normal((3, 3, 3, 4)) * 0.3; b = rng.standard_normal(4)
    assert np.abs(V.conv_s1_fwd(x, w, b) - V.naive_conv_s1(x, w, b)).max() < 1e-12
    w1 = rng.standard_normal((1, 1, 3, 5))
    assert np.abs(V.conv_s1_fwd(x, w1) - x @ w1[0, 0]).max() < 1e-12           # a 1 x 1 convolution is a Dense on a rank-4 tensor
    cfg = O.OracleConfig(size=8, pixel_size=4, max_size=8, octaves=2, batch_size=2)
    xs, ts, es = O.synthetic_batch(cfg, seed=1)
    cases = [dict(block_depth=1, residual=False, concat=True, objective=None),
             dict(block_depth=0, residual=True, concat=True, objective=dict(predict_x=False)),
             dict(block_depth=2, residual=False, concat=False, objective=dict(ordinary_differential_equation=True)),
             dict(block_depth=1, residual=True, concat=False, objective=dict(predict_x=False, predict_scaled_epsilon=True, prediction_weighting=True))]
    for case in cases:
        bd, res, cat = case["block_depth"], case["residual"], case["concat"]
        params = V.init_variant_params(cfg, bd, res, cat, seed=3)
        loss, pred, grads = V.variant_trainer_step(params, xs, ts, es, cfg, bd, res, cat, case["objective"])
        # torch: same structure, NCHW, autograd
        P = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in params.items()}
        conv = lambda h, n: F.relu(F.conv2d(h, P[n + ".w"].permute(3, 2, 0, 1), P[n + ".b"], stride=2, padding=1))
        convT = lambda h, n: F.relu(F.conv_transpose2d(h, P[n + ".w"].permute(3, 2, 0, 1), P[n + ".b"], stride=2, padding=1))
        c3 = lambda h, n: F.relu(F.conv2d(h, P[n + ".w"].permute(3, 2, 0, 1), P[n + ".b"], stride=1, padding=1))

        def block(h, name):
            for d in range(bd):
                h = c3(h, f"{name}.{d}")
            return h

        def level(i, v):
            h = block(conv(v, f"D{i}"), f"blkA{i}")
            h = level(i + 1, h) if i + 1 < cfg.octaves else block(h, "blkMid")
            h = convT(block(h, f"blkB{i}"), f"U{i}")
            if res:
                return v + torch.einsum("bchw,cd->bdhw", h, P[f"res{i}.dense.w"])
            return torch.cat([h, v], 1) if cat else h

        noised = torch.tensor(O.noise_image(xs, ts, es, cfg.steps)).permute(0, 3, 1, 2)
        h = block(level(0, block(noised, "blkTopA")), "blkTopB")
        tp = torch.einsum("bchw,cd->bhwd", h, P["dense.w"]) + P["dense.b"]
        target, wgt = O.objective_terms(xs, ts, es, cfg.steps, **(case["objective"] or {}))
        tl = torch.mean((tp * torch.tensor(wgt) - torch.tensor(target)) ** 2)
        tl.backward()
        assert abs(loss - float(tl)) < 1e-12 and np.abs(pred - tp.detach().numpy()).max() < 1e-11
        for k in grads:
            assert np.abs(grads[k] - P[k].grad.numpy()).max() <= 1e-11 * max(1.0, np.abs(grads[k]).max()), (case, k)
        assert set(grads) == set(params)


def test_objective_terms_and_fp16_noise_model():
    x, t, e = np.ones((2, 1, 1, 3)), np.array([1, 200]), np.full((2, 1, 1, 3), 2.0)
    a = O.alpha_dash(np.array([1.0, 200.0]))
    tg, w = O.objective_terms(x, t, e)
    assert tg is x and np.all(w == 1)
    tg, w = O.objective_terms(x, t, e, predict_x=False, predict_scaled_epsilon=True, prediction_weighting=True)
    assert np.allclose(tg[:, 0, 0, 0], 2.0 * (1 - a)) and np.allclose(w[:, 0, 0, 0], np.sqrt(1 - a))
    tg, w = O.objective_terms(x, t, e, ordinary_differential_equation=True)
    a0 = O.alpha_dash(np.array([0.0, 199.0]))
    assert np.allclose(tg[:, 0, 0, 0], np.sqrt(a0) + 2 * np.sqrt(1 - a0)) and np.all(w == 1)
    n16 = O.noise_image_f16(x, t, e)
    assert n16.dtype == np.float16 and np.abs(n16.astype(np.float64) - O.noise_image(x, t, e)).max() < 4e-3


def test_keras_strict_mode_adds_the_two_mixed_float16_rounding_points():
    """r06: trainer_step(keras_strict=True) = the rounding model + the two points of Keras' mixed_float16 policy that the HIP path
    leaves out (fp16 conv output before an fp16 bias add; fp16 variable gradients, inf on overflow).  Without a rounding type it is
    the plain step; with one every gradient it returns is representable in that type, it differs from the default model by the
    fp16 noise level (not more), and a gradient beyond 65504 overflows to inf as the reference's would."""
    cfg = O.OracleConfig(size=16, batch_size=2, octaves=2, pixel_size=8, max_size=16)
    rng = np.random.default_rng(5)
    params = {k: v.astype(np.float64) for k, v in O.init_params(cfg, seed=1).items()}
    for k in params:
        if k.endswith(".b"):
            params[k] = params[k] + 0.05 * rng.standard_normal(params[k].shape)
    x, t, e = O.synthetic_batch(cfg, seed=0)
    x, e = x.astype(np.float64), e.astype(np.float64)
    plain, strict_plain = O.trainer_step(params, x, t, e, cfg), O.trainer_step(params, x, t, e, cfg, keras_strict=True)
    assert plain[0] == strict_plain[0] and all(np.array_equal(plain[2][k], strict_plain[2][k]) for k in plain[2])
    a = O.trainer_step(params, x, t, e, cfg, operand_round="f16", loss_scale=2.0 ** 10)
    b = O.trainer_step(params, x, t, e, cfg, operand_round="f16", loss_scale=2.0 ** 10, keras_strict=True)
    assert abs(a[0] - b[0]) <= 1e-3 * a[0] and 0 < np.abs(a[1] - b[1]).max() <= 5e-3 * np.abs(a[1]).max()
    for k, g in b[2].items():
        assert np.array_equal(g, g.astype(np.float16).astype(np.float64)), k           # an fp16 tensor
        d = np.linalg.norm(a[2][k] - g) / np.linalg.norm(a[2][k])
        assert 0 < d <= 5e-2, (k, d)
    with np.errstate(over="ignore"):
        c = O.trainer_step(params, x, t, e, cfg, operand_round="f16", loss_scale=2.0 ** 40, keras_strict=True)
    assert any(not np.isfinite(g).all() for g in c[2].values())
