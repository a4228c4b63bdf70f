"""GPU parity of every C-ABI kernel against the CPU oracle (oracle/denoiser_oracle.py), same seeded inputs.

Tolerances (stated per SURVEY.md §4):
  * GCT2_F32  : rel-L2 <= 2e-6  (fp32 fma accumulation vs the fp64 oracle)
  * GCT2_BF16 / GCT2_F16 : inputs are pre-rounded to the 16-bit type, so only accumulation order (fp32) and the
    final output rounding differ from the oracle: rel-L2 <= 4e-3 (bf16, 2^-9 output rounding) / 6e-4 (f16).
PARITY UNPINNED w.r.t. TensorFlow: the reference holds no fixtures (see oracle/denoiser_oracle.py header).
"""
import numpy as np
import pytest
import torch

from oracle import denoiser_oracle as O

pytestmark = pytest.mark.gpu

F32, BF16, F16 = 0, 1, 2
TDT = {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}
TOL_OUT = {F32: 2e-6, BF16: 4e-3, F16: 6e-4}      # outputs stored in the compute dtype
TOL_F32OUT = {F32: 2e-6, BF16: 2e-5, F16: 2e-5}   # fp32 outputs (weight gradients) of pre-rounded operands


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def rnd(a, dt):
    """value-round a float64 array to the compute dtype."""
    if dt == F32:
        return a.astype(np.float32).astype(np.float64)
    return torch.tensor(a, dtype=torch.float64).to(TDT[dt]).to(torch.float64).numpy()


def dev(a, dt, device):
    return torch.tensor(np.asarray(a), dtype=torch.float64).to(TDT[dt]).to(device).contiguous()


def lib():
    import gan_class_transfer2_amd as g
    return g._lib


def stream():
    return torch.cuda.current_stream().cuda_stream


# every test gets its own call context (include/gct2.h gct2_ctx): scratch and tile knobs are per context, nothing is global
_CTX = [None]


@pytest.fixture(autouse=True)
def _fresh_ctx(gpu):
    import gan_class_transfer2_amd as g
    _CTX[0] = g._lib.Context()
    yield
    _CTX[0] = None


def ctx():
    return _CTX[0].handle


def ctx_obj():
    return _CTX[0]


def set_ws(t):
    _CTX[0].set_workspace(t)


def set_tuning(v):
    _CTX[0].set_tuning(v)


# (B, H, W, Cin, Cout): MFMA-eligible shapes (multiples of 8) incl. ragged tiles, and direct-path shapes
CONV_SHAPES = [
    (2, 8, 8, 64, 128),      # one full tile
    (1, 4, 12, 72, 136),     # K and N ragged w.r.t. 64/128, M ragged
    (3, 2, 2, 256, 64),      # tiny spatial grid (bottleneck regime), deep K
    (2, 16, 16, 3, 8),       # image layer: Cin = 3 -> rgb_mfma kernels (16-bit) / direct (fp32)
    (3, 32, 32, 3, 128),     # image layer at the real channel count, several r-steps per workgroup
    (1, 8, 8, 4, 136),       # Cin = 4, ragged N
    (1, 6, 10, 5, 7),        # nothing aligned -> direct kernels
]


@pytest.mark.parametrize("dt", [F32, BF16, F16])
@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv4s2_fwd(gpu, dt, shape):
    B, H, W, Cin, Cout = shape
    rng = np.random.default_rng(1)
    x = rnd(rng.standard_normal((B, H, W, Cin)), dt)
    w = rnd(rng.standard_normal((4, 4, Cin, Cout)) * 0.1, dt)
    b = rng.standard_normal(Cout).astype(np.float32).astype(np.float64)
    ref = np.maximum(O.conv4s2_fwd(x, w, b), 0)
    # input and output both live inside wider (concat-style) buffers: exercises ld != C and pointer offsets
    ldx, ldy, offx, offy = Cin + 8, Cout + 16, 8, 8
    xb = torch.zeros(B, H, W, ldx, dtype=TDT[dt], device=gpu)
    xb[..., offx:offx + Cin] = dev(x, dt, gpu)
    yb = torch.full((B, H // 2, W // 2, ldy), 7.0, dtype=TDT[dt], device=gpu)
    wd, bd = dev(w, dt, gpu), torch.tensor(b, dtype=torch.float32, device=gpu)
    es = xb.element_size()
    lib().call("gct2_conv4s2_fwd", ctx(), dt, xb.data_ptr() + offx * es, ldx, wd.data_ptr(), bd.data_ptr(),
               yb.data_ptr() + offy * es, ldy, B, H, W, Cin, Cout, 1, stream())
    torch.cuda.synchronize()
    out = yb[..., offy:offy + Cout].double().cpu().numpy()
    assert rel_l2(out, ref) <= TOL_OUT[dt]
    # nothing outside the output slice was touched
    assert float((yb[..., :offy].float() - 7).abs().max()) == 0 and float((yb[..., offy + Cout:].float() - 7).abs().max()) == 0


@pytest.mark.parametrize("dt", [F32, BF16, F16])
@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_convT4s2_fwd(gpu, dt, shape):
    B, H, W, Cin, Cout = shape
    rng = np.random.default_rng(2)
    x = rnd(rng.standard_normal((B, H, W, Cin)), dt)
    w = rnd(rng.standard_normal((4, 4, Cout, Cin)) * 0.1, dt)
    b = rng.standard_normal(Cout).astype(np.float32).astype(np.float64)
    ref = np.maximum(O.convT4s2_fwd(x, w, b), 0)
    xd, wd, bd = dev(x, dt, gpu), dev(w, dt, gpu), torch.tensor(b, dtype=torch.float32, device=gpu)
    ldy = Cout + 8
    yb = torch.zeros(B, 2 * H, 2 * W, ldy, dtype=TDT[dt], device=gpu)
    lib().call("gct2_convT4s2_fwd", ctx(), dt, xd.data_ptr(), Cin, wd.data_ptr(), bd.data_ptr(), yb.data_ptr(), ldy,
               B, H, W, Cin, Cout, 1, stream())
    torch.cuda.synchronize()
    assert rel_l2(yb[..., :Cout].double().cpu().numpy(), ref) <= TOL_OUT[dt]
    assert float(yb[..., Cout:].float().abs().max()) == 0


@pytest.mark.parametrize("dt", [F32, BF16, F16])
@pytest.mark.parametrize("shape", CONV_SHAPES)
@pytest.mark.parametrize("accumulate", [0, 1])
def test_conv4s2_dgrad(gpu, dt, shape, accumulate):
    B, H, W, Cin, Cout = shape
    rng = np.random.default_rng(3)
    x = rnd(np.maximum(rng.standard_normal((B, H, W, Cin)), 0), dt)      # ReLU output that fed the conv
    w = rnd(rng.standard_normal((4, 4, Cin, Cout)) * 0.1, dt)
    dz = rnd(rng.standard_normal((B, H // 2, W // 2, Cout)), dt)
    prev = rnd(rng.standard_normal((B, H, W, Cin)), dt)
    dx_ref, _, _ = O.conv4s2_bwd(x, w, dz)
    ref = dx_ref * (x > 0) + (prev if accumulate else 0)
    dzd, wd, actd = dev(dz, dt, gpu), dev(w, dt, gpu), dev(x, dt, gpu)
    dxd = dev(prev, dt, gpu)
    lib().call("gct2_conv4s2_dgrad", ctx(), dt, dzd.data_ptr(), Cout, wd.data_ptr(), actd.data_ptr(), Cin, dxd.data_ptr(), Cin,
               B, H, W, Cin, Cout, accumulate, None, 0, None, 0, stream())
    torch.cuda.synchronize()
    assert rel_l2(dxd.double().cpu().numpy(), ref) <= TOL_OUT[dt]


@pytest.mark.parametrize("dt", [F32, BF16, F16])
@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_convT4s2_dgrad(gpu, dt, shape):
    B, H, W, Cin, Cout = shape
    rng = np.random.default_rng(4)
    x = rnd(np.maximum(rng.standard_normal((B, H, W, Cin)), 0), dt)
    w = rnd(rng.standard_normal((4, 4, Cout, Cin)) * 0.1, dt)
    dz = rnd(rng.standard_normal((B, 2 * H, 2 * W, Cout)), dt)
    dx_ref, _, _ = O.convT4s2_bwd(x, w, dz)
    ref = dx_ref * (x > 0)
    dzd, wd, actd = dev(dz, dt, gpu), dev(w, dt, gpu), dev(x, dt, gpu)
    dxd = torch.zeros(B, H, W, Cin, dtype=TDT[dt], device=gpu)
    lib().call("gct2_convT4s2_dgrad", ctx(), dt, dzd.data_ptr(), Cout, wd.data_ptr(), actd.data_ptr(), Cin, dxd.data_ptr(), Cin,
               B, H, W, Cin, Cout, 0, None, 0, None, 0, stream())
    torch.cuda.synchronize()
    assert rel_l2(dxd.double().cpu().numpy(), ref) <= TOL_OUT[dt]
    # no mask: plain input gradient
    lib().call("gct2_convT4s2_dgrad", ctx(), dt, dzd.data_ptr(), Cout, wd.data_ptr(), None, 0, dxd.data_ptr(), Cin,
               B, H, W, Cin, Cout, 0, None, 0, None, 0, stream())
    torch.cuda.synchronize()
    assert rel_l2(dxd.double().cpu().numpy(), dx_ref) <= TOL_OUT[dt]


WGRAD_SHAPES = CONV_SHAPES + [(4, 32, 32, 64, 128)]   # long reduction: exercises the r-split + atomics


@pytest.mark.parametrize("dt", [F32, BF16, F16])
@pytest.mark.parametrize("shape", WGRAD_SHAPES)
def test_conv4s2_wgrad(gpu, dt, shape):
    B, H, W, Cin, Cout = shape
    rng = np.random.default_rng(5)
    x = rnd(rng.standard_normal((B, H, W, Cin)), dt)
    dz = rnd(rng.standard_normal((B, H // 2, W // 2, Cout)), dt)
    _, dw_ref, db_ref = O.conv4s2_bwd(x, np.zeros((4, 4, Cin, Cout)), dz)
    xd, dzd = dev(x, dt, gpu), dev(dz, dt, gpu)
    dw = torch.zeros(4, 4, Cin, Cout, dtype=torch.float32, device=gpu)
    db = torch.zeros(Cout, dtype=torch.float32, device=gpu)
    lib().call("gct2_conv4s2_wgrad", ctx(), dt, xd.data_ptr(), Cin, dzd.data_ptr(), Cout, dw.data_ptr(), db.data_ptr(),
               B, H, W, Cin, Cout, 1, None, stream())
    torch.cuda.synchronize()
    assert rel_l2(dw.cpu().numpy(), dw_ref) <= TOL_F32OUT[dt]
    assert rel_l2(db.cpu().numpy(), db_ref) <= TOL_F32OUT[dt]
    # the entry point ACCUMULATES: a second call doubles the result
    lib().call("gct2_conv4s2_wgrad", ctx(), dt, xd.data_ptr(), Cin, dzd.data_ptr(), Cout, dw.data_ptr(), db.data_ptr(),
               B, H, W, Cin, Cout, 1, None, stream())
    torch.cuda.synchronize()
    assert rel_l2(dw.cpu().numpy(), 2 * dw_ref) <= TOL_F32OUT[dt]


@pytest.mark.parametrize("shape", [(4, 32, 32, 64, 128), (2, 16, 16, 72, 136), (4, 8, 8, 256, 512)])
def test_wgrad_reduction_modes_agree(gpu, shape):
    """split reduction through workspace slabs (deterministic) == fp32 atomics == oracle; one-owner tiles (rsplit 1)."""
    B, H, W, Cin, Cout = shape
    dt = BF16
    rng = np.random.default_rng(14)
    x = rnd(rng.standard_normal((B, H, W, Cin)), dt)
    dz = rnd(rng.standard_normal((B, H // 2, W // 2, Cout)), dt)
    _, dw_ref, _ = O.conv4s2_bwd(x, np.zeros((4, 4, Cin, Cout)), dz)
    xd, dzd = dev(x, dt, gpu), dev(dz, dt, gpu)
    ws = torch.full((16 << 18,), float("nan"), dtype=torch.float32, device=gpu)
    res = []
    for use_ws in (False, True, True):
        set_ws(ws if use_ws else None)
        dw = torch.ones(4, 4, Cin, Cout, dtype=torch.float32, device=gpu)       # running buffer: the call ACCUMULATES
        lib().call("gct2_conv4s2_wgrad", ctx(), dt, xd.data_ptr(), Cin, dzd.data_ptr(), Cout, dw.data_ptr(), None, B, H, W, Cin, Cout, 1, None, stream())
        torch.cuda.synchronize()
        res.append(dw.cpu().numpy())
    set_ws(None)
    assert rel_l2(res[0] - 1, dw_ref) <= TOL_F32OUT[dt] and rel_l2(res[1] - 1, dw_ref) <= TOL_F32OUT[dt]
    assert np.array_equal(res[1], res[2])          # slab path is bitwise reproducible


@pytest.mark.parametrize("use_ws", [False, True])
@pytest.mark.parametrize("variant", [2, 3, 7])
@pytest.mark.parametrize("shape", [(4, 32, 32, 64, 128), (1, 12, 20, 72, 136), (2, 8, 8, 256, 512)])
def test_wgrad_tile_variants(gpu, variant, shape, use_ws):
    """weight-gradient tiles: 2 = 256x256 (8 waves of 128x64), five-stage ring with counted vmcnt across raw barriers and the lean
    stage (incremental gather addresses, fragment addresses computed once); 3 = 128x128 with two buffers (lean stage since r04);
    7 = 128x128 keeping fp32 atomics although a workspace is registered; without a workspace (atomics) and with one (ordered slabs)."""
    B, H, W, Cin, Cout = shape
    dt = BF16
    L = lib()
    ws = torch.empty(16 << 18, dtype=torch.float32, device=gpu)
    set_ws(ws if use_ws else None)
    set_tuning(variant << 16)
    try:
        rng = np.random.default_rng(16)
        x = rnd(rng.standard_normal((B, H, W, Cin)), dt)
        dz = rnd(rng.standard_normal((B, H // 2, W // 2, Cout)), dt)
        dzt = rnd(rng.standard_normal((B, 2 * H, 2 * W, Cout)), dt)
        xd, dzd, dztd = dev(x, dt, gpu), dev(dz, dt, gpu), dev(dzt, dt, gpu)
        dw = torch.zeros(4, 4, Cin, Cout, dtype=torch.float32, device=gpu)
        dwt = torch.zeros(4, 4, Cout, Cin, dtype=torch.float32, device=gpu)
        L.call("gct2_conv4s2_wgrad", ctx(), dt, xd.data_ptr(), Cin, dzd.data_ptr(), Cout, dw.data_ptr(), None, B, H, W, Cin, Cout, 1, None, stream())
        L.call("gct2_convT4s2_wgrad", ctx(), dt, xd.data_ptr(), Cin, dztd.data_ptr(), Cout, dwt.data_ptr(), None, B, H, W, Cin, Cout, 1, None, stream())
        torch.cuda.synchronize()
        assert rel_l2(dw.cpu().numpy(), O.conv4s2_bwd(x, np.zeros((4, 4, Cin, Cout)), dz)[1]) <= TOL_F32OUT[dt]
        assert rel_l2(dwt.cpu().numpy(), O.convT4s2_bwd(x, np.zeros((4, 4, Cout, Cin)), dzt)[1]) <= TOL_F32OUT[dt]
    finally:
        set_tuning(0)
        set_ws(None)


@pytest.mark.parametrize("shape", [(2, 64, 64, 64, 256), (3, 32, 32, 72, 136), (5, 16, 16, 128, 64), (2, 128, 64, 64, 64), (3, 40, 24, 64, 128),
                                   (21, 16, 16, 16, 64)])      # 21 steps of 64 rows over 5 pixel splits: the last split has ONE step = two stages
def test_wgrad_turns_and_scalar_addresses_equal_r03_order_bit_for_bit(gpu, shape):
    """The stage orders of the 256x256 weight-gradient tile: r05 (default: the next stage's fragments read under the multiplies, one
    barrier earlier in the ring), r04 (tuning 16-23 = 5: the two waves of a SIMD take turns between DMA issue and multiplies; stage
    position, offsets and border bits in SGPRs where a 32-row stage is aligned with the image rows) and r03 (4): same sources, same
    zero fill, same multiplies in the same order, hence the same bits - grids of 64 / 32 / 16 / 8 columns (stage = part of a row,
    one row, 2 and 4 rows), ragged channel counts, several images per split, and one shape that is not aligned (all three tunings
    then run the r03 order)."""
    B, H, W, Cin, Cout = shape
    dt, L = BF16, lib()
    rng = np.random.default_rng(71)
    x = rnd(rng.standard_normal((B, H, W, Cin)), dt)
    dz = rnd(rng.standard_normal((B, H // 2, W // 2, Cout)), dt)
    dzt = rnd(rng.standard_normal((B, 2 * H, 2 * W, Cout)), dt)
    xd, dzd, dztd = dev(x, dt, gpu), dev(dz, dt, gpu), dev(dzt, dt, gpu)
    ws = torch.empty(64 << 18, dtype=torch.float32, device=gpu)
    set_ws(ws)
    res = {}
    try:
        for variant in (2, 5, 4):
            set_tuning(variant << 16)
            dw = torch.full((4, 4, Cin, Cout), float("nan"), dtype=torch.float32, device=gpu)
            dwt = torch.full((4, 4, Cout, Cin), float("nan"), dtype=torch.float32, device=gpu)
            L.call("gct2_conv4s2_wgrad", ctx(), dt, xd.data_ptr(), Cin, dzd.data_ptr(), Cout, dw.data_ptr(), None, B, H, W, Cin, Cout, 0, None, stream())
            L.call("gct2_convT4s2_wgrad", ctx(), dt, xd.data_ptr(), Cin, dztd.data_ptr(), Cout, dwt.data_ptr(), None, B, H, W, Cin, Cout, 0, None, stream())
            torch.cuda.synchronize()
            res[variant] = (dw, dwt)
        assert torch.equal(res[2][0], res[4][0]) and torch.equal(res[2][1], res[4][1])
        assert torch.equal(res[5][0], res[4][0]) and torch.equal(res[5][1], res[4][1])
        assert rel_l2(res[2][0].cpu().numpy(), O.conv4s2_bwd(x, np.zeros((4, 4, Cin, Cout)), dz)[1]) <= TOL_F32OUT[dt]
        assert rel_l2(res[2][1].cpu().numpy(), O.convT4s2_bwd(x, np.zeros((4, 4, Cout, Cin)), dzt)[1]) <= TOL_F32OUT[dt]
    finally:
        set_tuning(0)
        set_ws(None)


@pytest.mark.parametrize("shape", [(8, 16, 16, 128, 64), (16, 8, 8, 64, 128), (64, 4, 4, 256, 72), (40, 2, 2, 128, 128), (3, 8, 8, 64, 64),
                                   (5, 16, 16, 72, 136), (9, 4, 8, 64, 64)])
def test_wgrad_image_aligned_addresses_equal_the_general_form_bit_for_bit(gpu, shape):
    """r05: the 128x128 weight-gradient tile on grids where a 64-row step covers whole images (64 % (H/2 * W/2) == 0: the bottleneck
    levels) takes its gather addresses as "lane constant + step x stride" and its fragment addresses as lane constants - against the
    general address code (tuning 16-23 = 6): same sources, same zero fill, same multiplies, hence the same bits.  8x8 ... 1x1 small
    grids, a last step with missing rows, ragged channel counts, and a grid (2x4) whose images do not divide a step into a power of
    two of rows per image row."""
    B, H, W, Cin, Cout = shape
    dt, L = BF16, lib()
    rng = np.random.default_rng(81)
    x = rnd(rng.standard_normal((B, H, W, Cin)), dt)
    dz = rnd(rng.standard_normal((B, H // 2, W // 2, Cout)), dt)
    dzt = rnd(rng.standard_normal((B, 2 * H, 2 * W, Cout)), dt)
    xd, dzd, dztd = dev(x, dt, gpu), dev(dz, dt, gpu), dev(dzt, dt, gpu)
    ws = torch.empty(64 << 18, dtype=torch.float32, device=gpu)
    set_ws(ws)
    res = {}
    try:
        for variant in (3, 6):
            set_tuning(variant << 16)
            dw = torch.full((4, 4, Cin, Cout), float("nan"), dtype=torch.float32, device=gpu)
            dwt = torch.full((4, 4, Cout, Cin), float("nan"), dtype=torch.float32, device=gpu)
            L.call("gct2_conv4s2_wgrad", ctx(), dt, xd.data_ptr(), Cin, dzd.data_ptr(), Cout, dw.data_ptr(), None, B, H, W, Cin, Cout, 0, None, stream())
            L.call("gct2_convT4s2_wgrad", ctx(), dt, xd.data_ptr(), Cin, dztd.data_ptr(), Cout, dwt.data_ptr(), None, B, H, W, Cin, Cout, 0, None, stream())
            torch.cuda.synchronize()
            res[variant] = (dw, dwt)
        assert torch.equal(res[3][0], res[6][0]) and torch.equal(res[3][1], res[6][1])
        assert rel_l2(res[3][0].cpu().numpy(), O.conv4s2_bwd(x, np.zeros((4, 4, Cin, Cout)), dz)[1]) <= TOL_F32OUT[dt]
        assert rel_l2(res[3][1].cpu().numpy(), O.convT4s2_bwd(x, np.zeros((4, 4, Cout, Cin)), dzt)[1]) <= TOL_F32OUT[dt]
    finally:
        set_tuning(0)
        set_ws(None)


@pytest.mark.parametrize("variant", [2, 3])
@pytest.mark.parametrize("shape", [(3, 40, 24, 64, 128), (5, 8, 8, 128, 64), (2, 64, 64, 64, 256), (7, 4, 12, 256, 72)])
def test_wgrad_incremental_gather_addresses(gpu, shape, variant):
    """both weight-gradient kernels advance their gather addresses incrementally (32 / 64 rows per stage, carries into the next image
    row and the next image) instead of decoding every row: checked against the oracle on grids narrower and wider than a stage,
    ragged row counts and several images per stage, in overwrite mode, twice (bitwise reproducible through the ordered slabs)."""
    B, H, W, Cin, Cout = shape
    dt, L = BF16, lib()
    rng = np.random.default_rng(61)
    x = rnd(rng.standard_normal((B, H, W, Cin)), dt)
    dz = rnd(rng.standard_normal((B, H // 2, W // 2, Cout)), dt)
    dzt = rnd(rng.standard_normal((B, 2 * H, 2 * W, Cout)), dt)
    xd, dzd, dztd = dev(x, dt, gpu), dev(dz, dt, gpu), dev(dzt, dt, gpu)
    ws = torch.empty(32 << 18, dtype=torch.float32, device=gpu)
    set_ws(ws)
    res = []
    try:
        set_tuning(variant << 16)
        for rep in range(2):
            dw = torch.full((4, 4, Cin, Cout), float("nan"), dtype=torch.float32, device=gpu)
            dwt = torch.full((4, 4, Cout, Cin), float("nan"), dtype=torch.float32, device=gpu)
            L.call("gct2_conv4s2_wgrad", ctx(), dt, xd.data_ptr(), Cin, dzd.data_ptr(), Cout, dw.data_ptr(), None, B, H, W, Cin, Cout, 0, None, stream())
            L.call("gct2_convT4s2_wgrad", ctx(), dt, xd.data_ptr(), Cin, dztd.data_ptr(), Cout, dwt.data_ptr(), None, B, H, W, Cin, Cout, 0, None, stream())
            torch.cuda.synchronize()
            res.append((dw, dwt))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
        assert rel_l2(res[0][0].cpu().numpy(), O.conv4s2_bwd(x, np.zeros((4, 4, Cin, Cout)), dz)[1]) <= TOL_F32OUT[dt]
        assert rel_l2(res[0][1].cpu().numpy(), O.convT4s2_bwd(x, np.zeros((4, 4, Cout, Cin)), dzt)[1]) <= TOL_F32OUT[dt]
    finally:
        set_tuning(0)
        set_ws(None)


@pytest.mark.parametrize("dt", [F32, BF16, F16])
@pytest.mark.parametrize("shape", WGRAD_SHAPES)
def test_convT4s2_wgrad(gpu, dt, shape):
    B, H, W, Cin, Cout = shape
    H, W = H // 2, W // 2
    rng = np.random.default_rng(6)
    x = rnd(rng.standard_normal((B, H, W, Cin)), dt)
    dz = rnd(rng.standard_normal((B, 2 * H, 2 * W, Cout)), dt)
    _, dw_ref, db_ref = O.convT4s2_bwd(x, np.zeros((4, 4, Cout, Cin)), dz)
    xd, dzd = dev(x, dt, gpu), dev(dz, dt, gpu)
    dw = torch.zeros(4, 4, Cout, Cin, dtype=torch.float32, device=gpu)
    db = torch.zeros(Cout, dtype=torch.float32, device=gpu)
    lib().call("gct2_convT4s2_wgrad", ctx(), dt, xd.data_ptr(), Cin, dzd.data_ptr(), Cout, dw.data_ptr(), db.data_ptr(),
               B, H, W, Cin, Cout, 1, None, stream())
    torch.cuda.synchronize()
    assert rel_l2(dw.cpu().numpy(), dw_ref) <= TOL_F32OUT[dt]
    assert rel_l2(db.cpu().numpy(), db_ref) <= TOL_F32OUT[dt]


@pytest.mark.parametrize("order", [0, 1, 2])
@pytest.mark.parametrize("dt", [BF16, F16])
def test_splitk_bottleneck_layers(gpu, dt, order):
    """small-M / deep-K layers (U-Net bottleneck) with a registered workspace: split-K slabs + finalize kernel; with the tile ->
    XCD order automatic (weight slices per XCD here: the weights are the bigger operand), forced to pixel bands, forced to slices."""
    L = lib()
    ws = torch.empty(32 << 18, dtype=torch.float32, device=gpu)        # 32 MiB, deliberately NOT zeroed
    ws.fill_(float("nan"))
    set_ws(ws)
    set_tuning(order << 26)
    try:
        B, H, W, Cin, Cout = 4, 4, 4, 512, 256
        rng = np.random.default_rng(11)
        x = rnd(np.maximum(rng.standard_normal((B, H, W, Cin)), 0), dt)
        # Conv2D forward + its input gradient
        w = rnd(rng.standard_normal((4, 4, Cin, Cout)) * 0.05, dt)
        b = rng.standard_normal(Cout).astype(np.float32).astype(np.float64)
        xd, wd, bd = dev(x, dt, gpu), dev(w, dt, gpu), torch.tensor(b, dtype=torch.float32, device=gpu)
        y = torch.zeros(B, H // 2, W // 2, Cout, dtype=TDT[dt], device=gpu)
        L.call("gct2_conv4s2_fwd", ctx(), dt, xd.data_ptr(), Cin, wd.data_ptr(), bd.data_ptr(), y.data_ptr(), Cout, B, H, W, Cin, Cout, 1, stream())
        torch.cuda.synchronize()
        assert rel_l2(y.double().cpu().numpy(), np.maximum(O.conv4s2_fwd(x, w, b), 0)) <= TOL_OUT[dt]
        dz = rnd(rng.standard_normal((B, H // 2, W // 2, Cout)), dt)
        prev = rnd(rng.standard_normal((B, H, W, Cin)), dt)
        dxd = dev(prev, dt, gpu)
        dz_dev = dev(dz, dt, gpu)     # named: a temporary would be freed (and reused) before the kernel runs
        L.call("gct2_conv4s2_dgrad", ctx(), dt, dz_dev.data_ptr(), Cout, wd.data_ptr(), xd.data_ptr(), Cin, dxd.data_ptr(), Cin,
               B, H, W, Cin, Cout, 1, None, 0, None, 0, stream())
        torch.cuda.synchronize()
        assert rel_l2(dxd.double().cpu().numpy(), O.conv4s2_bwd(x, w, dz)[0] * (x > 0) + prev) <= TOL_OUT[dt]
        # Conv2DTranspose forward + its input gradient
        wt = rnd(rng.standard_normal((4, 4, Cout, Cin)) * 0.05, dt)
        wtd = dev(wt, dt, gpu)
        yt = torch.zeros(B, 2 * H, 2 * W, Cout, dtype=TDT[dt], device=gpu)
        L.call("gct2_convT4s2_fwd", ctx(), dt, xd.data_ptr(), Cin, wtd.data_ptr(), bd.data_ptr(), yt.data_ptr(), Cout, B, H, W, Cin, Cout, 1, stream())
        torch.cuda.synchronize()
        assert rel_l2(yt.double().cpu().numpy(), np.maximum(O.convT4s2_fwd(x, wt, b), 0)) <= TOL_OUT[dt]
        dzt = rnd(rng.standard_normal((B, 2 * H, 2 * W, Cout)), dt)
        dzt_dev = dev(dzt, dt, gpu)     # named: a temporary would be freed (and reused) before the kernel runs
        dxt = torch.zeros(B, H, W, Cin, dtype=TDT[dt], device=gpu)
        L.call("gct2_convT4s2_dgrad", ctx(), dt, dzt_dev.data_ptr(), Cout, wtd.data_ptr(), xd.data_ptr(), Cin, dxt.data_ptr(), Cin,
               B, H, W, Cin, Cout, 0, None, 0, None, 0, stream())
        torch.cuda.synchronize()
        assert rel_l2(dxt.double().cpu().numpy(), O.convT4s2_bwd(x, wt, dzt)[0] * (x > 0)) <= TOL_OUT[dt]
        assert not bool(torch.isnan(ws).all())                          # the slabs were really used
    finally:
        set_ws(None)


@pytest.mark.parametrize("variant", [2, 5])
@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 128), (1, 12, 20, 72, 136), (3, 2, 2, 256, 64), (1, 32, 32, 128, 256), (1, 8, 12, 264, 328),
                                   (2, 16, 16, 512, 256)])
def test_tapgemm_tile_variants(gpu, variant, shape):
    """both tiles on every use: 2 = 128x128 with two LDS buffers, 5 = 256x128 with one buffer and 8 waves (N <= 64 falls through to
    the 256x64 tile); ragged M / K / N, one to 128 reduction steps per work-group."""
    B, H, W, Cin, Cout = shape
    dt = BF16
    L = lib()
    set_tuning(variant)
    try:
        rng = np.random.default_rng(13)
        x = rnd(np.maximum(rng.standard_normal((B, H, W, Cin)), 0), dt)
        w = rnd(rng.standard_normal((4, 4, Cin, Cout)) * 0.1, dt)
        wt = rnd(rng.standard_normal((4, 4, Cout, Cin)) * 0.1, dt)
        b = rng.standard_normal(Cout).astype(np.float32).astype(np.float64)
        xd, wd, wtd, bd = dev(x, dt, gpu), dev(w, dt, gpu), dev(wt, dt, gpu), torch.tensor(b, dtype=torch.float32, device=gpu)
        y = torch.zeros(B, H // 2, W // 2, Cout, dtype=TDT[dt], device=gpu)
        L.call("gct2_conv4s2_fwd", ctx(), dt, xd.data_ptr(), Cin, wd.data_ptr(), bd.data_ptr(), y.data_ptr(), Cout, B, H, W, Cin, Cout, 1, stream())
        yt = torch.zeros(B, 2 * H, 2 * W, Cout, dtype=TDT[dt], device=gpu)
        L.call("gct2_convT4s2_fwd", ctx(), dt, xd.data_ptr(), Cin, wtd.data_ptr(), bd.data_ptr(), yt.data_ptr(), Cout, B, H, W, Cin, Cout, 1, stream())
        dz = rnd(rng.standard_normal((B, H // 2, W // 2, Cout)), dt)
        dz_dev = dev(dz, dt, gpu)       # named: a temporary would be freed (and reused) before the kernel runs
        dx = torch.zeros(B, H, W, Cin, dtype=TDT[dt], device=gpu)
        L.call("gct2_conv4s2_dgrad", ctx(), dt, dz_dev.data_ptr(), Cout, wd.data_ptr(), xd.data_ptr(), Cin, dx.data_ptr(), Cin,
               B, H, W, Cin, Cout, 0, None, 0, None, 0, stream())
        dzt = rnd(rng.standard_normal((B, 2 * H, 2 * W, Cout)), dt)
        dzt_dev = dev(dzt, dt, gpu)
        dxt = torch.zeros(B, H, W, Cin, dtype=TDT[dt], device=gpu)
        L.call("gct2_convT4s2_dgrad", ctx(), dt, dzt_dev.data_ptr(), Cout, wtd.data_ptr(), xd.data_ptr(), Cin, dxt.data_ptr(), Cin,
               B, H, W, Cin, Cout, 0, None, 0, None, 0, stream())
        torch.cuda.synchronize()
        assert rel_l2(y.double().cpu().numpy(), np.maximum(O.conv4s2_fwd(x, w, b), 0)) <= TOL_OUT[dt]
        assert rel_l2(yt.double().cpu().numpy(), np.maximum(O.convT4s2_fwd(x, wt, b), 0)) <= TOL_OUT[dt]
        assert rel_l2(dx.double().cpu().numpy(), O.conv4s2_bwd(x, w, dz)[0] * (x > 0)) <= TOL_OUT[dt]
        assert rel_l2(dxt.double().cpu().numpy(), O.convT4s2_bwd(x, wt, dzt)[0] * (x > 0)) <= TOL_OUT[dt]
    finally:
        set_tuning(0)


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("shape,use_ws", [((2, 16, 16, 64, 128), False), ((1, 12, 20, 72, 136), False), ((4, 4, 4, 512, 256), True),
                                          ((1, 6, 10, 5, 7), False)])
def test_dgrad_fused_bias_gradients(gpu, dt, shape, use_ws):
    """db/db2 of the dgrad entry points = column sums of the masked gradient THIS call produced, split at db_split,
    on the MFMA epilogue, the split-K finalize kernel (use_ws) and the direct path (fp32 / unaligned), incl. accumulate."""
    B, H, W, Cin, Cout = shape
    L = lib()
    ws = torch.empty(32 << 18, dtype=torch.float32, device=gpu)
    set_ws(ws if use_ws else None)
    try:
        rng = np.random.default_rng(15)
        x = rnd(np.maximum(rng.standard_normal((B, H, W, Cin)), 0), dt)
        split = (Cin // 2) // 8 * 8 if Cin >= 16 else 2
        tol = 2e-3 if dt == BF16 else 2e-5
        # Conv2D input gradient, accumulate into a running buffer
        w = rnd(rng.standard_normal((4, 4, Cin, Cout)) * 0.1, dt)
        dz = rnd(rng.standard_normal((B, H // 2, W // 2, Cout)), dt)
        prev = rnd(rng.standard_normal((B, H, W, Cin)), dt)
        contrib = O.conv4s2_bwd(x, w, dz)[0] * (x > 0)
        dxd, dzd, wd, xd = dev(prev, dt, gpu), dev(dz, dt, gpu), dev(w, dt, gpu), dev(x, dt, gpu)   # keep every operand alive
        db = torch.full((split,), 3.0, device=gpu); db2 = torch.full((Cin - split,), -1.0, device=gpu)
        L.call("gct2_conv4s2_dgrad", ctx(), dt, dzd.data_ptr(), Cout, wd.data_ptr(), xd.data_ptr(), Cin,
               dxd.data_ptr(), Cin, B, H, W, Cin, Cout, 1, db.data_ptr(), split, db2.data_ptr(), 3, stream())
        torch.cuda.synchronize()
        cs = contrib.reshape(-1, Cin).sum(0)
        scale = np.abs(contrib).reshape(-1, Cin).sum(0).max()
        assert np.abs(db.cpu().numpy() - 3.0 - cs[:split]).max() <= tol * scale
        assert np.abs(db2.cpu().numpy() + 1.0 - cs[split:]).max() <= tol * scale
        assert rel_l2(dxd.double().cpu().numpy(), contrib + prev) <= TOL_OUT[dt]
        # Conv2DTranspose input gradient, only the second range wanted
        wt = rnd(rng.standard_normal((4, 4, Cout, Cin)) * 0.1, dt)
        dzt = rnd(rng.standard_normal((B, 2 * H, 2 * W, Cout)), dt)
        contrib_t = O.convT4s2_bwd(x, wt, dzt)[0] * (x > 0)
        dxt, dztd, wtd = torch.zeros(B, H, W, Cin, dtype=TDT[dt], device=gpu), dev(dzt, dt, gpu), dev(wt, dt, gpu)
        # db_accumulate = 0: the target is overwritten, whatever it held (here the sums of the previous call)
        L.call("gct2_convT4s2_dgrad", ctx(), dt, dztd.data_ptr(), Cout, wtd.data_ptr(), xd.data_ptr(), Cin,
               dxt.data_ptr(), Cin, B, H, W, Cin, Cout, 0, None, split, db2.data_ptr(), 0, stream())
        torch.cuda.synchronize()
        cst = contrib_t.reshape(-1, Cin).sum(0)
        assert np.abs(db2.cpu().numpy() - cst[split:]).max() <= tol * np.abs(contrib_t).reshape(-1, Cin).sum(0).max()
    finally:
        set_ws(None)


@pytest.mark.parametrize("queue_floats", [1 << 20, 3000])
def test_bias_queue_flush_equals_the_immediate_row_sums(gpu, queue_floats):
    """ABI v16: with a bias queue registered the input-gradient calls leave the partial rows of their fused bias gradients in the queue
    and ONE flush sums every recorded row set (overwritten targets first, added-to targets second) - same geometry, same order, hence
    the same bits as the per-call reduction launches.  A chain like the reverse pass: a Conv2DTranspose input gradient that OVERWRITES
    two targets (MFMA epilogue), a Conv2D input gradient through the split-K finalize that ADDS to the second of them, a halo-kernel
    call, then a second adder of the same target (forces the early flush); with a queue that holds everything and with one so small
    that calls fall back to the immediate form in the middle."""
    dt, L = BF16, lib()
    rng = np.random.default_rng(77)
    ws = torch.empty(32 << 18, dtype=torch.float32, device=gpu)
    set_ws(ws)
    B, H, W, Cin, Cout = 2, 16, 16, 128, 64
    split = 64
    x = rnd(np.maximum(rng.standard_normal((B, H, W, Cin)), 0), dt)
    wt = rnd(rng.standard_normal((4, 4, Cout, Cin)) * 0.1, dt)
    dzt = rnd(rng.standard_normal((B, 2 * H, 2 * W, Cout)), dt)
    w = rnd(rng.standard_normal((4, 4, Cin, Cout)) * 0.1, dt)
    dz = rnd(rng.standard_normal((B, H // 2, W // 2, Cout)), dt)
    xd, wtd, dztd, wd, dzd = dev(x, dt, gpu), dev(wt, dt, gpu), dev(dzt, dt, gpu), dev(w, dt, gpu), dev(dz, dt, gpu)
    # a halo-kernel shape (16-multiples, 64-channel multiples): Conv2D input gradient = convT form
    B2, H2, W2, C2in, C2out = 2, 32, 32, 64, 128
    x2 = rnd(np.maximum(rng.standard_normal((B2, H2, W2, C2in)), 0), dt)
    w2 = rnd(rng.standard_normal((4, 4, C2in, C2out)) * 0.1, dt)
    dz2 = rnd(rng.standard_normal((B2, H2 // 2, W2 // 2, C2out)), dt)
    x2d, w2d, dz2d = dev(x2, dt, gpu), dev(w2, dt, gpu), dev(dz2, dt, gpu)

    def chain(queue):
        if queue is not None:
            L.call("gct2_ctx_set_bias_queue", ctx(), queue.data_ptr(), queue.numel() * 4)
        dba = torch.full((split,), float("nan"), device=gpu); dbb = torch.full((Cin - split,), float("nan"), device=gpu)
        dbc = torch.full((C2in,), 7.0, device=gpu)
        dx = torch.zeros(B, H, W, Cin, dtype=TDT[dt], device=gpu)
        dx2 = torch.zeros(B2, H2, W2, C2in, dtype=TDT[dt], device=gpu)
        try:
            L.call("gct2_convT4s2_dgrad", ctx(), dt, dztd.data_ptr(), Cout, wtd.data_ptr(), xd.data_ptr(), Cin, dx.data_ptr(), Cin,
                   B, H, W, Cin, Cout, 0, dba.data_ptr(), split, dbb.data_ptr(), 0, stream())
            L.call("gct2_conv4s2_dgrad", ctx(), dt, dzd.data_ptr(), Cout, wd.data_ptr(), xd.data_ptr(), Cin, dx.data_ptr(), Cin,
                   B, H, W, Cin, Cout, 1, None, split, dbb.data_ptr(), 2, stream())
            L.call("gct2_conv4s2_dgrad", ctx(), dt, dz2d.data_ptr(), C2out, w2d.data_ptr(), x2d.data_ptr(), C2in, dx2.data_ptr(), C2in,
                   B2, H2, W2, C2in, C2out, 0, dbc.data_ptr(), C2in, None, 1, stream())
            L.call("gct2_conv4s2_dgrad", ctx(), dt, dzd.data_ptr(), Cout, wd.data_ptr(), xd.data_ptr(), Cin, dx.data_ptr(), Cin,
                   B, H, W, Cin, Cout, 1, None, split, dbb.data_ptr(), 2, stream())
            if queue is not None:
                L.call("gct2_bias_queue_flush", ctx(), stream())
            torch.cuda.synchronize()
        finally:
            if queue is not None:
                L.call("gct2_ctx_set_bias_queue", ctx(), None, 0)
        return dba, dbb, dbc, dx, dx2

    try:
        ref = chain(None)
        got = chain(torch.empty(queue_floats, dtype=torch.float32, device=gpu))
        for a, b in zip(ref, got):
            assert torch.equal(a, b)
        assert bool(torch.isfinite(ref[0]).all()) and bool(torch.isfinite(ref[1]).all())
        # ... and against the oracle: the first target holds the first call's sums, the second one three contributions
        c_t = O.convT4s2_bwd(x, wt, dzt)[0] * (x > 0)
        c_c = O.conv4s2_bwd(x, w, dz)[0] * (x > 0)
        want_b = (c_t.reshape(-1, Cin).sum(0) + 2 * c_c.reshape(-1, Cin).sum(0))[split:]
        scale = np.abs(c_t).reshape(-1, Cin).sum(0).max() + 2 * np.abs(c_c).reshape(-1, Cin).sum(0).max()
        assert np.abs(ref[0].cpu().numpy() - c_t.reshape(-1, Cin).sum(0)[:split]).max() <= 2e-3 * scale
        assert np.abs(ref[1].cpu().numpy() - want_b).max() <= 2e-3 * scale
    finally:
        set_ws(None)


@pytest.mark.parametrize("second", ["direct", "no_row_space", "overwrite_after_add", "two_overwrites"])
def test_bias_queue_keeps_program_order_with_immediate_writers(gpu, second):
    """a bias-gradient target keeps its program order whoever writes it.  First call: a Conv2DTranspose input gradient on
    the MFMA path whose OVERWRITE of two targets is queued.  Second call on the same targets:
      direct              - the direct kernels (force_direct) ADD at once: the queued overwrite must be reduced first, not behind it;
      no_row_space        - the MFMA launch has no room for partial rows (workspace withdrawn) and adds with atomics at once: same;
      overwrite_after_add - queued [overwrite, add], then an OVERWRITE: must not be reordered in front of the add;
      two_overwrites      - a second OVERWRITE of the same target: last writer wins, no race inside the first flush launch.
    Reference: the same chain without a queue (every reduction immediate)."""
    dt, L = BF16, lib()
    rng = np.random.default_rng(78)
    ws = torch.empty(32 << 18, dtype=torch.float32, device=gpu)
    B, H, W, Cin, Cout, split = 2, 16, 16, 128, 64, 64
    x = rnd(np.maximum(rng.standard_normal((B, H, W, Cin)), 0), dt)
    wt = rnd(rng.standard_normal((4, 4, Cout, Cin)) * 0.1, dt)
    dzt = rnd(rng.standard_normal((B, 2 * H, 2 * W, Cout)), dt)
    w = rnd(rng.standard_normal((4, 4, Cin, Cout)) * 0.1, dt)
    dz = rnd(rng.standard_normal((B, H // 2, W // 2, Cout)), dt)
    xd, wtd, dztd, wd, dzd = dev(x, dt, gpu), dev(wt, dt, gpu), dev(dzt, dt, gpu), dev(w, dt, gpu), dev(dz, dt, gpu)

    def up(dx, dba, dbb, acc):          # Conv2DTranspose input gradient: channels [0, split) -> dba, the rest -> dbb
        L.call("gct2_convT4s2_dgrad", ctx(), dt, dztd.data_ptr(), Cout, wtd.data_ptr(), xd.data_ptr(), Cin, dx.data_ptr(), Cin,
               B, H, W, Cin, Cout, 0, dba.data_ptr(), split, dbb.data_ptr(), acc, stream())

    def down(dx, dbb, acc):             # Conv2D input gradient accumulated into dx: its skip channels -> dbb
        L.call("gct2_conv4s2_dgrad", ctx(), dt, dzd.data_ptr(), Cout, wd.data_ptr(), xd.data_ptr(), Cin, dx.data_ptr(), Cin,
               B, H, W, Cin, Cout, 1, None, split, dbb.data_ptr(), acc, stream())

    def chain(queue):
        set_ws(ws)
        if queue is not None:
            L.call("gct2_ctx_set_bias_queue", ctx(), queue.data_ptr(), queue.numel() * 4)
        dba = torch.full((split,), float("nan"), device=gpu); dbb = torch.full((Cin - split,), float("nan"), device=gpu)
        dx = torch.zeros(B, H, W, Cin, dtype=TDT[dt], device=gpu)
        try:
            up(dx, dba, dbb, 0)                                # queued: overwrite dba, overwrite dbb
            if second == "direct":
                ctx_obj().force_direct(True)
                down(dx, dbb, 2)                               # immediate add (direct kernels, atomics)
                ctx_obj().force_direct(False)
            elif second == "no_row_space":
                set_ws(None)
                down(dx, dbb, 2)                               # immediate add (atomics in the MFMA epilogue)
                set_ws(ws)
            elif second == "overwrite_after_add":
                down(dx, dbb, 2)                               # queued: add to dbb
                up(dx, dba, dbb, 0)                            # overwrite both again: behind the add, not in front of it
            else:
                up(dx, dba, dbb, 0)                            # second overwrite of both targets
            if queue is not None:
                L.call("gct2_bias_queue_flush", ctx(), stream())
            torch.cuda.synchronize()
        finally:
            ctx_obj().force_direct(False)
            if queue is not None:
                L.call("gct2_ctx_set_bias_queue", ctx(), None, 0)
            set_ws(None)
        return dba, dbb

    ref = chain(None)
    got = chain(torch.empty(1 << 20, dtype=torch.float32, device=gpu))
    assert bool(torch.isfinite(ref[0]).all()) and bool(torch.isfinite(ref[1]).all())
    tol = 0.0 if second in ("overwrite_after_add", "two_overwrites") else 2e-6        # (atomics: the order of the adds is not fixed)
    for a, b in zip(ref, got):
        assert float((a - b).abs().max()) <= tol * float(a.abs().max()), second
    # ... and against the oracle (the second call's contribution must be there / the last overwrite must have won)
    c_t = (O.convT4s2_bwd(x, wt, dzt)[0] * (x > 0)).reshape(-1, Cin).sum(0)
    c_c = (O.conv4s2_bwd(x, w, dz)[0] * (x > 0)).reshape(-1, Cin).sum(0)
    want_b = c_t[split:] + (c_c[split:] if second in ("direct", "no_row_space") else 0.0)
    scale = np.abs(c_t).max() + np.abs(c_c).max()
    assert np.abs(got[0].cpu().numpy() - c_t[:split]).max() <= 4e-3 * scale
    assert np.abs(got[1].cpu().numpy() - want_b).max() <= 4e-3 * scale


@pytest.mark.parametrize("use_ws", [False, True])
@pytest.mark.parametrize("shape", [(2, 8, 8, 256, 64), (1, 16, 16, 328, 72), (8, 4, 4, 512, 256)])
def test_big_tile_dgrad_epilogues(gpu, shape, use_ws):
    """the 256 x 128 tile (variant 5: the automatic choice of the big levels at batch 64) through the input-gradient epilogues: ReLU
    mask, accumulation into a running buffer, the fused bias gradients (partial rows in the workspace / atomics without one), split-K
    slabs + finalize where the workspace allows it (third shape) - against the oracle."""
    B, H, W, Cin, Cout = shape
    dt, L = BF16, lib()
    ws = torch.empty(32 << 18, dtype=torch.float32, device=gpu)
    set_ws(ws if use_ws else None)
    set_tuning(5)
    try:
        rng = np.random.default_rng(23)
        x = rnd(np.maximum(rng.standard_normal((B, H, W, Cin)), 0), dt)
        split = (Cin // 2) // 8 * 8
        w = rnd(rng.standard_normal((4, 4, Cin, Cout)) * 0.1, dt)
        dz = rnd(rng.standard_normal((B, H // 2, W // 2, Cout)), dt)
        prev = rnd(rng.standard_normal((B, H, W, Cin)), dt)
        contrib = O.conv4s2_bwd(x, w, dz)[0] * (x > 0)
        dxd, dzd, wd, xd = dev(prev, dt, gpu), dev(dz, dt, gpu), dev(w, dt, gpu), dev(x, dt, gpu)
        db = torch.full((split,), 3.0, device=gpu); db2 = torch.full((Cin - split,), -1.0, device=gpu)
        L.call("gct2_conv4s2_dgrad", ctx(), dt, dzd.data_ptr(), Cout, wd.data_ptr(), xd.data_ptr(), Cin,
               dxd.data_ptr(), Cin, B, H, W, Cin, Cout, 1, db.data_ptr(), split, db2.data_ptr(), 3, stream())
        torch.cuda.synchronize()
        cs = contrib.reshape(-1, Cin).sum(0)
        scale = np.abs(contrib).reshape(-1, Cin).sum(0).max()
        assert np.abs(db.cpu().numpy() - 3.0 - cs[:split]).max() <= 2e-3 * scale
        assert np.abs(db2.cpu().numpy() + 1.0 - cs[split:]).max() <= 2e-3 * scale
        assert rel_l2(dxd.double().cpu().numpy(), contrib + prev) <= TOL_OUT[dt]
        wt = rnd(rng.standard_normal((4, 4, Cout, Cin)) * 0.1, dt)
        dzt = rnd(rng.standard_normal((B, 2 * H, 2 * W, Cout)), dt)
        contrib_t = O.convT4s2_bwd(x, wt, dzt)[0] * (x > 0)
        dxt, dztd, wtd = torch.zeros(B, H, W, Cin, dtype=TDT[dt], device=gpu), dev(dzt, dt, gpu), dev(wt, dt, gpu)
        L.call("gct2_convT4s2_dgrad", ctx(), dt, dztd.data_ptr(), Cout, wtd.data_ptr(), xd.data_ptr(), Cin,
               dxt.data_ptr(), Cin, B, H, W, Cin, Cout, 0, db.data_ptr(), split, db2.data_ptr(), 0, stream())
        torch.cuda.synchronize()
        cst = contrib_t.reshape(-1, Cin).sum(0)
        scale_t = np.abs(contrib_t).reshape(-1, Cin).sum(0).max()
        assert np.abs(db.cpu().numpy() - cst[:split]).max() <= 2e-3 * scale_t
        assert np.abs(db2.cpu().numpy() - cst[split:]).max() <= 2e-3 * scale_t
        assert rel_l2(dxt.double().cpu().numpy(), contrib_t) <= TOL_OUT[dt]
        # bit-reproducible (ordered partial rows / slabs) when a workspace is there
        if use_ws:
            dxt2 = torch.zeros_like(dxt); db_a, db2_a = db.clone(), db2.clone()
            for _ in range(3):
                L.call("gct2_convT4s2_dgrad", ctx(), dt, dztd.data_ptr(), Cout, wtd.data_ptr(), xd.data_ptr(), Cin,
                       dxt2.data_ptr(), Cin, B, H, W, Cin, Cout, 0, db.data_ptr(), split, db2.data_ptr(), 0, stream())
                torch.cuda.synchronize()
                assert torch.equal(dxt2, dxt) and torch.equal(db, db_a) and torch.equal(db2, db2_a)
    finally:
        set_ws(None)
        set_tuning(0)


def test_mfma_and_direct_paths_agree(gpu):
    """the same bf16 problem through the MFMA path and the direct path (forced): independent kernels."""
    B, H, W, Cin, Cout = 2, 8, 8, 128, 128
    rng = np.random.default_rng(7)
    x, w = dev(rng.standard_normal((B, H, W, Cin)), BF16, gpu), dev(rng.standard_normal((4, 4, Cin, Cout)) * 0.1, BF16, gpu)
    outs = []
    for force in (0, 1):
        _CTX[0].force_direct(force)
        y = torch.zeros(B, H // 2, W // 2, Cout, dtype=torch.bfloat16, device=gpu)
        lib().call("gct2_conv4s2_fwd", ctx(), BF16, x.data_ptr(), Cin, w.data_ptr(), None, y.data_ptr(), Cout, B, H, W, Cin, Cout, 0, stream())
        torch.cuda.synchronize()
        outs.append(y.double().cpu().numpy())
    _CTX[0].force_direct(0)
    assert rel_l2(outs[0], outs[1]) <= 4e-3


@pytest.mark.parametrize("dt", [F32, BF16, F16])
def test_dense_fwd_bwd(gpu, dt):
    M, Cin, Cout, ld, Cmask = 1000, 67, 3, 72, 64
    rng = np.random.default_rng(8)
    x = rnd(np.maximum(rng.standard_normal((M, Cin)), 0), dt)
    w = rng.standard_normal((Cin, Cout)).astype(np.float32).astype(np.float64)
    b = rng.standard_normal(Cout).astype(np.float32).astype(np.float64)
    dy = rng.standard_normal((M, Cout)).astype(np.float32).astype(np.float64)
    xb = torch.zeros(M, ld, dtype=TDT[dt], device=gpu)
    xb[:, :Cin] = dev(x, dt, gpu)
    wd, bd = torch.tensor(w, dtype=torch.float32, device=gpu), torch.tensor(b, dtype=torch.float32, device=gpu)
    y = torch.zeros(M, Cout, dtype=torch.float32, device=gpu)
    lib().call("gct2_dense_fwd", dt, xb.data_ptr(), ld, wd.data_ptr(), bd.data_ptr(), y.data_ptr(), M, Cin, Cout, stream())
    torch.cuda.synchronize()
    # GCT2_F16 = Keras mixed_float16: the Dense output is an fp16 tensor (cast to fp32 for the loss), and so is the gradient
    # entering it (include/gct2.h)
    y_ref = rnd(x @ w + b, F16) if dt == F16 else x @ w + b
    assert rel_l2(y.cpu().numpy(), y_ref) <= (3e-4 if dt == F16 else 2e-6)
    if dt == F16:
        assert torch.equal(y, y.half().float())
        dy = rnd(dy, F16)
    dyd = torch.tensor(dy, dtype=torch.float32, device=gpu)
    dxb = torch.full((M, ld), 5.0, dtype=TDT[dt], device=gpu)
    dw = torch.full((Cin, Cout), 9.0, dtype=torch.float32, device=gpu)       # accumulate = 0: overwritten
    db = torch.full((Cout,), -9.0, dtype=torch.float32, device=gpu)
    lib().call("gct2_dense_bwd", dt, xb.data_ptr(), ld, wd.data_ptr(), dyd.data_ptr(), dxb.data_ptr(), ld, dw.data_ptr(),
               db.data_ptr(), M, Cin, Cout, Cmask, 0, stream())
    torch.cuda.synchronize()
    dx_ref = (dy @ w.T) * (x > 0)
    assert rel_l2(dxb[:, :Cmask].double().cpu().numpy(), dx_ref[:, :Cmask]) <= TOL_OUT[dt]
    assert float((dxb[:, Cmask:].float() - 5).abs().max()) == 0       # image channels get no gradient
    assert rel_l2(dw.cpu().numpy(), x.T @ dy) <= 2e-5
    assert rel_l2(db.cpu().numpy(), dy.sum(0)) <= 2e-5


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("M", [1000, 256 * 5, 16 * 4 * 512 + 7])
@pytest.mark.parametrize("with_ws", [False, True, "split"])
def test_dense_head_train_fused(gpu, dt, M, with_ws):
    """fused Dense(3) + MSE + both gradients == the three separate kernels' definitions (ragged last tile at M=1000).
    with_ws: a registered workspace selects the matrix-core version (partial rows, no atomics); without it the LDS-tile
    version runs.  The largest M gives every wave of the 512-work-group grid more than one trip.
    "split": the image channels [64, 67) come from a packed second view (x2, ld 4); R_0's own slice holds NaN."""
    split, with_ws = with_ws == "split", bool(with_ws)
    Cin, Cout, ld, Cmask = 67, 3, 72, 64
    ws = torch.full((1 << 20,), float("nan"), device=gpu) if with_ws else None
    set_ws(ws if with_ws else None)
    rng = np.random.default_rng(12)
    x = rnd(np.maximum(rng.standard_normal((M, Cin)), 0), dt)
    w = rng.standard_normal((Cin, Cout)).astype(np.float32).astype(np.float64)
    b = rng.standard_normal(Cout).astype(np.float32).astype(np.float64)
    tgt = rng.uniform(-1, 1, (M, Cout)).astype(np.float32).astype(np.float64)
    xb = torch.zeros(M, ld, dtype=TDT[dt], device=gpu)
    if with_ws:
        xb[:, Cin:] = float("nan")                                  # pad channels must not leak into anything
    xb[:, :Cin] = dev(x, dt, gpu)
    x2 = torch.zeros(M, 4, dtype=TDT[dt], device=gpu)
    if split:
        x2[:, :3] = xb[:, Cmask:Cin]
        xb[:, Cmask:] = float("nan")
    t32 = lambda a: torch.tensor(a, dtype=torch.float32, device=gpu)
    wd, bd, td = t32(w), t32(b), t32(tgt)
    pred = torch.zeros(M, Cout, device=gpu); dxb = torch.full((M, ld), 5.0, dtype=TDT[dt], device=gpu)
    dw = torch.full((Cin, Cout), 7.0, device=gpu); db = torch.full((Cout,), 7.0, device=gpu)      # accumulate = 0: overwritten
    loss = torch.zeros(1, device=gpu); part = torch.zeros(1024, device=gpu)
    scale = torch.tensor([8.0], device=gpu); dbx = torch.full((Cmask,), 7.0, device=gpu)
    try:
        lib().call("gct2_dense_head_train", ctx(), dt, xb.data_ptr(), ld, wd.data_ptr(), bd.data_ptr(), td.data_ptr(), pred.data_ptr(),
                   dxb.data_ptr(), ld, dw.data_ptr(), db.data_ptr(), loss.data_ptr(), part.data_ptr(), M, Cin, Cout, Cmask,
                   scale.data_ptr(), dbx.data_ptr(), x2.data_ptr() if split else None, 4 if split else 0, 0, stream())
        torch.cuda.synchronize()
    finally:
        set_ws(None)
    pr = x @ w + b
    if dt == F16:       # Keras mixed_float16 rounding points: fp16 Dense output, fp16 gradient entering it
        pr = rnd(pr, F16)
    d = pr - tgt
    dp = 8.0 * 2 * d / d.size
    if dt == F16:
        dp = rnd(dp, F16)
    assert rel_l2(pred.cpu().numpy(), pr) <= (3e-4 if dt == F16 else 2e-6)
    assert abs(float(loss[0]) - np.mean(d * d)) <= (2e-4 if dt == F16 else 2e-6) * np.mean(d * d)
    if dt == F16:
        # the parity quantities below are taken against the kernel's own fp16 prediction (a rounding flip of pred moves d by one
        # fp16 ulp of pred, i.e. up to 1e-3 relative on a single element)
        pr = pred.double().cpu().numpy()
        d = pr - tgt
        dp = rnd(8.0 * 2 * d / d.size, F16)
    assert rel_l2(dxb[:, :Cmask].double().cpu().numpy(), ((dp @ w.T) * (x > 0))[:, :Cmask]) <= TOL_OUT[dt]
    assert float((dxb[:, Cmask:].float() - 5).abs().max()) == 0
    assert rel_l2(dw.cpu().numpy(), x.T @ dp) <= 2e-5 and rel_l2(db.cpu().numpy(), dp.sum(0)) <= 2e-5
    if with_ws:     # column sums of the fp32 gradient rows before they are rounded for the store
        assert rel_l2(dbx.cpu().numpy(), ((dp @ w.T) * (x > 0))[:, :Cmask].sum(0)) <= 2e-5
    else:           # column sums of the stored rows
        assert rel_l2(dbx.cpu().numpy(), dxb[:, :Cmask].double().cpu().numpy().sum(0)) <= 1e-5


def test_noise_mse(gpu):
    B, HW, C, steps = 5, 64, 3, 200
    rng = np.random.default_rng(9)
    x = rng.uniform(-1, 1, (B, HW, C)).astype(np.float32)
    eps = rng.standard_normal((B, HW, C)).astype(np.float32)
    t = rng.integers(1, steps + 1, B).astype(np.int32)
    ref = O.noise_image(x.astype(np.float64).reshape(B, HW, 1, C), t, eps.astype(np.float64).reshape(B, HW, 1, C), steps).reshape(B * HW, C)
    xd, ed, td = torch.tensor(x, device=gpu), torch.tensor(eps, device=gpu), torch.tensor(t, device=gpu)
    out = torch.zeros(B * HW, 8, dtype=torch.float32, device=gpu)
    out2 = torch.zeros(B * HW, 4, dtype=torch.float32, device=gpu)     # the packed copy (ld 4)
    lib().call("gct2_noise_image", F32, xd.data_ptr(), td.data_ptr(), ed.data_ptr(), out.data_ptr() + 4 * 2, 8, out2.data_ptr(), 4,
               B, HW, C, steps, stream())
    torch.cuda.synchronize()
    assert rel_l2(out[:, 2:5].cpu().numpy(), ref) <= 1e-6
    assert torch.equal(out2[:, :3], out[:, 2:5]) and float(out2[:, 3].abs().max()) == 0
    # MSE + gradient
    n = 100003
    pred, tgt = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    pd_, tg = torch.tensor(pred, device=gpu), torch.tensor(tgt, device=gpu)
    dp = torch.zeros(n, device=gpu); loss = torch.zeros(1, device=gpu); part = torch.zeros(1024, device=gpu)
    lib().call("gct2_mse_fwd_bwd", pd_.data_ptr(), tg.data_ptr(), dp.data_ptr(), loss.data_ptr(), part.data_ptr(), n, None, stream())
    torch.cuda.synchronize()
    d = pred.astype(np.float64) - tgt
    assert abs(float(loss[0]) - np.mean(d * d)) <= 1e-6 * np.mean(d * d)
    assert rel_l2(dp.cpu().numpy(), 2 * d / n) <= 1e-6


def test_noise_with_inkernel_rng_is_bit_identical(gpu):
    """gct2_noise_image_rng == gct2_rng_normal + gct2_noise_image on the same stream positions (and returns the draws)."""
    B, HW, C, steps, off = 3, 50, 3, 200, 12345
    n = B * HW * C
    x = torch.rand(B, HW, C, device=gpu) * 2 - 1
    t = torch.tensor([1, 77, 200], dtype=torch.int32, device=gpu)
    eps = torch.zeros(n, device=gpu)
    lib().call("gct2_rng_normal", 99, 2, off, eps.data_ptr(), n, stream())
    a = torch.zeros(B * HW, 8, dtype=torch.bfloat16, device=gpu); bb = torch.zeros_like(a)
    lib().call("gct2_noise_image", BF16, x.data_ptr(), t.data_ptr(), eps.data_ptr(), a.data_ptr() + 2 * 4, 8, None, 0, B, HW, C, steps,
               stream())
    eps2 = torch.zeros(n, device=gpu)
    packed = torch.zeros(B * HW, 4, dtype=torch.bfloat16, device=gpu)
    lib().call("gct2_noise_image_rng", BF16, x.data_ptr(), t.data_ptr(), 99, 2, off, eps2.data_ptr(), bb.data_ptr() + 2 * 4, 8,
               packed.data_ptr(), 4, B, HW, C, steps, stream())
    torch.cuda.synchronize()
    assert torch.equal(a, bb) and torch.equal(eps, eps2) and float(a[:, 4:7].float().abs().max()) > 0
    assert torch.equal(packed[:, :3], bb[:, 4:7]) and float(packed[:, 3].float().abs().max()) == 0


@pytest.mark.parametrize("dt", [BF16, F16])
def test_noise_rng_four_pixel_path_is_bit_identical(gpu, dt):
    """r06: the train step's call of gct2_noise_image_rng (3 channels into the packed 4-slot image, nothing else, stream position a
    multiple of 12, whole groups of four pixels) runs a kernel whose threads own three Philox counters = four whole pixels (16-byte
    loads and stores).  Same counters, normals and mix as the general kernel - forced here by asking for eps as well - hence the same
    bits; slot 3 of the packed image is written as zero."""
    B, HW, C, steps = 5, 4 * 37, 3, 200
    n = B * HW * C
    x = ((torch.randint(0, 256, (B, HW, C)).float() / 128) - 1).to(gpu)
    t = torch.tensor([1, 77, 200, 13, 150], dtype=torch.int32, device=gpu)
    for off in (0, 12 * 1000 + 0, 3 * n):
        general = torch.zeros(B * HW, 4, dtype=TDT[dt], device=gpu)
        eps = torch.zeros(n, device=gpu)
        lib().call("gct2_noise_image_rng", dt, x.data_ptr(), t.data_ptr(), 99, 2, off, eps.data_ptr(), general.data_ptr(), 4, None, 0,
                   B, HW, C, steps, stream())
        fast = torch.full((B * HW, 4), 3.0, dtype=TDT[dt], device=gpu)
        lib().call("gct2_noise_image_rng", dt, x.data_ptr(), t.data_ptr(), 99, 2, off, None, fast.data_ptr(), 4, None, 0, B, HW, C, steps, stream())
        torch.cuda.synchronize()
        assert torch.equal(general, fast), off
        assert float(fast[:, 3].float().abs().max()) == 0 and float(fast[:, :3].float().abs().max()) > 0
    # a stream position that is not a multiple of 12 takes the general kernel (slot 3 untouched) and still agrees with rng_normal
    off = 12 * 50 + 5
    fast = torch.full((B * HW, 4), 3.0, dtype=TDT[dt], device=gpu)
    lib().call("gct2_noise_image_rng", dt, x.data_ptr(), t.data_ptr(), 99, 2, off, None, fast.data_ptr(), 4, None, 0, B, HW, C, steps, stream())
    eps = torch.zeros(n, device=gpu)
    lib().call("gct2_rng_normal", 99, 2, off, eps.data_ptr(), n, stream())
    ref = torch.zeros(B * HW, 4, dtype=TDT[dt], device=gpu)
    lib().call("gct2_noise_image", dt, x.data_ptr(), t.data_ptr(), eps.data_ptr(), ref.data_ptr(), 4, None, 0, B, HW, C, steps, stream())
    torch.cuda.synchronize()
    assert torch.equal(fast[:, :3], ref[:, :3]) and float((fast[:, 3].float() - 3).abs().max()) == 0


def test_rng_streams(gpu):
    """distributional checks (TF's Philox stream cannot be reproduced, SURVEY.md §8c 'RNG')."""
    n = 1 << 20
    t = torch.zeros(n, dtype=torch.int32, device=gpu)
    e = torch.zeros(n, dtype=torch.float32, device=gpu)
    lib().call("gct2_rng_uniform_int", 123, 1, 0, t.data_ptr(), n, 1, 200, stream())
    lib().call("gct2_rng_normal", 123, 2, 0, e.data_ptr(), n, stream())
    torch.cuda.synchronize()
    tn, en = t.cpu().numpy(), e.cpu().numpy().astype(np.float64)
    assert tn.min() == 1 and tn.max() == 200
    counts = np.bincount(tn, minlength=201)[1:]
    assert abs(counts - n / 200).max() < 6 * np.sqrt(n / 200)
    assert abs(en.mean()) < 5e-3 and abs(en.std() - 1) < 5e-3
    assert abs(np.mean(en ** 3)) < 2e-2 and abs(np.mean(en ** 4) - 3) < 5e-2
    # counter-based: drawing [0,n) in two calls with offsets equals one call
    e2 = torch.zeros(n, dtype=torch.float32, device=gpu)
    h = n // 2 + 3
    lib().call("gct2_rng_normal", 123, 2, 0, e2.data_ptr(), h, stream())
    lib().call("gct2_rng_normal", 123, 2, h, e2.data_ptr() + 4 * h, n - h, stream())
    torch.cuda.synchronize()
    assert torch.equal(e, e2)


@pytest.mark.parametrize("sdt", [F32, BF16, F16])
def test_adam_keras(gpu, sdt):
    cfg = O.OracleConfig()
    n = 4099   # not a multiple of 4: exercises the tail
    rng = np.random.default_rng(10)
    p = rng.standard_normal(n).astype(np.float32); g = (rng.standard_normal(n) * 1e-4).astype(np.float32)
    m = (rng.standard_normal(n) * 1e-4).astype(np.float32); v = (rng.random(n) * 1e-8).astype(np.float32)
    k = 7
    pr, mr, vr = O.keras_adam_step(p, g, m, v, k, cfg)
    import math
    alpha = O.warmup_lr(k, cfg.base_lr, cfg.warm_up) * math.sqrt(1 - cfg.beta_2 ** (k + 1)) / (1 - cfg.beta_1 ** (k + 1))
    t = lambda a: torch.tensor(a, device=gpu)
    pd_, md, vd, gd = t(p), t(m), t(v), t(g)
    sh = torch.zeros(n, dtype=TDT[sdt], device=gpu)
    gd.mul_(4.0)    # as if summed over 4 data-parallel ranks: grad_mul = 1/4 restores the mean
    lib().call("gct2_adam_keras_multi", pd_.data_ptr(), md.data_ptr(), vd.data_ptr(), gd.data_ptr(), sh.data_ptr(), sdt, n,
               alpha, cfg.beta_1, cfg.beta_2, cfg.epsilon, 0.25, None, 1, stream())
    torch.cuda.synchronize()
    assert rel_l2(pd_.cpu().numpy(), pr) <= 1e-6 and rel_l2(md.cpu().numpy(), mr) <= 1e-6 and rel_l2(vd.cpu().numpy(), vr) <= 1e-6
    assert float(gd.abs().max()) == 0                       # zero_grad
    assert torch.equal(sh, pd_.to(TDT[sdt]))                # shadow = rounded copy of the new p
    # Keras epsilon placement: differs measurably from sqrt(v_hat)+eps
    upd = (p - pd_.cpu().numpy())
    torch_style = alpha * 0 + O.warmup_lr(k, cfg.base_lr, cfg.warm_up) * (mr / (1 - cfg.beta_1 ** (k + 1))) / (np.sqrt(vr / (1 - cfg.beta_2 ** (k + 1))) + cfg.epsilon)
    assert rel_l2(upd, torch_style) > 1e-3


def test_loss_scale_state_machine(gpu):
    """gct2_loss_scale_state: dynamic scale (x2 after growth_interval finite steps, /2 + skip on inf/nan) and the optimizer step
    counter it gates - a skipped step advances neither applied_steps nor the WarmUp / bias-correction factor alpha [TF]."""
    import math
    st = torch.zeros(8, dtype=torch.int32, device=gpu)
    L = lib()
    L.call("gct2_loss_scale_init", st.data_ptr(), 2.0 ** 15, stream())
    ref = O.LossScaleState(growth_interval=3)
    base_lr, warm, b1, b2 = 2e-5, 4, 0.9, 0.999
    g_ok = torch.ones(1000, device=gpu)
    g_bad = g_ok.clone(); g_bad[777] = float("inf")
    g_nan = g_ok.clone(); g_nan[3] = float("nan")
    p = torch.zeros(1000, device=gpu); m = torch.zeros(1000, device=gpu); v = torch.zeros(1000, device=gpu)
    applied_ref = 0
    for step, g in enumerate([g_ok, g_ok, g_bad, g_ok, g_ok, g_ok, g_nan, g_ok]):
        gg = g.clone()
        L.call("gct2_loss_scale_begin", st.data_ptr(), base_lr, warm, b1, b2, stream())
        L.call("gct2_scale_check_finite", gg.data_ptr(), gg.numel(), st.data_ptr(), stream())
        p_before = p.clone()
        torch.cuda.synchronize()
        alpha_dev = float(st.cpu()[5:6].view(torch.float32)[0])
        k = applied_ref
        lr = float(np.float32(base_lr) * np.float32(k + 1) / np.float32(warm + 1)) if k < warm else base_lr
        b1f, b2f = float(np.float32(b1)), float(np.float32(b2))
        alpha_ref = lr * math.sqrt(1 - b2f ** (k + 1)) / (1 - b1f ** (k + 1))
        assert abs(alpha_dev - alpha_ref) <= 2e-6 * alpha_ref, (step, alpha_dev, alpha_ref)
        # the host's alpha argument is ignored when the state is passed (1e3 would be visible)
        L.call("gct2_adam_keras_multi", p.data_ptr(), m.data_ptr(), v.data_ptr(), gg.data_ptr(), None, 0, 1000, 1e3, b1, b2,
               1e-7, 1.0, st.data_ptr(), 0, stream())
        L.call("gct2_loss_scale_update", st.data_ptr(), 3, stream())
        torch.cuda.synchronize()
        finite = bool(torch.isfinite(g).all())
        applied = ref.update(finite)
        applied_ref += int(applied)
        raw = st.cpu()
        assert float(raw[:1].view(torch.float32)[0]) == ref.scale and int(raw[2]) == ref.good_steps
        assert int(raw[4]) == applied_ref
        assert bool((p != p_before).any()) == applied          # update skipped on inf/nan
        assert bool(torch.isfinite(p).all()) and float(p.abs().max()) < 1.0


def test_noise_fp16_follows_mixed_float16_arithmetic(gpu):
    """GCT2_F16 noising = train.py:229-234 on fp16 tensors (mixed_float16: x, eps, t are fp16; every op rounds to fp16),
    bit for bit against the numpy float16 restatement; the in-kernel-RNG form is identical on the same draws."""
    B, HW, C, steps = 7, 96, 3, 200
    rng = np.random.default_rng(19)
    x = (rng.integers(0, 256, (B, HW, C)) / 128.0 - 1.0).astype(np.float32)
    eps = rng.standard_normal((B, HW, C)).astype(np.float32)
    t = np.array([1, 2, 25, 100, 137, 199, 200], dtype=np.int32)
    ref = O.noise_image_f16(x.reshape(B, HW, 1, C), t, eps.reshape(B, HW, 1, C), steps).reshape(B * HW, C)
    xd, ed, td = torch.tensor(x, device=gpu), torch.tensor(eps, device=gpu), torch.tensor(t, device=gpu)
    out = torch.zeros(B * HW, 4, dtype=torch.float16, device=gpu)
    lib().call("gct2_noise_image", F16, xd.data_ptr(), td.data_ptr(), ed.data_ptr(), out.data_ptr(), 4, None, 0, B, HW, C, steps, stream())
    torch.cuda.synchronize()
    got = out[:, :3].cpu().numpy()
    assert got.dtype == np.float16 and ref.dtype == np.float16
    mism = np.mean(got != ref)
    assert mism <= 2e-3, mism                    # pow/sqrt of the two libraries may differ in the last fp16 bit on rare inputs
    assert np.abs(got.astype(np.float64) - ref.astype(np.float64)).max() <= 2e-3


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 64), (1, 32, 16, 136, 72), (3, 16, 32, 256, 128), (1, 16, 16, 8, 8)])
def test_convT_fwd_halo_kernel(gpu, dt, shape):
    """halo-tile Conv2DTranspose forward (one staged 18x18 source patch for all phases and taps; shift-invariant LDS swizzle):
    forced on (variant bits 24-25 = 2) and compared with the oracle; shapes cover ragged K and N, several patches per image
    in both directions, several images, and the smallest channel counts."""
    B, H, W, Cin, Cout = shape
    L = lib()
    rng = np.random.default_rng(31)
    x = rnd(np.maximum(rng.standard_normal((B, H, W, Cin)), 0), dt)
    wt = rnd(rng.standard_normal((4, 4, Cout, Cin)) * 0.1, dt)
    b = rng.standard_normal(Cout).astype(np.float32).astype(np.float64)
    xd, wtd, bd = dev(x, dt, gpu), dev(wt, dt, gpu), torch.tensor(b, dtype=torch.float32, device=gpu)
    ref = np.maximum(O.convT4s2_fwd(x, wt, b), 0)
    outs = []
    for mode in ((2 << 24), (1 << 24)):                          # halo forced / off: each against the oracle
        set_tuning(mode)
        try:
            ld = Cout + 16                # view = channels [8, 8 + Cout): 16-byte aligned rows (the halo kernel's epilogue needs that)
            yt = torch.full((B, 2 * H, 2 * W, ld), 7.0, dtype=TDT[dt], device=gpu)
            L.call("gct2_convT4s2_fwd", ctx(), dt, xd.data_ptr(), Cin, wtd.data_ptr(), bd.data_ptr(), yt.data_ptr() + 8 * yt.element_size(), ld,
                   B, H, W, Cin, Cout, 1, stream())
            torch.cuda.synchronize()
        finally:
            set_tuning(0)
        assert rel_l2(yt[..., 8:8 + Cout].double().cpu().numpy(), ref) <= TOL_OUT[dt], mode
        assert float((yt[..., :8].float() - 7).abs().max()) == 0 and float((yt[..., 8 + Cout:].float() - 7).abs().max()) == 0
        outs.append(yt)
    assert rel_l2(outs[0].double().cpu().numpy(), outs[1].double().cpu().numpy()) <= TOL_OUT[dt]


@pytest.mark.parametrize("use_ws", [False, True])
@pytest.mark.parametrize("shape", [(2, 32, 32, 64, 24), (1, 32, 64, 136, 64)])
def test_conv_dgrad_halo_kernel(gpu, shape, use_ws):
    """the halo-tile kernel as the Conv2D input gradient (mask, accumulation into a running buffer, fused bias gradients with the
    split at db_split, partial rows with a workspace / atomics without): forced on and compared with the oracle."""
    B, H, W, Cin, Cout = shape                     # dgrad output [B,H,W,Cin]; small grid H/2 x W/2 must tile into 16 x 16
    dt = BF16
    L = lib()
    ws = torch.empty(8 << 18, dtype=torch.float32, device=gpu)
    set_ws(ws if use_ws else None)
    set_tuning(2 << 24)
    try:
        rng = np.random.default_rng(41)
        x = rnd(np.maximum(rng.standard_normal((B, H, W, Cin)), 0), dt)
        w = rnd(rng.standard_normal((4, 4, Cin, Cout)) * 0.1, dt)
        dz = rnd(rng.standard_normal((B, H // 2, W // 2, Cout)), dt)
        prev = rnd(rng.standard_normal((B, H, W, Cin)), dt)
        contrib = O.conv4s2_bwd(x, w, dz)[0] * (x > 0)
        split = (Cin // 2) // 8 * 8
        dxd, dzd, wd, xd = dev(prev, dt, gpu), dev(dz, dt, gpu), dev(w, dt, gpu), dev(x, dt, gpu)
        db = torch.full((split,), 3.0, device=gpu); db2 = torch.full((Cin - split,), -1.0, device=gpu)
        L.call("gct2_conv4s2_dgrad", ctx(), dt, dzd.data_ptr(), Cout, wd.data_ptr(), xd.data_ptr(), Cin,
               dxd.data_ptr(), Cin, B, H, W, Cin, Cout, 1, db.data_ptr(), split, db2.data_ptr(), 3, stream())
        torch.cuda.synchronize()
        cs = contrib.reshape(-1, Cin).sum(0)
        scale = np.abs(contrib).reshape(-1, Cin).sum(0).max()
        assert np.abs(db.cpu().numpy() - 3.0 - cs[:split]).max() <= 2e-3 * scale
        assert np.abs(db2.cpu().numpy() + 1.0 - cs[split:]).max() <= 2e-3 * scale
        assert rel_l2(dxd.double().cpu().numpy(), contrib + prev) <= TOL_OUT[dt]
    finally:
        set_tuning(0)
        set_ws(None)


def _packbits(y, C):
    """bit k of byte c <-> channel 8c + k (include/gct2.h, gct2_ctx_set_relu_bits)"""
    return np.packbits((y[..., :C] > 0).reshape(-1, C), axis=1, bitorder="little")


@pytest.mark.parametrize("words", [False, True])
@pytest.mark.parametrize("case", ["conv_wide", "conv_narrow_view", "conv_splitk", "conv_rgb_staged", "conv_rgb_small", "conv_f32",
                                  "convT_halo", "convT_tap", "convT_splitk"])
def test_relu_bit_plane_written_by_forward_calls(gpu, case, words):
    """gct2_ctx_set_relu_bits before a forward call: the call also leaves bits = (y > 0), one byte per 8 channels, written by the
    16-byte epilogues (tap GEMM, halo kernel, image layer) or derived from the stored y on the other paths (8-byte epilogue of an
    unaligned view, split-K finalize, fp32 direct kernels); y itself is what the call without a plane produces, bit for bit; the
    plane is one-shot (a second call without a new registration leaves the bytes alone); bytes outside the view stay untouched.
    words: a 4-byte aligned plane (the engine's): the four lane rows of a pixel merge their bytes into one 32-bit store."""
    L = lib()
    dt = F32 if case == "conv_f32" else BF16
    kind = "convT" if case.startswith("convT") else "conv"
    B, H, W, Cin, Cout, ld_extra, tuning = {
        "conv_wide": (2, 32, 32, 64, 128, 8, 0), "conv_narrow_view": (2, 16, 16, 64, 64, 4, 0), "conv_splitk": (2, 8, 8, 128, 256, 0, 0),
        "conv_rgb_staged": (2, 32, 32, 3, 128, 0, 0), "conv_rgb_small": (2, 32, 32, 3, 64, 0, 0), "conv_f32": (1, 16, 16, 16, 32, 0, 0),
        "convT_halo": (2, 16, 16, 64, 64, 8, 2 << 24), "convT_tap": (2, 16, 16, 64, 128, 0, 1 << 24), "convT_splitk": (2, 4, 4, 256, 128, 0, 0),
    }[case]
    rng = np.random.default_rng(91)
    ldx = 4 if Cin == 3 else Cin
    x = rnd(rng.standard_normal((B, H, W, Cin)), dt)
    xd = torch.zeros(B, H, W, ldx, dtype=TDT[dt], device=gpu)
    xd[..., :Cin] = dev(x, dt, gpu)
    wshape = (4, 4, Cin, Cout) if kind == "conv" else (4, 4, Cout, Cin)
    wd = dev(rnd(rng.standard_normal(wshape) * 0.1, dt), dt, gpu)
    bd = torch.tensor(rng.standard_normal(Cout), dtype=torch.float32, device=gpu)
    Ho, Wo = (H // 2, W // 2) if kind == "conv" else (2 * H, 2 * W)
    ld = Cout + ld_extra
    off = ld_extra // 2 if ld_extra == 4 else ld_extra          # narrow view: starts 2 elements in (4-byte aligned only)
    ldb, boff = (Cout // 8 + 3, 1) if not words else ((Cout // 8 + 7) // 4 * 4, 4)
    outs = []
    ws = torch.empty(4 << 18, dtype=torch.float32, device=gpu)
    set_ws(ws)
    set_tuning(tuning)
    try:
        for with_bits in (False, True):
            y = torch.full((B, Ho, Wo, ld), 7.0, dtype=TDT[dt], device=gpu)
            bits = torch.full((B * Ho * Wo, ldb), 0xA5, dtype=torch.uint8, device=gpu)
            for rep in range(2):                                # second call: no new registration -> the plane is not touched
                if with_bits and rep == 0:
                    ctx_obj().set_relu_bits(bits.data_ptr() + boff, ldb)
                if rep == 1:
                    bits_before = bits.clone()
                L.call("gct2_conv4s2_fwd" if kind == "conv" else "gct2_convT4s2_fwd", ctx(), dt, xd.data_ptr(), ldx, wd.data_ptr(), bd.data_ptr(),
                       y.data_ptr() + off * y.element_size(), ld, B, H, W, Cin, Cout, 1, stream())
                torch.cuda.synchronize()
                if rep == 1:
                    assert torch.equal(bits, bits_before)
            outs.append(y)
            if with_bits:
                got = bits.cpu().numpy()
                ref = _packbits(y[..., off:off + Cout].float().cpu().numpy(), Cout)
                assert np.array_equal(got[:, boff:boff + Cout // 8], ref), case
                assert (got[:, :boff] == 0xA5).all() and (got[:, boff + Cout // 8:] == 0xA5).all()
                assert 0.2 < np.unpackbits(ref).mean() < 0.8    # a real mask, not all zeros / ones
        assert torch.equal(outs[0], outs[1])
    finally:
        set_tuning(0)
        set_ws(None)


@pytest.mark.parametrize("case", ["conv_dgrad_halo", "conv_dgrad_tap", "convT_dgrad_tap", "convT_dgrad_big", "conv_dgrad_narrow"])
def test_relu_bit_plane_read_by_input_gradient_calls(gpu, case):
    """gct2_ctx_set_relu_bits before an input-gradient call: the 16-byte epilogues (tap GEMM, halo kernel) take their ReLU mask from
    the plane - proven by handing them an all-positive act with the REAL mask in the plane - and give, with the real act and a
    consistent plane, the bits of the call without a plane; kernels that cannot read planes (8-byte epilogue) keep using act."""
    L = lib()
    dt = BF16
    kind = "convT" if case.startswith("convT") else "conv"
    B, H, W, Cin, Cout, tuning, narrow = {
        "conv_dgrad_halo": (2, 32, 32, 64, 32, 2 << 24, False), "conv_dgrad_tap": (2, 32, 32, 64, 32, 1 << 24, False),
        "convT_dgrad_tap": (2, 16, 16, 128, 32, 0, False), "convT_dgrad_big": (4, 32, 32, 256, 64, 5, False),
        "conv_dgrad_narrow": (2, 16, 16, 64, 32, 0, True),
    }[case]
    rng = np.random.default_rng(93)
    x = rnd(np.maximum(rng.standard_normal((B, H, W, Cin)), 0), dt)
    if kind == "conv":
        w = rnd(rng.standard_normal((4, 4, Cin, Cout)) * 0.1, dt)
        dz = rnd(rng.standard_normal((B, H // 2, W // 2, Cout)), dt)
        ref = O.conv4s2_bwd(x, w, dz)[0] * (x > 0)
    else:
        w = rnd(rng.standard_normal((4, 4, Cout, Cin)) * 0.1, dt)
        dz = rnd(rng.standard_normal((B, 2 * H, 2 * W, Cout)), dt)
        ref = O.convT4s2_bwd(x, w, dz)[0] * (x > 0)
    lda = Cin + (4 if narrow else 0)
    xd = torch.zeros(B, H, W, lda, dtype=TDT[dt], device=gpu)
    xd[..., :Cin] = dev(x, dt, gpu)
    ones = torch.ones_like(xd)
    wd, dzd = dev(w, dt, gpu), dev(dz, dt, gpu)
    ldb = Cin // 8 + 2
    bits = torch.zeros(B * H * W, ldb, dtype=torch.uint8, device=gpu)
    bits[:, 2:] = torch.tensor(_packbits(x, Cin), device=gpu)
    fn = "gct2_conv4s2_dgrad" if kind == "conv" else "gct2_convT4s2_dgrad"
    ws = torch.empty(4 << 18, dtype=torch.float32, device=gpu)
    use_ws = kind == "conv"          # the small conv-form problems would take split-K with a workspace: its finalize kernel reads act
    set_ws(ws if use_ws else None)
    set_tuning(tuning)
    res = {}
    same = (lambda a, b: torch.equal(a, b)) if use_ws else (lambda a, b: torch.allclose(a, b, rtol=1e-4, atol=1e-3))   # bias sums: atomics without a workspace
    try:
        for name, act, use_bits in (("act", xd, False), ("act+bits", xd, True), ("ones+bits", ones, True)):
            dx = torch.zeros(B, H, W, Cin, dtype=TDT[dt], device=gpu)
            db = torch.zeros(Cin, device=gpu)
            if use_bits:
                ctx_obj().set_relu_bits(bits.data_ptr() + 2, ldb)
            L.call(fn, ctx(), dt, dzd.data_ptr(), Cout, wd.data_ptr(), act.data_ptr(), lda, dx.data_ptr(), Cin, B, H, W, Cin, Cout, 0,
                   db.data_ptr(), Cin, None, 0, stream())
            torch.cuda.synchronize()
            res[name] = (dx, db)
        assert rel_l2(res["act"][0].double().cpu().numpy(), ref) <= TOL_OUT[dt]
        assert torch.equal(res["act"][0], res["act+bits"][0]) and same(res["act"][1], res["act+bits"][1])
        if narrow:      # 8-byte epilogue: the plane is ignored, the all-positive act masks nothing
            assert rel_l2(res["ones+bits"][0].double().cpu().numpy(), ref) > 0.3
        else:
            assert torch.equal(res["act"][0], res["ones+bits"][0]) and same(res["act"][1], res["ones+bits"][1])
    finally:
        set_tuning(0)
        set_ws(None)


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("use_ws", [False, True])
@pytest.mark.parametrize("shape", [(2, 32, 32, 64, 128), (2, 8, 8, 256, 512), (1, 16, 16, 3, 128)])
def test_wgrad_overwrite_mode(gpu, dt, shape, use_ws):
    """accumulate = 0: dw is overwritten (no pre-zeroing, no read) on every path - workspace slabs, one-owner tiles, atomics after
    an internal memset (no workspace / direct fp32 kernels / the 3-channel layer); accumulate = 1 adds to what is there."""
    B, H, W, Cin, Cout = shape
    L = lib()
    ws = torch.empty(16 << 18, dtype=torch.float32, device=gpu)
    set_ws(ws if use_ws else None)
    try:
        rng = np.random.default_rng(51)
        x = rnd(rng.standard_normal((B, H, W, Cin)), dt)
        dz = rnd(rng.standard_normal((B, H // 2, W // 2, Cout)), dt)
        ref = O.conv4s2_bwd(x, np.zeros((4, 4, Cin, Cout)), dz)[1]
        ldx = 4 if Cin == 3 else Cin
        xd = torch.zeros(B, H, W, ldx, dtype=TDT[dt], device=gpu)
        xd[..., :Cin] = dev(x, dt, gpu)
        dzd = dev(dz, dt, gpu)
        dw = torch.full((4, 4, Cin, Cout), 1e6, dtype=torch.float32, device=gpu)          # garbage that must disappear
        L.call("gct2_conv4s2_wgrad", ctx(), dt, xd.data_ptr(), ldx, dzd.data_ptr(), Cout, dw.data_ptr(), None, B, H, W, Cin, Cout, 0, None, stream())
        torch.cuda.synchronize()
        assert rel_l2(dw.cpu().numpy(), ref) <= TOL_F32OUT[dt]
        L.call("gct2_conv4s2_wgrad", ctx(), dt, xd.data_ptr(), ldx, dzd.data_ptr(), Cout, dw.data_ptr(), None, B, H, W, Cin, Cout, 1, None, stream())
        torch.cuda.synchronize()
        assert rel_l2(dw.cpu().numpy(), 2 * ref) <= TOL_F32OUT[dt]
        if Cin != 3:
            dzt = rnd(rng.standard_normal((B, 2 * H, 2 * W, Cout)), dt)
            reft = O.convT4s2_bwd(x, np.zeros((4, 4, Cout, Cin)), dzt)[1]
            dztd = dev(dzt, dt, gpu)
            dwt = torch.full((4, 4, Cout, Cin), -3e5, dtype=torch.float32, device=gpu)
            L.call("gct2_convT4s2_wgrad", ctx(), dt, xd.data_ptr(), ldx, dztd.data_ptr(), Cout, dwt.data_ptr(), None, B, H, W, Cin, Cout, 0, None, stream())
            torch.cuda.synchronize()
            assert rel_l2(dwt.cpu().numpy(), reft) <= TOL_F32OUT[dt]
    finally:
        set_ws(None)


