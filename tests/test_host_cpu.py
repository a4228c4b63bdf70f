"""CPU tier: the C-ABI library loads and exports every symbol include/gct2.h declares (no compute calls), argument
validation returns errors without a GPU, and the host-side logic (topology, arena layout, schedules, config API)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "gct2.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gct2_[A-Za-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    import gan_class_transfer2_amd as g
    lib = ctypes.CDLL(g._lib.LIB_PATH)
    names = _header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gct2.h but not exported"
    # and the ctypes table binds exactly the declared entry points (minus last_error, bound separately)
    assert sorted(set(g._lib.SIGNATURES) | {"gct2_last_error"}) == names
    assert g._lib.load().gct2_abi_version() == g._lib.ABI_VERSION == 17
    # the shipped library is the PRODUCT build: no in-kernel stamps, and the diagnostic hook refuses
    assert g._lib.build_flags() == 0
    c = g._lib.Context()
    with pytest.raises(g.Gct2Error, match="product build"):
        g._lib.call("gct2_ctx_set_stamp_buffer", c.handle, 4096, 1 << 20)


def test_binding_refuses_a_diagnostic_library(monkeypatch, tmp_path):
    """a library built with -DGCT2_STAMP reports it through gct2_build_flags(); _lib.load() (hence the engine, bench.py and every
    test) refuses it unless GCT2_ALLOW_DIAGNOSTIC_BUILD=1 (scripts/stamp_*.py set it).  Checked with a stand-in library that only
    exports the two identification calls - the binding must refuse before touching anything else."""
    import subprocess
    import gan_class_transfer2_amd as g
    src = tmp_path / "fake.c"
    src.write_text("int gct2_abi_version(void){return %d;}\nint gct2_build_flags(void){return 1;}\n"
                   "const char* gct2_last_error(void){return \"\";}\n" % g._lib.ABI_VERSION)
    so = tmp_path / "libgct2.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)], check=True)
    monkeypatch.setattr(g._lib, "_lib", None)
    monkeypatch.setattr(g._lib, "LIB_PATH", str(so))
    monkeypatch.setattr(g._lib, "SIGNATURES", {"gct2_abi_version": [], "gct2_build_flags": []})
    monkeypatch.delenv("GCT2_ALLOW_DIAGNOSTIC_BUILD", raising=False)
    with pytest.raises(g.Gct2Error, match="DIAGNOSTIC"):
        g._lib.load()
    monkeypatch.setenv("GCT2_ALLOW_DIAGNOSTIC_BUILD", "1")
    assert g._lib.load().gct2_build_flags() == 1
    monkeypatch.setattr(g._lib, "_lib", None)


def test_argument_validation_needs_no_gpu():
    import gan_class_transfer2_amd as g
    L = g._lib.load()
    # odd height: the reference's concat would fail (train.py:114-119); nothing is launched
    rc = L.gct2_conv4s2_fwd(None, 0, 16, 8, 16, None, 16, 8, 1, 5, 4, 8, 8, 1, None)
    assert rc == 1 and b"even" in L.gct2_last_error()
    assert L.gct2_conv4s2_fwd(None, 7, 16, 8, 16, None, 16, 8, 1, 4, 4, 8, 8, 1, None) == 1      # bad dtype
    assert L.gct2_conv4s2_fwd(None, 0, None, 8, 16, None, 16, 8, 1, 4, 4, 8, 8, 1, None) == 1    # null pointer
    assert L.gct2_convT4s2_fwd(None, 0, 16, 4, 16, None, 16, 8, 1, 4, 4, 8, 8, 1, None) == 1     # ld < channels
    assert L.gct2_dense_fwd(0, 16, 67, 16, None, 16, 10, 67, 5, None) == 1                 # Cout > 4
    with pytest.raises(g.Gct2Error):
        g._lib.call("gct2_adam_keras_multi", 4, 16, 16, 16, None, 0, 8, 1e-3, 0.9, 0.999, 1e-7, 1.0, None, 0, None)  # misaligned


def test_call_context_is_host_only_and_independent():
    """gct2_ctx (ABI v11): created, configured and destroyed without a GPU; the library keeps no process-wide scratch, so two
    contexts are independent objects and a bad scratch pointer is rejected per context."""
    import gan_class_transfer2_amd as g
    L = g._lib.load()
    a, b = g._lib.Context(), g._lib.Context()
    assert a.handle and b.handle and a.handle != b.handle
    assert L.gct2_ctx_set_workspace(a.handle, 24, 1 << 20) == 1 and b"16-byte" in L.gct2_last_error()      # misaligned
    assert L.gct2_ctx_set_workspace(a.handle, 4096, 1 << 20) == 0 and L.gct2_ctx_set_workspace(a.handle, None, 0) == 0
    assert L.gct2_ctx_set_tuning(b.handle, 2 | (3 << 16) | (1 << 24)) == 0 and L.gct2_ctx_force_direct(b.handle, 1) == 0
    assert L.gct2_ctx_set_tuning(b.handle, 7) == 1 and b"unknown tuning" in L.gct2_last_error()           # a tile that was pruned in r04
    assert L.gct2_ctx_set_tuning(b.handle, 0x200) == 1                                                     # ... and a removed switch
    assert L.gct2_ctx_set_workspace(None, None, 0) == 1                                                    # null ctx
    for name in ("gct2_set_workspace", "gct2_set_wgrad_workspace", "gct2_debug_tapgemm_variant", "gct2_debug_force_direct"):
        assert not hasattr(L, name), name                           # the process-wide hooks of ABI v10 are gone
    del a, b


def test_product_path_fails_loudly_without_device_or_library(monkeypatch):
    import gan_class_transfer2_amd as g
    if not torch.cuda.is_available():
        with pytest.raises(Exception):
            g.UNetEngine(g.Topology(8, 16, 2), g.F32, torch.device("cpu"))
    monkeypatch.setattr(g._lib, "_lib", None)
    monkeypatch.setattr(g._lib, "LIB_PATH", "/nonexistent/libgct2.so")
    with pytest.raises(g.Gct2Error):
        g._lib.load()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gan-class-transfer2_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f


def test_topology_matches_reference_channel_rule():
    import gan_class_transfer2_amd as g
    t = g.Topology(128, 512, 6)
    s = t.param_shapes()
    assert s["D0.w"] == (4, 4, 3, 128) and s["D5.w"] == (4, 4, 512, 512)
    assert s["U0.w"] == (4, 4, 64, 256) and s["U4.w"] == (4, 4, 512, 1024) and s["U5.w"] == (4, 4, 512, 512)
    assert s["dense.w"] == (67, 3)
    assert sum(int(np.prod(v)) for v in s.values()) == 41_691_660
    assert t.layer_order() == ["dense", "U0", "U1", "U2", "U3", "U4", "U5", "D5", "D4", "D3", "D2", "D1", "D0"]


def test_arena_layout_is_aligned_and_bucketable():
    from gan_class_transfer2_amd.engine import ParamArena
    import gan_class_transfer2_amd as g
    A = ParamArena(g.Topology(128, 512, 6), g.BF16, torch.device("cpu"))
    prev_hi = 0
    assert A.ready_order() == ["U0", "U1", "U2", "U3", "U4", "U5", "D5", "D4", "D3", "D2", "D1", "D0", "fp32"]
    for layer in A.ready_order():
        lo, hi = A.layer_ranges[layer]
        assert lo == prev_hi and lo % 64 == 0          # contiguous, in backward order, 16-byte aligned in bf16
        if layer != "fp32":                            # a layer's range is its kernel: what its fused optimizer launch covers
            assert lo == A.offsets[layer + ".w"] and hi - lo >= A.numel(layer + ".w") > hi - lo - 64
        prev_hi = hi
    assert prev_hi == A.total and A.total >= 41_691_660
    # every parameter the kernels read in fp32 from the master arena sits in ONE zone behind the kernels (r04): the replicated last
    # bucket of the sharded data-parallel step, one small optimizer launch on a single GPU
    zlo, zhi = A.layer_ranges["fp32"]
    fp32_read = [k for k in A.shapes if k.endswith(".b") or k.startswith("dense.")]
    assert len(fp32_read) == 14 and all(zlo <= A.offsets[k] and A.offsets[k] + A.numel(k) <= zhi for k in fp32_read)
    assert all(A.offsets[k] + A.numel(k) <= zlo for k in A.shapes if k not in fp32_read)
    assert A.layer_ranges["dense"][0] == zlo and zhi - zlo <= 64 * 64 + 8192
    A.glorot_init(1)
    w = A.param("U3.w")
    lim = np.sqrt(6.0 / (16 * 512 + 16 * 1024))
    assert float(w.abs().max()) <= lim and float(w.abs().max()) > 0.99 * lim and float(A.param("U3.b").abs().max()) == 0
    A.grad("D1.w").fill_(1.0)
    lo, hi = A.layer_ranges["D1"]
    assert float(A.g[lo:hi].sum()) == 4 * 4 * 128 * 256 and float(A.g.sum()) == 4 * 4 * 128 * 256


def test_reference_config_surface_and_schedules():
    from gan_class_transfer2_amd import model as M
    # defaults of train.py:17-36
    assert (M.size, M.pixel_size, M.max_size, M.block_depth, M.octaves, M.batch_size, M.steps) == (256, 128, 512, 0, 6, 1, 200)
    assert (M.residual, M.concat, M.predict_x, M.mixed_precision, M.warm_up) == (False, True, True, False, 2000)
    w = M.WarmUp(2e-5, 2000)
    assert abs(w(0) - 2e-5 / 2001) < 1e-12 and abs(w(1999) - 2e-5 * 2000 / 2001) < 1e-11 and abs(w(2000) - 2e-5) < 1e-12
    assert abs(float(M.alpha_dash(torch.tensor(25.0))) - 0.25 * (1 - 25 / 201) ** 2) < 1e-7
    assert float(M.identity(None, torch.tensor([1.0, 3.0]))) == 2.0
    assert M.preferred_dtype_code() == 0
    M.configure(mixed_precision=True)
    try:
        assert M.preferred_dtype_code() == 2 and isinstance(M.default_optimizer(), M.LossScaleOptimizer)
    finally:
        M.configure(mixed_precision=False)
    with pytest.raises(AttributeError):
        M.configure(no_such_knob=1)
    M.configure(block_depth=2, residual=True)           # the off-by-default switches build their layers (train.py:104-143)
    try:
        blk = M.Block(8)
        assert len(blk.convs) == 2 and all(isinstance(c, M.Conv3x3) and c.filters == 8 for c in blk.convs)
        assert M.Denoiser.variant(None)
    finally:
        M.configure(block_depth=0, residual=False)
    assert not M.Denoiser.variant(None) and M.Block(8).convs == []


def test_bench_flop_model_matches_survey_appendix_b():
    import bench
    import gan_class_transfer2_amd as g
    t = g.Topology(128, 512, 6)
    assert abs(bench.f_train_per_image(t, 128, 128) / 1e9 - 32.1314) < 1e-3
    assert abs(bench.f_train_per_image(t, 64, 64) / 1e9 - 8.0328) < 1e-3
    assert abs(bench.f_train_per_image(g.Topology(128, 512, 5), 32, 32) / 1e9 - 1.9705) < 1e-3


def test_lds_swizzles_are_conflict_free_in_the_bank_model():
    """scripts/lds_swizzle_search.py: the three LDS image layouts of csrc/ against the ds_read_b128 lane-group model of
    MI355X_MICROARCH.md (the asserts inside main() are the check)."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "lds_swizzle_search.py")
    spec = importlib.util.spec_from_file_location("lds_swizzle_search", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main()


def test_one_shot_relu_plane_never_outlives_its_call():
    """gct2_ctx_set_relu_bits registers a plane for the NEXT layer call only.  Every layer entry point takes it out of the ctx first
    thing, so a call rejected by its argument checks cannot leak it to a later layer (r03 consumed it behind the checks), and an
    entry point that can neither write nor read a plane says so.  Host-side only: nothing is launched (every call is rejected)."""
    import gan_class_transfer2_amd as g
    L = g._lib.load()
    c = g._lib.Context()
    plane = 4096                                                    # never dereferenced: every call below fails its checks
    # (1) a rejected forward call consumes the plane: the weight-gradient call behind it fails for ITS reason, not for a pending plane
    c.set_relu_bits(plane, 16)
    assert L.gct2_conv4s2_fwd(c.handle, 1, 16, 8, 16, None, 16, 8, 1, 5, 4, 8, 8, 1, None) == 1 and b"even" in L.gct2_last_error()
    assert L.gct2_conv4s2_wgrad(c.handle, 1, 16, 8, 16, 8, 16, None, 1, 5, 4, 8, 8, 0, None, None) == 1
    assert b"even" in L.gct2_last_error() and b"plane" not in L.gct2_last_error()
    # (2) the same through a rejected input-gradient call
    c.set_relu_bits(plane, 16)
    assert L.gct2_convT4s2_dgrad(c.handle, 1, 16, 8, 16, 16, 4, 16, 8, 1, 4, 4, 8, 8, 0, None, 0, None, 0, None) == 1      # ldact < Cin
    assert L.gct2_convT4s2_wgrad(c.handle, 1, 16, 4, 16, 8, 16, None, 1, 4, 4, 8, 8, 0, None, None) == 1
    assert b"plane" not in L.gct2_last_error()
    # (3) entry points that cannot use a plane reject a pending one - and clear it
    for call in (lambda: L.gct2_conv4s2_wgrad(c.handle, 1, 16, 8, 16, 8, 16, None, 1, 4, 4, 8, 8, 0, None, None),
                 lambda: L.gct2_convT4s2_wgrad(c.handle, 1, 16, 8, 16, 8, 16, None, 1, 4, 4, 8, 8, 0, None, None),
                 lambda: L.gct2_conv2d_s1_fwd(c.handle, 1, 16, 8, 16, None, 16, 8, 1, 4, 4, 8, 8, 3, 1, None)):
        c.set_relu_bits(plane, 16)
        assert call() == 1 and b"ReLU bit plane was registered" in L.gct2_last_error()
    assert L.gct2_conv4s2_wgrad(c.handle, 1, 16, 8, 16, 8, 16, None, 1, 5, 4, 8, 8, 0, None, None) == 1 and b"even" in L.gct2_last_error()
    # (4) a plane that does not fit the call's channels is an error of that call, and gone afterwards
    c.set_relu_bits(plane, 1)
    assert L.gct2_conv4s2_fwd(c.handle, 1, 16, 8, 16, None, 16, 16, 1, 4, 4, 8, 16, 1, None) == 1 and b"ld_bytes" in L.gct2_last_error()
    assert L.gct2_conv4s2_wgrad(c.handle, 1, 16, 8, 16, 8, 16, None, 1, 5, 4, 8, 8, 0, None, None) == 1 and b"even" in L.gct2_last_error()


def test_launch_log_is_per_context_and_host_only():
    import gan_class_transfer2_amd as g
    c = g._lib.Context()
    assert c.read_launch_log() == []
    c.log_launches(True)
    assert c.read_launch_log() == []                                 # nothing launched yet; reading clears
    c.log_launches(False)


def _run_bench(extra_env, *argv):
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, env=env, timeout=300)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` with no launcher in front of it (the form the driver uses): the parent starts N ranks through
    torch.distributed.run before touching a GPU, relays rank 0's ONE JSON line and the exit code.  The rank body
    here is the launch-contract self-test (GCT2_BENCH_LAUNCH_TEST=1: gloo rendezvous, barrier, MAX over ranks; no GPU work)."""
    import json
    r = _run_bench({"GCT2_BENCH_LAUNCH_TEST": "1"}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["max_over_ranks"] == 2.0
    # a failing rank makes the launcher exit non-zero and print no result line
    r = _run_bench({"GCT2_BENCH_LAUNCH_TEST": "1", "GCT2_BENCH_LAUNCH_TEST_FAIL_RANK": "1"}, "--gpus", "2")
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_kernel_symbol_and_labels():
    import bench
    assert bench.kernel_symbol("wgrad:256q:rsplit=64:slabs") == "wgrad256q_kernel"
    assert bench.kernel_symbol("wgrad:128:rsplit=1:owner") == "wgrad_kernel"
    # one symbol per profiler row: the epilogue is part of it (r05 merged DownShuffle_1's forward with the UpShuffle input gradients)
    assert bench.kernel_symbol("tap:conv:256x128:mask:ksplit=1:bits") == "tapgemm_kernel<conv,256x128,mask>"
    assert bench.kernel_symbol("tap:conv:256x128:bias_act:ksplit=1:bits") == "tapgemm_kernel<conv,256x128,bias_act>"
    assert bench.kernel_symbol("tap:convT:128x128:mask:ksplit=2:wstat") == "tapgemm_kernel<convT,128x128,mask>"
    assert bench.kernel_symbol("halo:convT:head") == "halo_convT_kernel<head>"
    assert bench.kernel_symbol("halo:convT:mask:bits") == "halo_convT_kernel<mask>" != bench.kernel_symbol("halo:convT:bias_act:bits")
    # ... and the mangled-name pattern that joins a symbol with a rocprofv3 --mangled-kernels row (scripts/compare_bench_rocprof.py)
    assert bench.rocprof_pattern("tapgemm_kernel<conv,256x128,mask>") == "tapgemm_kernelI*Li0ELi256ELi128ELi1E"
    assert bench.rocprof_pattern("halo_convT_kernel<head>") == "halo_convT_kernelI*Li2E"
    assert bench.rocprof_pattern("wgrad256q_kernel") == "wgrad256q_kernelI"
    # per call the MEDIAN over the measured steps, per symbol the sum of its calls' medians: one stalled sample changes nothing
    names = [("wgrad", "wgrad256q_kernel", 1e11, ("U0", "wgrad")), ("wgrad", "wgrad256q_kernel", 1e11, ("U1", "wgrad")),
             ("conv_form", "tapgemm_kernel<conv,256x128,mask>", 1e11, ("U0", "dgrad"))]
    samples = [[0.10, 0.11, 0.10, 0.10, 0.12], [0.08, 0.08, 0.08, 0.09, 0.08], [0.14, 0.14, 33.0, 0.14, 0.15]]
    rows, syms = bench.symbol_table(names, samples, 1)
    assert abs(rows["wgrad256q_kernel"]["sum_median_ms"] - 0.18) < 1e-9 and syms["wgrad256q_kernel"]["launches_per_step"] == 2
    t = syms["tapgemm_kernel<conv,256x128,mask>"]
    assert t["avg_launch_us"] == 140.0 and t["max_us"] == 33000.0 and t["min_us"] == 140.0 and t["layers"] == ["U0.dgrad"]
    assert max(rows, key=lambda k: rows[k]["sum_median_ms"]) == "wgrad256q_kernel"
    bench.call_label.size = 128
    # gct2_convT4s2_wgrad(ctx, dtype, x, ldx, dz, lddz, dw, db, B, H, W, Cin, Cout, acc, adam, stream) for UpShuffle_0 at config 3
    a = (0, 1, 0, 256, 0, 64, 0, None, 64, 64, 64, 256, 64, 0, None, None)
    assert bench.call_label("gct2_convT4s2_wgrad", a) == ("U0", "wgrad")
    assert bench.call_flops("gct2_convT4s2_wgrad", a) == 2.0 * 64 * 128 * 128 * 64 * 4 * 256
    a = (0, 1, 0, 512, 0, 0, 0, 512, 64, 16, 16, 512, 512, 1, None)       # gct2_conv4s2_fwd of DownShuffle_3
    assert bench.call_label("gct2_conv4s2_fwd", a) == ("D3", "fwd")
    assert abs(bench.call_flops("gct2_conv4s2_fwd", a) / 1e9 - 34.360) < 1e-2


def test_step_plan_records_replays_and_reports_the_failing_record():
    """gct2_plan on the host alone (no GPU: every recorded call is rejected by its argument checks, which run before any launch):
    records are appended in order, slots are re-set by key, a run stops at the first failing record and says which one, segments
    end at the cuts, entry points a plan cannot hold are executed right away instead of being recorded."""
    import gan_class_transfer2_amd as g
    L = g._lib
    c = L.Context()
    p = L.Plan()
    p.begin(execute=False)
    c.set_tuning(2)                                                   # not plannable: executed immediately, not recorded
    assert c.tuning == 2 and p.n == 0
    L.call("gct2_rng_uniform_int", 1, 1, L.Slot("off", 0), None, 4, 1, 200, None)          # record 0: null output -> EINVAL at run time
    p.cut("after-rng")
    L.call("gct2_conv4s2_fwd", c.handle, 1, 16, 8, 16, None, 16, 8, 1, L.Slot("H", 5), 4, 8, 8, 1, None)   # record 1: odd H
    p.end()
    assert p.n == 2 and p.segments() == 2 and L._recording is None
    with pytest.raises(g.Gct2Error, match=r"record 0.*rng_uniform_int"):
        p.run_segment(0)
    with pytest.raises(g.Gct2Error, match=r"record 1.*must be even"):
        p.run_segment(1)
    p.set("H", 4)                                                     # the slot is re-set in place: the same record now fails later,
    with pytest.raises(g.Gct2Error, match=r"record 1"):               # for another reason (still before any launch)
        p.run_segment(1)
    assert "even" not in L.load().gct2_last_error().decode()
    # direct C calls: unknown entry point, wrong arity, bad indices
    lib = L.load()
    arr = (ctypes.c_uint64 * 2)(0, 0)
    assert lib.gct2_plan_add_call(p.handle, b"gct2_ctx_create", arr, 1, None) == 1 and b"not an entry point" in lib.gct2_last_error()
    assert lib.gct2_plan_add_call(p.handle, b"gct2_add", arr, 2, None) == 1 and b"takes 8 arguments" in lib.gct2_last_error()
    assert lib.gct2_plan_set_arg(p.handle, 7, 0, 0) == 1 and lib.gct2_plan_run(p.handle, 1, 5, None) == 1
    assert lib.gct2_plan_add_wait(p.handle, None, 3) == 1
