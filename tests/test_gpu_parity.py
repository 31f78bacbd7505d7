"""GPU parity tests: HIP kernels (through the C-ABI in include/mg_hip.h) against the oracle
and the golden vectors made by the reference's own code.  Run with -m gpu on an MI355X."""
import ctypes as C

import numpy as np
import pytest

from morphablegraphs_amd import _capi, synthetic
from oracle import c_oracle
from oracle import mg_oracle as orc

pytestmark = pytest.mark.gpu

# north-star tolerance on float32 pose values against the reference's float64 path:
# 1e-5 absolute, widened only where 1e-5 is below float32 resolution of the value itself
# (|v| >= 128: half an f32 ulp = 2^-24 |v|).
def pose_tol(ref):
    return 1e-5 + 2.0 ** -24 * np.abs(ref)


@pytest.fixture(scope="module")
def ctx():
    c = _capi.Context(0)
    yield c
    c.close()


@pytest.fixture(autouse=True)
def _default_options(request):
    """Tests that switch a library option (mg_context_set_option) leave the shared context at its defaults."""
    yield
    if "ctx" in request.fixturenames:
        c = request.getfixturevalue("ctx")
        if c.handle:
            for option in range(_capi.MG_OPT_COUNT):
                c.set_option(option, 0)


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_device_is_gfx950(ctx):
    info = ctx.device_info()
    assert "gfx950" in info["name"]
    assert info["n_cu"] >= 200


def test_frames_golden_all_paths(ctx, golden_case):
    name, data, g = golden_case
    prim = _capi.Primitive(ctx, data)
    cp = c_oracle.COraclePrimitive(data)
    S = g["S"]
    ref = g["frames"]
    model = cp.frames_f32model(S)
    paths = [_capi.MG_PATH_DIRECT] + ([_capi.MG_PATH_MFMA] if prim.mfma_supported else [])
    for path in paths:
        got = prim.back_project_frames(S, path=path)
        assert got.shape == ref.shape and got.dtype == np.float32
        err = np.abs(got.astype(np.float64) - ref)
        assert np.all(err <= pose_tol(ref)), (name, path, float((err / pose_tol(ref)).max()))
        # bit-exact against the CPU model of the f32 contract: pins frame/joint indexing exactly
        np.testing.assert_array_equal(_bits(got), _bits(model), err_msg="%s path %d" % (name, path))
    # float32 latents: same contract, the float32 values are the inputs
    S32 = S.astype(np.float32)
    model32 = cp.frames_f32model(S32.astype(np.float64))
    for path in paths:
        np.testing.assert_array_equal(_bits(prim.back_project_frames(S32, path=path)), _bits(model32))
    prim.close()


def test_frames_f64_and_coeffs_golden(ctx, golden_case):
    name, data, g = golden_case
    prim = _capi.Primitive(ctx, data)
    scale = max(1.0, np.abs(g["frames"]).max())
    got = prim.back_project_frames_f64(g["S"])
    np.testing.assert_allclose(got, g["frames"], rtol=0, atol=4e-12 * scale)
    coeffs = prim.back_project_coeffs(g["S"])
    assert coeffs.shape == g["coeffs"].shape
    np.testing.assert_allclose(coeffs, g["coeffs"], rtol=0, atol=4e-12 * scale)
    c32 = prim.back_project_coeffs(g["S"], dtype=np.float32)
    assert np.all(np.abs(c32 - g["coeffs"]) <= pose_tol(g["coeffs"]))
    # a batch large enough for the persistent kernel (identity "time grid" over the control points, carried-over
    # tiles included): same bits as the small-batch direct kernel row by row, same tolerance against float64
    S64 = np.tile(g["S"], (16, 1))
    big = prim.back_project_coeffs(S64, dtype=np.float32)
    np.testing.assert_array_equal(_bits(big), _bits(np.tile(c32, (16, 1, 1))), err_msg=name)
    ref64 = prim.back_project_coeffs(S64)
    assert np.all(np.abs(big - ref64) <= pose_tol(ref64))
    prim.close()


def test_evaluate_arbitrary_times_golden(ctx, golden_case):
    name, data, g = golden_case
    prim = _capi.Primitive(ctx, data)
    cp = c_oracle.COraclePrimitive(data)
    times = g["eval_times"]
    grid = prim.time_grid(times)
    i0, w, t = grid.tables()
    i0_o, w_o = orc.basis_rows(cp.knots, times)
    np.testing.assert_array_equal(i0, i0_o)
    np.testing.assert_array_equal(w, w_o)
    scale = max(1.0, np.abs(g["evals"]).max())
    np.testing.assert_allclose(prim.back_project_frames_f64(g["S"], grid), g["evals"], rtol=0, atol=4e-12 * scale)
    got = prim.back_project_frames(g["S"], grid)
    assert np.all(np.abs(got - g["evals"]) <= pose_tol(g["evals"]))
    np.testing.assert_array_equal(_bits(got), _bits(cp.frames_f32model(g["S"], times)))
    # spline evaluation from explicit (mutable) coefficient arrays
    ev = prim.spline_evaluate(g["coeffs"], grid)
    np.testing.assert_allclose(ev, g["evals"], rtol=0, atol=4e-12 * scale)
    fr = prim.spline_evaluate(g["coeffs"][0])
    np.testing.assert_allclose(fr[0], g["frames"][0], rtol=0, atol=4e-12 * scale)
    grid.close()
    prim.close()


def test_gmm_log_prob_golden(ctx, golden_case):
    name, data, g = golden_case
    prim = _capi.Primitive(ctx, data)
    pc = g["precisions_cholesky"]
    np.testing.assert_allclose(prim.precisions_cholesky(), pc, rtol=1e-9, atol=1e-9 * np.abs(pc).max())
    lp = prim.gmm_log_prob(g["X"])
    np.testing.assert_allclose(lp, g["logp"], rtol=1e-9, atol=1e-7)
    lp32 = prim.gmm_log_prob(g["X"].astype(np.float32), dtype=np.float32)
    cp = c_oracle.COraclePrimitive(data)
    ref32 = cp.log_prob_f64(g["X"].astype(np.float32).astype(np.float64))
    np.testing.assert_allclose(lp32, ref32, rtol=3e-7, atol=1e-6)
    prim.close()


@pytest.mark.parametrize("B", [1, 7, 16, 17, 255, 1000])
def test_frames_ragged_batches_bit_exact(ctx, B):
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    cp = c_oracle.COraclePrimitive(data)
    rng = np.random.default_rng(B)
    S = rng.standard_normal((B, 40)).astype(np.float32)
    model = cp.frames_f32model(S.astype(np.float64))
    for path in (_capi.MG_PATH_MFMA, _capi.MG_PATH_DIRECT, _capi.MG_PATH_AUTO):
        got = prim.back_project_frames(S, path=path)
        np.testing.assert_array_equal(_bits(got), _bits(model), err_msg="B=%d path=%d" % (B, path))
    prim.close()


def test_frames_repeated_launches_are_bitwise_stable(ctx):
    """The persistent kernel hands LDS buffers between producer and consumer waves: repeated launches over ragged
    and full batches must reproduce the oracle bit for bit every time (guards the barrier protocol)."""
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    cp = c_oracle.COraclePrimitive(data)
    rng = np.random.default_rng(77)
    for B in (1000, 4099):
        S = rng.standard_normal((B, 40)).astype(np.float32)
        model = cp.frames_f32model(S[:1000].astype(np.float64))
        for rep in range(4):
            got = prim.back_project_frames(S, path=_capi.MG_PATH_MFMA)
            np.testing.assert_array_equal(_bits(got[:1000]), _bits(model), err_msg="B=%d rep=%d" % (B, rep))
            if rep == 0:
                first = got
            else:
                np.testing.assert_array_equal(_bits(got), _bits(first), err_msg="B=%d rep=%d" % (B, rep))
    prim.close()


def _fused_step(ctx, prim, S, F, D):
    B, L = S.shape
    d_S = ctx.upload(S)
    d_frames = ctx.malloc(max(B * F * D * 4, 4))
    d_logp = ctx.malloc(max(B * 4, 4))
    prim.step_frames_and_logp_dev(d_S, S.dtype, B, L, d_frames, d_logp)
    ctx.synchronize()
    frames = ctx.download(d_frames, (B, F, D), np.float32)
    logp = ctx.download(d_logp, (B,), np.float32)
    for buf in (d_S, d_frames, d_logp):
        buf.free()
    return frames, logp


@pytest.mark.parametrize("B", [8, 16, 17, 255, 1000, 4099])
def test_fused_step_matches_separate_calls_bit_for_bit(ctx, B):
    """mg_step_frames_and_logp scores the mixture inside the frames kernel (the sweep waves do it while the
    pipeline fills): frames must equal the oracle's f32 model and log p must equal the stand-alone log-likelihood
    kernel bit for bit, for float32 and float64 latents, ragged and multi-tile batches, twice in a row."""
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    cp = c_oracle.COraclePrimitive(data)
    rng = np.random.default_rng(1000 + B)
    for dtype in (np.float32, np.float64):
        S = rng.standard_normal((B, 40)).astype(dtype)
        model = cp.frames_f32model(S.astype(np.float64))
        lp_sep = prim.gmm_log_prob(S, dtype=np.float32)
        ref = cp.log_prob_f64(S.astype(np.float64))
        for rep in range(2):
            frames, logp = _fused_step(ctx, prim, S, 156, 79)
            np.testing.assert_array_equal(_bits(frames), _bits(model), err_msg="B=%d %s rep=%d" % (B, dtype, rep))
            np.testing.assert_array_equal(_bits(logp), _bits(lp_sep), err_msg="B=%d %s rep=%d" % (B, dtype, rep))
            np.testing.assert_allclose(logp, ref, rtol=3e-7, atol=1e-6)
    prim.close()


def test_fused_step_golden_shapes(ctx, golden_case):
    """Every golden primitive through the step entry point (fused where the shape allows it, two launches
    otherwise): frames against the reference within the north-star tolerance, log p against the reference."""
    name, data, g = golden_case
    prim = _capi.Primitive(ctx, data)
    F, D = g["frames"].shape[1], g["frames"].shape[2]
    for reps in (1, 5):   # 4 samples: two launches (direct path); 20 samples: one fused launch where supported
        S = np.ascontiguousarray(np.tile(g["S"], (reps, 1)))
        ref = np.tile(g["frames"], (reps, 1, 1))
        ref_lp = np.tile(g["logp_S"], reps)
        frames, logp = _fused_step(ctx, prim, S, F, D)
        err = np.abs(frames.astype(np.float64) - ref)
        assert np.all(err <= pose_tol(ref)), (name, reps, float((err / pose_tol(ref)).max()))
        np.testing.assert_allclose(logp, ref_lp, rtol=3e-7, atol=1e-5 * max(1.0, float(np.abs(ref_lp).max())))
    prim.close()


def test_short_last_window_ignores_stale_lds(ctx):
    """The last chunk of the walk grid has a 6-row window; the root-tap MFMA must not pick up stale LDS (a previous
    launch's bytes) from rows 6-7 of the float64 root image -- 0 * NaN would poison the root channels.  A launch with
    NaN latents leaves NaN bit patterns in every CU's LDS; small launches whose workgroups START on the short chunk
    must still be exact."""
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    cp = c_oracle.COraclePrimitive(data)
    rng = np.random.default_rng(5)
    poison = np.full((4096, 40), np.nan, dtype=np.float32)
    for B in (16, 40, 16):
        S = rng.standard_normal((B, 40)).astype(np.float32)
        model = cp.frames_f32model(S.astype(np.float64))
        assert np.isnan(prim.back_project_frames(poison, path=_capi.MG_PATH_MFMA)).all()
        got = prim.back_project_frames(S, path=_capi.MG_PATH_MFMA)
        assert np.isfinite(got).all(), "stale LDS leaked into the output"
        np.testing.assert_array_equal(_bits(got), _bits(model))
    prim.close()


def test_frames_leading_dimension_and_extra_columns(ctx):
    """back_project uses s[:n_components] (reference motion_primitive.py:229); extra time
    columns in the latent rows are ignored."""
    data = synthetic.make_primitive(seed=3, n_components=10, n_frames=50, n_dim=23, n_gmm=2)
    prim = _capi.Primitive(ctx, data)
    cp = c_oracle.COraclePrimitive(data)
    rng = np.random.default_rng(0)
    S = rng.standard_normal((33, 14))
    got = prim.back_project_frames(S, path=_capi.MG_PATH_MFMA)
    np.testing.assert_array_equal(_bits(got), _bits(cp.frames_f32model(np.ascontiguousarray(S[:, :10]))))
    prim.close()


def test_empty_batch_and_bad_arguments(ctx):
    data = synthetic.make_tiny_primitive()
    prim = _capi.Primitive(ctx, data)
    assert prim.back_project_frames(np.zeros((0, 3), dtype=np.float32)).shape == (0, 12, 7)
    assert prim.gmm_log_prob(np.zeros((0, 3))).shape == (0,)
    with pytest.raises(_capi.MGError):
        prim.back_project_frames(np.zeros((4, 2), dtype=np.float32))      # ld < n_components
    bad = dict(data)
    cov = np.array(data["gmm_covars"])
    cov[0] = -np.eye(3)
    bad["gmm_covars"] = cov.tolist()
    with pytest.raises(_capi.MGError) as ei:
        _capi.Primitive(ctx, bad)
    assert ei.value.status == -5
    prim.close()


def test_full_size_properties(ctx):
    """BASELINE.json config 2 size (B = 8192): size-independent properties instead of a full
    oracle pass -- linearity of the delta in the latent, row independence, and a seeded
    subset against the oracle."""
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    cp = c_oracle.COraclePrimitive(data)
    B = 8192
    rng = np.random.default_rng(1)
    S = rng.standard_normal((B, 40)).astype(np.float32)
    F = prim.back_project_frames(S, path=_capi.MG_PATH_MFMA)
    assert F.shape == (B, 156, 79) and np.isfinite(F).all()
    idx = rng.choice(B, size=48, replace=False)
    np.testing.assert_array_equal(_bits(F[idx]), _bits(cp.frames_f32model(S[idx].astype(np.float64))))
    # a candidate's frames do not depend on its batch position or neighbours
    perm = rng.permutation(B)
    Fp = prim.back_project_frames(S[perm], path=_capi.MG_PATH_MFMA)
    np.testing.assert_array_equal(_bits(Fp), _bits(F[perm]))
    # linearity: frames(s) - frames(0) is linear in s (float64 entry point)
    z = prim.back_project_frames_f64(np.zeros((1, 40)))[0]
    a, b = S[:4].astype(np.float64), S[4:8].astype(np.float64)
    fa, fb = prim.back_project_frames_f64(a) - z, prim.back_project_frames_f64(b) - z
    fab = prim.back_project_frames_f64(2.0 * a - 0.5 * b) - z
    np.testing.assert_allclose(fab, 2.0 * fa - 0.5 * fb, rtol=0, atol=1e-9)
    # log-likelihood at full size against the oracle
    lp = prim.gmm_log_prob(S, dtype=np.float64)
    np.testing.assert_allclose(lp, cp.log_prob_f64(S.astype(np.float64)), rtol=1e-10, atol=1e-8)
    # the bench's step (one fused launch, two 16-candidate tiles per workgroup): identical to the separate kernels
    Ff, lpf = _fused_step(ctx, prim, S, 156, 79)
    np.testing.assert_array_equal(_bits(Ff), _bits(F))
    np.testing.assert_array_equal(_bits(lpf), _bits(prim.gmm_log_prob(S, dtype=np.float32)))
    # more than two tiles per workgroup: the fused kernel scores a second group of tiles, same bits
    S2 = np.vstack([S, S[:48]])
    Ff2, lpf2 = _fused_step(ctx, prim, S2, 156, 79)
    np.testing.assert_array_equal(_bits(Ff2[:B]), _bits(F))
    np.testing.assert_array_equal(_bits(Ff2[B:]), _bits(F[:48]))
    np.testing.assert_array_equal(_bits(lpf2[:B]), _bits(lpf))
    prim.close()


def test_score_constraints_and_argmin(ctx):
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    cp = c_oracle.COraclePrimitive(data)
    rng = np.random.default_rng(2)
    S = rng.standard_normal((513, 40))
    cons = [
        {"type": "position", "t": 155.0, "weight": 1.0, "target": [30.0, None, -40.0]},
        {"type": "position", "t": 77.5, "weight": 0.5, "target": [5.0, 90.0, 3.0]},
        {"type": "direction", "t": 155.0, "weight": 2.0, "target": [0.3, -1.0], "ref_dir": (0.0, 0.0, 1.0)},
    ]
    nan = np.nan
    cons_c = np.array([[0, 155.0, 1.0, 30.0, nan, -40.0, 0, 0],
                       [0, 77.5, 0.5, 5.0, 90.0, 3.0, 0, 0],
                       [1, 155.0, 2.0, 0.3, -1.0, 0.0, 0.0, 1.0]])
    cset = _capi.ConstraintSet(prim, cons)
    err = prim.score_constraints(cset, S)
    ref = cp.keyframe_errors_f64(S, cons_c)
    np.testing.assert_allclose(err, ref, rtol=1e-10, atol=1e-9)
    # argmin: first strict minimum, NaN never wins
    for vals in (err, np.array([3.0, 1.0, 1.0, 2.0]), np.array([np.nan, 2.0, np.nan, 2.0]),
                 np.array([np.inf, np.inf]), np.array([np.nan]), np.full(5000, 7.0)):
        for dt in (np.float64, np.float32):
            v = np.ascontiguousarray(vals, dtype=dt)
            buf = ctx.upload(v)
            i, m = ctx.argmin_first(buf, len(v), dt)
            io, mo = c_oracle.first_min_argmin(v)
            assert (i, m) == (io, mo) or (np.isinf(m) and np.isinf(mo) and i == io)
            buf.free()
    assert int(np.argmin(ref)) == ctx.argmin_first(ctx.upload(err), len(err), np.float64)[0]
    cset.close()
    prim.close()


def test_device_sampler_distribution(ctx):
    """Philox sampler is validated distributionally (not bit-compatible with sklearn)."""
    data = synthetic.make_primitive(seed=2, n_components=6, n_frames=40, n_dim=11, n_gmm=3)
    prim = _capi.Primitive(ctx, data)
    counts = np.array([40000, 25000, 35000])
    X, comp = prim.gmm_sample(counts, seed=1234)
    assert X.shape == (100000, 6)
    np.testing.assert_array_equal(comp, np.repeat(np.arange(3), counts))      # grouped by component
    means, covars = np.array(data["gmm_means"]), np.array(data["gmm_covars"])
    for k in range(3):
        xs = X[comp == k]
        se = np.sqrt(np.diag(covars[k]) / len(xs))
        assert np.all(np.abs(xs.mean(axis=0) - means[k]) < 5 * se)
        emp = np.cov(xs.T)
        assert np.max(np.abs(emp - covars[k])) < 0.05 * np.abs(covars[k]).max() + 0.02
    X2, _ = prim.gmm_sample(counts, seed=1234)
    np.testing.assert_array_equal(X, X2)                                         # reproducible
    X3, _ = prim.gmm_sample(counts, seed=1235)
    assert np.abs(X - X3).max() > 0.1
    prim.close()


def test_graph_of_primitives_shapes(ctx):
    """BASELINE.json config 3 shapes: 16 primitives with varying (L, F, K)."""
    rng = np.random.default_rng(5)
    for data in synthetic.make_graph_primitives(16)[:6]:
        prim = _capi.Primitive(ctx, data)
        cp = c_oracle.COraclePrimitive(data)
        S = rng.standard_normal((40, prim.n_components)).astype(np.float32)
        got = prim.back_project_frames(S)
        np.testing.assert_array_equal(_bits(got), _bits(cp.frames_f32model(S.astype(np.float64))))
        np.testing.assert_allclose(prim.gmm_log_prob(S, dtype=np.float64), cp.log_prob_f64(S.astype(np.float64)),
                                   rtol=1e-10, atol=1e-8)
        from oracle import mg_oracle as orc
        jac = prim.gmm_log_prob_jac(S[:12].astype(np.float64))
        ref = orc.OraclePrimitive(data).log_likelihood_jac(S[:12].astype(np.float64))
        np.testing.assert_allclose(jac, ref, rtol=1e-8, atol=1e-9 * max(1.0, np.abs(ref).max()))
        prim.close()


def test_more_than_64_latent_components_take_the_fallback_kernels(ctx):
    """n_components > 64 is outside the MFMA kernels' k-step templates: frames go through the direct kernel,
    log-likelihood and its Jacobian through the VALU kernels, the fused step through two launches; same contracts."""
    from oracle import mg_oracle as orc
    data = synthetic.make_primitive(seed=21, n_components=70, n_frames=30, n_basis=8, n_dim=15, n_gmm=3, name="wide")
    prim = _capi.Primitive(ctx, data)
    assert not prim.mfma_supported
    cp = c_oracle.COraclePrimitive(data)
    rng = np.random.default_rng(2)
    S = (0.3 * rng.standard_normal((70, 70))).astype(np.float32)
    got = prim.back_project_frames(S)
    np.testing.assert_array_equal(_bits(got), _bits(cp.frames_f32model(S.astype(np.float64))))
    with pytest.raises(_capi.MGError):
        prim.back_project_frames(S, path=_capi.MG_PATH_MFMA)
    lp = prim.gmm_log_prob(S, dtype=np.float64)
    np.testing.assert_allclose(lp, cp.log_prob_f64(S.astype(np.float64)), rtol=1e-10, atol=1e-8)
    jac = prim.gmm_log_prob_jac(S[:20].astype(np.float64))
    ref = orc.OraclePrimitive(data).log_likelihood_jac(S[:20].astype(np.float64))
    np.testing.assert_allclose(jac, ref, rtol=1e-8, atol=1e-9 * max(1.0, np.abs(ref).max()))
    frames, logp = _fused_step(ctx, prim, S, 30, 15)
    np.testing.assert_array_equal(_bits(frames), _bits(got))
    np.testing.assert_allclose(logp, lp, rtol=3e-7, atol=1e-5)
    prim.close()


def test_joint_position_constraints_forward_kinematics(ctx):
    """MG_CONSTRAINT_JOINT_POSITION: GlobalTransformConstraint._point_distance on the FK position of any joint
    (reference two_hand_constraint.py:51-93 / pose_constraint.py:48-67 through anim_utils' get_global_position;
    PARITY UNPINNED, anim_utils absent).  Pinned by (1) known-answer poses built into a primitive whose mean
    curve is a chosen pose, (2) an independent rotation-matrix oracle on random candidates, (3) the residual
    matrix / error sum identity, (4) the root joint agreeing with the FK-free position constraint."""
    from oracle import mg_oracle as orc
    joints, animated = synthetic.make_skeleton()
    sk = _capi.Skeleton(joints, animated)
    assert 3 + 4 * len(animated) == 79 and sk.chain("LeftHand_EndSite")[0] == 0
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    op = orc.OraclePrimitive(data)
    rng = np.random.default_rng(4)
    S = rng.standard_normal((50, 40))
    cons = [{"type": "joint_position", "joint": "LeftHand_EndSite", "t": 155.0, "weight": 1.0, "target": [30.0, 90.0, -20.0]},
            {"type": "joint_position", "joint": "RightFoot", "t": 40.5, "weight": 2.0, "target": [None, 0.0, None]},
            {"type": "joint_position", "joint": "Head_EndSite", "t": 0.0, "weight": 0.5, "target": [0.0, 160.0, None]},
            {"type": "joint_position", "joint": "Hips", "t": 77.0, "weight": 1.0, "target": [10.0, None, -5.0]},
            {"type": "position", "t": 77.0, "weight": 1.0, "target": [10.0, None, -5.0]},
            {"type": "direction", "t": 155.0, "weight": 0.3, "target": [0.2, 1.0]}]
    cset = _capi.ConstraintSet(prim, cons, sk)
    res = prim.score_constraint_residuals(cset, S)
    ref = op.joint_position_residuals(S, cons[:4], joints, animated)
    np.testing.assert_allclose(res[:, :4], ref, rtol=1e-10, atol=1e-8)
    np.testing.assert_allclose(res[:, 3], res[:, 4], rtol=1e-13, atol=1e-12)       # root joint == FK-free constraint
    np.testing.assert_allclose(res[:, 4:], op.keyframe_residuals(S, cons[4:]), rtol=1e-10, atol=1e-8)
    np.testing.assert_allclose(prim.score_constraints(cset, S), res.sum(axis=1), rtol=1e-13, atol=1e-12)
    cset.close()
    with pytest.raises(ValueError):
        _capi.ConstraintSet(prim, cons[:1])                                          # no skeleton
    prim.close()

    # known answers: a primitive whose every control point is one pose, zero variance -> FK of that pose
    tiny = synthetic.make_primitive(seed=2, n_components=3, n_frames=12, n_basis=7, n_dim=79, n_gmm=2, name="tiny79")
    nb = int(tiny["n_basis_spatial"])
    pose = np.zeros(79)
    pose[3::4][:19] = 1.0                                                            # identity quaternions
    pose[:3] = [1.0, 2.0, 3.0]
    half = np.sqrt(0.5)
    for name, quat, expect in (("identity", None, [76.0, 35.0, 3.5]),                # sum of the offsets along the chain
                               ("hips +90deg about y", (half, 0.0, half, 0.0), [1.5, 35.0, -72.0])):
        p = pose.copy()
        if quat is not None:
            p[3:7] = quat
        model = dict(tiny)
        model["mean_spatial_vector"] = np.tile(p, nb)
        model["eigen_vectors_spatial"] = np.zeros_like(np.asarray(tiny["eigen_vectors_spatial"], dtype=np.float64))
        model["translation_maxima"] = np.ones(3)
        pr = _capi.Primitive(ctx, model)
        cs = _capi.ConstraintSet(pr, [{"type": "joint_position", "joint": "LeftHand_EndSite", "t": 1.0, "weight": 1.0,
                                       "target": [0.0, 0.0, 0.0]}], sk)
        d = pr.score_constraints(cs, np.zeros((3, pr.n_components)))
        np.testing.assert_allclose(d, np.linalg.norm(expect), rtol=1e-12, err_msg=name)
        cs.close()
        pr.close()


def test_time_grids_that_run_backwards_or_jump(ctx):
    """Caller-defined time grids (MotionSpline.evaluate(t), reference motion_spline.py:89-92) need not be increasing:
    the persistent kernel carries row tiles over between consecutive chunks only when the next window starts at or
    after the previous one.  Descending, shuffled and repeated sample times on the MFMA path, bit for bit."""
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    cp = c_oracle.COraclePrimitive(data)
    rng = np.random.default_rng(8)
    S = rng.standard_normal((40, 40)).astype(np.float32)
    grids = {"descending": np.linspace(155.0, 0.0, 97), "shuffled": rng.permutation(np.linspace(0.0, 155.0, 120)),
             "repeated": np.repeat(np.linspace(3.0, 150.0, 9), 11), "zigzag": np.concatenate([np.linspace(0, 155, 40), np.linspace(155, 0, 40)])}
    for name, times in grids.items():
        grid = _capi.TimeGrid(prim, times)
        model = cp.frames_f32model(S.astype(np.float64), tp=times)
        for path in (_capi.MG_PATH_MFMA, _capi.MG_PATH_DIRECT):
            got = prim.back_project_frames(S, grid=grid, path=path)
            np.testing.assert_array_equal(_bits(got), _bits(model), err_msg="%s path %d" % (name, path))
        grid.close()
    prim.close()


def test_placed_output_buffer_and_options(ctx):
    """mg_device_malloc_placed hands out a usable buffer with its probe record (small buffers are not probed);
    mg_device_probe_placement reports the same quantities for any buffer; unknown options are refused."""
    small = ctx.malloc_placed(1 << 20)
    assert small.placement["probed"] == 0 and small.placement["fast"]
    small.free()
    nbytes = 2048 * 156 * 79 * 4          # 101 MB
    buf = ctx.malloc_placed(nbytes, max_candidates=3)
    pl = buf.placement
    assert 0 <= pl["probed"] <= 3 and pl["ratio"] > 0.5 and pl["pattern_us"] > 1.0   # 0: a piece of a region an earlier test left
    again = ctx.probe_placement(buf)
    assert again["probed"] == 1 and abs(again["ratio"] - pl["ratio"]) < 0.5
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    S = np.random.default_rng(3).standard_normal((2048, 40)).astype(np.float32)
    d_S = ctx.upload(S)
    prim.back_project_frames_dev(d_S, np.float32, 2048, 40, buf, path=_capi.MG_PATH_MFMA)
    got = ctx.download(buf, (2048, 156, 79), np.float32)
    np.testing.assert_array_equal(_bits(got), _bits(prim.back_project_frames(S, path=_capi.MG_PATH_MFMA)))
    for b in (d_S, buf):
        b.free()
    prim.close()
    with pytest.raises(_capi.MGError):
        ctx.set_option(99, 1)
    with pytest.raises(_capi.MGError):
        ctx.set_option(_capi.MG_OPT_RING_SLOTS, -1)


def test_large_buffers_are_pieces_of_placed_regions():
    """mg_device_malloc from 64 MiB on hands out pieces of the context's placed regions: a freed piece goes back to its region
    (the next request of that size reuses it without a new scan), two smaller requests share a region, the accounting and
    mg_context_trim_outputs work, MG_OPT_PLAIN_MALLOC opts out, and the scratch block behind a *_host call is placed too."""
    ctx = _capi.Context(0)                # a context of its own: the module's may hold a placed scratch block
    assert ctx.output_bytes() == (0, 0, 0, 0)
    nbytes = 3000 * 156 * 79 * 4          # 148 MB
    a = ctx.malloc(nbytes)
    reserved, used, regions, fast = ctx.output_bytes()
    assert regions == 1 and reserved >= nbytes and used == reserved and reserved % (2 << 20) == 0
    addr = a.address
    _capi._check(ctx.lib.mg_memset(ctx.handle, a.ptr, 0, nbytes))
    a.free()
    assert ctx.output_bytes()[1] == 0 and ctx.output_bytes()[2] == 1          # the region stays
    b = ctx.malloc_placed(nbytes)
    assert b.address == addr and b.placement["probed"] == 0                   # reused: no scan
    b.free()
    c1, c2 = ctx.malloc(70 << 20), ctx.malloc(70 << 20)                        # two pieces of the one region
    assert ctx.output_bytes()[2] == 1 and c1.address == addr and c2.address == addr + (70 << 20)
    c1.free()
    c2.free()
    d = ctx.malloc(nbytes)                                                      # the pieces merged again
    assert d.address == addr
    d.free()
    ctx.trim_outputs()
    assert ctx.output_bytes() == (0, 0, 0, 0)
    ctx.set_option(_capi.MG_OPT_PLAIN_MALLOC, 1)
    try:
        e = ctx.malloc(nbytes)
        assert ctx.output_bytes()[2] == 0
        e.free()
    finally:
        ctx.set_option(_capi.MG_OPT_PLAIN_MALLOC, 0)
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    S = np.random.default_rng(5).standard_normal((1500, 40)).astype(np.float32)
    frames = prim.back_project_frames(S, path=_capi.MG_PATH_MFMA)               # host entry point: scratch of 74 MB
    assert frames.shape == (1500, 156, 79) and ctx.output_bytes()[2] >= 1
    prim.close()
    ctx.close()


def test_placed_allocator_walks_both_recipes_when_nothing_is_fast(ctx):
    """With an acceptance ratio no buffer can meet, mg_device_malloc_placed probes its plain candidates, then the
    twelve assembled from physical chunks by the virtual-memory API, and returns the best of all of them: a buffer
    that works like any other and is released by mg_device_free whichever recipe it came from."""
    nbytes = 2048 * 156 * 79 * 4
    ctx.trim_outputs()                    # no region from an earlier test may serve this request
    ctx.set_option(_capi.MG_OPT_PLACED_FAST_PCT, 50)
    try:
        buf = ctx.malloc_placed(nbytes, max_candidates=20)
    finally:
        ctx.set_option(_capi.MG_OPT_PLACED_FAST_PCT, 0)
    pl = buf.placement
    assert pl["probed"] in (16, 20) and not pl["fast"] and pl["ratio"] > 0.5   # 16: a box without the virtual-memory API
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    S = np.random.default_rng(4).standard_normal((2048, 40)).astype(np.float32)
    d_S = ctx.upload(S)
    prim.back_project_frames_dev(d_S, np.float32, 2048, 40, buf, path=_capi.MG_PATH_MFMA)
    got = ctx.download(buf, (2048, 156, 79), np.float32)
    np.testing.assert_array_equal(_bits(got), _bits(prim.back_project_frames(S, path=_capi.MG_PATH_MFMA)))
    # where the frames go decides the kernel: a piece of a region whose scan found nothing fast is written by the tile-major kernel
    # (the faster one on slow-class memory), anything else by the batch-size rule; the bits are the same either way
    assert prim.step_plan(2048)["kernel"] == "mg_frames_cs_kernel"
    assert prim.step_plan(2048, buf)["kernel"] == "mg_frames_ws_kernel"
    plain = ctx.malloc(1 << 20)
    assert prim.step_plan(2048, plain)["kernel"] == "mg_frames_cs_kernel"        # not the arena's: class unknown
    plain.free()
    for b in (d_S, buf):
        b.free()
    prim.close()


def test_default_scan_goes_deep_when_its_first_candidates_are_all_slow():
    """Fast-class memory is sparse (tools/probes/deep_scan.py: 14 of 400 allocations of 404 MB on one box), and a scan of 32 misses
    it on one box in three.  The DEFAULT scan therefore goes on after 32 slow candidates -- plain allocations, held, up to 400 or six
    tenths of the free memory (fast memory comes in clusters: on one box the first was the 181st candidate) -- while an explicit budget is
    exact.  With an acceptance ratio nothing can meet: the default scan probes 400 candidates (388 on a box without the virtual-memory
    API), an explicit budget of 8 probes 8; both return working buffers and release every rejected candidate."""
    ctx = _capi.Context(0)
    nbytes = 64 << 20
    before = ctx.output_bytes()
    ctx.set_option(_capi.MG_OPT_PLACED_FAST_PCT, 1)
    try:
        deep = ctx.malloc_placed(nbytes)
        exact = ctx.malloc_placed(nbytes + (1 << 20), max_candidates=8)
    finally:
        ctx.set_option(_capi.MG_OPT_PLACED_FAST_PCT, 0)
    assert deep.placement["probed"] in (400, 388) and not deep.placement["fast"]
    assert exact.placement["probed"] == 8
    for buf in (deep, exact):
        x = np.arange(1 << 20, dtype=np.float32)
        ctx.upload_into(buf, x)
        np.testing.assert_array_equal(ctx.download(buf, x.shape, np.float32), x)
    assert ctx.output_bytes()[0] - before[0] <= 2 * (nbytes + (2 << 20))      # only the two regions stay reserved
    deep.free(); exact.free()
    ctx.close()


def test_placement_scan_that_drops_candidates_leaves_no_stale_mapping():
    """VERDICT r3 item 3a.  The builder's own probe had found that unmap + map at an overlapping virtual range makes kernels store
    through stale translations (profiles/r03_placement/t8_remap_stale_translation.log, "STORES LOST").  The product guard: the
    address range of a released chunked buffer stays reserved with the context (parked), so no later reservation can land on it.
    Here a scan that may hold only two candidates at once walks all 32 -- sixteen plain, twelve assembled from chunks, four
    plain: candidates are dropped all along -- and the frames kernel then writes a position-dependent result through the
    surviving region, which must read back bit for bit what a plain hipMalloc buffer reads back; and a chunked buffer
    allocated after another one was freed gets a range of its own and holds its data."""
    ctx = _capi.Context(0)
    nbytes = 2048 * 156 * 79 * 4           # 101 MB
    ctx.set_option(_capi.MG_OPT_PLACED_FAST_PCT, 1)      # no candidate can be accepted: the scan runs to its end
    ctx.set_option(_capi.MG_OPT_PLACED_HOLD, 2)
    buf = ctx.malloc_placed(nbytes)
    assert buf.placement["probed"] in (20, 32) and not buf.placement["fast"]      # (20: a box without the virtual-memory API)
    info = ctx.placement_info(buf)
    assert info["region"] and info["fast"] == buf.placement["fast"] and info["pattern_us"] == buf.placement["pattern_us"]
    ctx.set_option(_capi.MG_OPT_PLACED_FAST_PCT, 0)
    ctx.set_option(_capi.MG_OPT_PLACED_HOLD, 0)
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    S = np.random.default_rng(8).standard_normal((2048, 40)).astype(np.float32)
    d_S = ctx.upload(S)
    ctx.set_option(_capi.MG_OPT_PLAIN_MALLOC, 1)
    plain = ctx.malloc(nbytes)
    ctx.set_option(_capi.MG_OPT_PLAIN_MALLOC, 0)
    for kern in (1, 2):
        ctx.set_option(_capi.MG_OPT_FRAMES_KERNEL, kern)
        for target in (buf, plain):
            _capi._check(ctx.lib.mg_memset(ctx.handle, target.ptr, 0xff, nbytes))
            prim.back_project_frames_dev(d_S, np.float32, 2048, 40, target, path=_capi.MG_PATH_MFMA)
        a = ctx.download(buf, (2048, 156, 79), np.float32)
        b = ctx.download(plain, (2048, 156, 79), np.float32)
        np.testing.assert_array_equal(_bits(a), _bits(b), err_msg="kernel %d" % kern)
        assert np.isfinite(a).all()
    ctx.set_option(_capi.MG_OPT_FRAMES_KERNEL, 0)
    # chunked buffers one after the other: every one at an address range no earlier one had
    seen = set()
    for chunk_mib in (2, 8, 2, 32, 8):
        c = ctx.malloc(nbytes, chunk_bytes=chunk_mib << 20)
        assert c.address not in seen, "the address range of a released chunked buffer was handed out again"
        seen.add(c.address)
        prim.back_project_frames_dev(d_S, np.float32, 2048, 40, c, path=_capi.MG_PATH_MFMA)
        np.testing.assert_array_equal(_bits(ctx.download(c, (2048, 156, 79), np.float32)), _bits(b))
        c.free()
    for x in (d_S, buf, plain):
        x.free()
    prim.close()
    ctx.close()


def test_growing_requests_do_not_pile_up_idle_regions():
    """a workload whose output grows from call to call (each buffer freed before the next is asked for) must not keep
    every earlier, too small region reserved beside the new one: idle regions that cannot serve a request go before its scan."""
    ctx = _capi.Context(0)
    unit = 1024 * 156 * 79 * 4            # 50 MB
    last = 0
    for k in (2, 3, 5, 8):
        b = ctx.malloc(k * unit)
        reserved, used, regions, _ = ctx.output_bytes()
        assert regions == 1 and used == reserved and k * unit <= reserved < k * unit + (4 << 20), (k, reserved, regions)
        assert reserved > last
        last = reserved
        b.free()
    b = ctx.malloc(3 * unit)              # a smaller one afterwards is a piece of the region that is there
    assert ctx.output_bytes()[0] == last and ctx.output_bytes()[2] == 1
    b.free()
    ctx.close()


def _set_frames_kernel(ctx, which):
    ctx.set_option(_capi.MG_OPT_FRAMES_KERNEL, which)


@pytest.mark.parametrize("B", [16, 17, 255, 1000, 4099, 8192 + 5])
def test_chunk_stationary_kernel_is_bit_identical(ctx, B):
    """The chunk-stationary frames kernel (a workgroup keeps ONE time chunk's eigenvector window in registers and
    walks candidate tiles; chosen by itself for large batches) against the tile-major kernel and the oracle's f32
    model: ragged last tiles, fewer tiles than workgroups of a chunk (idle workgroups), float32 and float64 latents,
    stand-alone and inside the fused step, repeated launches -- bit for bit."""
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    cp = c_oracle.COraclePrimitive(data)
    rng = np.random.default_rng(7000 + B)
    for dtype in (np.float32, np.float64):
        S = rng.standard_normal((B, 40)).astype(dtype)
        n_model = min(B, 400)
        model = cp.frames_f32model(S[:n_model].astype(np.float64))
        model_tail = cp.frames_f32model(S[-40:].astype(np.float64))
        _set_frames_kernel(ctx, 1)
        tm = prim.back_project_frames(S, path=_capi.MG_PATH_MFMA)
        _set_frames_kernel(ctx, 2)
        for rep in range(2):
            cs = prim.back_project_frames(S, path=_capi.MG_PATH_MFMA)
            np.testing.assert_array_equal(_bits(cs), _bits(tm), err_msg="B=%d %s rep %d" % (B, dtype, rep))
        np.testing.assert_array_equal(_bits(cs[:n_model]), _bits(model))
        np.testing.assert_array_equal(_bits(cs[-40:][-min(B, 40):]), _bits(model_tail[-min(B, 40):]))
        frames, logp = _fused_step(ctx, prim, S, 156, 79)
        np.testing.assert_array_equal(_bits(frames), _bits(tm), err_msg="fused B=%d %s" % (B, dtype))
        np.testing.assert_array_equal(logp, prim.gmm_log_prob(S, dtype=np.float32))
    prim.close()


@pytest.mark.parametrize("shape", [dict(n_components=40, n_frames=156, n_dim=79, n_gmm=16), dict(n_components=40, n_frames=156, n_dim=79, n_gmm=1),
                                   dict(n_components=24, n_frames=60, n_dim=79, n_gmm=3), dict(n_components=13, n_frames=47, n_dim=15, n_gmm=5),
                                   dict(n_components=33, n_frames=97, n_dim=79, n_gmm=11)])
def test_fused_tail_with_staged_mixture_constants(ctx, shape):
    """The chunk-stationary kernel's mixture tail for float32 latents (round 5): the workgroup's two latent tiles, the components'
    C-in rows and constants staged in LDS at start-up by the four sweep waves that produce nothing, both components of a producer
    wave requested at once.  Mixtures of 1 .. 16 components (more C-in entries than one pass of the staging wave; an odd count: a
    wave with one component only), latent counts that are no multiple of four, one and two tiles per workgroup and workgroups without
    any: log p bit for bit the stand-alone kernel's, frames the tile-major kernel's; float64 latents keep the unstaged form."""
    data = synthetic.make_primitive(seed=77, name="tail", **shape)
    prim = _capi.Primitive(ctx, data)
    if not prim.mfma_supported:
        prim.close()
        pytest.skip("no LDS-staged kernel for this shape")
    F, D, L = shape["n_frames"], shape["n_dim"], shape["n_components"]
    rng = np.random.default_rng(5)
    ran = 0
    for B in (16, 1000, 4099, 8192):
        for dtype in (np.float32, np.float64):
            S = rng.standard_normal((B, L)).astype(dtype)
            _set_frames_kernel(ctx, 1)
            tm = prim.back_project_frames(S, path=_capi.MG_PATH_MFMA)
            _set_frames_kernel(ctx, 2)
            try:
                frames, logp = _fused_step(ctx, prim, S, F, D)
            except _capi.MGError as e:
                assert e.status == -4, e          # MG_ERR_UNSUPPORTED: the shape is the tile-major kernel's
                continue
            ran += 1
            np.testing.assert_array_equal(_bits(frames), _bits(tm), err_msg="B=%d %s" % (B, dtype))
            np.testing.assert_array_equal(logp, prim.gmm_log_prob(S, dtype=np.float32), err_msg="B=%d %s" % (B, dtype))
    _set_frames_kernel(ctx, 0)
    prim.close()
    if not ran:
        pytest.skip("the chunk-stationary kernel does not cover this shape")


def test_chunk_stationary_kernel_on_the_golden_shapes_and_other_grids(ctx, golden_case):
    """Every golden shape the chunk-stationary kernel covers (the others must say MG_ERR_UNSUPPORTED, never run
    something else): canonical grid against the reference's frames, and evaluation grids with fractional, repeated,
    backwards and out-of-range times (chunks of other lengths and windows) against the tile-major kernel."""
    name, data, g = golden_case
    prim = _capi.Primitive(ctx, data)
    if not prim.mfma_supported:
        prim.close()
        pytest.skip("no LDS-staged kernel for this shape")
    cp = c_oracle.COraclePrimitive(data)
    rng = np.random.default_rng(11)
    L = g["S"].shape[1]
    S = np.concatenate([g["S"], rng.standard_normal((333, L))]).astype(np.float64)
    _set_frames_kernel(ctx, 2)
    try:
        got = prim.back_project_frames(S, path=_capi.MG_PATH_MFMA)
    except _capi.MGError as e:
        assert e.status == -4, e          # MG_ERR_UNSUPPORTED
        prim.close()
        pytest.skip("window of %s does not fit the row producers' registers" % name)
    np.testing.assert_array_equal(_bits(got), _bits(cp.frames_f32model(S)))
    ref = g["frames"]
    assert np.all(np.abs(got[:len(ref)].astype(np.float64) - ref) <= pose_tol(ref))
    F = int(data["n_canonical_frames"])
    grids = [np.linspace(0.0, F, 2 * F + 3), np.array([F - 1.0, 0.25, 0.25, F / 2.0, -3.0, F + 5.0, 1.0]),
             np.linspace(F, 0.0, 77), np.array([0.5 * (F - 1)])]
    for times in grids:
        grid = prim.time_grid(times)
        _set_frames_kernel(ctx, 1)
        tm = prim.back_project_frames(S, grid=grid, path=_capi.MG_PATH_MFMA)
        _set_frames_kernel(ctx, 2)
        try:
            cs = prim.back_project_frames(S, grid=grid, path=_capi.MG_PATH_MFMA)
        except _capi.MGError as e:
            assert e.status == -4, e
            grid.close()
            continue
        np.testing.assert_array_equal(_bits(cs), _bits(tm), err_msg="%s grid of %d" % (name, len(times)))
        grid.close()
    prim.close()


def test_step_plan_names_what_a_step_launches(ctx):
    """mg_step_plan: small batches on the tile-major kernel, from two units per workgroup on the chunk-stationary one
    (whole workgroups per chunk), log p(x) inside the frames kernel for 'walk'; the option overrides the choice; a
    mixture over spatial AND time latents is scored by a second launch."""
    prim = _capi.Primitive(ctx, synthetic.make_walk_primitive(seed=0))
    small, large = prim.step_plan(256), prim.step_plan(8192)
    assert small["kernel"] == "mg_frames_ws_kernel" and small["fused"]
    assert large["kernel"] == "mg_frames_cs_kernel" and large["fused"] and large["workgroups"] % 4 == 0
    assert 0 < large["lds_bytes"] <= 160 * 1024
    _set_frames_kernel(ctx, 1)
    assert prim.step_plan(8192)["kernel"] == "mg_frames_ws_kernel"
    _set_frames_kernel(ctx, 0)
    assert prim.step_plan(4)["kernel"] == "mg_frames_direct_kernel"
    prim.close()
    timed = _capi.Primitive(ctx, synthetic.make_primitive(seed=13, n_components=12, n_frames=60, n_dim=15, n_gmm=3, name="timed",
                                                          n_time_components=3, n_basis_time=8))
    assert not timed.step_plan(4096)["fused"]
    timed.close()


def test_large_batches_of_every_golden_shape_take_the_same_bits_through_either_kernel(ctx, golden_case):
    """At a batch size where the library picks the chunk-stationary kernel by itself (other numbers of latent components:
    other template instances, other register budgets, other chunk plans): frames and fused log p(x) of 6000 + 3
    candidates, library's choice against the tile-major kernel forced, the first rows against the oracle's f32 model."""
    name, data, g = golden_case
    prim = _capi.Primitive(ctx, data)
    if not prim.mfma_supported:
        prim.close()
        pytest.skip("no LDS-staged kernel for this shape")
    cp = c_oracle.COraclePrimitive(data)
    L = g["S"].shape[1]
    F, D = int(data["n_canonical_frames"]), int(data["n_dim_spatial"])
    S = np.random.default_rng(17).standard_normal((6003, L)).astype(np.float32)
    plan = prim.step_plan(len(S))
    _set_frames_kernel(ctx, 0)
    auto = prim.back_project_frames(S, path=_capi.MG_PATH_MFMA)
    _set_frames_kernel(ctx, 1)
    tm = prim.back_project_frames(S, path=_capi.MG_PATH_MFMA)
    np.testing.assert_array_equal(_bits(auto), _bits(tm), err_msg="%s (%s)" % (name, plan["kernel"]))
    np.testing.assert_array_equal(_bits(auto[:200]), _bits(cp.frames_f32model(S[:200].astype(np.float64))))
    _set_frames_kernel(ctx, 0)
    frames, logp = _fused_step(ctx, prim, S, F, D)
    np.testing.assert_array_equal(_bits(frames), _bits(tm))
    np.testing.assert_array_equal(_bits(logp), _bits(prim.gmm_log_prob(S, dtype=np.float32)))
    prim.close()


def test_two_slot_ring(ctx):
    """Shapes whose three LDS slots do not fit 160 KiB run a two-slot ring; mg_context_set_option(MG_OPT_RING_SLOTS, 2) forces it on 'walk' (the
    planner reads it when a grid is built).  Carried-over tiles then need a full meeting of the row producers per
    unit: ragged and multi-tile batches, the fused step, repeated launches, bit for bit."""
    ctx.set_option(_capi.MG_OPT_RING_SLOTS, 2)
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)          # canonical grid planned with two ring slots
    cp = c_oracle.COraclePrimitive(data)
    rng = np.random.default_rng(12)
    for B in (17, 1000, 4099):
        S = rng.standard_normal((B, 40)).astype(np.float32)
        model = cp.frames_f32model(S[:600].astype(np.float64))
        for rep in range(3):
            got = prim.back_project_frames(S, path=_capi.MG_PATH_MFMA)
            np.testing.assert_array_equal(_bits(got[:600]), _bits(model), err_msg="B=%d rep=%d" % (B, rep))
        frames, logp = _fused_step(ctx, prim, S, 156, 79)
        np.testing.assert_array_equal(_bits(frames), _bits(got))
        np.testing.assert_array_equal(_bits(logp), _bits(prim.gmm_log_prob(S, dtype=np.float32)))
    prim.close()


def test_large_ragged_batch_on_device(ctx):
    """40 001 candidates (2 GB of frames, 157 units per workgroup, a ragged last tile): device-resident in and out,
    a seeded subset and both ends against the f32 model, per-candidate checksums computed on the host from two
    independent launches (fused step and stand-alone kernel) must agree bit for bit."""
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    cp = c_oracle.COraclePrimitive(data)
    B = 40001
    rng = np.random.default_rng(99)
    S = rng.standard_normal((B, 40)).astype(np.float32)
    d_S = ctx.upload(S)
    d_f = ctx.malloc(B * 156 * 79 * 4)
    d_l = ctx.malloc(B * 4)
    prim.back_project_frames_dev(d_S, np.float32, B, 40, d_f, path=_capi.MG_PATH_MFMA)
    ctx.synchronize()
    F1 = ctx.download(d_f, (B, 156, 79), np.float32)
    idx = np.concatenate([[0, 1, 15, 16, B - 17, B - 2, B - 1], rng.choice(B, size=40, replace=False)])
    np.testing.assert_array_equal(_bits(F1[idx]), _bits(cp.frames_f32model(S[idx].astype(np.float64))))
    assert np.isfinite(F1).all()
    prim.step_frames_and_logp_dev(d_S, np.float32, B, 40, d_f, d_l)     # > 2 tiles per workgroup: two launches
    ctx.synchronize()
    F2 = ctx.download(d_f, (B, 156, 79), np.float32)
    np.testing.assert_array_equal(_bits(F2), _bits(F1))
    lp = ctx.download(d_l, (B,), np.float32)
    np.testing.assert_allclose(lp[idx], cp.log_prob_f64(S[idx].astype(np.float64)), rtol=3e-7, atol=1e-5)
    for buf in (d_S, d_f, d_l):
        buf.free()
    prim.close()


def test_score_kernels_mfma_and_valu_agree_bit_for_bit(ctx):
    """The fused keyframe scorer has two kernels: channels = X . W^T on the f64 matrix pipe (n_components <= 64) and
    a dot product per channel on the VALU (fallback).  Both run the same k-ordered fma chain from the bias and share
    the residual code, so errors and residual matrices must be identical, ragged batches included."""
    joints, animated = synthetic.make_skeleton()
    sk = _capi.Skeleton(joints, animated)
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    cons = [{"type": "position", "t": 155.0, "weight": 1.0, "target": [40.0, None, -30.0]},
            {"type": "direction", "t": 155.0, "weight": 0.5, "target": [0.5, 1.0]},
            {"type": "joint_position", "joint": "LeftHand_EndSite", "t": 100.25, "weight": 1.0, "target": [30.0, 90.0, -20.0]},
            {"type": "joint_position", "joint": "RightFoot", "t": 40.5, "weight": 2.0, "target": [None, 0.0, None]},
            {"type": "position", "t": 0.0, "weight": 0.1, "target": [0.0, 0.0, 0.0]}]
    cset = _capi.ConstraintSet(prim, cons, sk)
    rng = np.random.default_rng(6)
    for B in (1, 15, 64, 65, 1000):
        for dtype in (np.float32, np.float64):
            S = rng.standard_normal((B, 40)).astype(dtype)
            ctx.set_option(_capi.MG_OPT_FORCE_VALU_SCORE, 0)
            e1, r1 = prim.score_constraints(cset, S), prim.score_constraint_residuals(cset, S)
            ctx.set_option(_capi.MG_OPT_FORCE_VALU_SCORE, 1)
            e2, r2 = prim.score_constraints(cset, S), prim.score_constraint_residuals(cset, S)
            np.testing.assert_array_equal(e1.view(np.uint64), e2.view(np.uint64), err_msg="B=%d %s" % (B, dtype))
            np.testing.assert_array_equal(r1.view(np.uint64), r2.view(np.uint64), err_msg="B=%d %s" % (B, dtype))
            np.testing.assert_allclose(r1.sum(axis=1), e1, rtol=1e-13, atol=1e-12)
    cset.close()
    prim.close()


def test_device_sampler_mfma_and_valu_produce_the_same_rows(ctx):
    """The device sampler has an MFMA kernel (16-row tiles inside one component, x = mu + z L^T on the f64 matrix
    pipe) and a lane-per-row VALU kernel; both draw z from the same Philox counters (row, group of four) and run the
    same ascending fma chain, so the same seed must give the same rows bit for bit -- ragged component counts
    (tiles that end inside a component), empty components, float32 and float64 output."""
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    for counts in ([100, 0, 17, 1, 250, 16, 33, 95], [0, 0, 0, 0, 0, 0, 0, 5], [4096] + [0] * 7):
        for dtype in (np.float64, np.float32):
            ctx.set_option(_capi.MG_OPT_FORCE_VALU_SAMPLE, 0)
            X1, c1 = prim.gmm_sample(counts, 1234, dtype=dtype)
            ctx.set_option(_capi.MG_OPT_FORCE_VALU_SAMPLE, 1)
            X2, c2 = prim.gmm_sample(counts, 1234, dtype=dtype)
            view = np.uint64 if dtype == np.float64 else np.uint32
            np.testing.assert_array_equal(X1.view(view), X2.view(view), err_msg=str(counts))
            np.testing.assert_array_equal(c1, c2)
            np.testing.assert_array_equal(c1, np.repeat(np.arange(8), counts))
            assert np.isfinite(X1).all()
    prim.close()


def test_fused_step_with_reserved_cus_and_up_to_four_tiles_per_workgroup(ctx):
    """mg_context_set_reserved_cus leaves CUs free for kernels on other streams (RCCL's all-gather): the persistent
    kernel then runs fewer, longer workgroups and the fused mixture scoring handles three or four 16-candidate tiles
    per workgroup in two groups.  Frames and scores must not change by a bit; above four tiles the entry point takes
    two launches."""
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    cp = c_oracle.COraclePrimitive(data)
    rng = np.random.default_rng(13)
    n_cu = ctx.device_info()["n_cu"]
    try:
        for reserved, B in ((0, 12 * n_cu + 5), (0, 64 * n_cu), (8, 32 * n_cu), (8, 33 * n_cu + 7), (n_cu - 3, 200), (0, 70 * n_cu)):
            ctx.set_reserved_cus(reserved)
            S = rng.standard_normal((B, 40)).astype(np.float32)
            frames, logp = _fused_step(ctx, prim, S, 156, 79)
            idx = np.concatenate([[0, B - 1], rng.choice(B, size=24, replace=False)])
            np.testing.assert_array_equal(_bits(frames[idx]), _bits(cp.frames_f32model(S[idx].astype(np.float64))),
                                          err_msg="reserved=%d B=%d" % (reserved, B))
            np.testing.assert_array_equal(_bits(logp), _bits(prim.gmm_log_prob(S, dtype=np.float32)), err_msg="reserved=%d B=%d" % (reserved, B))
            np.testing.assert_array_equal(_bits(frames), _bits(prim.back_project_frames(S, path=_capi.MG_PATH_MFMA)))
    finally:
        ctx.set_reserved_cus(0)
    prim.close()


def test_constraints_in_global_coordinates_align_every_candidate_to_the_previous_motion(ctx):
    """MotionPrimitiveConstraints.evaluate outside local mode (reference motion_primitive_constraints.py:110-114 ->
    anim_utils align_quaternion_frames_automatically; PARITY UNPINNED, anim_utils absent): per candidate a rotation
    about y and an xz translation attach its first control point to the previous motion's last frame, then the
    constraints are evaluated.  The device never transforms control points (closed form on the evaluated
    quantities); the oracle does, with a 4x4 matrix and a quaternion product like the library.  Also: known
    answers (the aligned start sits on the previous root, the aligned start heading is the previous heading), the
    MFMA and the VALU kernel agree, and alignment through a chain (aligning node != root)."""
    from oracle import mg_oracle as orc
    joints, animated = synthetic.make_skeleton()
    sk = _capi.Skeleton(joints, animated)
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    op = orc.OraclePrimitive(data)
    rng = np.random.default_rng(12)
    S = rng.standard_normal((70, 40))
    prev = op.back_project_frames(rng.standard_normal(40))[-1].copy()        # some last frame of a previous step
    prev[:3] = [120.0, 90.0, -340.0]
    prev[3:7] = [0.3, 0.1, 0.9, -0.2]                                        # not unit: normalised like quaternion_matrix
    cons = [{"type": "position", "t": 155.0, "weight": 1.0, "target": [150.0, None, -300.0]},
            {"type": "position", "t": 77.5, "weight": 0.5, "target": [130.0, 88.0, -320.0]},
            {"type": "direction", "t": 155.0, "weight": 0.3, "target": [0.2, 1.0]},
            {"type": "joint_position", "joint": "LeftHand_EndSite", "t": 100.0, "weight": 2.0, "target": [160.0, 120.0, -310.0]},
            {"type": "joint_position", "joint": "Hips", "t": 0.0, "weight": 1.0, "target": [0.0, None, 0.0]}]
    for node in ("Hips", "Spine1"):
        al = sk.alignment_to(prev, node)
        np.testing.assert_allclose(al["heading"], orc.node_heading(prev, joints, animated, node), rtol=1e-13, atol=1e-14)
        cset = _capi.ConstraintSet(prim, cons, sk, alignment=al)
        res = prim.score_constraint_residuals(cset, S)
        ref = op.aligned_residuals(S, cons, prev, joints, animated, node)
        np.testing.assert_allclose(res, ref, rtol=1e-9, atol=1e-8, err_msg=node)
        np.testing.assert_allclose(prim.score_constraints(cset, S), res.sum(axis=1), rtol=1e-13, atol=1e-12)
        # the aligned start: root xz of the first control point lands on the previous root -> |(prev_x, ., prev_z)|
        np.testing.assert_allclose(res[:, 4], np.hypot(prev[0], prev[2]), rtol=1e-12)
        ctx.set_option(_capi.MG_OPT_FORCE_VALU_SCORE, 1)
        np.testing.assert_array_equal(prim.score_constraint_residuals(cset, S), res)
        ctx.set_option(_capi.MG_OPT_FORCE_VALU_SCORE, 0)
        best, err = prim.best_candidate(cset, S)
        assert best == int(np.argmin(ref.sum(axis=1))) and abs(err - ref.sum(axis=1).min()) < 1e-7
        cset.close()
    # root alignment needs no skeleton; a heading constraint at t = 0 then reads the angle between the previous heading
    # and its own target for every candidate alike
    al = sk.alignment_to(prev, 0)
    cset = _capi.ConstraintSet(prim, [{"type": "direction", "t": 0.0, "weight": 1.0, "target": [1.0, 0.0]}], None,
                               alignment={"position": al["position"], "heading": al["heading"]})
    h = np.asarray(al["heading"])
    expect = abs(np.degrees(np.arccos(np.clip(h[0], -1.0, 1.0))))
    np.testing.assert_allclose(prim.score_constraints(cset, S), expect, rtol=1e-9, atol=1e-9)
    cset.close()
    # edges: no constraints at all (error 0 for every candidate), one candidate, a ragged batch, float32 latents
    al = sk.alignment_to(prev, "Spine1")
    cset = _capi.ConstraintSet(prim, [], sk, alignment=al)
    np.testing.assert_array_equal(prim.score_constraints(cset, S[:5]), np.zeros(5))
    assert prim.score_constraint_residuals(cset, S[:5]).shape == (5, 0)
    cset.close()
    cset = _capi.ConstraintSet(prim, cons, sk, alignment=al)
    for n in (1, 17):
        np.testing.assert_allclose(prim.score_constraint_residuals(cset, S[:n]), op.aligned_residuals(S[:n], cons, prev, joints, animated, "Spine1"),
                                   rtol=1e-9, atol=1e-8)
    S32 = S[:17].astype(np.float32)
    np.testing.assert_allclose(prim.score_constraint_residuals(cset, S32), op.aligned_residuals(S32.astype(np.float64), cons, prev, joints, animated, "Spine1"),
                               rtol=1e-9, atol=1e-8)
    assert prim.score_constraints(cset, S[:0]).shape == (0,)
    cset.close()
    with pytest.raises(_capi.MGError):
        _capi.ConstraintSet(prim, cons[:1], None, alignment={"joint": 3, "position": [0, 0, 0], "heading": [0, 1]})
    with pytest.raises(_capi.MGError):
        _capi.ConstraintSet(prim, cons[:1], None, alignment={"position": [0, 0, 0], "heading": [0, 0]})
    prim.close()


def test_start_pose_alignment_branch(ctx):
    """The other branch of the reference's align_quaternion_frames (optimization/objective_functions.py:38-47): no previous
    frames but a start pose.  Every candidate gets the same rotation about y; the reference's arithmetic (its `delta` IS
    the start pose's position list) puts the first root position on x = z = 0 and raises heights by the start height, call
    after call.  The oracle follows the reference's lines literally, mutation included, on the control points; the device
    applies the closed form to the evaluated quantities.  PARITY UNPINNED (get_transform_from_start_pose is anim_utils')."""
    from oracle import mg_oracle as orc
    from morphablegraphs_amd.candidate_scoring import alignment_from_start_pose, alignment_from_prev_frames
    joints, animated = synthetic.make_skeleton()
    sk = _capi.Skeleton(joints, animated)
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    op = orc.OraclePrimitive(data)
    S = np.random.default_rng(21).standard_normal((40, 40))
    cons = [{"type": "position", "t": 155.0, "weight": 1.0, "target": [30.0, None, -20.0]},
            {"type": "position", "t": 0.0, "weight": 1.0, "target": [0.0, None, 0.0]},
            {"type": "position", "t": 60.0, "weight": 0.5, "target": [10.0, 95.0, 5.0]},
            {"type": "direction", "t": 155.0, "weight": 0.3, "target": [0.2, 1.0]},
            {"type": "joint_position", "joint": "LeftHand_EndSite", "t": 100.0, "weight": 2.0, "target": [40.0, 120.0, -10.0]},
            {"type": "joint_orientation", "joint": "Hips", "t": 20.0, "weight": 1.0, "orientation": [0.9, 0.1, 0.3, -0.2]}]
    for angle in (0.0, 37.5, -110.0):
        start_pose = {"position": [55.0, 7.5, -80.0], "orientation": [0.0, angle, 0.0]}
        al = alignment_from_start_pose(start_pose)
        assert start_pose["position"] == [55.0, 7.5, -80.0]                    # the device path leaves the caller's object alone
        cset = _capi.ConstraintSet(prim, cons, sk, alignment=al)
        res = prim.score_constraint_residuals(cset, S)
        ref = op.start_pose_residuals(S, cons, {"position": [55.0, 7.5, -80.0], "orientation": [0.0, angle, 0.0]}, joints, animated)
        np.testing.assert_allclose(res, ref, rtol=1e-9, atol=1e-8, err_msg=str(angle))
        np.testing.assert_allclose(res[:, 1], 0.0, atol=1e-9)                   # the aligned start sits on x = z = 0
        ctx.set_option(_capi.MG_OPT_FORCE_VALU_SCORE, 1)
        np.testing.assert_array_equal(prim.score_constraint_residuals(cset, S), res)
        ctx.set_option(_capi.MG_OPT_FORCE_VALU_SCORE, 0)
        # the values of a cached set are rewritten in place (another angle, another height)
        al2 = alignment_from_start_pose({"position": [1.0, -3.0, 2.0], "orientation": [0.0, angle + 15.0, 0.0]})
        cset.update(cons, alignment=al2)
        ref2 = op.start_pose_residuals(S, cons, {"position": [1.0, -3.0, 2.0], "orientation": [0.0, angle + 15.0, 0.0]}, joints, animated)
        np.testing.assert_allclose(prim.score_constraint_residuals(cset, S), ref2, rtol=1e-9, atol=1e-8)
        cset.close()

    class _C(object):                                                           # what MotionPrimitiveConstraints carries
        is_local = False
        start_pose = {"position": [5.0, 1.0, 5.0], "orientation": [0.0, 90.0, 0.0]}
    rec = alignment_from_prev_frames(None, _C(), None)
    assert rec["joint"] == _capi.MG_ALIGN_START_POSE and abs(rec["heading"][1] - 1.0) < 1e-12 and rec["position"] == (0.0, 1.0, 0.0)
    _C.start_pose = {"position": [0.0, 0.0, 0.0], "orientation": [10.0, 0.0, 0.0]}
    with pytest.raises(NotImplementedError):
        alignment_from_prev_frames(None, _C(), None)
    prim.close()


def test_wide_scoring_kernel_is_bit_identical(ctx):
    """mg_score_constraints' large-batch kernel (a wave per 64 candidates, a lane per candidate walking the constraints in order)
    against the tile kernel (a wave per 16 candidates, (candidate, constraint) pairs on the lanes) and the VALU kernel: the same
    errors and residual matrices bit for bit, for ragged batches, both latent types, forward-kinematics constraints and candidates
    aligned to a start pose."""
    from morphablegraphs_amd.candidate_scoring import alignment_from_start_pose
    joints, animated = synthetic.make_skeleton()
    sk = _capi.Skeleton(joints, animated)
    prim = _capi.Primitive(ctx, synthetic.make_walk_primitive(seed=0))
    cons = [{"type": "position", "t": 155.0, "weight": 1.0, "target": [30.0, None, -20.0]},
            {"type": "direction", "t": 155.0, "weight": 0.3, "target": [0.2, 1.0]},
            {"type": "joint_position", "joint": "LeftHand_EndSite", "t": 100.0, "weight": 2.0, "target": [40.0, 120.0, -10.0]},
            {"type": "joint_orientation", "joint": "Hips", "t": 20.0, "weight": 1.0, "orientation": [0.9, 0.1, 0.3, -0.2]},
            {"type": "look_at", "joint": "Head", "t": 77.5, "weight": 0.7, "target": [100.0, 150.0, 30.0]}]
    rng = np.random.default_rng(5)
    for alignment in (None, alignment_from_start_pose({"position": [55.0, 7.5, -80.0], "orientation": [0.0, 37.5, 0.0]})):
        cset = _capi.ConstraintSet(prim, cons, sk, alignment=alignment)
        for B in (1, 63, 64, 65, 4099, 40000):
            S = rng.standard_normal((B, 40))
            for dt in (np.float32, np.float64):
                out = []
                for mode in (1, 2):
                    ctx.set_option(_capi.MG_OPT_SCORE_KERNEL, mode)
                    out.append((prim.score_constraints(cset, S.astype(dt), dtype=np.float64), prim.score_constraints(cset, S.astype(dt), dtype=np.float32),
                                prim.score_constraint_residuals(cset, S.astype(dt))))
                ctx.set_option(_capi.MG_OPT_SCORE_KERNEL, 0)
                for a, b in zip(out[0], out[1]):
                    np.testing.assert_array_equal(a.view(np.uint8), b.view(np.uint8))
                if B == 4099:
                    ctx.set_option(_capi.MG_OPT_FORCE_VALU_SCORE, 1)
                    np.testing.assert_array_equal(prim.score_constraint_residuals(cset, S.astype(dt)), out[1][2])
                    ctx.set_option(_capi.MG_OPT_FORCE_VALU_SCORE, 0)
        cset.close()
    prim.close()


def test_two_hand_midpoint_and_joint_orientation_constraints(ctx):
    """MG_CONSTRAINT_JOINT_MIDPOINT (first residual of TwoHandConstraint, reference two_hand_constraint.py:66-74) and
    MG_CONSTRAINT_JOINT_ORIENTATION (GlobalTransformConstraint._quaternion_distance, global_transform_constraint.py:
    109-121; radians) against the matrix oracle, in local coordinates and aligned to a previous motion, MFMA and VALU
    kernels bit for bit, plus known answers on a constant-pose primitive."""
    from oracle import mg_oracle as orc
    joints, animated = synthetic.make_skeleton()
    sk = _capi.Skeleton(joints, animated)
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    op = orc.OraclePrimitive(data)
    rng = np.random.default_rng(15)
    S = rng.standard_normal((60, 40))
    cons = [{"type": "joint_midpoint", "joint": "LeftHand", "joint2": "RightHand", "t": 120.0, "weight": 1.5, "target": [5.0, 95.0, 20.0]},
            {"type": "joint_position", "joint": "LeftHand", "t": 120.0, "weight": 1.5, "target": [35.0, 95.0, 20.0]},
            {"type": "joint_position", "joint": "RightHand", "t": 120.0, "weight": 1.5, "target": [-25.0, 95.0, 20.0]},
            {"type": "joint_orientation", "joint": "Head", "t": 33.25, "weight": 2.0, "orientation": [0.9, 0.1, -0.3, 0.2]},
            {"type": "joint_orientation", "joint": "Hips", "t": 155.0, "weight": 1.0, "orientation": [0.7, 0.0, 0.7, 0.0]},
            {"type": "joint_midpoint", "joint": "Hips", "joint2": "LeftToeBase" if "LeftToeBase" in sk.names else "LeftFoot", "t": 0.0,
             "weight": 1.0, "target": [0.0, None, 0.0]}]
    cset = _capi.ConstraintSet(prim, cons, sk)
    res = prim.score_constraint_residuals(cset, S)
    np.testing.assert_allclose(res, op.skeleton_residuals(S, cons, joints, animated), rtol=1e-9, atol=1e-8)
    ctx.set_option(_capi.MG_OPT_FORCE_VALU_SCORE, 1)
    np.testing.assert_array_equal(prim.score_constraint_residuals(cset, S), res)
    ctx.set_option(_capi.MG_OPT_FORCE_VALU_SCORE, 0)
    cset.close()
    prev = op.back_project_frames(rng.standard_normal(40))[-1].copy()
    prev[:3] = [-60.0, 90.0, 210.0]
    cset = _capi.ConstraintSet(prim, cons, sk, alignment=sk.alignment_to(prev, "Hips"))
    res_al = prim.score_constraint_residuals(cset, S)
    np.testing.assert_allclose(res_al, op.aligned_residuals(S, cons, prev, joints, animated, "Hips"), rtol=1e-9, atol=1e-8)
    assert np.abs(res_al - res).max() > 1.0                                       # the alignment does something
    cset.close()
    # the root's orientation needs no skeleton
    cset = _capi.ConstraintSet(prim, [dict(cons[4], joint=0)])
    np.testing.assert_allclose(prim.score_constraint_residuals(cset, S)[:, 0], res[:, 4], rtol=1e-13, atol=1e-13)
    cset.close()
    with pytest.raises(ValueError):
        _capi.ConstraintSet(prim, [dict(cons[3], joint=3)])                        # a chain without a skeleton
    with pytest.raises(ValueError):
        _capi.ConstraintSet(prim, [cons[0]])
    prim.close()

    # known answers: every control point is the identity pose at (1, 2, 3)
    tiny = synthetic.make_primitive(seed=2, n_components=3, n_frames=12, n_basis=7, n_dim=79, n_gmm=2, name="tiny79")
    pose = np.zeros(79)
    pose[3::4][:19] = 1.0
    pose[:3] = [1.0, 2.0, 3.0]
    model = dict(tiny)
    model["mean_spatial_vector"] = np.tile(pose, int(tiny["n_basis_spatial"]))
    model["eigen_vectors_spatial"] = np.zeros_like(np.asarray(tiny["eigen_vectors_spatial"], dtype=np.float64))
    model["translation_maxima"] = np.ones(3)
    pr = _capi.Primitive(ctx, model)
    half = np.sqrt(0.5)
    known = [({"type": "joint_midpoint", "joint": "LeftHand_EndSite", "joint2": "RightHand_EndSite", "t": 4.0, "weight": 1.0,
               "target": [1.0, 35.0, 0.5]}, 3.0),
             ({"type": "joint_orientation", "joint": "LeftHand", "t": 4.0, "weight": 1.0, "orientation": [half, 0.0, half, 0.0]}, np.pi / 2),
             ({"type": "joint_orientation", "joint": "Hips", "t": 4.0, "weight": 2.0, "orientation": [1.0, 0.0, 0.0, 0.0]}, 0.0)]
    for c, expect in known:
        cs = _capi.ConstraintSet(pr, [c], sk)
        np.testing.assert_allclose(pr.score_constraints(cs, np.zeros((2, pr.n_components))), expect, rtol=1e-12, atol=1e-12, err_msg=str(c))
        cs.close()
    pr.close()


def test_relative_point_and_look_at_constraints(ctx):
    """A point given in a joint's own frame (RelativeTransformConstraint, reference relative_transform_constraint.py:
    46-50) and LookAtConstraint (look_at_constraint.py:55-66), local and aligned, against the matrix oracle, plus
    answers known by hand on a constant identity pose."""
    from oracle import mg_oracle as orc
    joints, animated = synthetic.make_skeleton()
    sk = _capi.Skeleton(joints, animated)
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    op = orc.OraclePrimitive(data)
    rng = np.random.default_rng(19)
    S = rng.standard_normal((50, 40))
    cons = [{"type": "joint_position", "joint": "RightHand", "offset": [0.0, -3.0, 12.0], "t": 60.0, "weight": 1.0, "target": [-30.0, 90.0, 25.0]},
            {"type": "joint_position", "joint": "Hips", "offset": [5.0, 0.0, 0.0], "t": 155.0, "weight": 0.5, "target": [10.0, None, 0.0]},
            {"type": "look_at", "joint": "Head", "t": 100.5, "weight": 2.0, "target": [50.0, 150.0, 400.0]},
            {"type": "look_at", "joint": "Hips", "t": 0.0, "weight": 1.0, "target": [-100.0, 0.0, 30.0]},
            {"type": "joint_position", "joint": "RightHand", "t": 60.0, "weight": 1.0, "target": [-30.0, 90.0, 25.0]}]
    cset = _capi.ConstraintSet(prim, cons, sk)
    res = prim.score_constraint_residuals(cset, S)
    np.testing.assert_allclose(res, op.skeleton_residuals(S, cons, joints, animated), rtol=1e-9, atol=1e-8)
    assert np.abs(res[:, 0] - res[:, 4]).max() > 1.0                                  # the offset matters
    ctx.set_option(_capi.MG_OPT_FORCE_VALU_SCORE, 1)
    np.testing.assert_array_equal(prim.score_constraint_residuals(cset, S), res)
    ctx.set_option(_capi.MG_OPT_FORCE_VALU_SCORE, 0)
    cset.close()
    prev = op.back_project_frames(rng.standard_normal(40))[-1].copy()
    prev[:3] = [15.0, 90.0, -75.0]
    cset = _capi.ConstraintSet(prim, cons, sk, alignment=sk.alignment_to(prev, "Hips"))
    np.testing.assert_allclose(prim.score_constraint_residuals(cset, S), op.aligned_residuals(S, cons, prev, joints, animated, "Hips"),
                               rtol=1e-9, atol=1e-8)
    cset.close()
    prim.close()

    tiny = synthetic.make_primitive(seed=2, n_components=3, n_frames=12, n_basis=7, n_dim=79, n_gmm=2, name="tiny79")
    pose = np.zeros(79)
    pose[3::4][:19] = 1.0
    pose[:3] = [1.0, 2.0, 3.0]
    half = np.sqrt(0.5)
    pose[3:7] = [half, 0.0, half, 0.0]                                                 # hips a quarter turn about y: z -> x, x -> -z
    model = dict(tiny)
    model["mean_spatial_vector"] = np.tile(pose, int(tiny["n_basis_spatial"]))
    model["eigen_vectors_spatial"] = np.zeros_like(np.asarray(tiny["eigen_vectors_spatial"], dtype=np.float64))
    model["translation_maxima"] = np.ones(3)
    pr = _capi.Primitive(ctx, model)
    known = [({"type": "joint_position", "joint": "Hips", "offset": [0.0, 0.0, 5.0], "t": 3.0, "weight": 1.0, "target": [6.0, 2.0, 3.0]}, 0.0),
             ({"type": "joint_position", "joint": "Hips", "offset": [2.0, 0.0, 0.0], "t": 3.0, "weight": 1.0, "target": [1.0, 2.0, 3.0]}, 2.0),
             ({"type": "look_at", "joint": "Hips", "t": 3.0, "weight": 1.0, "target": [11.0, 2.0, 3.0]}, 0.0),     # looks along +x
             ({"type": "look_at", "joint": "Hips", "t": 3.0, "weight": 1.0, "target": [1.0, 2.0, 9.0]}, np.pi / 2),
             ({"type": "look_at", "joint": "Head", "t": 3.0, "weight": 3.0, "target": [1.0 - 50.0, 2.0 + 44.0 + 0.0, 3.0]}, 3.0 * np.pi)]
    for c, expect in known:
        cs = _capi.ConstraintSet(pr, [c], sk)
        np.testing.assert_allclose(pr.score_constraints(cs, np.zeros((2, pr.n_components))), expect, rtol=1e-12, atol=1e-7, err_msg=str(c))
        cs.close()
    pr.close()


def test_constraint_set_update_keeps_the_structure_and_swaps_the_values(ctx):
    """mg_constraint_set_update: new targets / weights / previous frame for a set of the same structure, stream
    ordered and without a new allocation.  Results must be those of a freshly built set, bit for bit; a launch
    enqueued before the update keeps the old values; a different structure is refused; more than 60 constraints
    take the copying path."""
    from morphablegraphs_amd.candidate_scoring import cached_constraint_set, clear_constraint_cache
    joints, animated = synthetic.make_skeleton()
    sk = _capi.Skeleton(joints, animated)
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    rng = np.random.default_rng(23)
    S = rng.standard_normal((333, 40))
    prev_a, prev_b = np.zeros(79), np.zeros(79)
    prev_a[3::4][:19], prev_b[3::4][:19] = 1.0, 1.0
    prev_a[:7] = [10.0, 90.0, 20.0, 0.9, 0.0, 0.4, 0.0]
    prev_b[:7] = [-40.0, 91.0, 5.0, 0.3, 0.1, -0.9, 0.0]

    def cons(k):
        return [{"type": "position", "t": 155.0, "weight": 1.0 + k, "target": [40.0 + 7 * k, None, -30.0 - k]},
                {"type": "direction", "t": 155.0, "weight": 0.5, "target": [0.5 + k, 1.0]},
                {"type": "joint_position", "joint": "LeftHand", "t": 80.0, "weight": 1.0, "target": [30.0, 95.0 + k, 10.0]},
                {"type": "joint_orientation", "joint": "Head", "t": 33.0, "weight": 2.0, "orientation": [0.9, 0.1 * k, -0.3, 0.2]},
                {"type": "look_at", "joint": "Head", "t": 50.0, "weight": 1.0, "target": [50.0 * k, 150.0, 400.0]}]
    fresh_a = _capi.ConstraintSet(prim, cons(0), sk, alignment=sk.alignment_to(prev_a, "Hips"))
    fresh_b = _capi.ConstraintSet(prim, cons(1), sk, alignment=sk.alignment_to(prev_b, "Hips"))
    ra, rb = prim.score_constraint_residuals(fresh_a, S), prim.score_constraint_residuals(fresh_b, S)
    assert np.abs(ra - rb).max() > 1.0
    cs = _capi.ConstraintSet(prim, cons(0), sk, alignment=sk.alignment_to(prev_a, "Hips"))
    # enqueue a scoring with the old values, update, enqueue one with the new values, only then read both back
    d_S, d_e0, d_e1 = ctx.upload(S), ctx.malloc(len(S) * 8), ctx.malloc(len(S) * 8)
    prim.score_constraints_dev(cs, d_S, np.float64, len(S), 40, d_e0, np.float64)
    cs.update(cons(1), sk.alignment_to(prev_b, "Hips"))
    prim.score_constraints_dev(cs, d_S, np.float64, len(S), 40, d_e1, np.float64)
    np.testing.assert_array_equal(ctx.download(d_e0, (len(S),), np.float64), prim.score_constraints(fresh_a, S))
    np.testing.assert_array_equal(ctx.download(d_e1, (len(S),), np.float64), prim.score_constraints(fresh_b, S))
    np.testing.assert_array_equal(prim.score_constraint_residuals(cs, S), rb)
    cs.update(cons(0), sk.alignment_to(prev_a, "Hips"))
    np.testing.assert_array_equal(prim.score_constraint_residuals(cs, S), ra)
    for bad, al in ((cons(0)[:4], sk.alignment_to(prev_a, "Hips")),                                  # fewer constraints
                    ([dict(cons(0)[0], t=154.0)] + cons(0)[1:], sk.alignment_to(prev_a, "Hips")),    # another keyframe
                    (cons(0)[:2] + [dict(cons(0)[2], joint="RightHand")] + cons(0)[3:], sk.alignment_to(prev_a, "Hips")),
                    (cons(0)[:2] + [dict(cons(0)[2], offset=[0.0, 1.0, 0.0])] + cons(0)[3:], sk.alignment_to(prev_a, "Hips")),
                    (cons(0), None),                                                                 # alignment dropped
                    (cons(0), sk.alignment_to(prev_a, "Spine"))):                                    # another aligning joint
        with pytest.raises(_capi.MGError):
            cs.update(bad, al)
    np.testing.assert_array_equal(prim.score_constraint_residuals(cs, S), ra)                         # refused updates change nothing
    for c in (cs, fresh_a, fresh_b):
        c.close()

    # the cache hands out ONE set per structure and rewrites its values
    clear_constraint_cache()
    c0 = cached_constraint_set(prim, cons(0), sk, sk.alignment_to(prev_a, "Hips"))
    r0 = prim.score_constraint_residuals(c0, S)
    c1 = cached_constraint_set(prim, cons(1), sk, sk.alignment_to(prev_b, "Hips"))
    assert c1 is c0
    np.testing.assert_array_equal(prim.score_constraint_residuals(c1, S), rb)
    np.testing.assert_array_equal(r0, ra)
    c2 = cached_constraint_set(prim, cons(1), sk, None)                                               # local mode: another structure
    assert c2 is not c0
    clear_constraint_cache()

    # 70 constraints: the values no longer fit kernel arguments
    many = [{"type": "position", "t": float(i), "weight": 1.0, "target": [float(i), None, 1.0]} for i in range(70)]
    many2 = [dict(c, target=[c["target"][0] + 3.0, 2.0, None]) for c in many]
    cs = _capi.ConstraintSet(prim, many)
    cs.update(many2)
    ref = _capi.ConstraintSet(prim, many2)
    np.testing.assert_array_equal(prim.score_constraint_residuals(cs, S[:40]), prim.score_constraint_residuals(ref, S[:40]))
    cs.close()
    ref.close()
    prim.close()


def test_c_abi_rejects_misuse_with_status_codes_not_crashes(ctx):
    """Every entry point validates what it is given and answers with a negative status and a message
    (mg_last_error), never a fault: NULL handles and pointers, unknown dtype / path codes, negative sizes, a time
    grid or constraint set that belongs to another primitive, non-finite constraint data."""
    lib = ctx.lib
    vp = C.c_void_p
    prim_a = _capi.Primitive(ctx, synthetic.make_tiny_primitive())
    prim_b = _capi.Primitive(ctx, synthetic.make_walk_primitive(seed=0))
    S = np.zeros((4, 40), dtype=np.float32)
    d_S, d_out = ctx.upload(S), ctx.malloc(4 * 156 * 79 * 4)
    null = vp(0)

    def bad(rc, what):
        assert rc < 0, what
        assert len(lib.mg_last_error()) > 0, what
    bad(lib.mg_back_project_frames(null, null, d_S.ptr, 0, 4, 40, d_out.ptr, 0), "NULL primitive")
    bad(lib.mg_back_project_frames(prim_b.handle, null, null, 0, 4, 40, d_out.ptr, 0), "NULL latents")
    bad(lib.mg_back_project_frames(prim_b.handle, null, d_S.ptr, 7, 4, 40, d_out.ptr, 0), "unknown dtype")
    bad(lib.mg_back_project_frames(prim_b.handle, null, d_S.ptr, 0, -1, 40, d_out.ptr, 0), "negative batch")
    bad(lib.mg_back_project_frames(prim_b.handle, null, d_S.ptr, 0, 4, 39, d_out.ptr, 0), "ld < n_components")
    bad(lib.mg_back_project_frames(prim_b.handle, null, d_S.ptr, 0, 4, 40, d_out.ptr, 9), "unknown path")
    bad(lib.mg_back_project_frames(prim_b.handle, null, d_S.ptr, 0, 4, 40, null, 0), "NULL output")
    bad(lib.mg_back_project_frames(prim_b.handle, prim_a.canonical_grid.handle, d_S.ptr, 0, 4, 40, d_out.ptr, 0), "foreign grid")
    bad(lib.mg_gmm_log_prob(prim_b.handle, d_S.ptr, 0, 4, 40, null, 0), "NULL log p output")
    bad(lib.mg_gmm_log_prob(prim_b.handle, d_S.ptr, 0, 4, 40, d_out.ptr, 5), "unknown output dtype")
    bad(lib.mg_step_frames_and_logp(null, d_S.ptr, 0, 4, 40, d_out.ptr, d_out.ptr), "NULL primitive (fused step)")
    cs_a = _capi.ConstraintSet(prim_a, [{"type": "position", "t": 1.0, "weight": 1.0, "target": [0.0, 0.0, 0.0]}])
    bad(lib.mg_score_constraints(prim_b.handle, cs_a.handle, d_S.ptr, 0, 4, 40, d_out.ptr, 1), "foreign constraint set")
    bad(lib.mg_score_constraints(prim_b.handle, null, d_S.ptr, 0, 4, 40, d_out.ptr, 1), "NULL constraint set")
    bad(lib.mg_constraint_set_update(null, null, 0, null), "NULL set update")
    bad(lib.mg_argmin_first_dev(ctx.handle, d_out.ptr, 3, 4, d_out.ptr), "argmin dtype")
    bad(lib.mg_argmin_first_dev(ctx.handle, null, 0, 4, d_out.ptr), "argmin NULL values")
    bad(lib.mg_context_set_reserved_cus(ctx.handle, -1), "negative reserved CUs")
    bad(lib.mg_context_arena_begin(null, 0), "NULL context")
    h = vp()
    bad(lib.mg_time_grid_create(prim_b.handle, null, 5, C.byref(h)), "NULL times")
    t = np.array([0.0, np.nan, 3.0])
    bad(lib.mg_time_grid_create(prim_b.handle, t.ctypes.data_as(vp), 3, C.byref(h)), "NaN sample time")
    for kw in ({"t": float("inf")}, {"type": "joint_orientation", "orientation": [1, 0, 0, 0], "ref_dir": (0.0, 0.0, 0.0)}):
        c = {"type": "position", "t": 1.0, "weight": 1.0, "target": [0.0, 0.0, 0.0]}
        c.update(kw)
        with pytest.raises(_capi.MGError):
            _capi.ConstraintSet(prim_b, [c])
    # round-2 entry points
    i4, d4 = (C.c_int32 * 4)(), (C.c_double * 4)()
    bad(lib.mg_step_plan(null, 8192, i4), "NULL primitive (step plan)")
    bad(lib.mg_step_plan(prim_b.handle, -1, i4), "negative batch (step plan)")
    bad(lib.mg_context_set_option(ctx.handle, _capi.MG_OPT_COUNT, 1), "unknown option")
    bad(lib.mg_device_probe_placement(ctx.handle, d_out.ptr, 4 * 156 * 79 * 4, d4), "probe of a small buffer")
    bad(lib.mg_device_malloc_placed(null, 1 << 28, 0, C.byref(h), d4), "NULL context (placed malloc)")
    bad(lib.mg_time_function_canonical(prim_b.handle, d_S.ptr, 0, 4, 40, d_out.ptr), "time function of a primitive without a time model")
    bad(lib.mg_joint_positions(ctx.handle, null, null, 1, d_out.ptr, 4, 79, d_out.ptr), "NULL skeleton")
    cps = np.zeros((1, 3))
    bad(lib.mg_trajectory_create(prim_b.handle, cps.ctypes.data_as(vp), 1, 1000, C.byref(h)), "trajectory of one control point")
    cps = np.array([[0.0, 0.0, 0.0], [np.nan, 0.0, 1.0]])
    bad(lib.mg_trajectory_create(prim_b.handle, cps.ctypes.data_as(vp), 2, 1000, C.byref(h)), "NaN control point")
    bad(lib.mg_score_trajectory(prim_b.handle, null, null, d_S.ptr, 0, 4, 40, 0.0, 1.0, null, d_out.ptr, 0, null), "NULL trajectory")
    cs_b = _capi.ConstraintSet(prim_b, [{"type": "position", "t": 1.0, "weight": 1.0, "target": [0.0, 0.0, 0.0]}])
    one = lambda v: (vp * 1)(v)   # noqa: E731
    cnt = np.array([4, 0, 0, 0, 0, 0, 0, 0], dtype=np.int64)
    seeds, lds = (C.c_uint64 * 1)(1), (C.c_int64 * 1)(40)
    ok_args = (1, one(prim_b.handle), one(cs_b.handle), 4, one(cnt.ctypes.data), seeds, one(d_S.ptr.value), 0, lds, one(d_out.ptr.value), d_out.ptr)
    bad(lib.mg_options_step(*ok_args, 16 + 8 * 40 - 8, null), "result stride too small")
    bad(lib.mg_options_step(0, *ok_args[1:], 16 + 8 * 40, null), "no options")
    bad(lib.mg_options_step(1, one(None), *ok_args[2:], 16 + 8 * 40, null), "NULL option")
    # and the library still works afterwards
    assert prim_b.back_project_frames(S).shape == (4, 156, 79)
    cs_b.close()
    cs_a.close()
    prim_a.close()
    prim_b.close()


def test_rccl_entry_points_of_the_c_abi_on_one_rank():
    """mg_dist_*: RCCL loaded on first use, a communicator of one rank on the context's device, the all-gather of
    the scores on the context's stream (with one rank: gathered == local).  More ranks need more GPUs than the
    test box has; the exchange logic above the collective is covered by the 2-rank gloo test."""
    ctx = _capi.Context(0)
    with pytest.raises(_capi.MGError):
        ctx.dist_all_gather(ctx.malloc(16), ctx.malloc(16), 4)             # no communicator yet
    uid = ctx.dist_unique_id()
    assert len(uid) == 128 and any(uid)
    ctx.dist_init(0, 1, uid)
    with pytest.raises(_capi.MGError):
        ctx.dist_init(0, 1, uid)                                           # one communicator per context
    for dtype in (np.float32, np.float64):
        local = np.random.default_rng(0).standard_normal(8192).astype(dtype)
        d_l, d_g = ctx.upload(local), ctx.malloc(local.nbytes)
        ctx.dist_all_gather(d_l, d_g, len(local), dtype)
        np.testing.assert_array_equal(ctx.download(d_g, (len(local),), dtype), local)
    # the torch-free sharded argmin on top of it (distributed.mg_sharded_best_candidate), one rank
    from morphablegraphs_amd import distributed
    S = np.random.default_rng(1).standard_normal((37, 8))
    S[20] = S[3]
    score = lambda blk: np.abs(blk).sum(axis=1)
    idx, val, scores = distributed.mg_sharded_best_candidate(ctx, S, score, 0, 1)
    np.testing.assert_array_equal(scores, score(S))
    assert idx == int(np.argmin(score(S))) and val == score(S).min()
    ctx.dist_finalize()
    ctx.dist_finalize()                                                    # idempotent
    ctx.close()


def test_point_cloud_pose_constraint(ctx):
    """MG_CONSTRAINT_POSE (PoseConstraint.evaluate_motion_spline, reference pose_constraint.py:48-67; the fit and the
    distance are anim_utils', absent: PARITY UNPINNED): forward kinematics of every listed joint, the optimal weighted
    2-D fit onto the wanted cloud, mean distance, velocity of the first joint -- against the NumPy oracle in local
    and global coordinates, MFMA and VALU kernels bit for bit, mixed with other constraint types, and a known answer
    (a wanted cloud that IS the pose, turned and shifted: error 0)."""
    from oracle import mg_oracle as orc
    joints, animated = synthetic.make_skeleton()
    sk = _capi.Skeleton(joints, animated)
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    op = orc.OraclePrimitive(data)
    rng = np.random.default_rng(29)
    S = rng.standard_normal((40, 40))
    names = ["Hips", "Spine1", "Head", "LeftArm", "LeftHand", "RightArm", "RightHand", "LeftLeg", "LeftFoot", "RightLeg", "RightFoot",
             "LeftHand_EndSite", "Head_EndSite"]
    ref_frame = op.back_project_frames(rng.standard_normal(40))[70]
    wanted = np.array([orc.joint_global_position(ref_frame, joints, animated, j) for j in names]) + rng.standard_normal((len(names), 3))
    weights = rng.uniform(0.5, 2.0, len(names))
    cons = [{"type": "position", "t": 155.0, "weight": 1.0, "target": [40.0, None, -30.0]},
            {"type": "pose", "t": 0.0, "weight": 0.7, "joints": names, "points": wanted, "weights": weights, "velocity": [0.3, 0.0, 1.2]},
            {"type": "joint_position", "joint": "LeftHand", "t": 80.0, "weight": 1.0, "target": [30.0, 95.0, 10.0]},
            {"type": "pose", "t": 155.0, "weight": 1.0, "joints": names[:5], "points": wanted[:5], "weights": weights[:5], "velocity": None}]
    cset = _capi.ConstraintSet(prim, cons, sk)
    res = prim.score_constraint_residuals(cset, S)
    np.testing.assert_allclose(res, op.skeleton_residuals(S, cons, joints, animated), rtol=1e-9, atol=1e-8)
    np.testing.assert_allclose(prim.score_constraints(cset, S), res.sum(axis=1), rtol=1e-13, atol=1e-12)
    ctx.set_option(_capi.MG_OPT_FORCE_VALU_SCORE, 1)
    np.testing.assert_array_equal(prim.score_constraint_residuals(cset, S), res)
    ctx.set_option(_capi.MG_OPT_FORCE_VALU_SCORE, 0)
    with pytest.raises(_capi.MGError):
        cset.update(cons)                                              # sets with poses are rebuilt, not updated
    cset.close()
    prev = op.back_project_frames(rng.standard_normal(40))[-1].copy()
    prev[:3] = [-20.0, 90.0, 45.0]
    cset = _capi.ConstraintSet(prim, cons, sk, alignment=sk.alignment_to(prev, "Hips"))
    res_al = prim.score_constraint_residuals(cset, S)
    np.testing.assert_allclose(res_al, op.aligned_residuals(S, cons, prev, joints, animated, "Hips"), rtol=1e-9, atol=1e-8)
    # the fit absorbs the alignment: a pose error without a velocity term is the same in both coordinate systems
    np.testing.assert_allclose(res_al[:, 3], res[:, 3], rtol=1e-9, atol=1e-9)
    cset.close()
    with pytest.raises(ValueError):
        _capi.ConstraintSet(prim, cons[1:2])                           # no skeleton
    with pytest.raises(_capi.MGError):
        _capi.ConstraintSet(prim, [dict(cons[1], weights=np.zeros(len(names)))], sk)
    prim.close()

    tiny = synthetic.make_primitive(seed=2, n_components=3, n_frames=12, n_basis=7, n_dim=79, n_gmm=2, name="tiny79")
    pose = np.zeros(79)
    pose[3::4][:19] = 1.0
    pose[:3] = [1.0, 2.0, 3.0]
    model = dict(tiny)
    model["mean_spatial_vector"] = np.tile(pose, int(tiny["n_basis_spatial"]))
    model["eigen_vectors_spatial"] = np.zeros_like(np.asarray(tiny["eigen_vectors_spatial"], dtype=np.float64))
    model["translation_maxima"] = np.ones(3)
    pr = _capi.Primitive(ctx, model)
    cloud = np.array([orc.joint_global_position(pose, joints, animated, j) for j in names])
    turned = orc.transform_point_cloud(cloud, -0.9, 25.0, 4.0)
    cs = _capi.ConstraintSet(pr, [{"type": "pose", "t": 5.0, "weight": 2.0, "joints": names, "points": turned, "weights": weights,
                                   "velocity": [0.0, 0.0, 0.0]}], sk)
    np.testing.assert_allclose(pr.score_constraints(cs, np.zeros((2, pr.n_components))), 0.0, atol=1e-9)
    cs.close()
    lifted = turned + [0.0, 3.0, 0.0]                                   # y is not fitted: every point 3 off -> mean distance 3
    cs = _capi.ConstraintSet(pr, [{"type": "pose", "t": 5.0, "weight": 2.0, "joints": names, "points": lifted, "weights": weights}], sk)
    np.testing.assert_allclose(pr.score_constraints(cs, np.zeros((2, pr.n_components))), 6.0, rtol=1e-10)
    cs.close()
    pr.close()


def test_frames_beyond_four_gigabytes_of_output(ctx):
    """262 144 candidates = 12.9 GB of frames: output offsets pass 2^32 bytes (candidate 87 127) and 2^33, the unit
    index passes 2^16 tiles.  Spot rows around those boundaries and at the very end against the float32 contract
    model, bit for bit, for the stand-alone kernel and the fused step (two-launch fallback at this size)."""
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    cp = c_oracle.COraclePrimitive(data)
    B, L, F, D = 262144, 40, 156, 79
    rng = np.random.default_rng(31)
    S = rng.standard_normal((B, L)).astype(np.float32)
    d_S = ctx.upload(S)
    d_f = ctx.malloc(B * F * D * 4)
    d_l = ctx.malloc(B * 4)
    spots = [0, 15, 16, 87126, 87127, 87128, 174254, 174255, 174256, 200000, B - 17, B - 16, B - 1]
    model = cp.frames_f32model(S[spots].astype(np.float64))
    for fused in (False, True):
        _capi._check(ctx.lib.mg_memset(ctx.handle, d_f.ptr, 0xFF, B * F * D * 4))     # NaN pattern: stale rows cannot pass
        if fused:
            prim.step_frames_and_logp_dev(d_S, np.float32, B, L, d_f, d_l)
        else:
            prim.back_project_frames_dev(d_S, np.float32, B, L, d_f, path=_capi.MG_PATH_MFMA)
        for k, b in enumerate(spots):
            row = ctx.download(d_f.ptr.value + b * F * D * 4, (F, D), np.float32)
            np.testing.assert_array_equal(_bits(row), _bits(model[k]), err_msg="candidate %d fused=%s" % (b, fused))
    lp = ctx.download(d_l, (B,), np.float32)
    ref = cp.log_prob_f64(S[spots].astype(np.float64))
    np.testing.assert_allclose(lp[spots], ref, rtol=3e-7, atol=1e-5)
    assert np.isfinite(lp).all()
    d_f.free(); d_l.free(); d_S.free()
    prim.close()


def test_chunked_device_allocation(ctx):
    """mg_device_malloc_chunked: a buffer assembled from separate physical chunks behaves like any other device
    buffer -- the frames kernel writes the same bits into it, copies work across chunk boundaries, and
    mg_device_free releases it."""
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    B = 300
    S = np.random.default_rng(41).standard_normal((B, 40)).astype(np.float32)
    ref = prim.back_project_frames(S, path=_capi.MG_PATH_MFMA)
    d_S = ctx.upload(S)
    for chunk in (1 << 20, 8 << 20):
        buf = ctx.malloc(ref.nbytes, chunk_bytes=chunk)                # 14.8 MB: 15 chunks of 1 MiB / 2 of 8 MiB
        prim.back_project_frames_dev(d_S, np.float32, B, 40, buf, path=_capi.MG_PATH_MFMA)
        got = ctx.download(buf, ref.shape, np.float32)
        np.testing.assert_array_equal(_bits(got), _bits(ref))
        buf.free()
    with pytest.raises(_capi.MGError):
        ctx.malloc(0, chunk_bytes=1 << 20)
    prim.close()


@pytest.mark.parametrize("case", ["walk_seed0", "odd_shape", "k1_near_singular"])
def test_lds_resident_mixture_kernel_is_bit_identical(ctx, case):
    """mg_gmm_log_prob's large-batch kernel (persistent workgroups, the mixture's fragments staged in LDS once, a wave per
    16-candidate tile) against the one-tile-per-workgroup kernel: the same log p(x), bit for bit, for ragged batches and every
    input / output type -- and so within 1e-9 of sklearn's score_samples on the golden rows."""
    from conftest import golden_model
    data, g = golden_model(case)
    prim = _capi.Primitive(ctx, data)
    L = prim.n_gmm_dims
    rng = np.random.default_rng(11)
    for B in (1, 17, 4099, 40000):
        X = rng.standard_normal((B, L))
        for xdt in (np.float32, np.float64):
            for odt in (np.float32, np.float64):
                d_x, d_o = ctx.upload(X.astype(xdt)), ctx.malloc(B * np.dtype(odt).itemsize)
                res = []
                for mode in (1, 2):
                    ctx.set_option(_capi.MG_OPT_GMM_KERNEL, mode)
                    _capi._check(ctx.lib.mg_memset(ctx.handle, d_o.ptr, 0xff, B * np.dtype(odt).itemsize))
                    prim.gmm_log_prob_dev(d_x, xdt, B, L, d_o, odt)
                    res.append(ctx.download(d_o, (B,), odt))
                ctx.set_option(_capi.MG_OPT_GMM_KERNEL, 0)
                np.testing.assert_array_equal(res[0].view(np.uint8), res[1].view(np.uint8))
                d_x.free()
                d_o.free()
    ctx.set_option(_capi.MG_OPT_GMM_KERNEL, 2)
    np.testing.assert_allclose(prim.gmm_log_prob(g["X"], dtype=np.float64), g["logp"], rtol=1e-9, atol=1e-7)
    ctx.set_option(_capi.MG_OPT_GMM_KERNEL, 0)
    prim.close()


# ---------------------------------------------------------------------------------------------------------------------
# The root channels' two modes (include/mg_hip.h, mg_primitive_root_mode): float64 pipeline and mean/delta split
# ---------------------------------------------------------------------------------------------------------------------
def test_root_mode_gate_agrees_with_the_oracle(ctx, golden_case):
    """The accuracy gate of the optional mean/delta split is part of the float32 contract: under MG_OPT_ROOT_MODE 3 the library
    and the oracle's restatement of the gate pick the same mode for every golden primitive; the default is the float64 pipeline."""
    name, data, g = golden_case
    prim = _capi.Primitive(ctx, data)
    cp = c_oracle.COraclePrimitive(data)
    np.testing.assert_allclose(prim.root_split_estimate, cp.root_split_estimate, rtol=1e-12)
    assert prim.root_split is False
    ctx.set_option(_capi.MG_OPT_ROOT_MODE, 3)
    assert prim.root_split == cp.root_split, name
    ctx.set_option(_capi.MG_OPT_ROOT_MODE, 1)
    assert prim.root_split is False
    ctx.set_option(_capi.MG_OPT_ROOT_MODE, 2)
    assert prim.root_split is True
    prim.close()


@pytest.mark.parametrize("mode", [1, 2])
def test_root_modes_bit_exact_on_every_kernel(ctx, golden_case, mode):
    """Both modes forced on every golden primitive: the direct kernel, the tile-major and the chunk-stationary kernel (where
    they cover the shape), float64 and float32 latents, canonical grid and an evaluation grid -- bit for bit the oracle's model
    of that mode; the float64 pipeline also within the north-star tolerance of the reference's frames whatever the primitive
    (the split only where the gate allows it: test_root_split_where_the_gate_allows_it)."""
    name, data, g = golden_case
    ctx.set_option(_capi.MG_OPT_ROOT_MODE, mode)
    prim = _capi.Primitive(ctx, data)
    cp = c_oracle.COraclePrimitive(data)
    split = mode == 2
    assert prim.root_split == split
    rng = np.random.default_rng(41)
    L = g["S"].shape[1]
    S = np.concatenate([g["S"], rng.standard_normal((300, L))]).astype(np.float64)
    model = cp.frames_f32model(S, root_split=split)
    got = prim.back_project_frames(S, path=_capi.MG_PATH_DIRECT)
    np.testing.assert_array_equal(_bits(got), _bits(model), err_msg="%s direct" % name)
    ref = g["frames"]
    if not split:
        assert np.all(np.abs(got[:len(ref)].astype(np.float64) - ref) <= pose_tol(ref))
    if prim.mfma_supported:
        S32 = S.astype(np.float32)
        model32 = cp.frames_f32model(S32.astype(np.float64), root_split=split)
        times = np.array([0.0, 0.25, (prim.n_canonical_frames - 1) / 2.0, prim.n_canonical_frames - 1.0, prim.n_canonical_frames + 2.0, -1.0])
        grid = prim.time_grid(times)
        model_t = cp.frames_f32model(S, tp=times, root_split=split)
        for kern in (1, 2):
            _set_frames_kernel(ctx, kern)
            try:
                got = prim.back_project_frames(S, path=_capi.MG_PATH_MFMA)
            except _capi.MGError as e:
                assert kern == 2 and e.status == -4, e   # MG_ERR_UNSUPPORTED: the window does not fit the registers
                continue
            np.testing.assert_array_equal(_bits(got), _bits(model), err_msg="%s kernel %d" % (name, kern))
            np.testing.assert_array_equal(_bits(prim.back_project_frames(S32, path=_capi.MG_PATH_MFMA)), _bits(model32))
            try:
                got_t = prim.back_project_frames(S, grid=grid, path=_capi.MG_PATH_MFMA)
            except _capi.MGError as e:
                assert kern == 2 and e.status == -4, e
                continue
            np.testing.assert_array_equal(_bits(got_t), _bits(model_t), err_msg="%s kernel %d, evaluation grid" % (name, kern))
        grid.close()
    prim.close()


@pytest.mark.parametrize("B", [17, 1000, 8192 + 5])
def test_root_split_where_the_gate_allows_it(ctx, B):
    """A 'walk'-sized primitive whose root translation varies little between candidates (root rows of the eigenvectors not
    scaled up): under MG_OPT_ROOT_MODE 3 the gate picks the split; results are bit for bit the oracle's model, within the north-star
    tolerance of the float64 frames (the reference's arithmetic, device float64 kernel pinned by the golden tests), the same
    from all three kernels and from the fused step."""
    data = synthetic.make_walk_primitive(seed=3, realistic=False)
    m = np.array(data["mean_spatial_vector"]).reshape(-1, 79)
    m[:, :3] *= 100.0                       # root translation of realistic size, its variation small
    data["mean_spatial_vector"] = m.reshape(-1).tolist()
    prim = _capi.Primitive(ctx, data)
    cp = c_oracle.COraclePrimitive(data)
    assert prim.root_split is False
    ctx.set_option(_capi.MG_OPT_ROOT_MODE, 3)
    assert cp.root_split and prim.root_split and prim.root_split_estimate <= 5e-6
    rng = np.random.default_rng(B)
    for dtype in (np.float32, np.float64):
        S = (2.2 * rng.standard_normal((B, 40))).astype(dtype)   # the spread of the primitive's mixture (second moment ~5)
        n_model = min(B, 300)
        model = cp.frames_f32model(S[:n_model].astype(np.float64), root_split=True)
        ref = prim.back_project_frames_f64(S[:n_model].astype(np.float64))
        direct = prim.back_project_frames(S[:n_model], path=_capi.MG_PATH_DIRECT)
        np.testing.assert_array_equal(_bits(direct), _bits(model))
        err = np.abs(model.astype(np.float64) - ref)
        assert np.all(err <= pose_tol(ref)), float((err / pose_tol(ref)).max())
        outs = []
        for kern in (1, 2):
            _set_frames_kernel(ctx, kern)
            outs.append(prim.back_project_frames(S, path=_capi.MG_PATH_MFMA))
            np.testing.assert_array_equal(_bits(outs[-1][:n_model]), _bits(model), err_msg="kernel %d B=%d" % (kern, B))
        np.testing.assert_array_equal(_bits(outs[0]), _bits(outs[1]))
        _set_frames_kernel(ctx, 0)
        frames, logp = _fused_step(ctx, prim, S, 156, 79)
        np.testing.assert_array_equal(_bits(frames), _bits(outs[0]))
        np.testing.assert_array_equal(logp, prim.gmm_log_prob(S, dtype=np.float32))
    prim.close()


def test_configs0_thirty_two_samples_of_the_walk_primitive(ctx):
    """BASELINE configs[0] at its stated size (VERDICT r3 missing 5): 32 latent samples of the 'walk' primitive, drawn and
    back-projected by the reference (tests/golden/walk_32.npz) -- every frames path of the library against the reference's frames
    and bit for bit against the float32 model, log p against sklearn's."""
    from conftest import load_golden
    g = load_golden("walk_32")
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    cp = c_oracle.COraclePrimitive(data)
    S, rows, want = g["S"], g["frame_rows"], g["frames_at_rows"]
    model = cp.frames_f32model(S)
    for path in (_capi.MG_PATH_DIRECT, _capi.MG_PATH_MFMA):
        got = prim.back_project_frames(S, path=path)
        np.testing.assert_array_equal(_bits(got), _bits(model))
        assert np.all(np.abs(got[:, rows].astype(np.float64) - want) <= pose_tol(want))
    scale = max(1.0, np.abs(want).max())
    np.testing.assert_allclose(prim.back_project_frames_f64(S)[:, rows], want, rtol=0, atol=4e-12 * scale)
    np.testing.assert_allclose(prim.gmm_log_prob(S), g["logp"], rtol=1e-9, atol=1e-7)
    frames, logp = _fused_step(ctx, prim, S.astype(np.float32), 156, 79)
    np.testing.assert_array_equal(_bits(frames), _bits(cp.frames_f32model(S.astype(np.float32).astype(np.float64))))
    prim.close()
