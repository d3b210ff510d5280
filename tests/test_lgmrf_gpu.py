"""LGMRF on the GPU (through the C ABI) against the reference's golden vectors and the oracle.

Tolerances: the device applies the same iteration as the reference but inverts I + rho M by
Gauss-Jordan (general path) or through the Woodbury-maintained (Theta + rho I)^-1 (fast path)
instead of LAPACK's inv, and sums in a different order: iterates agree to rounding times the
conditioning; the ADMM / cycle stopping rules are discontinuous at their tolerances (1e-6 / 1e-3),
so final matrices are compared to 1e-6 relative to their scale, and `converged_` exactly.
"""

import numpy as np
import pytest

from conftest import c3_covariance
from graph_learning_amd import LGMRF, nonnegative_qp_solver, relative_error
from graph_learning_amd.synth import er_laplacian
from oracle import lgmrf as olg

pytestmark = pytest.mark.gpu


import os

# The forms the library chooses by itself (wave / quad / tagged / wide) and the automatic choice run in every default
# session.  The superseded forms -- one launch per step, the single-workgroup LDS form, the grid-barrier persistent form;
# DESIGN.md section 5 -- run the whole matrix only with GRAPHLEARN_AMD_TEST_ALL=1; by default they keep ONE smoke case each
# (test_superseded_admm_forms_smoke, below), so the suite's wall time stays inside the driver's step limit.
ALL_FORMS = os.environ.get("GRAPHLEARN_AMD_TEST_ALL", "0") not in ("", "0")
SUPERSEDED = ["launch", "one", "resident"]        # (+ "wide_lds", which only exists from n = 256 on: covered at n = 512 below)
FORMS = ["wave", "quad", "tagged", "wide", "auto"] + (SUPERSEDED + ["wide_lds"] if ALL_FORMS else [])


def set_form(monkeypatch, form):
    monkeypatch.delenv("GRAPHLEARN_AMD_LGMRF_WIDE", raising=False)
    monkeypatch.delenv("GRAPHLEARN_AMD_LGMRF_REGS", raising=False)
    if form == "auto":
        monkeypatch.delenv("GRAPHLEARN_AMD_LGMRF_ADMM", raising=False)
    elif form in ("tagged", "wide", "wide_lds"):
        # "tagged": up to 8 rows per workgroup, in LDS; "wide": 16 rows, all in registers (round 4; the automatic choice from
        # n = 256 on); "wide_lds": round 2's wide form, 8 of the 16 rows in LDS
        monkeypatch.setenv("GRAPHLEARN_AMD_LGMRF_ADMM", "tagged")
        monkeypatch.setenv("GRAPHLEARN_AMD_LGMRF_WIDE", "0" if form == "tagged" else "1")
        if form == "wide_lds":
            monkeypatch.setenv("GRAPHLEARN_AMD_LGMRF_REGS", "0")
    else:
        monkeypatch.setenv("GRAPHLEARN_AMD_LGMRF_ADMM", form)


@pytest.fixture(autouse=True, params=FORMS)
def admm_form(request, monkeypatch):
    """Every test runs with every form of the QP's ADMM loop: one launch per step, the single-wave kernel
    (n <= 64, rows of the inverse in registers), the single-workgroup kernel with four lanes per row
    (64 < n <= 160, rows in registers), the single-workgroup kernel with the inverse in LDS (n <= 64), the
    persistent multi-workgroup kernels (matrix rows resident in LDS; steps separated by grid barriers, or
    the right-hand side handed over as self-validating tagged records -- "tagged": up to 8 rows per workgroup,
    all in LDS; "wide": up to 16, half of them in registers) and the automatic choice (wave to n = 64, quad to
    n = 160, tagged above, wide from n = 256)."""
    set_form(monkeypatch, request.param)
    return request.param


@pytest.mark.parametrize("form", SUPERSEDED)
def test_superseded_admm_forms_smoke(golden, monkeypatch, admm_form, form):
    """One golden fit (n = 32, fast and general path) and one golden QP on each superseded form of the ADMM loop: they stay
    selectable (GRAPHLEARN_AMD_LGMRF_ADMM) as comparison points, so they stay correct."""
    if admm_form != "auto":
        pytest.skip("once per session")
    set_form(monkeypatch, form)
    g = golden("lgmrf")
    for path in (1, 2):
        m = LGMRF(norm_par=1, regularization_type=1).fit(g["cov_32"].copy(), path=path)
        assert m.converged_ == int(g["conv_32_np1_rt1"])
        assert close(m.laplacian_, g["lapl_32_np1_rt1"]) and close(m.inv_laplacian_, g["inv_32_np1_rt1"])
    q, p = g["qp_a_Q"], g["qp_a_p"]
    tol, mi = float(g["qp_a_tol"]), int(g["qp_a_maxit"])
    z, iters = nonnegative_qp_solver(q, p, step=0.1, tol=tol, max_iter=mi, return_iters=True)
    _, ref_iters = olg.nonneg_qp_admm(q, p, step=0.1, tol=tol, max_iter=mi, return_iters=True)
    assert abs(iters - ref_iters) <= 1
    np.testing.assert_allclose(z, g["qp_a_z"], rtol=0, atol=1e-7 if iters != ref_iters else 1e-11)


def close(a, b, rel=1e-6):
    return np.abs(a - b).max() <= rel * max(np.abs(b).max(), 1e-300)


def test_relative_error(golden):
    g = golden("lgmrf")
    assert relative_error(g["re_a"], g["re_b"]) == pytest.approx(float(g["re_ab"]), rel=1e-13)
    assert relative_error(np.zeros((5, 5)), g["re_b"]) == pytest.approx(float(g["re_0b"]), rel=1e-13)
    with pytest.raises(ValueError):
        relative_error(np.zeros(3), np.zeros(4))


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_qp_solver_golden(golden, tag):
    g = golden("lgmrf")
    q, p = g[f"qp_{tag}_Q"], g[f"qp_{tag}_p"]
    tol, mi = float(g[f"qp_{tag}_tol"]), int(g[f"qp_{tag}_maxit"])
    z, iters = nonnegative_qp_solver(q, p, step=0.1, tol=tol, max_iter=mi, return_iters=True)
    _, ref_iters = olg.nonneg_qp_admm(q, p, step=0.1, tol=tol, max_iter=mi, return_iters=True)
    assert abs(iters - ref_iters) <= 1
    np.testing.assert_allclose(z, g[f"qp_{tag}_z"], rtol=0, atol=1e-7 if iters != ref_iters else 1e-11)
    assert (z >= 0).all()


@pytest.mark.parametrize("path", [1, 2])
@pytest.mark.parametrize("n,npar,rt", [(8, 1, 1), (8, 1, 2), (8, 10, 1), (8, 10, 2), (16, 1, 1), (16, 10, 2), (32, 1, 1)])
def test_lgmrf_fit_golden(golden, n, npar, rt, path):
    g = golden("lgmrf")
    cov = g[f"cov_{n}"].copy()
    m = LGMRF(norm_par=npar, regularization_type=rt).fit(cov, path=path)
    assert np.array_equal(cov, g[f"cov_{n}"])                      # input not modified
    assert m.n_nodes_ == n and m.converged_ == int(g[f"conv_{n}_np{npar}_rt{rt}"])
    assert close(m.laplacian_, g[f"lapl_{n}_np{npar}_rt{rt}"])
    assert close(m.inv_laplacian_, g[f"inv_{n}_np{npar}_rt{rt}"])
    # structure: symmetric, non-positive off-diagonals (generalized Laplacian), inverse consistent
    assert np.allclose(m.laplacian_, m.laplacian_.T, rtol=0, atol=1e-12)
    off = m.laplacian_[~np.eye(n, dtype=bool)]
    assert (off <= 0).all()
    assert np.allclose(m.laplacian_ @ m.inv_laplacian_, np.eye(n), atol=1e-8)


def test_lgmrf_mask_and_cycle_cap(golden):
    g = golden("lgmrf")
    m = LGMRF(adj_mask=g["mask_16"].copy()).fit(g["cov_16"].copy())
    assert m.converged_ == int(g["conv_16_mask"])
    assert close(m.laplacian_, g["lapl_16_mask"]) and close(m.inv_laplacian_, g["inv_16_mask"])
    assert (m.laplacian_[g["mask_16"] == 0] == 0).all()            # masked-out pairs stay disconnected
    m = LGMRF(max_cycle=3).fit(g["cov_16"].copy())
    assert m.converged_ == -1 == int(g["conv_16_cyc3"]) and m.n_cycles_ == 3
    assert close(m.laplacian_, g["lapl_16_cyc3"])


@pytest.mark.parametrize("n,fused", [(64, "1"), (64, "0"), (96, "1")])
def test_lgmrf_sparse_mask_against_reference_golden(golden, admm_form, monkeypatch, n, fused):
    """A SPARSE adj_mask (the true support plus 6 % extra candidates: 4..19 per node) run to convergence, against what the
    reference produced (tests/golden/make_golden.py::gen_lgmrf_mask; lgmrf.py:191-202).  n = 64 through the single-launch
    fit and, with it switched off, through the launch path; n = 96 through the launch path: there every node's QP is
    gathered to its na x na active set (k_lg_compact_*), inverted in one launch and solved by the form that fits na."""
    if fused == "0" or n > 64:
        if admm_form not in ("auto", "launch", "wave"):
            pytest.skip("the gathered problems have 4..19 unknowns: the single-wave form serves them, or one launch per step")
    elif admm_form != "auto":
        pytest.skip("the single-launch fit has one form")
    monkeypatch.setenv("GRAPHLEARN_AMD_LGMRF_FUSED", fused)
    g = golden("lgmrf_mask")
    mask = g[f"mask_{n}"]
    m = LGMRF(adj_mask=mask.copy()).fit(g[f"cov_{n}"].copy())
    assert m.converged_ == int(g[f"conv_{n}"]) == 5
    err = np.abs(m.laplacian_ - g[f"lapl_{n}"]).max() / np.abs(g[f"lapl_{n}"]).max()
    print(f"[lgmrf sparse mask n={n} fused={fused} form={admm_form}] max|Theta - ref| / max|ref| = {err:.1e}")
    assert close(m.laplacian_, g[f"lapl_{n}"]) and close(m.inv_laplacian_, g[f"inv_{n}"])
    assert (m.laplacian_[mask == 0] == 0).all()                    # masked-out pairs stay disconnected
    assert np.array_equal(m.laplacian_, m.laplacian_.T)


def test_lgmrf_sparse_mask_n256_against_oracle(admm_form):
    """n = 256 with a sparse mask (true support + 3 % extra pairs), three cycles against the oracle: the launch path with
    gathered QPs (4..25 unknowns each), and a node whose column of the mask is empty (its QP has no unknown at all)."""
    if admm_form not in ("auto", "launch"):
        pytest.skip("the gathered problems are served by the single-wave form (auto) or by one launch per step")
    n = 256
    rng = np.random.default_rng(21)
    lap = er_laplacian(n, 6.0 / n, 0.3, 2.0, seed=rng)
    x = rng.multivariate_normal(np.zeros(n), np.linalg.inv(lap + np.eye(n)), size=20 * n)
    cov = x.T @ x / x.shape[0]
    extra = np.triu(rng.uniform(size=(n, n)) < 0.03, 1)
    mask = ((lap != 0) | extra | extra.T | np.eye(n, dtype=bool)).astype(float)
    mask[7, :] = mask[:, 7] = 0.0                                   # node 7: no candidate neighbour
    mask[7, 7] = 1.0
    o = olg.LGMRFOracle(adj_mask=mask.copy(), max_cycle=3).fit(cov.copy())
    m = LGMRF(adj_mask=mask.copy(), max_cycle=3).fit(cov.copy())
    assert m.converged_ == o.converged_ == -1 and m.n_cycles_ == 3
    err = np.abs(m.laplacian_ - o.laplacian_).max() / np.abs(o.laplacian_).max()
    print(f"[lgmrf sparse mask n=256 form={admm_form}] max|Theta - oracle| / max = {err:.1e}")
    assert close(m.laplacian_, o.laplacian_, 1e-9) and close(m.inv_laplacian_, o.inv_laplacian_, 1e-9)
    assert (m.laplacian_[mask == 0] == 0).all() and (np.delete(m.laplacian_[7], 7) == 0).all()


def test_lgmrf_errors():
    x = np.eye(4) + 1.0
    with pytest.raises(ValueError):
        LGMRF(laplacian_set="nope").fit(x.copy())
    with pytest.raises(ValueError):
        LGMRF(regularization_type=3).fit(x.copy())
    with pytest.raises(ValueError):
        LGMRF(adj_mask=np.ones((3, 3))).fit(x.copy())
    with pytest.raises(NotImplementedError):
        LGMRF(laplacian_set="c").fit(x.copy())
    assert LGMRF().get_params()["qp_step"] == 0.1


def test_fast_path_matches_general_path_and_oracle_n96():
    """Beyond the golden sizes: n=96, precision L + I, both device paths vs the oracle."""
    rng = np.random.default_rng(3)
    n = 96
    e = n * (n - 1) // 2
    w = (rng.uniform(size=e) < 6.0 / n) * rng.uniform(0.3, 2.0, size=e)
    lap = np.zeros((n, n))
    lap[np.triu_indices(n, 1)] = -w
    lap = lap + lap.T
    lap[np.arange(n), np.arange(n)] = -lap.sum(1)
    x = rng.multivariate_normal(np.zeros(n), np.linalg.inv(lap + np.eye(n)), size=30 * n)
    cov = x.T @ x / x.shape[0]
    o = olg.LGMRFOracle().fit(cov.copy())
    for path in (1, 2):
        m = LGMRF().fit(cov.copy(), path=path)
        assert m.converged_ == o.converged_
        assert close(m.laplacian_, o.laplacian_) and close(m.inv_laplacian_, o.inv_laplacian_)


def test_persistent_forms_fall_back_when_not_co_resident(golden, monkeypatch, admm_form):
    """The persistent kernels need all their workgroups on the chip at once.  With the test hook the
    start-up rendezvous can never complete: the QP entry point reruns with one launch per step, the fit
    restores the start of the cycle and redoes it that way -- same results, about a second later."""
    if admm_form not in ("resident", "tagged", "wide"):
        pytest.skip("only the persistent forms have a rendezvous")
    monkeypatch.setenv("GRAPHLEARN_AMD_LGMRF_TEST_NOT_RESIDENT", "1")
    g = golden("lgmrf")
    z = nonnegative_qp_solver(g["qp_a_Q"], g["qp_a_p"], step=0.1, tol=float(g["qp_a_tol"]), max_iter=int(g["qp_a_maxit"]))
    np.testing.assert_allclose(z, g["qp_a_z"], rtol=0, atol=1e-7)
    m = LGMRF().fit(g["cov_16"].copy())
    assert m.converged_ == int(g["conv_16_np1_rt1"])
    assert close(m.laplacian_, g["lapl_16_np1_rt1"]) and close(m.inv_laplacian_, g["inv_16_np1_rt1"])


def test_qp_solver_many_iterations():
    """More steps than the default 100 (the tagged form numbers its hand-offs by step) against the oracle."""
    rng = np.random.default_rng(5)
    n = 40
    a = rng.standard_normal((n, 2 * n))
    q, p = a @ a.T / (2 * n) + 0.1 * np.eye(n), rng.standard_normal(n)
    z, iters = nonnegative_qp_solver(q, p, step=0.1, tol=1e-13, max_iter=400, return_iters=True)
    zr, ref_iters = olg.nonneg_qp_admm(q, p, step=0.1, tol=1e-13, max_iter=400, return_iters=True)
    assert abs(iters - ref_iters) <= 1 and iters > 120
    np.testing.assert_allclose(z, zr, rtol=0, atol=1e-10)


@pytest.mark.parametrize("n", [65, 100, 128, 129, 160, 161])
def test_qp_solver_mid_sizes(n):
    """Both sides of the limits of the register-resident single-workgroup form (64 < n <= 160) against the oracle."""
    rng = np.random.default_rng(n)
    a = rng.standard_normal((n, n + 5))
    q, p = a @ a.T / n + 0.05 * np.eye(n), rng.standard_normal(n)
    for tol, mi in ((1e-6, 100), (1e-3, 100), (0.0, 7)):
        z, iters = nonnegative_qp_solver(q, p, step=0.1, tol=tol, max_iter=mi, return_iters=True)
        zr, ref_iters = olg.nonneg_qp_admm(q, p, step=0.1, tol=tol, max_iter=mi, return_iters=True)
        assert abs(iters - ref_iters) <= 1
        if iters == ref_iters:
            np.testing.assert_allclose(z, zr, rtol=0, atol=1e-10)


@pytest.mark.parametrize("n", [200, 300, 1000])
def test_qp_solver_blocked_inverse_sizes(n):
    """n > 128: (I + rho M)^-1 by the BLOCKED Gauss-Jordan elimination (k_bgj_*: 64-pivot steps, f64 MFMA products, the
    next pivot block inverted by the update kernel), blocks padded by the identity (n = 200, 300, 1000 are not multiples
    of 64): against the oracle (numpy.linalg.inv + the same ADMM steps), and bit-identical run to run."""
    rng = np.random.default_rng(n)
    a = rng.standard_normal((n, n + 5))
    q, p = a @ a.T / n + 0.05 * np.eye(n), rng.standard_normal(n)
    z, iters = nonnegative_qp_solver(q, p, step=0.1, tol=0.0, max_iter=12, return_iters=True)
    zr, ref_iters = olg.nonneg_qp_admm(q, p, step=0.1, tol=0.0, max_iter=12, return_iters=True)
    assert iters == ref_iters == 12
    np.testing.assert_allclose(z, zr, rtol=0, atol=1e-10)
    z2 = nonnegative_qp_solver(q, p, step=0.1, tol=0.0, max_iter=12)
    assert np.array_equal(z, z2)


# ------------------------------------------------------------------ BASELINE configs[2] shape
LARGE_FORMS = ("auto", "tagged", "wide", "wide_lds", "resident", "launch")


def triu_to_square(v, n):
    m = np.zeros((n, n))
    m[np.triu_indices(n)] = v
    return m + np.triu(m, 1).T


@pytest.mark.parametrize("path", [1, 2])
def test_lgmrf_c3_shape_n512_against_reference_fixture(golden, admm_form, path):
    """Input shaped like BASELINE configs[2] (S = X'X/d, X ~ N(0, (L+I)^-1), d = n/4: rank deficient,
    mean(S) = 2e-3), n = 512, two cycles of lgmrf.py:169-226 -- against the matrices the REFERENCE produced
    (tests/golden/make_golden.py::gen_lgmrf_c3).  On this input every QP runs its 100 ADMM steps and the
    iterate grows (max|Theta| = 22.58 after two cycles): the comparison is relative to that scale, and the
    tolerance follows from the measured sensitivity of the REFERENCE on this input: the fixture (made with ONE BLAS
    thread, recorded in its `versions`) stores how far the reference's own result moves when only the thread count
    changes to 8 (n512_thread_sensitivity = 3.2e-9 of max|Theta|); the device path sums in yet another order
    (wave-wide butterflies, 100 ADMM steps x 512 nodes x 2 cycles) and is held to 300 x that = 1e-6."""
    if admm_form not in LARGE_FORMS:
        pytest.skip("single-wave / single-workgroup forms end at n = 160")
    if path == 1 and admm_form != "auto":
        pytest.skip("the general path (explicit Gauss-Jordan inverse per node) once: ~10 s at this size")
    g = golden("lgmrf_c3")
    n = 512
    cov = c3_covariance(g["n512_x32"])
    assert cov.mean() == float(g["n512_cov_mean"])                      # bit-identical input
    m = LGMRF(max_cycle=int(g["n512_cycles"])).fit(cov.copy(), path=path)
    ref = triu_to_square(g["n512_lapl_triu"], n)
    scale = np.abs(ref).max()
    assert scale == pytest.approx(22.578421555988328, rel=1e-12)
    assert m.converged_ == int(g["n512_conv"]) == -1 and m.n_cycles_ == 2
    sens = float(g["n512_thread_sensitivity"][0])
    assert 1e-9 < sens < 1e-8
    err = np.abs(m.laplacian_ - ref).max() / scale
    print(f"[lgmrf c3 n=512 form={admm_form} path={path}] max|Theta - ref| / max|ref| = {err:.2e} "
          f"(reference, 1 vs 8 BLAS threads: {sens:.1e})")
    assert err < 300 * sens
    rows = g["n512_rows"]
    assert np.abs(m.inv_laplacian_[rows] - g["n512_inv_rows"]).max() <= 1e-6 * np.abs(g["n512_inv_rows"]).max()
    assert np.abs(np.diag(m.inv_laplacian_) - g["n512_inv_diag"]).max() <= 1e-6 * np.abs(g["n512_inv_diag"]).max()
    assert np.array_equal(m.laplacian_, m.laplacian_.T)


def test_wide_form_with_rows_in_lds_smoke(golden, admm_form, monkeypatch):
    """Round 2's wide form (8 of a workgroup's 16 rows in LDS) stays selectable (GRAPHLEARN_AMD_LGMRF_REGS=0) beside round 4's
    default (all 16 in registers): the n = 512 reference fixture once in every default session, and EQUAL matrices from the
    two (same columns per thread, same butterflies: the same order of summation)."""
    if admm_form != "auto":
        pytest.skip("once per session")
    g = golden("lgmrf_c3")
    n = 512
    cov = c3_covariance(g["n512_x32"])
    ref = triu_to_square(g["n512_lapl_triu"], n)
    out = {}
    for form in ("wide_lds", "wide"):
        set_form(monkeypatch, form)
        m = LGMRF(max_cycle=int(g["n512_cycles"])).fit(cov.copy())
        out[form] = (m.laplacian_, m.inv_laplacian_)
        assert np.abs(m.laplacian_ - ref).max() / np.abs(ref).max() < 300 * float(g["n512_thread_sensitivity"][0])
    assert np.array_equal(out["wide"][0], out["wide_lds"][0]) and np.array_equal(out["wide"][1], out["wide_lds"][1])


def test_lgmrf_c3_shape_n1024_blow_up_matches_reference(golden, admm_form):
    """n = 1024, d = 256, ONE cycle: the reference itself diverges on this input (max|Theta| = 1.5348e4 after
    the cycle, fixture made by the reference in 85 s); the device path reproduces diagonal, sampled rows,
    Frobenius norm and maximum.  This is the evidence for DESIGN.md section 5 ("C3 as specified is ill-posed").
    Tolerance: the fixture (ONE BLAS thread) records that the reference's own Theta moves by 9.7e-7 of max|Theta| and
    its inverse by 1.7e-6 of max|C| when only the thread count changes to 8 (n1024_thread_sensitivity); the device
    path is held to 20 x those figures (2e-5 / 3.5e-5), relative to the REFERENCE's maxima (max|Theta| and max|C| both
    sit on the diagonals the fixture stores); the observed errors are printed (DESIGN.md section 5 records them)."""
    if admm_form != "auto":
        pytest.skip("once, on the default form")
    g = golden("lgmrf_c3")
    cov = c3_covariance(g["n1024_x32"])
    assert cov.mean() == float(g["n1024_cov_mean"])
    m = LGMRF(max_cycle=1).fit(cov.copy())
    scale = float(g["n1024_lapl_absmax"])
    assert scale == pytest.approx(15348.149994978496, rel=1e-12)
    assert m.converged_ == -1 and m.n_cycles_ == 1
    s_th, s_inv = (float(v) for v in g["n1024_thread_sensitivity"])
    assert 3e-7 < s_th < 3e-6 and 5e-7 < s_inv < 5e-6
    tol_th, tol_inv = 20 * s_th, 20 * s_inv
    rows = g["n1024_rows"]
    e_rows = np.abs(m.laplacian_[rows] - g["n1024_lapl_rows"]).max() / scale
    e_diag = np.abs(np.diag(m.laplacian_) - g["n1024_lapl_diag"]).max() / scale
    # (normalised by REFERENCE quantities only: C is positive definite, so max|C| sits on its diagonal -- the fixture's)
    inv_scale = np.abs(g["n1024_inv_diag"]).max()
    assert inv_scale == pytest.approx(72801073.87114467, rel=1e-12)
    e_inv = np.abs(m.inv_laplacian_[rows] - g["n1024_inv_rows"]).max() / inv_scale
    e_inv_diag = np.abs(np.diag(m.inv_laplacian_) - g["n1024_inv_diag"]).max() / inv_scale
    print(f"[lgmrf c3 n=1024] Theta rows {e_rows:.2e} diag {e_diag:.2e} (of max|Theta| = {scale:.6g}; reference 1 vs 8 threads "
          f"{s_th:.1e}); C rows {e_inv:.2e} diag {e_inv_diag:.2e} (of the reference's max|C| = {inv_scale:.4g}; reference {s_inv:.1e})")
    assert e_rows < tol_th and e_diag < tol_th     # 1.5e4-fold growth in one cycle: rounding is amplified accordingly
    assert np.abs(m.laplacian_).max() == pytest.approx(scale, rel=tol_th)
    assert np.linalg.norm(m.laplacian_) == pytest.approx(float(g["n1024_lapl_fro"]), rel=tol_th)
    assert e_inv < tol_inv and e_inv_diag < tol_inv


def test_lgmrf_c3_full_size_one_cycle_properties(admm_form):
    """BASELINE configs[2] at full size (n = 2048, d = 512), one cycle = 2048 node updates: size-independent
    properties of a block-coordinate cycle -- Theta symmetric, Theta C = I (C is maintained by rank-one
    updates, lgmrf.py:207-219), non-positive off-diagonals, C_uu = K_uu for the last node (lgmrf.py:213),
    bit-reproducible
    -- and the ms per node update printed for the record."""
    import time

    from scipy.linalg import solve_triangular

    if admm_form not in ("auto", "tagged", "wide"):
        pytest.skip("the default (tagged) form; the others are covered at n = 512")
    n, d = 2048, 512
    rng = np.random.default_rng(7)
    lap = er_laplacian(n, 6.0 / n, 0.3, 2.0, seed=rng)
    x = solve_triangular(np.linalg.cholesky(lap + np.eye(n)), rng.standard_normal((n, d)), lower=True, trans="T").T
    cov = x.T @ x / d
    t0 = time.perf_counter()
    m = LGMRF(max_cycle=1).fit(cov.copy())
    dt = time.perf_counter() - t0
    print(f"[lgmrf c3 n=2048 d=512 form={admm_form}] one cycle {dt:.2f} s = {1e3 * dt / n:.3f} ms per node update "
          f"(transfers included), max|Theta| = {np.abs(m.laplacian_).max():.4g}")
    th, c = m.laplacian_, m.inv_laplacian_
    assert m.n_cycles_ == 1 and m.converged_ == -1
    assert np.isfinite(th).all() and np.isfinite(c).all()
    assert np.array_equal(th, th.T) and np.array_equal(c, c.T)
    assert (th[~np.eye(n, dtype=bool)] <= 0).all()
    k = cov / cov.mean() + 2 * np.eye(n) - 1.0                         # lgmrf.py:229, 154-161 (type 1)
    assert c[n - 1, n - 1] == pytest.approx(k[n - 1, n - 1], rel=1e-12)   # the last node updated (lgmrf.py:213)
    resid = np.abs(th @ c - np.eye(n)).max()
    print(f"[lgmrf c3 n=2048] max|Theta C - I| = {resid:.2e}")
    assert resid < 1e-6 * max(1.0, np.abs(th).max() * np.abs(c).max())
    m2 = LGMRF(max_cycle=1).fit(cov.copy())
    assert np.array_equal(m2.laplacian_, th) and np.array_equal(m2.inv_laplacian_, c)
