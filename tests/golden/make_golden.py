"""Generate the golden vectors that pin the oracle (and through it the HIP path).

Run ONLY in the build container, where the reference is mounted read-only:

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports the real reference (``graph_learn``), runs it on small seeded inputs
and stores inputs + outputs as ``.npz`` files next to this script.  Only data
is stored -- no reference source travels.  The GPU box never runs this file.
Versions used: see the ``versions`` entry written into every file.
"""

import os
import sys

# ONE BLAS / OpenMP thread: numpy.linalg.inv and the matrix products of the reference (lgmrf.py:53-60) sum in an order
# that depends on the thread count, and on the ill-conditioned lgmrf_c3 input that moves the reference's own result
# (measured below and stored as *_thread_sensitivity).  Recorded in every file's `versions`.
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[_v] = "1"

import numpy as np
import scipy
from numpy.random import default_rng
from scipy.spatial.distance import pdist, squareform

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

from graph_learn import lgmrf as ref_lgmrf  # noqa: E402
from graph_learn import operators as ref_ops  # noqa: E402
from graph_learn import smooth_learning as ref_sl  # noqa: E402
from graph_learn.evaluation import relative_error as ref_relerr  # noqa: E402
from graph_learn.sampling.graphs import sample_er_laplacian  # noqa: E402
from graph_learn.sampling.signals import sample_lgmrf  # noqa: E402
from graph_learn.temporal.smooth import WindowLogModel  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
VERS = np.array([f"numpy {np.__version__}", f"scipy {scipy.__version__}",
                 "reference LTS4/graph-learning @ 2024-12-20", "BLAS/OpenMP threads 1"])


def save(name, **arrays):
    """Write the fixture.  A regeneration must reproduce the committed DATA bit for bit (only `versions` may change):
    asserted here, array by array, before anything is overwritten."""
    path = os.path.join(HERE, name + ".npz")
    if os.path.exists(path) and not os.environ.get("GOLDEN_ALLOW_CHANGE"):
        old = np.load(path)
        assert sorted(k for k in old.files if k != "versions") == sorted(arrays), f"{name}: the set of arrays changed"
        for k, v in arrays.items():
            a, b = np.asarray(old[k]), np.asarray(v)
            assert a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes(), f"{name}[{k}] differs from the committed fixture"
    np.savez_compressed(path, versions=VERS, **arrays)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def smooth_signals(n, d, seed, p=None):
    """ER Laplacian + LGMRF signals, exactly as tests/test_clustering.py builds them."""
    rng = default_rng(seed)
    lap = sample_er_laplacian(n, (6.0 / n) if p is None else p, 0.3, 2.0, seed=rng)
    return sample_lgmrf(lap, d, seed=rng), lap


def raw_iterates(z, alpha, beta, iters, step_size=0.5, edge_init=None):
    """w and dvar after exactly k iterations: run the reference with tol=0 (never
    breaks: rel < 0 is impossible) and edge_tol=-inf (threshold disabled), and
    recover dvar = S w_k ... no: dvar is a separate variable, not S w.  The
    reference does not return it, so only w is captured here; dvar is pinned
    through the next iterate (w_{k+1} depends on dvar_k)."""
    out = {}
    for k in iters:
        ei = None if edge_init is None else edge_init.copy()
        w = ref_sl.gsp_learn_graph_log_degrees(z, alpha, beta, edge_init=ei, maxit=k, tol=0.0,
                                               step_size=step_size, edge_tol=-np.inf)
        out[k] = squareform(w, checks=False)
    return out


def count_iterations(fn):
    """Number of loop iterations the reference ran = norm calls / 4 + 1
    (4 per iteration for i > 0, smooth_learning.py:158-159)."""
    calls = {"n": 0}
    real = np.linalg.norm

    def spy(*a, **k):
        calls["n"] += 1
        return real(*a, **k)

    np.linalg.norm = spy
    try:
        res = fn()
    finally:
        np.linalg.norm = real
    return res, calls["n"] // 4 + 1


def gen_logmodel():
    iters = (1, 2, 3, 10, 50, 200)
    for n, d, seed in ((7, 23, 11), (16, 40, 12), (64, 256, 1234)):
        x, lap = smooth_signals(n, d, seed)
        z = pdist(x.T) ** 2
        arrays = {"x": x, "laplacian": lap, "z": z}
        for k in (None, 2, 5, 6):
            if k is not None and k >= n:
                continue
            m = ref_sl.LogModel(avg_degree=k)
            m._initialize(x)
            arrays[f"theta_deg{k}"] = np.float64(m.theta_)
        deg = 6 if n > 7 else 2
        theta = float(arrays[f"theta_deg{deg}"])
        zt = z * theta
        its = iters + ((2000,) if n <= 16 else ())
        for k, w in raw_iterates(zt, 1, 1, its).items():
            arrays[f"raw_w_it{k}"] = w
        # defaults (maxit 1000, tol 1e-5, edge_tol 1e-3)
        (model, n_it) = count_iterations(lambda: ref_sl.LogModel(avg_degree=deg).fit(x))
        arrays["weights_default"] = model.weights_
        arrays["n_iter_default"] = np.int64(n_it)
        arrays["deg"] = np.int64(deg)
        # converge mode with generous maxit
        (w_conv, n_conv) = count_iterations(
            lambda: ref_sl.gsp_learn_graph_log_degrees(zt, 1, 1, maxit=20000, tol=1e-5, edge_tol=-np.inf))
        arrays["raw_w_converged"] = squareform(w_conv, checks=False)
        arrays["n_iter_converged"] = np.int64(n_conv)
        # non-unit alpha/beta/step
        arrays["raw_w_a2_b05_s03_it50"] = squareform(
            ref_sl.gsp_learn_graph_log_degrees(zt, 2.0, 0.5, maxit=50, tol=0.0, step_size=0.3,
                                               edge_tol=-np.inf), checks=False)
        # warm start
        rng = default_rng(seed + 1000)
        ei = rng.uniform(0, 0.2, size=z.shape) * (rng.uniform(size=z.shape) < 0.3)
        arrays["edge_init"] = ei
        arrays["raw_w_warm_it30"] = raw_iterates(zt, 1, 1, (30,), edge_init=ei)[30]
        save(f"logmodel_n{n}_d{d}", **arrays)


def gen_batched():
    """2-D (B, E) path with its spectral-norm stopping rule (smooth_learning.py:128-129,158-159)."""
    x, _ = smooth_signals(12, 60, 21)
    xs = x.reshape(3, 20, 12)
    z = np.stack([pdist(xx.T) ** 2 for xx in xs])
    out_fixed = ref_sl.gsp_learn_graph_log_degrees(z.copy(), 1, 1, maxit=40, tol=0.0, edge_tol=-np.inf)
    (out_conv, n_it) = count_iterations(
        lambda: ref_sl.gsp_learn_graph_log_degrees(z.copy(), 1, 1, maxit=5000, tol=1e-4, edge_tol=1e-3))
    data = default_rng(240920).standard_normal((23, 7))
    wl = WindowLogModel(window_size=5, avg_degree=2).fit(data)
    wz = WindowLogModel(window_size=5, avg_degree=2)._initialize(data)
    save("batched_B3_n12", z=z, raw_w_it40=out_fixed, w_converged=out_conv, n_iter_converged=np.int64(n_it),
         win_x=data, win_theta=np.asarray(wl.theta_), win_weights=wl.weights_, win_z=wz)


def gen_theta():
    rng = default_rng(5)
    z = squareform(pdist(rng.standard_normal((30, 4))) ** 2)
    # a row with ties: duplicate two points
    pts = rng.standard_normal((20, 3))
    pts[3] = pts[7]
    pts[11] = pts[7]
    zt = squareform(pdist(pts) ** 2)
    save("theta", z=z, k=np.array([1, 2, 5, 10]),
         theta=np.array([ref_sl.get_theta(z, k) for k in (1, 2, 5, 10)]),
         z_ties=zt, k_ties=np.array([3, 4]),
         theta_ties=np.array([ref_sl.get_theta(zt, k) for k in (3, 4)]))


def gen_sum_squareform():
    arrays = {}
    for n in (2, 3, 5, 8, 33):
        s, st = ref_sl.sum_squareform(n)
        arrays[f"S_{n}"] = s.toarray()
        arrays[f"S_{n}_indptr"] = s.indptr
        arrays[f"S_{n}_indices"] = s.indices
        arrays[f"St_{n}_indptr"] = st.indptr
        arrays[f"St_{n}_indices"] = st.indices
    save("sum_squareform", **arrays)


def gen_lgmrf():
    arrays = {}
    for n in (8, 16, 32):
        rng = default_rng(7)
        lap = sample_er_laplacian(n, 6.0 / n if n > 8 else 0.5, 0.3, 2.0, seed=rng)
        prec = lap + np.eye(n)
        x = rng.multivariate_normal(np.zeros(n), np.linalg.inv(prec), size=40 * n)
        cov = x.T @ x / x.shape[0]
        arrays[f"cov_{n}"] = cov
        for npar in (1, 10):
            for rt in (1, 2):
                if n == 32 and (npar, rt) != (1, 1):
                    continue
                m = ref_lgmrf.LGMRF(norm_par=npar, regularization_type=rt).fit(cov.copy())
                arrays[f"lapl_{n}_np{npar}_rt{rt}"] = m.laplacian_
                arrays[f"inv_{n}_np{npar}_rt{rt}"] = m.inv_laplacian_
                arrays[f"conv_{n}_np{npar}_rt{rt}"] = np.int64(m.converged_)
        if n == 16:
            mask = (rng.uniform(size=(n, n)) < 0.6).astype(float)
            mask = np.triu(mask, 1)
            mask = mask + mask.T + np.eye(n)
            m = ref_lgmrf.LGMRF(adj_mask=mask.copy()).fit(cov.copy())
            arrays["mask_16"] = mask
            arrays["lapl_16_mask"] = m.laplacian_
            arrays["inv_16_mask"] = m.inv_laplacian_
            arrays["conv_16_mask"] = np.int64(m.converged_)
            # capped cycles (never converges in 3)
            m = ref_lgmrf.LGMRF(max_cycle=3).fit(cov.copy())
            arrays["lapl_16_cyc3"] = m.laplacian_
            arrays["conv_16_cyc3"] = np.int64(m.converged_)
    # direct QP pairs
    rng = default_rng(99)
    for tag, m_, tol, mi in (("a", 6, 1e-6, 100), ("b", 20, 1e-6, 100), ("c", 20, 1e-14, 7)):
        a = rng.standard_normal((m_, m_))
        q = a @ a.T / m_ + 0.1 * np.eye(m_)
        p = rng.standard_normal(m_)
        arrays[f"qp_{tag}_Q"] = q
        arrays[f"qp_{tag}_p"] = p
        arrays[f"qp_{tag}_z"] = ref_lgmrf.nonnegative_qp_solver(q, p, step=0.1, tol=tol, max_iter=mi)
        arrays[f"qp_{tag}_tol"] = np.float64(tol)
        arrays[f"qp_{tag}_maxit"] = np.int64(mi)
    # relative_error, both branches
    a = rng.standard_normal((5, 5))
    b = rng.standard_normal((5, 5))
    arrays["re_a"], arrays["re_b"] = a, b
    arrays["re_ab"] = np.float64(ref_relerr(a, b))
    arrays["re_0b"] = np.float64(ref_relerr(np.zeros((5, 5)), b))
    save("lgmrf", **arrays)


def gen_operators():
    """A few seeds of tests/test_operators.py:25-72,209-226 with outputs stored."""
    arrays = {}
    for seed in (3, 11, 1):
        rng = default_rng(seed)
        n = int(rng.integers(20, 100))
        k = int(rng.integers(1, 10))
        w = rng.standard_normal((k, n * (n - 1) // 2))
        dual = rng.standard_normal((k, n, n))
        arrays[f"w_{seed}"] = w
        arrays[f"dual_{seed}"] = dual
        arrays[f"lap_{seed}"] = ref_ops.laplacian_squareform_vec(w)
        arrays[f"adj_{seed}"] = ref_ops.laplacian_squareform_adj_vec(dual)
        arrays[f"opadj_{seed}"] = ref_ops.op_adj_dual(dual)
        arrays[f"s2v_{seed}"] = ref_ops.square_to_vec(dual)
    rng = default_rng(3)
    sig = rng.standard_normal((37, 13))
    arrays["signals"] = sig
    arrays["sq_pdiffs"] = ref_ops.squared_pdiffs(sig)
    save("operators", **arrays)


def gen_graphdict():
    """GraphDictLog (dictionary/log_model.py + base_model.py): fitted state on small seeded inputs,
    the state after the first steps (through the reference's own callback hook), every
    initialisation strategy, predict and score."""
    import warnings

    from graph_learn.dictionary import GraphDictLog

    warnings.simplefilter("ignore")
    arrays = {}

    def snap(store, tag, steps):
        def cb(model, i):
            if i in steps:
                store[f"{tag}_c_it{i}"] = model.coefficients_.copy()
                store[f"{tag}_w_it{i}"] = model.weights_.copy()
                store[f"{tag}_dual_it{i}"] = model.dual_.copy()
        return cb

    # A: the shapes of the reference's own test (tests/test_models.py:11-23)
    data = default_rng(240920).standard_normal((23, 7))
    arrays["A_x"] = data
    for k in (1, 3):
        for l1 in (0.0, 0.01):
            tag = f"A_k{k}_l1w{l1}"
            m = GraphDictLog(n_atoms=k, l1_w=l1, random_state=240920, max_iter=100)
            m.fit(data, callback=snap(arrays, tag, {0, 1, 2, 10}))
            arrays[f"{tag}_c"], arrays[f"{tag}_w"], arrays[f"{tag}_dual"] = m.coefficients_, m.weights_, m.dual_
            arrays[f"{tag}_converged"] = np.int64(m.converged_)
            arrays[f"{tag}_history"] = m.history_.to_numpy()

    # B: windows (with padding), every regulariser on, separate steps
    rng = default_rng(77)
    xb = rng.standard_normal((50, 12)) * 0.7
    arrays["B_x"] = xb
    kw = dict(n_atoms=3, window_size=4, l1_w=0.02, ortho_w=0.1, l1_c=0.01, log_c=0.05, l1_diff_c=0.001,
              step_c=0.5, step_w=0.3, step_dual=0.7, tol=1e-12, max_iter=60, random_state=5)
    m = GraphDictLog(**kw)
    m.fit(xb, callback=snap(arrays, "B", {0, 1, 2, 3, 20}))
    arrays["B_c"], arrays["B_w"], arrays["B_dual"] = m.coefficients_, m.weights_, m.dual_
    arrays["B_converged"] = np.int64(m.converged_)
    arrays["B_history"] = m.history_.to_numpy()

    # C: two ER graphs, smooth signals, step 0.5: stops on the tolerance (step 466)
    rng = default_rng(2024)
    laps = [sample_er_laplacian(16, 0.3, 0.3, 2.0, seed=rng) for _ in range(2)]
    xc = np.concatenate([sample_lgmrf(lap, 32, seed=rng) for lap in laps])
    arrays["C_x"] = xc
    m = GraphDictLog(n_atoms=2, random_state=11, l1_w=0.05, step=0.5, max_iter=1000)
    m.fit(xc, callback=snap(arrays, "C", {0}))
    arrays["C_c"], arrays["C_w"], arrays["C_dual"] = m.coefficients_, m.weights_, m.dual_
    arrays["C_converged"] = np.int64(m.converged_)
    arrays["C_history"] = m.history_.to_numpy()
    arrays["C_score"] = np.float64(m.score(xc))
    xp = np.concatenate([sample_lgmrf(lap, 8, seed=rng) for lap in laps])
    arrays["C_predict_x"] = xp
    m.random_state = np.random.RandomState(3)
    m.max_iter = 50
    arrays["C_predict"] = m.predict(xp)

    # D: every initialisation strategy (state right after _initialize)
    xd = default_rng(9).standard_normal((24, 6))
    arrays["D_x"] = xd
    for tag, kwd in {
        "uniform_prior": dict(init_strategy="uniform", weight_prior=0.5, coefficient_prior=0.8),
        "discrete": dict(init_strategy="discrete", weight_prior=0.3),
        "exp": dict(init_strategy="exp", weight_prior=0.2, coefficient_prior=0.4),
        "exact": dict(init_strategy="exact", weight_prior=0.25, coefficient_prior=0.6),
        "kmd_corr_a": dict(init_strategy_c="k-means++d", init_strategy_w="correlation_a"),
        "kmc_corr_a_pinv": dict(init_strategy_c="k-means++c", init_strategy_w="correlation_a_pinv"),
        "window_uniform": dict(init_strategy="uniform", window_size=3),
    }.items():
        m = GraphDictLog(n_atoms=3, random_state=42, **kwd)
        m._initialize(xd)
        arrays[f"D_{tag}_c"], arrays[f"D_{tag}_w"] = m.coefficients_, m.weights_
    save("graphdict", **arrays)


def gen_clustering():
    """Clustering M-step (clustering/glmm.py:18-85) and get_theta's blocks branch (:220-235)."""
    from graph_learn.clustering.glmm import _estimate_gauss_laplacian_parameters as mstep

    arrays = {}
    rng = default_rng(31)
    n, t, k = 12, 80, 3
    laps = [sample_er_laplacian(n, 0.4, 0.3, 2.0, seed=rng) for _ in range(k)]
    x = np.concatenate([sample_lgmrf(lap, t // k + 1, seed=rng) for lap in laps])[:t]
    logits = rng.standard_normal((t, k)) * 2
    resp = np.exp(logits) / np.exp(logits).sum(1, keepdims=True)
    blocks = np.array([0] * 5 + [1] * 7)
    degs = {(0, 0): 2, (0, 1): 3, (1, 1): 2}
    arrays.update(x=x, resp=resp, blocks=blocks, delta=np.float64(2.0))
    cases = {
        "deg3": dict(avg_degree=3),
        "theta05": dict(theta=0.5),
        "blocks_int": dict(avg_degree=2, blocks=blocks),
        "blocks_dict": dict(avg_degree=degs, blocks=blocks),
    }
    for tag, kw in cases.items():
        w, m, lap = mstep(x, resp, 2.0, **kw)
        arrays[f"{tag}_weights"], arrays[f"{tag}_means"], arrays[f"{tag}_laplacians"] = w, m, lap
    # warm start from previous Laplacians
    _, _, lap0 = mstep(x, resp, 2.0, avg_degree=3)
    w, m, lap = mstep(x, resp * 0.9 + 0.1 / k, 2.0, avg_degree=3, laplacians=lap0)
    arrays["warm_weights"], arrays["warm_means"], arrays["warm_laplacians"] = w, m, lap
    # get_theta blocks branch on a plain distance matrix
    z = squareform(pdist(rng.standard_normal((n, 6))) ** 2)
    arrays["theta_z"] = z
    arrays["theta_blocks_int"] = ref_sl.get_theta(z, 2, blocks=blocks)
    arrays["theta_blocks_dict"] = ref_sl.get_theta(z, degs, blocks=blocks)
    save("clustering", **arrays)


def c3_covariance(x):
    """S = X'X / d summed sample by sample with elementwise operations only (no BLAS), so that the
    test rebuilds bit-identical input from the stored signals on any machine."""
    d, n = x.shape
    cov = np.zeros((n, n))
    for t in range(d):
        cov += np.outer(x[t], x[t])
    return cov / d


def gen_lgmrf_c3():
    """LGMRF on inputs shaped like BASELINE configs[2] (S = X'X/d, X ~ N(0, (L+I)^-1), d = n/4 < n: rank
    deficient, mean(S) ~ 1e-3).  lgmrf.py:169-226 after a capped number of cycles: n = 512 after two
    (full matrices, ~24 s) and n = 1024 after one (~85 s; diagonal, sampled rows and norms: this is the case
    DESIGN.md section 5 cites for the reference's own blow-up on this input).  The signals are stored in
    float32 (the input IS their float64 widening) to keep the fixture small."""
    from scipy.linalg import solve_triangular

    arrays = {}
    for n, d, cycles in ((512, 128, 2), (1024, 256, 1)):
        rng = default_rng(7)
        lap = sample_er_laplacian(n, 6.0 / n, 0.3, 2.0, seed=rng)
        chol = np.linalg.cholesky(lap + np.eye(n))
        x = solve_triangular(chol, rng.standard_normal((n, d)), lower=True, trans="T").T
        x32 = x.astype(np.float32)
        cov = c3_covariance(x32.astype(np.float64))
        m = ref_lgmrf.LGMRF(max_cycle=cycles).fit(cov.copy())
        iu = np.triu_indices(n)
        tag = f"n{n}"
        # how far the REFERENCE itself moves on this input when only the BLAS thread count changes (1 -> 8): the tolerance
        # of the tests that compare with this fixture is set from it (relative to the largest entry of each quantity)
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=8):
            m8 = ref_lgmrf.LGMRF(max_cycle=cycles).fit(cov.copy())
        rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
        arrays[f"{tag}_thread_sensitivity"] = np.array([rel(m8.laplacian_, m.laplacian_), rel(m8.inv_laplacian_, m.inv_laplacian_)])
        print(tag, "1 thread vs 8 threads: laplacian", arrays[f"{tag}_thread_sensitivity"][0], "inverse",
              arrays[f"{tag}_thread_sensitivity"][1], flush=True)
        arrays[f"{tag}_x32"] = x32
        arrays[f"{tag}_cycles"] = np.int64(cycles)
        arrays[f"{tag}_cov_mean"] = np.float64(cov.mean())
        arrays[f"{tag}_conv"] = np.int64(m.converged_)
        rows = np.array([0, n // 2 - 3, n - 1])
        arrays[f"{tag}_rows"] = rows
        if n == 512:
            arrays[f"{tag}_lapl_triu"] = m.laplacian_[iu]
        else:
            arrays[f"{tag}_lapl_diag"] = np.diag(m.laplacian_).copy()
            arrays[f"{tag}_lapl_rows"] = m.laplacian_[rows].copy()
            arrays[f"{tag}_lapl_fro"] = np.float64(np.linalg.norm(m.laplacian_))
            arrays[f"{tag}_lapl_absmax"] = np.float64(np.abs(m.laplacian_).max())
        arrays[f"{tag}_inv_diag"] = np.diag(m.inv_laplacian_).copy()
        arrays[f"{tag}_inv_rows"] = m.inv_laplacian_[rows].copy()
        print(tag, "max|Theta| =", np.abs(m.laplacian_).max(), " mean(S) =", cov.mean(), flush=True)
    save("lgmrf_c3", **arrays)


def gen_lgmrf_mask():
    """LGMRF with a SPARSE adj_mask (lgmrf.py:191-202: every node's QP runs on M[nz][:, nz] only): the support of the true
    graph plus 6 % random extra candidate pairs, n = 64 and n = 96, well-conditioned input (precision L + I, 40 n samples),
    default parameters, run to convergence."""
    arrays = {}
    for n in (64, 96):
        rng = default_rng(11)
        lap = sample_er_laplacian(n, 6.0 / n, 0.3, 2.0, seed=rng)
        x = rng.multivariate_normal(np.zeros(n), np.linalg.inv(lap + np.eye(n)), size=40 * n)
        cov = x.T @ x / x.shape[0]
        extra = np.triu(rng.uniform(size=(n, n)) < 0.06, 1)
        mask = ((lap != 0) | extra | extra.T | np.eye(n, dtype=bool)).astype(float)
        m = ref_lgmrf.LGMRF(adj_mask=mask.copy()).fit(cov.copy())
        arrays[f"cov_{n}"], arrays[f"mask_{n}"] = cov, mask
        arrays[f"lapl_{n}"], arrays[f"inv_{n}"] = m.laplacian_, m.inv_laplacian_
        arrays[f"conv_{n}"] = np.int64(m.converged_)
        deg = (mask.sum(0) - 1)
        print(f"n={n}: candidates per node {deg.min():.0f}..{deg.max():.0f}, converged_ = {m.converged_}, "
              f"edges learned {(np.abs(np.triu(m.laplacian_, 1)) > 1e-8).sum()}", flush=True)
    save("lgmrf_mask", **arrays)


if __name__ == "__main__":
    if sys.argv[1:] == ["lgmrf_c3"]:      # the two-minute one on its own
        gen_lgmrf_c3()
        sys.exit(0)
    if sys.argv[1:] == ["lgmrf_mask"]:
        gen_lgmrf_mask()
        sys.exit(0)
    if sys.argv[1:] == ["fast"]:          # everything but the two slow LGMRF fixtures
        for gen in (gen_sum_squareform, gen_theta, gen_logmodel, gen_batched, gen_lgmrf, gen_operators, gen_graphdict, gen_clustering):
            gen()
        sys.exit(0)
    gen_sum_squareform()
    gen_theta()
    gen_logmodel()
    gen_batched()
    gen_lgmrf()
    gen_operators()
    gen_graphdict()
    gen_clustering()
    gen_lgmrf_mask()
    gen_lgmrf_c3()
