#!/usr/bin/env python
"""Headline benchmark: primal-dual iterations/sec of the LogModel solver.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): LogModel n=4096 nodes, d=1024 samples,
fp64, synthetic smooth signals (ER graph, X ~ N(0, pinv(L))), avg_degree=10,
fixed-iteration mode (tol = 0).  One "step" = one primal-dual iteration
(graph_learn/smooth_learning.py:146-167) over the whole edge lattice.

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank solves
its own independent problem (weak scaling, seed = rank), no collective on the
iteration path; RCCL is used for the device barrier around the timed region and
for the final gather of the learned adjacencies (BASELINE config 4).

Prints ONE JSON line on rank 0.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

# One BLAS thread: the synthetic-data Cholesky would otherwise leave a pool of spinning OpenBLAS
# workers behind that starve the HIP runtime's submission thread during the timed region
# (observed: 2x longer GPU-side span for the same kernels).  torch.distributed.run sets
# OMP_NUM_THREADS=1 itself; this makes the single-process run behave the same.
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PROFILE_JSON = "r04_rocprof_summary.json"   # falls back to nothing when absent
CENSUS_JSON = "r04_tile_census.json"        # instruction census of the timed kernel's pass (tools/tile_pass_census.py)
HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md), the roofline the streaming edge pass is bound by
FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X fp64 vector spec = 256 CUs x 4 SIMDs x 16 lanes/clk x 2 flop x 2.4 GHz: the tile engine's roof


def algorithmic_bytes_per_iteration(n: int) -> int:
    """read w, read z, write w over E = n(n-1)/2 fp64 edges = 24 E bytes."""
    return 12 * n * (n - 1)


def cpu_baseline(z_scaled: np.ndarray, budget_its=(5, 65)):
    """Oracle (NumPy/SciPy restatement with the reference's CSR strategy) on the host cores:
    (t(65 its) - t(5 its)) / 60 -- about 11 s of CPU work at n = 4096 -- which excludes the one-time operator build."""
    from oracle import logdeg as old

    n = old.n_nodes_from_edges(z_scaled.size)
    ops = old.degree_operator(n)

    def run(k):
        t = time.perf_counter()
        old.learn_log_degrees(z_scaled, 1, 1, maxit=k, tol=0.0, edge_tol=-np.inf, ops=ops)
        return time.perf_counter() - t

    lo, hi = budget_its
    t_lo, t_hi = run(lo), run(hi)
    per_it = (t_hi - t_lo) / (hi - lo)
    return {"value": 1.0 / per_it, "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": f"same n={n} problem, ({hi}-{lo}) iterations of the NumPy/SciPy-CSR oracle, single thread"}


def other_configs(ctx):
    """Bounded measurements of the other BASELINE configs on this GPU (rank 0, N = 1 only; ~3 s together).
    They are reported under config.other_configs and are NOT the graded `value`.

    lgmrf      BASELINE configs[2]: LGMRF n=2048, d=512 (S = X'X/d, X ~ N(0, (L+I)^-1)): node updates 64..191 of the
               block-coordinate descent (lgmrf.py:175-219), timed through the C ABI between HIP events
               (gl_lgmrf_time_node_updates); every QP runs its 100 ADMM steps on this input.
    graphdict  BASELINE configs[4]: GraphDictLog K=8, n=1024, T=4096: ms per primal-dual step, fp64 (what the reference
               computes) and fp32 contractions (what the config names); 50 steps of one fit call behind a fit of the same
               length (a fit captures its step graphs once per history length).
    batched_n256    the 2-D branch (B = 8 graphs of n = 256): us per iteration of the bare loop and of a run to tol.
    window_n256     WindowLogModel.fit end to end, 10 windows of one smooth process on 256 nodes: the second fit of the shape
               (ms_end_to_end) and the first one, which builds the context's stand-in session for the cruise beside it.
    logmodel_n8192  the headline loop on a lattice four times as large (SURVEY section 8 row d quotes the roofline for
               n >= 4096): time per iteration and its fraction of 24 E bytes / 8 TB/s."""
    import numpy as np
    from graph_learning_amd import _lib
    out = {}
    try:
        n, t, k = 1024, 4096, 8
        rng = np.random.default_rng(0)
        x = rng.standard_normal((t, n)) * 0.3
        rs = np.random.RandomState(0)
        c0, w0 = rs.uniform(size=(k, t)), rs.uniform(size=(k, n * (n - 1) // 2))
        for bits in (64, 32):
            with _lib.GraphDictSession(n, t, k, ctx=ctx) as s:
                s.set_precision(bits)
                s.set_signals(x)
                s.set_params(l1_w=0.01, step_c=0.5, step_w=0.5, step_dual=0.5)
                s.set_state(c0, w0)
                s.fit(50, 0.0)     # warm-up of the same length: a fit captures its step graphs once per history length
                t0 = time.perf_counter()
                s.fit(50, 0.0)
                dt = (time.perf_counter() - t0) / 50
                gc, gw = s.time_gradients(5)
            out[f"graphdict_fp{bits}"] = {"workload": "GraphDictLog K=8 n=1024 T=4096 (BASELINE configs[4]), 50 steps of one fit call", "ms_per_step": dt * 1e3,
                                         "steps_per_s": 1 / dt, "grad_c_us": gc, "grad_w_us": gw,
                                         "dtype": "f64" if bits == 64 else "f32 contractions, f64 folds and state",
                                         "contraction_tflops_fma_only": 2.0 * t * (n * (n - 1) // 2) * k / (gw * 1e-6) / 1e12}
    except Exception as exc:  # a leg that cannot run is reported, never silently dropped
        out["graphdict_error"] = repr(exc)
    try:
        from scipy.linalg import solve_triangular
        from graph_learning_amd import LGMRF
        from graph_learning_amd.synth import er_laplacian
        n, d = 2048, 512
        rng = np.random.default_rng(7)
        lap = er_laplacian(n, 6.0 / n, 0.3, 2.0, seed=rng)
        xs = solve_triangular(np.linalg.cholesky(lap + np.eye(n)), rng.standard_normal((n, d)), lower=True, trans="T").T
        cov = xs.T @ xs / d
        from graph_learning_amd.lgmrf import time_node_updates
        # node updates 64 .. 191 of the first cycle (the first ones meet a still diagonal Theta and their QPs stop early),
        # between HIP events on the stream, through a measurement entry point of its own (the public fit cannot be cut short)
        lo, hi = 64, 192
        time_node_updates(cov, 0, 8, ctx=ctx)            # code objects loaded
        dt = time_node_updates(cov, lo, hi - lo, ctx=ctx)
        nodes = hi - lo
        m = n - 1
        out["lgmrf"] = {"workload": "LGMRF n=2048 d=512 (BASELINE configs[2]), block-coordinate node updates, fast path",
                        "node_updates_timed": nodes, "ms_per_node_update": dt * 1e3 / nodes, "node_updates_per_s": nodes / dt,
                        "ms_per_cycle_projected": dt * 1e3 / nodes * n, "admm_steps_per_node_update": 100,
                        # bytes >= 3 x 8 m^2 per node update (read C, write it back, read the ADMM matrix once)
                        "byte_floor_frac_of_hbm_peak": 3 * 8 * m * m / (dt / nodes) / 1e9 / HBM_PEAK_GBS,
                        "bound": "latency: 100 dependent ADMM steps per node update (DESIGN.md section 5)", "dtype": "f64"}
        # the same input with a SPARSE adj_mask (the true support + 1 % extra candidate pairs): every node's QP is gathered
        # to its active set (general path, lgmrf.py:191-202)
        extra = np.triu(rng.uniform(size=(n, n)) < 0.01, 1)
        mask = ((lap != 0) | extra | extra.T | np.eye(n, dtype=bool)).astype(np.float64)
        time_node_updates(cov, 0, 8, adj_mask=mask, ctx=ctx)
        dtm = time_node_updates(cov, lo, hi - lo, adj_mask=mask, ctx=ctx)
        out["lgmrf_masked"] = {"workload": "LGMRF n=2048 d=512, sparse adj_mask (true support + 1 % extra pairs), general path, "
                                           "QPs gathered to their active sets", "node_updates_timed": nodes,
                               "candidates_per_node_mean": float(mask.sum(0).mean() - 1), "ms_per_node_update": dtm * 1e3 / nodes,
                               "ms_per_cycle_projected": dtm * 1e3 / nodes * n, "dtype": "f64"}
    except Exception as exc:
        out["lgmrf_error"] = repr(exc)
    try:  # the 2-D (batched) branch above n = 128: cruise on the tile engine, deciding stretch on batched launches
        from graph_learning_amd import gsp_learn_graph_log_degrees
        n, b = 256, 8
        z = np.random.default_rng(n).uniform(0.2, 3.0, size=(b, n * (n - 1) // 2))
        gsp_learn_graph_log_degrees(z, 1, 1, maxit=100, tol=0.0)      # code objects, the cruise's stand-in session
        t0 = time.perf_counter(); gsp_learn_graph_log_degrees(z, 1, 1, maxit=0); t_setup = time.perf_counter() - t0
        t0 = time.perf_counter(); gsp_learn_graph_log_degrees(z, 1, 1, maxit=2000, tol=0.0); t_bare = time.perf_counter() - t0 - t_setup
        t0 = time.perf_counter()
        _, info = gsp_learn_graph_log_degrees(z, 1, 1, maxit=20000, tol=1e-6, return_info=True)
        t_tol = time.perf_counter() - t0 - t_setup
        out["batched_n256"] = {"workload": "gsp_learn_graph_log_degrees on a (8, E) matrix, n=256 (WindowLogModel / M-step shape), "
                                           "coupled spectral-norm rule (smooth_learning.py:158-159)", "dtype": "f64",
                               "us_per_iteration_bare_loop": t_bare / 2000 * 1e6, "to_tol_1e-6": {"iterations": int(info["n_iter"]),
                               "converged": bool(info["converged"]), "us_per_iteration": t_tol / max(int(info["n_iter"]), 1) * 1e6},
                               "timing": "host wall clock around the call minus the same call with maxit=0 (uploads, set-up)"}
        # WindowLogModel end to end on windows of ONE smooth process (the case whose coupled rule stays undecided by the cheap
        # bounds longest): 10 windows of 256 samples, 256 nodes; distances, theta and the solver on the device
        from graph_learning_amd.temporal import WindowLogModel
        rs = np.random.default_rng(5)
        adj = np.triu((rs.random((n, n)) < 6.0 / n) * rs.uniform(0.3, 2.0, (n, n)), 1)
        adj = adj + adj.T
        chol = np.linalg.cholesky(np.linalg.inv(np.diag(adj.sum(1)) - adj + np.eye(n)))
        xw = (chol @ rs.standard_normal((n, 10 * 256))).T
        # (warm-up = the same fit: the first batch of a new (n, B) shape builds the context's stand-in session for the cruise,
        #  ~0.8 ms that every later fit of that shape does not pay; reported beside the steady figure)
        t0 = time.perf_counter()
        WindowLogModel(window_size=256, avg_degree=6, maxit=20000, tol=1e-4).fit(xw)
        t_first = time.perf_counter() - t0
        t0 = time.perf_counter()
        wm = WindowLogModel(window_size=256, avg_degree=6, maxit=20000, tol=1e-4).fit(xw)
        t_fit = time.perf_counter() - t0
        out["window_n256"] = {"workload": "WindowLogModel.fit, 10 windows of 256 samples of one smooth process on 256 nodes (temporal/smooth.py:12-61), "
                                          "host signals in, (10, E) weights out", "dtype": "f64", "iterations": int(wm.n_iter_),
                              "converged": bool(wm.converged_), "ms_end_to_end": t_fit * 1e3, "ms_first_fit_of_the_shape": t_first * 1e3,
                              "us_per_iteration_all_in": t_fit / max(int(wm.n_iter_), 1) * 1e6}
    except Exception as exc:
        out["batched_error"] = repr(exc)
    try:  # a lattice four times the headline's: what does not fit the register files streams (rounds)
        n = 8192
        e = n * (n - 1) // 2
        z = np.random.default_rng(3).uniform(0.05, 3.0, size=e)
        with _lib.LogDegreeSession(n, ctx) as s:
            s.set_distances(z)
            s.init(1.0, 1.0, 0.5, None)
            kern = s.engine_kernel
            s.iterate_async(16)
            ctx.sync()
            ctx.timer_start()
            s.iterate_async(200)
            us = ctx.timer_stop() * 1e3 / 200
        out["logmodel_n8192"] = {"workload": "LogModel loop n=8192 (E = 33.5 M edges), one 200-iteration call", "engine_kernel": kern,
                                 "us_per_iteration": us, "iterations_per_s": 1e6 / us, "dtype": "f64",
                                 "frac_of_24E_over_hbm_peak": 24.0 * e / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}
    except Exception as exc:
        out["logmodel_n8192_error"] = repr(exc)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--nodes", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=1024)
    ap.add_argument("--avg-degree", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-converge", action="store_true")
    ap.add_argument("--converge-maxit", type=int, default=200000)
    ap.add_argument("--no-other-configs", action="store_true", help="skip the LGMRF / GraphDictLog legs (config.other_configs)")
    ap.add_argument("--comm-timeout", type=float, default=300.0,
                    help="N > 1: wall-clock bound in seconds of every barrier / reduction / gather (a rank that never arrives "
                         "ends the run with an error and a non-zero exit code instead of the launcher's kill)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    from graph_learning_amd import _lib
    from graph_learning_amd.synth import logmodel_problem

    n, d = args.nodes, args.samples
    ctx = _lib.Context(local_rank)
    comm = None
    if world > 1:
        from graph_learning_amd.batch import make_comm

        # RCCL over xGMI (bootstrap via a /tmp file); a run that cannot bring RCCL up fails instead of
        # quietly timing a gloo gather
        comm = make_comm(rank, world, local_rank, ctx, allow_fallback=os.environ.get("GRAPHLEARN_AMD_COMM") == "gloo",
                         timeout_s=args.comm_timeout)
        if getattr(comm, "kind", None) != "rccl" and os.environ.get("GRAPHLEARN_AMD_COMM") != "gloo":
            sys.exit(f"bench.py --gpus {world}: RCCL communicator not available (got {getattr(comm, 'kind', None)})")
    x, _ = logmodel_problem(n, d, seed=rank)

    sess = _lib.LogDegreeSession(n, ctx)
    t0 = time.perf_counter()
    theta = sess.set_signals(x, args.avg_degree)       # A1 + A2, inputs now resident in HBM
    setup_s = time.perf_counter() - t0
    sess.init(1.0, 1.0, 0.5, None)

    # ---- warm-up: W untimed steps as asked, then -- whatever W says -- one untimed call of the SAME length as
    # the timed one, so that the timed call runs the same engine with its code object loaded (a process's first
    # launch of a kernel costs hundreds of microseconds; the driver's `--steps 20 --warmup 5` used to time that)
    if args.warmup > 0:
        sess.iterate_async(args.warmup)
    ctx.sync()
    ctx.timer_start()
    sess.iterate_async(args.steps)
    first_call_us = ctx.timer_stop() * 1e3                # the process's first call of this length (cold), HIP events
    if comm is not None:
        comm.barrier()
    ctx.sync()
    ctx.timer_start()
    t0 = time.perf_counter()
    sess.iterate_async(args.steps)
    gpu_ms = ctx.timer_stop()                            # HIP events on the solver's stream (+ sync)
    wall_s = time.perf_counter() - t0
    if comm is not None:
        comm.barrier()
        wall_s = comm.max_over_ranks(wall_s)
        gpu_ms = comm.max_over_ranks(gpu_ms)

    engine = sess.engine                                 # "resident" (n=4096 default) or "stream"
    engine_kernel = sess.engine_kernel                   # "k_tile" (default), "k_resident" or "k_edge"
    if engine == "resident" and args.steps < (4 if engine_kernel == "k_tile" else 16):   # shorter calls run on the streaming engine
        engine, engine_kernel = "stream", "k_edge"
    # steady state and fixed cost per call: a long call (2000 steps, or the timed one if it is longer) gives the
    # per-iteration rate; what the timed call took beyond steps x that rate is the call's fixed cost
    # (launch, loading the iterate into registers / LDS, storing it back)
    long_steps = max(args.steps, 2000)
    ctx.timer_start()
    sess.iterate_async(long_steps)
    long_us = ctx.timer_stop() * 1e3
    steady_us = long_us / long_steps
    fixed_call_us = max(gpu_ms * 1e3 - args.steps * steady_us, 0.0) if long_steps > args.steps else None
    edge_us = sess.time_edge_pass(200)                   # streaming edge pass alone, HIP events on its stream
    if comm is not None:
        edge_us = comm.max_over_ranks(edge_us)

    gather_s = None
    if comm is not None and not args.no_gather:
        t0 = time.perf_counter()
        gathered = comm.allgather_records(sess, 1e-3, problem_id=rank, n_iter=sess.total_iterations, theta=theta)
        gather_s = comm.max_over_ranks(time.perf_counter() - t0)
        assert gathered.shape == (world, 8 + n * (n - 1) // 2)
        from graph_learning_amd import wire
        assert [wire.unpack(r)["problem_id"] for r in gathered] == list(range(world))

    # time-to-converge (the second half of BASELINE's metric): fresh start, reference stopping rule
    converge = None
    if not args.no_converge:
        sess.init(1.0, 1.0, 0.5, None)
        t0 = time.perf_counter()
        n_it, conv, rel = sess.iterate(args.converge_maxit, 1e-5)
        converge = {"tol": 1e-5, "iterations": n_it, "converged": conv, "seconds": time.perf_counter() - t0}

    if rank == 0:
        its_per_s = world * args.steps / wall_s
        bytes_it = algorithmic_bytes_per_iteration(n)
        iter_us = gpu_ms * 1e3 / args.steps
        # Dominant kernel of the timed region.  Resident engine: ONE launch of k_resident runs all
        # `steps` iterations (HIP events on its stream bracket exactly that launch), so a launch
        # processes steps x 24 E algorithmic bytes.  Streaming engine: k_edge<false>, one launch per
        # iteration, timed alone by gl_logdeg_time_edge_pass.
        if engine == "resident":
            kprefix = engine_kernel
            kernel = ("k_tile (persistent: iterate w in registers, z half in LDS / half streamed, degree partials and node "
                      "values handed over as tagged records)" if engine_kernel == "k_tile" else
                      "k_resident (persistent: edge phase + node phase + stopping rule, iterate in registers/LDS, grid barriers)")
            launch_us, launch_bytes, its_per_launch = gpu_ms * 1e3, bytes_it * args.steps, args.steps
        else:
            kernel, kprefix = "k_edge<false> (edge pass: prox update of all edges + degree sums)", "k_edge"
            launch_us, launch_bytes, its_per_launch = edge_us, bytes_it, 1
        achieved = launch_bytes / (launch_us * 1e-6) / 1e9
        # HBM traffic: PMC counters cannot be read from inside this process; the figure below comes from the committed
        # rocprofv3 --pmc passes of THIS command line (FETCH_SIZE x 2 + WRITE_SIZE per launch, MI355X_MICROARCH.md): the
        # launches of `--steps 20` and of the default `--steps 2000` were both profiled, other step counts are priced by
        # the line through those two points (a call's fixed traffic + its per-iteration traffic)
        traffic, traffic_src = None, None
        try:
            with open(os.path.join(ROOT, "profiles", PROFILE_JSON)) as f:
                prof = json.load(f)
            pts = prof["traffic_bytes_per_launch"].get(kprefix) if n == 4096 else None   # {"20": bytes, "2000": bytes}
            if pts and engine == "resident":
                if str(its_per_launch) in pts:
                    traffic = float(pts[str(its_per_launch)])
                    how = f"the profiled {its_per_launch}-iteration launch"
                else:
                    (s0, b0), (s1, b1) = sorted((int(k), float(v)) for k, v in pts.items())[:2]
                    traffic = b0 + (b1 - b0) / (s1 - s0) * (its_per_launch - s0)
                    how = f"the line through the profiled {s0}- and {s1}-iteration launches"
                traffic_src = f"profile-derived: profiles/{PROFILE_JSON} (rocprofv3 --pmc, FETCH_SIZE x 2 + WRITE_SIZE; {how})"
            elif pts:
                traffic = float(pts["1"])
                traffic_src = f"profile-derived: profiles/{PROFILE_JSON} (rocprofv3 --pmc, FETCH_SIZE x 2 + WRITE_SIZE per launch)"
        except Exception:
            traffic = None
        hbm = {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
               "algorithmic_bytes_per_launch": launch_bytes,
               "definition": "SURVEY 8d: 24 E algorithmic bytes per iteration (read w, read z, write w) / launch time"}
        if engine == "resident" and engine_kernel == "k_tile":
            # The tile engine keeps w in registers and half of z in LDS: those 24 E bytes do not move, so the HBM figure
            # above exceeds 1 and bounds nothing.  What binds the kernel is the fp64 vector pipe: the pass is a chain of fp64
            # VALU instructions (census of the shipped kernel's pass: profiles/r04_tile_census.json, made by
            # tools/tile_pass_census.py from the assembly of the shipped flags), two waves per SIMD.
            #   achieved / frac   REAL flops (a fused multiply-add = 2, add / mul / max = 1) / the 78.6 TFLOP/s fp64 vector spec
            #   valu_issue_frac   fp64 vector instructions issued / the pipe's issue slots (one wave-instruction per 4 cycles
            #                     per SIMD whatever it computes: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3 T lane-ops/s)
            # A census that was not made from THIS gl_tile.hip (sha of the source) prices another kernel: the figures are null then.
            census, census_ok = None, False
            try:
                import hashlib
                with open(os.path.join(ROOT, "profiles", CENSUS_JSON)) as f:
                    census = json.load(f)
                src = b"".join(open(os.path.join(ROOT, "graph_learning_amd", "csrc", f), "rb").read() for f in census["sources"])
                census_ok = hashlib.sha256(src).hexdigest()[:16] == census["source_sha16"]
            except Exception:
                census = None
            streamed_bytes = 2 * n * (n - 1)      # half of z per iteration (4 E); records ~3 MB on top
            roof = {"bound": "valu_f64", "unit": "TFLOP/s", "peak": FP64_VALU_PEAK_TFLOPS, "achieved": None, "frac": None,
                    "valu_issue_frac": None}
            if census is not None and census_ok:
                nb = (n + 63) // 64
                # row pairs the launch really runs per iteration: 32 per bulk tile + the 32 of every diagonal tile (dealt out
                # as fringe groups; whole rows are computed, the entries outside i < j are masked) = 64 512 + 2 048 at n = 4096
                row_pairs = nb * (nb + 1) // 2 * 32
                per_rp = census["per_row_pair"]
                lanes = row_pairs * 64
                flops = per_rp["flop_f64"] * lanes * its_per_launch
                roof["achieved"] = flops / (launch_us * 1e-6) / 1e12
                roof["frac"] = roof["achieved"] / FP64_VALU_PEAK_TFLOPS
                roof["valu_issue_frac"] = per_rp["valu_f64"] * lanes * its_per_launch / (launch_us * 1e-6) / (FP64_VALU_PEAK_TFLOPS * 0.5e12)
                roof["flop_convention"] = ("achieved = real fp64 flops of the pass (fused multiply-add 2, add / max 1) x row pairs "
                                          "executed (diagonal tiles' masked halves included); valu_issue_frac = share of the fp64 "
                                          "pipe's issue slots (every instruction takes one slot)")
                roof["steady_state_frac"] = per_rp["flop_f64"] * lanes / (steady_us * 1e-6) / 1e12 / FP64_VALU_PEAK_TFLOPS
                roof["steady_state_valu_issue_frac"] = per_rp["valu_f64"] * lanes / (steady_us * 1e-6) / (FP64_VALU_PEAK_TFLOPS * 0.5e12)
            roof["census"] = {"source": f"profiles/{CENSUS_JSON}", "matches_this_build": census_ok,
                              "per_row_pair": census["per_row_pair"] if census is not None else None}
            roof.update({"traffic": traffic, "traffic_source": traffic_src, "kernel": kernel,
                         "iterations_per_launch": its_per_launch, "avg_launch_us": launch_us,
                         "streamed_bytes_per_iteration": streamed_bytes,
                         "streamed_frac_of_hbm_peak": streamed_bytes * its_per_launch / (launch_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "hbm_algorithmic": hbm,
                         "note": ("bound = the resource that binds this kernel: fp64 vector-instruction issue (plus one all-to-all "
                                  "hand-off chain per iteration, pure latency: DESIGN.md section 3a).  `hbm_algorithmic` keeps "
                                  "SURVEY 8d's figure (24 E bytes per iteration / time; > 1 because the iterate never leaves the "
                                  "register files); `traffic` = what really crossed the fabric (PMC)")})
        else:
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic, "kernel": kernel, "iterations_per_launch": its_per_launch,
                    "algorithmic_bytes_per_launch": launch_bytes, "avg_launch_us": launch_us, "traffic_source": traffic_src}
        out = {
            "metric": "primal-dual iterations/sec, LogModel n=4096 d=1024",
            "value": its_per_s,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall_s * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"LogModel fixed-iteration loop n={n} d={d} avg_degree={args.avg_degree} "
                                   f"(BASELINE configs[1]); one independent problem per GPU",
                       "edges": n * (n - 1) // 2, "theta": theta, "setup_s_distances_theta": setup_s,
                       "gather_s": gather_s, "comm": getattr(comm, "kind", None), "gpu_us_per_iteration": iter_us,
                       "engine": engine, "engine_kernel": engine_kernel, "streaming_edge_pass_us": edge_us,
                       "first_call_us": first_call_us, "fixed_call_us": fixed_call_us,
                       "steady_state": {"steps": long_steps, "us_per_iteration": steady_us,
                                        "iterations_per_s": 1e6 / steady_us,
                                        "frac_of_hbm_peak_algorithmic": bytes_it / (steady_us * 1e-6) / 1e9 / HBM_PEAK_GBS},
                       "time_to_converge": converge},
            "roofline": roof,
        }
        if not args.no_other_configs and world == 1:
            out["config"]["other_configs"] = other_configs(ctx)
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is reported at N = 1 only
            z = sess.get_distances() * theta
            out["cpu_baseline"] = cpu_baseline(z)
        print(json.dumps(out), flush=True)
    sess.close()
    if comm is not None:
        comm.close()
    ctx.close()


if __name__ == "__main__":
    main()
