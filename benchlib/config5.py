"""benchlib.config5 -- BASELINE config 5: the NMFk sweep k = 2..16 x 20 perturbations, HALS/FRO on bf16-stored X (bench.py --config 5), and
the reference's own swim example reported beside the headline."""
import json
import os
import sys
import time

from .common import CONFIGS, PEAK_CLOCK_GHZ, PEAK_FP32_MFMA_TFLOPS, PEAK_HBM_GBS, ROOT, event_time_ms, parse_grid, pmc_traffic, rccl_record
from .cpu_baseline import _cpu_hals_rank, _cpu_pool, host_cpu


def swim_example(npz, dev):
    """One warm-up sweep + one timed sweep of the reference's swim example (see tools/swimbench.py)."""
    import contextlib
    import tempfile
    import numpy as np
    import torch
    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.pyDNMFk import PyNMFk
    from pydnmfk_amd.utils import parse
    X = torch.from_numpy(np.ascontiguousarray(np.load(npz)["A"].astype(np.float32))).to(dev)
    itr = 5000

    def sweep():
        comms = MPI_comm(None, 1, 1)
        q = parse()
        q.size, q.rank, q.comm, q.p_r, q.p_c = 1, 0, comms, 1, 1
        q.row_comm, q.col_comm, q.comm1 = comms.cart_1d_row(), comms.cart_1d_column(), comms.comm
        q.fpath, q.fname, q.ftype = "../data/", "swim", "mat"
        q.start_k, q.end_k, q.sill_thr, q.itr, q.init = 14, 18, 0.6, itr, "rand"
        q.noise_var, q.verbose, q.norm, q.method, q.checkpoint = 0.016, False, "kl", "mu", False
        q.prune, q.rng, q.results_path = False, "device", tempfile.mkdtemp(prefix="dnmf_swim_") + "/"
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(sys.stderr):
            nopt = PyNMFk(X, factors=None, params=q).fit()
        torch.cuda.synchronize()
        return int(nopt), time.perf_counter() - t0

    sweep()
    nopt, secs = sweep()
    steps = 5 * 21 * itr
    return {"workload": "NMFk on swim %d x %d (tests/golden/data_swim.npz = the reference's data/swim.mat): KL/MU, k = 14..18, 20 perturbations + the "
                        "regression fit per k, %d iterations each, one rank, device resident" % (X.shape[0], X.shape[1], itr),
            "nopt": nopt, "known_answer": 16, "seconds_per_sweep": secs, "kl_steps_per_sec": steps / secs, "fits_per_sec": 105 / secs,
            "note": "every fit is ONE persistent kernel (csrc/dnmf_small.h): slab of A, rows of W and H in LDS across the steps"}


def run_config5(a, job):
    """BASELINE config 5: the NMFk sweep k = 2..16, 20 perturbations, HALS / Frobenius on bf16-STORED X (reference
    pyDNMFk.py:169-258 over dist_nmf.py:873-934, clustering dist_clustering.py:84-160).  X = 65536 x 4096 of planted rank 6
    (+ 1 % noise), per-rank blocks of the 1 x N grid (the grid for HALS: W is replicated there, its column norms are local).
    A "step" is one WHOLE sweep: (end_k - start_k + 1) x perturbations fits of `itr` HALS iterations + the regression fit,
    clustering and statistics of every k, device resident (params.rng = 'device'); value = fits per second."""
    import contextlib
    import torch
    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.engine import ops_for
    from pydnmfk_amd.pyDNMFk import PyNMFk
    from pydnmfk_amd.utils import determine_block_params, parse

    world, rank, dev = job.world, job.rank, job.dev
    # N GPUs share the sweep the MI355X way unless a grid is asked for: every GPU holds the WHOLE X (0.5 GB of bf16 here, 288 GB of
    # HBM) and fits its share of the perturbations as one-rank problems -- no exchange inside a fit (params.nmfk_split =
    # 'perturbations', pydnmfk_amd/pyDNMFk.py); --grid RxC cuts X into the reference's blocks instead (every fit on all ranks)
    shared = world > 1 and not a.grid
    m, n = a.m, a.n
    if shared:
        from pydnmfk_amd.dist_comm import COMM_WORLD, SoloGrid
        p_r = p_c = 1
        comms, whole = SoloGrid(rank), COMM_WORLD()
        s, e = determine_block_params(0, (1, 1), (m, n)).determine_block_index_range_asymm()
    else:
        p_r, p_c = parse_grid(a.grid, world, (1, world))
        comms = MPI_comm(None, p_r, p_c)
        whole = comms.comm
        s, e = determine_block_params(rank, (p_r, p_c), (m, n)).determine_block_index_range_asymm()
    # planted rank 6, identifiable (the recipe of tests/test_gpu_nmfk_sweep.py at this size): six Gaussian bumps along the rows,
    # sparse uniform mixing, 0.5 % noise -- the sweep must come back with estimated_k = 6
    g = torch.Generator(device=dev)
    g.manual_seed(7)                                                   # the planted factors: the same on every rank
    x = torch.arange(m, device=dev, dtype=torch.float32)[:, None]
    cen = torch.linspace(0.075 * m, m - 0.075 * m, 6, device=dev)[None, :]
    Wt = torch.exp(-(x - cen) ** 2 / (2 * (0.044 * m) ** 2))
    Ht = torch.rand(6, n, device=dev, generator=g) * (torch.rand(6, n, device=dev, generator=g) < 0.7)
    g.manual_seed(1234 + (0 if shared else rank))                      # (shared sweep: the same X on every rank)
    X = (Wt[s[0]:e[0] + 1] @ Ht[:, s[1]:e[1] + 1])
    X += 0.005 * torch.rand(X.shape, device=dev, generator=g)
    del x, cen
    Xb = X.to(torch.bfloat16)
    del X, Wt, Ht
    import tempfile
    tmp = tempfile.mkdtemp(prefix="dnmf_c5_") if rank == 0 else None
    tmp = whole.bcast(tmp, root=0) if world > 1 else tmp

    def params(start_k, end_k, pert, itr):
        q = parse()
        q.comm1, q.comm, q.p_r, q.p_c = (whole if shared else comms.comm), comms, p_r, p_c
        q.row_comm, q.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
        q.size, q.rank = world, rank
        if shared:
            q.nmfk_split = "perturbations"
        q.norm, q.method, q.init, q.itr, q.verbose, q.prune = "fro", "hals", "rand", itr, False, False
        q.start_k, q.end_k, q.step_k, q.fname, q.checkpoint = start_k, end_k, 1, "c5", False
        q.perturbations, q.noise_var, q.sampling, q.sill_thr = pert, 0.03, "uniform", 0.8
        q.precision, q.results_path, q.timing_stats, q.rng = "bfloat16", tmp + "/", False, "device"
        q.nmfk_batch = True if a.nmfk_batch == "auto" else int(a.nmfk_batch)
        if a.fit_loop == "python":
            q.fit_loop = "python"
        return q

    nopt = [None]

    def sweep(_i, small=False):
        q = params(2, 3, 2, 10) if small else params(a.start_k, a.end_k, a.perturbations, a.itr)
        with contextlib.redirect_stdout(sys.stderr):                   # PyNMFk reports progress on stdout: the JSON line stays alone there
            nopt[0] = PyNMFk(Xb, factors=None, params=q).fit()

    for i in range(max(1, a.warmup)):
        sweep(i, small=True)                                           # loads every kernel class of the sweep; not a full sweep
    elapsed = job.timed(a.steps, sweep)
    nk = a.end_k - a.start_k + 1
    fits = nk * a.perturbations
    ms = elapsed / a.steps * 1e3
    out = None
    if rank == 0:
        out = {
            "metric": "nmfk_fits_per_sec", "value": fits * a.steps / elapsed, "unit": "fits/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16 storage of X, f32 arithmetic and factors", "data": "synthetic",
            "config": {"workload": "NMFk k=%d..%d x %d perturbations, HALS/FRO %d iterations per fit, X=%dx%d stored bf16 (planted rank 6: estimated_k must be 6), "
                                   "%d x %d grid (%s); a step = one whole sweep (fits + regression fit + clustering per k)" % (
                                       a.start_k, a.end_k, a.perturbations, a.itr, m, n, p_r, p_c, CONFIGS[5]["label"]),
                       "m": m, "n": n, "k_range": [a.start_k, a.end_k], "perturbations": a.perturbations, "itr": a.itr,
                       "block_per_gpu": [e[0] - s[0] + 1, e[1] - s[1] + 1],
                       "parallelism": "single GPU" if world == 1 else (
                           "perturbations over %d ranks, the whole X on every GPU (no exchange inside a fit; factors gathered per k over %s)" % (
                               world, "RCCL" if a.backend == "nccl" else a.backend) if shared else
                           "%d x %d blocks of X over %s" % (p_r, p_c, "RCCL" if a.backend == "nccl" else a.backend))},
            "fits_per_step": fits, "estimated_k": int(nopt[0]), "seconds_per_sweep": elapsed / a.steps,
            "nmfk_batch": a.nmfk_batch, "fit_loop": a.fit_loop,
            "hals_iterations_per_sec": (fits + nk) * a.itr * a.steps / elapsed,
        }
    if not a.no_kernel_timing:
        # the kernel the sweep spends most of its time in: A H^T on the bf16-stored block at the top rank of the sweep (HBM bound)
        ops = ops_for(None)
        kk = min(16, a.end_k)
        m_l, n_l = Xb.shape
        Hk = torch.rand(kk, n_l, device=dev, generator=g)
        Wk = torch.rand(m_l, kk, device=dev, generator=g)
        V = torch.empty(m_l, kk, device=dev)
        Y = torch.empty(kk, n_l, device=dev)
        t_nt, _ = event_time_ms(lambda: ops.aht(Xb, Hk, V), reps=20, warm=5)
        t_tn, _ = event_time_ms(lambda: ops.wta(Xb, Wk, Y), reps=20, warm=5)
        if rank == 0:
            by = 2.0 * m_l * n_l

            def entry(kernel, ms_):
                ach = by / ms_ / 1e6
                return {"kernel": kernel, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
                        "traffic": None, "bytes_per_launch": by, "ms_per_launch": ms_, "note": "algorithmic bytes = one read of the bf16 block"}
            out["roofline"] = entry("nt16_kernel<bf16 X> (dnmf_aht_bf16a, k=%d: A H^T of the HALS W phase, dist_nmf.py:884)" % kk, t_nt)
            out["rooflines"] = [out["roofline"], entry("tn16_kernel<bf16 X> + reduce_partials (dnmf_wta_bf16a, k=%d: W^T A of the H phase, dist_nmf.py:903)" % kk, t_tn)]
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cores, model = host_cpu()
        P = min(8, cores)
        ks = sorted({a.start_k, (a.start_k + a.end_k) // 2, a.end_k})
        got = _cpu_pool(_cpu_hals_rank, [(r, m, n // P, ks, 3) for r in range(P)])
        if got is None:
            out["cpu_baseline"] = {"value": None, "unit": "fits/s", "cores": P, "kind": "port", "sample": "FAILED: a CPU rank died or timed out"}
        else:
            import numpy as np
            t_k = {k_: max(v[k_] for v in got.values()) for k_ in ks}                      # iteration time = slowest rank
            t_all = np.interp(np.arange(a.start_k, a.end_k + 1), ks, [t_k[k_] for k_ in ks])
            t_sweep = float(np.sum(t_all) * (a.perturbations + 1) * a.itr)
            out["cpu_baseline"] = {
                "value": fits / t_sweep, "unit": "fits/s", "cores": P, "kind": "port", "host_cores": cores, "host_cpu": model,
                "seconds_per_sweep": t_sweep,
                "sample": "oracle fro_hals_step_local (float32: numpy has no bfloat16) in the reference's process model: %d processes x 1 BLAS "
                          "thread = the ranks of a 1 x %d grid, each on its %d x %d column block; 1 warm-up + 3 timed HALS iterations at k = %s "
                          "(%s s, slowest rank), interpolated over k = %d..%d and multiplied by (%d perturbations + 1 regression fit) x %d "
                          "iterations = %.0f s per sweep; no exchange, no clustering counted" % (
                              P, P, m, n // P, ks, [round(t_k[k_], 4) for k_ in ks], a.start_k, a.end_k, a.perturbations, a.itr, t_sweep)}
    import shutil
    if rank == 0:
        shutil.rmtree(tmp, ignore_errors=True)
    return out
