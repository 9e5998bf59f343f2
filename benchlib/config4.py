"""benchlib.config4 -- BASELINE config 4: MU/KL 131072 x 65536, k = 128 on the 2D grid (bench.py --config 4)."""
import json
import os
import sys
import time

from .common import CONFIGS, PEAK_CLOCK_GHZ, PEAK_FP32_MFMA_TFLOPS, PEAK_HBM_GBS, ROOT, event_time_ms, parse_grid, pmc_traffic, rccl_record
from .cpu_baseline import _cpu_kl_rank, _cpu_pool, host_cpu


def run_config4(a, job):
    """BASELINE config 4: MU/KL, X = 131072 x 65536 fp32, k = 128, on the p_r x p_c grid of the job (8 ranks: 4 x 2; reference
    dist_nmf.py:268-407 over the grid of dist_comm.py:16-56).  Strong scaling: the global X is fixed.  A step = one
    nmf_algorithms_2D(...).update() (1D class on 1 x 1 / N x 1 grids), clamp on every 10th; `--emulate-ranks R` = this
    process is rank 0 of the R-rank grid on one GPU, its sub-communicators EmulatedGroup objects (dist_comm.py)."""
    import torch
    import torch.distributed as dist
    from pydnmfk_amd.dist_comm import EmulatedGroup, MPI_comm, NullExchange
    from pydnmfk_amd.dist_nmf import nmf_algorithms_1D, nmf_algorithms_2D
    from pydnmfk_amd.engine import ops_for
    from pydnmfk_amd.utils import determine_block_params, parse

    world, rank, dev, emu = job.world, job.rank, job.dev, job.emu
    nr = emu or world
    p_r, p_c = parse_grid(a.grid, nr, {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (4, 2)}.get(nr, (nr, 1)))
    m, n, k = a.m, a.n, a.k
    two_d = p_r > 1 and p_c > 1
    p = parse()
    if emu:
        base = MPI_comm(None, 1, 1)
        p.comm1, p.comm = base.comm, base
        p.row_comm, p.col_comm = EmulatedGroup(base.comm, p_r), EmulatedGroup(base.comm, p_c)
    else:
        comms = MPI_comm(None, p_r, p_c)
        p.comm1, p.comm, p.row_comm, p.col_comm = comms.comm, comms, comms.cart_1d_row(), comms.cart_1d_column()
    p.p_r, p.p_c, p.k, p.m, p.n = p_r, p_c, k, m, n
    p.norm, p.method, p.W_update, p.eps, p.gemm = a.norm, "mu", True, 1.1920929e-07, "fp32"
    ops = ops_for(p)
    i, j = divmod(rank, p_c)
    m_l, n_l = determine_block_params(rank, (p_r, p_c), (m, n)).determine_block_shape_asymm()
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    A = torch.rand(m_l, n_l, device=dev, generator=g)
    if two_d:      # the rank's SLICES of the factors (utils.py:99-103): W_ij = rows of W_i split over the p_c ranks of grid row i
        m_w = determine_block_params(j, (p_c, 1), (m_l, k)).determine_block_shape_asymm()[0]
        n_h = determine_block_params(i, (1, p_r), (k, n_l)).determine_block_shape_asymm()[1]
    else:          # 1D (and 1 x 1): the sharded factor whole, the other one replicated
        m_w, n_h = m_l, n_l
    g.manual_seed(4321 + rank)
    W = torch.rand(m_w, k, device=dev, generator=g)
    g.manual_seed(99 + (rank if two_d else (j if p_c > 1 else 0)))
    H = torch.rand(k, n_h, device=dev, generator=g)
    if not two_d and world > 1:
        H = p.comm1.bcast(H, root=0) if p_c == 1 else H
        W = p.comm1.bcast(W, root=0) if p_r == 1 else W
    cls = nmf_algorithms_2D if two_d else nmf_algorithms_1D

    if getattr(a, "overlap_2d", False) and two_d:
        p.overlap_2d = True

    def step(it, params=p):
        if two_d:
            cls(A, W, H, params=params).update(clamp=(it % 10 == 0), more=True)
        else:
            cls(A, W, H, params=params).update(clamp=(it % 10 == 0))

    multi = world > 1 or emu > 1
    mg = None
    if multi:
        mg = {"rccl_ranks_seen": job.rccl_ranks_seen, "backend": a.backend, "grid": [p_r, p_c],
              "rccl": rccl_record() if a.backend == "nccl" else None}
        if emu:
            mg["emulated"] = ("rank 0 of a %d x %d grid on a single GPU: its own %d x %d block, real kernels and launches, the collectives of the "
                              "step issued on one-rank groups -- no wire time -- by torch.distributed between the launches (EmulatedGroup) or inside "
                              "the library (dnmf_comm_create_emulated: one call per step)" % (p_r, p_c, m_l, n_l))
        modes = ["torch"] if a.exchange in ("auto", "torch") else []
        if emu and a.backend == "nccl" and a.exchange in ("auto", "native"):
            # the library-sequenced step of the emulated member: the same kernels, every collective issued inside libdnmf_hip.so on a
            # one-rank RCCL communicator -- what `params.exchange = 'native'` runs on the real grid, minus the wire
            try:
                from pydnmfk_amd.engine import NativeComm
                p._native_comm = NativeComm.emulated(p_r, p_c, 0)
                p.exchange = "native"
                step(1)
                torch.cuda.synchronize()
                if p._native_comm.steps < 1:
                    raise RuntimeError("the step did not run inside the library")
                modes.append("native")
            except Exception as exc:  # noqa: BLE001
                mg["native_exchange_unavailable"] = repr(exc)
            p.exchange = "torch"
        if world > 1 and a.backend == "nccl" and a.exchange in ("auto", "native"):
            p.exchange = "native"
            from pydnmfk_amd.engine import native_comm_for
            ok = 1
            try:
                ok = int(native_comm_for(p) is not None)           # collective-safe: raises / returns None on every rank together
                if ok:
                    step(1)
                    torch.cuda.synchronize()
            except Exception as exc:  # noqa: BLE001
                ok, mg["native_exchange_unavailable"] = 0, repr(exc)
            if int(-job.max_over_ranks(-float(ok))):
                modes.append("native")
            p.exchange = "torch"
        if not modes:
            sys.exit("bench.py: --exchange %s is not available (backend %s%s)" % (a.exchange, a.backend, ", emulated" if emu else ""))
        if len(modes) > 1:
            ab, nab = {}, max(3, min(10, a.steps))
            for mode in modes:
                p.exchange = mode
                for it in range(2):
                    step(it)
                ab[mode] = job.timed(nab, step) / nab * 1e3
            mg["exchange_ab_ms_per_step"], mg["exchange_ab_steps"] = ab, nab
            p.exchange = min(ab, key=ab.get)
        else:
            p.exchange = modes[0]
        mg["exchange_used"] = p.exchange
        mg["overlap_2d"] = bool(getattr(p, "overlap_2d", False)) and p.exchange == "torch"

    for it in range(a.warmup):
        step(it)
    elapsed = job.timed(a.steps, step)
    assert torch.isfinite(W).all() and torch.isfinite(H).all()
    ms = elapsed / a.steps * 1e3

    if multi:      # the same step with every exchange stubbed out (timing only): this rank's compute
        pn = parse()
        pn.__dict__.update(vars(p))
        native_used = p.exchange == "native"
        if native_used:
            p._native_comm.set_null_exchange(True)
        else:
            pn.comm1, pn.row_comm, pn.col_comm = NullExchange(p.comm1), NullExchange(p.row_comm), NullExchange(p.col_comm)
        keep = (W.clone(), H.clone())
        for it in range(2):
            step(it, pn)
        nn = max(3, min(20, a.steps))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(nn):
            step(it, pn)
        torch.cuda.synchronize()
        mine = (time.perf_counter() - t0) / nn * 1e3
        if native_used:
            p._native_comm.set_null_exchange(False)
        per = [mine]
        if world > 1:
            per = [None] * world
            dist.all_gather_object(per, mine)
        W.copy_(keep[0]); H.copy_(keep[1])
        del keep
        mg["compute_only_ms_per_rank"], mg["compute_only_ms"] = per, max(per)
        mg["full_step_ms"], mg["exposed_comm_ms"] = ms, ms - max(per)
        kb = 4 * k
        if two_d:
            mg["exchange_bytes_per_step"] = {"allgather_H_recv": kb * n_l, "allgather_W_recv": 2 * kb * m_l, "reduce_scatter_W_send": kb * m_l,
                                             "reduce_scatter_H_send": kb * n_l, "allreduce_k_vectors": 2 * kb}
        else:      # 1D grid: the sharded factor's phase is local, the replicated one's product is allreduced (dist_nmf.py:776-869)
            mg["exchange_bytes_per_step"] = {"allreduce_product": kb * (n_l if p_c == 1 else m_l), "allreduce_k_vectors": 2 * kb}
        mg["note"] = ("compute_only = the same step with every exchange stubbed out (NullExchange / the library's null mode; timing only); "
                      "exposed_comm_ms = full step - slowest rank's compute-only step")

    out = None
    flops_iter = 8.0 * m * n * k + 6.0 * (m + n) * k                  # SURVEY 8d, whole job
    if rank == 0:
        out = {
            "metric": "mu_iterations_per_sec", "value": a.steps / elapsed, "unit": "iter/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "MU/%s X=%dx%d fp32 k=%d, %s grid p_r=%d p_c=%d (%s)%s" % (
                a.norm.upper(), m, n, k, "2D" if two_d else "1D", p_r, p_c, CONFIGS[4]["label"],
                " -- EMULATED: one rank's block on one GPU, not a whole-job number" if emu else ""),
                "m": m, "n": n, "k": k, "block_per_gpu": [m_l, n_l], "gemm": "fp32",
                "parallelism": ("%d x %d blocks of X; per step: allreduce of k-vectors, allgather of the H / W slices in the size-p_r / size-p_c "
                                "groups, reduce-scatter of U H^T / W^T U over %s" % (p_r, p_c, "RCCL" if a.backend == "nccl" else a.backend + " (host staged)"))
                if multi else "single GPU"},
            "step_tflops_per_gpu": flops_iter / nr / (ms * 1e-3) / 1e12,
            "step_mfma_frac": flops_iter / nr / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            "step_algorithmic_hbm_gbs_per_gpu": 4.0 * m_l * n_l / (ms * 1e-3) / 1e9,          # ONE read of the block (the step makes two)
        }
        if mg is not None:
            out["multi_gpu"] = mg

    if not a.no_kernel_timing:
        # the two KL products in situ (HIP events on the stream the launches go to), on this rank's block with the gathered factors
        W_i = torch.rand(m_l, k, device=dev, generator=g)
        H_j = torch.rand(k, n_l, device=dev, generator=g)
        V = torch.empty(m_l, k, device=dev)
        Y = torch.empty(k, n_l, device=dev)
        t_uht, _ = event_time_ms(lambda: ops.kl_uht(A, W_i, H_j, p.eps, V), reps=10, warm=3)
        t_wtu, _ = event_time_ms(lambda: ops.kl_wtu(A, W_i, H_j, p.eps, Y), reps=10, warm=3)
        del W_i, H_j, V, Y
        if rank == 0:
            fl = 4.0 * m_l * n_l * k
            kt = max(1, (k + 31) // 32)
            pipe = k > 16 and m_l >= 128 and n_l % 32 == 0

            def entry(kernel, ms_, role):
                ach = fl / ms_ / 1e9
                e = {"kernel": kernel, "bound": "mfma", "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": ach / PEAK_FP32_MFMA_TFLOPS, "traffic": None, "flops_per_launch": fl, "ms_per_launch": ms_,
                     "algorithmic_bytes_per_launch": 4.0 * m_l * n_l}
                tr = pmc_traffic(role, "kl") if (m_l, n_l, k) == (32768, 32768, 128) else None
                if tr is not None:
                    e["traffic"] = tr["bytes"]
                    e["traffic_note"] = ("HBM bytes per launch from the committed PMC pass %s, kernel %s (FETCH_SIZE x2 + WRITE_SIZE); a "
                                         "constant of that profile, not a live counter" % (tr["source"], tr["kernel"]))
                return e

            r_uht = entry(("kl_uht_pipe_kernel<KT=%d>" if pipe else "kl_uht_kernel<KT=%d>") % kt +
                          " + reduce_partials (dnmf_kl_uht: U H^T, the W phase's product, dist_nmf.py:806,810)", t_uht,
                          "kl_uht_pipe_kernel<4" if pipe else "kl_uht_kernel<4")
            r_wtu = entry("kl_wtu_kernel<KT=%d> + reduce_partials (dnmf_kl_wtu: W^T U, the H phase's product, dist_nmf.py:806,808)" % kt,
                          t_wtu, "kl_wtu_kernel<4")
            out["roofline"], out["rooflines"] = r_uht, [r_uht, r_wtu]
            out["kernels"] = {"dnmf_kl_uht": {"ms": t_uht, "tflops": fl / t_uht / 1e9}, "dnmf_kl_wtu": {"ms": t_wtu, "tflops": fl / t_wtu / 1e9},
                              "rest_of_step_ms": ms - t_uht - t_wtu}

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        # the reference's process model for this config: the 4 x 2 grid = 8 ranks x 1 BLAS thread, each with a 32768 x 32768 block;
        # bounded sample = a row slab of every rank's block (the KL step is linear in the rows)
        cores, model = host_cpu()
        P = min(8, cores)
        pr_, pc_ = (4, 2) if P == 8 else (P, 1)
        mb, nb = m // pr_, n // pc_
        rows_s = max(32, min(mb, int(2.4e10 / (8.0 * nb * k))))            # ~1-2 s per slab step on one core
        got = _cpu_pool(_cpu_kl_rank, [(r, rows_s, nb, k, 2) for r in range(P)])
        if got is None:
            out["cpu_baseline"] = {"value": None, "unit": "iter/s", "cores": P, "kind": "port", "sample": "FAILED: a CPU rank died or timed out"}
        else:
            t_slab = max(got.values())
            t_it = t_slab * (mb / rows_s)
            out["cpu_baseline"] = {
                "value": 1.0 / t_it, "unit": "iter/s", "cores": P, "kind": "port", "host_cores": cores, "host_cpu": model,
                "seconds_per_iter": t_it, "gflops_whole_job": flops_iter / t_it / 1e9,
                "sample": "oracle kl_mu_step_local in the reference's process model: %d processes x 1 BLAS thread = the ranks of a %d x %d grid, "
                          "each on a %d x %d row slab of its %d x %d block, k=%d, 1 warm-up + 2 timed steps; slab step %.2f s (slowest rank) x %.0f "
                          "= %.1f s per iteration (the step is linear in the rows); no exchange simulated" % (
                              P, pr_, pc_, rows_s, nb, mb, nb, k, t_slab, mb / rows_s, t_it)}
    if multi and getattr(p, "_native_comm", None) is not None:
        p._native_comm.close()
    return out
