#!/usr/bin/env python3
"""bench.py -- MU iterations/sec of the MI355X update engine on BASELINE.json's headline workload.

Workload (config.workload): synthetic dense fp32 X = 262144 x 8192, k = 64, MU / Frobenius, 1D row grid
p_r = N, p_c = 1 (BASELINE config 3).  The global X is fixed, so N GPUs each hold m/N rows: strong scaling.
A "step" is one full MU iteration exactly as PyNMF.fit runs it (W update, H update with the new W, the
allreduce of [W^T A | W^T W] when N > 1, and the clamp on every 10th step), X already resident in HBM.

  python bench.py                      # N=1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline     : the dominant kernel (fused A H^T + W update, one launch), algorithmic flops / HIP-event time
  kernels      : the same measurement for every kernel class of the step
  cpu_baseline : the numpy oracle (port of the reference's path) timed on the host, bounded sample (N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec
# measured on an MI355X with tools/ceilbench.hip (profiles/r01d_ceilings.txt): what the matrix pipe sustains while the
# contraction's HBM stream (128 B per 32x32x2 MFMA at k = 64) runs beside it, and plain streaming reads
MEASURED_MFMA_WITH_STREAM_TFLOPS = {128: 138.9, 64: 126.2, 32: 108.4}
MEASURED_STREAM_GBS = 6760.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", dest="m", type=int, default=262144)
    ap.add_argument("--cols", dest="n", type=int, default=8192)
    ap.add_argument("--rank", dest="k", type=int, default=64)
    ap.add_argument("--norm", default="fro")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for debugging)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    return ap.parse_args()


def event_time_ms(fn, reps=5, warm=2):
    import torch
    for _ in range(warm):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in evs:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return sum(ts) / len(ts), ts[0]


def cpu_baseline(n, k, m_full):
    """The oracle's single-rank MU/FRO step (same numpy calls as dist_nmf.py:716-751) on a row slab, 1 BLAS thread
    (the reference forces OMP_NUM_THREADS=1 per rank, main.py:3)."""
    import numpy as np
    from oracle import nmf_oracle as orc
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        threadpool_limits = None
    m_s = 8192
    rs = np.random.RandomState(0)
    A = rs.rand(m_s, n).astype(np.float32)
    W = rs.rand(m_s, k).astype(np.float32)
    H = rs.rand(k, n).astype(np.float32)
    eps = np.finfo(np.float32).eps

    def run(iters):
        t0 = time.perf_counter()
        for _ in range(iters):
            orc.fro_mu_step_local(A, W, H, eps)
        return (time.perf_counter() - t0) / iters

    def timed():
        # bounded sample of about 10 s of CPU work: one warm-up step sizes the number of timed steps
        t1 = run(1)
        steps = int(min(100, max(6, round(10.0 / max(t1, 1e-3)))))
        return run(steps), steps

    if threadpool_limits is not None:
        with threadpool_limits(limits=1):
            t, steps = timed()
    else:
        t, steps = timed()
    t_full = t * (m_full / m_s)
    return {"value": 1.0 / t_full, "unit": "iter/s", "cores": 1, "kind": "port",
            "sample": "oracle fro_mu_step_local on a %dx%d row slab (1/%d of X), k=%d, 1 BLAS thread, %d timed steps; "
                      "%.3f s/step on the slab, scaled by rows to the full X" % (m_s, n, m_full // m_s, k, steps, t),
            "slab_seconds_per_step": t,
            "gflops": (4.0 * m_s * n * k + 4.0 * (m_s + n) * k * k) / t / 1e9}


def pmc_traffic_bytes():
    """HBM bytes per launch of the fused NT kernel from the committed PMC pass (profiles/*_pmc.json; collected by
    tools/collect_profiles.sh in separate --pmc runs, FETCH_SIZE doubled per MI355X_MICROARCH.md).  None if absent."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        # the k = 64 (KT = 2) NT instantiations of the step: the gram H H^T launch only reads H, the fused
        # A H^T + W-update launch streams A -> it is the one with the most HBM bytes (template argument lists
        # change between rounds, so the kernel is picked by role, not by its exact name)
        cand = [(c["hbm_bytes"], name) for name, c in d.items()
                if name.startswith("nt_kernel<2,") and isinstance(c, dict) and "hbm_bytes" in c]
        if cand:
            by, name = max(cand)
            best = {"bytes": by, "source": os.path.basename(f), "kernel": name}
    return best


def main():
    a = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py: --gpus %d needs a torch.distributed.run launch with that many ranks" % a.gpus)
        a.gpus = world
    ndev = torch.cuda.device_count()
    if world > ndev and a.backend == "nccl" and not os.environ.get("DNMF_BENCH_OVERSUBSCRIBE"):   # (the override exercises the fallback below)
        sys.exit("bench.py: %d ranks but %d GPUs" % (world, ndev))
    local = local % max(1, ndev)            # gloo debugging runs may stack ranks on one device
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(a.backend)      # nccl = RCCL; communicators are created lazily on the current device
        if a.backend == "nccl":
            # one tiny collective up front: if RCCL cannot come up on this node the run continues over gloo (host staged,
            # slower, and labelled as such in config.parallelism) instead of producing no number at all
            try:
                t = torch.ones(1, device=dev)
                dist.all_reduce(t)
                torch.cuda.synchronize()
                if int(t.item()) != world:
                    raise RuntimeError("allreduce returned %s" % t.item())
            except Exception as exc:  # noqa: BLE001
                sys.stderr.write("bench.py: RCCL unavailable (%s); falling back to gloo\n" % exc)
                try:
                    dist.destroy_process_group()
                except Exception:  # noqa: BLE001
                    pass
                a.backend = "gloo"
                dist.init_process_group("gloo")

    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.dist_nmf import nmf_algorithms_1D
    from pydnmfk_amd.engine import HIP_OPS as ops, new_gram
    from pydnmfk_amd.utils import determine_block_params, parse

    m, n, k = a.m, a.n, a.k
    comms = MPI_comm(None, world, 1)
    p = parse()
    p.comm1, p.comm, p.p_r, p.p_c, p.k, p.m, p.n = comms.comm, comms, world, 1, k, m, n
    p.row_comm, p.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
    p.norm, p.method, p.W_update, p.eps = a.norm, "mu", True, 1.1920929e-07
    m_l = determine_block_params(rank, (world, 1), (m, n)).determine_block_shape_asymm()[0]

    # synthetic data, generated on device: X ~ U[0,1), W0 per rank, H0 from rank 0
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    A = torch.rand(m_l, n, device=dev, generator=g)
    g.manual_seed(4321 + rank)
    W = torch.rand(m_l, k, device=dev, generator=g)
    g.manual_seed(99)
    H = torch.rand(k, n, device=dev, generator=g)
    if world > 1:
        H = comms.comm.bcast(H, root=0)

    def barrier():
        if a.backend == "nccl":
            dist.barrier(device_ids=[local])
        else:
            dist.barrier()

    def step(i):
        nmf_algorithms_1D(A, W, H, params=p).update(clamp=(i % 10 == 0))

    for i in range(a.warmup):
        step(i)
    if world > 1:
        barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(W).all() and torch.isfinite(H).all()

    out = None
    if rank == 0:
        ms = elapsed / a.steps * 1e3
        flops_iter = 4.0 * m * n * k + 4.0 * (m + n) * k * k      # SURVEY 8d, whole job
        out = {
            "metric": "mu_iterations_per_sec", "value": a.steps / elapsed, "unit": "iter/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "MU/%s X=%dx%d fp32 k=%d, 1D row grid p_r=%d p_c=1 (BASELINE config 3)" % (
                a.norm.upper(), m, n, k, world), "m": m, "n": n, "k": k, "rows_per_gpu": m_l,
                "parallelism": ("row-sharded X, allreduce[W^T A | W^T W] over %s" % ("RCCL" if a.backend == "nccl" else a.backend + " (host staged)")) if world > 1 else "single GPU"},
            "step_tflops_per_gpu": flops_iter / world / (ms * 1e-3) / 1e12,
            "step_mfma_frac": flops_iter / world / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            "step_algorithmic_hbm_gbs_per_gpu": (4.0 * m_l * n + 12.0 * (m_l + n) * k) / (ms * 1e-3) / 1e9,
        }

    if not a.no_kernel_timing and a.norm == "fro":
        # Per-kernel HIP-event timings IN SITU: the step is replayed primitive by primitive (same launches, same order
        # as dnmf_mu_fro_step / the 1D-row choreography) with an event pair around every library call, so each kernel
        # runs in the cache / clock state it sees inside the real step.  Events are recorded on torch's current stream,
        # which is the stream every launch goes to.
        G = new_gram(k, dev)
        AtW = torch.empty(k, n, device=dev)
        reps = 10
        calls = [("gram_hht", lambda: ops.gram_hht(H, G)),
                 ("aht_update_w", lambda: ops.aht_update_w(A, H, G, W, p.eps)),
                 ("gram_wtw", lambda: ops.gram_wtw(W, G)),
                 ("wta", lambda: ops.wta(A, W, AtW)),
                 ("mu_update_h", lambda: ops.mu_update_h(H, AtW, G, p.eps, False))]
        evs = {name: [] for name, _ in calls}
        for it in range(reps + 2):
            for name, fn in calls:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                fn()
                e.record()
                if it >= 2:
                    evs[name].append((s, e))
            if world > 1:
                pass  # kernels are timed on this rank's slab; the exchange is part of ms_per_step only
        torch.cuda.synchronize()
        t = {name: sum(s.elapsed_time(e) for s, e in v) / len(v) for name, v in evs.items()}
        t3, _ = event_time_ms(lambda: ops.sqnorm(A), reps=6, warm=2)
        fl_nt = 2.0 * m_l * n * k + 2.0 * m_l * k * k
        fl_tn = 2.0 * m_l * n * k
        kern = {
            "dnmf_aht_update_w = nt_kernel<FUSED_W> (A.H^T + W update, 1 launch)": {
                "ms": t["aht_update_w"], "tflops": fl_nt / t["aht_update_w"] / 1e9,
                "frac_mfma": fl_nt / t["aht_update_w"] / 1e9 / PEAK_FP32_MFMA_TFLOPS,
                "algorithmic_gbs": (4.0 * m_l * n + 8.0 * m_l * k) / t["aht_update_w"] / 1e6},
            "dnmf_wta = tn_kernel<PARTIAL> + reduce_partials (W^T.A)": {
                "ms": t["wta"], "tflops": fl_tn / t["wta"] / 1e9,
                "frac_mfma": fl_tn / t["wta"] / 1e9 / PEAK_FP32_MFMA_TFLOPS,
                "algorithmic_gbs": (4.0 * m_l * n + 4.0 * m_l * k) / t["wta"] / 1e6},
            "dnmf_mu_update_h = tn_kernel<UPDATE_H> (eltwise multiply-divide + k x k product)": {
                "ms": t["mu_update_h"], "algorithmic_gbs": 12.0 * n * k / t["mu_update_h"] / 1e6},
            "dnmf_gram_wtw (tn_kernel + reduce)": {"ms": t["gram_wtw"]},
            "dnmf_gram_hht (nt_kernel split + reduce)": {"ms": t["gram_hht"]},
            "dnmf_sqnorm = sqnorm_kernel (||A||^2; eltwise/norm class, HBM bound)": {
                "ms": t3, "algorithmic_gbs": 4.0 * m_l * n / t3 / 1e6, "frac_hbm": 4.0 * m_l * n / t3 / 1e6 / PEAK_HBM_GBS},
        }
        if rank == 0:
            ach = fl_nt / t["aht_update_w"] / 1e9
            kname = "nt16_kernel<FUSED_W>" if (k <= 16 and n % 32 == 0) else "nt_kernel<KT=%d,FUSED_W>" % max(1, (k + 31) // 32)
            out["roofline"] = {"kernel": kname + " (dnmf_aht_update_w)",
                               "bound": "mfma", "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                               "flops_per_launch": fl_nt, "ms_per_launch": t["aht_update_w"]}
            if k in MEASURED_MFMA_WITH_STREAM_TFLOPS:       # informational: fraction of the MEASURED mixed ceiling
                out["roofline"]["measured_ceiling"] = MEASURED_MFMA_WITH_STREAM_TFLOPS[k]
                out["roofline"]["frac_of_measured_ceiling"] = ach / MEASURED_MFMA_WITH_STREAM_TFLOPS[k]
            tr = pmc_traffic_bytes() if (m, n, k, world) == (262144, 8192, 64, 1) else None
            if tr is not None:
                out["roofline"]["traffic"] = tr["bytes"]
                out["roofline"]["traffic_note"] = "HBM bytes per launch, PMC pass %s (FETCH_SIZE x2 + WRITE_SIZE)" % tr["source"]
            out["roofline_hbm"] = {"kernel": "sqnorm_kernel", "bound": "hbm", "achieved": 4.0 * m_l * n / t3 / 1e6,
                                   "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": 4.0 * m_l * n / t3 / 1e6 / PEAK_HBM_GBS,
                                   "traffic": None, "measured_ceiling": MEASURED_STREAM_GBS,
                                   "frac_of_measured_ceiling": 4.0 * m_l * n / t3 / 1e6 / MEASURED_STREAM_GBS}
            out["kernels"] = kern

    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, k, m)
        print(json.dumps(out), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
