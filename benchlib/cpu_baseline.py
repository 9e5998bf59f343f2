"""benchlib.cpu_baseline -- the `cpu_baseline` legs of bench.py: the numpy checker (oracle/nmf_oracle.py, a port of the reference's path) timed
on the host in the reference's process model.  Test infrastructure used as the thing measured ONLY here (the contract's cpu_baseline)."""
import json
import os
import sys
import time

from .common import ROOT


def _cpu_rank(rank, P, m_l, n, k, steps, q):
    """One rank of the reference's process model: single BLAS thread (main.py:3 pins OMP_NUM_THREADS=1), its own row
    slab of X, the oracle's step (the numpy calls of dist_nmf.py:716-751)."""
    os.environ["OMP_NUM_THREADS"] = "1"
    import numpy as np
    from oracle import nmf_oracle as orc
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=1)
    except ImportError:
        ctx = None
    rng = np.random.default_rng(1234 + rank)
    A = rng.random((m_l, n), dtype=np.float32)
    W = rng.random((m_l, k), dtype=np.float32)
    H = np.random.default_rng(99).random((k, n), dtype=np.float32)
    eps = np.finfo(np.float32).eps
    orc.fro_mu_step_local(A, W, H, eps)                      # warm-up (page faults, BLAS init)
    q.put(("ready", rank))
    t0 = time.perf_counter()
    for _ in range(steps):
        orc.fro_mu_step_local(A, W, H, eps)
    q.put(("done", rank, (time.perf_counter() - t0) / steps))
    del ctx


def cpu_baseline(n, k, m_full, steps=3):
    """The reference's CPU path beside the GPU number: P = min(8, host cores) processes x 1 thread, each
    holding the 1/P row slab a rank of the p_r = P grid would hold, each running the oracle's MU/FRO step; the iteration
    time of the job is the slowest rank's (the 2 MiB allreduce the reference adds is not simulated: it only makes the
    CPU figure slightly optimistic).  Bounded: 1 warm-up + `steps` timed steps per process."""
    import multiprocessing as mp
    cores = os.cpu_count() or 1
    P = min(8, cores)
    m_l = m_full // P
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cpu_rank, args=(r, P, m_l, n, k, steps, q)) for r in range(P)]
    for pr in procs:
        pr.start()
    import queue
    times, deadline = {}, time.time() + 600
    try:
        while len(times) < P and time.time() < deadline:
            try:
                msg = q.get(timeout=2)
            except queue.Empty:
                if any(pr.exitcode not in (None, 0) for pr in procs):
                    break                                    # a rank died: report that instead of waiting
                continue
            if msg[0] == "done":
                times[msg[1]] = msg[2]
    finally:
        for pr in procs:
            pr.join(timeout=5)
            if pr.is_alive():
                pr.kill()
    if len(times) < P:
        return {"value": None, "unit": "iter/s", "cores": P, "kind": "port", "host_cores": cores, "host_cpu": model,
                "sample": "FAILED: %d of %d CPU ranks finished (exit codes %s)" % (len(times), P, [pr.exitcode for pr in procs])}
    t = max(times.values())
    flops = 4.0 * m_full * n * k + 4.0 * (m_full + n * P) * k * k          # every rank forms its own k x k products
    return {"value": 1.0 / t, "unit": "iter/s", "cores": P, "kind": "port",
            "sample": "oracle fro_mu_step_local in the reference's process model: %d processes x 1 BLAS thread, each on "
                      "its %dx%d row slab (1/%d of X), k=%d, 1 warm-up + %d timed steps; iteration time = slowest rank "
                      "(%.2f s; fastest %.2f s), no allreduce simulated" % (P, m_l, n, P, k, steps, t, min(times.values())),
            "host_cores": cores, "host_cpu": model, "seconds_per_iter": t, "gflops_whole_job": flops / t / 1e9}


def _cpu_allcores_rank(m_s, n, k, steps, q):
    """The non-reference threading variant: ONE process, the BLAS library free to use every host core."""
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.pop(v, None)
    import numpy as np
    from oracle import nmf_oracle as orc
    try:
        from threadpoolctl import threadpool_info
        nthreads = max([int(d.get("num_threads", 1)) for d in threadpool_info()] or [1])
    except ImportError:
        nthreads = os.cpu_count() or 1
    rng = np.random.default_rng(1234)
    A = rng.random((m_s, n), dtype=np.float32)
    W = rng.random((m_s, k), dtype=np.float32)
    H = np.random.default_rng(99).random((k, n), dtype=np.float32)
    eps = np.finfo(np.float32).eps
    orc.fro_mu_step_local(A, W, H, eps)
    t0 = time.perf_counter()
    for _ in range(steps):
        orc.fro_mu_step_local(A, W, H, eps)
    q.put(((time.perf_counter() - t0) / steps, nthreads))


def cpu_baseline_allcores(n, k, m_full, steps=3):
    """One process with all host cores' BLAS threads on a 1/8 row slab of X (the oracle's step; the whole-X iteration time
    is 8 x the slab's: the step is linear in the rows).  Labelled non-reference: the reference pins one BLAS thread per
    rank (main.py:3)."""
    import multiprocessing as mp
    import queue
    frac = 8
    m_s = m_full // frac
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    saved = {v: os.environ.pop(v) for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS") if v in os.environ}
    try:
        pr = ctx.Process(target=_cpu_allcores_rank, args=(m_s, n, k, steps, q))
        pr.start()
    finally:
        os.environ.update(saved)
    try:
        t, nthreads = q.get(timeout=300)
    except queue.Empty:
        pr.kill()
        return {"value": None, "unit": "iter/s", "kind": "port", "sample": "FAILED: no result within 300 s"}
    pr.join(timeout=5)
    t_full = t * frac
    flops = 4.0 * m_full * n * k + 4.0 * (m_full + n) * k * k
    return {"value": 1.0 / t_full, "unit": "iter/s", "cores": nthreads, "kind": "port", "threading": "non-reference",
            "sample": "oracle fro_mu_step_local, ONE process with %d BLAS threads on a %dx%d row slab (1/%d of X), k=%d, 1 warm-up "
                      "+ %d timed steps: %.3f s per slab step, x %d = %.2f s per iteration of the whole X (the step is linear in "
                      "the rows); non-reference threading (the reference pins one BLAS thread per rank)" % (
                          nthreads, m_s, n, frac, k, steps, t, frac, t_full),
            "seconds_per_iter": t_full, "gflops_whole_job": flops / t_full / 1e9}


def _cpu_kl_rank(rank, rows_s, n_l, k, steps, q):
    """One rank of the reference's process model for config 4 (single BLAS thread, main.py:3): the oracle's MU/KL step
    (dist_nmf.py:806-849) on a ROW SLAB of the rank's block -- the step is linear in the rows."""
    os.environ["OMP_NUM_THREADS"] = "1"
    import numpy as np
    from oracle import nmf_oracle as orc
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=1)
    except ImportError:
        ctx = None
    rng = np.random.default_rng(1234 + rank)
    A = rng.random((rows_s, n_l), dtype=np.float32)
    W = rng.random((rows_s, k), dtype=np.float32)
    H = np.random.default_rng(99).random((k, n_l), dtype=np.float32)
    eps = np.finfo(np.float32).eps
    orc.kl_mu_step_local(A, W, H, eps)
    t0 = time.perf_counter()
    for _ in range(steps):
        orc.kl_mu_step_local(A, W, H, eps)
    q.put((rank, (time.perf_counter() - t0) / steps))
    del ctx


def _cpu_hals_rank(rank, m_l, n_l, ks, steps, q):
    """One rank of the reference's process model for config 5: the oracle's HALS/FRO step (dist_nmf.py:873-934) on the rank's
    block (float32: numpy has no bfloat16) for a few ranks k."""
    os.environ["OMP_NUM_THREADS"] = "1"
    import numpy as np
    from oracle import nmf_oracle as orc
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=1)
    except ImportError:
        ctx = None
    rng = np.random.default_rng(1234 + rank)
    A = rng.random((m_l, n_l), dtype=np.float32)
    eps = np.finfo(np.float32).eps
    res = {}
    for k in ks:
        W = rng.random((m_l, k), dtype=np.float32)
        H = rng.random((k, n_l), dtype=np.float32)
        orc.fro_hals_step_local(A, W, H, eps)
        t0 = time.perf_counter()
        for _ in range(steps):
            orc.fro_hals_step_local(A, W, H, eps)
        res[k] = (time.perf_counter() - t0) / steps
    q.put((rank, res))
    del ctx


def _cpu_pool(target, argsets, timeout=600):
    """P single-thread processes, one result each (rank, value); None when any of them fails."""
    import multiprocessing as mp
    import queue
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=tuple(args) + (q,)) for args in argsets]
    for pr in procs:
        pr.start()
    got, deadline = {}, time.time() + timeout
    try:
        while len(got) < len(procs) and time.time() < deadline:
            try:
                r, v = q.get(timeout=2)
                got[r] = v
            except queue.Empty:
                if any(pr.exitcode not in (None, 0) for pr in procs):
                    break
    finally:
        for pr in procs:
            pr.join(timeout=5)
            if pr.is_alive():
                pr.kill()
    return got if len(got) == len(procs) else None


def host_cpu():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return os.cpu_count() or 1, model
