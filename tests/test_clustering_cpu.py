"""Clustering / silhouettes against the reference's own golden file (tests/sill.npy of lanl/pyDNMFk, kept as
tests/golden/ref_sill.npy): same data recipe as the reference's tests/test_dist_clustering.py:15-59, rtol = atol = 1e-3."""
import numpy as np
import pytest
import torch

from tests._mp import free_port


def _problem():
    np.random.seed(100)

    def gauss(n, mean, std):
        return np.exp(-(np.linspace(1, n, n) - mean) ** 2 / std)

    m, p, k = 16, 4, 3
    W = np.vstack([gauss(m, 3, 3), gauss(m, 8, 2), gauss(m, 14, 3)]).T
    W_all = np.stack([W[:, np.random.permutation(k)] + np.random.rand(m, k) * .1 for _ in range(p)], axis=-1)
    H_dist = np.random.rand(k, 5, p)
    return W_all, H_dist


def _cluster(W_all, H_all, p_r, comm1):
    from pydnmfk_amd.dist_clustering import custom_clustering
    from pydnmfk_amd.utils import parse
    args = parse()
    args.comm1, args.p_r, args.p_c, args.eps = comm1, p_r, 1, np.finfo(np.float64).eps
    c = custom_clustering(W_all, H_all, args)
    res = c.fit()
    return c, res, c.dist_silhouettes()


def test_serial_matches_reference_golden(golden_dir):
    from pydnmfk_amd.dist_comm import COMM_WORLD
    W_all, H_all = _problem()
    c, res, sils = _cluster(W_all, H_all, 1, COMM_WORLD())
    ref = np.load(golden_dir + "/ref_sill.npy")
    assert sils.shape == ref.shape == (3, 4)
    assert np.allclose(ref, sils, rtol=1e-3, atol=1e-3)
    centroids, cent_std, H_ord, sil_k, sil_avg, orders = res
    assert centroids.shape == (16, 3) and cent_std.shape == (16, 3) and H_ord.shape == (3, 5, 4)
    assert np.allclose(np.linalg.norm(centroids.numpy(), axis=0), 1.0, atol=1e-6)
    assert abs(sil_avg - sils.mean()) < 0.05
    # every group's columns are now ordered like group 0's: correlation with the centroids is diagonal-dominant
    for p in range(4):
        sim = centroids.numpy().T @ c.W_all[:, :, p].numpy()
        assert (np.argmax(sim, axis=1) == np.arange(3)).all()


def _rank_body(rank, world, port, q):
    import os
    import torch.distributed as dist
    from pydnmfk_amd.dist_comm import MPI_comm
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        W_all, H_all = _problem()
        comms = MPI_comm(None, world, 1)
        rows = W_all.shape[0] // world
        _, _, sils = _cluster(W_all[rank * rows:(rank + 1) * rows], H_all, world, comms.comm)
        q.put((rank, sils, None))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, None, traceback.format_exc()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_match_reference_golden(golden_dir):
    """The reference test itself runs on a (2,1) grid (tests/test_dist_clustering.py:26-50)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_rank_body, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    ref = np.load(golden_dir + "/ref_sill.npy")
    for rank, sils, err in res:
        assert err is None, err
        assert np.allclose(ref, sils, rtol=1e-3, atol=1e-3)


def test_early_exit_is_bitwise_the_full_100_rounds():
    """The fixed-point exit of dist_custom_clustering must not change anything: compare with all 100 rounds (the
    reference's loop, dist_clustering.py:114) on shuffled, noisy groups -- centroids, reordered W / H, the MAD, the
    silhouettes and the complete list of orders."""
    import numpy as np
    import torch
    from pydnmfk_amd.dist_clustering import custom_clustering
    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.utils import parse
    comms = MPI_comm(None, 1, 1)
    args = parse()
    args.comm1, args.p_r, args.p_c, args.eps = comms.comm, 1, 1, 1.1920929e-07
    rs = np.random.RandomState(3)
    for (m, n, k, P, noise) in ((60, 40, 5, 7, 0.05), (33, 21, 3, 4, 0.4), (50, 30, 6, 6, 0.0)):
        W0, H0 = rs.rand(m, k).astype(np.float32), rs.rand(k, n).astype(np.float32)
        Wall, Hall = np.empty((m, k, P), np.float32), np.empty((k, n, P), np.float32)
        for p in range(P):
            perm = rs.permutation(k)
            Wall[:, :, p] = W0[:, perm] + noise * rs.rand(m, k)
            Hall[:, :, p] = H0[perm, :] + noise * rs.rand(k, n)
        fast = custom_clustering(Wall, Hall, args)
        full = custom_clustering(Wall, Hall, args)
        full.early_exit = False
        a, b = fast.fit(), full.fit()
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
        assert np.array_equal(a[3], b[3]) and a[4] == b[4]
        assert a[5] == b[5] and len(a[5]) == 100 * P
