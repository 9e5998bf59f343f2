"""The C step entry points (dnmf_mu_{fro,kl}_step_{1d,2d}, csrc/dnmf_comm.hip) on MORE THAN ONE RANK: a hosted communicator
(dnmf_comm_create_hosted) hands every collective of a step to the host, which runs it over gloo -- world_size 2..8 processes
stacked on the one GPU, the real kernels, the library's own sequencing of kernels and exchanges, its group / member / block
order.  Each rank must end with the factors the Python choreography (pinned by the reference's goldens on the same grids,
tests/test_gpu_multirank.py) computes from the same inputs over the same transport: bit for bit.
What RCCL adds on a real node is the wire, not the order: ncclCommSplit(colour, key) builds the same groups
(dist_comm.py:25-51)."""
import os
import traceback

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from tests.conftest import long_param  # noqa: E402
pytestmark = pytest.mark.gpu


def _rank(rank, world, port, cfg, q):
    try:
        import torch.distributed as dist
        from pydnmfk_amd.dist_comm import MPI_comm
        from pydnmfk_amd.dist_nmf import nmf_algorithms_1D, nmf_algorithms_2D
        from pydnmfk_amd.utils import parse
        torch.set_num_threads(1)
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        p_r, p_c, m, n, k, norm, w_update, chunks = cfg
        comms = MPI_comm(None, p_r, p_c)
        i, j = rank // p_c, rank % p_c
        rs = np.random.RandomState(7 * m + n + k)
        A = (rs.rand(m, n) + 0.05).astype(np.float32)
        A[rs.rand(m, n) < 0.1] = 0.0
        W0, H0 = (rs.rand(m, k) + 0.05).astype(np.float32), (rs.rand(k, n) + 0.05).astype(np.float32)
        def part(total, p, q):        # the partition rule (utils.py:36-41): (start, count) of member q
            return q * (total // p) + min(q, total % p), total // p + (1 if q < total % p else 0)
        (r0, m_l), (c0, n_l) = part(m, p_r, i), part(n, p_c, j)
        A_ij = torch.from_numpy(np.ascontiguousarray(A[r0:r0 + m_l, c0:c0 + n_l])).to(dev)
        two_d = p_r > 1 and p_c > 1
        if two_d:                      # the rank's slices of its grid row's W_i and its grid column's H_j (utils.py:99-115)
            (ws_, m_w), (hs_, n_h) = part(m_l, p_c, j), part(n_l, p_r, i)
            w0, h0 = r0 + ws_, c0 + hs_
        else:                          # 1D: the factor along the split axis is the rank's block, the other is replicated
            m_w, n_h, w0, h0 = m_l, n_l, r0, c0
        Wb, Hb = W0[w0:w0 + m_w], H0[:, h0:h0 + n_h]
        eps = float(np.finfo(np.float32).eps)

        def args(exchange):
            a = parse()
            a.comm1, a.comm, a.p_r, a.p_c, a.k, a.m, a.n = comms.comm, comms, p_r, p_c, k, m, n
            a.row_comm, a.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
            a.eps, a.W_update, a.norm, a.method = eps, w_update, ("fro" if norm == "hals" else norm), ("hals" if norm == "hals" else "mu")
            a.overlap_min_cols, a.overlap_chunks = 64, chunks
            if exchange:
                a.exchange = exchange
            return a
        cls = nmf_algorithms_2D if two_d else nmf_algorithms_1D
        ap, an = args(None), args("native-hosted")
        Wp, Hp = (torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (Wb, Hb))
        Wn, Hn = Wp.clone(), Hp.clone()
        took = 0
        # (HALS on unstructured data is chaotic -- columns that collapse to eps are renormalised from rounding noise: a 1e-7
        #  difference in ring order is 3e-2 after the third sweep of the ragged 2 x 3 case -- so its cases stop after two)
        steps = 2 if norm == "hals" else 3
        for it in range(steps):
            cls(A_ij, Wp, Hp, params=ap).update(clamp=(it == 1))
            alg = cls(A_ij, Wn, Hn, params=an)
            took += int(bool(alg._native_step(clamp=(it == 1))))
        eq = bool(torch.equal(Wp, Wn) and torch.equal(Hp, Hn))
        dw = float((Wp - Wn).norm() / Wp.norm())
        dh = float((Hp - Hn).norm() / Hp.norm())
        moved = float((Wp - torch.from_numpy(np.ascontiguousarray(Wb)).to(dev)).norm() / Wp.norm()) > 1e-3 or not w_update
        q.put((rank, (took == steps, eq, dw, dh, moved), None))
        dist.barrier()
        an._native_comm.close()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        q.put((rank, None, traceback.format_exc()))


# Ragged grids (a dimension that does not divide): the library exchanges blocks at the pitch of the largest one where the gloo
# fall-back of the choreography allreduces the whole buffer -- the same sums in a different ring order once a group has more than
# two members, so those cases are held to 2e-6 instead of bit equality.
CASES = [   # p_r, p_c, m, n, k, norm, W_update, overlap chunks of the 1D row grid
    (2, 1, 512, 256, 16, "fro", True, 1), (4, 1, 1024, 512, 64, "fro", True, 2), (3, 1, 300, 260, 5, "kl", True, 1),
    (1, 2, 256, 512, 32, "fro", True, 1), (1, 3, 200, 384, 8, "kl", True, 1),
    (2, 2, 512, 256, 16, "kl", True, 1), (2, 3, 240, 192, 33, "fro", True, 1),
    (4, 2, 200, 256, 64, "kl", True, 1),
    # ragged / narrow slices: rows and columns that do not divide, equal slices that are not whole tiles, 16-wide and 32-wide kernels
    (2, 2, 515, 262, 16, "fro", True, 1), (2, 2, 515, 262, 16, "kl", True, 1), (2, 2, 512, 264, 24, "kl", True, 1),
    (2, 3, 241, 199, 33, "fro", True, 1), (3, 2, 301, 197, 5, "kl", True, 1), (4, 2, 203, 259, 64, "fro", True, 1), (2, 4, 150, 140, 7, "kl", False, 1),
    (3, 1, 301, 260, 16, "fro", True, 1),
    # HALS / Frobenius (dnmf_hals_fro_step_{1d,2d}): the column norms of the W sweep are 8-byte allreduces between the column kernels
    (2, 1, 512, 256, 16, "hals", True, 1), (3, 1, 301, 260, 5, "hals", True, 1), (1, 2, 256, 512, 32, "hals", True, 1),
    (2, 2, 512, 256, 16, "hals", True, 1), (2, 3, 241, 199, 33, "hals", True, 1), (4, 2, 200, 256, 64, "hals", False, 1),
]


# the default tier keeps one case of every family (row / column / 2D grids, overlapped chunks, ragged slices, W_update off, HALS
# 1D and 2D); the rest runs with DNMF_LONG_TESTS=1
DEFAULT = {(2, 1, 512, 256, 16, "fro"), (4, 1, 1024, 512, 64, "fro"), (1, 2, 256, 512, 32, "fro"), (2, 2, 512, 256, 16, "kl"),
           (4, 2, 200, 256, 64, "kl"), (2, 3, 241, 199, 33, "fro"), (3, 2, 301, 197, 5, "kl"), (2, 4, 150, 140, 7, "kl"),
           (2, 1, 512, 256, 16, "hals"), (2, 3, 241, 199, 33, "hals"), (4, 2, 200, 256, 64, "hals")}


def _id(c):
    return "%dx%d_%dx%d_k%d_%s%s" % (c[0], c[1], c[2], c[3], c[4], c[5], "" if c[6] else "_noW")


@pytest.mark.parametrize("cfg", [pytest.param(c, id=_id(c)) if tuple(c[:6]) in DEFAULT else long_param(c, id=_id(c)) for c in CASES])
def test_c_steps_over_a_hosted_transport_equal_the_choreography(cfg):
    import torch.multiprocessing as mp
    from tests._mp import free_port
    world = cfg[0] * cfg[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_rank, args=(r, world, port, cfg, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, out, err in res:
        assert err is None, "rank %d failed:\n%s" % (rank, err)
        took, eq, dw, dh, moved = out
        assert took, (rank, "the library's step entry point was not taken", out)
        p_r, p_c, m, n = cfg[:4]
        two_d = p_r > 1 and p_c > 1
        ragged = two_d and (m % (p_r * p_c) or n % (p_r * p_c)) and max(p_r, p_c) > 2
        if ragged:
            assert dw <= 2e-6 and dh <= 2e-6, (rank, dw, dh)
        else:
            assert eq, (rank, dw, dh)
        assert moved
