"""The update steps contain no atomics and every split-K / partial-slab reduction adds in a fixed order, so the same
inputs give bit-identical factors run after run (the norms of pyDNMF.py:205-218 accumulate with fp64 atomics and are
exempt).  Checked for every kernel family: 32-wide NT / TN, the one-round three-tile NT loop, the 16-wide kernels,
bf16-stored X, the KL kernels (pipelined and edge paths)."""
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m,n,k,norm,bf16", [(4096, 1024, 64, "fro", False), (33000, 512, 64, "fro", False),
                                             (8192, 2048, 16, "fro", False), (8192, 2048, 5, "fro", True),
                                             (3000, 1500, 32, "fro", True), (4096, 2048, 128, "kl", False),
                                             (1000, 700, 40, "kl", False)])
def test_steps_are_bitwise_reproducible(m, n, k, norm, bf16):
    from pydnmfk_amd.engine import HIP_OPS as ops
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(3)
    A = torch.rand(m, n, device=dev, generator=g)
    if bf16:
        A = A.to(torch.bfloat16)
    W0 = torch.rand(m, k, device=dev, generator=g)
    H0 = torch.rand(k, n, device=dev, generator=g)
    step = ops.mu_fro_step if norm == "fro" else ops.mu_kl_step
    outs = []
    for _ in range(2):
        W, H = W0.clone(), H0.clone()
        for i in range(12):
            step(A, W, H, 1.1920929e-07, True, i % 10 == 0)
        outs.append((W, H))
    assert torch.isfinite(outs[0][0]).all() and torch.isfinite(outs[0][1]).all()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("m,n,k", [(40000, 1024, 16), (8192, 512, 64), (3001, 700, 7)])
def test_hals_steps_are_bitwise_reproducible(m, n, k):
    """HALS joins the list: the persistent W sweep reduces the column norms over per-workgroup slots in a fixed order."""
    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.dist_nmf import nmf_algorithms_1D
    from pydnmfk_amd.utils import parse
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(5)
    A = torch.rand(m, n, device=dev, generator=g)
    W0 = torch.rand(m, k, device=dev, generator=g)
    H0 = torch.rand(k, n, device=dev, generator=g)
    comms = MPI_comm(None, 1, 1)
    p = parse()
    p.comm1, p.comm, p.p_r, p.p_c, p.k, p.m, p.n = comms.comm, comms, 1, 1, k, m, n
    p.row_comm, p.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
    p.norm, p.method, p.W_update, p.eps = "fro", "hals", True, 1.1920929e-07
    outs = []
    for _ in range(2):
        W, H = W0.clone(), H0.clone()
        for i in range(8):
            nmf_algorithms_1D(A, W, H, params=p).update(clamp=(i % 10 == 0))
        outs.append((W, H))
    assert torch.isfinite(outs[0][0]).all() and torch.isfinite(outs[0][1]).all()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
