"""Multi-rank product path on the GPU: the real HIP kernels driven by the p_r x p_c choreography with 2-4
processes, all on cuda:0 (a GPU box has one device), exchanging through gloo (host-staged) instead of RCCL.
Checks W, H per rank and recon_err against the golden vectors captured from the reference on the same grids.
What this does not cover is the RCCL transport itself (the 8-GPU runs are the driver's)."""
import pytest

torch = pytest.importorskip("torch")
from tests.conftest import long_param  # noqa: E402
pytestmark = pytest.mark.gpu

CASES = ["t24x12_2x1_fro_float32", "t24x12_1x2_kl_float32", "t24x12_2x2_fro_float32", "r25x13_3x1_fro_float32",
         "r25x13_2x2_kl_float32", "swim_4x1_fro_float32", "swim_1x4_fro_float32", "swim_2x2_kl_float32",
         "lr200x136k64_2x2_fro_float32", "lr200x136k64_2x1_kl_float32", "lr150x140k128_1x2_fro_float32",
         "lr136x100k32_2x2_kl_float32",
         "t24x12_2x1_hals_float32", "t24x12_2x2_hals_float32", "r25x13_3x1_hals_float32", "swim_4x1_hals_float32",
         "lr200x136k64_1x2_hals_float32",
         "t24x12z_2x1_fro_float32_prune", "t24x12z_1x2_fro_float32_prune", "t24x12z_2x2_fro_float32_prune",
         # non-square 2D grids (config 4 is 4x2): 6 or 8 ranks on the one GPU
         "r50x39_4x2_fro_float32", "r50x39_4x2_kl_float32", "r50x39_4x2_hals_float32",
         "r50x39_2x3_fro_float32", "r50x39_2x3_kl_float32", "r50x39_2x3_hals_float32",
         "r50x39_3x2_fro_float32", "r50x39_2x4_kl_float32",
         "lr200x136k64_4x2_fro_float32", "lr200x136k64_4x2_kl_float32", "lr200x136k64_2x3_hals_float32",
         "lr150x140k128_4x2_kl_float32", "swim_4x2_kl_float32",
         # 2D Frobenius / HALS with 32-column-aligned slices: the W phase reads the allgathered H as column blocks (aht_hblocks)
         "swim_2x2_fro_float32", "swim_2x2_hals_float32"]


@pytest.mark.parametrize("name", CASES)
def test_multirank_hip_matches_reference(name):
    from tests._mp import run_case
    run_case(name, use_hip=True, timeout=400)


NATIVE_CASES = ["t24x12_2x1_fro_float32", "t24x12_1x2_kl_float32", "swim_4x1_fro_float32", "lr200x136k64_2x1_kl_float32", "lr150x140k128_1x2_fro_float32",
                "t24x12_2x2_fro_float32", "r25x13_2x2_kl_float32", "swim_2x2_kl_float32", "swim_2x2_fro_float32",
                "lr200x136k64_2x2_fro_float32", "r50x39_4x2_fro_float32", "r50x39_4x2_kl_float32",
                "lr150x140k128_4x2_kl_float32",
                "t24x12_2x1_hals_float32", "r25x13_3x1_hals_float32", "lr200x136k64_1x2_hals_float32",
                "r50x39_4x2_hals_float32", "lr200x136k64_2x3_hals_float32", "swim_2x2_hals_float32"]


@pytest.mark.parametrize("name", NATIVE_CASES)
def test_multirank_library_sequenced_steps_match_reference(name):
    """The same golden fits with every MU / HALS step sequenced INSIDE the library (dnmf_mu_{fro,kl}_step_{1d,2d}, dnmf_hals_fro_step_{1d,2d}: kernels, the
    exchanges, kernels in one call; params.exchange = 'native-hosted' hands the collectives to gloo through
    dnmf_comm_create_hosted) -- 1D and 2D grids, even and ragged blocks, against the reference's W, H and recon_err."""
    from tests._mp import run_case
    run_case(name, use_hip=True, timeout=400, extra={"exchange": "native-hosted"})


@pytest.mark.parametrize("name", ["swim_4x1_fro_float32", "lr200x136k64_2x1_kl_float32", "lr150x140k128_1x2_fro_float32",
                                  "t24x12_2x1_hals_float32", "r25x13_3x1_hals_float32", "swim_4x1_hals_float32",
                                  # (HALS: the W sweep is ONE persistent launch across the ranks, csrc/dnmf_hals.h HalsPeers.  The 4 x 2 case stacks EIGHT
                                  # processes with spin-waiting kernels on the one GPU of the test box -- 39 s of time slicing: the long tier)
                                  long_param("r50x39_4x2_hals_float32")])
def test_multirank_direct_allreduce_matches_reference(name):
    """Reference golden fits on 1D grids with every step inside the library AND its packed exchange through the direct two-shot
    allreduce over IPC peer buffers (params.direct_allreduce; csrc/dnmf_comm.hip): four / two ranks stacked on the one GPU map
    each other's regions; W, H and recon_err against the reference at the usual tolerances."""
    from tests._mp import run_case
    run_case(name, use_hip=True, timeout=400, extra={"exchange": "native-hosted", "direct_allreduce": True})


@pytest.mark.parametrize("grid,method", [((2, 1), "hals"), ((1, 2), "mu"), ((2, 2), "hals")])
def test_multirank_hip_bf16_storage(grid, method):
    """bf16-stored data blocks on a grid, real HIP kernels (the *_bf16a entry points), gloo transport."""
    from tests._mp import run_bf16
    run_bf16(grid, method, use_hip=True)


@pytest.mark.parametrize("grid,method", [((2, 1), "hals"), long_param((2, 1), "mu"), long_param((1, 2), "mu"), long_param((2, 2), "hals"),
                                         ((2, 2), "mu"), long_param((3, 2), "hals")])
def test_multirank_bf16_storage_library_sequenced(grid, method):
    """bf16-stored blocks with every step sequenced inside the library (dnmf_{mu,hals}_fro_step_{1d,2d}_bf16a over the hosted
    gloo transport): BASELINE config 5's arithmetic (HALS / Frobenius on bf16 data) on row, column and 2D grids, even and ragged,
    against the oracle's grid simulation on float(bf16(A))."""
    from tests._mp import run_bf16
    run_bf16(grid, method, use_hip=True, cfg={"exchange": "native-hosted"})
    if grid != (2, 1):
        run_bf16(grid, method, use_hip=True, cfg={"exchange": "native-hosted", "shape": (256, 192, 16, 6)})


@pytest.mark.parametrize("grid,method", [((2, 1), "mu"), long_param((1, 2), "mu"), ((2, 2), "mu"), long_param((2, 1), "hals"), long_param((4, 1), "mu")])
def test_multirank_hip_bf16x6_gemm(grid, method):
    """params.gemm = 'bf16x6' on a grid: the split kernels on every rank's block (k = 40, local n a multiple of 128; the
    4 x 1 grid takes the overlapped H phase on column halves of A), checked against the oracle's grid simulation at the fp32
    path's tolerances."""
    from tests._mp import run_bf16
    run_bf16(grid, method, use_hip=True, cfg={"shape": (512, 512, 40, 8), "precision": "float32", "gemm": "bf16x6",
                                              "overlap_min_cols": 128})


@pytest.mark.parametrize("grid", [long_param((2, 1)), long_param((1, 2)), (2, 2)])
def test_multirank_hip_bf16x6_kl(grid):
    """KL updates with params.gemm = 'bf16x6' on a grid (k = 12: the split KL kernels on every block, 2D slices included)."""
    from tests._mp import run_bf16
    run_bf16(grid, "mu", use_hip=True, cfg={"shape": (512, 512, 12, 8), "precision": "float32", "gemm": "bf16x6", "norm": "kl"})


@pytest.mark.parametrize("grid,method", [((2, 1), "mu"), long_param((2, 2), "hals")])
def test_multirank_hip_bf16_storage_with_bf16x6(grid, method):
    """bf16-stored blocks AND params.gemm = 'bf16x6' on a grid: the three-product kernels against the oracle on float(bf16(A))."""
    from tests._mp import run_bf16
    run_bf16(grid, method, use_hip=True, cfg={"shape": (512, 512, 40, 8), "precision": "bfloat16", "gemm": "bf16x6"})


def test_multirank_hip_overlapped_h_phase():
    """The chunked / overlapped H phase of row grids with more than two ranks, real kernels on column views of A and H."""
    from tests._mp import run_case
    run_case("swim_4x1_fro_float32", use_hip=True, timeout=400, extra={"overlap_min_cols": 32, "overlap_chunks": 4})
