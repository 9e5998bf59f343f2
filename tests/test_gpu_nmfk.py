"""NMFk driver on the GPU (real HIP kernels inside every PyNMF.fit) against the statistics captured from the reference's
PyNMFk (tests/golden/nmfk_1x1.npz): same data, seeds, parameters; numpy input -> the reference's numpy RNG stream."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_nmfk_on_gpu_matches_reference_statistics(tmp_path, golden_dir):
    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.pyDNMFk import PyNMFk
    from tests.test_nmfk_cpu import _args, check_against_golden
    z = np.load(golden_dir + "/nmfk_1x1.npz")
    comms = MPI_comm(None, 1, 1)
    nmfk = PyNMFk(z["A"], factors=None, params=_args(tmp_path, comms))
    nopt = nmfk.fit()
    check_against_golden(nmfk, nopt, z)


def test_nmfk_device_resident_input(tmp_path, golden_dir):
    """CUDA-tensor input: perturbations are drawn on the device (different stream, same distribution) -> the estimate
    and the error levels still match, silhouettes are compared loosely."""
    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.pyDNMFk import PyNMFk
    from tests.test_nmfk_cpu import _args
    z = np.load(golden_dir + "/nmfk_1x1.npz")
    comms = MPI_comm(None, 1, 1)
    np.random.seed(123)
    nmfk = PyNMFk(torch.from_numpy(z["A"]).cuda(), factors=None, params=_args(tmp_path, comms))
    assert nmfk.fit() == 3
    for k, tol in ((1, 0.05), (2, 0.05), (3, 0.6)):   # k = 3 is at the noise floor: its level depends on the draws
        assert abs(nmfk.stats[k]["avgErr"] / float(z["k%d_avgErr" % k]) - 1) < tol, k
    assert np.min(nmfk.stats[3]["clusterSilhouetteCoefficients"]) > 0.8
