"""Readers / writers on the host: block partition of npy / csv / mat / folder inputs (reference data_io.py:12-105,
tests/test_dist_file_split.py) and re-assembly of saved factors (data_io.py:212-261)."""
import numpy as np
import pytest

from tests._golden import GOLDEN


class _FakeComm:
    def __init__(self, rank):
        self.rank = rank


def _args(tmp, ftype, fname, rank, p_r, p_c):
    from pydnmfk_amd.utils import parse
    a = parse()
    a.fpath, a.ftype, a.fname, a.p_r, a.p_c, a.comm1, a.precision = str(tmp) + "/", ftype, fname, p_r, p_c, _FakeComm(rank), "float32"
    return a


@pytest.mark.parametrize("ftype", ["npy", "csv", "mat"])
def test_block_reads(tmp_path, ftype):
    from scipy.io import savemat
    from pydnmfk_amd.data_io import data_read
    X = np.load(GOLDEN + "/data_swim.npz")["A"][:96, :21].astype(np.float64)   # same shape as wtsi (96 x 21)
    if ftype == "npy":
        np.save(tmp_path / "x.npy", X)
    elif ftype == "csv":
        np.savetxt(tmp_path / "x.csv", X, delimiter=",")
    else:
        savemat(tmp_path / "x.mat", {"X": X})
    # known answer of the reference's tests/test_dist_file_split.py:25-30 on a (2,1) grid
    b0 = data_read(_args(tmp_path, ftype, "x", 0, 2, 1)).read()
    b1 = data_read(_args(tmp_path, ftype, "x", 1, 2, 1)).read()
    assert b0.shape == b1.shape == (48, 21) and b0.dtype == np.float32 and b0.flags.c_contiguous
    assert np.array_equal(b0, X[:48].astype(np.float32)) and np.array_equal(b1, X[48:].astype(np.float32))
    # ragged 2D grid
    blocks = [data_read(_args(tmp_path, ftype, "x", r, 3, 2)).read() for r in range(6)]
    assert [b.shape for b in blocks] == [(32, 11), (32, 10)] * 3
    assert np.array_equal(np.block([[blocks[0], blocks[1]], [blocks[2], blocks[3]], [blocks[4], blocks[5]]]), X.astype(np.float32))


def test_folder_read_and_factor_roundtrip(tmp_path):
    from pydnmfk_amd.data_io import data_read, data_write, read_factors
    from pydnmfk_amd.utils import parse
    rs = np.random.RandomState(0)
    for r in range(2):
        np.save(tmp_path / ("blk%d.npy" % r), rs.rand(5, 4))
    b = data_read(_args(tmp_path, "folder", "blk", 1, 2, 1)).read()
    assert b.shape == (5, 4) and b.dtype == np.float32
    # save_factors on a 2x3 grid, then read_factors re-assembles W (rows) and H (columns) in global order
    p_r, p_c, m, n, k = 2, 3, 12, 18, 2
    W, H = rs.rand(m, k).astype(np.float32), rs.rand(k, n).astype(np.float32)
    from oracle import nmf_oracle as orc
    for rank in range(p_r * p_c):
        (w0, w1), (h0, h1) = orc.factor_ranges(rank, p_r, p_c, m, n)
        a = parse()
        a.p_r, a.p_c, a.comm1, a.results_paths, a.ftype = p_r, p_c, _FakeComm(rank), str(tmp_path) + "/res/", None
        data_write(a).save_factors([W[w0:w1], H[:, h0:h1]], reg=True)
    W2, H2 = read_factors(str(tmp_path) + "/res/", (p_r, p_c)).load_factors()
    assert np.array_equal(W2, W) and np.array_equal(H2, H)
