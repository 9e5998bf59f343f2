"""The C-ABI library loads and exports every symbol include/dnmf.h declares; argument validation
works without touching a GPU (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "dnmf.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dnmf_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported_and_bound():
    from pydnmfk_amd import _lib
    names = _declared()
    assert len(names) >= 24
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "symbol %s declared in include/dnmf.h but not exported" % n
        assert n in _lib.SIGNATURES, "symbol %s has no ctypes signature in pydnmfk_amd/_lib.py" % n
    assert sorted(_lib.SIGNATURES) == names


def test_argument_validation_without_gpu():
    from pydnmfk_amd._lib import lib
    assert lib.dnmf_version() >= 100
    assert [lib.dnmf_kp(k) for k in (1, 4, 32, 33, 64, 65, 128, 129, 256)] == [32, 32, 32, 64, 64, 128, 128, 256, 256]
    assert lib.dnmf_kp(0) < 0 and lib.dnmf_kp(257) < 0
    # wide ranks (128 < k <= 256): the workspace carries the m x n quotient image of the KL products; the library-sequenced grid
    # steps and the float64 path stop at the tuned kernels' rank
    assert lib.dnmf_ws_bytes(4096, 1024, 200) >= lib.dnmf_ws_bytes(4096, 1024, 128) + 4 * 4096 * 1024
    assert lib.dnmf_ws_bytes_1d(4096, 1024, 200) == 0 and lib.dnmf_f64_ws_bytes(4096, 1024, 200) == 0
    assert lib.dnmf_aht_update_w(None, 8, 8, 8, None, 200, 8, None, None, 200, 1e-7, None) == -1 and b"fused form" in lib.dnmf_last_error()
    assert lib.dnmf_ws_bytes(0, 10, 4) == 0
    assert lib.dnmf_ws_bytes(262144, 8192, 64) > 64 * 8192 * 4
    # null pointers / bad rank are rejected before any HIP call
    assert lib.dnmf_aht(None, 8, 8, 8, None, 4, 8, None, 4, None) == -1
    assert b"aht" in lib.dnmf_last_error()
    assert lib.dnmf_mu_fro_step(None, 8, 8, 8, None, 4, None, 8, 300, 1e-7, 1, 0, None, 0, None) == -1
    # the exchange entry points (csrc/dnmf_comm.hip): argument checks come before any RCCL / HIP call
    assert lib.dnmf_ws_bytes_1d(0, 10, 4) == 0
    assert lib.dnmf_ws_bytes_1d(32768, 8192, 64) >= lib.dnmf_ws_bytes(32768, 8192, 64) + 4 * (64 * 8192 + 64 * 64)
    assert lib.dnmf_comm_create(None, 2, 0, 2, 1, None) == -1 and b"comm_create" in lib.dnmf_last_error()
    assert lib.dnmf_comm_unique_id(None) == -1
    assert lib.dnmf_comm_allreduce(None, None, 4, 0, None) == -1
    assert lib.dnmf_mu_fro_step_1d(None, 8, 8, 8, None, 4, None, 8, 4, 1e-7, 1, 0, None, 0, None, None) == -1
    assert lib.dnmf_mu_kl_step_1d(None, 8, 8, 8, None, 4, None, 8, 4, 1e-7, 1, 0, None, 0, None, None) == -1
    # 2D steps: the workspace query is host arithmetic (ragged grids included; 0 for a grid with more members than rows / columns
    # or a bad grid), null arguments are refused
    assert lib.dnmf_ws_bytes_2d(32768, 32768, 128, 4, 2) >= lib.dnmf_ws_bytes(32768, 32768, 128) + 4 * 128 * (2 * 32768 + 2 * 32768)
    assert lib.dnmf_ws_bytes_2d(1000, 1000, 8, 3, 2) > lib.dnmf_ws_bytes(1000, 1000, 8)
    assert lib.dnmf_ws_bytes_2d(2, 1000, 8, 2, 3) == 0 and lib.dnmf_ws_bytes_2d(1000, 1024, 8, 2, 0) == 0
    assert lib.dnmf_mu_fro_step_2d(None, 8, 8, 8, None, 8, 4, None, 8, 8, 4, 1e-7, 1, 0, None, 0, None, None) == -1
    assert b"mu_fro_step_2d" in lib.dnmf_last_error()
    assert lib.dnmf_mu_kl_step_2d(None, 8, 8, 8, None, 8, 4, None, 8, 8, 4, 1e-7, 1, 0, None, 0, None, None) == -1
    assert lib.dnmf_hals_fro_step_1d(None, 8, 8, 8, None, 4, None, 8, 4, 1e-7, 1, 0, 0, None, 0, None, None) == -1
    assert lib.dnmf_hals_fro_step_2d(None, 8, 8, 8, None, 8, 4, None, 8, 8, 4, 1e-7, 1, 0, None, 0, None, None) == -1
    assert lib.dnmf_hals_fro_step_1d_bf16a(None, 8, 8, 8, None, 4, None, 8, 4, 1e-7, 1, 0, 0, None, 0, None, None) == -1
    assert lib.dnmf_mu_fro_step_2d_bf16a(None, 8, 8, 8, None, 8, 4, None, 8, 8, 4, 1e-7, 1, 0, None, 0, None, None) == -1
    assert lib.dnmf_wta_gram(None, 8, 8, 8, None, 4, 4, None, 8, None, None, 0, None) == -1
    assert lib.dnmf_clock_probe(None, 4, 1, None) == -1
    assert lib.dnmf_comm_destroy(None) == 0
