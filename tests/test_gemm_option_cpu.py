"""The opt-in `gemm` knob (bf16x6 contractions) on the host side: parser flag, operator-set selection, ABI surface.
No kernel runs here (CPU tier)."""
import pytest


def test_cli_flag_and_default():
    import main
    ap = main.build_parser()
    ns = ap.parse_args(["--p_r", "1", "--p_c", "1"])
    assert ns.gemm == "fp32"
    assert ap.parse_args(["--p_r", "1", "--p_c", "1", "--gemm", "bf16x6"]).gemm == "bf16x6"


def test_ops_for_selects_the_operator_set():
    from pydnmfk_amd.engine import HIP_OPS, HIP_OPS_BF16X6, ops_for
    from pydnmfk_amd.utils import parse
    p = parse()
    assert ops_for(p) is HIP_OPS and ops_for(None) is HIP_OPS          # absent -> the fp32 MFMA path (the parity reference)
    p.gemm = "fp32"
    assert ops_for(p) is HIP_OPS
    p.gemm = "bf16x6"
    assert ops_for(p) is HIP_OPS_BF16X6 and HIP_OPS_BF16X6.name == "hip-bf16x6"
    p.gemm = "fp16"
    with pytest.raises(ValueError):
        ops_for(p)


def test_split_entry_points_are_declared_and_exported():
    """every bf16x6 entry point of include/dnmf.h is in the ctypes table and resolves in the library"""
    import os
    import re
    from pydnmfk_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "dnmf.h")).read()
    names = sorted(set(re.findall(r"\b(dnmf_\w*bf16x6)\s*\(", hdr)))
    assert len(names) == 12, names
    for n in names:
        assert n in _lib.SIGNATURES, n
        assert getattr(_lib.lib, n) is not None
    # the workspace query is pure host arithmetic: larger than the fp32 one where split kernels exist, equal where not
    base = _lib.lib.dnmf_ws_bytes(4096, 1024, 64)
    assert _lib.lib.dnmf_ws_bytes_bf16x6(4096, 1024, 64) > base
    base_ragged = _lib.lib.dnmf_ws_bytes(4096, 1000, 64)               # n % 128 != 0: no split kernel, nothing extra
    assert _lib.lib.dnmf_ws_bytes_bf16x6(4096, 1000, 64) == (base_ragged + 255) // 256 * 256
    assert _lib.lib.dnmf_ws_bytes_bf16x6(0, 1024, 64) == 0


def test_operator_set_refuses_cpu_tensors():
    torch = pytest.importorskip("torch")
    from pydnmfk_amd.engine import HIP_OPS_BF16X6 as x6
    A, W, H = torch.rand(256, 128), torch.rand(256, 40), torch.rand(40, 128)
    with pytest.raises(TypeError):
        x6.wta(A, W, torch.empty(40, 128))
    with pytest.raises(TypeError):
        x6.mu_kl_step(A, W, H, 1e-7)
