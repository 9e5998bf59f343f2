"""ctypes binding of libdnmf_hip.so (C ABI declared in include/dnmf.h).

There is NO CPU fallback: if the library is missing the import of this module fails
loudly, and every compute entry point requires CUDA(HIP) device pointers.
"""
import ctypes
import os

# torch bundles its own libamdhip64; it MUST be loaded first so that libdnmf_hip.so binds to the same HIP
# runtime instance (two runtimes in one process = "no ROCm-capable device" + foreign device pointers).
import torch  # noqa: F401  isort:skip

_HERE = os.path.dirname(os.path.abspath(__file__))
# DNMF_LIB_PATH: measurement tools point this at the tuning build (tools/_build/libdnmf_hip_tune.so, same ABI)
LIB_PATH = os.environ.get("DNMF_LIB_PATH") or os.path.join(_HERE, "libdnmf_hip.so")

c_float_p = ctypes.c_void_p
c_long, c_int, c_float, c_size_t, c_void_p = ctypes.c_long, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/dnmf.h one to one
SIGNATURES = {
    "dnmf_last_error": [],
    "dnmf_version": [],
    "dnmf_kp": [c_int],
    "dnmf_ws_bytes": [c_long, c_long, c_int],
    "dnmf_gram_hht": [c_void_p, c_int, c_long, c_long, c_void_p, c_void_p, c_size_t, c_void_p],
    "dnmf_gram_wtw": [c_void_p, c_long, c_int, c_long, c_void_p, c_void_p, c_size_t, c_void_p],
    "dnmf_aht": [c_void_p, c_long, c_long, c_long, c_void_p, c_int, c_long, c_void_p, c_long, c_void_p],
    "dnmf_aht_hblocks": [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_int, c_void_p, c_long, c_void_p],
    "dnmf_wta": [c_void_p, c_long, c_long, c_long, c_void_p, c_int, c_long, c_void_p, c_long, c_void_p, c_size_t,
                 c_void_p],
    "dnmf_wta_gram": [c_void_p, c_long, c_long, c_long, c_void_p, c_int, c_long, c_void_p, c_long, c_void_p, c_void_p,
                      c_size_t, c_void_p],
    "dnmf_mu_update_w": [c_void_p, c_long, c_int, c_long, c_void_p, c_long, c_void_p, c_float, c_void_p],
    "dnmf_mu_update_h": [c_void_p, c_int, c_long, c_long, c_void_p, c_long, c_void_p, c_float, c_int, c_void_p],
    "dnmf_aht_update_w": [c_void_p, c_long, c_long, c_long, c_void_p, c_int, c_long, c_void_p, c_void_p, c_long,
                          c_float, c_void_p],
    "dnmf_mu_fro_step": [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_float, c_int,
                         c_int, c_void_p, c_size_t, c_void_p],
    "dnmf_mu_fro_onepass": [c_long, c_long, c_int],
    "dnmf_set_onepass": [c_int],
    "dnmf_set_persistent": [c_int],
    "dnmf_hals_w_col": [c_void_p, c_long, c_int, c_long, c_void_p, c_long, c_void_p, c_int, c_void_p, c_float, c_void_p,
                        c_void_p],
    "dnmf_hals_w_scale": [c_void_p, c_long, c_long, c_int, c_void_p, c_void_p],
    "dnmf_hals_update_w": [c_void_p, c_long, c_int, c_long, c_void_p, c_long, c_void_p, c_float, c_void_p, c_void_p],
    "dnmf_hals_sweep_w": [c_void_p, c_long, c_int, c_long, c_void_p, c_long, c_void_p, c_float, c_void_p, c_size_t, c_void_p],
    "dnmf_hals_sweep_status": [ctypes.POINTER(c_int), c_void_p],
    "dnmf_hals_update_h": [c_void_p, c_int, c_long, c_long, c_void_p, c_long, c_void_p, c_float, c_void_p],
    "dnmf_kl_uht": [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_float, c_void_p,
                    c_long, c_void_p, c_size_t, c_void_p],
    "dnmf_kl_wtu": [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_float, c_void_p,
                    c_long, c_void_p, c_size_t, c_void_p],
    "dnmf_kl_uht_hblocks": [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_float, c_void_p,
                            c_long, c_void_p, c_size_t, c_void_p],
    "dnmf_rowsum": [c_void_p, c_int, c_long, c_long, c_void_p, c_void_p],
    "dnmf_colsum": [c_void_p, c_long, c_int, c_long, c_void_p, c_void_p, c_size_t, c_void_p],
    "dnmf_kl_update_w": [c_void_p, c_long, c_int, c_long, c_void_p, c_long, c_void_p, c_float, c_void_p],
    "dnmf_kl_update_h": [c_void_p, c_int, c_long, c_long, c_void_p, c_long, c_void_p, c_float, c_int, c_void_p],
    "dnmf_mu_kl_step": [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_float, c_int,
                        c_int, c_void_p, c_size_t, c_void_p],
    "dnmf_clamp_min": [c_void_p, c_long, c_long, c_long, c_float, c_void_p],
    "dnmf_scale_cols_div": [c_void_p, c_long, c_int, c_long, c_void_p, c_float, c_void_p],
    "dnmf_scale_rows_mul": [c_void_p, c_int, c_long, c_long, c_void_p, c_void_p],
    "dnmf_sqnorm": [c_void_p, c_long, c_long, c_long, c_void_p, c_void_p],
    "dnmf_clock_probe": [c_void_p, c_int, c_int, c_void_p],
    "dnmf_perturb_uniform": [c_void_p, c_void_p, c_long, c_long, c_long, c_long, c_float, ctypes.c_ulonglong, c_int, c_void_p],
    "dnmf_column_err": [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_void_p, c_void_p,
                        c_void_p],
    "dnmf_resid_sqnorm": [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_void_p,
                          c_void_p],
}
SIGNATURES["dnmf_resid_sqnorm_ws"] = SIGNATURES["dnmf_resid_sqnorm"][:-1] + [c_void_p, c_size_t, c_void_p]
# bf16 storage of A: same argument lists as the fp32 twins (A is passed as a device pointer either way)
for _n in ("aht", "wta", "wta_gram", "aht_update_w", "mu_fro_step", "sqnorm", "resid_sqnorm", "resid_sqnorm_ws", "column_err"):
    SIGNATURES["dnmf_%s_bf16a" % _n] = SIGNATURES["dnmf_" + _n]
# bf16x6 contractions: the fp32 argument lists, aht / aht_update_w with a workspace added
SIGNATURES["dnmf_ws_bytes_bf16x6"] = SIGNATURES["dnmf_ws_bytes"]
SIGNATURES["dnmf_aht_bf16x6"] = SIGNATURES["dnmf_aht"][:-1] + [c_void_p, c_size_t, c_void_p]
SIGNATURES["dnmf_wta_bf16x6"] = SIGNATURES["dnmf_wta"]
SIGNATURES["dnmf_aht_update_w_bf16x6"] = SIGNATURES["dnmf_aht_update_w"][:-1] + [c_void_p, c_size_t, c_void_p]
SIGNATURES["dnmf_mu_fro_step_bf16x6"] = SIGNATURES["dnmf_mu_fro_step"]
for _n in ("aht", "wta", "aht_update_w", "mu_fro_step"):
    SIGNATURES["dnmf_%s_bf16a_bf16x6" % _n] = SIGNATURES["dnmf_%s_bf16x6" % _n]
SIGNATURES["dnmf_kl_uht_bf16x6"] = SIGNATURES["dnmf_kl_uht"]
SIGNATURES["dnmf_kl_wtu_bf16x6"] = SIGNATURES["dnmf_kl_wtu"]
SIGNATURES["dnmf_mu_kl_step_bf16x6"] = SIGNATURES["dnmf_mu_kl_step"]
# grid exchanges inside the library (csrc/dnmf_comm.hip); the communicator handle is an opaque pointer
SIGNATURES["dnmf_comm_unique_id"] = [c_void_p]
SIGNATURES["dnmf_comm_create"] = [c_void_p, c_int, c_int, c_int, c_int, ctypes.POINTER(c_void_p)]
COLLECTIVE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p)   # dnmf_collective_fn
SIGNATURES["dnmf_comm_create_hosted"] = [c_int, c_int, c_int, c_int, COLLECTIVE_FN, c_void_p, ctypes.POINTER(c_void_p)]
SIGNATURES["dnmf_comm_create_emulated"] = [c_int, c_int, c_int, ctypes.POINTER(c_void_p)]
SIGNATURES["dnmf_comm_destroy"] = [c_void_p]
SIGNATURES["dnmf_comm_rccl_version"] = [ctypes.POINTER(c_int), c_void_p, c_size_t]
SIGNATURES["dnmf_comm_direct_init"] = [c_void_p, c_size_t, c_void_p]
SIGNATURES["dnmf_comm_direct_connect"] = [c_void_p, c_void_p]
SIGNATURES["dnmf_comm_set_direct"] = [c_void_p, c_int]
SIGNATURES["dnmf_comm_direct_teardown"] = [c_void_p]
SIGNATURES["dnmf_comm_hals_xsweeps"] = [c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]
SIGNATURES["dnmf_comm_set_direct_timeout"] = [c_void_p, ctypes.c_double]
SIGNATURES["dnmf_comm_fit_begin"] = [c_void_p]
DIRECT_HANDLE_BYTES = 80           # DNMF_DIRECT_HANDLE_BYTES
SIGNATURES["dnmf_comm_allreduce_direct"] = [c_void_p, c_void_p, c_size_t, c_void_p]
SIGNATURES["dnmf_comm_allreduce_direct_f64"] = [c_void_p, c_void_p, c_size_t, c_void_p]
SIGNATURES["dnmf_comm_direct_status"] = [c_void_p, ctypes.POINTER(c_int)]
SIGNATURES["dnmf_comm_info"] = [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]
SIGNATURES["dnmf_comm_set_overlap_chunks"] = [c_void_p, c_int]
SIGNATURES["dnmf_comm_set_always_exchange"] = [c_void_p, c_int]
SIGNATURES["dnmf_comm_set_null_exchange"] = [c_void_p, c_int]
SIGNATURES["dnmf_comm_allreduce"] = [c_void_p, c_void_p, c_size_t, c_int, c_void_p]
SIGNATURES["dnmf_ws_bytes_1d"] = SIGNATURES["dnmf_ws_bytes"]
SIGNATURES["dnmf_ws_bytes_2d"] = [c_long, c_long, c_int, c_int, c_int]
SIGNATURES["dnmf_hals_fro_step_1d"] = [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_float, c_int, c_int,
                                       c_int, c_void_p, c_size_t, c_void_p, c_void_p]
for _n in ("fro", "kl"):      # A, m_l, n_l, lda, W, m_w, ldw, H, n_h, ldh, k, eps, w_update, clamp, ws, ws_bytes, comm, stream
    SIGNATURES["dnmf_mu_%s_step_2d" % _n] = [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_long, c_void_p, c_long, c_long,
                                             c_int, c_float, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p]
SIGNATURES["dnmf_hals_fro_step_2d"] = SIGNATURES["dnmf_mu_fro_step_2d"]
SIGNATURES["dnmf_mu_fro_step_1d"] = SIGNATURES["dnmf_mu_fro_step"][:-1] + [c_void_p, c_void_p]
SIGNATURES["dnmf_mu_kl_step_1d"] = SIGNATURES["dnmf_mu_fro_step_1d"]
SIGNATURES["dnmf_ws_bytes_hblocks"] = [c_long, c_long, c_int, c_long]
for _n in ("mu_fro_step_1d", "mu_fro_step_2d", "hals_fro_step_1d", "hals_fro_step_2d"):     # bf16-stored A: same argument lists
    SIGNATURES["dnmf_%s_bf16a" % _n] = SIGNATURES["dnmf_" + _n]
# whole fits (csrc/dnmf_fit.hip): A, m, n, lda, W, ldw, H, ldh, k, eps, w_update, itr, [column_sweep,] batch, a_stride, w_stride, h_stride,
# sq_out, ws, ws_bytes, stream
SIGNATURES["dnmf_ws_bytes_fit"] = [c_long, c_long, c_int, c_int]
SIGNATURES["dnmf_mu_fit_persistent"] = [c_long, c_long, c_int]
SIGNATURES["dnmf_hals_fit_persistent"] = [c_long, c_long, c_int]
SIGNATURES["dnmf_fit_set_timeout"] = [ctypes.c_double]
SIGNATURES["dnmf_mu_fro_fit"] = [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_float, c_int, c_int,
                                 c_int, c_long, c_long, c_long, c_void_p, c_void_p, c_size_t, c_void_p]
SIGNATURES["dnmf_mu_kl_fit"] = SIGNATURES["dnmf_mu_fro_fit"]
SIGNATURES["dnmf_mu_fro_fit_bf16a"] = SIGNATURES["dnmf_mu_fro_fit"]
SIGNATURES["dnmf_hals_fro_fit"] = SIGNATURES["dnmf_mu_fro_fit"][:12] + [c_int] + SIGNATURES["dnmf_mu_fro_fit"][12:]
SIGNATURES["dnmf_hals_fro_fit_bf16a"] = SIGNATURES["dnmf_hals_fro_fit"]
# the float64 path (csrc/dnmf_f64.hip)
c_double = ctypes.c_double
SIGNATURES["dnmf_f64_ws_bytes"] = [c_long, c_long, c_int]
SIGNATURES["dnmf_f64_aht"] = [c_void_p, c_long, c_long, c_long, c_void_p, c_int, c_long, c_void_p, c_long, c_void_p, c_size_t, c_void_p]
SIGNATURES["dnmf_f64_wta"] = SIGNATURES["dnmf_f64_aht"]
SIGNATURES["dnmf_f64_ws_bytes_fit"] = [c_long, c_long, c_int]
SIGNATURES["dnmf_f64_fit_tiny"] = [c_long, c_long, c_int, c_int]
SIGNATURES["dnmf_f64_fit"] = [c_int, c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_double, c_int, c_int, c_void_p, c_void_p,
                              c_size_t, c_void_p]
SIGNATURES["dnmf_f64_mu_update_w"] = [c_void_p, c_long, c_int, c_long, c_void_p, c_long, c_void_p, c_long, c_double, c_void_p]
SIGNATURES["dnmf_f64_mu_update_h"] = [c_void_p, c_int, c_long, c_long, c_void_p, c_long, c_void_p, c_long, c_double, c_int, c_void_p]
SIGNATURES["dnmf_f64_kl_quot"] = [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_double, c_void_p, c_long, c_void_p]
for _n in ("dnmf_f64_kl_uht", "dnmf_f64_kl_wtu"):      # A m n lda W ldw H ldh k eps S lds U ws ws_bytes stream
    SIGNATURES[_n] = [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_double, c_void_p, c_long, c_void_p, c_void_p, c_size_t, c_void_p]
SIGNATURES["dnmf_f64_sqdiff"] = [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_void_p, c_long, c_void_p]
SIGNATURES["dnmf_f64_sum"] = [c_void_p, c_long, c_long, c_long, c_int, c_void_p, c_void_p, c_size_t, c_void_p]
SIGNATURES["dnmf_f64_colsum"] = SIGNATURES["dnmf_f64_sum"]
SIGNATURES["dnmf_f64_rowsum"] = [c_void_p, c_int, c_long, c_long, c_void_p, c_void_p]
SIGNATURES["dnmf_f64_ew"] = [c_int, c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_void_p, c_double, c_int, c_void_p]
SIGNATURES["dnmf_f64_hals_w_col"] = [c_void_p, c_long, c_int, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_void_p, c_double, c_void_p,
                                     c_void_p, c_size_t, c_void_p]
SIGNATURES["dnmf_f64_hals_w_scale"] = [c_void_p, c_long, c_long, c_int, c_void_p, c_void_p]
SIGNATURES["dnmf_f64_hals_update_h"] = [c_void_p, c_int, c_long, c_long, c_void_p, c_long, c_void_p, c_long, c_double, c_void_p]
_RESTYPES = {"dnmf_ws_bytes_fit": c_size_t, "dnmf_f64_ws_bytes": c_size_t, "dnmf_f64_ws_bytes_fit": c_size_t, "dnmf_last_error": ctypes.c_char_p, "dnmf_ws_bytes": c_size_t, "dnmf_ws_bytes_bf16x6": c_size_t,
             "dnmf_ws_bytes_1d": c_size_t, "dnmf_ws_bytes_hblocks": c_size_t, "dnmf_ws_bytes_2d": c_size_t}


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "pydnmfk_amd: %s not found. Build it with `python -m pydnmfk_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI drifted from include/dnmf.h
        fn.argtypes = args
        fn.restype = _RESTYPES.get(name, c_int)
    return lib


lib = load()


class DnmfError(RuntimeError):
    pass


class PersistentTimeout(DnmfError):
    """A kernel whose workgroups wait for each other gave up (dnmf_hals_sweep_status): its workgroups were not all resident -- the GPU is
    shared with another process or stream.  The factors it was working on are invalid."""


def check(rc):
    if rc != 0:
        raise DnmfError("libdnmf_hip: %s (code %d)" % (lib.dnmf_last_error().decode(), rc))
