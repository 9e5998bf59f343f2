"""pydnmfk_amd -- MI355X-native multiplicative-update engine behind pyDNMFk's PyNMF / nmf_algorithms API.

Host side mirrors the reference module names (pyDNMF, dist_nmf, dist_comm, utils, data_io); the arithmetic
lives in libdnmf_hip.so (csrc/dnmf.hip + csrc/dnmf_*.h, C ABI in include/dnmf.h).  Importing `engine` (or running
anything numeric) requires the built library: there is no CPU fallback.
"""
__version__ = "0.1.0"
