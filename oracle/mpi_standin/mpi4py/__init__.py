"""Thread-simulated stand-in for the `mpi4py` package.

TEST INFRASTRUCTURE ONLY.  mpi4py is not installed in the build container, so
the reference (lanl/pyDNMFk, pure Python) cannot be imported as-is.  This
package provides exactly the slice of the mpi4py API that the reference's MU
hot path touches, simulating P ranks as P threads of
one process.  It exists only so that `tests/golden/make_golden.py` can run the
UNMODIFIED reference here and capture golden vectors; it is never imported by
the product package.
"""
from . import MPI  # noqa: F401
