"""cProfile of the swim KL known-answer sweep on one rank (tests/test_gpu_nmfk.py::test_swim_kl_known_answer_on_one_rank)."""
import cProfile, os, pstats, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pydnmfk_amd.dist_comm import MPI_comm
from pydnmfk_amd.pyDNMFk import PyNMFk
from pydnmfk_amd.utils import parse
A = torch.from_numpy(np.ascontiguousarray(np.load("tests/golden/data_swim.npz")["A"].astype(np.float32))).cuda()
def run():
    comms = MPI_comm(None, 1, 1)
    args = parse()
    args.size, args.rank, args.comm, args.p_r, args.p_c = 1, 0, comms, 1, 1
    args.row_comm, args.col_comm, args.comm1 = comms.cart_1d_row(), comms.cart_1d_column(), comms.comm
    args.fpath, args.fname, args.ftype = "../data/", "swim", "mat"
    args.start_k, args.end_k, args.sill_thr, args.itr, args.init = 14, 18, 0.6, int(os.environ.get("ITR", "5000")), "rand"
    args.noise_var, args.verbose, args.norm, args.method, args.checkpoint = 0.016, False, "kl", "mu", False
    args.prune, args.rng, args.results_path = False, "device", tempfile.mkdtemp() + "/"
    t0 = time.time(); n = PyNMFk(A, factors=None, params=args).fit(); torch.cuda.synchronize()
    return n, time.time() - t0
print("warm", run())
pr = cProfile.Profile(); pr.enable(); out = run(); pr.disable()
print("profiled", out)
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
