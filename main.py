#!/usr/bin/env python3
"""Command-line entry point, flag-compatible with the reference's main.py (:13-88), one process per GPU:

    python -m torch.distributed.run --nproc-per-node 4 main.py --process=pyDNMF --p_r=4 --p_c=1 \\
        --fpath=data/ --fname=swim --ftype=mat --k=4 --itr=100 --norm=fro --method=mu --results_path=results/

(reference: `mpirun -n 4 python main.py ...`).  A single process needs no launcher.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _flag(text):
    """yes/no style command-line booleans"""
    if isinstance(text, bool):
        return text
    word = text.strip().lower()
    if word in {'1', 'y', 'yes', 't', 'true'}:
        return True
    if word in {'0', 'n', 'no', 'f', 'false'}:
        return False
    raise NameError('Boolean value expected.')


# The reference CLI surface (main.py:13-58), kept flag for flag so existing launch scripts work:
#   (flag, type, default, required, help)
FLAGS = [
    # single factorisation
    ('process', str, 'pyDNMF', False, 'pyDNMF (one factorisation) or pyDNMFk (rank estimation)'),
    ('p_r', int, None, True, 'process-grid rows'),
    ('p_c', int, None, True, 'process-grid columns'),
    ('k', int, 4, False, 'rank of the factorisation'),
    ('fpath', str, 'data/', False, 'directory of the input (with trailing slash)'),
    ('ftype', str, 'mat', False, 'input format: mat / npy / csv / txt / folder'),
    ('fname', str, 'swim', False, 'input file name without extension'),
    ('init', str, 'rand', False, 'factor initialisation: rand / nnsvd'),
    ('itr', int, 5000, False, 'update iterations'),
    ('norm', str, 'kl', False, 'objective: kl / fro'),
    ('method', str, 'mu', False, 'update rule: mu / hals'),
    ('verbose', _flag, False, False, 'print the relative error of every fit'),
    ('results_path', str, 'results/', False, 'output directory'),
    ('checkpoint', _flag, False, False, 'keep a coarse NMFk checkpoint'),
    ('timing_stats', _flag, False, False, 'accepted for compatibility; ignored'),
    ('prune', _flag, False, False, 'drop all-zero rows / columns before factorising'),
    ('precision', str, 'float32', False, 'dtype the input file is cast to (reference main.py:29): float32 (the tuned path), float64 (factorised in float64 on the fp64 matrix cores), float16 (read as such, computed in float32) or bfloat16 (storage of the data on the GPU, Frobenius mu / hals; fp32 arithmetic)'),
    ('gemm', str, 'fp32', False, 'arithmetic of the two big Frobenius contractions: fp32 (fp32 MFMA) or bf16x6 (six bf16 piece products, fp32-grade)'),
    ('rng', str, 'device', False, 'where random numbers are drawn: device (the data block goes to the GPU once, perturbations and the rand init are drawn there) or numpy (the reference\'s host stream: every fit draws and uploads host arrays)'),
    ('exchange', str, 'auto', False, 'who sequences the exchanges of a multi-rank step: torch (torch.distributed between the kernel launches), native (whole steps inside libdnmf_hip.so over its own RCCL communicators: one call per step) or auto (native for hals on grids with p_r > 1, where the per-column norm exchanges make the Python-sequenced step host-bound; torch otherwise)'),
    ('direct_allreduce', _flag, False, False, 'with --exchange native on one node: the packed allreduce of a 1D step goes through IPC-mapped peer buffers (two-shot, rank-ordered sums) instead of RCCL, and the HALS W sweep on p_r > 1 runs as ONE persistent launch whose column norms cross the ranks through slots in those buffers; checked against RCCL on first contact, all ranks fall back together'),
    ('shared_gpu', _flag, False, False, 'the GPU is shared with other processes or streams: never use the kernels whose workgroups wait for each other (whole fits of small problems, the one-launch HALS W sweep, the one-pass MU/FRO step) -- dnmf_set_persistent(0); without it a fit that loses its residency is detected and fitted again on the launch-chain kernels, at the cost of one time-out'),
    ('hals_sweep', str, 'persistent', False, 'W sweep of method hals on a rank with local norms: persistent (one launch; needs the GPU to itself) or columns'),
    # NMFk
    ('perturbations', int, 20, False, 'perturbed copies per rank'),
    ('noise_var', float, 0.015, False, 'perturbation amplitude'),
    ('start_k', int, 1, False, 'first rank of the sweep'),
    ('end_k', int, 10, False, 'last rank of the sweep'),
    ('step_k', int, 1, False, 'rank increment'),
    ('sill_thr', float, 0.6, False, 'silhouette threshold of the rank estimate'),
    ('sampling', str, 'uniform', False, 'perturbation law: uniform / poisson'),
    ('nmfk_split', str, 'data', False, 'how the ranks of a pyDNMFk job share the sweep: data (the reference: blocks of X on a p_r x p_c grid) or '
                                       'perturbations (every rank holds the whole X and fits its share of the perturbations as one-rank problems: '
                                       '--p_r=1 --p_c=1 with any number of ranks; no exchange inside a fit)'),
    ('nmfk_batch', int, 0, False, 'perturbation fits of one rank that run together in one batched library call (0 = as many as fit in memory; 1 = one by one)'),
]


def build_parser():
    ap = argparse.ArgumentParser(description='pyDNMF / pyDNMFk on MI355X (reference-compatible flags)')
    for name, kind, default, required, text in FLAGS:
        ap.add_argument('--' + name, type=kind, default=default, required=required, help=text)
    return ap


def main():
    args = build_parser().parse_args()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")
    from pydnmfk_amd.data_io import data_read
    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.pyDNMF import PyNMF
    from pydnmfk_amd.pyDNMFk import PyNMFk
    args.nmfk_batch = True if args.nmfk_batch == 0 else args.nmfk_batch
    if args.nmfk_split not in ('data', 'perturbations'):
        raise SystemExit("--nmfk_split must be data or perturbations")
    shared = args.process == 'pyDNMFk' and args.nmfk_split == 'perturbations' and world > 1
    if shared:
        # every rank reads the WHOLE matrix (a 1 x 1 grid of its own); the job's ranks share the perturbations (PyNMFk)
        if args.p_r * args.p_c != 1:
            raise SystemExit("--nmfk_split=perturbations fits one-rank problems: --p_r=1 --p_c=1 (the %d ranks share the perturbations)" % world)
        from pydnmfk_amd.dist_comm import COMM_WORLD, SoloGrid
        whole = COMM_WORLD()
        comms = SoloGrid(whole.world_rank)
        args.size, args.rank, args.comm1, args.comm = 1, 0, comms.comm, comms
    else:
        comms = MPI_comm(None, args.p_r, args.p_c)
        args.size, args.rank, args.comm1, args.comm = comms.size, comms.rank, comms.comm, comms
    args.row_comm, args.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
    A_ij = data_read(args).read()
    if shared:
        args.size, args.rank, args.comm1 = whole.size, whole.rank, whole      # PyNMFk builds the fits' one-rank bag from here
    if args.rank == 0:
        print('Starting ', args.process, '...')
    if args.rng == 'device':
        # the rank's block goes to the GPU ONCE (bf16 storage is rounded here); PyNMF / PyNMFk then work on device tensors:
        # perturbations and the rand init are drawn on the GPU, factors stay there between the fits of an NMFk sweep
        from pydnmfk_amd.pyDNMF import storage_dtype
        A_ij = torch.from_numpy(np.ascontiguousarray(A_ij)).to(device=torch.device("cuda", torch.cuda.current_device()),
                                                               dtype=storage_dtype(A_ij, args))
    elif args.rng != 'numpy':
        raise SystemExit("--rng must be device or numpy")
    if args.exchange not in ('auto', 'torch', 'native'):
        raise SystemExit("--exchange must be auto, torch or native")
    if args.exchange == 'auto':                            # measured: tools/rankbench.py --method hals
        args.exchange = 'native' if (args.method.lower() == 'hals' and args.p_r > 1) else 'torch'
    if args.method.lower() == 'hals' and args.p_r > 1 and args.rank == 0:
        # the W sweep's column norms are global when W's rows are spread over ranks (dist_nmf.py:889, utils.py:388-391): k
        # dependent 8-byte allreduces per iteration -- said once, with the grid that avoids them
        print("note: method=hals on a %d x %d grid exchanges one column norm per factor column: k dependent 8-byte allreduces "
              "per iteration (k = %s%s).  On a 1 x %d grid W is replicated and its column norms are local -- prefer "
              "--p_r=1 --p_c=%d for HALS / NMFk sweeps when the matrix shape allows it."
              % (args.p_r, args.p_c, getattr(args, 'k', None) if args.process == 'pyDNMF' else "%s..%s" % (args.start_k, args.end_k),
                 ", sequenced inside the library: --exchange=%s" % args.exchange, world, world), flush=True)
    if args.exchange == 'torch' or world == 1:
        del args.exchange                                  # (the choreography reads params.exchange only when it is set)
    if args.process == 'pyDNMF':
        args.results_paths = args.results_path
        nmf = PyNMF(A_ij, factors=None, save_factors=True, params=args)
        W, H, err = nmf.fit()
        if args.rank == 0:
            print('relative error =', float(err))
    elif args.process == 'pyDNMFk':
        nopt = PyNMFk(A_ij, factors=None, params=args).fit()
        if args.rank == 0:
            print('Estimated k with NMFk is ', nopt)
    else:
        raise SystemExit("--process must be pyDNMF or pyDNMFk")
    if world > 1:
        args.comm1.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
