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


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    if v.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    raise NameError('Boolean value expected.')


def parser_pyNMF(parser):
    parser.add_argument('--process', type=str, default='pyDNMF', help='pyDNMF/pyDNMFk')
    parser.add_argument('--p_r', type=int, required=True, help='Now of row processors')
    parser.add_argument('--p_c', type=int, required=True, help='Now of column processors')
    parser.add_argument('--k', type=int, default=4, help='feature count')
    parser.add_argument('--fpath', type=str, default='data/', help='data path to read(eg: tmp/)')
    parser.add_argument('--ftype', type=str, default='mat', help='data type : mat/folder/h5')
    parser.add_argument('--fname', type=str, default='swim', help='File name')
    parser.add_argument('--init', type=str, default='rand', help='NMF initializations: rand/nnsvd')
    parser.add_argument('--itr', type=int, default=5000, help='NMF iterations, default:1000')
    parser.add_argument('--norm', type=str, default='kl', help='Reconstruction Norm for NMF to optimize:KL/FRO')
    parser.add_argument('--method', type=str, default='mu', help='NMF update method:MU/BCD/HALS')
    parser.add_argument('--verbose', type=str2bool, default=False)
    parser.add_argument('--results_path', type=str, default='results/', help='Path for saving results')
    parser.add_argument('--checkpoint', type=str2bool, default=False, help='Enable checkpoint to track the pyNMFk state')
    parser.add_argument('--timing_stats', type=str2bool, default=False, help='accepted for compatibility; ignored')
    parser.add_argument('--prune', type=str2bool, default=False, help='Prune zero row/column.')
    parser.add_argument('--precision', type=str, default='float32', help='Storage precision of the data: float32, or bfloat16 (Frobenius mu/hals; arithmetic stays float32).')
    return parser


def parser_pyNMFk(parser):
    parser.add_argument('--perturbations', type=int, default=20, help='perturbation for NMFk')
    parser.add_argument('--noise_var', type=float, default=0.015, help='Noise variance for NMFk')
    parser.add_argument('--start_k', type=int, default=1, help='Start index of K for NMFk')
    parser.add_argument('--end_k', type=int, default=10, help='End index of K for NMFk')
    parser.add_argument('--step_k', type=int, default=1, help='step for K search')
    parser.add_argument('--sill_thr', type=float, default=0.6, help='SIll Threshold for K estimation')
    parser.add_argument('--sampling', type=str, default='uniform', help='Sampling noise for NMFk i.e uniform/poisson')
    return parser


def main():
    parser = parser_pyNMFk(parser_pyNMF(argparse.ArgumentParser(description='Arguments for pyDNMF/pyDNMFk on MI355X')))
    args = parser.parse_args()
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
    comms = MPI_comm(None, args.p_r, args.p_c)
    args.size, args.rank, args.comm1, args.comm = comms.size, comms.rank, comms.comm, comms
    args.row_comm, args.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
    if args.rank == 0:
        print('Starting ', args.process, '...')
    A_ij = data_read(args).read()
    if args.process == 'pyDNMF':
        args.results_paths = args.results_path
        nmf = PyNMF(A_ij, factors=None, save_factors=True, params=args)
        W, H, err = nmf.fit()
        if args.rank == 0:
            print('relative error =', err)
    elif args.process == 'pyDNMFk':
        nopt = PyNMFk(A_ij, factors=None, params=args).fit()
        if args.rank == 0:
            print('Estimated k with NMFk is ', nopt)
    else:
        raise SystemExit("--process must be pyDNMF or pyDNMFk")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
