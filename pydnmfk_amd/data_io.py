"""On-disk factor layout, drop-in with reference pyDNMFk/data_io.py:143-196 (`data_write.save_factors`)."""
import os

import numpy as np


class data_write:
    """Writes per-rank factor blocks as plain .npy under params.results_paths (data_io.py:158-196):
    p_c == 1: every rank W_factors/W_<rank>.npy, rank 0 H_factors/H.npy; p_r == 1: rank 0 W_factors/W.npy,
    every rank H_factors/H_<rank>.npy; 2D (and 1x1): every rank both.  reg=True -> W_reg_factors/H_reg_factors."""

    def __init__(self, args):
        self.p_r, self.p_c = args.p_r, args.p_c
        self.pgrid = [self.p_r, self.p_c]
        self.ftype = getattr(args, "ftype", None)
        self.comm = args.comm1
        self.params = args
        self.fpath = self.params.results_paths
        self.rank = self.comm.rank

    @staticmethod
    def create_folder_dir(fpath):
        try:
            os.mkdir(fpath)
        except OSError:
            pass

    def save_factors(self, factors, reg=False):
        self.create_folder_dir(self.fpath)
        sub = ('W_reg_factors/', 'H_reg_factors/') if reg else ('W_factors/', 'H_factors/')
        W_pth, H_pth = self.fpath + sub[0], self.fpath + sub[1]
        self.create_folder_dir(W_pth)
        self.create_folder_dir(H_pth)
        W, H = np.asarray(factors[0]), np.asarray(factors[1])
        if self.p_r == 1 and self.p_c != 1:
            if self.rank == 0:
                np.save(W_pth + 'W.npy', W)
            np.save(H_pth + 'H_' + str(self.rank) + '.npy', H)
        elif self.p_c == 1 and self.p_r != 1:
            if self.rank == 0:
                np.save(H_pth + 'H.npy', H)
            np.save(W_pth + 'W_' + str(self.rank) + '.npy', W)
        else:
            np.save(H_pth + 'H_' + str(self.rank) + '.npy', H)
            np.save(W_pth + 'W_' + str(self.rank) + '.npy', W)

    def save_cluster_results(self, params):
        """Rank 0 writes the per-k NMFk statistics (data_io.py:199-209).  Dataset names are the reference's
        (clusterSilhouetteCoefficients, avgSilhouetteCoefficients, L_err, L_errDist, avgErr, ErrTol, AIC).  HDF5
        (`results.h5`) when h5py is importable -- byte-compatible with the reference's readers; otherwise the same
        keys go to `results.npz` (this image has no h5py)."""
        if self.rank != 0:
            return
        data = {'clusterSilhouetteCoefficients': np.asarray(params['clusterSilhouetteCoefficients']),
                'avgSilhouetteCoefficients': np.asarray(params['avgSilhouetteCoefficients']),
                'L_err': np.asarray(params['L_err']), 'L_errDist': np.asarray(params['L_errDist']),
                'avgErr': np.asarray(params['avgErr']), 'ErrTol': np.asarray(params['recon_err']),
                'AIC': np.asarray(params['AIC'])}
        try:
            import h5py
        except ImportError:
            np.savez(self.fpath + 'results.npz', **data)
            return
        with h5py.File(self.fpath + 'results.h5', 'w') as hf:
            for key, val in data.items():
                hf.create_dataset(key, data=val)


def read_cluster_results(path):
    """Per-k statistics written by save_cluster_results (either container)."""
    if os.path.exists(path + 'results.h5'):
        import h5py
        with h5py.File(path + 'results.h5', 'r') as hf:
            return {key: np.array(hf[key]) for key in hf.keys()}
    z = np.load(path + 'results.npz')
    return {key: z[key] for key in z.files}
