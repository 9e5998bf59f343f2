"""On-disk factor layout, drop-in with reference pyDNMFk/data_io.py:143-196 (`data_write.save_factors`)."""
import glob
import os

import numpy as np

from .utils import determine_block_params


class data_read:
    """Read a dense matrix and return this rank's block -- reference data_io.py:12-105.

    args.fpath + args.fname + '.' + args.ftype with ftype in {npy, csv, txt, mat}; ftype == 'folder' reads the
    pre-split block fpath + fname + <rank> + '.npy' as is.  The block is rows/cols
    determine_block_params(rank, [p_r, p_c], shape) (data_io.py:81-83), cast to args.precision (default float32).
    Like the reference every rank opens the file; .npy files are memory-mapped here so only the block is paged in
    (the reference loads the whole matrix on every rank, data_io.py:57)."""

    def __init__(self, args):
        self.fpath = args.fpath
        if "grid" in vars(args) and args.grid:
            self.pgrid = args.grid
        else:
            self.pgrid = [args.p_r, args.p_c]
        self.ftype = args.ftype
        self.fname = args.fname
        self.comm = args.comm1
        self.rank = self.comm.rank
        self.precision = getattr(args, "precision", None) or 'float32'
        self.data = 0
        if self.ftype == 'folder':
            self.file_path = self.fpath + self.fname + str(self.comm.rank) + '.npy'
        else:
            self.file_path = self.fpath + self.fname + '.' + self.ftype

    def read(self):
        return self.read_dat()

    def read_file_npy(self):
        self.data = np.load(self.file_path, mmap_mode='r')

    def read_file_csv(self):
        self.data = np.loadtxt(self.file_path, delimiter=',', ndmin=2)

    def read_file_mat(self):
        from scipy.io import loadmat
        self.data = loadmat(self.file_path)['X']

    def data_partition(self):
        blk = determine_block_params(self.rank, self.pgrid, self.data.shape)
        s, e = blk.determine_block_index_range_asymm()
        self.data = self.data[s[0]:e[0] + 1, s[1]:e[1] + 1]

    def read_dat(self):
        if self.ftype == 'npy':
            self.read_file_npy()
            self.data_partition()
        elif self.ftype in ('csv', 'txt'):
            self.read_file_csv()
            self.data_partition()
        elif self.ftype == 'mat':
            self.read_file_mat()
            self.data_partition()
        elif self.ftype == 'folder':
            self.read_file_npy()
        else:
            raise ValueError("unknown ftype '%s' (npy/csv/txt/mat/folder)" % self.ftype)
        prec = 'float32' if str(self.precision).lower() in ('bfloat16', 'bf16') else self.precision
        return np.ascontiguousarray(self.data).astype(prec)   # bf16: numpy has no such dtype; PyNMF rounds on upload


class read_factors:
    """Re-assemble saved regression factors -- reference data_io.py:212-261.  W blocks are stacked by rank along
    rows.  H blocks: on a 2D grid rank r = i * p_c + j holds the i-th column slice of column block j, so the global
    column order is (j, i) -> rank i * p_c + j.  (The reference's transform_H_index uses i * p_r + j, utils.py:357,
    which is only right for square grids.)"""

    def __init__(self, factors_path, pgrid):
        self.factors_path = factors_path
        self.W_path = self.factors_path + 'W_reg_factors/*'
        self.H_path = self.factors_path + 'H_reg_factors/*'
        self.p_grid = pgrid

    @staticmethod
    def _by_rank(files):
        def rank_of(f):
            stem = os.path.splitext(os.path.basename(f))[0]
            return int(stem.split('_')[1]) if '_' in stem else 0
        return sorted(files, key=rank_of)

    def load_factors(self):
        wf, hf = self._by_rank(glob.glob(self.W_path)), self._by_rank(glob.glob(self.H_path))
        W = [np.load(f) for f in wf]
        H = [np.load(f) for f in hf]
        W_data = np.vstack(W) if len(W) > 1 else W[0]
        if len(H) > 1:
            if len(W) > 1:
                p_r, p_c = self.p_grid
                order = [i * p_c + j for j in range(p_c) for i in range(p_r)]
                H_data = np.hstack([H[r] for r in order])
            else:
                H_data = np.hstack(H)
        else:
            H_data = H[0]
        return W_data, H_data


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
