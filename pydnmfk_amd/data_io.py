"""On-disk factor layout, drop-in with reference pyDNMFk/data_io.py:143-196 (`data_write.save_factors`)."""
import glob
import os

import numpy as np

from .utils import determine_block_params


def _load_npy(path):
    return np.load(path, mmap_mode='r')                      # only the rank's block is paged in


def _load_text(path):
    return np.loadtxt(path, delimiter=',', ndmin=2)


def _load_mat(path):
    from scipy.io import loadmat
    return loadmat(path)['X']


# ftype -> (reader, the file holds the WHOLE matrix: cut this rank's block out of it)
_READERS = {'npy': (_load_npy, True), 'csv': (_load_text, True), 'txt': (_load_text, True), 'mat': (_load_mat, True),
            'folder': (_load_npy, False)}


class data_read:
    """This rank's block of a dense matrix on disk -- what reference data_io.py:12-105 returns.

    File: args.fpath + args.fname + '.' + args.ftype for ftype in {npy, csv, txt, mat}; ftype == 'folder' names the pre-split
    block fpath + fname + <rank> + '.npy', taken as is.  Whole-matrix files are cut by determine_block_params(rank, [p_r, p_c],
    shape) (data_io.py:81-83); the block is cast to args.precision (default float32).  Like the reference every rank opens the
    file; .npy files are memory-mapped here (the reference loads the whole matrix on every rank, data_io.py:57)."""

    def __init__(self, args):
        opts = vars(args)
        self.ftype = args.ftype
        if self.ftype not in _READERS:
            raise ValueError("unknown ftype '%s' (npy/csv/txt/mat/folder)" % self.ftype)
        self.pgrid = opts.get("grid") or [args.p_r, args.p_c]
        self.rank = args.comm1.rank
        self.precision = opts.get("precision") or 'float32'
        whole = _READERS[self.ftype][1]
        self.file_path = args.fpath + args.fname + ('.' + self.ftype if whole else '%d.npy' % self.rank)

    def read(self):
        load, whole = _READERS[self.ftype]
        data = load(self.file_path)
        if whole:
            s, e = determine_block_params(self.rank, self.pgrid, data.shape).determine_block_index_range_asymm()
            data = data[s[0]:e[0] + 1, s[1]:e[1] + 1]
        prec = 'float32' if str(self.precision).lower() in ('bfloat16', 'bf16') else self.precision
        return np.ascontiguousarray(data).astype(prec)       # bf16: numpy has no such dtype; PyNMF rounds on upload

    read_dat = read                                          # (the reference's name for the same call)


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
    """Per-rank factor blocks as plain .npy under params.results_paths -- the layout of reference data_io.py:158-196, which
    read_factors, the CLI and downstream scripts rely on: a factor that is REPLICATED over the ranks (H on a p_r x 1 grid, W on a
    1 x p_c grid) is written once, by rank 0, without a rank suffix; every other block as <name>_<rank>.npy by its owner.
    reg=True -> W_reg_factors / H_reg_factors (the regression fits of an NMFk sweep)."""

    def __init__(self, args):
        self.params = args
        self.p_r, self.p_c = args.p_r, args.p_c
        self.pgrid = [self.p_r, self.p_c]
        self.ftype = getattr(args, "ftype", None)
        self.rank = args.comm1.rank
        self.fpath = args.results_paths

    @staticmethod
    def create_folder_dir(fpath):
        os.makedirs(fpath, exist_ok=True)

    def save_factors(self, factors, reg=False):
        replicated = {'W': self.p_r == 1 and self.p_c != 1, 'H': self.p_c == 1 and self.p_r != 1}
        for name, block in zip(('W', 'H'), factors):
            folder = self.fpath + name + ('_reg_factors/' if reg else '_factors/')
            self.create_folder_dir(folder)
            if not replicated[name]:
                np.save(folder + '%s_%d.npy' % (name, self.rank), np.asarray(block))
            elif self.rank == 0:
                np.save(folder + name + '.npy', np.asarray(block))

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
