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
