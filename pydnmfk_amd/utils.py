"""Partition math, the `args` attribute bag and dimension bookkeeping.

Mirrors the hot-path parts of reference pyDNMFk/utils.py: `determine_block_params` (:15-46),
`data_operations.compute_global_dim/compute_local_dim` (:73-115), `var_init` (:473-477),
`parse` (:480-483), zero row / column pruning (:117-217) and the NMFk `Checkpoint` (:486-536).
"""
import pickle

import numpy as np


class parse:
    """Empty attribute bag populated by assignment (utils.py:480-483)."""

    def __init__(self):
        pass


def var_init(clas, var, default):
    """Return clas.<var>, first setting it to `default` if absent (utils.py:473-477)."""
    if not hasattr(clas, var):
        setattr(clas, var, default)
    return getattr(clas, var)


class determine_block_params:
    """Block index ranges of a `shape` array on a `pgrid` grid for one rank (utils.py:15-46)."""

    def __init__(self, comm, pgrid, shape):
        self.rank = comm if isinstance(comm, (int, np.integer)) else comm.rank
        self.pgrid = tuple(int(p) for p in pgrid)
        self.rank = self.rank if int(np.prod(self.pgrid)) > 1 else 0
        self.shape = tuple(int(s) for s in shape)

    def determine_block_index_range_asymm(self):
        """(start_inds, end_inds), end inclusive; the first n % p blocks hold one extra item (utils.py:36-41)."""
        chunk_ind = np.unravel_index(self.rank, self.pgrid)
        start = [int(i) * (n // k) + min(int(i), n % k) for n, k, i in zip(self.shape, self.pgrid, chunk_ind)]
        end = [(int(i) + 1) * (n // k) + min(int(i) + 1, n % k) - 1 for n, k, i in zip(self.shape, self.pgrid, chunk_ind)]
        return start, end

    def determine_block_shape_asymm(self):
        s, e = self.determine_block_index_range_asymm()
        return [j - i + 1 for i, j in zip(s, e)]


class data_operations:
    """Global/local dimension bookkeeping for a rank's block (utils.py:49-115).  Writes
    params.m, params.n, params.m_loc, params.n_loc, params.W_start/W_end/H_start/H_end."""

    def __init__(self, data, params):
        self.ten = data
        self.params = params
        self.comm1 = params.comm1
        self.cart_1d_row = params.row_comm
        self.cart_1d_column = params.col_comm
        self.rank = self.comm1.rank
        self.p_r, self.p_c = params.p_r, params.p_c
        self.topo = params.topo
        self.k = params.k
        self.compute_global_dim()
        self.compute_local_dim()
        self.m, self.n = params.m, params.n

    def compute_global_dim(self):
        """utils.py:73-93: global m, n from the local block shapes via integer allreduces."""
        loc_m, loc_n = int(self.ten.shape[0]), int(self.ten.shape[1])
        p = self.params
        if self.p_r != 1 and self.p_c == 1:
            p.n = loc_n
            p.m = self.comm1.allreduce(loc_m)
        elif self.p_c != 1 and self.p_r == 1:
            p.n = self.comm1.allreduce(loc_n)
            p.m = loc_m
        else:
            p.m = self.comm1.allreduce(loc_m if self.rank % self.p_c == 0 else 0)
            p.n = self.comm1.allreduce(loc_n if self.rank // self.p_c == 0 else 0)

    def compute_local_dim(self):
        """utils.py:97-115: factor slice sizes and offsets."""
        p = self.params
        if self.topo == "2d":
            bm = determine_block_params(self.cart_1d_column, (self.p_c, 1), (self.ten.shape[0], self.k))
            bn = determine_block_params(self.cart_1d_row, (1, self.p_r), (self.k, self.ten.shape[1]))
        else:
            bm = determine_block_params(self.comm1, (self.p_r, 1), (p.m, self.k))
            bn = determine_block_params(self.comm1, (1, self.p_c), (self.k, p.n))
        m_loc = bm.determine_block_shape_asymm()[0]
        n_loc = bn.determine_block_shape_asymm()[1]
        w = bm.determine_block_index_range_asymm()
        h = bn.determine_block_index_range_asymm()
        p.m_loc, p.n_loc = m_loc, n_loc
        p.W_start, p.W_end = w[0][0], w[1][0] + 1
        p.H_start, p.H_end = h[0][1], h[1][1] + 1

    # ---- zero row / column pruning (utils.py:117-217) on torch tensors
    def zero_idx_prune(self):
        """Boolean keep-masks (rows of A, cols of A, rows of W, cols of H).  Non-zero counts are summed inside the
        sub-groups on a 2D grid (utils.py:121-123) and over the world along the split axis on a 1D grid (:124-126)."""
        import torch
        nz = self.ten != 0
        row_sum, col_sum = nz.sum(1), nz.sum(0)
        if self.topo == '2d':
            row_sum = self.cart_1d_column.allreduce(row_sum)
            col_sum = self.cart_1d_row.allreduce(col_sum)
        else:
            if self.p_c > 1:
                row_sum = self.comm1.allreduce(row_sum)
            if self.p_r > 1:
                col_sum = self.comm1.allreduce(col_sum)
        row_x, col_x = row_sum > 0, col_sum > 0
        if self.topo == '2d':
            col_h = col_x[self.params.H_start:self.params.H_end]
            row_w = row_x[self.params.W_start:self.params.W_end]
        else:
            row_w, col_h = row_x, col_x
        return row_x, col_x, row_w, col_h

    def prune_all(self, W, H):
        """utils.py:156-172: drop all-zero rows / columns of the block and the matching factor rows / columns."""
        p = self.params
        p.row_zero_idx_x, p.col_zero_idx_x, p.row_zero_idx_w, p.col_zero_idx_h = self.zero_idx_prune()
        if bool(p.row_zero_idx_x.all()) and bool(p.col_zero_idx_x.all()):
            return self.ten, W, H                      # nothing to prune: keep the caller's buffers (no copy)
        self.ten = self.ten[p.row_zero_idx_x][:, p.col_zero_idx_x].contiguous()
        return self.ten, W[p.row_zero_idx_w].contiguous(), H[:, p.col_zero_idx_h].contiguous()

    def unprune_factors(self, W, H):
        """utils.py:176-217: scatter the factors back (zeros at pruned rows / columns).  Stays float32 on the device
        (the reference returns float64 here because np.zeros defaults to it, utils.py:195,198)."""
        import torch
        p = self.params
        if W.shape[0] != p.row_zero_idx_w.numel():
            Wf = torch.zeros(p.row_zero_idx_w.numel(), W.shape[1], dtype=W.dtype, device=W.device)
            Wf[p.row_zero_idx_w] = W
            W = Wf
        if H.shape[1] != p.col_zero_idx_h.numel():
            Hf = torch.zeros(H.shape[0], p.col_zero_idx_h.numel(), dtype=H.dtype, device=H.device)
            Hf[:, p.col_zero_idx_h] = H
            H = Hf
        return W, H


_REF_PARSE = b"cpyDNMFk.utils\nparse\n"          # how a protocol-2 pickle names the reference's bag class
_OUR_PARSE = b"cpydnmfk_amd.utils\nparse\n"


class _BagUnpickler(pickle.Unpickler):
    """Restricted unpickler for checkpoint.p: the only class a checkpoint may name is the attribute bag -- the reference's
    `pyDNMFk.utils.parse` (a checkpoint written by lanl/pyDNMFk) or this package's -- and both load as `parse` here."""

    def find_class(self, module, name):
        if name == "parse" and module in ("pyDNMFk.utils", "pydnmfk_amd.utils"):
            return parse
        raise pickle.UnpicklingError("checkpoint.p may only contain a parse bag, found %s.%s" % (module, name))


class Checkpoint:
    """Coarse NMFk resume state, file-compatible with reference utils.py:486-536 in BOTH directions: rank 0 pickles an
    attribute bag (flag, perturbation, k) to results_path + "checkpoint.p".  The file names the bag class the way the
    reference does (`pyDNMFk.utils.parse`, protocol 2), so lanl/pyDNMFk can resume from a checkpoint written here, and
    the loader maps that name (and this package's own) to `parse`, so a checkpoint written by lanl/pyDNMFk resumes here
    (tests/golden/ref_checkpoint.p is one).  Loading copies the bag's attributes onto this object.  W/H are not part of
    a checkpoint (the interrupted k restarts from its first perturbation)."""

    def __init__(self, checkpoint_save, params):
        self.checkpoint_save = checkpoint_save if checkpoint_save else False
        self.params = params
        self.perturbation = 0
        self.k = 0
        self.flag = 0

    def load_from_checkpoint(self):
        if self.checkpoint_save:
            with open(self.params.results_path + "/checkpoint.p", "rb") as f:          # utils.py:514
                saved = _BagUnpickler(f).load()
            if getattr(self.params, "rank", 0) == 0:
                print("Checkpoint loaded")
            self._set_params(vars(saved))
            if getattr(self.params, "rank", 0) == 0:
                print("Continuing from checkpoint for k=", self.k, "perturbation=", self.perturbation)

    @staticmethod
    def dumps(flag, perturbation, k):
        """The checkpoint bytes: a protocol-2 pickle of the bag, its class named as the reference names it."""
        bag = parse()
        bag.flag, bag.perturbation, bag.k = flag, perturbation, k
        raw = pickle.dumps(bag, protocol=2)
        assert raw.count(_OUR_PARSE) == 1
        return raw.replace(_OUR_PARSE, _REF_PARSE)

    def _save_checkpoint(self, flag, perturbation, k):
        if self.checkpoint_save and getattr(self.params, "rank", 0) == 0:
            with open(self.params.results_path + "checkpoint.p", "wb") as f:           # utils.py:530
                f.write(self.dumps(flag, perturbation, k))

    def _set_params(self, class_parameters):
        for name, value in class_parameters.items():
            setattr(self, name, value)
