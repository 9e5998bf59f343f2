"""CPU oracle: numpy restatement of pyDNMFk's distributed multiplicative-update path.

THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it, and
only as the checker / reported CPU baseline.  The product package
(`pydnmfk_amd`) never imports it and has no CPU fallback.

Parity status: PINNED.  Every function below is checked against golden vectors
captured from the unmodified reference run in the build container
(tests/golden/make_golden.py -> tests/golden/case_*.npz; tests/test_oracle_golden.py).
The reference's own tests hold no golden W/H for this path, only
convergence thresholds, which tests/test_oracle_golden.py also re-asserts.

All citations are file:line in /root/reference (lanl/pyDNMFk).  The P ranks of
the reference's p_r x p_c cartesian grid are simulated inside one process;
collectives become explicit rank-ordered sums / concatenations over Python lists.
"""
import numpy as np

__all__ = [
    "block_range", "data_block", "factor_ranges", "split_problem",
    "fro_mu_step_local", "kl_mu_step_local", "fro_hals_step_local", "SimGrid", "fit_single",
]


# ------------------------------------------------------------------ partition math
def block_range(i, nblk, n):
    """[start, end) of block i when n items are split into nblk blocks.

    pyDNMFk/utils.py:39-40 (determine_block_index_range_asymm): the first n % nblk
    blocks get one extra item.  (The reference returns an inclusive end.)
    """
    q, r = divmod(n, nblk)
    return i * q + min(i, r), (i + 1) * q + min(i + 1, r)


def grid_coords(rank, p_r, p_c):
    """rank = i * p_c + j (row-major; dist_comm.py:22 reorder=False, utils.py:38)."""
    return divmod(rank, p_c) if p_r * p_c > 1 else (0, 0)


def data_block(rank, p_r, p_c, m, n):
    """(r0, r1, c0, c1) of the block of A owned by `rank` (utils.py:36-41, data_io.py:81-83)."""
    i, j = grid_coords(rank, p_r, p_c)
    r0, r1 = block_range(i, p_r, m)
    c0, c1 = block_range(j, p_c, n)
    return r0, r1, c0, c1


def factor_ranges(rank, p_r, p_c, m, n):
    """((w0, w1), (h0, h1)): global rows of W and columns of H held by `rank`.

    1D (pyDNMF.py:115-129, utils.py:104-108): the sharded factor follows A's block,
    the other factor is replicated.  2D (utils.py:99-103): W_ij is the j-th of p_c
    row slices of A_ij's rows, H_ij the i-th of p_r column slices of A_ij's columns.
    """
    r0, r1, c0, c1 = data_block(rank, p_r, p_c, m, n)
    if p_r != 1 and p_c != 1:
        i, j = grid_coords(rank, p_r, p_c)
        ws, we = block_range(j, p_c, r1 - r0)
        hs, he = block_range(i, p_r, c1 - c0)
        return (r0 + ws, r0 + we), (c0 + hs, c0 + he)
    if p_c == 1:
        return (r0, r1), (0, n)
    return (0, m), (c0, c1)


def split_problem(A, W0, H0, p_r, p_c):
    """Per-rank (A_ij, W block, H block) copies for a p_r x p_c grid."""
    m, n = A.shape
    out = []
    for rank in range(p_r * p_c):
        r0, r1, c0, c1 = data_block(rank, p_r, p_c, m, n)
        (w0, w1), (h0, h1) = factor_ranges(rank, p_r, p_c, m, n)
        out.append((np.ascontiguousarray(A[r0:r1, c0:c1]),
                    np.array(W0[w0:w1], dtype=A.dtype), np.array(H0[:, h0:h1], dtype=A.dtype)))
    return out


def _rsum(vals):
    """Rank-ordered sum ((v0 + v1) + v2) + ... -- stands for MPI allreduce(SUM)."""
    acc = np.array(vals[0], copy=True)
    for v in vals[1:]:
        acc = acc + v
    return acc


# ------------------------------------------------------------------ single-rank steps
def fro_mu_step_local(A, W, H, eps, W_update=True):
    """One MU/Frobenius step on one rank, in place (dist_nmf.py:716-771 with p_r=p_c=1).

    W-update: HHT = H H^T (:729); AH = A H^T (:730); W *= AH / (W HHT + eps) (:731-732)
    H-update (uses the NEW W): WTW = W^T W (:748); AtW = W^T A (:749);
              H *= AtW / (H^T WTW + eps)^T (:750-751)
    """
    if W_update:
        HHT = np.matmul(H, H.T)
        AH = np.matmul(A, H.T)
        W *= AH / (np.matmul(W, HHT) + eps)
    WTW = np.matmul(W.T, W)
    AtW = np.matmul(W.T, A)
    H *= AtW / (np.matmul(H.T, WTW) + eps).T
    return W, H


def kl_mu_step_local(A, W, H, eps, W_update=True):
    """One MU/KL step on one rank, in place (dist_nmf.py:776-869 with p_r=p_c=1).

    W-update: x2 = rowsum(H) (:827); U = A / (W H + eps) (:806); W *= (U H^T) / (x2 + eps) (:828-830)
    H-update (NEW W): x = colsum(W) (:846); U recomputed (:806); H *= (W^T U) / (x + eps) (:847-849)
    """
    if W_update:
        x2 = H.sum(axis=1)
        U = A / (W @ H + eps)
        W *= (U @ H.T) / (x2[None, :] + eps)
    x = W.sum(axis=0)
    U = A / (W @ H + eps)
    H *= (W.T @ U) / (x[:, None] + eps)
    return W, H


def fro_hals_step_local(A, W, H, eps, W_update=True):
    """One HALS/Frobenius step on one rank, in place (dist_nmf.py:873-934 with p_r=p_c=1).

    W sweep (:884-891): for kk: W[:,kk] = max(W[:,kk]*HHT[kk,kk] + AH[:,kk] - W @ HHT[:,kk], eps); W[:,kk] /= ||W[:,kk]||_2
    H sweep (:905-909): for kk: H[kk,:] = max(H[kk,:] + AtW[kk,:] - WTW[kk,:] @ H, eps)   (updated rows are used at once)
    """
    k = W.shape[1]
    if W_update:
        HHT = np.matmul(H, H.T)
        AH = np.matmul(A, H.T)
        for kk in range(k):
            t = W[:, kk] * HHT[kk, kk] + AH[:, kk] - W.dot(HHT[:, kk])
            W[:, kk] = np.maximum(t, eps)
            ss = np.sqrt(np.linalg.norm(W[:, kk], ord=2) ** 2)               # utils.py:388-391
            if ss > 0:
                W[:, kk] /= ss
    WTW = np.matmul(W.T, W)
    AtW = np.matmul(W.T, A)
    for kk in range(k):
        t = H[kk, :] + AtW[kk, :] - WTW[kk, :].dot(H)
        H[kk, :] = np.maximum(t, eps)
    return W, H


# ------------------------------------------------------------------ simulated grid
class SimGrid:
    """The reference's SPMD program with P = p_r * p_c ranks simulated in one process.

    State: lists (indexed by rank) of A_ij, W block, H block, laid out exactly as
    `PyNMF` holds them (pyDNMF.py:83-129).  `update()` = one `nmf_algorithms_*.update()`
    on every rank; `fit(itr)` = `PyNMF.fit()` (pyDNMF.py:138-182).
    """

    def __init__(self, A, W0, H0, p_r=1, p_c=1, norm="fro", W_update=True, method="mu", prune=False, eps=None):
        self.method = method
        self.prune = prune
        self.p_r, self.p_c, self.P = p_r, p_c, p_r * p_c
        self.m, self.n = A.shape
        self.dtype = A.dtype
        # pyDNMF.py:68: the machine epsilon of the data's dtype.  `eps` overrides it for ONE use: a float64 evaluation of a float32
        # run (tests of fp32 kernels against the same loop without fp32 rounding keep the fp32 epsilon in the denominators)
        self.eps = np.finfo(A.dtype).eps if eps is None else eps
        self.norm = norm
        self.W_update = W_update
        self.topo = "2d" if (p_r != 1 and p_c != 1) else "1d"  # pyDNMF.py:83-87
        blocks = split_problem(A, W0, H0, p_r, p_c)
        self.A = [b[0] for b in blocks]
        self.W = [b[1] for b in blocks]
        self.H = [b[2] for b in blocks]
        if norm.upper() not in ("FRO", "KL"):
            raise Exception("Not a valid norm: Choose (fro/kl)")  # dist_nmf.py:91,659
        if prune:
            self._prune_all()

    # ---- zero row / column pruning (pyDNMF.py:99-101; utils.py:117-217), restated literally
    def _prune_all(self):
        R = range(self.P)
        row_sum = [np.sum(self.A[r] != 0, 1) for r in R]                 # utils.py:119-120
        col_sum = [np.sum(self.A[r] != 0, 0) for r in R]
        if self.topo == "2d":                                            # utils.py:121-123: sums inside the sub-groups
            row_sum = [_rsum([row_sum[q] for q in self._col_group(r)]) for r in R]
            col_sum = [_rsum([col_sum[q] for q in self._row_group(r)]) for r in R]
        else:                                                            # :124-126: world sums along the split axis
            if self.p_c > 1:
                row_sum = [_rsum(row_sum)] * self.P
            if self.p_r > 1:
                col_sum = [_rsum(col_sum)] * self.P
        self._wmask, self._hmask = [], []
        for r in R:
            rx, cx = row_sum[r] > 0, col_sum[r] > 0
            if self.topo == "2d":                                        # :129-131: the factor slices of the block
                (w0, w1), (h0, h1) = factor_ranges(r, self.p_r, self.p_c, self.m, self.n)
                r0, _, c0, _ = data_block(r, self.p_r, self.p_c, self.m, self.n)
                rw, ch = rx[w0 - r0: w1 - r0], cx[h0 - c0: h1 - c0]
            else:
                rw, ch = rx, cx
            self.A[r] = self.A[r][np.ix_(rx, cx)]                        # :151
            self.W[r] = self.W[r][rw]
            self.H[r] = self.H[r][:, ch]
            self._wmask.append(rw)
            self._hmask.append(ch)

    def _unprune(self):
        """utils.py:176-217: scatter the factors back into zero matrices (float64, as np.zeros defaults)."""
        W, H = [], []
        for r in range(self.P):
            Wf = np.zeros((len(self._wmask[r]), self.W[r].shape[1]))
            Wf[self._wmask[r], :] = self.W[r]
            Hf = np.zeros((self.H[r].shape[0], len(self._hmask[r])))
            Hf[:, self._hmask[r]] = self.H[r]
            W.append(Wf)
            H.append(Hf)
        return W, H

    # ---- communicator membership (dist_comm.py:25-51)
    def _row_group(self, rank):
        """cartesian1d_row = Sub([True, False]): same grid column j, size p_r, ordered by i."""
        _, j = grid_coords(rank, self.p_r, self.p_c)
        return [i * self.p_c + j for i in range(self.p_r)]

    def _col_group(self, rank):
        """cartesian1d_column = Sub([False, True]): same grid row i, size p_c, ordered by j."""
        i, _ = grid_coords(rank, self.p_r, self.p_c)
        return [i * self.p_c + j for j in range(self.p_c)]

    # ---- one update step
    def update(self):
        fro = self.norm.upper() == "FRO"
        if self.method.upper() == "HALS":
            if not fro:
                raise Exception("Not a valid method: Choose (mu)")     # dist_nmf.py:89,657
            return (self._hals_1d if self.topo == "1d" else self._hals_2d)()
        if self.method.upper() != "MU":
            raise Exception("Not a valid method: Choose (mu/hals/bcd)")
        if self.topo == "1d":
            (self._fro_1d if fro else self._kl_1d)()
        else:
            (self._fro_2d if fro else self._kl_2d)()

    # 1D: dist_nmf.py:663-771
    def _fro_1d(self):
        R = range(self.P)
        A, W, H, eps = self.A, self.W, self.H, self.eps
        if self.W_update:
            HHT = [np.matmul(H[r], H[r].T) for r in R]                   # :729 -> :679
            AH = [np.matmul(A[r], H[r].T) for r in R]                    # :730 -> :705
            if self.p_c != 1:                                            # allreduce iff p != 1 (:680,:706)
                HHT = [_rsum(HHT)] * self.P
                AH = [_rsum(AH)] * self.P
            for r in R:
                W[r] *= AH[r] / (np.matmul(W[r], HHT[r]) + eps)          # :731-732
        WTW = [np.matmul(W[r].T, W[r]) for r in R]                       # :748
        AtW = [np.matmul(W[r].T, A[r]) for r in R]                       # :749
        if self.p_r != 1:
            WTW = [_rsum(WTW)] * self.P
            AtW = [_rsum(AtW)] * self.P
        for r in R:
            H[r] *= AtW[r] / (np.matmul(H[r].T, WTW[r]) + eps).T         # :750-751

    # 1D KL: dist_nmf.py:776-869
    def _kl_1d(self):
        R = range(self.P)
        A, W, H, eps = self.A, self.W, self.H, self.eps
        if self.W_update:
            x2 = [H[r].sum(axis=1) for r in R]                           # :827 -> :793-795
            sk = [(A[r] / (W[r] @ H[r] + eps)) @ H[r].T for r in R]      # :806,:810
            if self.p_c != 1:
                x2 = [_rsum(x2)] * self.P
                sk = [_rsum(sk)] * self.P
            for r in R:
                W[r] *= sk[r] / (x2[r][None, :] + eps)                   # :828-830
        x1 = [W[r].sum(axis=0) for r in R]                               # :846
        sk = [W[r].T @ (A[r] / (W[r] @ H[r] + eps)) for r in R]          # :806,:808
        if self.p_r != 1:
            x1 = [_rsum(x1)] * self.P
            sk = [_rsum(sk)] * self.P
        for r in R:
            H[r] *= sk[r] / (x1[r][:, None] + eps)                       # :847-849

    # HALS sweeps shared by 1D and 2D (the per-rank loops of dist_nmf.py:884-891 / :905-909 and :428-434 / :449-452)
    def _hals_w_sweep(self, AH, HHT, allreduce_norm):
        """AH, HHT: per-rank lists.  Column kk is clamped, then divided by its GLOBAL 2-norm
        (utils.py:367-391 `norm`: local squared norms are allreduced iff p != 1) before column kk+1 is touched."""
        k = self.W[0].shape[1]
        for kk in range(k):
            sq = []
            for r in range(self.P):
                W = self.W[r]
                t = W[:, kk] * HHT[r][kk, kk] + AH[r][:, kk] - W.dot(HHT[r][:, kk])
                W[:, kk] = np.maximum(t, self.eps)
                sq.append(np.linalg.norm(W[:, kk], ord=2) ** 2)
            for r in range(self.P):
                ss = np.sqrt(_rsum(sq)) if allreduce_norm else np.sqrt(sq[r])
                if ss > 0:
                    self.W[r][:, kk] /= ss

    def _hals_h_sweep(self, AtW, WTW):
        for r in range(self.P):
            H = self.H[r]
            for kk in range(H.shape[0]):
                t = H[kk, :] + AtW[r][kk, :] - WTW[r][kk, :].dot(H)
                H[kk, :] = np.maximum(t, self.eps)

    # 1D HALS: dist_nmf.py:873-934
    def _hals_1d(self):
        R = range(self.P)
        A, W, H = self.A, self.W, self.H
        if self.W_update:
            HHT = [np.matmul(H[r], H[r].T) for r in R]                   # :882
            AH = [np.matmul(A[r], H[r].T) for r in R]                    # :883
            if self.p_c != 1:
                HHT = [_rsum(HHT)] * self.P
                AH = [_rsum(AH)] * self.P
            self._hals_w_sweep(AH, HHT, allreduce_norm=(self.p_r != 1))  # norm(..., p=self.p_r) :889
        WTW = [np.matmul(W[r].T, W[r]) for r in R]                       # :902
        AtW = [np.matmul(W[r].T, A[r]) for r in R]                       # :903
        if self.p_r != 1:
            WTW = [_rsum(WTW)] * self.P
            AtW = [_rsum(AtW)] * self.P
        self._hals_h_sweep(AtW, WTW)

    # 2D HALS: dist_nmf.py:411-470
    def _hals_2d(self):
        R = range(self.P)
        A, W, H = self.A, self.W, self.H
        if self.W_update:
            HHT = _rsum([np.matmul(H[r], H[r].T) for r in R])            # :426 (world)
            V = [np.matmul(A[r], self._gather_H(r).T) for r in R]        # AH_glob :427
            AH = []
            for r in R:
                grp = self._col_group(r)
                AH.append(self._scatter_rows(r, grp, _rsum([V[q] for q in grp]), [W[q].shape[0] for q in grp]))
            self._hals_w_sweep(AH, [HHT] * self.P, allreduce_norm=True)  # p = p_r != 1 on a 2D grid (:432)
        WTW = _rsum([np.matmul(W[r].T, W[r]) for r in R])                # :447
        Y = [np.matmul(self._gather_W(r).T, A[r]) for r in R]            # ATW_glob :448
        AtW = []
        for r in R:
            grp = self._row_group(r)
            ks = self._scatter_rows(r, grp, _rsum([Y[q].T.copy() for q in grp]), [H[q].shape[1] for q in grp])
            AtW.append(ks.T)
        self._hals_h_sweep(AtW, [WTW] * self.P)

    # 2D helpers
    def _gather_W(self, rank):
        return np.vstack([self.W[q] for q in self._col_group(rank)])     # :163-165 / :289-291

    def _gather_H(self, rank):
        return np.hstack([self.H[q] for q in self._row_group(rank)])     # :195-197 / :285-287

    def _scatter_rows(self, rank, group, full, counts):
        """This rank's row slice of a reduce-scattered (rows x k) buffer (MPI Reduce_scatter;
        each member receives as many rows as its own local factor slice holds)."""
        g = group.index(rank)
        off = sum(counts[:g])
        return full[off: off + counts[g]]

    # 2D FRO: dist_nmf.py:95-263
    def _fro_2d(self):
        R = range(self.P)
        A, W, H, eps = self.A, self.W, self.H, self.eps
        if self.W_update:
            HHT = _rsum([np.matmul(H[r], H[r].T) for r in R])            # :241 -> :113-114 (world)
            V = [np.matmul(A[r], self._gather_H(r).T) for r in R]        # :195-198
            for r in R:
                grp = self._col_group(r)
                AH = self._scatter_rows(r, grp, _rsum([V[q] for q in grp]),
                                        [W[q].shape[0] for q in grp])      # :202
                W[r] *= AH / (np.matmul(W[r], HHT) + eps)                # :244-245
        WTW = _rsum([np.matmul(W[r].T, W[r]) for r in R])                # :222
        Y = [np.matmul(self._gather_W(r).T, A[r]) for r in R]            # :163-166
        for r in R:
            grp = self._row_group(r)
            ks = self._scatter_rows(r, grp, _rsum([Y[q].T.copy() for q in grp]),
                                    [H[q].shape[1] for q in grp])          # :169
            AtW = ks.T                                                   # :171
            H[r] *= AtW / (np.matmul(H[r].T, WTW) + eps).T               # :224-225

    # 2D KL: dist_nmf.py:268-407
    def _kl_2d(self):
        R = range(self.P)
        A, W, H, eps = self.A, self.W, self.H, self.eps
        if self.W_update:
            x2 = _rsum([H[r].sum(axis=1) for r in R])                    # :365 -> :347-348
            UHT = []
            for r in R:
                Wi, Hj = self._gather_W(r), self._gather_H(r)            # :367
                U = A[r] / (Wi.dot(Hj) + eps)                            # :337
                UHT.append(U.dot(Hj.T))                                  # :338
            for r in R:
                grp = self._col_group(r)
                sk = self._scatter_rows(r, grp, _rsum([UHT[q] for q in grp]),
                                        [W[q].shape[0] for q in grp])      # :340
                W[r] *= sk / (x2[None, :] + eps)                         # :366,:369
        x1 = _rsum([W[r].sum(axis=0) for r in R])                        # :385
        WTU = []
        for r in R:
            Wi, Hj = self._gather_W(r), self._gather_H(r)                # :387
            U = A[r] / (Wi.dot(Hj) + eps)                                # :311
            WTU.append(Wi.T.dot(U))                                      # :312
        for r in R:
            grp = self._row_group(r)
            ks = self._scatter_rows(r, grp, _rsum([WTU[q].T.copy() for q in grp]),
                                    [H[q].shape[1] for q in grp]).T        # :314-316
            H[r] *= ks / (x1[:, None] + eps)                             # :386,:389

    # ---- PyNMF.fit (pyDNMF.py:138-182)
    def clamp(self):
        """pyDNMF.py:155-157 / :170-172."""
        for r in range(self.P):
            self.H[r] = np.maximum(self.H[r], self.eps)
            self.W[r] = np.maximum(self.W[r], self.eps)

    def normalize_features(self):
        """pyDNMF.py:185-194: column sums of W (allreduce iff 2D or p_r != 1); W /= s+eps; H *= s^T."""
        s = [self.W[r].sum(axis=0, keepdims=True) for r in range(self.P)]
        if self.topo == "2d" or self.p_r != 1:
            s = [_rsum(s)] * self.P
        for r in range(self.P):
            self.W[r] /= s[r] + self.eps
            self.H[r] *= s[r].T

    def relative_err(self):
        """pyDNMF.py:205-218: ||A - W H||_F / ||A||_F with per-rank norms squared, summed, sqrt'ed."""
        num, den = [], []
        for r in range(self.P):
            if self.topo == "2d":
                Wi, Hj = self._gather_W(r), self._gather_H(r)            # :197-202
            else:
                Wi, Hj = self.W[r], self.H[r]
            num.append(np.linalg.norm(self.A[r] - Wi @ Hj) ** 2)        # :207,:215-217
            den.append(np.linalg.norm(self.A[r]) ** 2)
        return np.sqrt(_rsum(num)) / np.sqrt(_rsum(den))                 # :218,:210

    def fit(self, itr):
        """Returns (list of W blocks, list of H blocks, relative error) like PyNMF.fit on each rank."""
        for i in range(itr):
            self.update()                                                # :154 / :169
            if i % 10 == 0:                                              # :155 / :170
                self.clamp()
            if i == itr - 1:                                             # :158 / :173
                self.normalize_features()
                err = self.relative_err()
                if self.prune:                                           # pyDNMF.py:180-181
                    W, H = self._unprune()
                    return W, H, float(err)
                return self.W, self.H, float(err)
        raise ValueError("itr must be >= 1")


def fit_single(A, W0, H0, itr, norm="fro", W_update=True, method="mu", eps=None):
    """Single-rank convenience wrapper: returns (W, H, err)."""
    g = SimGrid(A, W0, H0, 1, 1, norm=norm, W_update=W_update, method=method, eps=eps)
    W, H, err = g.fit(itr)
    return W[0], H[0], err
