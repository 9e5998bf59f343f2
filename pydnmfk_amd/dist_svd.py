"""Distributed truncated SVD and NNDSVD initialisation for 1D grids -- behavioural mirror of reference
pyDNMFk/dist_svd.py (`DistSVD.svd` :149-186, `.nnsvd` :200-267).  A one-off pre-step outside the MU hot path
(SURVEY.md 8f row 4); it runs on torch tensors wherever the block lives.

Method: the reference finds the leading singular triplets one by one by power iteration on the (allreduced) n x n or
m x m Gram matrix with deflation.  Here the same Gram matrix is formed once in float64 and its leading eigenpairs are
taken with `torch.linalg.eigh` (identical subspace, no iteration count / random start vector), then the sharded factor
is recovered as A V / s (or A^T U / s).  NNDSVD (Boutsidis & Gallopoulos, the reference's flag=1 branch): per component
keep the dominant signed part of (u, v), scaled by sqrt(s * |u+-| * |v+-|), with the norms of the sharded factor taken
globally.  Finally columns of W are scaled to unit sum and H inversely (dist_svd.py:66-76), as the reference does.
"""
import numpy as np
import torch


class DistSVD:
    def __init__(self, args, A):
        self.args = args
        self.globalm, self.globaln = args.m, args.n
        self.k = args.k if args.k else min(self.globalm, self.globaln)
        self.comm1 = args.comm1
        self.rank = self.comm1.rank
        self.proc_rows, self.proc_cols = args.p_r, args.p_c
        if self.proc_rows != 1 and self.proc_cols != 1:
            raise Exception('NNSVD init only available for 1D topology, please try with 1d topo.')   # pyDNMF.py:135
        self.eps = float(args.eps)
        self.A = A if isinstance(A, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(A))

    def _gram64(self, rows_are_contraction):
        """sum over ranks of A^T A (n x n) or A A^T (m x m) in float64, formed in slabs (no float64 copy of A)."""
        A = self.A
        d = A.shape[1] if rows_are_contraction else A.shape[0]
        G = torch.zeros(d, d, dtype=torch.float64, device=A.device)
        step = max(1, (1 << 24) // max(1, A.shape[1] if rows_are_contraction else A.shape[0]))
        if rows_are_contraction:
            for r0 in range(0, A.shape[0], step):
                a = A[r0:r0 + step].double()
                G += a.t() @ a
        else:
            for c0 in range(0, A.shape[1], step):
                a = A[:, c0:c0 + step].double()
                G += a @ a.t()
        return self.comm1.allreduce_(G) if hasattr(self.comm1, "allreduce_") else self.comm1.allreduce(G)

    def svd(self):
        """(singular values [k], U [m_loc x k], V [k x n_loc]); the factor along the sharded axis is local, the other
        one replicated (dist_svd.py:149-186)."""
        A = self.A
        row_sharded = self.proc_cols == 1          # A is m_loc x n: contraction over the (distributed) rows
        G = self._gram64(rows_are_contraction=row_sharded)
        lam, vec = torch.linalg.eigh(G)
        lam, vec = lam.flip(0)[: self.k], vec.flip(1)[:, : self.k]
        s = torch.sqrt(torch.clamp(lam, min=0.0))
        if row_sharded:
            V = vec                                 # n x k, replicated
            U = (A.double() @ V) / s                # m_loc x k
            return s, U, V.t().contiguous()
        U = vec                                     # m x k, replicated
        V = (A.double().t() @ U) / s                # n_loc x k
        return s, U, V.t().contiguous()

    def rel_error(self, U, S, V):
        X = (U @ S @ V).to(self.A.dtype)
        num = self.comm1.allreduce(float(((self.A - X).double() ** 2).sum()))
        den = self.comm1.allreduce(float((self.A.double() ** 2).sum()))
        return float(np.sqrt(num) / np.sqrt(den))

    def normalize_by_W(self, W, H):
        """dist_svd.py:66-76: unit column sums of W (global when W is row-sharded), H scaled inversely."""
        s = W.sum(dim=0, keepdim=True)
        if self.proc_rows != 1:
            s = self.comm1.allreduce(s)
        s = s + self.eps
        return W / s, H * s.t()

    def nnsvd(self, flag=1, verbose=0):
        """dist_svd.py:200-267.  Returns (W_i, H_j) [, {'recon_err_svd', 'recon_err_nnsvd'} when verbose]."""
        s, U, V = self.svd()                        # V: k x n_loc
        err = {}
        if verbose:
            err['recon_err_svd'] = self.rel_error(U, torch.diag(s), V)
        if flag == 0:
            W, H = U.clone(), torch.diag(s) @ V
            W[W < 0] = 0
            H[H < 0] = 0
        else:
            Vt = V.t()                              # n_loc x k
            UP, UN = torch.clamp(U, min=0), torch.clamp(-U, min=0)
            VP, VN = torch.clamp(Vt, min=0), torch.clamp(-Vt, min=0)

            def gnorm(X, sharded):
                q = (X * X).sum(dim=0)
                return torch.sqrt(self.comm1.allreduce(q) if sharded else q)

            u_sharded = self.proc_cols == 1 and self.proc_rows != 1
            v_sharded = self.proc_rows == 1 and self.proc_cols != 1
            UPn, UNn = gnorm(UP, u_sharded), gnorm(UN, u_sharded)
            VPn, VNn = gnorm(VP, v_sharded), gnorm(VN, v_sharded)
            mp = torch.sqrt(UPn * VPn * s)
            mn = torch.sqrt(UNn * VNn * s)
            pos = mp > mn
            W = torch.where(pos, mp * UP / (UPn + self.eps), mn * UN / (UNn + self.eps))
            H = torch.where(pos, mp * VP / (VPn + self.eps), mn * VN / (VNn + self.eps)).t()
        if verbose:
            err['recon_err_nnsvd'] = self.rel_error(W, torch.eye(self.k, dtype=W.dtype, device=W.device), H)
        W, H = self.normalize_by_W(W, H)
        W, H = W.to(torch.float32).contiguous(), H.to(torch.float32).contiguous()
        return ((W, H), err) if verbose else (W, H)
