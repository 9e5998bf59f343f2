"""Distributed truncated SVD and NNDSVD initialisation for 1D grids -- behavioural mirror of reference
pyDNMFk/dist_svd.py (`DistSVD.svd` :149-186, `.nnsvd` :200-267).  A one-off pre-step of the MU path.

Method: the reference finds the leading singular triplets one by one by power iteration on the (allreduced) Gram
operator with deflation.  Here the k leading triplets are found together by block subspace iteration, and every pass
over the data block is one of the hot path's own contractions: Y = A V is `dnmf_aht` (A H^T with H = V^T), Z = A^T Y is
`dnmf_wta` (W^T A with W = Y), the k x k Ritz problem comes from `dnmf_gram_wtw`; the exchanges are the path's allreduce.
Only k-sized dense algebra (the QR of the replicated n x k / m x k basis, a k x k symmetric eigenproblem) runs in
torch.linalg, in float64.  Nothing of the size of A is ever copied or widened.  NNDSVD (Boutsidis & Gallopoulos, the
reference's flag=1 branch): per component keep the dominant signed part of (u, v), scaled by sqrt(s |u+-| |v+-|), with the
norms of the sharded factor taken globally.  Finally columns of W are scaled to unit sum and H inversely
(dist_svd.py:66-76), as the reference does.
"""
import numpy as np
import torch


class DistSVD:
    def __init__(self, args, A, ops=None):
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
        if ops is None:
            if not self.A.is_cuda:
                raise TypeError("DistSVD: the data block is on %s; the contractions run on the GPU (no CPU fallback)" % self.A.device)
            from .engine import HIP_OPS
            ops = HIP_OPS
        self.ops = ops
        self.max_iter = int(getattr(args, "svd_itr", 60))

    def _allreduce(self, t):
        return self.comm1.allreduce_(t) if hasattr(self.comm1, "allreduce_") else self.comm1.allreduce(t)

    def _orth(self, Z):
        """Orthonormal basis of the columns of the replicated (d x k) block, float64 QR (d k^2 work)."""
        Q, _ = torch.linalg.qr(Z.double())
        return Q.to(torch.float32).contiguous()

    def svd(self):
        """(singular values [k], U [m_loc x k], V [k x n_loc]); the factor along the sharded axis is local, the other
        one replicated (dist_svd.py:149-186).  Block subspace iteration on A^T A (row-sharded A) or A A^T (column-
        sharded A); stops when every wanted Ritz value has settled to 1e-7 relative to ITSELF (floor: the fp32 operator's
        resolution, singular values below ~3e-4 s_1) or after `params.svd_itr` (60) passes."""
        ops, A, kk = self.ops, self.A, self.k
        dev = A.device
        row_sharded = self.proc_cols == 1           # A is m_loc x n: the replicated basis lives in the n-space
        d = A.shape[1] if row_sharded else A.shape[0]
        # a few extra basis vectors (dropped after the Rayleigh-Ritz step) speed the trailing components up when the
        # spectrum is flat around s_k; the contractions pad the rank to 32 / 64 / 128 anyway
        k = min(d, 128, kk + max(4, kk // 2))
        g = torch.Generator(device="cpu").manual_seed(1234)            # same start on every rank, no exchange needed
        B = self._orth(torch.rand(d, k, generator=g, dtype=torch.float64).to(dev))
        m_l, n_l = A.shape
        kp = 32 if k <= 32 else (64 if k <= 64 else 128)
        lam_prev = None
        for it in range(self.max_iter):
            if row_sharded:
                Y = ops.aht(A, B.t().contiguous(), ops.empty((m_l, k), A))               # Y = A B         (m_loc x k)
                Z = self._allreduce(ops.wta(A, Y, ops.empty((k, n_l), A)))                # Z = A^T Y       (k x n), summed
            else:
                Y = ops.wta(A, B, ops.empty((k, n_l), A)).t().contiguous()                # Y = A^T B       (n_loc x k)
                Z = self._allreduce(ops.aht(A, Y.t().contiguous(), ops.empty((m_l, k), A))).t()   # Z = A Y (m x k), as k x m
            # Ritz values of the current basis: B^T (A^T A) B = B^T Z^T
            T = (B.double().t() @ Z.double().t())
            lam = torch.linalg.eigvalsh(0.5 * (T + T.t()))
            B = self._orth(Z.t())
            if lam_prev is not None:
                # per COMPONENT: every wanted Ritz value settled relative to ITSELF (|d lam_i| <= 1e-7 lam_i), with an absolute
                # floor at the resolution of the fp32 operator (values below 1e-7 lam_max -- singular values below ~3e-4 s_1 --
                # are rounding noise of A^T A applied in float32 and cannot settle further).  A test against lam_max alone
                # accepted trailing vectors (s_k / s_1 < 1e-3) that had not converged.
                w, wp = lam[-kk:], lam_prev[-kk:]                       # (eigvalsh ascends: the last kk are the wanted ones)
                # (the Ritz values come from A^T A B applied in float32, whose noise is ~6e-8 lam_max ABSOLUTE for every
                # component -- a component with 1e-7 < lam_i / lam_max << 1 cannot settle to 1e-7 lam_i and the loop ran all its
                # passes; hence the absolute term 4 eps lam_max in the tolerance)
                lam_max = float(lam.abs().max())
                tol = torch.clamp(1e-7 * w.abs(), min=4 * 1.1920929e-07 * lam_max)
                floor = w.abs() <= 1e-7 * lam_max
                if bool((((w - wp).abs() <= tol) | floor).all()):
                    break
            lam_prev = lam
        # Rayleigh-Ritz in the converged basis: Y = A B (or A^T B), T = Y^T Y = B^T A^T A B
        if row_sharded:
            Y = ops.aht(A, B.t().contiguous(), ops.empty((m_l, k), A))
        else:
            Y = ops.wta(A, B, ops.empty((k, n_l), A)).t().contiguous()
        G = ops.gram_wtw(Y, ops.zeros((kp, kp), A))
        T = self._allreduce(G)[:k, :k].double()
        lam, R = torch.linalg.eigh(0.5 * (T + T.t()))
        lam, R = lam.flip(0)[:kk], R.flip(1)[:, :kk]                    # keep the kk leading Ritz pairs
        s = torch.sqrt(torch.clamp(lam, min=0.0))
        sdiv = torch.clamp(s, min=self.eps * float(s.max()) if float(s.max()) > 0 else self.eps)   # rank-deficient data: no inf / nan
        rep = (B.double() @ R)                                          # replicated factor (d x k)
        loc = ops.aht(Y, R.t().to(torch.float32).contiguous(), ops.empty((Y.shape[0], kk), A)).double() / sdiv  # Y R / s
        # sign convention: make the largest-magnitude entry of every replicated vector positive (deterministic output)
        sg = torch.sign(rep.gather(0, rep.abs().argmax(dim=0, keepdim=True))).squeeze(0)
        sg[sg == 0] = 1
        rep, loc = rep * sg, loc * sg
        if row_sharded:
            return s, loc, rep.t().contiguous()                         # U local (m_loc x k), V replicated (k x n)
        return s, rep, loc.t().contiguous()                             # U replicated (m x k), V local (k x n_loc)

    def rel_error(self, U, S, V):
        """||A - U S V|| / ||A|| over all ranks, through the path's residual / norm kernels (no m_loc x n_loc temporary)."""
        Wf = (U.double() @ S.double()).to(torch.float32).contiguous()
        Vf = V.to(torch.float32).contiguous()
        num = self.comm1.allreduce(float(self.ops.resid_sqnorm(self.A, Wf, Vf)))
        den = self.comm1.allreduce(float(self.ops.sqnorm(self.A)))
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
