"""Checker back end for the update choreography (TEST INFRASTRUCTURE, lives under tests/ only).

Implements the `ops` interface of pydnmfk_amd.engine.HipOps with the oracle's numpy arithmetic on CPU
torch tensors, so that pydnmfk_amd.dist_nmf / pyDNMF / dist_comm (kernel sequencing + collectives) can
be exercised under gloo without a GPU.  It deliberately has no `mu_fro_step` / `mu_kl_step`, so the
primitive-by-primitive path -- the one multi-rank runs take -- is what gets tested.
"""
import numpy as np
import torch


def _n(t):
    # a bfloat16-STORED data block (params.precision = 'bfloat16') means its exact float32 widening
    return t.float().numpy() if t.dtype == torch.bfloat16 else t.numpy()


class OracleOps:
    name = "oracle-double"

    def gram_hht(self, H, out):
        k = H.shape[0]
        out.zero_()
        out[:k, :k] = torch.from_numpy(np.matmul(_n(H), _n(H).T))        # dist_nmf.py:679
        return out

    def gram_wtw(self, W, out):
        k = W.shape[1]
        out.zero_()
        out[:k, :k] = torch.from_numpy(np.matmul(_n(W).T, _n(W)))
        return out

    def aht(self, A, H, out):
        out.copy_(torch.from_numpy(np.matmul(_n(A), _n(H).T)))          # :705
        return out

    def wta(self, A, W, out):
        out.copy_(torch.from_numpy(np.matmul(_n(W).T, _n(A))))
        return out

    def wta_gram(self, A, W, out, G):
        self.gram_wtw(W, G)
        return self.wta(A, W, out)

    def mu_update_w(self, W, AH, G, eps):
        k = W.shape[1]
        w = _n(W)
        w *= _n(AH) / (np.matmul(w, _n(G)[:k, :k]) + np.float32(eps))   # :731-732

    def mu_update_h(self, H, AtW, G, eps, clamp=False):
        k = H.shape[0]
        h = _n(H)
        h *= _n(AtW) / (np.matmul(h.T, _n(G)[:k, :k]) + np.float32(eps)).T   # :750-751
        if clamp:
            np.maximum(h, np.float32(eps), out=h)

    def aht_update_w(self, A, H, G, W, eps):
        AH = np.matmul(_n(A), _n(H).T)
        self.mu_update_w(W, torch.from_numpy(AH), G, eps)

    # ---- HALS sweeps (dist_nmf.py:884-891, :905-909)
    def hals_ss2(self, k, like):
        return torch.zeros(k, dtype=torch.float64)

    def hals_w_col(self, W, AH, G, kk, ss2, eps):
        k = W.shape[1]
        w, g = _n(W), _n(G)[:k, :k]
        if kk > 0 and float(ss2[kk - 1]) > 0:
            w[:, kk - 1] /= np.float32(np.sqrt(float(ss2[kk - 1])))
        t = w[:, kk] * g[kk, kk] + _n(AH)[:, kk] - w.dot(g[:, kk])
        w[:, kk] = np.maximum(t, np.float32(eps))
        ss2[kk] = float(np.linalg.norm(w[:, kk], ord=2)) ** 2

    def hals_w_scale(self, W, col, ss2):
        if float(ss2[col]) > 0:
            _n(W)[:, col] /= np.float32(np.sqrt(float(ss2[col])))

    def hals_update_w(self, W, AH, G, eps):
        k = W.shape[1]
        ss2 = self.hals_ss2(k, W)
        for kk in range(k):
            self.hals_w_col(W, AH, G, kk, ss2, eps)
        self.hals_w_scale(W, k - 1, ss2)

    def hals_update_h(self, H, AtW, G, eps):
        k = H.shape[0]
        h, g, a = _n(H), _n(G)[:k, :k], _n(AtW)
        for kk in range(k):
            t = h[kk, :] + a[kk, :] - g[kk, :].dot(h)
            h[kk, :] = np.maximum(t, np.float32(eps))

    def kl_uht(self, A, W, H, eps, out):
        U = _n(A) / (_n(W) @ _n(H) + np.float32(eps))                   # :806
        out.copy_(torch.from_numpy(U @ _n(H).T))                        # :810
        return out

    def aht_hblocks(self, A, Hs, out):
        """H as the allgather's stacked column blocks [p][k][n_h]: np.hstack (dist_nmf.py:195-197), then aht."""
        assert Hs.dim() == 3 and Hs.is_contiguous()
        return self.aht(A, torch.cat(list(Hs), dim=1), out)

    def kl_uht_hblocks(self, A, W, Hs, eps, out):
        """H as the allgather's stacked column blocks [p][k][n_h]: np.hstack (dist_nmf.py:283-287), then kl_uht."""
        assert Hs.dim() == 3 and Hs.is_contiguous()
        return self.kl_uht(A, W, torch.cat(list(Hs), dim=1), eps, out)

    def kl_wtu(self, A, W, H, eps, out):
        U = _n(A) / (_n(W) @ _n(H) + np.float32(eps))
        out.copy_(torch.from_numpy(_n(W).T @ U))                        # :808
        return out

    def rowsum(self, H, out):
        out.copy_(torch.from_numpy(_n(H).sum(axis=1)))
        return out

    def colsum(self, W, out):
        out.copy_(torch.from_numpy(_n(W).sum(axis=0)))
        return out

    def kl_update_w(self, W, S, x, eps):
        w = _n(W)
        w *= _n(S) / (_n(x)[None, :] + np.float32(eps))                 # :828-830

    def kl_update_h(self, H, S, x, eps, clamp=False):
        h = _n(H)
        h *= _n(S) / (_n(x)[:, None] + np.float32(eps))                 # :847-849
        if clamp:
            np.maximum(h, np.float32(eps), out=h)

    def clamp_min(self, X, eps):
        x = _n(X)
        np.maximum(x, np.float32(eps), out=x)

    def scale_cols_div(self, W, s, eps):
        w = _n(W)
        w /= _n(s)[None, :] + np.float32(eps)                           # pyDNMF.py:192

    def scale_rows_mul(self, H, s):
        h = _n(H)
        h *= _n(s)[:, None]                                             # pyDNMF.py:193

    def sqnorm(self, A):
        return torch.tensor([float(np.linalg.norm(_n(A))) ** 2], dtype=torch.float64)

    def resid_sqnorm(self, A, W, H):
        return torch.tensor([float(np.linalg.norm(_n(A) - _n(W) @ _n(H))) ** 2], dtype=torch.float64)

    def column_err_sums(self, A, W, H):
        a = _n(A).astype(np.float64)
        d = a - (_n(W) @ _n(H)).astype(np.float64)                      # pyDNMF.py:229-231 (float32 product, as numpy forms it)
        return torch.from_numpy((d * d).sum(0)), torch.from_numpy((a * a).sum(0))

    def empty(self, shape, like):
        return torch.empty(shape, dtype=torch.float32)

    def zeros(self, shape, like):
        return torch.zeros(shape, dtype=torch.float32)
