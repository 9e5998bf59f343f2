"""Custom clustering of NMF factors across perturbations + cosine silhouettes (NMFk).

Behavioural mirror of reference pyDNMFk/dist_clustering.py (`custom_clustering`, :5-188): given P groups of k column
vectors (W_all: m_loc x k x P, row-sharded over the p_r grid rows) build k clusters containing one vector of every
group by a greedy approximation of the linear-sum assignment against median centroids (100 fixed rounds, :114), then
score the clustering with cosine-distance silhouettes (:130-160).  This is control logic around the MU hot path
(SURVEY.md 8f row 2): it runs on torch tensors wherever the factors live (GPU in production, CPU in tests); the only
cross-rank traffic is a handful of tiny allreduces (k x k x P similarities, (kP)^2 Gram matrix).

Differences from the reference, none of which changes results: the P similarity matrices of one round are computed
and allreduced together (the centroids are fixed within a round, :115-119), the unused similarity product of
:111-113 is dropped, and the 100 fixed rounds stop at the first exact fixed point (a round after the first in which no
group is reordered: every later round would repeat it bit for bit; typically round 2 -- the clustering was 70 % of
the wall time of an NMFk sweep on a 65536 x 4096 matrix before).
"""
import numpy as np
import torch


def _as_tensor(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))


class custom_clustering:
    early_exit = True      # stop the 100 rounds at the first exact fixed point (False = run them all, for the tests)

    def __init__(self, Wall, Hall, params, ops=None):
        self.W_all = _as_tensor(Wall).clone()          # m_loc x k x P
        self.H_all = _as_tensor(Hall).clone()          # k x n_loc x P
        self.H_all = self.H_all.to(self.W_all.device)
        # the two contractions over the (long) row index -- similarities to the centroids, the silhouette Gram matrix --
        # run through the update engine's W^T A kernel when the factors live on the GPU (`ops` = HipOps); a checker back
        # end / CPU tensors take the torch expressions
        if ops is None and self.W_all.is_cuda:
            from .engine import HIP_OPS
            ops = HIP_OPS
        self.ops = ops if (ops is not None and hasattr(ops, "wta") and self.W_all.is_cuda) else None
        self.p_r, self.p_c = params.p_r, params.p_c
        self.comm1 = params.comm1
        self.eps = float(params.eps)
        self.p = self.p_r * self.p_c

    def _allreduce(self, t):
        """SUM over the world when W is row-sharded (the reference allreduces iff p_r != 1, e.g. :35, :77, :123)."""
        if self.p_r != 1:
            return self.comm1.allreduce(t)
        return t

    def normalize_by_W(self):
        """:30-40  unit 2-norm columns of W (global norm), H scaled inversely."""
        nrm = self._allreduce((self.W_all * self.W_all).sum(dim=0)) + self.eps      # k x P
        temp = torch.sqrt(nrm)
        self.W_all /= temp.unsqueeze(0)
        self.H_all *= temp.unsqueeze(1)

    @staticmethod
    def mad(data, flag=1, axis=-1):
        """:42-49  median (flag=1) / mean (flag=0) absolute deviation along `axis`."""
        if flag == 1:
            med = torch.nanmedian(data, dim=axis, keepdim=True).values
            return torch.nanmedian((data - med).abs(), dim=axis).values
        mean = torch.nanmean(data, dim=axis, keepdim=True)
        return torch.nanmean((data - mean).abs(), dim=axis)

    @staticmethod
    def greedy_lsa(A):
        """:59-69  repeatedly take the largest remaining similarity, strike its row and column."""
        X = np.array(A, dtype=np.float64, copy=True)
        pairs = []
        for _ in range(X.shape[0]):
            ind = np.unravel_index(int(np.argmax(X)), X.shape)
            pairs.append(ind)
            X[:, ind[1]] = -np.inf
            X[ind[0], :] = -np.inf
        return pairs

    @staticmethod
    def greedy_orders(dist):
        """`change_order(greedy_lsa(dist[:, :, p]))` for every group p at once: the k strike-out rounds run on a [P][k k] array
        (np.argmax takes the FIRST maximum in the row-major order of dist[:, :, p], as the per-group loop does), so the cost is
        k vectorised rounds instead of k P small ones -- at the NMFk sweep shape the per-group Python loop was a third of the
        clustering's wall time.  Returns a list of P orders (order[centroid] = feature)."""
        k, _, P = dist.shape
        X = np.array(np.transpose(dist, (2, 0, 1)), dtype=np.float64, copy=True)          # [P][centroid][feature]
        orders = np.tile(np.arange(k), (P, 1))
        rows = np.arange(P)
        for _ in range(k):
            flat = X.reshape(P, k * k).argmax(axis=1)
            c, f = flat // k, flat % k
            orders[rows, c] = f
            X[rows, :, f] = -np.inf
            X[rows, c, :] = -np.inf
        return [list(map(int, o)) for o in orders]

    @staticmethod
    def change_order(pairs):
        """:51-57  order[centroid] = feature."""
        ans = list(range(len(pairs)))
        for c, f in pairs:
            ans[int(c)] = int(f)
        return ans

    def dist_custom_clustering(self, centroids=None, vb=0):
        """:84-127  returns (centroids, W_all, H_all, permute_order)."""
        permute_order = []
        self.normalize_by_W()
        if centroids is None:
            centroids = self.W_all[:, :, 0].clone()
        P = self.W_all.shape[-1]
        k = self.W_all.shape[1]
        identity = list(range(k))
        rounds = 100
        for rnd in range(rounds):
            # similarities of every group's vectors to the centroids: k x k x P, one allreduce per round
            dist = self._allreduce(self._centroid_similarities(centroids)).cpu().numpy()
            orders = self.greedy_orders(dist)           # = [change_order(greedy_lsa(dist[:, :, p])) for p in range(P)]
            permute_order.extend(orders)
            if any(j != identity for j in orders):
                # all P groups reordered with one gather each (feature index = dim 1 of W_all, dim 0 of H_all)
                idx = torch.as_tensor(np.asarray(orders, dtype=np.int64).T.copy(), device=self.W_all.device)   # k x P
                self.W_all = torch.gather(self.W_all, 1, idx.unsqueeze(0).expand(self.W_all.shape[0], -1, -1))
                self.H_all = torch.gather(self.H_all, 0, idx.unsqueeze(1).expand(-1, self.H_all.shape[1], -1))
            elif rnd > 0 and self.early_exit:
                # Fixed point: these centroids were computed from the current W_all (rnd > 0) and nothing moved, so the
                # next round would recompute the same centroids and repeat this one exactly.  The reference grinds
                # through all 100 rounds (:114); the result -- including the list of orders -- is identical.
                permute_order.extend([list(identity) for _ in range((rounds - 1 - rnd) * P)])
                break
            centroids = _median_lower_upper_mean(self.W_all)
            cn = self._allreduce((centroids ** 2).sum(dim=0)) + self.eps
            centroids = centroids / torch.sqrt(cn)
        return centroids, self.W_all, self.H_all, permute_order

    def _centroid_similarities(self, centroids):
        """sim[c][f][p] = sum_m centroids[m][c] W_all[m][f][p]  (k x k x P): `dnmf_wta` with W = centroids (m x k) and the
        data operand = W_all seen as an m x (k P) matrix."""
        N, k, P = self.W_all.shape
        if self.ops is None or self.W_all.dtype != torch.float32:
            return torch.einsum("mc,mfp->cfp", centroids, self.W_all)
        flat = self.W_all.reshape(N, k * P)
        out = torch.empty(k, k * P, dtype=torch.float32, device=flat.device)
        self.ops.wta(flat, centroids.to(torch.float32).contiguous(), out)
        return out.view(k, k, P)

    def _gram_of_all_vectors(self):
        """(k P) x (k P) Gram matrix of all column vectors, in row blocks of at most 128 (the engine's rank limit)."""
        N, k, P = self.W_all.shape
        flat = self.W_all.reshape(N, k * P)
        if self.ops is None or self.W_all.dtype != torch.float32:
            return flat.t() @ flat
        flat = flat.contiguous()
        out = torch.empty(k * P, k * P, dtype=torch.float32, device=flat.device)
        for b0 in range(0, k * P, 128):
            b1 = min(k * P, b0 + 128)
            self.ops.wta(flat, flat[:, b0:b1].contiguous(), out[b0:b1])
        return out

    def dist_silhouettes(self):
        """:130-160  k x P cosine-distance silhouettes (re-runs the clustering first, as the reference does)."""
        self.dist_custom_clustering()
        N, k, n_pert = self.W_all.shape
        gram = self._allreduce(self._gram_of_all_vectors()).reshape(k, n_pert, k, n_pert)
        distances = torch.arccos(torch.clamp(gram, -1.0, 1.0)).cpu().numpy().astype(np.float64)
        if k == 1:
            return np.ones((k, n_pert))
        return self._silhouettes_from_distances(distances)

    @staticmethod
    def _silhouettes_from_distances(distances):
        """:147-160  a = mean distance to the own cluster's other members, b = mean distance to the nearest other cluster, per
        (cluster, perturbation); the reference's double loop as array operations (same sums over the same contiguous axis)."""
        k, n_pert = distances.shape[0], distances.shape[1]
        S = np.sum(distances, axis=3)                                   # [k][P][k]: sum over the members of every cluster
        idx = np.arange(k)
        a = 1 / (n_pert - 1) * S[idx, :, idx]                           # own cluster
        S = S.copy()
        S[idx, :, idx] = np.inf
        b = 1 / n_pert * np.min(S, axis=2)
        return (b - a) / np.maximum(a, b)

    def fit(self):
        """:163-188  [centroids, MAD of W around them, ordered H_all, per-cluster mean silhouette, mean silhouette, orders]."""
        centroids, _, _, orders = self.dist_custom_clustering()
        cent_std = self.mad(self.W_all, axis=-1)
        sils = self.dist_silhouettes()
        return [centroids, cent_std, self.H_all, sils.mean(axis=1), float(sils.flatten().mean()), orders]


def _median_lower_upper_mean(t):
    """np.median semantics along the last axis (mean of the two middle values for an even count); torch.median alone
    returns the lower one."""
    P = t.shape[-1]
    s = torch.sort(t, dim=-1).values
    if P % 2:
        return s[..., P // 2].clone()
    return 0.5 * (s[..., P // 2 - 1] + s[..., P // 2])
