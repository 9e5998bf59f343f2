"""One multiplicative-update step on a 1D or 2D process grid -- the MI355X update engine.

Drop-in for reference pyDNMFk/dist_nmf.py: `nmf_algorithms_1D(A_ij, W_i, H_j, params).update()`
(:618,:634-660) and `nmf_algorithms_2D(A_ij, W_ij, H_ij, params).update()` (:51,:66-92), MU method
for the Frobenius and KL objectives.  Arrays are torch float32 tensors on the GPU; W and H are
updated IN PLACE and returned, A_ij is never written (same ownership as the reference).

All arithmetic runs in libdnmf_hip.so through `ops` (pydnmfk_amd.engine.HipOps by default).  This
module only sequences kernels and the grid exchanges, placed exactly where the reference calls
mpi4py (allreduce / allgather / Reduce_scatter), minus its barriers.  There is no CPU path: `ops`
exists as a parameter so the choreography can be exercised by tests with a checker back end.

Methods: 'mu' (Frobenius and KL) and 'hals' (Frobenius; dist_nmf.py:411-470, :873-934 -- the same contractions, with
the multiply-divide replaced by column-sequential sweeps).  BCD (dist_nmf.py:474-579, :939-1047) is out of scope.
"""
import torch

_cache = {}


def _buf(key, numel, like):
    """Persistent scratch tensors: a fresh nmf_algorithms_* object is built every iteration (pyDNMF.py:154,169), so
    buffers live in a module cache -- ONE buffer per role and device, grown on demand (an NMFk sweep over k reuses it;
    every use is stream-ordered).  `release_buffers()` drops them."""
    dtype = torch.float64 if like.dtype == torch.float64 else torch.float32     # (bf16-stored data: float32 factors and products)
    key = (like.device, key[0], dtype)
    t = _cache.get(key)
    if t is None or t.numel() < numel:
        t = torch.zeros(numel, dtype=dtype, device=like.device)
        _cache[key] = t
    return t


def release_buffers():
    """Free the exchange buffers of this module and the kernel workspaces of the engine (the next step re-creates them)."""
    _cache.clear()
    from . import engine
    engine._ws_cache.clear()


def _default_ops(params=None, like=None):
    from .engine import ops_for
    return ops_for(params, like.dtype if like is not None else None)


def _kp(k):
    """the padded rank of the k x k buffers (dnmf_kp): 32 / 64 / 128, and 256 for the wide ranks 128 < k <= 256 (csrc/dnmf_wide.hip)"""
    if k < 1 or k > 256:
        raise ValueError("rank k=%d unsupported (1 <= k <= 256)" % k)
    return 32 if k <= 32 else (64 if k <= 64 else (128 if k <= 128 else 256))


def _pad64(x):
    return (x + 63) // 64 * 64


class _Base:
    def _dispatch(self, clamp):
        norm, method = self.norm.upper(), self.method.upper()
        if norm == 'FRO':
            if method == 'MU':
                self.Fro_MU_update(self.W_update, clamp)
            elif method == 'HALS':
                self.FRO_HALS_update(self.W_update, clamp)
            elif method == 'BCD':
                raise NotImplementedError("method 'bcd' is not part of the MI355X engine (mu / hals)")
            else:
                raise Exception('Not a valid method: Choose (mu/hals/bcd)')    # dist_nmf.py:84,652
        elif norm == 'KL':
            if method == 'MU':
                self.KL_MU_update(self.W_update, clamp)
            else:
                raise Exception('Not a valid method: Choose (mu)')             # dist_nmf.py:89,657
        else:
            raise Exception('Not a valid norm: Choose (fro/kl)')               # dist_nmf.py:91,659


    def _hals_w_sweep(self, W, AH, G, allreduce_norm):
        """Column-sequential W sweep (dist_nmf.py:884-891 / :428-434).  Without a cross-rank norm it is one library
        call; with p_r > 1 the 8-byte sum of squares of every column is allreduced between the column kernels, where
        the reference calls utils.norm (utils.py:388-391)."""
        ops, eps, k = self.ops, self.eps, self.k
        # (`params.hals_force_exchange`: measurement aid -- a one-rank group still runs the exchanged sweep, tools/rankbench.py)
        if not allreduce_norm or (self.comm1.size == 1 and not getattr(self.params, "hals_force_exchange", False)):
            # `params.hals_sweep = 'columns'`: k column launches instead of the persistent sweep, whose workgroups wait for
            # each other and must all be resident (a GPU shared with another process or stream cannot promise that)
            if getattr(self.params, "hals_sweep", None) == "columns" and hasattr(ops, "hals_update_w_columns"):
                ops.hals_update_w_columns(W, AH, G, eps)
            else:
                ops.hals_update_w(W, AH, G, eps)
            return
        ss2 = ops.hals_ss2(k, W)
        for kk in range(k):
            ops.hals_w_col(W, AH, G, kk, ss2, eps)
            self.comm1.allreduce_(ss2[kk:kk + 1])
        ops.hals_w_scale(W, k - 1, ss2)


class nmf_algorithms_1D(_Base):
    """1D grids: p_c == 1 (A and W row-sharded, H replicated) or p_r == 1 (A and H column-sharded,
    W replicated).  Reference dist_nmf.py:582-869."""

    def __init__(self, A_ij, W_i, H_j, params=None, ops=None):
        self.m, self.n, self.p_r, self.p_c, self.k = params.m, params.n, params.p_r, params.p_c, params.k
        self.params = params
        self.comm1 = params.comm1
        self.norm, self.method = params.norm, params.method
        self.A_ij, self.W_i, self.H_j = A_ij, W_i, H_j
        self.eps = float(params.eps)
        self.p = self.p_r * self.p_c
        self.W_update = params.W_update
        self.rank = self.comm1.rank
        self.local_W_m = self.W_i.shape[0]
        self.local_H_n = self.H_j.shape[1]
        self.ops = ops if ops is not None else _default_ops(params, A_ij)

    def update(self, clamp=False):
        """One step; `clamp=True` additionally applies H = max(H, eps), W = max(W, eps) after it
        (what PyNMF.fit does when i % 10 == 0, pyDNMF.py:170-172)."""
        if self._native_step(clamp):
            return self.W_i, self.H_j
        self._dispatch(clamp)
        return self.W_i, self.H_j

    def _native_step(self, clamp):
        """`params.exchange = 'native'`: the whole MU step, exchanges included, is ONE library call (dnmf_mu_*_step_1d over
        the RCCL communicator inside libdnmf_hip.so) -- same kernels in the same order as the choreography below, no Python
        between the launches.  float32 data, the product's own operator set, more than one rank."""
        hals = self.method.upper() == 'HALS' and self.norm.upper() == 'FRO'
        if (self.p == 1 and not getattr(self.params, "native_always", False)) or self.k > 128 or \
                not (hals or (self.method.upper() == 'MU' and self.norm.upper() in ('FRO', 'KL'))):
            return False            # (k > 128: the library-sequenced steps stop at the tuned kernels' rank; the choreography below does not)
        if getattr(self.params, "exchange", None) not in ("native", "native-hosted") or getattr(self.ops, "name", "") != "hip":
            return False
        fro = self.norm.upper() == 'FRO'
        if self.A_ij.dtype != torch.float32 and not (fro and self.A_ij.dtype == torch.bfloat16):   # bf16 storage: Frobenius only
            return False
        from .engine import native_comm_for
        nc = native_comm_for(self.params)
        if nc is None:
            return False
        if hals:
            nc.hals_step_1d(self.A_ij, self.W_i, self.H_j, self.eps, self.W_update, clamp,
                            column_sweep=(getattr(self.params, "hals_sweep", None) == "columns"))
            nc.steps += 1
            return True
        want = self._overlap_chunks(self.A_ij.shape[1]) if self.norm.upper() == 'FRO' else 1
        if nc.overlap_chunks != want:
            nc.set_overlap_chunks(want)
        nc.step_1d(self.norm, self.A_ij, self.W_i, self.H_j, self.eps, self.W_update, clamp)
        nc.steps += 1
        return True

    # ---- Frobenius (dist_nmf.py:716-771)
    def Fro_MU_update(self, W_update=True, clamp=False):
        ops, A, W, H, eps, k = self.ops, self.A_ij, self.W_i, self.H_j, self.eps, self.k
        if self.p == 1 and hasattr(ops, "mu_fro_step"):
            ops.mu_fro_step(A, W, H, eps, W_update, clamp)         # whole local step, one library call
            return
        kp = _kp(k)
        m_l, n_l = A.shape
        if W_update:                                               # Fro_MU_update_W :716-732
            if self.p_c == 1:                                      # no exchange: fused single pass over A
                G = _buf(("G", kp), kp * kp, A)[: kp * kp].view(kp, kp)
                ops.gram_hht(H, G)
                ops.aht_update_w(A, H, G, W, eps)
            else:                                                  # allreduce [A H^T | H H^T] (:681,:707)
                off = _pad64(m_l * k)
                buf = _buf(("ahg", m_l, k), off + kp * kp, A)
                AH, G = buf[: m_l * k].view(m_l, k), buf[off: off + kp * kp].view(kp, kp)
                ops.gram_hht(H, G)
                ops.aht(A, H, AH)
                self.comm1.allreduce_(buf[: off + kp * kp])
                ops.mu_update_w(W, AH, G, eps)
        nch = self._overlap_chunks(n_l)
        if nch > 1:
            self._fro_h_phase_overlapped(nch, clamp)
        else:
            off = _pad64(k * n_l)                                  # Fro_MU_update_H :736-751
            buf = _buf(("atwg", k, n_l), off + kp * kp, A)
            AtW, G = buf[: k * n_l].view(k, n_l), buf[off: off + kp * kp].view(kp, kp)
            ops.wta_gram(A, W, AtW, G)                             # W^T A and W^T W (:705), one call
            if self.p_r != 1:                                      # allreduce [W^T A | W^T W] (:681,:707)
                self.comm1.allreduce_(buf[: off + kp * kp])
            ops.mu_update_h(H, AtW, G, eps, clamp)
        if clamp:
            ops.clamp_min(W, eps)

    def _overlap_chunks(self, n_l):
        """Column chunks of the H phase on a row grid of more than two ranks (the 8-GPU configuration): the exchange of
        chunk c runs on the communicator's stream while W^T A of chunk c+1 is computed, only the last exchange is
        exposed.  Up to two ranks the single packed allreduce stays (one latency, nothing worth hiding behind).
        `params.overlap_chunks` (default 2 beyond two ranks; an explicit value is honoured from two ranks on -- bench.py
        times 1 / 2 / 4 in its warm-up and keeps the fastest) / `params.overlap_min_cols` (default 4096) tune it; 1 switches it off.
        Compute-side cost of the chunks on the 8-GPU shard (32768 x 8192, k = 64, tools/chunkbench.py, no exchange):
        1 / 2 / 4 / 8 chunks = 0.654 / 0.668 / 0.752 / 0.816 ms per step -- two chunks cost 13 us and hide about half of
        the exchange, four cost more than a 2 MiB allreduce is expected to take, hence the default."""
        if self.p_c != 1 or self.p_r < 2:
            return 1
        nch = getattr(self.params, "overlap_chunks", None)
        if nch is None:                                            # default: two ranks keep the single packed allreduce
            nch = 2 if self.p_r > 2 else 1
        nch = int(nch)
        if nch <= 1 or n_l < int(getattr(self.params, "overlap_min_cols", 4096)):
            return 1
        return max(1, min(nch, n_l // 64))

    def _fro_h_phase_overlapped(self, nch, clamp):
        """Fro_MU_update_H (:736-751) with the allreduce of [W^T A | W^T W] (:681,:707) cut into column chunks: chunk 0
        carries W^T W.  Each chunk is a contiguous buffer [k x cw (| KP x KP)], W^T A of the chunk is written with
        ld = cw, and the H update runs per chunk on column views of H -- same arithmetic, same per-element sums."""
        ops, A, W, H, eps, k = self.ops, self.A_ij, self.W_i, self.H_j, self.eps, self.k
        kp = _kp(k)
        m_l, n_l = A.shape
        cw = -(-(-(-n_l // nch)) // 64) * 64                       # chunk width: ceil(n_l / nch) rounded up to 64 columns
        starts = list(range(0, n_l, cw))
        total = sum(_pad64(k * min(cw, n_l - c0)) for c0 in starts) + kp * kp
        buf = _buf(("atwg_chunks", k, n_l), total, A)
        G = buf[_pad64(k * min(cw, n_l)): _pad64(k * min(cw, n_l)) + kp * kp].view(kp, kp)
        ops.gram_wtw(W, G)
        pending, off = [], 0
        for ci, c0 in enumerate(starts):
            c1 = min(n_l, c0 + cw)
            ne = _pad64(k * (c1 - c0))
            AtW = buf[off: off + k * (c1 - c0)].view(k, c1 - c0)
            ops.wta(A[:, c0:c1], W, AtW)
            span = ne + (kp * kp if ci == 0 else 0)                # chunk 0: [W^T A chunk | W^T W] in one message
            pending.append((c0, c1, AtW, self.comm1.allreduce_begin(buf[off: off + span])))
            off += span
        for c0, c1, AtW, work in pending:
            work.wait()
            ops.mu_update_h(H[:, c0:c1], AtW, G, eps, clamp)

    # ---- HALS / Frobenius (dist_nmf.py:873-934)
    def FRO_HALS_update(self, W_update=True, clamp=False):
        ops, A, W, H, eps, k = self.ops, self.A_ij, self.W_i, self.H_j, self.eps, self.k
        kp = _kp(k)
        m_l, n_l = A.shape
        if W_update:                                               # FRO_HALS_update_W :873-891
            off = _pad64(m_l * k)
            buf = _buf(("ahg", m_l, k), off + kp * kp, A)
            AH, G = buf[: m_l * k].view(m_l, k), buf[off: off + kp * kp].view(kp, kp)
            ops.gram_hht(H, G)                                     # :882
            ops.aht(A, H, AH)                                      # :883
            if self.p_c != 1:
                self.comm1.allreduce_(buf[: off + kp * kp])
            self._hals_w_sweep(W, AH, G, allreduce_norm=(self.p_r != 1))   # norm(..., p=self.p_r) :889
        off = _pad64(k * n_l)                                      # FRO_HALS_update_H :893-909
        buf = _buf(("atwg", k, n_l), off + kp * kp, A)
        AtW, G = buf[: k * n_l].view(k, n_l), buf[off: off + kp * kp].view(kp, kp)
        ops.wta_gram(A, W, AtW, G)                                 # :902-903
        if self.p_r != 1:
            self.comm1.allreduce_(buf[: off + kp * kp])
        ops.hals_update_h(H, AtW, G, eps)                          # :905-909
        if clamp:                                                  # pyDNMF.py:170-172
            ops.clamp_min(H, eps)
            ops.clamp_min(W, eps)

    # ---- KL (dist_nmf.py:776-869)
    def KL_MU_update(self, W_update=True, clamp=False):
        ops, A, W, H, eps, k = self.ops, self.A_ij, self.W_i, self.H_j, self.eps, self.k
        if self.p == 1 and hasattr(ops, "mu_kl_step"):
            ops.mu_kl_step(A, W, H, eps, W_update, clamp)
            return
        m_l, n_l = A.shape
        if W_update:                                               # KL_MU_update_W :813-830
            off = _pad64(m_l * k)
            buf = _buf(("uhx", m_l, k), off + 256, A)
            UHT, x2 = buf[: m_l * k].view(m_l, k), buf[off: off + k]
            ops.rowsum(H, x2)                                      # :827
            ops.kl_uht(A, W, H, eps, UHT)                          # :806,:810
            if self.p_c != 1:
                self.comm1.allreduce_(buf[: off + k])              # :797,:707
            ops.kl_update_w(W, UHT, x2, eps)
        off = _pad64(k * n_l)                                      # KL_MU_update_H :832-849
        buf = _buf(("wux", k, n_l), off + 256, A)
        WTU, x1 = buf[: k * n_l].view(k, n_l), buf[off: off + k]
        ops.colsum(W, x1)                                          # :846
        ops.kl_wtu(A, W, H, eps, WTU)                              # :806,:808
        if self.p_r != 1:
            self.comm1.allreduce_(buf[: off + k])
        ops.kl_update_h(H, WTU, x1, eps, clamp)
        if clamp:
            ops.clamp_min(W, eps)


class nmf_algorithms_2D(_Base):
    """2D grid (p_r > 1 and p_c > 1).  Reference dist_nmf.py:7-407.  `row_comm` is the size-p_r
    group sharing grid column j, `col_comm` the size-p_c group sharing grid row i (dist_comm.py:34,48)."""

    def __init__(self, A_ij, W_ij, H_ij, params=None, ops=None):
        self.params = params
        self.m, self.n, self.p_r, self.p_c, self.k = params.m, params.n, params.p_r, params.p_c, params.k
        self.comm1 = params.comm1
        self.cartesian1d_row, self.cartesian1d_column, self.comm = params.row_comm, params.col_comm, params.comm
        self.A_ij, self.W_ij, self.H_ij = A_ij, W_ij, H_ij
        self.eps = float(params.eps)
        self.p = self.p_r * self.p_c
        self.W_update = params.W_update
        self.norm, self.method = params.norm, params.method
        self.rank = self.comm1.rank
        self.local_W_m = self.W_ij.shape[0]
        self.local_H_n = self.H_ij.shape[1]
        self.ops = ops if ops is not None else _default_ops(params, A_ij)
        # per-member slice sizes inside the sub-communicators: the partition rule (utils.py:99-103), unless pruning
        # changed them (then PyNMF has exchanged the actual sizes once and left them on params)
        from .utils import determine_block_params
        m_l, n_l = A_ij.shape
        counts = getattr(params, "_slice_counts", None)
        if counts is not None:
            self.w_counts, self.h_counts = counts
        else:
            self.w_counts = [determine_block_params(q, (self.p_c, 1), (m_l, self.k)).determine_block_shape_asymm()[0]
                             for q in range(self.p_c)]
            self.h_counts = [determine_block_params(q, (1, self.p_r), (self.k, n_l)).determine_block_shape_asymm()[1]
                             for q in range(self.p_r)]

    def update(self, clamp=False, more=False):
        """One step.  `more` (PyNMF.fit passes it for every step but the last): another step on the SAME factor tensors follows --
        with `params.overlap_2d` the allgather of the updated H slices that the next step's W phase begins with (dist_nmf.py:
        195-197, :283-287) is started now, behind this step's H update, and runs while this step's last kernels, the host's way
        to the next step and that step's first kernels and allreduce are issued."""
        if not self._native_step(clamp):
            self._dispatch(clamp)
            if more and getattr(self.params, "overlap_2d", False) and self.W_update and hasattr(self.cartesian1d_row, "allgather_blocks_begin"):
                self._prefetch_h()
        return self.W_ij, self.H_ij

    def _prefetch_h(self):
        import weakref
        h = self.cartesian1d_row.allgather_blocks_begin(self.H_ij, [(self.k, c) for c in self.h_counts])
        # valid for this very tensor object in this very state: the next step's gather_H takes it, anything else discards it
        self.params._h_prefetch = (weakref.ref(self.H_ij), self.H_ij._version, self.H_ij.data_ptr(), tuple(self.H_ij.shape), h)

    def _take_prefetched_h(self):
        pf = self.params.__dict__.pop("_h_prefetch", None)
        if pf is None:
            return None
        ref, ver, ptr, shape, h = pf
        blocks = h.wait()                      # (always completed: an abandoned collective must not outlive the step)
        if ref() is self.H_ij and ver == self.H_ij._version and ptr == self.H_ij.data_ptr() and shape == tuple(self.H_ij.shape):
            self.params._h_prefetch_hits = getattr(self.params, "_h_prefetch_hits", 0) + 1      # (tests read it)
            return blocks
        return None

    def _native_step(self, clamp):
        """`params.exchange = 'native'`: the whole MU step -- kernels, allreduce / allgather / reduce-scatter over the RCCL
        communicators inside libdnmf_hip.so, kernels -- is ONE library call (dnmf_mu_*_step_2d): same kernels in the same
        order as the choreography below, even and ragged grids.  Pruned factors, bf16-stored A and the other operator sets
        keep the choreography."""
        if getattr(self.params, "exchange", None) not in ("native", "native-hosted") or getattr(self.ops, "name", "") != "hip" or self.k > 128:
            return False
        hals = self.method.upper() == 'HALS' and self.norm.upper() == 'FRO'
        if not (hals or (self.method.upper() == 'MU' and self.norm.upper() in ('FRO', 'KL'))):
            return False
        if self.A_ij.dtype != torch.float32 and not (self.norm.upper() == 'FRO' and self.A_ij.dtype == torch.bfloat16):
            return False
        from .engine import native_comm_for
        nc = native_comm_for(self.params)
        if nc is None:
            return False
        ok = nc.step_2d_ok(self.A_ij, self.W_ij, self.H_ij)
        if getattr(self.params, "_slice_counts", None) is not None:
            # pruning is on (the reference's default): the slice sizes were exchanged, and the library's steps apply only if
            # EVERY rank's slices still follow the partition rule of its pruned block -- decided once per shape by all ranks
            # together (a rank-local decision would let the grid split between the two sequencings)
            key = (tuple(self.A_ij.shape), tuple(map(tuple, self.params._slice_counts)))
            cache = self.params.__dict__.setdefault("_native_2d_ok", {})
            if key not in cache:
                m_l, n_l = self.A_ij.shape
                rule = ([m_l // self.p_c + (1 if q < m_l % self.p_c else 0) for q in range(self.p_c)],
                        [n_l // self.p_r + (1 if q < n_l % self.p_r else 0) for q in range(self.p_r)])
                mine = ok and [list(self.params._slice_counts[0]), list(self.params._slice_counts[1])] == [rule[0], rule[1]]
                cache[key] = int(self.comm1.allreduce(0 if mine else 1)) == 0
            ok = cache[key]
        if not ok:
            return False
        nc.step_2d("hals" if hals else self.norm, self.A_ij, self.W_ij, self.H_ij, self.eps, self.W_update, clamp)
        nc.steps += 1
        return True

    # ---- gathers (dist_nmf.py:163-165, :195-197, :268-291)
    def gather_W(self):
        blocks = self.cartesian1d_column.allgather_blocks(self.W_ij, [(c, self.k) for c in self.w_counts])
        if len(blocks) == 1:
            return blocks[0]
        if len(set(self.w_counts)) == 1:       # equal row blocks: the receive buffer already IS the vstack (no copy)
            b0 = blocks[0]
            return torch.as_strided(b0, (sum(self.w_counts), self.k), (self.k, 1))
        return torch.cat(blocks, dim=0)                                             # vstack -> W_i [m_l x k]

    def gather_H(self, stacked=None):
        """H_j [k x n_l] = hstack of the row group's slices (:283-287).  `stacked` names the operator that will consume it
        ('aht_hblocks' / 'kl_uht_hblocks'): when the slices are equal and whole k-tiles wide and the operator set has it, the
        allgather's receive buffer itself is returned, viewed [p_r][k][n_h] -- the kernels read H as column blocks and nothing
        is re-assembled.  Otherwise (ragged slices, bf16-stored A, the bf16x6 operator set) the blocks are concatenated:
        one copy of k x n_l floats."""
        blocks = self._take_prefetched_h()     # (params.overlap_2d: started behind the previous step's H update)
        if blocks is None:
            blocks = self.cartesian1d_row.allgather_blocks(self.H_ij, [(self.k, c) for c in self.h_counts])
        if len(blocks) == 1:
            return blocks[0]
        if stacked and self._h_blockable(stacked):
            nh = self.h_counts[0]
            return torch.as_strided(blocks[0], (len(blocks), self.k, nh), (self.k * nh, nh, 1))
        return torch.cat(blocks, dim=1)                                              # hstack -> H_j [k x n_l]

    def _h_blockable(self, op):
        hc = self.h_counts
        return (len(set(hc)) == 1 and hc[0] % 32 == 0 and getattr(self.ops, op, None) is not None
                and self.A_ij.dtype == torch.float32)

    def _aht(self, H_j, V):
        """A_ij H_j^T with H_j either one matrix or the stacked column blocks of gather_H (AH_glob :198)."""
        return self.ops.aht_hblocks(self.A_ij, H_j, V) if H_j.dim() == 3 else self.ops.aht(self.A_ij, H_j, V)

    def _scatter_to_W(self, V):
        """Reduce_scatter over the column group of an (m_l x k) buffer -> (m_w x k)  (:202, :340)."""
        return self.cartesian1d_column.reduce_scatter_rows(V, self.w_counts)

    def _scatter_to_H(self, Y):
        """Reduce_scatter over the row group of Y^T (n_l x k) -> (n_h x k) -> transpose (:169-171, :314-316).  Only for
        ragged column slices; equal slices take `_product_scattered_to_H` (no transposes)."""
        ks = self.cartesian1d_row.reduce_scatter_rows(Y.t().contiguous(), self.h_counts)
        return ks.t().contiguous()

    def _product_scattered_to_H(self, product):
        """The k x n_l product of the H phase, reduce-scattered over the row group into this rank's k x n_h slice.
        `product(c0, c1, out)` writes the product restricted to the local columns [c0, c1) into the k x (c1 - c0) block
        `out`.  The reference transposes to n_l x k so that MPI's Reduce_scatter can cut row blocks (:169-171, :314-316).
        With equal slices the product is instead formed slice by slice straight into a [p_r][k][n_h] buffer -- member q's
        slice is a contiguous block, which is what reduce_scatter_tensor cuts -- and nothing is transposed or copied."""
        k, hc = self.k, self.h_counts
        n_l = sum(hc)
        # slices that do not start on 16-byte boundaries (n_h % 4 != 0) would send every launch but the first down the
        # predicated kernels (and bf16x6 needs whole 128-column tiles): one aligned full-width product + the transposes then
        sliceable = len(set(hc)) == 1 and hc[0] % 4 == 0 and (getattr(self.ops, "name", "") != "hip-bf16x6" or hc[0] % 128 == 0)
        if not sliceable:
            Y = _buf(("Y", k, n_l), k * n_l, self.A_ij)[: k * n_l].view(k, n_l)
            product(0, n_l, Y)
            return self._scatter_to_H(Y)
        nh, p = hc[0], len(hc)
        buf = _buf(("Yb", k, n_l), p * k * nh, self.A_ij)[: p * k * nh]
        for q in range(p):
            product(q * nh, (q + 1) * nh, buf[q * k * nh: (q + 1) * k * nh].view(k, nh))
        return self.cartesian1d_row.reduce_scatter_rows(buf.view(p * k, nh), [k] * p)

    # ---- Frobenius (dist_nmf.py:207-263)
    def Fro_MU_update(self, W_update=True, clamp=False):
        ops, A, W, H, eps, k = self.ops, self.A_ij, self.W_ij, self.H_ij, self.eps, self.k
        kp = _kp(k)
        m_l, n_l = A.shape
        G = _buf(("G", kp), kp * kp, A)[: kp * kp].view(kp, kp)
        if W_update:                                               # Fro_MU_update_W :227-245
            ops.gram_hht(H, G)
            self.comm1.allreduce_(G)                               # global_gram :114
            H_j = self.gather_H(stacked="aht_hblocks")             # AH_glob :195-197
            V = self._aht(H_j, _buf(("V", m_l, k), m_l * k, A)[: m_l * k].view(m_l, k))     # :198
            AH = self._scatter_to_W(V)                             # :202
            ops.mu_update_w(W, AH, G, eps)                         # :244-245
        ops.gram_wtw(W, G)                                         # Fro_MU_update_H :207-225
        self.comm1.allreduce_(G)
        W_i = self.gather_W()                                      # ATW_glob :163-165
        AtW = self._product_scattered_to_H(lambda c0, c1, out: ops.wta(A[:, c0:c1], W_i, out))   # :166, :169-171
        ops.mu_update_h(H, AtW, G, eps, clamp)                     # :224-225
        if clamp:
            ops.clamp_min(W, eps)

    # ---- HALS / Frobenius (dist_nmf.py:411-470)
    def FRO_HALS_update(self, W_update=True, clamp=False):
        ops, A, W, H, eps, k = self.ops, self.A_ij, self.W_ij, self.H_ij, self.eps, self.k
        kp = _kp(k)
        m_l, n_l = A.shape
        G = _buf(("G", kp), kp * kp, A)[: kp * kp].view(kp, kp)
        if W_update:                                               # FRO_HALS_update_W :411-434
            ops.gram_hht(H, G)
            self.comm1.allreduce_(G)                               # :426
            H_j = self.gather_H(stacked="aht_hblocks")
            V = self._aht(H_j, _buf(("V", m_l, k), m_l * k, A)[: m_l * k].view(m_l, k))     # AH_glob :427
            AH = self._scatter_to_W(V).contiguous()
            self._hals_w_sweep(W, AH, G, allreduce_norm=True)      # norm(..., p=self.p_r), p_r > 1 on a 2D grid :432
        ops.gram_wtw(W, G)                                         # FRO_HALS_update_H :436-452
        self.comm1.allreduce_(G)
        W_i = self.gather_W()
        AtW = self._product_scattered_to_H(lambda c0, c1, out: ops.wta(A[:, c0:c1], W_i, out))   # ATW_glob :448
        ops.hals_update_h(H, AtW, G, eps)                          # :449-452
        if clamp:
            ops.clamp_min(H, eps)
            ops.clamp_min(W, eps)

    # ---- KL (dist_nmf.py:351-407)
    def KL_MU_update(self, W_update=True, clamp=False):
        ops, A, W, H, eps, k = self.ops, self.A_ij, self.W_ij, self.H_ij, self.eps, self.k
        m_l, n_l = A.shape
        x = _buf(("x", 256), 256, A)[:k]
        if W_update:                                               # KL_MU_update_W :351-369
            ops.rowsum(H, x)
            self.comm1.allreduce_(x)                               # sum_axis :346-349
            W_i, H_j = self.gather_W(), self.gather_H(stacked="kl_uht_hblocks")   # gather_W_H :367
            V = _buf(("V", m_l, k), m_l * k, A)[: m_l * k].view(m_l, k)
            if H_j.dim() == 3:                                     # the receive buffer as it is: H as column blocks
                UHT = ops.kl_uht_hblocks(A, W_i, H_j, eps, V)      # :337-338
            else:
                UHT = ops.kl_uht(A, W_i, H_j, eps, V)
            sk = self._scatter_to_W(UHT)                           # :340
            ops.kl_update_w(W, sk, x, eps)                         # :369
        ops.colsum(W, x)                                           # KL_MU_update_H :371-389
        self.comm1.allreduce_(x)
        W_i = self.gather_W()                                      # :387 (the W phase changed W, not H: the H_j it
        if not W_update:                                           #  gathered is still current -- one exchange less)
            H_j = self.gather_H(stacked="kl_uht_hblocks")
        if H_j.dim() == 3:                                         # member q's columns are block q of the stack
            # :311-312, :314-316.  Round 5: ONE full-width product on H_j assembled from the gathered blocks, then the k x n_l result
            # is cut into the reduce-scatter's member blocks -- p_r sliced launches that wrote the blocks directly cost 4.19 ms
            # against 3.97 + 0.04 ms on the config-4 block (each slice pays its own partial slabs and tail; tools/dbg/kl_slices.py)
            nh, p = self.h_counts[0], len(self.h_counts)
            Hf = _buf(("Hjf", k, n_l), k * n_l, A)[: k * n_l].view(k, n_l)
            Hf.view(k, p, nh).copy_(H_j.permute(1, 0, 2))
            Y = _buf(("Y", k, n_l), k * n_l, A)[: k * n_l].view(k, n_l)
            ops.kl_wtu(A, W_i, Hf, eps, Y)
            buf = _buf(("Yb", k, n_l), p * k * nh, A)[: p * k * nh]
            buf.view(p, k, nh).copy_(Y.view(k, p, nh).permute(1, 0, 2))
            ks = self.cartesian1d_row.reduce_scatter_rows(buf.view(p * k, nh), [k] * p)
        else:
            ks = self._product_scattered_to_H(
                lambda c0, c1, out: ops.kl_wtu(A[:, c0:c1], W_i, H_j[:, c0:c1], eps, out))
        ops.kl_update_h(H, ks, x, eps, clamp)                      # :389
        if clamp:
            ops.clamp_min(W, eps)
