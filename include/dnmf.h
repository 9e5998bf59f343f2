/*
 * dnmf.h -- C ABI of libdnmf_hip.so: the MI355X (gfx950) multiplicative-update engine
 * that replaces the numpy hot path of lanl/pyDNMFk.
 *
 * The reference has no FFI layer: its hot path is reached by plain Python calls
 * (pyDNMFk/dist_nmf.py, pyDNMFk/pyDNMF.py).  Each entry point below names the reference
 * lines it replaces.  Calling convention:
 *   - every matrix is dense row-major fp32 in DEVICE memory, given as pointer + leading
 *     dimension (elements); the caller owns every buffer, nothing is retained;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is
 *     stream-ordered, no entry point synchronises the device or allocates memory; the library
 *     reads no environment variables;
 *   - every leading dimension must be >= the logical row length (lda >= n, ...);
 *   - `ws` is caller-provided device scratch of at least dnmf_ws_bytes(m, n, k) bytes;
 *   - return value: 0 on success, negative DNMF_E* on error (dnmf_last_error() has text);
 *   - k <= DNMF_MAX_K.  Internally k is padded to KP = 32/64/128/256; "gram" buffers G are
 *     always KP x KP, ld = KP, zero padded (dnmf_kp(k) returns KP).
 * Collectives: the kernels above the "Grid exchanges" section never communicate -- a host may issue the p_r x p_c grid
 * exchanges itself between these calls, exactly where the reference calls mpi4py (dist_nmf.py:681,707,114,163,169,195,
 * 202; pydnmfk_amd/dist_nmf.py does so over torch.distributed).  The last section binds RCCL at run time and offers
 * communicators plus whole 1D steps that enqueue kernels -> allreduce -> kernels on one stream.
 */
#ifndef DNMF_H
#define DNMF_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DNMF_MAX_K 256        /* the primitives, the local whole steps and the whole fits */
#define DNMF_TUNED_MAX_K 128  /* the tuned kernels' own limit: beyond it the contractions run as two passes over A and the k x k products
                               * on plain MFMA kernels (csrc/dnmf_wide.hip).  Also the limit of the float64 path, the bf16x6 arithmetic,
                               * bf16 storage of A in the error evaluation, the fused dnmf_aht_update_w, the block-column (hblocks)
                               * products and the library-sequenced grid steps (dnmf_*_step_1d / _2d: the host keeps its choreography) */
#define DNMF_OK 0
#define DNMF_EINVAL (-1)   /* bad shape / null pointer / k too large */
#define DNMF_EWS (-2)      /* workspace too small */
#define DNMF_EHIP (-3)     /* HIP launch error */
#define DNMF_ECOMM (-4)    /* RCCL missing or an RCCL call failed */

const char* dnmf_last_error(void);
int dnmf_version(void);
/* padded rank used for internal k x k buffers (32, 64, 128 or 256); <0 if k unsupported */
int dnmf_kp(int k);
/* scratch bytes sufficient for ANY entry point below on an m x n block with rank k */
size_t dnmf_ws_bytes(long m, long n, int k);

/* ---- Gram matrices (dist_nmf.py:679 `np.matmul(A.T, A)` in global_gram; :113 in 2D) ---- */
/* G[KP x KP] = H H^T for H [k x n] (called as global_gram(H.T), dist_nmf.py:729,241) */
int dnmf_gram_hht(const float* H, int k, long n, long ldh, float* G, void* ws, size_t ws_bytes, void* stream);
/* G[KP x KP] = W^T W for W [m x k] (global_gram(W), dist_nmf.py:748,222) */
int dnmf_gram_wtw(const float* W, long m, int k, long ldw, float* G, void* ws, size_t ws_bytes, void* stream);

/* ---- The two big contractions (dist_nmf.py:705 `np.matmul(A, B)` in global_mm; :166,:198 in 2D) ---- */
/* AH[m x k] = A[m x n] H[k x n]^T        (global_mm(A_ij, H_j.T), dist_nmf.py:730; AH_glob :198) */
int dnmf_aht(const float* A, long m, long n, long lda, const float* H, int k, long ldh,
             float* AH, long ldah, void* stream);
/* The same product with H given as n / nh COLUMN BLOCKS stacked [q][k][nh] -- the receive buffer of the allgather of the
 * ranks' k x nh slices (AH_glob, dist_nmf.py:195-197; np.hstack there), so that the 2D step needs no re-assembly copy.
 * n % nh == 0 and nh % 32 == 0 (DNMF_EINVAL otherwise: assemble H and call dnmf_aht). */
int dnmf_aht_hblocks(const float* A, long m, long n, long lda, const float* Hs, long nh, int k,
                     float* AH, long ldah, void* stream);
/* AtW[k x n] = W[m x k]^T A[m x n]       (global_mm(W_i.T, A_ij), dist_nmf.py:749; ATW_glob :166) */
int dnmf_wta(const float* A, long m, long n, long lda, const float* W, int k, long ldw,
             float* AtW, long ldatw, void* ws, size_t ws_bytes, void* stream);
/* W^T A AND the Gram matrix of the same W in one call: AtW as dnmf_wta, G as dnmf_gram_wtw (KP x KP, zero padded) -- the two
 * reductions the H phase sends through ONE allreduce (dist_nmf.py:705 twice, :681 / :707, :747-748).  For k <= 16 the Gram
 * rides in the same two launches: the wave of column block 0 of every row chunk sums W^T W over its rows (the W value a lane
 * holds is the A and the B operand of that product: one more MFMA per step, no more loads), the reduction launch sums the
 * partial tiles.  For k > 16 it is dnmf_gram_wtw followed by dnmf_wta (a riding Gram in the 32-wide kernel was measured:
 * no gain over the two launches it saves).  `ws` >= dnmf_ws_bytes(m, n, k). */
int dnmf_wta_gram(const float* A, long m, long n, long lda, const float* W, int k, long ldw, float* AtW, long ldatw,
                  float* G, void* ws, size_t ws_bytes, void* stream);

/* ---- Multiplicative updates (element-wise multiply/divide with the small k x k product) ----
 * G must be symmetric (the Gram matrices are).  These two kernels address a tile through 32-bit offsets: leading
 * dimensions up to 2^23 (W side) / 2^24 (H side) elements, DNMF_EINVAL beyond. */
/* W *= AH / (W G + eps), G = H H^T       (dist_nmf.py:731-732, :244-245) */
int dnmf_mu_update_w(float* W, long m, int k, long ldw, const float* AH, long ldah, const float* G,
                     float eps, void* stream);
/* H *= AtW / (H^T G + eps)^T, G = W^T W  (dist_nmf.py:750-751, :224-225); clamp!=0 also applies
 * H = max(H, eps) afterwards (pyDNMF.py:156,171) */
int dnmf_mu_update_h(float* H, int k, long n, long ldh, const float* AtW, long ldatw, const float* G,
                     float eps, int clamp, void* stream);
/* Fused W phase for p_c == 1 (no exchange between the GEMM and the update):
 * W *= (A H^T) / (W G + eps) in one pass over A  (dist_nmf.py:716-732) */
int dnmf_aht_update_w(const float* A, long m, long n, long lda, const float* H, int k, long ldh,
                      const float* G, float* W, long ldw, float eps, void* stream);

/* ---- One whole local MU/Frobenius step on one rank (dist_nmf.py:755-771 with p_r = p_c = 1),
 * W first then H with the new W; clamp!=0 applies pyDNMF.py:155-157 after the step. ---- */
/* k <= 32 on fp32 A with n a multiple of 4 up to 4096 (8192 at k <= 16; by default from 2048), m >= 4096, 16-byte aligned rows (and w_update != 0): the step reads A
 * ONCE (csrc/dnmf_team.h): teams of ceil(n / 512) workgroups share 16-row slabs by columns, exchange their 16 x k partials of A H^T inside the
 * kernel, update the slab's W rows and add W_new^T A from the LDS copy of the slab -- dist_nmf.py:729-732 feeding :748-751 without a
 * second pass.  Same update rule; the sums over n run in another association than the two-pass kernels' (results agree to fp32
 * rounding, not bit for bit; runs of the same shape are bit-identical).  All its workgroups (one per CU) must be resident together: the
 * launch starts with a census and writes nothing unless every workgroup has been seen (a shared GPU: dnmf_hals_sweep_status reports the
 * time-out, W is untouched by that launch).  dnmf_mu_fro_onepass(m, n, k) != 0: steps of this shape take it on this device;
 * dnmf_set_onepass(0) switches it off process-wide (returns the previous setting). */
/* Process-wide switch of every kernel whose workgroups wait for each other (the whole fits of small problems, the persistent HALS W sweep
 * -- local and across ranks --, the one-pass MU/FRO step): on (default) / off = their launch-chain forms everywhere (same update rules, no
 * co-residency needed: for a GPU shared with another process or stream).  Returns the previous setting.  PyNMF switches it off for the
 * rest of the process, and fits again from the initial factors, when dnmf_hals_sweep_status reports a time-out. */
int dnmf_set_persistent(int on);
int dnmf_mu_fro_onepass(long m, long n, int k);
int dnmf_set_onepass(int on);
int dnmf_mu_fro_step(const float* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh,
                     int k, float eps, int w_update, int clamp, void* ws, size_t ws_bytes, void* stream);

/* ---- HALS / Frobenius sweeps (dist_nmf.py:873-934; 2D :411-470).  They reuse dnmf_gram_*, dnmf_aht, dnmf_wta. ---- */
/* One column of the W sweep (dist_nmf.py:886-887): first applies the pending normalisation of column kk-1
 * (W[:,kk-1] /= sqrt(*prev_ss2), skipped if prev_ss2 is NULL or 0), then
 * W[:,kk] = max(W[:,kk]*G[kk][kk] + AH[:,kk] - W G[:,kk], eps) and *ss2_out = sum_i W[i][kk]^2 (device double).
 * With p_r > 1 the caller allreduces *ss2_out between calls (utils.py:390 inside `norm`, dist_nmf.py:889). */
int dnmf_hals_w_col(float* W, long m, int k, long ldw, const float* AH, long ldah, const float* G, int kk,
                    const double* prev_ss2, float eps, double* ss2_out, void* stream);
/* W[:,col] /= sqrt(*ss2) if > 0 (dist_nmf.py:890-891, the normalisation of the last column) */
int dnmf_hals_w_scale(float* W, long m, long ldw, int col, const double* ss2, void* stream);
/* the whole W sweep on one rank (no allreduce of the norms): k column launches + the final scale; ss2 = k doubles */
int dnmf_hals_update_w(float* W, long m, int k, long ldw, const float* AH, long ldah, const float* G, float eps,
                       double* ss2, void* stream);
/* The same W sweep in ONE launch when the rank's rows fit on the device at once (one lane per row for the whole sweep:
 * W and AH are read once, W written once; grid-wide column norms through per-workgroup fp64 slots reduced in a fixed
 * order -- bitwise reproducible); otherwise it runs dnmf_hals_update_w.  `ws` >= dnmf_ws_bytes(m, k, k). dist_nmf.py:884-891
 * The persistent kernel's workgroups wait for each other: do not run two of these sweeps concurrently on one device
 * (two streams, or two processes sharing a GPU with factors of tens of thousands of rows) -- each could hold the
 * resources the other's remaining workgroups need.  One process per GPU and one stream, the product's model, is safe. */
int dnmf_hals_sweep_w(float* W, long m, int k, long ldw, const float* AH, long ldah, const float* G, float eps,
                      void* ws, size_t ws_bytes, void* stream);
/* Did a persistent W sweep on the current device give up waiting for its other workgroups since the last call?  (Its
 * workgroups were not co-resident -- see above; the sweep then ends after about a second with NaN column norms instead of
 * hanging the GPU, and sets a sticky per-device word.)  *timed_out = 0 / 1; the word is cleared.  This call SYNCHRONISES
 * `stream` (the one entry point that does): call it where the host waits anyway, e.g. before the factors leave the GPU. */
int dnmf_hals_sweep_status(int* timed_out, void* stream);
/* H sweep: for kk: H[kk,:] = max(H[kk,:] + AtW[kk,:] - G[kk,:] H, eps), rows updated in sequence (dist_nmf.py:905-909) */
int dnmf_hals_update_h(float* H, int k, long n, long ldh, const float* AtW, long ldatw, const float* G, float eps,
                       void* stream);

/* ---- KL pieces (dist_nmf.py:776-869; 2D :294-343).  U = A / (W H + eps) is never materialised. ---- */
/* UHT[m x k] = (A / (W H + eps)) H^T     (glob_UX(axis=0), dist_nmf.py:806,810; UHT_glob :337-338) */
int dnmf_kl_uht(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh,
                int k, float eps, float* UHT, long ldo, void* ws, size_t ws_bytes, void* stream);
/* The same product with H given as n / nh COLUMN BLOCKS stacked [q][k][nh] -- the receive buffer of the allgather of the
 * ranks' k x nh slices (gather_W_H, dist_nmf.py:283-287; np.hstack there) -- so that the 2D step needs no re-assembly copy.
 * n % nh == 0 and nh % 32 == 0 (DNMF_EINVAL otherwise: assemble H and call dnmf_kl_uht). */
int dnmf_kl_uht_hblocks(const float* A, long m, long n, long lda, const float* W, long ldw, const float* Hs, long nh,
                        int k, float eps, float* UHT, long ldo, void* ws, size_t ws_bytes, void* stream);
/* scratch bytes for dnmf_kl_uht_hblocks (one partial slab per column split, and a split never straddles a block: narrow
 * blocks need more slabs than dnmf_ws_bytes reserves); >= dnmf_ws_bytes(m, n, k) */
size_t dnmf_ws_bytes_hblocks(long m, long n, int k, long nh);
/* WTU[k x n] = W^T (A / (W H + eps))     (glob_UX(axis=1), dist_nmf.py:806,808; WTU_glob :311-312) */
int dnmf_kl_wtu(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh,
                int k, float eps, float* WTU, long ldo, void* ws, size_t ws_bytes, void* stream);
/* x[k] = row sums of H [k x n]           (sum_along_axis(H, axis=1), dist_nmf.py:793-795) */
int dnmf_rowsum(const float* H, int k, long n, long ldh, float* x, void* stream);
/* x[k] = column sums of W [m x k]        (sum_along_axis(W, axis=0), dist_nmf.py:793; pyDNMF.py:187) */
int dnmf_colsum(const float* W, long m, int k, long ldw, float* x, void* ws, size_t ws_bytes, void* stream);
/* W[i][j] *= S[i][j] / (x[j] + eps)      (dist_nmf.py:828-830) */
int dnmf_kl_update_w(float* W, long m, int k, long ldw, const float* S, long lds_, const float* x, float eps,
                     void* stream);
/* H[j][c] *= S[j][c] / (x[j] + eps); clamp as above (dist_nmf.py:847-849) */
int dnmf_kl_update_h(float* H, int k, long n, long ldh, const float* S, long lds_, const float* x, float eps,
                     int clamp, void* stream);
/* One whole local MU/KL step on one rank (dist_nmf.py:851-869 with p_r = p_c = 1) */
int dnmf_mu_kl_step(const float* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh,
                    int k, float eps, int w_update, int clamp, void* ws, size_t ws_bytes, void* stream);

/* ---- PyNMF.fit helpers (pyDNMF.py:155-157, 185-194, 205-218) ---- */
/* X = max(X, eps) over a rows x cols matrix */
int dnmf_clamp_min(float* X, long rows, long cols, long ldx, float eps, void* stream);
/* W[i][j] /= (s[j] + eps) (pyDNMF.py:192) */
int dnmf_scale_cols_div(float* W, long m, int k, long ldw, const float* s, float eps, void* stream);
/* H[j][c] *= s[j] (pyDNMF.py:193) */
int dnmf_scale_rows_mul(float* H, int k, long n, long ldh, const float* s, void* stream);
/* *out (double, device) = sum A^2  (np.linalg.norm(A)**2, pyDNMF.py:208,215) */
int dnmf_sqnorm(const float* A, long m, long n, long lda, double* out, void* stream);
/* *out (double, device) = sum (A - W H)^2 without materialising the residual (pyDNMF.py:207,215) */
int dnmf_resid_sqnorm(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh,
                      int k, double* out, void* stream);
/* the same with a workspace (dnmf_ws_bytes(m, n, k) is enough; NULL = dnmf_resid_sqnorm): ranks that are not a whole number of
 * 32-wide tiles (every k an NMFk sweep visits) are evaluated on zero-padded factor images in it by the LDS-staged kernel */
int dnmf_resid_sqnorm_ws(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh,
                         int k, double* out, void* ws, size_t ws_bytes, void* stream);
/* per-column pieces of PyNMF.column_err (pyDNMF.py:221-239) over this rank's rows: num[c] += sum_i (A - W H)[i][c]^2,
 * den[c] += sum_i A[i][c]^2 (device doubles, n each, ACCUMULATED: the caller zeroes them; A - W H is never materialised) */
int dnmf_column_err(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh,
                    int k, double* num, double* den, void* stream);

/* NMFk perturbation (pyDNMFk.py:42-44, sample.randM): X_per = X * (1 + noise_var + 2 noise_var U), U ~ U[0,1) per element from a
 * counter-based generator keyed by `seed` and the element's position (stateless: the same (seed, position) gives the same value).
 * One pass; bf16 != 0: X and X_per are bfloat16 (scaled in fp32, rounded once).  Any shape and alignment (16-byte aligned rows of
 * whole 8-element vectors take a vector kernel, everything else one element per thread -- the values are the same either way). */
int dnmf_perturb_uniform(const void* X, void* X_per, long rows, long cols, long ldx, long ldo, float noise_var,
                         unsigned long long seed, int bf16, void* stream);

/* ---- bf16 STORAGE of the data matrix (BASELINE config 5, "mixed precision"; no reference counterpart: numpy has no
 * bf16).  A is bfloat16 in device memory (pointer to 16-bit words, lda in elements, rows 8-byte aligned for the vector
 * path), widened exactly to fp32 in registers; W, H, every product and every accumulation stay fp32, so each call equals
 * its fp32 twin applied to float(A).  Same arguments otherwise. ---- */
int dnmf_aht_bf16a(const void* A, long m, long n, long lda, const float* H, int k, long ldh,
                   float* AH, long ldah, void* stream);
int dnmf_wta_bf16a(const void* A, long m, long n, long lda, const float* W, int k, long ldw,
                   float* AtW, long ldatw, void* ws, size_t ws_bytes, void* stream);
int dnmf_wta_gram_bf16a(const void* A, long m, long n, long lda, const float* W, int k, long ldw, float* AtW, long ldatw,
                        float* G, void* ws, size_t ws_bytes, void* stream);
int dnmf_aht_update_w_bf16a(const void* A, long m, long n, long lda, const float* H, int k, long ldh,
                            const float* G, float* W, long ldw, float eps, void* stream);
int dnmf_mu_fro_step_bf16a(const void* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh,
                           int k, float eps, int w_update, int clamp, void* ws, size_t ws_bytes, void* stream);
int dnmf_sqnorm_bf16a(const void* A, long m, long n, long lda, double* out, void* stream);
int dnmf_resid_sqnorm_bf16a(const void* A, long m, long n, long lda, const float* W, long ldw, const float* H,
                            long ldh, int k, double* out, void* stream);
int dnmf_resid_sqnorm_ws_bf16a(const void* A, long m, long n, long lda, const float* W, long ldw, const float* H,
                               long ldh, int k, double* out, void* ws, size_t ws_bytes, void* stream);
int dnmf_column_err_bf16a(const void* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh,
                          int k, double* num, double* den, void* stream);

/* ---- bf16x6: the two big contractions on the bf16 matrix cores with fp32-grade products (no reference counterpart; an
 * alternative arithmetic for global_mm, dist_nmf.py:705).  Every fp32 operand is cut into three bf16 pieces (x = x1 + x2 + x3
 * to 2^-24 |x|) and a product is the sum of the six largest piece products, each exact in the fp32 accumulator; the dropped
 * terms are below one fp32 rounding of the product (csrc/dnmf_split.h has the bound, tests/test_gpu_split.py the measurement
 * against float64).  Opt-in: the fp32-MFMA entry points above stay the default and the reference for parity.
 * Kernels exist for 32 < k <= 128, 16-byte aligned rows of A and n % 128 == 0; any other shape is forwarded to the fp32
 * entry point of the same name.  Same arguments as the fp32 twins plus a workspace of dnmf_ws_bytes_bf16x6(m, n, k) bytes
 * (the bf16 images of H and W^T, partial sums). ---- */
size_t dnmf_ws_bytes_bf16x6(long m, long n, int k);
int dnmf_aht_bf16x6(const float* A, long m, long n, long lda, const float* H, int k, long ldh,
                    float* AH, long ldah, void* ws, size_t ws_bytes, void* stream);
int dnmf_wta_bf16x6(const float* A, long m, long n, long lda, const float* W, int k, long ldw,
                    float* AtW, long ldatw, void* ws, size_t ws_bytes, void* stream);
int dnmf_aht_update_w_bf16x6(const float* A, long m, long n, long lda, const float* H, int k, long ldh,
                             const float* G, float* W, long ldw, float eps, void* ws, size_t ws_bytes, void* stream);
int dnmf_mu_fro_step_bf16x6(const float* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh,
                            int k, float eps, int w_update, int clamp, void* ws, size_t ws_bytes, void* stream);
/* The same four with A STORED as bfloat16 (dnmf_*_bf16a above): A is then its own single piece, a product is the sum of three
 * bf16 piece products (A times the three pieces of the fp32 factor), each exact in the fp32 accumulator -- the result equals
 * the fp32-MFMA twin on float(A) to fp32 rounding, at half the HBM bytes and a quarter of the matrix work of the fp32 A case.
 * Kernels for 16 < k <= 128 (k <= 16: the 16-wide fp32 kernels of the bf16a entry points are as fast). */
int dnmf_aht_bf16a_bf16x6(const void* A, long m, long n, long lda, const float* H, int k, long ldh,
                          float* AH, long ldah, void* ws, size_t ws_bytes, void* stream);
int dnmf_wta_bf16a_bf16x6(const void* A, long m, long n, long lda, const float* W, int k, long ldw,
                          float* AtW, long ldatw, void* ws, size_t ws_bytes, void* stream);
int dnmf_aht_update_w_bf16a_bf16x6(const void* A, long m, long n, long lda, const float* H, int k, long ldh,
                                   const float* G, float* W, long ldw, float eps, void* ws, size_t ws_bytes, void* stream);
int dnmf_mu_fro_step_bf16a_bf16x6(const void* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh,
                                  int k, float eps, int w_update, int clamp, void* ws, size_t ws_bytes, void* stream);
/* The two KL products (dist_nmf.py:806-810) in the same arithmetic: S = W H, U H^T and W^T U as bf16 piece products, U = A / (S
 * + eps) in fp32 between them.  Kernels for every k <= 128 with n % 128 == 0 and 16-byte aligned rows of A; other shapes are forwarded to
 * dnmf_kl_uht / dnmf_kl_wtu / dnmf_mu_kl_step.  Workspace: dnmf_ws_bytes_bf16x6(m, n, k). */
int dnmf_kl_uht_bf16x6(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh,
                       int k, float eps, float* UHT, long ldo, void* ws, size_t ws_bytes, void* stream);
int dnmf_kl_wtu_bf16x6(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh,
                       int k, float eps, float* WTU, long ldo, void* ws, size_t ws_bytes, void* stream);
int dnmf_mu_kl_step_bf16x6(const float* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh,
                           int k, float eps, int w_update, int clamp, void* ws, size_t ws_bytes, void* stream);

/* ---- Whole fits on one rank (PyNMF.fit, pyDNMF.py:138-182 with p_r = p_c = 1): `itr` update steps with the clamp to eps
 * after the steps i % 10 == 0 (:155-157, :170-172), then normalize_features (:185-194) and the two squared norms of
 * relative_err (:205-218) -- everything enqueued on `stream` by ONE call: no host code between the launches, nothing
 * synchronises.  sq_out (device doubles) receives {sum (A - W H)^2, sum A^2} per problem; recon_err = sqrt of their ratio.
 * `batch` > 1: that many INDEPENDENT problems of one shape in every launch (blockIdx.z = problem) -- the perturbation fits of an
 * NMFk sweep (pyDNMFk.py:226-231), which the reference runs one after another.  Problem b's operands live at A + b a_stride,
 * W + b w_stride, H + b h_stride (strides in ELEMENTS, multiples of 16 bytes, each spanning at least one problem: e.g. stacked
 * [batch][m][lda] arrays) and its scratch at ws + b (ws_bytes_fit / batch); sq_out is [batch][2].  Problem b of a batched fit
 * runs the same kernels on the same operands as a fit of its own: results are bit-identical to `batch` single fits (strides
 * are ignored for batch == 1).  `ws` >= dnmf_ws_bytes_fit(m, n, k, batch).  The persistent HALS W sweep runs on as many problems
 * at a time as the device holds resident (a problem that does not fit on its own takes the column launches); `column_sweep` != 0 forces those.
 * After a HALS fit dnmf_hals_sweep_status tells whether a persistent sweep timed out.
 * SMALL MU problems (MU/KL on fp32 A; MU/FRO on fp32 or bf16-stored A; k <= 32, a 128-row slab of A -- in LDS or streamed from the L2 -- or a 64-row slab, all of H
 * and the slab's rows of W in the 160 KiB of LDS of a CU -- n up to ~2400 at k <= 16, ~1200 beyond -- and at most 64 slabs (m <= 8192): the reference's example sizes, swim 1024
 * x 256, wtsi 96 x 21) run the whole loop as ONE persistent kernel per batch (csrc/dnmf_small.h): a workgroup per slab keeps its data in
 * LDS across the steps, the problem's workgroups meet at ONE barrier per step and pass the new H as {value, step} granules (the waits of both are bounded the same way).  Same update rules, fp32 sums in another association
 * than the step kernels: results agree with dnmf_mu_{kl,fro}_step to fp32 rounding (not bit for bit); a batched fit still equals
 * `batch` single fits bit for bit.  All workgroups of a launch must be resident together (the library splits a batch into as many
 * launches as that takes; do not share the GPU with another stream meanwhile); a barrier that waits longer than 2 s gives up and
 * dnmf_hals_sweep_status reports it.  dnmf_mu_fit_persistent(m, n, k) != 0: fits of this shape take that kernel. */
size_t dnmf_ws_bytes_fit(long m, long n, int k, int batch);
int dnmf_mu_fit_persistent(long m, long n, int k);
/* != 0: dnmf_hals_fro_fit[_bf16a] with w_update != 0 and column_sweep == 0 runs fits of this shape on the persistent kernel too (A streamed
 * from the L2, fp32 or bf16-stored; the k column norms of a W sweep cross the problem's workgroups through value-as-flag slots) */
int dnmf_hals_fit_persistent(long m, long n, int k);
/* seconds a barrier of the persistent small fit may wait before it gives up (default 2; process-wide, read at the next fit call) */
int dnmf_fit_set_timeout(double seconds);
int dnmf_mu_fro_fit(const float* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k, float eps, int w_update,
                    int itr, int batch, long a_stride, long w_stride, long h_stride, double* sq_out, void* ws, size_t ws_bytes,
                    void* stream);
int dnmf_mu_kl_fit(const float* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k, float eps, int w_update,
                   int itr, int batch, long a_stride, long w_stride, long h_stride, double* sq_out, void* ws, size_t ws_bytes,
                   void* stream);
int dnmf_hals_fro_fit(const float* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k, float eps,
                      int w_update, int itr, int column_sweep, int batch, long a_stride, long w_stride, long h_stride, double* sq_out,
                      void* ws, size_t ws_bytes, void* stream);
/* the Frobenius fits with A STORED as bfloat16 (dnmf_*_bf16a above; a_stride in bf16 elements) */
int dnmf_mu_fro_fit_bf16a(const void* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k, float eps,
                          int w_update, int itr, int batch, long a_stride, long w_stride, long h_stride, double* sq_out, void* ws,
                          size_t ws_bytes, void* stream);
int dnmf_hals_fro_fit_bf16a(const void* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k, float eps,
                            int w_update, int itr, int column_sweep, int batch, long a_stride, long w_stride, long h_stride,
                            double* sq_out, void* ws, size_t ws_bytes, void* stream);

/* ---- The update path in float64.  The reference computes in the dtype of A_ij (pyDNMF.py:68; its own tests feed float64,
 * tests/test_dist_nmf_1d.py:14-20): these are the float64 twins of the primitives above, on the fp64 matrix cores
 * (v_mfma_f64_16x16x4_f64), one tile shape each (0.61 / 0.72 of the fp64 MFMA peak per MU/FRO / MU/KL step at 65536 x 4096, k = 64).  Everything is `double`
 * in device memory, row-major, leading dimensions in elements; eps = 2.220446049250313e-16; k <= DNMF_TUNED_MAX_K; Gram matrices are
 * plain k x k blocks with their own leading dimension `ldg` (no padding contract).  `ws` >= dnmf_f64_ws_bytes(m, n, k) where an
 * entry point takes one.  The KL products (dnmf_f64_kl_uht / dnmf_f64_kl_wtu) keep the quotient U = A / (W H + eps) in registers up to
 * k = 64 and go through the materialised image beyond (dnmf_f64_kl_quot into an m x n buffer of the caller -- the reference materialises
 * it too, dist_nmf.py:806 -- followed by dnmf_f64_aht / dnmf_f64_wta on U); the error evaluation through the materialised squared residual (dnmf_f64_sqdiff) and the ordered sums.  Host sequencing of these
 * primitives (1D and 2D grids, exchanges over torch.distributed): pydnmfk_amd/dist_nmf.py with engine.HipOpsF64. ---- */
size_t dnmf_f64_ws_bytes(long m, long n, int k);
/* A whole float64 fit on one rank (PyNMF.fit, pyDNMF.py:138-182): `itr` steps of method 0 = MU/FRO (dist_nmf.py:716-751), 1 = MU/KL
 * (:806-849), 2 = HALS/FRO (:873-934) with the clamp after the steps i % 10 == 0, then normalize_features and the two squared norms
 * {sum (A - W H)^2, sum A^2} -> sq_out (device) -- the primitives below in the order pydnmfk_amd/dist_nmf.py issues them, enqueued by
 * ONE call (bit-identical to that step loop).  ws >= dnmf_f64_ws_bytes_fit(m, n, k): it holds the m x n quotient / residual image too. */
size_t dnmf_f64_ws_bytes_fit(long m, long n, int k);
/* != 0: a fit of this shape and method runs as ONE single-workgroup launch with A, W, H in LDS for all its steps (csrc/dnmf_f64_tiny.hip:
 * k <= 16 and everything within 160 KiB of LDS -- the reference's own test sizes, 24 x 12 with k = 2 over 2000 iterations,
 * tests/test_dist_nmf_1d.py:14-46): same update rules in plain float64 FMA chains (index-order sums: a few ulp from the primitives'
 * MFMA sums, so such a fit equals the step loop to ~1e-13, not bit for bit); no workgroup waits for another one (nothing to keep resident) */
int dnmf_f64_fit_tiny(long m, long n, int k, int method);
int dnmf_f64_fit(int method, const double* A, long m, long n, long lda, double* W, long ldw, double* H, long ldh, int k, double eps,
                 int w_update, int itr, double* sq_out, void* ws, size_t ws_bytes, void* stream);
/* C[m x kc] = X[m x n] Y[kc x n]^T   (A H^T: dist_nmf.py:730; H H^T = global_gram(H.T), :729, with X = Y = H) */
int dnmf_f64_aht(const double* X, long m, long n, long ldx, const double* Y, int kc, long ldy, double* C, long ldc, void* ws,
                 size_t ws_bytes, void* stream);
/* C[kc x n] = W[m x kc]^T A[m x n]   (W^T A: dist_nmf.py:749; W^T W = global_gram(W), :748, with A = W) */
int dnmf_f64_wta(const double* A, long m, long n, long lda, const double* W, int kc, long ldw, double* C, long ldc, void* ws,
                 size_t ws_bytes, void* stream);
/* W *= AH / (W G + eps)   (dist_nmf.py:731-732, :244-245) */
int dnmf_f64_mu_update_w(double* W, long m, int k, long ldw, const double* AH, long ldah, const double* G, long ldg, double eps,
                         void* stream);
/* H *= AtW / (G H + eps); clamp != 0: H = max(H, eps) afterwards   (dist_nmf.py:750-751, :224-225; pyDNMF.py:156) */
int dnmf_f64_mu_update_h(double* H, int k, long n, long ldh, const double* AtW, long ldatw, const double* G, long ldg, double eps,
                         int clamp, void* stream);
/* U[m x n] = A / (W H + eps)   (dist_nmf.py:806) */
int dnmf_f64_kl_quot(const double* A, long m, long n, long lda, const double* W, long ldw, const double* H, long ldh, int k, double eps,
                     double* U, long ldu, void* stream);
/* The two KL products without the image: S[m x k] = (A / (W H + eps)) H^T (dist_nmf.py:806, :810) and S[k x n] = W^T (A / (W H + eps))
 * (:806, :808).  k <= 64: ONE pass over A each -- a wave forms a 16 x 16 tile of W H on the matrix cores, divides in registers and feeds
 * the quotient to the second product as an MFMA operand (csrc/dnmf_f64_kl.h); U may be NULL.  k > 64: dnmf_f64_kl_quot into U (m x n,
 * leading dimension n; DNMF_EINVAL if NULL) followed by dnmf_f64_aht / dnmf_f64_wta.  ws >= dnmf_f64_ws_bytes(m, n, k). */
int dnmf_f64_kl_uht(const double* A, long m, long n, long lda, const double* W, long ldw, const double* H, long ldh, int k, double eps,
                    double* S, long lds_, double* U, void* ws, size_t ws_bytes, void* stream);
int dnmf_f64_kl_wtu(const double* A, long m, long n, long lda, const double* W, long ldw, const double* H, long ldh, int k, double eps,
                    double* S, long lds_, double* U, void* ws, size_t ws_bytes, void* stream);
/* R[m x n] = (A - W H)^2 element-wise   (pyDNMF.py:207, :229) */
int dnmf_f64_sqdiff(const double* A, long m, long n, long lda, const double* W, long ldw, const double* H, long ldh, int k, double* R,
                    long ldr, void* stream);
/* *out = sum X (sq == 0) or sum X^2 (sq != 0), fixed summation order   (np.linalg.norm(.)**2, pyDNMF.py:208,215) */
int dnmf_f64_sum(const double* X, long rows, long cols, long ldx, int sq, double* out, void* ws, size_t ws_bytes, void* stream);
/* x[c] = sum_r X[r][c] (or of squares)   (sum_along_axis(W, axis=0), dist_nmf.py:793; the column sums of column_err, pyDNMF.py:229) */
int dnmf_f64_colsum(const double* X, long m, long n, long ldx, int sq, double* x, void* ws, size_t ws_bytes, void* stream);
/* x[r] = sum_c H[r][c]   (sum_along_axis(H, axis=1), dist_nmf.py:793-795) */
int dnmf_f64_rowsum(const double* H, int k, long n, long ldh, double* x, void* stream);
/* element-wise passes.  op 0: X = max(X, eps) (pyDNMF.py:156); 1: X[r][c] /= x[c] + eps (:192); 2: X[r][c] *= x[r] (:193);
 * 3: X[r][c] *= S[r][c] / (x[r] + eps) (dist_nmf.py:847-849); 4: X[r][c] *= S[r][c] / (x[c] + eps) (:828-830); clamp != 0 (ops 3, 4):
 * max(., eps) afterwards */
int dnmf_f64_ew(int op, double* X, long rows, long cols, long ldx, const double* S, long lds_, const double* x, double eps, int clamp,
                void* stream);
/* HALS (dist_nmf.py:873-934): one column of the W sweep (as dnmf_hals_w_col; *ss2_out = this rank's sum of squares of the new
 * column, WRITTEN, not accumulated), the final scale, the H sweep */
int dnmf_f64_hals_w_col(double* W, long m, int k, long ldw, const double* AH, long ldah, const double* G, long ldg, int kk,
                        const double* prev_ss2, double eps, double* ss2_out, void* ws, size_t ws_bytes, void* stream);
int dnmf_f64_hals_w_scale(double* W, long m, long ldw, int col, const double* ss2, void* stream);
int dnmf_f64_hals_update_h(double* H, int k, long n, long ldh, const double* AtW, long ldatw, const double* G, long ldg, double eps,
                           void* stream);

/* ---- Grid exchanges inside the library (RCCL over xGMI; replaces MPI_comm, dist_comm.py:16-56, and the mpi4py calls of
 * global_gram / global_mm, dist_nmf.py:681,707).  RCCL is bound at run time (dlopen; an RCCL already in the process -- the
 * PyTorch host's -- is preferred), so the library loads without it; these entry points then return DNMF_ECOMM.
 * One communicator handle per rank (= per process = per GPU; the current HIP device at dnmf_comm_create is the rank's GPU).
 * Rank r sits at grid coordinates (i, j) = (r / p_c, r % p_c), the reference's Create_cart(reorder=False). ---- */
#define DNMF_UNIQUE_ID_BYTES 128
typedef struct dnmf_comm dnmf_comm_t;
/* rank 0: fill 128 bytes that every rank must pass to dnmf_comm_create (the host broadcasts them: mpi4py bcast,
 * torch.distributed broadcast_object_list, a file ...) */
int dnmf_comm_unique_id(void* id_out);
/* collective over all nranks processes; p_r * p_c == nranks; 2D grids also get the two sub-communicators
 * (cart_1d_row = the p_r ranks of one grid column, cart_1d_column = the p_c ranks of one grid row, dist_comm.py:25-51) */
int dnmf_comm_create(const void* unique_id, int nranks, int rank, int p_r, int p_c, dnmf_comm_t** out);
/* A communicator whose collectives the HOST performs (no RCCL in the process, or a transport of the host's own -- a GPU-aware
 * MPI under the reference's mpi4py driver; the multi-rank tests of this repository drive it with gloo):
 * every exchange of the step entry points calls fn(user, op, group, send, recv, count, stream) on the calling thread.
 *   op    DNMF_ALLREDUCE       recv == send: in-place SUM of `count` floats over the group (dist_nmf.py:114, :681, :707);
 *         DNMF_ALLREDUCE_F64 the same on doubles (the HALS steps only)
 *         DNMF_ALLGATHER       recv[q * count ...] = member q's `count` floats, members in group order (:163-165, :195-197)
 *         DNMF_REDUCE_SCATTER  send = members x count floats; recv = this member's block of the SUM (:169-171, :202)
 *   group 0 = all ranks, 1 = cart_1d_row (the p_r ranks of a grid column, ordered by grid row), 2 = cart_1d_column (the p_c
 *         ranks of a grid row, ordered by grid column) -- dist_comm.py:16-56; groups of one member are never passed
 *   send / recv are DEVICE pointers; everything enqueued on `stream` before the call produces `send`, everything enqueued
 *   after it consumes `recv`: the function either enqueues its transfer on `stream` or synchronises it, moves the data and
 *   returns when `recv` is complete.  Return 0, or non-zero to fail the step (DNMF_ECOMM). */
#define DNMF_ALLREDUCE 0
#define DNMF_ALLGATHER 1
#define DNMF_REDUCE_SCATTER 2
#define DNMF_ALLREDUCE_F64 3   /* as DNMF_ALLREDUCE on `count` DOUBLES (send / recv point to doubles): the HALS column norms */
/* send / recv are untyped: the element type is the op's (float32 for ops 0-2, float64 for DNMF_ALLREDUCE_F64) */
typedef int (*dnmf_collective_fn)(void* user, int op, int group, const void* send, void* recv, size_t count, void* stream);
int dnmf_comm_create_hosted(int nranks, int rank, int p_r, int p_c, dnmf_collective_fn fn, void* user, dnmf_comm_t** comm);
/* MEASUREMENT aid (bench.py --emulate-ranks, a one-GPU box): member `rank` of a p_r x p_c grid whose collectives are all issued
 * for real on a ONE-rank RCCL communicator -- launch and stream-ordering costs without a wire, buffers of the real grid's sizes.
 * The step entry points run as on the grid; the numbers they produce are not a factorisation of anything. */
int dnmf_comm_create_emulated(int p_r, int p_c, int rank, dnmf_comm_t** comm);
/* the RCCL this library bound at run time: *version = ncclGetVersion's code (e.g. 22604; 0 if the symbol is absent), `origin` =
 * how it was found ("already in the process" = the host framework's copy, or the library name that was loaded).  For run records. */
int dnmf_comm_rccl_version(int* version, char* origin, size_t origin_bytes);
/* ---- direct (two-shot) allreduce over IPC peer buffers -- an alternative to RCCL's ring for the small packed message of a 1D
 * step (2 MiB at BASELINE config 3; SURVEY section 5 "measure both").  Every rank exports a region of device memory, the host
 * moves the 64-byte handles (allgather over whatever it has), every rank maps its peers' regions.  One node, at most
 * DNMF_DIRECT_MAX_RANKS ranks, messages of an even number of floats.  Rank-ordered sums with one owner per element: every rank
 * ends with identical bits.  Works on RCCL and on hosted communicators alike (it needs neither). ---- */
#define DNMF_DIRECT_MAX_RANKS 16
#define DNMF_DIRECT_HANDLE_BYTES 80
/* allocate this rank's region (uncached device memory; DNMF_EHIP when the device has none to give -- there is no cached
 * fall-back) for messages of up to max_floats floats; handle_out receives DNMF_DIRECT_HANDLE_BYTES bytes (the IPC handle and the
 * region's capacity).  EVERY rank must pass the same max_floats: peers address each other's regions with one capacity. */
int dnmf_comm_direct_init(dnmf_comm_t* comm, size_t max_floats, void* handle_out);
/* handles = nranks x DNMF_DIRECT_HANDLE_BYTES bytes in rank order; DNMF_EINVAL when any rank's capacity differs from this rank's
 * (every rank sees every entry, so a mismatch fails the call on all of them) */
int dnmf_comm_direct_connect(dnmf_comm_t* comm, const void* handles);
/* undo init / connect (a set-up that failed on some rank): device synchronise, unmap the peers, free the own region.  The host
 * makes sure no peer is still reading this rank's region (a barrier); dnmf_comm_destroy does the same unmapping */
int dnmf_comm_direct_teardown(dnmf_comm_t* comm);
/* how long a wait of a direct allreduce may see no progress before it gives up (default 30 s; rank skew of seconds is ordinary:
 * result I/O on rank 0, first-call module loads) */
int dnmf_comm_set_direct_timeout(dnmf_comm_t* comm, double seconds);
/* Call on EVERY rank before the first step of a fit on new data (PyNMF.fit does): the next HALS W sweep over the ranks then agrees anew
 * on whether the one-launch cross-rank sweep applies (a host-synchronous 16-float allreduce).  The ranks re-open that agreement only at
 * events all of them see -- this call, another k, dnmf_set_persistent -- because a rank's local row count may change on some ranks only;
 * a step whose local row count differs from the agreed one without this call returns DNMF_EINVAL. */
int dnmf_comm_fit_begin(dnmf_comm_t* comm);
/* on != 0: allreduces over ALL ranks that fit the regions (the packed exchange of the 1D steps, dnmf_comm_allreduce with
 * group 0) take the direct path; everything else stays on RCCL / the hosted function */
int dnmf_comm_set_direct(dnmf_comm_t* comm, int on);
/* With the direct path on, the W sweep of the HALS steps on grids whose W rows are spread over ranks (p_r > 1; every 2D grid) is ONE
 * persistent launch instead of k column launches + k allreduces: every workgroup publishes its column partial into its slot of every
 * rank's slab (two 1 MiB slabs by sweep parity live in the exported regions), polls its own rank's slab and sums all ranks' slots in
 * slot order -- identical bits on every rank (csrc/dnmf_hals.h, HalsPeers).  Taken when EVERY rank can keep its rows resident (the
 * ranks agree once per shape through a 16-float allreduce; otherwise all of them keep the column launches).  A peer that stalls
 * longer than the direct time-out sets the sticky word dnmf_hals_sweep_status reports.  *count = such sweeps issued so far. */
int dnmf_comm_hals_xsweeps(dnmf_comm_t* comm, unsigned long long* count);
/* in-place SUM allreduce of `count` floats over all ranks through the peer regions, on `stream` */
int dnmf_comm_allreduce_direct(dnmf_comm_t* comm, float* buf, size_t count, void* stream);
/* the ONE-launch form for 1..8 doubles (the column norms of the HALS W sweep: k dependent 8-byte allreduces per iteration on
 * p_r > 1, pure latency): every rank pushes its values into its peers' regions, sums in rank order.  With dnmf_comm_set_direct
 * the HALS step entry points use it for their norm exchanges. */
int dnmf_comm_allreduce_direct_f64(dnmf_comm_t* comm, double* buf, size_t count, void* stream);
/* *timed_out != 0: a wait of a direct allreduce saw no progress for the time-out and gave up (a peer is gone): every result since
 * is INVALID.  Sticky.  Callers check it where they synchronise anyway -- PyNMF.fit at the end of a fit, all ranks agreeing on the
 * outcome before they raise -- and treat it as fatal. */
int dnmf_comm_direct_status(dnmf_comm_t* comm, int* timed_out);
int dnmf_comm_destroy(dnmf_comm_t* comm);
int dnmf_comm_info(const dnmf_comm_t* comm, int* nranks, int* rank, int* p_r, int* p_c);
/* column chunks of the overlapped H phase of dnmf_mu_fro_step_1d on a row grid (1 = one packed allreduce; <= 8) */
int dnmf_comm_set_overlap_chunks(dnmf_comm_t* comm, int chunks);
/* testing aid: with on != 0 a ONE-rank communicator still issues its RCCL calls (a 1 x 1 grid then behaves as a row grid),
 * so that the exchange path can be exercised on a single GPU */
int dnmf_comm_set_always_exchange(dnmf_comm_t* comm, int on);
/* measurement aid: with on != 0 the step entry points skip their RCCL calls (everything else -- kernels, events, the internal
 * stream -- as usual), so a rank's compute can be timed without the exchange; results are WRONG on more than one rank */
int dnmf_comm_set_null_exchange(dnmf_comm_t* comm, int on);
/* in-place SUM allreduce of `count` floats on `stream`; group 0 = all ranks, 1 = cart_1d_row, 2 = cart_1d_column
 * (comm.allreduce of dist_nmf.py:114,681,707 and the host's scalar reductions) */
int dnmf_comm_allreduce(dnmf_comm_t* comm, float* buf, size_t count, int group, void* stream);
/* workspace of the two 1D steps below for a rank holding an m_l x n_l block (kernel scratch + the packed exchange buffers) */
size_t dnmf_ws_bytes_1d(long m_l, long n_l, int k);
/* One whole MU / Frobenius step of one rank of a 1D grid, exchanges included (nmf_algorithms_1D.update, dist_nmf.py:755-771
 * with global_gram / global_mm :663-708): p_c == 1 (A, W row blocks; H replicated): fused W phase, then ONE allreduce of the
 * packed [W^T A | W^T W] message (or overlapped column chunks, see dnmf_comm_set_overlap_chunks); p_r == 1: the mirror
 * image with [A H^T | H H^T].  Everything is enqueued on `stream` (the chunked exchange on an internal stream, ordered by
 * events); the call returns without synchronising.  Same arithmetic as the per-kernel entry points called in that order. */
int dnmf_mu_fro_step_1d(const float* A, long m_l, long n_l, long lda, float* W, long ldw, float* H, long ldh, int k,
                        float eps, int w_update, int clamp, void* ws, size_t ws_bytes, dnmf_comm_t* comm, void* stream);
/* The same for MU / KL (dist_nmf.py:851-869 with sum_along_axis / glob_UX :776-811): [U H^T | rowsum H] is reduced when
 * p_r == 1, [W^T U | colsum W] when p_c == 1. */
int dnmf_mu_kl_step_1d(const float* A, long m_l, long n_l, long lda, float* W, long ldw, float* H, long ldh, int k,
                       float eps, int w_update, int clamp, void* ws, size_t ws_bytes, dnmf_comm_t* comm, void* stream);

/* The 2D grid (p_r > 1 and p_c > 1; nmf_algorithms_2D.update, dist_nmf.py:66-91).  The rank at grid position (i, j) = rank / p_c,
 * rank % p_c holds A_ij (m_l x n_l), the slice W_ij (m_w x k, contiguous: ldw == k) of its grid row's W_i and the slice H_ij
 * (k x n_h, contiguous: ldh == n_h) of its grid column's H_j; the slice sizes follow the reference's partition rule (utils.py:36-41:
 * of p members the first total % p hold one item more), m_w of m_l over the p_c ranks of the grid row, n_h of n_l over the p_r
 * ranks of the grid column -- anything else returns DNMF_EINVAL (pruned factors: the host keeps its choreography).  Exchanges:
 * allreduce of the Gram matrix / the factor sums over all ranks (:114, :346-349), allgather of H_ij over the p_r ranks of a grid
 * column and of W_ij over the p_c ranks of a grid row (:163-165, :195-197, :268-291), reduce-scatter of the products back to the
 * slices (:169-171, :202, :314-316, :340) -- all enqueued on `stream`, no host synchronisation.  Equal slices of whole
 * 32-column tiles: the kernels read the gathered H as column blocks and the H-phase product is formed slice by slice into the
 * reduce-scatter's send buffer (nothing is re-assembled); ragged slices (a dimension that does not divide: real data) are
 * exchanged at the pitch of the largest one, as MPI's Allgatherv / Reduce_scatter counts do, and H_j / W_i are assembled by
 * device copies. */
size_t dnmf_ws_bytes_2d(long m_l, long n_l, int k, int p_r, int p_c);
/* Fro_MU_update on the 2D grid (dist_nmf.py:207-263 with global_gram :107-117, AH_glob :186-205, ATW_glob :154-172) */
int dnmf_mu_fro_step_2d(const float* A, long m_l, long n_l, long lda, float* W, long m_w, long ldw, float* H, long n_h, long ldh,
                        int k, float eps, int w_update, int clamp, void* ws, size_t ws_bytes, dnmf_comm_t* comm, void* stream);
/* KL_MU_update on the 2D grid (dist_nmf.py:351-407 with sum_axis :346-349, gather_W_H :268-291, UHT_glob :330-343, WTU_glob :294-318) */
int dnmf_mu_kl_step_2d(const float* A, long m_l, long n_l, long lda, float* W, long m_w, long ldw, float* H, long n_h, long ldh,
                       int k, float eps, int w_update, int clamp, void* ws, size_t ws_bytes, dnmf_comm_t* comm, void* stream);

/* HALS / Frobenius steps with the exchanges inside (FRO_HALS_update, dist_nmf.py:873-934 on 1D grids, :411-470 on the 2D grid):
 * products and Gram matrices exchanged as in the MU steps above; the W sweep column by column -- where W's rows are spread over
 * ranks (p_r > 1, and every 2D grid) the 8-byte sum of squares of each column is allreduced between the column kernels
 * (utils.norm, utils.py:388-391); with local norms (1D, p_r == 1) it is dnmf_hals_sweep_w, or k column launches when
 * `column_sweep` != 0 -- then the H sweep; `clamp` != 0: H = max(H, eps), W = max(W, eps) afterwards (pyDNMF.py:170-172).
 * Workspaces: dnmf_ws_bytes_1d / dnmf_ws_bytes_2d. */
int dnmf_hals_fro_step_1d(const float* A, long m_l, long n_l, long lda, float* W, long ldw, float* H, long ldh, int k,
                          float eps, int w_update, int clamp, int column_sweep, void* ws, size_t ws_bytes, dnmf_comm_t* comm,
                          void* stream);
int dnmf_hals_fro_step_2d(const float* A, long m_l, long n_l, long lda, float* W, long m_w, long ldw, float* H, long n_h, long ldh,
                          int k, float eps, int w_update, int clamp, void* ws, size_t ws_bytes, dnmf_comm_t* comm, void* stream);

/* The Frobenius steps above with A STORED as bfloat16 (dnmf_*_bf16a: `A` points to 16-bit words, lda in elements; BASELINE config
 * 5's "mixed precision"): same arguments, same exchanges; on the 2D grid H_j is always assembled (the column-block form of A H^T
 * exists for fp32 A only). */
int dnmf_mu_fro_step_1d_bf16a(const void* A, long m_l, long n_l, long lda, float* W, long ldw, float* H, long ldh, int k,
                              float eps, int w_update, int clamp, void* ws, size_t ws_bytes, dnmf_comm_t* comm, void* stream);
int dnmf_mu_fro_step_2d_bf16a(const void* A, long m_l, long n_l, long lda, float* W, long m_w, long ldw, float* H, long n_h, long ldh,
                              int k, float eps, int w_update, int clamp, void* ws, size_t ws_bytes, dnmf_comm_t* comm, void* stream);
int dnmf_hals_fro_step_1d_bf16a(const void* A, long m_l, long n_l, long lda, float* W, long ldw, float* H, long ldh, int k,
                                float eps, int w_update, int clamp, int column_sweep, void* ws, size_t ws_bytes, dnmf_comm_t* comm,
                                void* stream);
int dnmf_hals_fro_step_2d_bf16a(const void* A, long m_l, long n_l, long lda, float* W, long m_w, long ldw, float* H, long n_h, long ldh,
                                int k, float eps, int w_update, int clamp, void* ws, size_t ws_bytes, dnmf_comm_t* comm, void* stream);

/* ---- measurement aid (no counterpart in the reference) ----
 * Which shader clock does the GPU hold right now?  Launches ONE wave on `stream` that writes `n` pairs {s_memtime (shader
 * cycles), wall_clock64 (the constant 100 MHz reference)} into samples[2 n], sleeping `naps` x ~4 us between two pairs, and
 * returns at once.  Launched on its own stream BEFORE the work of interest, it keeps one wave slot of one CU while that work
 * runs; (d memtime / d wall) x 0.1 between consecutive pairs is the clock in GHz over that interval.  The fp32 passes of a
 * step hold about 2.0 GHz of the 2.4 GHz peak clock (full-rate fp32 MFMAs + a 4 TB/s HBM stream meet the board's power limit):
 * bench.py quotes roofline fractions against the peak clock and reports this one next to them.  1 <= n, 0 <= naps <= 1000. */
int dnmf_clock_probe(unsigned long long* samples, int n, int naps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DNMF_H */
