/*
 * xmca_hip.h - C ABI of the MI355X (gfx950) implementation of the xmca solve / rotate / rule_n path.
 *
 * The reference (nicrie/xmca v1.4.2) is pure Python: it has no FFI of its own.  The drop-in boundary is
 * therefore the private numerical core of xmca.array.MCA, and every entry point below names the reference
 * lines it replaces.  The host side (xmca_amd/array.py) binds these symbols through ctypes; INTEGRATION.md
 * shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only.  Return value: 0 = ok, negative = error code below;
 *     xmca_last_error() returns the message of the last failure on that handle.
 *   - The caller owns every host buffer.  The library owns all device memory inside the opaque handle.
 *   - One handle = one HIP device + one stream.  A handle is not thread-safe; use one per thread / rank.
 *   - Matrices are row-major.  Complex data is interleaved (re, im) in host buffers.
 *   - dtype: 0 = float32, 1 = float64 (of the real components).
 */
#ifndef XMCA_HIP_H
#define XMCA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XMCA_OK 0
#define XMCA_ERR_INVALID (-1)        /* bad argument                                   -> ValueError            */
#define XMCA_ERR_HIP (-2)            /* HIP runtime failure                            -> RuntimeError          */
#define XMCA_ERR_NOT_CONVERGED (-3)  /* Varimax hit max_iter (rotation.py:66-71)       -> RuntimeError          */
#define XMCA_ERR_STATE (-4)          /* call order (e.g. vectors before solve)         -> RuntimeError          */
#define XMCA_ERR_UNSUPPORTED (-5)    /* outside device limits                          -> NotImplementedError   */
#define XMCA_ERR_NUMERIC (-6)        /* NaN / singular matrix (array.py:575-578)       -> numpy LinAlgError     */

#define XMCA_F32 0
#define XMCA_F64 1
#define XMCA_HOST 0
#define XMCA_DEVICE 1

typedef struct xmca_handle xmca_handle;

/* library / device management ------------------------------------------------------------------------- */
const char* xmca_version(void);
/* Number of this header's ABI (XMCA_ABI_VERSION): the binding refuses a library built from another revision. */
#define XMCA_ABI_VERSION 9
int xmca_abi_version(void);
int xmca_device_count(void);
int xmca_create(int device, xmca_handle** out);
void xmca_destroy(xmca_handle* h);
const char* xmca_last_error(xmca_handle* h);

/* Input of solve(): the centered, weighted, NaN-free T x N field of one side (0 = left, 1 = right), i.e.
 * MCA._fields[key] as returned by MCA._get_X()                         (xmca/array.py:95, :117, :289-298).
 *   re        real part, T*N elements of `dtype`
 *   im        imaginary part (analytic signal computed by the host, array.py:455-470) or NULL
 *   location  XMCA_HOST: copied to the device;  XMCA_DEVICE: device pointers, adopted without a copy
 *             (they must stay valid until the next xmca_set_field / xmca_destroy). */
int xmca_set_field(xmca_handle* h, int side, const void* re, const void* im, int64_t T, int64_t N, int dtype, int location);

/* Hilbert complexify on the device: X_im = Ht * X_re for every field set so far.  Ht (T x T) is the imaginary
 * part of the analytic-signal operator, imag(scipy.signal.hilbert(eye(T), axis=0)); it is circulant, so the
 * caller passes only its first column `hilbert_col` (T float64, host): Ht[t][s] = hilbert_col[(t - s) mod T].
 * Replaces scipy.signal.hilbert(field, axis=0) of array.py:464 for extend=False.
 * hilbert_col == NULL reverts to the real fields (a later solve on the same resident fields is a real one again). */
int xmca_complexify(xmca_handle* h, const double* hilbert_col);

/* MCA.solve numerical core (xmca/array.py:549-584): per-field SVD, kernel, kernel SVD, back-projection.
 *   n_fields  1 (EOF/PCA) or 2 (MCA)
 *   n_vec     number of leading modes to back-project into grid space; -1 = all `rank` modes
 *             (two float64 fields: the re-solve of weak modes - deflation / weak-block refinement, DESIGN.md 1 - covers the
 *             modes that get vectors; with 0 <= n_vec < rank the singular values beyond n_vec are those of the single solve,
 *             accurate to ~5e-14 (sigma_1 / sigma_i)^2 relative)
 *   rank_out  min(T, Nx, Ny)  (array.py:597) */
int xmca_solve(xmca_handle* h, int n_fields, int64_t n_vec, int64_t* rank_out);

/* MCA._singular_values (array.py:590): all `rank` singular values of A^H B / (T-1), descending, float64. */
int xmca_get_singular_values(xmca_handle* h, double* out, int64_t n);

/* MCA._V[key] (array.py:584), transposed: out[m * N + n] = V[n][m] for m < n_modes; complex interleaved when
 * the model is complex.  dtype selects float32 / float64 components. */
int xmca_get_vectors(xmca_handle* h, int side, void* out, int64_t n_modes, int dtype);

/* MCA.eofs() in its final layout (array.py:615-646 `_get_V` + :676-721 `_get_eofs`, ABI 9): out[n * q + c] = sum_{mm < m} V[n][mm] W[mm][c]
 * for the N grid points of `side` - the N x q array the reference reshapes to (space..., modes) - mixed on the device from the
 * resident mode-major vectors, so the host neither transposes nor multiplies an N x m array.  W (m x q, row-major float64,
 * complex interleaved when w_is_complex) = diag(sqrt(s)) R / norm with its columns ordered and selected by the caller; W == NULL:
 * the first q = m vectors as they are.  The output is complex (interleaved) when the model or W is; dtype: float32 / float64. */
int xmca_get_eofs(xmca_handle* h, int side, const double* W, int64_t m, int64_t q, int w_is_complex, void* out, int dtype);

/* PC projection of MCA._get_U (xmca/array.py:648-674, the product `fields[k] @ V[k]`): U = X~ V with X~ the field of
 * `side` as solve() saw it - still resident on the device; the analytic signal X + i Ht X when complexify was
 * requested (the imaginary field plane is not needed: U = W + i Ht W with W = X V).
 *   V      N x m row-major, float64 (is_complex = 0) or interleaved complex128 (is_complex = 1), host memory
 *   U_out  T x m row-major float64, interleaved complex128 when *out_is_complex = 1 (model or V complex)
 * The scaling by 1/sqrt(sigma) and the rotation (array.py:391-393) stay with the caller (m x m work). */
int xmca_project(xmca_handle* h, int side, const void* V, int64_t N, int64_t m, int is_complex, void* U_out,
                 int* out_is_complex);

/* Correlation maps of MCA.homogeneous_patterns / heterogeneous_patterns (xmca/array.py:1188-1261, the Pearson
 * correlation of tools/array.py:76-88): r[n][j] = corr(real part of field column n of `side`, Y[:, j]) on the resident
 * field - one tall GEMM X^T Y plus column moments instead of the reference's (N + m)^2 corrcoef matrix.
 *   Y      T x m row-major float64 (the real parts of the PCs), host memory
 *   r_out  N x m row-major float64
 * p-values (scipy.stats.beta) stay with the caller. */
int xmca_correlate(xmca_handle* h, int side, const double* Y, int64_t T, int64_t m, double* r_out);

/* Constructor preprocessing on the device (xmca/array.py:199-215 `_set_field_means` / `_set_field_stds` / `_center`;
 * SURVEY 8f row 3): the field of `side` set with xmca_set_field (raw, uncentered) is centered in place, column by
 * column; mean_out / std_out (ddof = 0, float64 accumulation) get N values each and *n_nan_out the number of NaN
 * entries found - when it is not zero nothing was changed and the caller takes its own NaN-column path. */
int xmca_center_field(xmca_handle* h, int side, double* mean_out, double* std_out, int64_t* n_nan_out);
/* NaN-column handling of the constructor on the device (xmca/array.py:191-197 `_set_no_nan_idx`, `_remove_nan_cols`,
 * tools/array.py:27-62): keep_out[c] = 1 for every column of the resident raw field of `side` that holds no NaN
 * (N ints), *n_keep_out their number; the resident field is replaced by those columns (T x n_keep, order kept).
 * With n_keep = 0 the field is left as it is (the caller raises the reference's error). */
int xmca_compact_field(xmca_handle* h, int side, int* keep_out, int64_t* n_keep_out);
/* Weights / normalisation of the constructor stage on the device (xmca/array.py:317-349 `apply_weights`, :351-365
 * `normalize`): every column c of the resident real field of `side` is multiplied (divide = 0) or divided (divide = 1)
 * by w[c]; w holds N values of the field's own element type, so the result equals the host operation bit for bit. */
int xmca_scale_field(xmca_handle* h, int side, const void* w, int divide);
/* Real plane of the resident field of `side` (T x N row-major, dtype XMCA_F32 / XMCA_F64 as it was set) -> host. */
int xmca_get_field(xmca_handle* h, int side, void* out);

/* MCA.bootstrapping (xmca/array.py:1813-1952): replicates on the device.
 * xmca_bootstrap_begin copies the real planes of the fields set with xmca_set_field (the caller's X_surr,
 * array.py:1925-1933) into working buffers.  Each xmca_bootstrap_run then
 *   resamples them along time, X <- X[idx, :] - cumulatively, like the reference's loop, which overwrites X_surr
 *   (idx_left / idx_right: T row indices drawn by the caller exactly as tools/array.py:91-138 does, or NULL when that
 *   side is not resampled),
 *   centers a copy (the MCA constructor, array.py:117), complexifies when hilbert_col != NULL, solves, rotates
 *   (rotated != 0: n_rot = p, power, tol) and returns the variance spectrum of `_get_variance` (array.py:755-779):
 *   `rank` singular values, or the p sorted norm products of the rotated model; *kept_out = 0 when Varimax failed. */
int xmca_bootstrap_begin(xmca_handle* h, int n_fields);
int xmca_bootstrap_run(xmca_handle* h, const double* hilbert_col, const int64_t* idx_left, const int64_t* idx_right, int rotated,
                       int p, int power, double tol, double* spectrum_out, int* kept_out, int64_t n_out);
/* All replicates of one bootstrap in one call.  idx_left / idx_right: n_runs x T row indices INTO THE FIELDS AS THEY WERE AT
 * xmca_bootstrap_begin - the caller composes the reference's cumulative resampling, c_r = c_{r-1}[idx_r] (idx_r drawn exactly
 * as tools/array.py:91-138 does) - or NULL for a side that is not resampled.  The replicates are then independent on the
 * device and several are kept in flight (lanes, as in xmca_rule_n).  spectra_out: n_runs x n_out, kept_out: n_runs.
 * Does not touch the cumulative state of xmca_bootstrap_run. */
int xmca_bootstrap_runs(xmca_handle* h, const double* hilbert_col, const int64_t* idx_left, const int64_t* idx_right, int64_t n_runs,
                        int rotated, int p, int power, double tol, double* spectra_out, int* kept_out, int64_t n_out);
int xmca_is_complex(xmca_handle* h);
/* 1 when the singular vectors of `side` from the last xmca_solve are resident in float32: a real float32 field decomposed on
 * its dual side (N > T) keeps `_V` in the input's dtype as the reference does (xmca/array.py:584, the dtype of
 * `VLT.conjugate().T`), and xmca_rotate_solved then multiplies float32 vectors by float32 sqrt(singular values) like the
 * reference's host code (array.py:818-822).  Every other result is float64 planes. */
int xmca_vectors_are_f32(xmca_handle* h, int side);
/* Persistent launches of this process (the register-resident tridiagonal reduction, the one-launch Varimax loop) that ran out
 * of their bounded waits because a workgroup never became resident, and were repeated on the launch-per-step path.  All
 * persistent kernels of a device pass one gate (csrc/common.h PersistGate), so this stays 0 unless ANOTHER process holds CUs. */
long long xmca_persistent_giveups(void);
/* Diagnostics of the last solve: for each of the up to three eigen-decompositions (left Gram, right Gram, kernel):
 * info[3*i + 0] = outer sweeps, info[3*i + 1] = tile size, info[3*i + 2] = pair slots (i = 0..2), then
 * info[9 + i]: bit 0 = the eigensolver inserted a Cholesky LR step (graded spectrum), bit 1 = the problem was solved by
 * reduction to tridiagonal form (csrc/tridiag.h: then no sweeps, tile and slots are 0).  n <= 12. */
int xmca_get_solve_info(xmca_handle* h, int* info, int n);

/* promax / varimax of xmca/tools/rotation.py:84-149, :15-78 on a host loading matrix L (N x p row-major,
 * float64, interleaved complex when is_complex), as built by MCA.rotate (array.py:821-822).
 *   n_left        rows belonging to the left field (array.py:818, :827-828)
 *   varimax_only  1: stop after Varimax (tools.rotation.varimax), 0: full Promax (also for power = 1)
 *   gamma         the `gamma` of tools.rotation.varimax (rotation.py:15, :56-57): 1 = Varimax (what promax / MCA.rotate
 *                 use), 0 = Quartimax; any real value is accepted
 *   B_out         NULL or N x p rotated loadings
 *   R_out/Phi_out p x p (interleaved complex when is_complex); norm_left/right: p column norms of the two
 *                 row blocks of the rotated loadings (array.py:827-828); iters_out: Varimax iterations run.
 * Returns XMCA_ERR_NOT_CONVERGED when max_iter iterations did not satisfy |d - d_old| / d < tol. */
int xmca_rotate_loadings(xmca_handle* h, const double* L, int64_t N, int64_t n_left, int p, int is_complex, int power,
                         double tol, int max_iter, int varimax_only, double gamma, double* B_out, double* R_out,
                         double* Phi_out, double* norm_left, double* norm_right, int* iters_out);

/* MCA.rotate (xmca/array.py:815-833) on the result of the last xmca_solve, without the vectors leaving the device: the
 * loadings V sqrt(sigma) of both fields are stacked (array.py:818-822) from the resident singular vectors, then as
 * xmca_rotate_loadings (full Promax, gamma = 1).  p <= the number of back-projected modes.  Outputs as there
 * (is_complex = xmca_is_complex(h)); XMCA_ERR_NOT_CONVERGED / XMCA_ERR_NUMERIC as there. */
int xmca_rotate_solved(xmca_handle* h, int p, int power, double tol, int max_iter, double* R_out, double* Phi_out,
                       double* norm_left, double* norm_right, int* iters_out);

/* MCA.rule_n surrogate loop (xmca/array.py:1753-1765) for runs [run_begin, run_end): N(0,1) surrogates
 * (Philox4x32-10 keyed by seed, run, side) generated on the device, centered, optionally complexified
 * (hilbert_col != NULL, see xmca_complexify), solved and, when `rotated`, rotated with (p, power, tol); each kept run contributes
 * MCA._get_variance() (array.py:772-779): n_out = rank values (unrotated) or p values (rotated), descending.
 *   spectra_out  (run_end - run_begin) x n_out float64;  kept_out[i] = 0 when run i was dropped because
 *                Varimax did not converge (array.py:1762-1763).
 * The final normalisation svals /= svals.sum(0) / ref.sum() (array.py:1767-1769) is left to the caller
 * because it needs the runs of every rank. */
int xmca_rule_n(xmca_handle* h, int64_t T, int64_t Nx, int64_t Ny, int n_fields, const double* hilbert_col, int rotated,
                int p, int power, double tol, int64_t run_begin, int64_t run_end, uint64_t seed, int dtype,
                double* spectra_out, int* kept_out, int64_t n_out);

/* Run sharding across the GPUs of a node - the only collective of the path (SURVEY 8(b) `mca_comm_*`, 8(e)).  The reference's
 * surrogate loop (xmca/array.py:1753-1765) is serial; its runs are independent, so rank r of `world` takes a contiguous block
 * of run indices on its own GPU and the per-run spectra are combined by ONE ncclAllGather over xGMI (RCCL, bound at run time:
 * XMCA_ERR_UNSUPPORTED when librccl.so.1 cannot be loaded).  One process per GPU, one communicator per handle's device.
 *   xmca_comm_unique_id  rank 0 fills XMCA_COMM_ID_BYTES bytes (an ncclUniqueId); the caller ships them to the other ranks
 *                        (file, MPI, a torch store - the library has no transport of its own besides RCCL).
 *   xmca_comm_create     collective over all `world` ranks (ncclCommInitRank on the device of `h`).
 *   xmca_comm_allgather  recv_host[r * count + i] = send_host[i] of rank r (float64, host buffers; staged through device memory).
 *   xmca_comm_broadcast  `count` float64 of `root` to every rank (used for the seed: every rank must key the generator alike).
 *   xmca_comm_info       rank, world, number of collectives carried out and bytes received in them (bench.py reports these).
 *   xmca_rule_n_sharded  xmca_rule_n for runs [0, n_runs) split over the ranks of `comm` (rank r: runs r*n/world ... as
 *                        xmca_amd/dist.py shard_range), seed taken from rank 0, then the all-gather: every rank receives all
 *                        n_runs x n_out spectra and kept flags - what array.py:1767-1771 normalises.  The generator is keyed by
 *                        (seed, run, side): the result does not depend on `world`. */
#define XMCA_COMM_ID_BYTES 128
typedef struct xmca_comm xmca_comm;
int xmca_comm_unique_id(void* id_out);
int xmca_comm_create(xmca_handle* h, const void* unique_id, int rank, int world, xmca_comm** out);
void xmca_comm_destroy(xmca_comm* c);
const char* xmca_comm_last_error(xmca_comm* c);
int xmca_comm_allgather(xmca_comm* c, const double* send_host, double* recv_host, int64_t count);
int xmca_comm_broadcast(xmca_comm* c, double* buf_host, int64_t count, int root);
int xmca_comm_info(xmca_comm* c, int* rank, int* world, int64_t* collectives, int64_t* bytes);
int xmca_rule_n_sharded(xmca_handle* h, xmca_comm* c, int64_t n_runs, int64_t T, int64_t Nx, int64_t Ny, int n_fields,
                        const double* hilbert_col, int rotated, int p, int power, double tol, uint64_t seed, int dtype,
                        double* spectra_out, int* kept_out, int64_t n_out);

/* Surrogate generator on its own (tests): T*N standard normals of (seed, run, side) as float64 on the host. */
int xmca_surrogate(xmca_handle* h, int64_t n, uint64_t seed, uint32_t run, uint32_t side, double* out);

/* hipEvent stage timers of the calls since the last reset: names are written as a ';'-separated list.  Two
 * kernel-level entries follow the stages when the eigensolver ran: "jacobi_round_kernel_ms" (total duration of the
 * jacobi_fused_round_kernel launches, events around the rounds of every sweep) and "jacobi_round_kernel_launches"
 * (their number, in the ms array). */
int xmca_get_timings(xmca_handle* h, char* names, int names_len, double* ms, int max_n);

/* The kernels of the handle's last tridiagonal reduction by name (instantiation, column range of every launch of the chain), e.g.
 * "chain of 6 persistent launches: trd_resident_kernel<real,NC=24,RR=3,tagged> columns [0,512) -> ...": what `roofline.kernel`
 * of bench.py reports (ABI 9; until round 5 the bench synthesised the name from T).  Returns the length of the description. */
int xmca_get_reduction_info(xmca_handle* h, char* out, int out_len);
int xmca_reset_timings(xmca_handle* h);

/* Batched complex DFT used by the analytic-signal path (csrc/fft.h), host in / host out, for the parity tests:
 * out[b][k] = sum_t in[b][t] exp(sign 2 pi i k t / n), b < batch, k < n; planes of batch x n float64, in_im may be NULL.
 * Returns XMCA_ERR_UNSUPPORTED when n has a prime factor above 7 or exceeds 5120 (the solver then uses GEMMs). */
int xmca_fft(xmca_handle* h, const double* in_re, const double* in_im, int batch, int n, int sign, double* out_re, double* out_im);

/* Device memory kept by the handle.  The solver's temporaries come from a per-handle pool (hipFree waits for the whole
 * device): blocks are kept after a call, up to XMCA_POOL_LIMIT_GB (default 16) in total; an allocation failure empties every pool of the process first.
 * xmca_pool_bytes reports what is held right now (lanes of rule_n included), xmca_trim_pool gives it back to the driver. */
int xmca_pool_bytes(xmca_handle* h, int64_t* held_bytes);
int xmca_trim_pool(xmca_handle* h);

/* Kernel-level entry points used by the parity tests and the roofline leg of bench.py ------------------- */
/* C (M x N, float64, host) = alpha * op(A) op(B);  a_kfast: A(m,k) = A[m*lda + k] else A[k*lda + m];
 * b_nfast: B(k,n) = B[k*ldb + n] else B[n*ldb + k];  dtype of A and B; upper_only/mirror as in gemm.h. */
int xmca_gemm(xmca_handle* h, const void* A, int64_t lda, int a_kfast, const void* B, int64_t ldb, int b_nfast, double* C,
              int M, int N, int K, int dtype, double alpha, int upper_only, int mirror, int splits);
/* Hermitian eigendecomposition of an n x n host matrix (interleaved complex when is_complex):
 * lam (n, descending) and Zh (n x n, row i = conj(u_i); NULL: eigenvalues only); info (4 ints): sweeps, tile, slots,
 * bit 0: Cholesky LR step taken, bit 1: solved by tridiagonalisation (csrc/tridiag.h; then no sweeps).  The device time of the
 * solve is recorded under "eigh_vectors" / "eigh_values" (xmca_get_timings). */
int xmca_eigh(xmca_handle* h, const double* A, int n, int is_complex, double* lam, double* Zh, int* info);
/* Blocked Cholesky of an n x n Hermitian host matrix (interleaved complex when is_complex): R (n x n, upper
 * triangular, row-major, same element layout) with R^H R = A + rel_shift * max(diag A) * I; *ok = 0 when a pivot was
 * not positive.  (The device routine behind the values-only two-field solve; exported for the kernel tests.) */
int xmca_cholesky(xmca_handle* h, const double* A, int n, int is_complex, double rel_shift, double* R, int* ok);
/* Time `reps` Gram products G = X X^T of the resident field `side` with hipEvents on the library's stream;
 * avg_ms = mean duration of one product (all launches it needs), kernel_ms = mean duration of the MFMA
 * kernel launches alone, flops = useful flops of one product, T (T+1) N. */
int xmca_bench_gram(xmca_handle* h, int side, int reps, double* avg_ms, double* kernel_ms, double* flops);

/* Time `reps` launches of C = op(A) op(B) on device-resident pseudo-random operands (no host traffic):
 * avg_ms per product including the split-K reduction when one is used. */
int xmca_bench_gemm(xmca_handle* h, int M, int N, int K, int dtype, int a_kfast, int b_nfast, int upper_only, int splits,
                    int reps, double* avg_ms);

#ifdef __cplusplus
}
#endif
#endif /* XMCA_HIP_H */
