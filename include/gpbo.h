/*
 * gpbo.h - C ABI of libgpbo: the MI355X (gfx950) GP-surrogate acquisition path.
 *
 * The reference has no FFI: its boundary is the Python attribute protocol between
 * select_parameters.py:146-157 / 282-293 and class PointSelector (point_selector.py:13-207).
 * bayesian_optimisation_amd.PointSelector keeps that protocol and binds the entry points below
 * through ctypes.  Each entry point names the reference code it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host; all buffers are
 *     caller-allocated and caller-owned, nothing is retained across calls;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls only enqueue
 *     work on it and never synchronise, so they may be captured into a hipGraph;
 *   - matrices are row-major fp64; observation-sized dimensions are padded to
 *     Np = gpbo_padded_n(N) (a multiple of 128): the padding carries the identity in K/L/U and
 *     zeros in y/alpha, so results on the leading N entries are those of the unpadded problem;
 *   - return value: GPBO_OK or a negative GPBO_ERR_* (no exceptions cross the ABI).  Numerical
 *     failures that are only known on the device (non-positive Cholesky pivot, NaN acquisition)
 *     are reported through the device-side `info` / `result` words the caller reads back.
 */
#ifndef GPBO_H
#define GPBO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPBO_VERSION 151 /* 0.5.1: + gpbo_nlml_grid_wave_f64 / _wave_logdet_f64 (the likelihood grid of N <= 64 observations, a wave per cell); 0.5.0: the likelihood grid of any N in one launch (one workgroup per cell), a second likelihood mode (log det from the factor: gpbo_nlml_grid_*logdet*) */

/* Environment switches the SHIPPED library reads (each once per process; none changes a result beyond the rounding of a
 * different summation order, none is needed for normal use - they select between measured alternatives for A/B runs):
 *   GPBO_F64_GROUPS=g     column groups of the fp64 variance kernel for large calls (default 8: one per XCD)
 *   GPBO_OVERLAP=1        K(X*,X) of chunk c + 1 on a second stream beside the variance launch of chunk c (measured: no gain)
 *   GPBO_PREFIX_VALU=1    the prefix bound's first-level mean from the difference-form kernel instead of the MFMA one
 *   GPBO_FACTOR_OLD=1     the round-2 factorisation chain (Cholesky + triangular inverse) instead of the fused sweep
 *   GPBO_NO_LOOKAHEAD=1   that chain without its helper stream
 *   GPBO_CI_OPTS=a,b,c,d  launch-plan options of the fused sweep (csrc/cholinv_plan.h)
 * Read only by DIAGNOSTICS builds (-DGPBO_DIAGNOSTICS: tools/build_variant.sh, never the shipped library; some produce
 * wrong results on purpose to time a phase): GPBO_KSTAR_VARIANT, GPBO_SIGMA_VARIANT, GPBO_CI_STAMPS, GPBO_FPS_MUTE (fault
 * injection), GPBO_FPS_SHAPE, and the compile-time GPBO_ARD_SKIP / GPBO_ARD_STAMPS of csrc/ard.hip. */

#define GPBO_OK 0
#define GPBO_ERR_ARG (-1)      /* null pointer, bad size/alignment, unsupported d */
#define GPBO_ERR_LAUNCH (-2)   /* HIP reported a launch/runtime error */
#define GPBO_ERR_WORKSPACE (-3) /* workspace too small */

#define GPBO_MAX_D 16          /* compile-time-unrolled feature counts 1..16 (every route) */
#define GPBO_MAX_D_ANY 1024    /* fp64 route only (kxx, factorise, posterior_acq_f64, select_next_host): any d up to this, slow path */
#define GPBO_NPAD 128          /* observation padding granule (column-block width of the variance kernel) */
#define GPBO_CHUNK_GRANULE 512 /* candidate-chunk granule */
#define GPBO_CHUNK_MAX (1 << 24) /* largest chunk (16-row tile offsets inside K*^T are 32-bit element offsets) */

#define GPBO_ACQ_LCB 0 /* acq = p0*sigma - mu            (point_selector.py:204, p0 = explore) */
#define GPBO_ACQ_EI 1  /* acq = EI for minimisation, p0 = f_best, p1 = xi (not in the reference) */

/* Result record written by gpbo_posterior_acq_f64 (device memory, 32 bytes). */
typedef struct gpbo_result {
    double best_val;   /* max acquisition over the candidates given to this call          */
    int64_t best_idx;  /* LOWEST global index attaining it (point_selector.py:207 tie rule) */
    int64_t nan_count; /* number of candidates whose acquisition is NaN (reference: IndexError) */
    int64_t reserved;
} gpbo_result;

/* Optional timing of the dominant kernel (sigma/acquisition) and of the K(X*,X) build: hipEvents recorded on
 * the caller's stream by gpbo_posterior_acq_f64 when a profile is passed.  The launches of one call form a chain
 * K*(0), variance(0), K*(1), variance(1), ... on one stream, so one event per boundary serves as the end of one
 * launch and the beginning of the next (an event record costs a few microseconds of idle GPU).  Host-side object. */
typedef struct gpbo_profile {
    int32_t capacity, count;
    void **begin, **end;   /* hipEvent_t before / after each variance/acquisition launch */
    int64_t *cands;        /* candidates processed by each recorded launch */
    void **kbegin;         /* hipEvent_t before the K(X*,X) launch of the slot (used when kmode == 1) */
    int32_t *kmode;        /* K(X*,X) launch of the slot: 0 not timed, 1 kbegin[i]..begin[i], 2 end[i-1]..begin[i] */
    void **qend;           /* gpbo_posterior_qei_f64: hipEvent_t after the qEI launch of the slot (end[i]..qend[i]) */
    int32_t *qmode;        /* 1: the slot has a qEI interval */
} gpbo_profile;
int gpbo_profile_create(int32_t capacity, gpbo_profile **out);
void gpbo_profile_reset(gpbo_profile *p);
/* Waits for the recorded events; sums elapsed ms, launches and candidates over all recorded launches. */
int gpbo_profile_read(gpbo_profile *p, double *total_ms_host, int64_t *launches_host, int64_t *cands_host);
/* Same for the K(X*,X) launches (fp64 path). */
int gpbo_profile_read_kstar(gpbo_profile *p, double *total_ms_host, int64_t *launches_host, int64_t *cands_host);
/* Same for the qEI launches (gpbo_posterior_qei_f64 with a profile). */
int gpbo_profile_read_qei(gpbo_profile *p, double *total_ms_host, int64_t *launches_host, int64_t *cands_host);
void gpbo_profile_destroy(gpbo_profile *p);

int gpbo_version(void);
const char *gpbo_strerror(int status);
int64_t gpbo_padded_n(int64_t N);

/* K1 - replaces kernel_rbf(X, X) + jitter assembly (point_selector.py:166-195 with :79, :116).
 * Kp[Np x Np]: leading N x N = exp(-1/2 sum_k (x_ik-x_jk)^2 / ls_k^2), diagonal = (1 + jitter1) + jitter2
 * (the reference adds 1e-4 inside kernel_rbf and 1e-6 at assembly, in that order); padding = identity.
 * X: [N x d] row-major, ls: [d] length scales (kernel_params). */
int gpbo_kxx_f64(const double *X, int64_t N, int32_t d, const double *ls_host, double jitter1, double jitter2,
                 double *Kp, int64_t Np, void *stream);

/* K4 - replaces np.linalg.inv(cov_meas) (point_selector.py:89) by a blocked right-looking Cholesky.
 * In place: lower triangle of Kp becomes L (upper triangle is left untouched). dinv [Np/64][64][64]
 * receives the inverses of the diagonal blocks of L. info (device int32): 0, or 1-based column of the
 * first non-positive / non-finite pivot.  Np: any multiple of 64. */
int gpbo_potrf_f64(double *Kp, int64_t Np, double *dinv, int32_t *info, void *stream);

/* U = (L^-1)^T, upper triangular, [Np x Np] row-major (strict lower triangle zero-filled).
 * work: Np*Np doubles. Together with gpbo_potrf_f64 this is the factorisation the posterior uses. */
int gpbo_trtri_f64(const double *L, const double *dinv, int64_t Np, double *U, double *work, void *stream);

/* K4, round 3 - the same inverse (point_selector.py:89) in ONE sweep: fused Cholesky + inverse factor by row
 * operations on the stacked matrix S = [A | W], [Np x ld] row-major with ld >= 2 Np (csrc/cholinv.hip).
 * In: columns [0, Np) = the symmetric positive definite matrix (both triangles), columns [Np, 2 Np) = zeros.
 * Out: columns [Np, 2 Np) = inv(L), lower triangular (U of gpbo_trtri_f64 is its transpose); the upper block triangle
 * of columns [0, Np) holds L^T except on its 128 x 128 diagonal blocks, which keep their last Schur complements.
 * info as gpbo_potrf_f64.  Np: a multiple of 128.  opt: NULL, or int32[7] {win, far_k, far_kind (3: 128 x 128 tiles,
 * 4: 256 x 128), defer + 1, max_launches, group_from + 1, small_w (64 or 32)}, 0 = default: schedule choices (csrc/cholinv_plan.h) and, for tests, the
 * first max_launches launches of the plan only.  The first call for a size uploads its plan (synchronous). */
int gpbo_cholinv_f64(double *S, int64_t ld, int64_t Np, int32_t *info, const int32_t *opt, void *stream);
/* The launch plan of gpbo_cholinv_f64 as data (no GPU needed; tests/test_cholinv_plan_cpu.py executes it with NumPy).
 * opt: as above (max_launches ignored).  *n_launch / *n_tile: in = capacity of launches[] / tiles[] in entries (ignored
 * when the array is NULL), out = entries of the plan.  launches: 5 words each {pair or -1, workgroups of PAIR(pair),
 * first tile, tiles, tiles per workgroup}; tiles: 8 words each {kind (2: 64 x w, 3: 128 x 128, 4: 256 x 128), k0, K, row0, col0, r1, wlim, w (kind 2: 64 or 32)}:
 * S[row0.., col0..] -= sum over source rows [k0, k0 + K) of S[k, row0..]^T S[k, col0..], rows < r1, live columns only. */
int gpbo_cholinv_plan(int64_t Np, const int32_t *opt, int64_t *n_launch, int64_t *n_tile, int32_t *launches, int32_t *tiles);
/* One launch = the PAIR workgroups of `pair` (< 0: none) + the given tiles (host array, format above; `group`
 * consecutive tiles per workgroup), `reps` times;
 * synchronous.  For tests of a tile kind by itself and tools/bench_ci_jobs.py.  info is not cleared here. */
int gpbo_cholinv_tiles_f64(double *S, int64_t ld, int64_t Np, int32_t *info, int32_t pair, const int32_t *tiles,
                           int64_t ntile, int32_t group, int32_t reps, void *stream);

/* alpha = K^-1 y = U (U^T y)   (point_selector.py:90 `inv @ measured_vals`).
 * y: [N]; alpha: [Np] (zero on the padding); tmp: [Np]. */
int gpbo_alpha_f64(const double *U, const double *y, int64_t N, int64_t Np, double *tmp, double *alpha,
                   void *stream);

/* One call = kxx + cholinv + transpose + alpha (round 2: kxx + potrf + trtri + alpha).  work: gpbo_factorise_workspace_bytes(Np) bytes.
 * Outputs: Kp keeps K (the reference's `cov_meas`, point_selector.py:79; L lives in the workspace),
 * U [Np x Np], alpha [Np], info (device int32, as gpbo_potrf_f64). */
int64_t gpbo_factorise_workspace_bytes(int64_t Np);
int gpbo_factorise_f64(const double *X, const double *y, int64_t N, int32_t d, const double *ls_host,
                       double jitter1, double jitter2, int64_t Np, double *Kp, double *U, double *alpha,
                       int32_t *info, void *work, int64_t work_bytes, void *stream);

/* Append one observation to an existing factorisation in O(N^2) instead of refactorising (SURVEY.md §8f rank 4:
 * consecutive BO iterations differ by one observed row, select_parameters.py:163,299, while
 * point_selector.py:89 inverts the whole matrix again).  Valid only while the length scales and jitters are
 * those the factors were built with.  In place, N -> N+1:
 *   X [>= N+1 rows x d], y [>= N+1]: row N receives x_new / y_new (device pointers to d and 1 doubles);
 *   U [Np x Np]: column N is written (U' = [U, -U l/lambda; 0, 1/lambda], l = U^T k, lambda^2 = K_NN - l.l);
 *   alpha [Np]: recomputed as U'(U'^T y');   Kp [Np x Np] or NULL: row and column N of K.
 * Needs N + 1 <= Np (Np a multiple of 64; the caller re-pads into larger buffers when the padding is used up:
 * identity on the new diagonal).  info (device int32): 0, or N+1 when the new pivot is not positive - U and alpha are
 * then unchanged (the surrogate of the N old observations stays valid) and the caller refactorises.  work: gpbo_append_workspace_bytes(Np). */
int64_t gpbo_append_workspace_bytes(int64_t Np);
int gpbo_append_f64(double *X, double *y, int64_t N, int32_t d, const double *ls_host, double jitter1,
                    double jitter2, int64_t Np, const double *x_new, const double *y_new, double *Kp, double *U,
                    double *alpha, int32_t *info, void *work, int64_t work_bytes, void *stream);

/* Xsc[Np x d] = X / (ls sqrt 2) (rows >= N zero): the pre-scaled observations gpbo_kstar_mu_f64 reads. */
int gpbo_scale_points_f64(const double *X, int64_t N, int64_t Np, int32_t d, const double *ls_host, double *Xsc,
                          void *stream);

/* K2+K5 - replaces kernel_rbf(X, X*).T and the mean product (point_selector.py:81, :90).
 * Builds the transposed cross-covariance chunk KsT[Np x ldk] (row n = observation n, column c =
 * candidate c of the chunk; rows n >= N are zero) and per-64-observation partial sums of
 * mu_c = sum_n k(x*_c, x_n) alpha_n into mu_part[(Np/64) x ldk].
 * Xs: [Mc x d] candidates of this chunk (unscaled); Xsc: output of gpbo_scale_points_f64;
 * Mc <= ldk, ldk a multiple of 512.
 * diag_add / cand_base: when the caller's full candidate set has the SAME SHAPE as X the reference
 * adds 1e-4 where observation index == global candidate index (point_selector.py:173,191-193);
 * pass diag_add = 0 otherwise. */
int gpbo_kstar_mu_f64(const double *Xs, int64_t Mc, const double *Xsc, int64_t N, int64_t Np, int32_t d,
                      const double *ls_host, const double *alpha, double diag_add, int64_t cand_base,
                      double *KsT, int64_t ldk, double *mu_part, void *stream);

/* K5..K8, whole candidate set of this rank: chunks of `chunk` candidates through gpbo_kstar_mu_f64 and
 * the fused sigma/acquisition/argmax kernel (point_selector.py:90-98, :204-207).
 *   sigma_c = sqrt(|prior_var - |U^T k_c|^2|)     (abs and sqrt as at :98; "cov_func" is a std-dev)
 *   acq_c   = LCB or EI;  result = first-index argmax over c in [0, M), reported as idx_offset + c.
 * mu_out / sigma_out / acq_out: optional dense [M] outputs (NULL to skip).
 * work: gpbo_posterior_workspace_bytes(Np, chunk, M) bytes. */
int64_t gpbo_posterior_workspace_bytes(int64_t Np, int64_t chunk, int64_t M);
int gpbo_posterior_acq_f64(const double *Xs, int64_t M, const double *X, int64_t N, int64_t Np, int32_t d,
                           const double *ls_host, const double *U, const double *alpha, double prior_var,
                           int32_t acq_kind, double p0, double p1, double diag_add, int64_t idx_offset,
                           int64_t chunk, double *mu_out, double *sigma_out, double *acq_out,
                           gpbo_result *result, void *work, int64_t work_bytes, gpbo_profile *prof /* or NULL */,
                           void *stream);

/* q = 8 Monte-Carlo Expected Improvement over consecutive batches of 8 candidates (BASELINE config 5; not in the
 * reference).  qEI_b = mean_s max(0, max_j(f_best - xi - (mu_b + chol(Sigma_b) z_s)_j)); Z: [S x 8] fixed base
 * samples (device); M a multiple of 8 (chunks must not split a batch: chunk is a multiple of 512).
 * result.best_idx = batch_offset + index of the first batch attaining the maximum; qei_out: optional [M/8]. */
int64_t gpbo_qei_workspace_bytes(int64_t Np, int64_t chunk, int64_t M);
int gpbo_posterior_qei_f64(const double *Xs, int64_t M, const double *X, int64_t N, int64_t Np, int32_t d,
                           const double *ls_host, const double *U, const double *alpha, double prior_var, double f_best,
                           double xi, const double *Z, int32_t S, int64_t batch_offset, int64_t chunk, double *qei_out,
                           gpbo_result *result, void *work, int64_t work_bytes, gpbo_profile *prof /* or NULL */,
                           void *stream);

/* fp32-screened scoring (BASELINE config 4).  The factorisation stays fp64; U is rounded to fp32 once per step
 * (gpbo_prepare_f32, re-padded to Np32 = gpbo_padded_n_f32(N), a multiple of 256; alpha32 may be NULL).
 * gpbo_posterior_acq_f32 = the screen over all M candidates: K(X*,X) entries and the mean in fp64 exactly as in the
 * fp64 path (mu_out is the fp64 path's mean bit for bit), K*^T stored in fp32, the N^2-per-candidate triangular
 * product on the fp32 matrix cores.  Dense outputs are doubles holding fp32-accurate values; var_out is the signed
 * prior_var - |v|^2 the screen decides on.  alpha: the fp64 alpha of the factorisation.  chunk: a multiple of 1024.
 * result: the screen's own arg-max (NOT final - see gpbo_rescore_f64). */
int64_t gpbo_padded_n_f32(int64_t N);
int gpbo_prepare_f32(const double *U, const double *alpha, int64_t Np, float *U32, float *alpha32, int64_t Np32,
                     void *stream);
int64_t gpbo_posterior_workspace_bytes_f32(int64_t Np32, int64_t chunk, int64_t M);
int gpbo_posterior_acq_f32(const double *Xs, int64_t M, const double *X, int64_t N, int64_t Np32, int32_t d,
                           const double *ls_host, const float *U32, const double *alpha, double prior_var,
                           int32_t acq_kind, double p0, double p1, double diag_add, int64_t idx_offset, int64_t chunk,
                           double *mu_out, double *sigma_out, double *acq_out, double *var_out, gpbo_result *result,
                           void *work, int64_t work_bytes, gpbo_profile *prof /* or NULL */, void *stream);

/* The decision behind the screen: the selected point must be the fp64 path's (point_selector.py:204-207: first index
 * of the maximum).  Given the screen's dense mu [M] and var32 [M]: every candidate whose acquisition could still reach
 * the best lower bound when its variance is off by up to tau - plus every sample_stride-th candidate - is gathered and
 * re-scored through the fp64 kernels (gpbo_posterior_acq_f64 on the gathered rows); result = fp64 maximum over those,
 * lowest original index on ties, idx_offset added.  tau starts at tau0 and is CHECKED on every call: the largest
 * |var64 - var32| on the re-scored set must stay below tau / 4, else tau is raised and the selection repeated.
 * stats_host->fallback = 1 (result untouched) when more than `cap` candidates survive or tau does not settle in four
 * rounds: the caller then runs gpbo_posterior_acq_f64 over all candidates.  This call synchronises `stream` (it
 * reads the survivor count back); not supported with the N == M diagonal quirk (diag_add != 0: use the fp64 path). */
typedef struct gpbo_screen_stats {
    int64_t survivors; /* candidates selected in the last round (may exceed cap when fallback is set) */
    int64_t rescored;  /* candidates pushed through the fp64 kernels, all rounds */
    int32_t rounds;
    int32_t fallback;
    double tau;        /* variance tolerance of the last round */
    double err_max;    /* largest |sigma64^2 - |var32|| seen on the last round's re-scored set */
} gpbo_screen_stats;
int64_t gpbo_rescore_workspace_bytes(int64_t Np, int64_t cap, int64_t chunk64);
int gpbo_rescore_f64(const double *Xs, int64_t M, const double *mu, const double *var32, const double *X, int64_t N,
                     int64_t Np, int32_t d, const double *ls_host, const double *U, const double *alpha,
                     double prior_var, int32_t acq_kind, double p0, double p1, int64_t idx_offset, double tau0,
                     int64_t sample_stride, int64_t cap, int64_t chunk64, gpbo_result *result,
                     gpbo_screen_stats *stats_host, void *work, int64_t work_bytes, void *stream);

/* Prefix-bound screen: exact branch and bound, everything in fp64 (csrc/sigma_acq.hip, csrc/rescore.hip).
 * |U^T k_c|^2 summed over the FIRST n_prefix components is a lower bound of the whole sum - it is the variance reduction
 * from the first n_prefix observations alone - so  sigma_ub = sqrt(|prior_var - partial|) >= cov_func_c  and, both
 * acquisitions being increasing in sigma (LCB for explore >= 0), acq_ub_c >= acq_func_eval_c (point_selector.py:98,204).
 * gpbo_posterior_prefix_f64: one pass over all candidates - the mean over all N observations (exactly the fp64 path's),
 * the variance product over the first n_prefix columns only ((n_prefix / Np)^2 of the work; n_prefix a multiple of 128).
 * Same workspace as gpbo_posterior_acq_f64; `result` receives the arg-max of the BOUNDS (not the answer).
 * gpbo_bound_select_f64: the strided sample goes through the fp64 kernels (a lower bound of the maximum), every
 * candidate whose bound reaches it survives and is re-scored by the fp64 kernels, which decide (maximum, lowest index,
 * NaN count as the plain pass); too many survivors -> stats->fallback = 1 and the caller runs gpbo_posterior_acq_f64.
 * work: gpbo_rescore_workspace_bytes(Np, cap, chunk64).  This call synchronises the stream.
 * When the bound is one: U, X, alpha must be ONE factorisation (the same the plain pass would use): the bound's |v[:J]|^2 is
 * then a partial sum of the squares gpbo_posterior_acq_f64 adds up.  The plain pass takes sqrt(|var|) (point_selector.py:98),
 * and a NEGATIVE computed variance - possible only when its rounding error exceeds the jitter tau of K = k(X,X) + tau I,
 * since the true variance is >= tau for prior_var >= 1 + tau - is the one value a prefix cannot bound: callers take this route
 * for tau >= 1e-6 (DeviceGP.BOUND_MIN_JITTER, gpbo_select_next_host_f64) and the plain pass otherwise. */
int gpbo_posterior_prefix_f64(const double *Xs, int64_t M, const double *X, int64_t N, int64_t Np, int32_t d,
                              const double *ls_host, const double *U, const double *alpha, double prior_var,
                              int32_t acq_kind, double p0, double p1, int64_t idx_offset, int64_t chunk, int64_t n_prefix,
                              double *mu_out, double *sigma_ub_out, double *acq_ub_out, gpbo_result *result, void *work,
                              int64_t work_bytes, gpbo_profile *prof /* or NULL */, void *stream);
int gpbo_bound_select_f64(const double *Xs, int64_t M, const double *acq_ub, const double *X, int64_t N, int64_t Np,
                          int32_t d, const double *ls_host, const double *U, const double *alpha, double prior_var,
                          int32_t acq_kind, double p0, double p1, int64_t idx_offset, int64_t sample_stride, int64_t cap,
                          int64_t chunk64, int64_t n_prefix2 /* 0, or a longer prefix (multiple of 128) that tightens the
                          bounds of the survivors before they are re-scored */, gpbo_result *result,
                          gpbo_screen_stats *stats_host, void *work, int64_t work_bytes, void *stream);

/* Order independence of that bound (csrc/subset.hip; version 140, replaces the gpbo_*_subset_f64 calls of 130): which
 * observations come FIRST decides how much the prefix prunes, and a history sorted along an axis or started inside one
 * cluster is a bad prefix.  gpbo_fps_order_f64 writes a permutation of the observations - J members by farthest-point
 * sampling in length-scale units (first member: the observation farthest from the centroid; then the one farthest from
 * the members so far; ties to the lowest index), then every other observation in index order - and gathers X / y in that
 * order.  The caller factorises the PERMUTED problem (a GP's posterior does not depend on the order of its observations:
 * same mean_func / cov_func within rounding, point_selector.py:89-98) and runs every pass - gpbo_posterior_acq_f64 as
 * well as the two calls above - on that one factorisation, so the bound's |v[:J]|^2 is a partial sum of the squares the
 * plain pass itself adds up, whatever the history.
 * perm_out [N] int64, Xp_out [N x d], yp_out [N] (both optional; y may be NULL when yp_out is): device memory;
 * 1 <= J <= N, d <= GPBO_MAX_D; work: gpbo_fps_order_workspace_bytes(N) bytes, 256-byte aligned. */
int64_t gpbo_fps_order_workspace_bytes(int64_t N);
int gpbo_fps_order_f64(const double *X, const double *y, int64_t N, int32_t d, const double *ls_host, int64_t J,
                       int64_t *perm_out, double *Xp_out, double *yp_out, void *work, int64_t work_bytes, void *stream);
/* The selection runs on up to 16 co-operating workgroups that wait for each other with BOUNDED polls; should one of them
 * never be scheduled the others give up after about a second and the ARRIVAL order is installed (exact as well, it only
 * prunes less).  gpbo_fps_order_status writes 1 into *fell_back (device int32) when the last gpbo_fps_order_f64 on this
 * workspace ended that way, else 0 - a caller whose peers must hold the same factorisation (candidate shards) checks it. */
int gpbo_fps_order_status(const void *work, int64_t N, int32_t *fell_back, void *stream);

/* int8-sliced variance screen (Ozaki-style splitting on the integer matrix cores, csrc/ozaki.hip): same role and
 * outputs as gpbo_posterior_acq_f32 - mean exactly the fp64 path's, variance from 20 exact int8 slice products
 * (|dsigma| ~ 1e-10 at N = 4096), decision by gpbo_rescore_f64 - at N <= GPBO_I8_MAX_N (int32 accumulators).
 * gpbo_prepare_i8: once per factorisation, U -> column scales + int8 MFMA fragments in `u8`
 * (gpbo_prepare_i8_bytes(Np) bytes, 256-byte aligned).  chunk: a multiple of 512.  No N == M diagonal quirk. */
#define GPBO_I8_MAX_N 16384
int64_t gpbo_prepare_i8_bytes(int64_t Np);
int gpbo_prepare_i8(const double *U, int64_t Np, void *u8, int64_t u8_bytes, void *stream);
int64_t gpbo_posterior_workspace_bytes_i8(int64_t Np, int64_t chunk, int64_t M);
int gpbo_posterior_acq_i8(const double *Xs, int64_t M, const double *X, int64_t N, int64_t Np, int32_t d,
                          const double *ls_host, const void *u8, const double *alpha, double prior_var,
                          int32_t acq_kind, double p0, double p1, int64_t idx_offset, int64_t chunk, double *mu_out,
                          double *sigma_out, double *acq_out, double *var_out, gpbo_result *result, void *work,
                          int64_t work_bytes, gpbo_profile *prof /* or NULL */, void *stream);
/* Coarse variant of the same screen: the three leading digits of both operands, SIX slice products, 256-row tiles
 * (|dsigma^2| ~ 2e-4 at N = 4096: start gpbo_rescore_f64 with tau0 ~ 1e-3).  Same arguments, buffers and workspace
 * size as gpbo_posterior_acq_i8 (it reads slices 0-2 of the same `u8`); the mean is still the fp64 path's. */
int gpbo_posterior_acq_i8c(const double *Xs, int64_t M, const double *X, int64_t N, int64_t Np, int32_t d,
                           const double *ls_host, const void *u8, const double *alpha, double prior_var,
                           int32_t acq_kind, double p0, double p1, int64_t idx_offset, int64_t chunk, double *mu_out,
                           double *sigma_out, double *acq_out, double *var_out, gpbo_result *result, void *work,
                           int64_t work_bytes, gpbo_profile *prof /* or NULL */, void *stream);

/* K7+K8 on a posterior already on the device: acq = LCB/EI of (mu, sigma), first-index arg-max
 * (point_selector.py:204-207).  Used for a second acquisition on the same surrogate.
 * work: gpbo_acq_workspace_bytes() bytes, 256-byte aligned. */
int64_t gpbo_acq_workspace_bytes(void);
int gpbo_acq_argmax_f64(const double *mu, const double *sigma, int64_t M, int32_t acq_kind, double p0, double p1,
                        int64_t idx_offset, double *acq_out, gpbo_result *result, void *work, int64_t work_bytes,
                        void *stream);

/* ---- Host-pointer entry points: the reference's call sequence on NumPy-style arrays, no device handling by the
 * caller (device buffers and a private stream live inside the call).  What a ctypes stub in the reference binds.
 *
 * gpbo_select_next_host_f64 = PointSelector.update_surrogate() once kernel_params are chosen + the acquisition
 * arg-max (point_selector.py:76-98, 197-207; caller: select_parameters.py:149-158, 285-294):
 *   X [N x d], y [N], ls [d] (kernel_params), Xs [M x d] (predicted_pts, row-major grid) - host, fp64;
 *   jitter1 / jitter2 = 1e-4 / 1e-6 for the reference's arithmetic; acq_kind / p0 / p1 as gpbo_posterior_acq_f64;
 *   diag_add = 1e-4 when Xs has the same shape as X (point_selector.py:173), else 0; chunk = 0 for the default;
 *   mu_out / sigma_out / acq_out: optional host [M] (mean_func, cov_func, acq_func_eval before reshaping);
 *   cov_meas_out: optional host [N x N] (the cov_meas attribute);
 *   result (host): best value, lowest flat index attaining it, NaN count (> 0: the reference raises IndexError);
 *   info (host): 0, or the 1-based failing pivot (the reference's inv() raises LinAlgError or returns garbage) -
 *   then nothing is scored and result->best_idx = -1.
 * gpbo_nlml_grid_host_f64 = tune_kernel()'s float32 likelihood grid (point_selector.py:104-163): ls_cells [G x d]
 *   host, out [G] host float32; any N (the wave-per-cell kernel up to 64 observations, the one-launch workgroup-per-cell
 *   kernel beyond). */
/* (When mu_out, sigma_out and acq_out are all NULL, M >= 32768, N > 896 and the acquisition increases with sigma, the next
 * point is found by branch and bound on the exact prefix bound - gpbo_posterior_prefix_f64 / gpbo_bound_select_f64 below -
 * instead of computing every variance: same index and NaN count, the value within 1e-12 relative of the plain pass.) */
int gpbo_select_next_host_f64(const double *X_host, const double *y_host, int64_t N, int32_t d, const double *ls_host,
                              double jitter1, double jitter2, const double *Xs_host, int64_t M, int32_t acq_kind,
                              double p0, double p1, double diag_add, int64_t chunk, double *mu_out_host,
                              double *sigma_out_host, double *acq_out_host, double *cov_meas_out_host,
                              gpbo_result *result_host, int32_t *info_host);
/* q = 8 Monte-Carlo qEI on host arrays (gpbo_posterior_qei_f64 behind the same device handling): Z_host [S x 8] base
 * samples, M a multiple of 8; qei_out_host optional [M/8]; result->best_idx = first batch with the largest qEI. */
int gpbo_select_qei_host_f64(const double *X_host, const double *y_host, int64_t N, int32_t d, const double *ls_host,
                             double jitter1, double jitter2, const double *Xs_host, int64_t M, double f_best, double xi,
                             const double *Z_host, int32_t S, int64_t chunk, double *qei_out_host,
                             gpbo_result *result_host, int32_t *info_host);
int gpbo_nlml_grid_host_f64(const double *X_host, const double *y_host, int64_t N, int32_t d,
                            const double *ls_cells_host, int64_t G, double jitter, float *out_host);
/* the log-det likelihood mode (gpbo_nlml_grid_batched_logdet_f64) on host arrays: out_host [G] fp64 */
int gpbo_nlml_grid_logdet_host_f64(const double *X_host, const double *y_host, int64_t N, int32_t d,
                                   const double *ls_cells_host, int64_t G, double jitter, double *out_host);

/* K9 - replaces tune_kernel / eval_log_marginal (point_selector.py:104-163): float32 grid of
 * nlml = 0.5 (y^T K^-1 y + log det K + N log 2pi), K = k(X,X) + jitter I, one value per grid cell.
 * ls_cells: [G x d] length scales of each cell (device); out: [G] float32 (device).
 * N <= gpbo_nlml_grid_max_n() (the bordered matrix is factorised in LDS). */
int gpbo_nlml_grid_max_n(void);
int gpbo_nlml_grid_f64(const double *X, const double *y, int64_t N, int32_t d, const double *ls_cells, int64_t G,
                       double jitter, float *out, void *stream);

/* The same grid at the reference's own sizes, N <= gpbo_nlml_grid_wave_max_n() (= 64): one WAVE per cell, the cell's
 * matrix in registers (lane = row), the column steps unrolled with v_readlane broadcasts, y as a right-hand side
 * (csrc/ard_wave.hip; ABI 151).  Same arguments and the same value per cell as gpbo_nlml_grid_f64 up to the rounding
 * of another elimination order; a pivot that is not positive gives NaN (the reference's log(det < 0)).
 * ..._wave_logdet_f64: the second likelihood mode (fp64 output, log det K straight from the factor: see below). */
int gpbo_nlml_grid_wave_max_n(void);
int gpbo_nlml_grid_wave_f64(const double *X, const double *y, int64_t N, int32_t d, const double *ls_cells, int64_t G,
                            double jitter, float *out, void *stream);
int gpbo_nlml_grid_wave_logdet_f64(const double *X, const double *y, int64_t N, int32_t d, const double *ls_cells,
                                   int64_t G, double jitter, double *out, void *stream);

/* The same grid for ANY N, one launch: a persistent workgroup per cell runs a left-looking blocked Cholesky of the
 * cell's K (entries generated on the fly, the factor kept in MFMA fragment order in a scratch slot of the workspace,
 * y carried as one more row so that y^T K^-1 y = |L^-1 y|^2).  Same value per cell as gpbo_nlml_grid_f64 up to the
 * rounding of a blocked elimination order.  work: ..._workspace_bytes(N, G), 256-byte aligned.
 *
 * Two likelihood modes (INTEGRATION.md "ARD likelihood modes"):
 *   gpbo_nlml_grid_batched_f64         the REFERENCE's value: float32, log det K evaluated as log(exp(logdet)) because
 *                                      point_selector.py:118 takes np.log(np.linalg.det(K)) - det underflows to 0 beyond
 *                                      N ~ 100 and the cell becomes -inf (reproduced on purpose: parity);
 *   gpbo_nlml_grid_batched_logdet_f64  a documented DEPARTURE for the sizes where that is useless: fp64 output,
 *                                      log det K = 2 sum log L_ii straight from the factor (finite at any N), NaN when a
 *                                      pivot is not positive.  Any N >= 1 (small N included). */
int64_t gpbo_nlml_grid_batched_workspace_bytes(int64_t N, int64_t G);
int gpbo_nlml_grid_batched_f64(const double *X, const double *y, int64_t N, int32_t d, const double *ls_cells,
                               int64_t G, double jitter, float *out, void *work, int64_t work_bytes, void *stream);
int gpbo_nlml_grid_batched_logdet_f64(const double *X, const double *y, int64_t N, int32_t d, const double *ls_cells,
                                      int64_t G, double jitter, double *out, void *work, int64_t work_bytes,
                                      void *stream);

/* One cell of the same grid for any N, from a factorisation made with (jitter1, jitter2) = (1e-4, 0):
 * log det K = -2 sum log U_ii, y^T K^-1 y = y . alpha; NaN when info != 0 (the reference's log of a negative det). */
int gpbo_nlml_cell_f64(const double *U, const double *alpha, const double *y, int64_t N, int64_t Np,
                       const int32_t *info, float *out, void *stream);

/* Strided-batched fp64 MFMA GEMM used by the factorisation (exported for tests):
 * C_b = alpha * A_b * op(B_b) + beta * C_b, row-major, M and N multiples of 64, K a multiple of 16;
 * transB = 0: B is [K x N]; transB = 1: B is [N x K].  lower_only = 1 skips 64x64 tiles strictly above
 * the block diagonal (SYRK-style update of a lower triangle). */
int gpbo_gemm_f64(int32_t transB, int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda,
                  int64_t strideA, const double *B, int64_t ldb, int64_t strideB, double beta, double *C,
                  int64_t ldc, int64_t strideC, int32_t batch, int32_t lower_only, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GPBO_H */
