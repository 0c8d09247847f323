// Host-pointer entry points: one call = one SELECT_PARAMETERS surrogate step.
//
// The reference's boundary is a Python attribute protocol fed with NumPy arrays
// (/root/reference/select_parameters.py:146-158, 282-294 -> /root/reference/point_selector.py:42-102, 197-207).
// These two functions take exactly those arrays as plain host pointers, so a maintainer can bind the GPU path
// with ctypes + NumPy alone (no PyTorch, no device-memory handling on the caller's side): device buffers are
// allocated, filled, used and released inside the call, on the library's own stream.
//   gpbo_select_next_host_f64  = update_surrogate() after the length scales are chosen + the acquisition arg-max
//   gpbo_nlml_grid_host_f64    = tune_kernel()'s likelihood grid
#include "gpbo_internal.h"

#include <vector>

namespace {

// Device allocations of one call, released on every exit path.
struct DeviceArena {
    std::vector<void *> ptrs;
    hipStream_t stream = nullptr;
    bool ok = true;
    DeviceArena() { ok = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) == hipSuccess; }
    ~DeviceArena() {
        if (stream) {
            (void)hipStreamSynchronize(stream);
            (void)hipStreamDestroy(stream);
        }
        for (void *p : ptrs) (void)hipFree(p);
    }
    template <typename T>
    T *alloc(int64_t count) {
        void *p = nullptr;
        if (count < 1) count = 1;
        if (hipMalloc(&p, sizeof(T) * (size_t)count) != hipSuccess) {
            ok = false;
            return nullptr;
        }
        ptrs.push_back(p);
        return reinterpret_cast<T *>(p);
    }
    bool h2d(void *dst, const void *src, size_t bytes) {
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream) == hipSuccess;
    }
    bool d2h(void *dst, const void *src, size_t bytes) {
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream) == hipSuccess;
    }
    bool sync() { return hipStreamSynchronize(stream) == hipSuccess; }
};

}  // namespace

extern "C" int gpbo_select_next_host_f64(const double *X, const double *y, int64_t N, int32_t d, const double *ls,
                                         double jitter1, double jitter2, const double *Xs, int64_t M, int32_t acq_kind,
                                         double p0, double p1, double diag_add, int64_t chunk, double *mu_out,
                                         double *sigma_out, double *acq_out, double *cov_meas_out, gpbo_result *result,
                                         int32_t *info) {
    if (!X || !y || !ls || !Xs || !result || !info) return GPBO_ERR_ARG;
    if (N < 1 || M < 1 || d < 1 || d > GPBO_MAX_D_ANY) return GPBO_ERR_ARG;
    if (acq_kind != GPBO_ACQ_LCB && acq_kind != GPBO_ACQ_EI) return GPBO_ERR_ARG;
    if (chunk == 0) chunk = (int64_t)1 << 17;
    if (chunk < GPBO_CHUNK_GRANULE || chunk % GPBO_CHUNK_GRANULE || chunk > GPBO_CHUNK_MAX) return GPBO_ERR_ARG;
    for (int k = 0; k < d; ++k)
        if (!(ls[k] > 0.0)) return GPBO_ERR_ARG;
    {  // no more chunk than the candidates need
        const int64_t need = (M + GPBO_CHUNK_GRANULE - 1) / GPBO_CHUNK_GRANULE * GPBO_CHUNK_GRANULE;
        if (chunk > need) chunk = need;
    }
    const int64_t Np = gpbo_padded_n(N);
    const int64_t wfact = gpbo_factorise_workspace_bytes(Np);
    const int64_t wpost = gpbo_posterior_workspace_bytes(Np, chunk, M);
    if (wpost < 0) return GPBO_ERR_ARG;

    DeviceArena A;
    if (!A.ok) return GPBO_ERR_LAUNCH;
    double *dX = A.alloc<double>(N * d), *dy = A.alloc<double>(N), *dXs = A.alloc<double>(M * d);   // (dX / dy: re-pointed below)
    double *dK = A.alloc<double>(Np * Np), *dU = A.alloc<double>(Np * Np), *dalpha = A.alloc<double>(Np);
    int32_t *dinfo = A.alloc<int32_t>(1);
    gpbo_result *dres = A.alloc<gpbo_result>(1);
    // the factorisation workspace is dead once U and alpha exist: the posterior workspace reuses the allocation
    const int64_t wbytes = (wfact > wpost ? wfact : wpost) + 256;
    char *dwork = A.alloc<char>(wbytes);
    const bool dense = mu_out || sigma_out || acq_out;
    double *dmu = dense ? A.alloc<double>(M) : nullptr;
    double *dsig = dense ? A.alloc<double>(M) : nullptr;
    double *dacq = dense ? A.alloc<double>(M) : nullptr;
    // A caller that asks for the next point only (no dense arrays) gets it by branch and bound on the exact prefix bound
    // (same point, same NaN count, the plain pass when the bound does not separate the candidates); the same
    // rule and the same prefix lengths as DeviceGP.score_bound.
    // (jitter: with K = k(X,X) + tau I the true variance is >= tau, so the plain pass's sqrt(|var|) never reflects a
    //  negative value unless its rounding error exceeds tau - the one case a prefix cannot bound; DeviceGP.BOUND_MIN_JITTER)
    const bool bound_route = !dense && diag_add == 0.0 && M >= 32768 && Np >= 1024 && d <= GPBO_MAX_D &&
                             (acq_kind == GPBO_ACQ_EI || p0 >= 0.0) && jitter1 + jitter2 >= 1e-6;
    const int64_t J1 = (Np / 16) / 128 * 128 < 128 ? 128 : (Np / 16) / 128 * 128, J2 = (8 * J1 <= Np) ? 4 * J1 : 0;
    // every first-level survivor may go on to the second-level bound (1/16 of a plain pass per candidate); the plain pass
    // takes over when more than M / 8 reach the fp64 kernels (rescore.hip)
    const int64_t bcap = M < 4096 ? 4096 : M;
    const int64_t bchunk = 1 << 14;
    const int64_t wresc = bound_route ? gpbo_rescore_workspace_bytes(Np, bcap, bchunk) : 0;
    if (wresc < 0) return GPBO_ERR_ARG;
    double *dub = bound_route ? A.alloc<double>(M) : nullptr;
    char *dresc = bound_route ? A.alloc<char>(wresc + 256) : nullptr;
    // the bound prunes by the FIRST J1 observations of the factorised problem: factorise the farthest-point order of the
    // observations (subset.hip), so that the pruning does not depend on the order in which they arrived.  cov_meas_out is
    // the reference's matrix in ARRIVAL order: a caller who wants it gets the arrival-order factorisation.
    const bool fps_order = bound_route && J1 < N && !cov_meas_out;
    const int64_t word = fps_order ? gpbo_fps_order_workspace_bytes(N) : 0;
    int64_t *dperm = fps_order ? A.alloc<int64_t>(N) : nullptr;
    double *dXp = fps_order ? A.alloc<double>(N * d) : nullptr;
    double *dyp = fps_order ? A.alloc<double>(N) : nullptr;
    char *dword = fps_order ? A.alloc<char>(word + 256) : nullptr;
    if (!A.ok) return GPBO_ERR_WORKSPACE;
    void *st = reinterpret_cast<void *>(A.stream);

    if (!A.h2d(dX, X, sizeof(double) * N * d) || !A.h2d(dy, y, sizeof(double) * N) ||
        !A.h2d(dXs, Xs, sizeof(double) * M * d))
        return GPBO_ERR_LAUNCH;
    int rc;
    if (fps_order) {
        char *wo = reinterpret_cast<char *>(((uintptr_t)dword + 255) & ~(uintptr_t)255);
        rc = gpbo_fps_order_f64(dX, dy, N, d, ls, J1, dperm, dXp, dyp, wo, word, st);
        if (rc != GPBO_OK) return rc;
        dX = dXp;   // every later pass reads the observations in the factorisation's order
        dy = dyp;
    }
    rc = gpbo_factorise_f64(dX, dy, N, d, ls, jitter1, jitter2, Np, dK, dU, dalpha, dinfo, dwork, wfact, st);
    if (rc != GPBO_OK) return rc;
    if (!A.d2h(info, dinfo, sizeof(int32_t)) || !A.sync()) return GPBO_ERR_LAUNCH;
    if (cov_meas_out) {  // the reference's cov_meas attribute (point_selector.py:79), N x N without the padding
        if (hipMemcpy2DAsync(cov_meas_out, sizeof(double) * N, dK, sizeof(double) * Np, sizeof(double) * N, (size_t)N,
                             hipMemcpyDeviceToHost, A.stream) != hipSuccess)
            return GPBO_ERR_LAUNCH;
    }
    if (*info != 0) {  // not positive definite: nothing to score (the reference's inv() raises or returns garbage)
        if (fps_order && *info >= 1 && *info <= N) {   // report the failing observation by its ARRIVAL index, as documented
            int64_t row = 0;
            if (!A.d2h(&row, dperm + (*info - 1), sizeof(int64_t)) || !A.sync()) return GPBO_ERR_LAUNCH;
            *info = (int32_t)(row + 1);
        }
        result->best_val = 0.0;
        result->best_idx = -1;
        result->nan_count = 0;
        result->reserved = 0;
        return A.sync() ? GPBO_OK : GPBO_ERR_LAUNCH;
    }
    const double prior_var = (1.0 + jitter1) + jitter2;  // diagonal of cov_pred as the reference rounds it
    bool decided = false;
    if (bound_route) {
        rc = gpbo_posterior_prefix_f64(dXs, M, dX, N, Np, d, ls, dU, dalpha, prior_var, acq_kind, p0, p1, 0, chunk, J1,
                                       nullptr, nullptr, dub, dres, dwork, wpost, nullptr, st);
        if (rc != GPBO_OK) return rc;
        gpbo_screen_stats stats;
        char *wr = reinterpret_cast<char *>(((uintptr_t)dresc + 255) & ~(uintptr_t)255);
        int64_t stride = M / 1024;
        if (stride < 1) stride = 1;
        rc = gpbo_bound_select_f64(dXs, M, dub, dX, N, Np, d, ls, dU, dalpha, prior_var, acq_kind, p0, p1, 0, stride, bcap,
                                   bchunk, J2, dres, &stats, wr, wresc, st);
        if (rc != GPBO_OK) return rc;
        decided = !stats.fallback;
    }
    if (!decided) {
        rc = gpbo_posterior_acq_f64(dXs, M, dX, N, Np, d, ls, dU, dalpha, prior_var, acq_kind, p0, p1, diag_add, 0, chunk,
                                    dmu, dsig, dacq, dres, dwork, wpost, nullptr, st);
        if (rc != GPBO_OK) return rc;
    }
    bool okc = A.d2h(result, dres, sizeof(gpbo_result));
    if (mu_out) okc = okc && A.d2h(mu_out, dmu, sizeof(double) * M);
    if (sigma_out) okc = okc && A.d2h(sigma_out, dsig, sizeof(double) * M);
    if (acq_out) okc = okc && A.d2h(acq_out, dacq, sizeof(double) * M);
    if (!okc || !A.sync()) return GPBO_ERR_LAUNCH;
    return GPBO_OK;
}

// q = 8 Monte-Carlo qEI on host arrays (same conventions as gpbo_select_next_host_f64; Z: [S x 8] base samples)
extern "C" int gpbo_select_qei_host_f64(const double *X, const double *y, int64_t N, int32_t d, const double *ls,
                                        double jitter1, double jitter2, const double *Xs, int64_t M, double f_best,
                                        double xi, const double *Z, int32_t S, int64_t chunk, double *qei_out,
                                        gpbo_result *result, int32_t *info) {
    if (!X || !y || !ls || !Xs || !Z || !result || !info) return GPBO_ERR_ARG;
    if (N < 1 || M < 8 || M % 8 || d < 1 || d > GPBO_MAX_D || S < 1) return GPBO_ERR_ARG;
    if (chunk == 0) chunk = (int64_t)1 << 15;
    if (chunk < GPBO_CHUNK_GRANULE || chunk % GPBO_CHUNK_GRANULE || chunk > GPBO_CHUNK_MAX) return GPBO_ERR_ARG;
    for (int k = 0; k < d; ++k)
        if (!(ls[k] > 0.0)) return GPBO_ERR_ARG;
    {
        const int64_t need = (M + GPBO_CHUNK_GRANULE - 1) / GPBO_CHUNK_GRANULE * GPBO_CHUNK_GRANULE;
        if (chunk > need) chunk = need;
    }
    const int64_t Np = gpbo_padded_n(N);
    const int64_t wfact = gpbo_factorise_workspace_bytes(Np);
    const int64_t wq = gpbo_qei_workspace_bytes(Np, chunk, M);
    if (wq < 0) return GPBO_ERR_ARG;
    DeviceArena A;
    if (!A.ok) return GPBO_ERR_LAUNCH;
    double *dX = A.alloc<double>(N * d), *dy = A.alloc<double>(N), *dXs = A.alloc<double>(M * d);
    double *dZ = A.alloc<double>((int64_t)S * 8);
    double *dK = A.alloc<double>(Np * Np), *dU = A.alloc<double>(Np * Np), *dalpha = A.alloc<double>(Np);
    int32_t *dinfo = A.alloc<int32_t>(1);
    gpbo_result *dres = A.alloc<gpbo_result>(1);
    char *dwork = A.alloc<char>((wfact > wq ? wfact : wq) + 256);
    double *dq = qei_out ? A.alloc<double>(M / 8) : nullptr;
    if (!A.ok) return GPBO_ERR_WORKSPACE;
    void *st = reinterpret_cast<void *>(A.stream);
    if (!A.h2d(dX, X, sizeof(double) * N * d) || !A.h2d(dy, y, sizeof(double) * N) ||
        !A.h2d(dXs, Xs, sizeof(double) * M * d) || !A.h2d(dZ, Z, sizeof(double) * S * 8))
        return GPBO_ERR_LAUNCH;
    int rc = gpbo_factorise_f64(dX, dy, N, d, ls, jitter1, jitter2, Np, dK, dU, dalpha, dinfo, dwork, wfact, st);
    if (rc != GPBO_OK) return rc;
    if (!A.d2h(info, dinfo, sizeof(int32_t)) || !A.sync()) return GPBO_ERR_LAUNCH;
    if (*info != 0) {
        result->best_val = 0.0;
        result->best_idx = -1;
        result->nan_count = 0;
        result->reserved = 0;
        return GPBO_OK;
    }
    const double prior_var = (1.0 + jitter1) + jitter2;
    rc = gpbo_posterior_qei_f64(dXs, M, dX, N, Np, d, ls, dU, dalpha, prior_var, f_best, xi, dZ, S, 0, chunk, dq, dres,
                                dwork, wq, nullptr, st);
    if (rc != GPBO_OK) return rc;
    bool okc = A.d2h(result, dres, sizeof(gpbo_result));
    if (qei_out) okc = okc && A.d2h(qei_out, dq, sizeof(double) * (M / 8));
    if (!okc || !A.sync()) return GPBO_ERR_LAUNCH;
    return GPBO_OK;
}

// mode 0: the reference's float32 likelihood; mode 1: fp64, log det from the factor (gpbo.h)
static int nlml_grid_host(const double *X, const double *y, int64_t N, int32_t d, const double *ls_cells, int64_t G,
                          double jitter, void *out, int mode) {
    if (!X || !y || !ls_cells || !out || N < 1 || d < 1 || d > GPBO_MAX_D || G < 1 || G > (1 << 30))
        return GPBO_ERR_ARG;
    for (int64_t e = 0; e < G * d; ++e)
        if (!(ls_cells[e] > 0.0)) return GPBO_ERR_ARG;
    DeviceArena A;
    if (!A.ok) return GPBO_ERR_LAUNCH;
    const size_t osz = mode ? sizeof(double) : sizeof(float);
    double *dX = A.alloc<double>(N * d), *dy = A.alloc<double>(N);
    char *dout = A.alloc<char>((int64_t)(G * osz));
    void *st = reinterpret_cast<void *>(A.stream);
    if (!A.ok) return GPBO_ERR_WORKSPACE;
    if (!A.h2d(dX, X, sizeof(double) * N * d) || !A.h2d(dy, y, sizeof(double) * N)) return GPBO_ERR_LAUNCH;
    int rc;
    double *dcells = A.alloc<double>(G * d);
    if (!A.ok) return GPBO_ERR_WORKSPACE;
    if (!A.h2d(dcells, ls_cells, sizeof(double) * G * d)) return GPBO_ERR_LAUNCH;
    if (N <= gpbo_nlml_grid_wave_max_n()) {   // the wave-per-cell kernel (the same switch as the tensor-resident binding)
        rc = mode ? gpbo_nlml_grid_wave_logdet_f64(dX, dy, N, d, dcells, G, jitter, reinterpret_cast<double *>(dout), st)
                  : gpbo_nlml_grid_wave_f64(dX, dy, N, d, dcells, G, jitter, reinterpret_cast<float *>(dout), st);
        if (rc != GPBO_OK) return rc;
    } else {         // one persistent workgroup per cell, the whole factorisation in one launch
        const int64_t wb = gpbo_nlml_grid_batched_workspace_bytes(N, G);
        if (wb < 0) return GPBO_ERR_ARG;
        char *dwork = A.alloc<char>(wb);
        if (!A.ok) return GPBO_ERR_WORKSPACE;
        rc = mode ? gpbo_nlml_grid_batched_logdet_f64(dX, dy, N, d, dcells, G, jitter, reinterpret_cast<double *>(dout), dwork, wb, st)
                  : gpbo_nlml_grid_batched_f64(dX, dy, N, d, dcells, G, jitter, reinterpret_cast<float *>(dout), dwork, wb, st);
        if (rc != GPBO_OK) return rc;
    }
    if (!A.d2h(out, dout, G * osz) || !A.sync()) return GPBO_ERR_LAUNCH;
    return GPBO_OK;
}

extern "C" int gpbo_nlml_grid_host_f64(const double *X, const double *y, int64_t N, int32_t d, const double *ls_cells,
                                       int64_t G, double jitter, float *out) {
    return nlml_grid_host(X, y, N, d, ls_cells, G, jitter, out, 0);
}

extern "C" int gpbo_nlml_grid_logdet_host_f64(const double *X, const double *y, int64_t N, int32_t d,
                                              const double *ls_cells, int64_t G, double jitter, double *out) {
    return nlml_grid_host(X, y, N, d, ls_cells, G, jitter, out, 1);
}
