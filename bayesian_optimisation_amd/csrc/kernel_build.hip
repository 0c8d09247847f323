// ARD squared-exponential covariance builds (SURVEY.md §2.2 K1, K2+K5).
//
// Replaces PointSelector.kernel_rbf (/root/reference/point_selector.py:166-195): the reference
// broadcasts an (n1, n2, d) temporary; here one thread owns a column (a candidate, or an observed
// point for K(X,X)), keeps its d coordinates in VGPRs, and walks the other set of points, whose
// coordinates are wave-uniform and therefore travel through the scalar cache into SGPRs.
// Per entry the arithmetic follows the reference's order: subtract, square, scale by 1/ls_k^2
// (the reference divides by ls_k^2; multiplying by the host-rounded reciprocal differs by <= 1 ulp
// per term), sum over features in index order, times -0.5, exp.
#include "gpbo_internal.h"

struct LsArgs {
    double il2[GPBO_MAX_D];  // 1 / ls_k^2, computed on the host in fp64
};

template <int D>
__device__ __forceinline__ double sqdist(const double (&xc)[D], const double *__restrict__ xo, const LsArgs &ls) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const double diff = xc[k] - xo[k];
        acc = fma(diff * diff, ls.il2[k], acc);
    }
    return acc;
}

// ---------------------------------------------------------------------------------------------
// K1: Kp[Np x Np] = k(X, X) with diagonal (1 + j1) + j2; identity on the padding.
// grid (ceil(Np/256), Np/64), block 256: thread = column j, block row-slice of 64 rows.
// ---------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void kxx_kernel(const double *__restrict__ X, int N, LsArgs ls, double j1,
                                                  double j2, double *__restrict__ K, int Np) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= Np) return;
    const int i0 = blockIdx.y * 64;
    double xj[D];
#pragma unroll
    for (int k = 0; k < D; ++k) xj[k] = (j < N) ? X[(int64_t)j * D + k] : 0.0;
    for (int i = i0; i < i0 + 64; ++i) {
        double v;
        if (i < N) {  // wave-uniform
            const double acc = sqdist<D>(xj, X + (int64_t)i * D, ls);
            v = exp(-0.5 * acc);
            if (i == j) v = (v + j1) + j2;  // reference: kernel_rbf adds 1e-4 (:193), assembly adds 1e-6 (:79)
            if (j >= N) v = 0.0;
        } else {
            v = (i == j) ? 1.0 : 0.0;
        }
        K[(int64_t)i * Np + j] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// K2+K5: KsT[n][c] = k(x_n, x*_c) for one candidate chunk, plus per-slice partial means.
// grid (ldk_used/512, Np/128), block 256: thread = two adjacent candidates (16-byte stores, two
// independent exp chains), blockIdx.y = slice of 128 observations.
// ---------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void kstar_mu_kernel(const double *__restrict__ Xs, int64_t Mc,
                                                       const double *__restrict__ X, int N, LsArgs ls,
                                                       const double *__restrict__ alpha, double diag_add,
                                                       int64_t cand_base, double *__restrict__ KsT, int64_t ldk,
                                                       double *__restrict__ mu_part) {
    const int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
    const int n0 = blockIdx.y * 128;
    double xa[D], xb[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        xa[k] = (c0 < Mc) ? Xs[c0 * D + k] : 0.0;
        xb[k] = (c0 + 1 < Mc) ? Xs[(c0 + 1) * D + k] : 0.0;
    }
    double mua = 0.0, mub = 0.0;
    const bool has_diag = diag_add != 0.0;  // uniform
#pragma unroll 2
    for (int n = n0; n < n0 + 128; ++n) {
        d2_t kv = {0.0, 0.0};
        if (n < N) {  // wave-uniform: padded observations contribute exact zeros
            const double *xo = X + (int64_t)n * D;
            const double sa = sqdist<D>(xa, xo, ls);
            const double sb = sqdist<D>(xb, xo, ls);
            double ka = exp(-0.5 * sa);
            double kb = exp(-0.5 * sb);
            if (has_diag) {  // N == M shape-coincidence quirk (point_selector.py:173,191-193)
                if ((int64_t)n == cand_base + c0) ka += diag_add;
                if ((int64_t)n == cand_base + c0 + 1) kb += diag_add;
            }
            const double an = alpha[n];
            mua = fma(ka, an, mua);
            mub = fma(kb, an, mub);
            kv.x = ka;
            kv.y = kb;
        }
        *reinterpret_cast<d2_t *>(KsT + (int64_t)n * ldk + c0) = kv;
    }
    d2_t m = {mua, mub};
    *reinterpret_cast<d2_t *>(mu_part + (int64_t)blockIdx.y * ldk + c0) = m;
}

static int make_ls(const double *ls_host, int d, LsArgs *out) {
    if (!ls_host || d < 1 || d > GPBO_MAX_D) return GPBO_ERR_ARG;
    for (int k = 0; k < GPBO_MAX_D; ++k) out->il2[k] = 0.0;
    for (int k = 0; k < d; ++k) {
        const double l = ls_host[k];
        if (!(l > 0.0)) return GPBO_ERR_ARG;
        out->il2[k] = 1.0 / (l * l);
    }
    return GPBO_OK;
}

#define GPBO_DISPATCH_D(d, CALL) \
    switch (d) {                 \
        case 1: CALL(1); break;  \
        case 2: CALL(2); break;  \
        case 3: CALL(3); break;  \
        case 4: CALL(4); break;  \
        case 5: CALL(5); break;  \
        case 6: CALL(6); break;  \
        case 7: CALL(7); break;  \
        case 8: CALL(8); break;  \
        case 9: CALL(9); break;  \
        case 10: CALL(10); break; \
        case 11: CALL(11); break; \
        case 12: CALL(12); break; \
        case 13: CALL(13); break; \
        case 14: CALL(14); break; \
        case 15: CALL(15); break; \
        case 16: CALL(16); break; \
        default: return GPBO_ERR_ARG; \
    }

extern "C" int gpbo_kxx_f64(const double *X, int64_t N, int32_t d, const double *ls_host, double jitter1,
                            double jitter2, double *Kp, int64_t Np, void *stream) {
    if (!X || !Kp || N < 1 || Np < N || Np % 64 != 0 || Np > (1 << 20)) return GPBO_ERR_ARG;
    LsArgs ls;
    int rc = make_ls(ls_host, d, &ls);
    if (rc != GPBO_OK) return rc;
    dim3 grid((unsigned)((Np + 255) / 256), (unsigned)(Np / 64));
#define CALL(DD) \
    hipLaunchKernelGGL(kxx_kernel<DD>, grid, dim3(256), 0, gpbo_stream(stream), X, (int)N, ls, jitter1, jitter2, Kp, (int)Np)
    GPBO_DISPATCH_D(d, CALL)
#undef CALL
    GPBO_CHECK_LAUNCH();
    return GPBO_OK;
}

extern "C" int gpbo_kstar_mu_f64(const double *Xs, int64_t Mc, const double *X, int64_t N, int64_t Np, int32_t d,
                                 const double *ls_host, const double *alpha, double diag_add, int64_t cand_base,
                                 double *KsT, int64_t ldk, double *mu_part, void *stream) {
    if (!Xs || !X || !alpha || !KsT || !mu_part) return GPBO_ERR_ARG;
    if (Mc < 1 || N < 1 || Np < N || Np % 128 != 0 || ldk % GPBO_CHUNK_GRANULE != 0 || Mc > ldk)
        return GPBO_ERR_ARG;
    LsArgs ls;
    int rc = make_ls(ls_host, d, &ls);
    if (rc != GPBO_OK) return rc;
    const int64_t used = (Mc + GPBO_CHUNK_GRANULE - 1) / GPBO_CHUNK_GRANULE * GPBO_CHUNK_GRANULE;
    dim3 grid((unsigned)(used / 512), (unsigned)(Np / 128));
#define CALL(DD)                                                                                              \
    hipLaunchKernelGGL(kstar_mu_kernel<DD>, grid, dim3(256), 0, gpbo_stream(stream), Xs, Mc, X, (int)N, ls, alpha, \
                       diag_add, cand_base, KsT, ldk, mu_part)
    GPBO_DISPATCH_D(d, CALL)
#undef CALL
    GPBO_CHECK_LAUNCH();
    return GPBO_OK;
}
