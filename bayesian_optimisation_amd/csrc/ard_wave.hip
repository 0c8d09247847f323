// ARD likelihood grid at the reference's OWN sizes (N <= 64 observations): one WAVE per grid cell, the matrix in registers.
//
// Replaces PointSelector.tune_kernel / eval_log_marginal (/root/reference/point_selector.py:104-163) like ard.hip does:
//     nlml = 0.5 * (y^T inv(K) y + log(det(K)) + N log(2 pi)),  K = kernel_rbf(X, X) (1e-4 jitter, :116,193), float32.
// The in-LDS kernel of round 2 (ard.hip, nlml_grid_kernel: a workgroup per cell, one barrier and one LDS round trip per
// column, index arithmetic with integer divisions, library exp / sqrt / division) needs 0.116 ms for the reference's 50 x 50
// grid at N = 32; the DAG runs exactly this shape once per iteration.  Here lane i of a wave holds row i of the cell's K in
// registers (x[k] = K[i][k], k < NMAX), the column steps of a right-looking Cholesky are fully unrolled, the pivot and the
// multipliers L[k][c] travel by v_readlane, and y rides along as a right-hand side (b_i -> z = L^-1 y by forward
// substitution in the same steps): no LDS traffic beyond the exp table, no barrier after the first, four cells per workgroup.
// Work per cell: N^2 / 2 fused multiply-adds per lane-instruction, i.e. N^2 / 2 wave instructions (+ two readlanes each) -
// the upper triangle is carried along unused (lanes i < k hold garbage in x[k]; nothing reads it).
// Rows / columns beyond N are the identity (pivot 1, log 1 = 0), lanes beyond NMAX idle.  NMAX = 16 / 32 / 48 / 64.
// det under- / overflow of the reference (np.log(np.linalg.det(K)), :117-119) as in ard.hip: log(exp(logdet)); the second
// likelihood mode (fp64, log det straight from the factor: gpbo.h) is a kernel argument.
// Measured (one MI355X, 2,500 cells, d = 2 / 8 / 16, ms): N = 16: 0.014 / 0.016 / 0.023, N = 32: 0.026 / 0.031 / 0.038
// (in-LDS kernel: 0.044 and 0.116 at d = 2), N = 48: 0.049 / 0.065 / 0.075, N = 64: 0.068 / 0.083 / 0.096 (the fused kernel of
// ard.hip, which serves every larger N: 0.122 / 0.140 / 0.178); the float32 cells of all three kernels are equal.
#include "gpbo_internal.h"
#include "exp_neg.h"
#include "potrf_diag64.h"

#include <cmath>

#ifndef GPBO_WAVE_MAX_N
#define GPBO_WAVE_MAX_N 64
#endif

namespace {

using gpbo_pd::readlane_f64;
using gpbo_pd::rsqrt_refined;

template <int NMAX, int D>
__global__ __launch_bounds__(256, NMAX > 32 ? 2 : (D == 16 ? 3 : 4)) void nlml_wave_kernel(const double *__restrict__ X, const double *__restrict__ y, int N, int d,
                                                        const double *__restrict__ ls_cells, int G, double jitter,
                                                        void *__restrict__ out, int logdet_mode) {
    __shared__ double tab[GPBO_EXP_E];
    __shared__ double Xs[NMAX * D];   // the observations, padded with zeros to NMAX rows of D features (the same for the four cells)
    __shared__ double ys[NMAX];
    if (threadIdx.x < GPBO_EXP_E) tab[threadIdx.x] = kExp2Tab256[threadIdx.x * (256 / GPBO_EXP_E)];
    for (int e = threadIdx.x; e < NMAX * D; e += 256) {
        const int r = e / D, q = e % D;   // (D is a power of two)
        Xs[e] = (r < N && q < d) ? X[r * d + q] : 0.0;
    }
    if (threadIdx.x < NMAX) ys[threadIdx.x] = ((int)threadIdx.x < N) ? y[threadIdx.x] : 0.0;
    gpbo_syncthreads();
    const int lane = threadIdx.x & 63;
    const int g = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);   // this wave's cell
    if (g >= G) return;

    const int li = ((NMAX & (NMAX - 1)) == 0) ? (lane & (NMAX - 1)) : (lane >= NMAX ? lane - NMAX : lane);   // a row < NMAX for every lane
    double il2[D], xi[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const double l = (k < d) ? ls_cells[(int64_t)g * d + k] : 1.0;        // wave-uniform
        il2[k] = (k < d) ? 1.0 / (l * l) : 0.0;
        xi[k] = Xs[li * D + k];                                               // this lane's row
    }
    // K[i][k] for the lane's row i: column k's coordinates are wave-uniform (one LDS address for the wave)
    double x[NMAX];
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < D; ++q) {
            const double diff = Xs[k * D + q] - xi[q];
            a = fma(diff * diff, il2[q], a);
        }
        double v = exp_neg(0.5 * a, tab);
        if (k == lane) v += jitter;
        if (lane >= N || k >= N) v = (k == lane) ? 1.0 : 0.0;
        x[k] = v;
        // two entries in flight at a time: left to itself the scheduler starts all NMAX chains at once (264 registers)
        if (k & 1) __builtin_amdgcn_sched_barrier(0);
    }
    // (a plain load and a select: written as one conditional expression, the register allocation of the WHOLE kernel
    //  collapses - 124 spilled registers at NMAX = 32 - and the launch moves 157 MB of scratch)
    double b = ys[li];
    if (lane >= NMAX) b = 0.0;

    double quad = 0.0, lcc = 1.0;
    bool bad = false;
#pragma unroll
    for (int c = 0; c < NMAX; ++c) {
        const double piv = readlane_f64(x[c], c);
        bad |= !(piv > 0.0) | !(piv < 1.0e300);
        const double r = rsqrt_refined(piv);
        const double lc = x[c] * r;                   // lane i >= c: L[i][c]
        if (lane == c) lcc = lc;                      // L[c][c], for log det
        const double zc = readlane_f64(b, c) * r;     // z_c = (y_c - sum_{k<c} L[c][k] z_k) / L[c][c]
        quad = fma(zc, zc, quad);
        b = fma(-lc, zc, b);
#pragma unroll
        for (int k = c + 1; k < NMAX; ++k) x[k] = fma(-lc, readlane_f64(lc, k), x[k]);
    }
    double logdet = log(lcc);                         // one log per lane, all at once (1 on the padded lanes)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) logdet += __shfl_xor(logdet, off);
    logdet *= 2.0;
    if (lane == 0) {
        // reference mode: the reference takes log of a det that under- / overflows
        const double ld = logdet_mode ? logdet : log(exp(logdet));
        double nlml = 0.5 * (quad + ld + (double)N * 1.8378770664093453);   // log(2 pi)
        if (bad) nlml = __builtin_nan("");            // not positive definite: the reference's log(det < 0) is NaN
        if (logdet_mode) reinterpret_cast<double *>(out)[g] = nlml;
        else reinterpret_cast<float *>(out)[g] = (float)nlml;
    }
}

template <int NMAX>
int launch_d(const double *X, const double *y, int64_t N, int32_t d, const double *ls_cells, int64_t G, double jitter,
             void *out, int mode, hipStream_t st) {
    const unsigned grid = (unsigned)((G + 3) / 4);
#define GPBO_WAVE_LAUNCH(DD)                                                                                          \
    hipLaunchKernelGGL((nlml_wave_kernel<NMAX, DD>), dim3(grid), dim3(256), 0, st, X, y, (int)N, (int)d, ls_cells, (int)G, \
                       jitter, out, mode)
    if (d <= 2) GPBO_WAVE_LAUNCH(2);
    else if (d <= 4) GPBO_WAVE_LAUNCH(4);
    else if (d <= 8) GPBO_WAVE_LAUNCH(8);
    else GPBO_WAVE_LAUNCH(16);
#undef GPBO_WAVE_LAUNCH
    GPBO_CHECK_LAUNCH();
    return GPBO_OK;
}

}  // namespace

constexpr int WAVE_MAX_N = GPBO_WAVE_MAX_N;
extern "C" int gpbo_nlml_grid_wave_max_n(void) { return WAVE_MAX_N; }

static int wave_run(const double *X, const double *y, int64_t N, int32_t d, const double *ls_cells, int64_t G, double jitter,
                    void *out, int mode, void *stream) {
    if (!X || !y || !ls_cells || !out || N < 1 || N > WAVE_MAX_N || d < 1 || d > GPBO_MAX_D || G < 1 || G > (1 << 30))
        return GPBO_ERR_ARG;
    hipStream_t st = gpbo_stream(stream);
    if (N <= 16) return launch_d<16>(X, y, N, d, ls_cells, G, jitter, out, mode, st);
    if (N <= 32) return launch_d<32>(X, y, N, d, ls_cells, G, jitter, out, mode, st);
#if GPBO_WAVE_MAX_N > 32
    if (N <= 48) return launch_d<48>(X, y, N, d, ls_cells, G, jitter, out, mode, st);
    return launch_d<64>(X, y, N, d, ls_cells, G, jitter, out, mode, st);
#else
    return GPBO_ERR_ARG;
#endif
}

extern "C" int gpbo_nlml_grid_wave_f64(const double *X, const double *y, int64_t N, int32_t d, const double *ls_cells,
                                       int64_t G, double jitter, float *out, void *stream) {
    return wave_run(X, y, N, d, ls_cells, G, jitter, out, 0, stream);
}

extern "C" int gpbo_nlml_grid_wave_logdet_f64(const double *X, const double *y, int64_t N, int32_t d, const double *ls_cells,
                                              int64_t G, double jitter, double *out, void *stream) {
    return wave_run(X, y, N, d, ls_cells, G, jitter, out, 1, stream);
}
