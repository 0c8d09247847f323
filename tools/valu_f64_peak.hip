// Microbenchmark: sustained v_fma_f64 (vector fp64) rate on MI355X, and v_add/v_mul mix.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NCH, int MODE>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a, double b) {
    double x[NCH];
    for (int i = 0; i < NCH; ++i) x[i] = a + i * 1e-3 + threadIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (MODE == 0) x[i] = fma(x[i], b, a);
            else if (MODE == 1) x[i] = x[i] + b;
            else x[i] = x[i] * b;
        }
    }
    double s = 0;
    for (int i = 0; i < NCH; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NCH, int MODE>
void run(int blocks, int iters, const char *name) {
    double *out; hipMalloc(&out, sizeof(double) * 256 * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NCH, MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, 0.5, 0.999999);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NCH, MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, 0.5, 0.999999);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double ops = (double)NCH * iters * 256.0 * blocks;
    printf("%s NCH=%d blocks=%d: %.3f ms, %.2f T instr-lanes/s (x2 = %.1f TFLOP/s for fma)\n", name, NCH, blocks, best,
           ops / best / 1e9, 2 * ops / best / 1e9);
    hipFree(out);
}
int main() {
    run<8, 0>(2048, 20000, "fma");   // 8 waves/SIMD
    run<8, 0>(1024, 20000, "fma");   // 4 waves/SIMD
    run<2, 0>(2048, 80000, "fma");
    run<1, 0>(2048, 80000, "fma");   // dependent chain, 8 waves/SIMD
    run<1, 0>(256, 80000, "fma");    // dependent chain, 1 wave/SIMD
    run<8, 1>(2048, 20000, "add");
    run<8, 2>(2048, 20000, "mul");
    return 0;
}
