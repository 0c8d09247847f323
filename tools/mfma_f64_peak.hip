// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate on MI355X (operands in registers, no memory).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o gpurun_out/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(512) void k(double *out, int iters, double seed) {
    d4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4_t{0, 0, 0, 0};
    double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int threads, int blocks, int iters) {
    double *out;
    hipMalloc(&out, sizeof(double) * threads * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, 0.37);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, 0.37);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double flops = 2048.0 * NACC * (double)iters * (threads / 64) * blocks;
    printf("NACC=%d threads=%d blocks=%d iters=%d: %.3f ms  %.2f TFLOP/s\n", NACC, threads, blocks, iters, best,
           flops / best / 1e9);
    hipFree(out);
}

int main() {
    run<16>(256, 256, 20000);   // 1 wave per SIMD
    run<16>(512, 256, 20000);   // 2 waves per SIMD
    run<16>(512, 512, 10000);   // 2 rounds of 2 waves per SIMD
    run<4>(256, 256, 80000);
    run<1>(256, 256, 80000);    // dependent chain
    run<2>(256, 256, 80000);
    return 0;
}
