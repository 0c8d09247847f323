"""TFLOP/s of the factorisation's fp64 MFMA GEMM (gpbo_gemm_f64) on the shapes the Cholesky / triangular inverse use.
M a multiple of 128 takes the 128 x 128 tile, M = 64 (2k+1) the 64 x 64 tile: python tools/bench_gemm.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesian_optimisation_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
for transB, M, K, lower in [(1, 8192, 256, 1), (1, 8128, 256, 1), (1, 4096, 128, 1), (1, 4032, 128, 1), (0, 2048, 2048, 0),
                            (0, 1984, 1984, 0), (0, 4096, 4096, 0), (0, 4032, 4032, 0)]:
    N = M
    A = torch.randn(M, K, dtype=torch.float64, device=dev)
    B = torch.randn((N, K) if transB else (K, N), dtype=torch.float64, device=dev)
    Cm = torch.zeros(M, N, dtype=torch.float64, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def run():
        rc = lib.gpbo_gemm_f64(transB, M, N, K, 1.0, C.c_void_p(A.data_ptr()), K, 0, C.c_void_p(B.data_ptr()), K if transB else N, 0,
                               0.0, C.c_void_p(Cm.data_ptr()), N, 0, 1, lower, st)
        assert rc == 0
    for _ in range(3): run()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    flop = 2.0 * M * N * K * (0.5 if lower else 1.0)
    print(f"transB={transB} M=N={M} K={K} lower={lower}: {dt*1e3:.3f} ms  {flop/dt/1e12:.1f} TFLOP/s", flush=True)
