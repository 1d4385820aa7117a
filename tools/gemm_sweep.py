"""GEMM micro-benchmark (developer tool): per-shape time of k_gemm_bf16 through mi355_gemm_bf16."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import os, sys, time, torch
import imageretrievalresearch_amd as M
from imageretrievalresearch_amd._lib import lib, check, stream_ptr
dev = "cuda:0"
shapes = [(12544, 232, 1408), (12544, 232, 1392), (12544, 384, 2304), (50176, 136, 832), (50176, 136, 816), (50176, 96, 576),
          (401408, 384, 128), (401408, 128, 128), (401408, 512, 128), (401408, 128, 512), (100352, 768, 256), (100352, 1024, 256), (25088, 1536, 512), (25088, 2048, 512), (25088, 512, 2048), (6272, 3072, 1024), (6272, 4096, 1024), (6272, 1024, 4096), (12544, 1536, 384), (50176, 96, 576),
          (12544, 1392, 232), (12544, 232, 1392), (50176, 576, 96), (50176, 96, 576), (50176, 816, 136), (50176, 136, 816),
          (802816, 192, 32), (802816, 32, 192), (3211264, 144, 24), (12544, 1536, 384), (12544, 192, 32), (12544, 192, 1392),
          (1568, 1392, 232), (128, 192, 32), (128, 192, 1392)]
if os.environ.get("SHAPES"):
    shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["SHAPES"].split(",")]
for (Mm, N, K) in shapes:
    ldw = (K + 31) // 32 * 32
    Np = (N + 15) // 16 * 16
    A = (torch.randn(Mm, K, device=dev) * 0.1).bfloat16()
    W = torch.zeros(Np, ldw, device=dev, dtype=torch.bfloat16)
    W[:N, :K] = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.zeros(Np, device=dev)
    out = torch.empty(Mm, N, device=dev, dtype=torch.bfloat16)
    def run():
        check(lib().mi355_gemm_bf16(A.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr(), Mm, N, K, ldw, int(os.environ.get("ACT","1")), stream_ptr(dev)))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 20
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    by = 2.0 * Mm * (K + N)
    ref = torch.nn.functional.silu(A[:256].float() @ W[:N, :K].float().t())
    err = float((out[:256].float() - ref).abs().max())
    print(f"M={Mm:8d} N={N:5d} K={K:5d}  {ms*1e3:8.1f} us  {by/ms/1e6:7.0f} GB/s  {2.0*Mm*N*K/ms/1e9:7.1f} TFLOP/s  maxerr {err:.3f}")
