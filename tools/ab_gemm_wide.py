"""Developer tool: the persistent wide-tile GEMM (gemm_wide.hip) against k_gemm_big on Swin's linear shapes, in one process:
bits (must be identical), time, TFLOP/s.  MI355_GEMM_WIDE is read per call, MI355_GEMM_WIDE_MI (5..8, forces the tile height) at
first use.   python tools/ab_gemm_wide.py [MxNxK,...]"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import os, sys, torch
import imageretrievalresearch_amd as M
from imageretrievalresearch_amd._lib import lib, check, stream_ptr
dev = "cuda:0"
shapes = [(25088, 1536, 512), (25088, 2048, 512), (25088, 512, 2048), (25088, 512, 512), (6272, 3072, 1024), (6272, 4096, 1024),
          (6272, 1024, 4096), (6272, 1024, 1024), (100352, 768, 256), (100352, 1024, 256), (100352, 256, 1024), (100352, 256, 256),
          (401408, 512, 128), (401408, 384, 128), (6272, 1024, 2048), (25088, 512, 1024), (5000, 520, 192)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in t.split("x")) for t in sys.argv[1].split(",")]
act = int(os.environ.get("ACT", "4"))
VAR = os.environ.get("AB_VAR", "MI355_GEMM_WIDE")      # the switch to A/B: MI355_GEMM_WIDE or MI355_GEMM_BN256
BASE = {"MI355_GEMM_WIDE": "0", "MI355_GEMM_BN256": "0"}
for k_, v_ in BASE.items(): os.environ[k_] = v_
for (Mm, N, K) in shapes:
    ldw = (K + 31) // 32 * 32
    Np = (N + 15) // 16 * 16
    g = torch.Generator(device=dev).manual_seed(Mm + N + K)
    A = (torch.randn(Mm, K, device=dev, generator=g) * 0.5).bfloat16()
    W = torch.zeros(Np, ldw, device=dev, dtype=torch.bfloat16)
    W[:N, :K] = (torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16()
    bias = torch.randn(Np, device=dev, generator=g) * 0.1
    outs, times = {}, {}
    for mode in ("0", "1"):
        os.environ[VAR] = mode
        out = torch.full((Mm, N), 7.0, device=dev, dtype=torch.bfloat16)
        def run():
            check(lib().mi355_gemm_bf16(A.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr(), Mm, N, K, ldw, act, stream_ptr(dev)))
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n): run()
        e1.record(); torch.cuda.synchronize()
        times[mode] = e0.elapsed_time(e1) / n
        outs[mode] = out
    same = torch.equal(outs["0"].view(torch.int16), outs["1"].view(torch.int16))
    nbad = int((outs["0"].view(torch.int16) != outs["1"].view(torch.int16)).sum())
    tf = lambda ms: 2.0 * Mm * N * K / ms / 1e9
    print(f"M={Mm:7d} N={N:5d} K={K:5d}  off {times['0']*1e3:7.1f} us {tf(times['0']):6.0f} TF | on {times['1']*1e3:7.1f} us {tf(times['1']):6.0f} TF"
          f" | x{times['0']/times['1']:.2f}  bits {'identical' if same else f'DIFFER in {nbad}'}", flush=True)
