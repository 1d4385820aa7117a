import torch, time
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
dev="cuda:0"
for mb in (64, 308, 1024, 4096):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, dtype=torch.bfloat16, device=dev).normal_()
    b = torch.empty_like(a)
    def t(fn, it=20):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it
    tc = t(lambda: b.copy_(a)); tz = t(lambda: b.zero_()); ts = t(lambda: a.sum())
    print(f"{mb:5d} MB  copy {2*mb/1024/tc*1e3/1024*1024:7.0f} GB/s   zero(write) {mb/1024/tz*1e3/1024*1024:7.0f} GB/s   sum(read) {mb/1024/ts*1e3/1024*1024:7.0f} GB/s")
