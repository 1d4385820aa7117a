"""Small-batch latency of one EfficientNet-B3a forward (developer tool): wall clock per forward and host enqueue time."""
import os as _os
import sys, time, torch
sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import imageretrievalresearch_amd as M
from imageretrievalresearch_amd import synth
dev = "cuda:0"
model = M.create_model("efficientnet_b3a", num_classes=0).to(dev).eval()
for B in (1, 4, 16, 64, 256):
    x = M.synth_fill(B * 3 * 224 * 224, 1, synth.UNIFORM, dev).view(B, 3, 224, 224)
    for _ in range(5): model(x)
    torch.cuda.synchronize(); t = time.perf_counter(); n = 50
    for _ in range(n): model(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print(f"B={B}: {dt*1e3:.3f} ms/forward  {B/dt:.0f} img/s")
# host-side enqueue cost of one forward (no sync between calls; the queue absorbs the launches)
for B in (1, 16):
    x = M.synth_fill(B * 3 * 224 * 224, 1, synth.UNIFORM, dev).view(B, 3, 224, 224)
    torch.cuda.synchronize(); t = time.perf_counter()
    model(x)
    t1 = time.perf_counter() - t
    torch.cuda.synchronize()
    print(f"B={B}: host time of one forward call {t1*1e3:.3f} ms (enqueue only)")
