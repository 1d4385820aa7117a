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
# host-side enqueue cost of one forward (a sync between calls, so the queue is empty and the call only enqueues): median of 21
# calls at a batch size the plan has already been carved for (a single first call also pays the re-plan: 0.4 - 0.9 ms, box-dependent)
import statistics
for B in (1, 16):
    x = M.synth_fill(B * 3 * 224 * 224, 1, synth.UNIFORM, dev).view(B, 3, 224, 224)
    model(x); torch.cuda.synchronize()
    ts = []
    for _ in range(21):
        t = time.perf_counter()
        model(x)
        ts.append(time.perf_counter() - t)
        torch.cuda.synchronize()
    print(f"B={B}: host time of one forward call {statistics.median(ts)*1e3:.3f} ms median, {min(ts)*1e3:.3f} ms min (enqueue only)")
