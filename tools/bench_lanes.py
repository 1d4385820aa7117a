"""Developer tool: forward time vs the "lanes" option (concurrent chunks on internal streams)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imageretrievalresearch_amd as M
from imageretrievalresearch_amd import synth
dev = "cuda:0"
name = sys.argv[1] if len(sys.argv) > 1 else "efficientnet_b3a"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
model = M.create_model(name, num_classes=0).to(dev).eval()
x = M.synth_fill(B * 3 * 224 * 224, 1, synth.UNIFORM, dev).view(B, 3, 224, 224)
ref = None
for lanes in (1, 2, 3, 4, 1):
    model.set_option("lanes", lanes)
    for _ in range(3): out = model(x)
    torch.cuda.synchronize(); t = time.perf_counter(); n = 30
    for _ in range(n): out = model(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    if ref is None: ref = out.clone()
    print(f"{name} B={B} lanes={lanes}: {dt*1e3:.3f} ms/forward {B/dt:.0f} img/s  bit-identical to lanes=1: {bool(torch.equal(out, ref))}")
