"""Developer tool: where the whole-block kernel starts to pay (option fuse_block_min_batch) and what two concurrent lanes buy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imageretrievalresearch_amd as M
from imageretrievalresearch_amd import synth
dev = "cuda:0"
model = M.create_model("efficientnet_b3a", num_classes=0).to(dev).eval()

def t(x, n=20):
    for _ in range(4): model(x)
    torch.cuda.synchronize(); s = time.perf_counter()
    for _ in range(n): model(x)
    torch.cuda.synchronize(); return (time.perf_counter() - s) / n * 1e3

for B in (32, 64, 96, 128, 160, 192, 256, 384, 512):
    x = M.synth_fill(B * 3 * 224 * 224, 1, synth.UNIFORM, dev).view(B, 3, 224, 224)
    model.set_option("fuse_block_min_batch", 1); a = t(x)
    model.set_option("fuse_block_min_batch", 100000); b = t(x)
    model.set_option("fuse_block_min_batch", 96)
    print(f"B={B}: block kernel {a:.3f} ms ({B/a*1e3:.0f} img/s)   unfused chain {b:.3f} ms ({B/b*1e3:.0f} img/s)")
x = M.synth_fill(256 * 3 * 224 * 224, 1, synth.UNIFORM, dev).view(256, 3, 224, 224)
for lanes in (1, 2, 4):
    model.set_option("lanes", lanes)
    print(f"B=256 lanes={lanes}: {t(x):.3f} ms")
model.set_option("lanes", 1)
