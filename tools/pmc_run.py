"""Tiny workload for PMC runs: a few forwards of one model (developer tool)."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import sys, torch
import imageretrievalresearch_amd as M
from imageretrievalresearch_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else "efficientnet_b3a"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
model = M.create_model(name, num_classes=0).to("cuda:0").eval()
x = M.synth_fill(B * 3 * 224 * 224, 1, synth.UNIFORM, "cuda:0").view(B, 3, 224, 224)
for _ in range(3):
    model(x)
torch.cuda.synchronize()
