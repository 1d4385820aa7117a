"""Per-op timing table of one forward (developer tool): python tools_profile_ops.py [model] [batch]"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import sys
import torch
import imageretrievalresearch_amd as M
from imageretrievalresearch_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "efficientnet_b3a"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = "cuda:0"
model = M.create_model(name, num_classes=0).to(dev).eval()
import os
for kv in os.environ.get("MI355_OPTS", "").split(","):
    if "=" in kv:
        model.set_option(kv.split("=")[0], int(kv.split("=")[1]))
x = M.synth_fill(B * 3 * 224 * 224, 1, synth.UNIFORM, dev).view(B, 3, 224, 224)
for _ in range(3):
    model(x)
torch.cuda.synchronize()
model.set_option("profile", 1)
for _ in range(5):
    model(x)
fam = model.profile_read()
tot = 0.0
print(f"{'op':38s} {'kind':5s} {'ms':>8s} {'MB':>9s} {'GB/s':>8s}")
for lab, kind, ms, by in model.profile_ops(B):
    tot += ms
    print(f"{lab:38s} {kind:5s} {ms:8.4f} {by / 1e6:9.1f} {by / max(ms, 1e-9) / 1e6:8.0f}")
print("sum of op ms per forward", tot, {k: round(v['ms'] / 5, 3) for k, v in fam.items() if v['launches']})
