"""Tiny workload for PMC runs: a few forwards of one model, and (argv[3]) a JSON side file with the labels of the forward's
op launches in launch order, so that tools/pmc_aggregate.py can attribute counters per launch (developer tool)."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import json, sys, torch
import imageretrievalresearch_amd as M
from imageretrievalresearch_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else "efficientnet_b3a"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
model = M.create_model(name, num_classes=0).to("cuda:0").eval()
x = M.synth_fill(B * 3 * 224 * 224, 1, synth.UNIFORM, "cuda:0").view(B, 3, 224, 224)
for _ in range(3):
    model(x)
torch.cuda.synchronize()
if len(sys.argv) > 3:
    model.set_option("profile", 1)
    model(x)
    model.profile_read()
    groups = []
    for lab, kind, ms, by in model.profile_ops(B):
        if ms > 0:
            groups.append({"label": lab, "kind": kind, "ms": ms, "bytes": by, "ops": 1})
        elif groups:
            groups[-1]["bytes"] += by
            groups[-1]["ops"] += 1
    json.dump({"model": name, "batch": B, "forwards": 4, "launches": groups}, open(sys.argv[3], "w"))
