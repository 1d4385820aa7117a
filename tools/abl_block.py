"""Developer tool: ablation of the whole-block kernel's phases (option "block_variant" bits, results are garbage):
per-op time of the late blocks with phases switched off, B = 256.  python tools/abl_block.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imageretrievalresearch_amd as M
from imageretrievalresearch_amd import synth
dev = "cuda:0"
model = M.create_model("efficientnet_b3a", num_classes=0).to(dev).eval()
Bt = 256
x = M.synth_fill(Bt * 3 * 224 * 224, 1, synth.UNIFORM, dev).view(Bt, 3, 224, 224)
names = {0: "full", 1: "-expand", 2: "-dw", 3: "-expand-dw", 4: "-se", 8: "-proj", 12: "-se-proj", 15: "-all phases", 16: "-wreq", 31: "-everything", 19: "-expand-dw-wreq"}
for v in names:
    model.set_option("block_variant", v)
    model.set_option("profile", 1)
    for _ in range(4): model(x)
    model.profile_read()
    rows = model.profile_ops(Bt)
    model.set_option("profile", 0)
    sel = {}
    for lab, kind, ms, by in rows:
        if ms > 0 and ("@14x14" in lab or "@7x7" in lab) and "->" in lab:
            key = lab.split()[1] + lab.split()[2]
            sel.setdefault(key, []).append(ms * 1e3)
    print(f"{names[v]:16s} " + "  ".join(f"{k}:{sum(t)/len(t):.0f}" for k, t in sel.items() if k in ("96->576@14x14", "136->816@14x14", "232->1392@7x7", "384->2304@7x7")))
