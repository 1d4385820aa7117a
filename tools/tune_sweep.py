"""Developer tool: time the row-sweep kernel's variants (band height / occupancy class, workgroups per image, phase skips) on
the EfficientNet-B3a early-stage shapes at B=256 through the executor's per-op profile.
    python tools/tune_sweep.py [variants] [csplits] [skips]     e.g.  0,1,2,3  0  0,1,2,4"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imageretrievalresearch_amd as M
from imageretrievalresearch_amd import synth
from oracle import effnet

dev = "cuda:0"
B = 256
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2,3,4").split(",")]
csplits = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0").split(",")]
skips = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0").split(",")]
model = M.create_model("efficientnet_b3a", num_classes=0).to(dev).eval()
model.load_state_dict(effnet.init_state_dict(2), strict=False)
x = M.synth_fill(B * 3 * 224 * 224, 1, synth.UNIFORM, dev).view(B, 3, 224, 224)
model.set_option("fuse_sweep", 1)
SWEEP_OPS = (7, 11, 15, 19, 23, 27, 31)
for v in variants:
    for c in csplits:
        for sk in skips:
            model.set_option("sweep_variant", v); model.set_option("sweep_csplit", c); model.set_option("sweep_skip", sk)
            model.set_option("profile", 0)
            for _ in range(2): model(x)
            model.set_option("profile", 1)
            for _ in range(5): model(x)
            model.profile_read()
            rows = model.profile_ops(B)
            model.set_option("profile", 0)
            print(f"variant {v} csplit {c} skip {sk}: " + "  ".join(f"{rows[i][0].split('@')[-1]}:{rows[i][2]*1e3:6.1f}" for i in SWEEP_OPS)
                  + f"  | sum {sum(rows[i][2] for i in SWEEP_OPS)*1e3:7.1f} us", flush=True)

if len(sys.argv) > 4:       # phase stamps of one variant: python tools/tune_sweep.py 0 0 0 stamps
    model.set_option("sweep_variant", variants[0]); model.set_option("sweep_csplit", csplits[0]); model.set_option("sweep_skip", 0)
    model.set_option("block_stamps", 1)
    model(x); torch.cuda.synchronize()
    rows = model.profile_ops(B)
    print("kcycles summed over an image's workgroups: init | slabconst | halo | xissue | xwait | expand | bar | dw | bar | squeeze")
    for i, v in model.block_stamps():
        if i in SWEEP_OPS:
            print(f"{i:3d} {rows[i][0]:24s} " + " ".join(f"{c/1e3:8.1f}" for c in v[:10]) + f" | {sum(v)/1e3:8.1f}")
    model.set_option("block_stamps", 0)
