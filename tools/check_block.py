"""Developer tool: whole-block kernel (fuse_block=1) against the unfused op chain (fuse_block=0) on the same weights —
per-block taps, then timing and the in-kernel phase stamps at B=256.  python tools/check_block.py [B_time]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import imageretrievalresearch_amd as M
from imageretrievalresearch_amd import synth
from oracle import effnet

dev = "cuda:0"
sd = effnet.init_state_dict(2)
model = M.create_model("efficientnet_b3a", num_classes=0).to(dev).eval()
model.load_state_dict(sd, strict=False)
names = ["stem"] + [f"blocks.{s}.{b}" for s, r in enumerate([2, 3, 3, 5, 5, 6, 2]) for b in range(r)] + ["head"]

def taps(B, opt):
    model.set_option("fuse_block", opt)
    x = M.synth_fill(B * 3 * 224 * 224, 1, synth.UNIFORM, dev).view(B, 3, 224, 224)
    model.enable_taps(True)
    out = model(x)
    t = {n: model.read_tap(n).float().cpu() for n in names}
    model.enable_taps(False)
    return t, out.float().cpu()

for B in (3,):
    t0, o0 = taps(B, 0)
    t1, o1 = taps(B, 1)
    worst = 0.0
    for n in names:
        a, b = t0[n], t1[n]
        rel = float((a - b).norm() / (a.norm() + 1e-12))
        mx = float((a - b).abs().max())
        frac = float(((a - b).abs() > 0).float().mean())
        worst = max(worst, rel)
        print(f"B={B} tap {n:12s} relL2 {rel:.3e} maxabs {mx:.3e} frac_diff {frac:.4f} finite {bool(torch.isfinite(b).all())}")
    print("embedding relL2", float((o0 - o1).norm() / o0.norm()), "worst tap", worst)

Bt = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = M.synth_fill(Bt * 3 * 224 * 224, 1, synth.UNIFORM, dev).view(Bt, 3, 224, 224)
for opt in (0, 1):
    model.set_option("fuse_block", opt)
    for _ in range(3): model(x)
    torch.cuda.synchronize(); t = time.perf_counter(); n = 20
    for _ in range(n): model(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print(f"fuse_block={opt}: B={Bt} {dt*1e3:.3f} ms/forward {Bt/dt:.0f} img/s")
model.set_option("fuse_block", 1)
model.set_option("profile", 1)
for _ in range(5): model(x)
fam = model.profile_read()
tot = 0.0
rows = model.profile_ops(Bt)
for i, (lab, kind, ms, by) in enumerate(rows):
    tot += ms
    if ms > 0: print(f"{i:3d} {lab:38s} {kind:5s} {ms:8.4f} ms {by/1e6:9.1f} MB {by/max(ms,1e-9)/1e6:8.0f} GB/s")
print("sum of op ms per forward", tot, {k: round(v['ms'] / 5, 3) for k, v in fam.items() if v['launches']})
model.set_option("profile", 0)
model.set_option("block_stamps", 1)
model(x); torch.cuda.synchronize()
print("op  | xload expand dw tailwait fc1 fc2 proj epi | total  (kcycles of wave 0, mean over images)")
for i, v in model.block_stamps():
    print(f"{i:3d} {rows[i][0]:24s} " + " ".join(f"{c/1e3:6.1f}" for c in v[:8]) + f" | {sum(v)/1e3:7.1f}")
model.set_option("block_stamps", 0)
