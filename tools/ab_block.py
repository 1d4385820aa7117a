"""Developer tool: A/B of the whole-block kernel's tuning variants (option "block_variant") in ONE process.
  parity : every variant against the unfused op chain at B = 3 (fuse_block_min_batch = 1), per-block taps
  timing : interleaved rounds at B = 256, median and min per variant
  detail : per-op table and in-kernel phase stamps of the default variant
python tools/ab_block.py [variants, comma separated] [B]"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imageretrievalresearch_amd as M
from imageretrievalresearch_amd import synth
from oracle import effnet

variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
Bt = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = "cuda:0"
sd = effnet.init_state_dict(2)
model = M.create_model("efficientnet_b3a", num_classes=0).to(dev).eval()
model.load_state_dict(sd, strict=False)
names = ["stem"] + [f"blocks.{s}.{b}" for s, r in enumerate([2, 3, 3, 5, 5, 6, 2]) for b in range(r)] + ["head"]


def taps(B, fuse_block, variant):
    model.set_option("fuse_block", fuse_block)
    model.set_option("fuse_block_min_batch", 1)
    model.set_option("block_variant", variant)
    x = M.synth_fill(B * 3 * 224 * 224, 1, synth.UNIFORM, dev).view(B, 3, 224, 224)
    model.enable_taps(True)
    out = model(x)
    t = {n: model.read_tap(n).float().cpu() for n in names}
    model.enable_taps(False)
    model.set_option("fuse_block_min_batch", 96)
    return t, out.float().cpu()


t0, o0 = taps(3, 0, 0)
for v in variants:
    t1, o1 = taps(3, 1, v)
    worst, wn = 0.0, ""
    for n in names:
        rel = float((t0[n] - t1[n]).norm() / (t0[n].norm() + 1e-12))
        if not torch.isfinite(t1[n]).all(): rel = float("inf")
        if rel > worst: worst, wn = rel, n
    print(f"variant {v}: vs unfused chain: embedding relL2 {float((o0 - o1).norm() / o0.norm()):.3e}, worst tap {wn} {worst:.3e}")

x = M.synth_fill(Bt * 3 * 224 * 224, 1, synth.UNIFORM, dev).view(Bt, 3, 224, 224)
model.set_option("fuse_block", 1)
res = {v: [] for v in variants}
for v in variants:
    model.set_option("block_variant", v)
    for _ in range(3): model(x)
torch.cuda.synchronize()
for rnd in range(5):
    for v in variants:
        model.set_option("block_variant", v)
        model(x); torch.cuda.synchronize()
        t = time.perf_counter(); n = 10
        for _ in range(n): model(x)
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t) / n * 1e3)
for v in variants:
    print(f"variant {v}: B={Bt} median {statistics.median(res[v]):.3f} ms  min {min(res[v]):.3f} ms  ({Bt / statistics.median(res[v]) * 1e3:.0f} img/s)")

for v in variants:
    model.set_option("block_variant", v)
    model.set_option("profile", 1)
    for _ in range(5): model(x)
    fam = model.profile_read()
    rows = model.profile_ops(Bt)
    print(f"variant {v}: families", {k: round(val['ms'] / 5, 3) for k, val in fam.items() if val['launches']})
    late = [(i, lab, ms) for i, (lab, kind, ms, by) in enumerate(rows) if ms > 0 and ("@14x14" in lab or "@7x7" in lab)]
    print("   " + "  ".join(f"{lab.split()[1]}{lab.split()[2]}:{ms * 1e3:.0f}" for i, lab, ms in late))
    model.set_option("profile", 0)

model.set_option("block_variant", variants[0])
model.set_option("block_stamps", 1)
model(x); torch.cuda.synchronize()
print("op  | xload expand dw tailwait fc1 fc2 proj epi | total  (kcycles of wave 0, mean over images)")
for i, vv in model.block_stamps():
    if sum(vv) > 0 and ("@14x14" in rows[i][0] or "@7x7" in rows[i][0]):
        print(f"{i:3d} {rows[i][0]:24s} " + " ".join(f"{c / 1e3:6.1f}" for c in vv[:8]) + f" | {sum(vv) / 1e3:7.1f}")
model.set_option("block_stamps", 0)
