"""Embed-only throughput of the BASELINE configs 2-4 (wall clock, device-resident synthetic input)."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import json, time, torch
import imageretrievalresearch_amd as M
from imageretrievalresearch_amd import synth

dev = "cuda:0"
out = {}
for name, B in (("efficientnet_b3a", 256), ("rexnet_150", 256), ("rexnet_200", 256), ("swin_base_patch4_window7_224", 128)):
    model = M.create_model(name, num_classes=0).to(dev).eval()
    x = M.synth_fill(B * 3 * 224 * 224, 1, synth.UNIFORM, dev).view(B, 3, 224, 224)
    for _ in range(5):
        model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        model(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    tr = model.traffic(B)
    out[name] = {"batch": B, "ms_per_batch": dt * 1e3, "images_per_s": B / dt,
                 "algorithmic_GBps": (tr["act_bytes"] + tr["weight_bytes"]) / dt / 1e9,
                 "tflops": 2 * tr["macs"] / dt / 1e12, "act_MB_per_img": tr["act_bytes"] / B / 1e6,
                 "gflop_per_img": 2 * tr["macs"] / B / 1e9}
    print(name, json.dumps(out[name]))
    del model
print(json.dumps(out))
