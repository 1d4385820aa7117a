import os as _os, sys as _sys
_sys.path.insert(0, "/root/repo")
import json, time, torch
import imageretrievalresearch_amd as M
from imageretrievalresearch_amd import synth
dev="cuda:0"; B=128
model = M.create_model("swin_base_patch4_window7_224", num_classes=0).to(dev).eval()
x = M.synth_fill(B*3*224*224, 1, synth.UNIFORM, dev).view(B,3,224,224)
for _ in range(5): y=model(x)
torch.cuda.synchronize(); t0=time.perf_counter(); n=30
for _ in range(n): model(x)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/n
print("DEEP", _os.environ.get("MI355_GEMM_DEEP"), "ms", dt*1e3, "img/s", B/dt, "sum", float(y.float().sum()), float(y.float().abs().max()))
