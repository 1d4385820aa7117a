"""One EfficientNet-B3a forward (B=256) + one rank call with roctx ranges on (developer tool):
    MI355_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --output-format csv -d out -- python3 tools/roctx_demo.py
The marker trace then holds one range per executor op ("embed/block: pw 96->576 @14x14", ...) and per rank phase."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import torch
import imageretrievalresearch_amd as M
from imageretrievalresearch_amd import synth
dev = "cuda:0"
model = M.create_model("efficientnet_b3a", num_classes=0).to(dev).eval()
x = M.synth_fill(256 * 3 * 224 * 224, 1, synth.UNIFORM, dev).view(256, 3, 224, 224)
g = M.l2_normalize_rows(M.synth_fill(100000 * 1536, 5, synth.NORMAL, dev).view(100000, 1536))
model.set_option("roctx", 0)
for _ in range(2):
    emb = model(x)
    M.cosine_topk(emb, g, 3, gallery_is_normalized=True)
torch.cuda.synchronize()
model.set_option("roctx", 1)
emb = model(x)
v, i = M.cosine_topk(emb, g, 3, gallery_is_normalized=True)
torch.cuda.synchronize()
model.set_option("roctx", 0)
print("ok", tuple(v.shape))
