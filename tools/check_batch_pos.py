"""Developer tool: the same image at different batch positions must give bit-identical taps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imageretrievalresearch_amd as M
from imageretrievalresearch_amd import synth
from oracle import effnet
dev = "cuda:0"
sd = effnet.init_state_dict(2)
model = M.create_model("efficientnet_b3a", num_classes=0).to(dev).eval()
model.load_state_dict(sd, strict=False)
names = ["stem"] + [f"blocks.{s}.{b}" for s, r in enumerate([2, 3, 3, 5, 5, 6, 2]) for b in range(r)] + ["head"]
x1 = M.synth_fill(3 * 224 * 224, 7, synth.UNIFORM, dev).view(1, 3, 224, 224)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
x = x1.repeat(B, 1, 1, 1).contiguous()
for norot in (0,):
  model.set_option("block_norot", norot)
  print("== norot", norot)
  model.enable_taps(True)
  model(x)
  for n in names:
    t = model.read_tap(n)
    bad = [i for i in range(1, B) if not torch.equal(t[0], t[i])]
    if bad:
        d = (t[0] - t[bad[0]]).abs()
        print(n, "positions differing from 0:", bad, "max", float(d.max()), "count", int((d > 0).sum()), "of", d.numel(),
              "where", (d > 0).nonzero()[:6].tolist())
        break
print("done")
