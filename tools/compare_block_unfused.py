"""Developer tool: per late block, the whole-block kernel against expand+depthwise / SE / gated projection as separate launches
(EfficientNet-B3a, B=256).  python tools/compare_block_unfused.py"""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imageretrievalresearch_amd as M
from imageretrievalresearch_amd import synth
model = M.create_model("efficientnet_b3a", num_classes=0).to("cuda:0").eval()
x = M.synth_fill(256*3*224*224, 1, synth.UNIFORM, "cuda:0").view(256,3,224,224)
res={}
for fb in (0,1):
    model.set_option("fuse_block", fb)
    for _ in range(3): model(x)
    model.set_option("profile", 1)
    for _ in range(5): model(x)
    model.profile_read(); rows = model.profile_ops(256); model.set_option("profile", 0)
    res[fb]=rows
# group per block: find indices where label starts with 'pw' and '@14x14' or '@7x7' expands
r0,r1=res[0],res[1]
i=0
while i < len(r1):
    lab=r1[i][0]
    if r1[i][2]>0 and i+3 < len(r1) and r1[i+1][2]==0 and r1[i+2][2]==0 and r1[i+3][2]==0 and ('@14x14' in lab or '@7x7' in lab):
        t1=r1[i][2]; t0=sum(r0[j][2] for j in range(i,i+4))
        print(f"{i:3d} {lab:26s} block {t1*1e3:7.1f} us   unfused {t0*1e3:7.1f} us ({' + '.join(f'{r0[j][2]*1e3:.0f}' for j in range(i,i+4))})")
        i+=4
    else: i+=1
