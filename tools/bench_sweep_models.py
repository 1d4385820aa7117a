import sys, time, torch
sys.path.insert(0, '/root/repo')
import imageretrievalresearch_amd as M
from imageretrievalresearch_amd import synth
for name, B in (("rexnet_150", 256), ("rexnet_200", 256), ("efficientnet_b3a", 256), ("efficientnet_b3a", 64), ("efficientnet_b3a", 16), ("efficientnet_b3a", 1)):
    model = M.create_model(name, num_classes=0).to("cuda:0").eval()
    x = M.synth_fill(B * 3 * 224 * 224, 1, synth.UNIFORM, "cuda:0").view(B, 3, 224, 224)
    outs = {}
    for opt in (0, 1):
        model.set_option("fuse_sweep", opt)
        for _ in range(3): outs[opt] = model(x)
        torch.cuda.synchronize(); t = time.perf_counter(); n = 20
        for _ in range(n): model(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
        print(f"{name} B={B} fuse_sweep={opt}: {dt*1e3:.3f} ms {B/dt:.0f} img/s", flush=True)
    a, b = outs[0].float(), outs[1].float()
    print("   rel diff", float((a - b).norm() / a.norm()), "finite", bool(torch.isfinite(b).all()))
