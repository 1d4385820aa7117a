#!/usr/bin/env python3
"""Headline benchmark: EfficientNet-B3a embed + cosine top-k over a 100k-row gallery (BASELINE.json).

One step = one pass of the hot path over one batch: embed 256 synthetic 224x224 images per GPU
(bf16 activations, fp32 accumulate) and rank the resulting 256 embeddings per GPU (k=3, fp32) against a
100 000 x 1536 gallery resident in HBM.  With N > 1 ranks (one process per GPU, torchrun) the gallery is
row-sharded 100000/N rows per GPU and the rank step is all-gather(queries) -> local top-k ->
all-gather(candidates) -> merge; images per GPU are fixed, so scaling is weak.

Prints ONE JSON line on rank 0 (contract in the task brief) carrying `roofline` (the single kernel launch that takes
the most time per forward, timed live with hipEvents on the launch stream; the per-family table sits beside it) and
`cpu_baseline` (the CPU oracle = a port of the reference's CPU path, timed on a bounded sample on this box's host cores).
With N > 1 rank 0 also re-computes the first 64 queries against the WHOLE gallery on its own GPU and requires the sharded
result to be bit-identical, so the first real RCCL run verifies itself.
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import imageretrievalresearch_amd as M  # noqa: E402
from imageretrievalresearch_amd import synth  # noqa: E402

PMC_EMBED = "r03_pmc_traffic_effnet_b256.json"     # committed PMC summaries (tools/refresh_profiles.sh), used for `traffic`
PMC_RANK = "r03_pmc_rank_kernels.json"
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
D = 1536
GALLERY_ROWS = 100_000
BATCH = 256
TOPK = 3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--gallery", type=int, default=GALLERY_ROWS)
    ap.add_argument("--microbatch", type=int, default=int(os.environ.get("MI355_MICROBATCH", "0")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=1, choices=(1, 2),
                    help="1 (default): embed then rank on one HIP stream; 2: rank of batch i on a second stream beside the embed "
                         "of batch i+1 (measured: hides <= 1 %% - the embed kernels fill the CUs - and adds launch-time variance)")
    ap.add_argument("--model", default="efficientnet_b3a")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    return ap.parse_args()


def csrc_sha16():
    """Fingerprint of the kernel sources: the PMC summary under profiles/ carries the same value when it was collected
    from this code (tools/refresh_profiles.sh), so a stale `traffic` is detectable."""
    h = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(ROOT, "imageretrievalresearch_amd", "csrc", "*.h*"))):
        with open(fn, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def cpu_baseline(model_name, gallery_rows):
    """Reference-style CPU path on a bounded sample (about 10-30 s): the oracle's fp32 forward on 16
    images (BASELINE configs[0] batch) and the per-query CosineSimilarity+topk loop of train/train.py:250-251
    for 8 queries against the same-size gallery."""
    from oracle import effnet, rank as orank
    # the GPU box gives one GPU job a 16-core share (os.cpu_count() reports the whole 256-thread host and
    # oversubscribing it made the oracle 20x slower); use what we are actually allowed to run on
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    sd = effnet.init_state_dict(2, num_classes=0)
    x = torch.from_numpy(synth.uniform(1, (16, 3, 224, 224)))
    with torch.no_grad():
        effnet.forward(sd, x[:2])                      # warm-up
        t0 = time.perf_counter()
        emb = effnet.forward(sd, x)
        t_embed = (time.perf_counter() - t0) / 16
    rows = min(gallery_rows, 100_000)
    g = torch.from_numpy(synth.normal(5, (rows, D)))
    q = torch.from_numpy(synth.normal(13, (8, D)))
    orank.rank_reference_loop(q[:1], g, TOPK)
    t0 = time.perf_counter()
    orank.rank_reference_loop(q, g, TOPK)
    t_rank = (time.perf_counter() - t0) / 8 * (gallery_rows / rows)
    del emb
    return {"value": 1.0 / (t_embed + t_rank), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32 forward on 16 images ({t_embed * 1e3:.1f} ms/img) + reference-style "
                      f"cos+topk loop for 8 queries vs {rows} rows ({t_rank * 1e3:.1f} ms/query); torch "
                      f"{torch.__version__} CPU, {cores} threads",
            "embed_images_per_s": 1.0 / t_embed, "rank_queries_per_s": 1.0 / t_rank}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        a.gpus = world
    dist = torch.distributed
    if a.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)

    # ---- resident inputs (timed region starts with everything in HBM)
    model = M.create_model(a.model, num_classes=0, seed=0).to(dev).eval()
    if a.microbatch:
        model.set_option("microbatch", a.microbatch)
    x = M.synth_fill(a.batch * 3 * 224 * 224, 1000 + rank, synth.UNIFORM, dev).view(a.batch, 3, 224, 224)
    lo = a.gallery * rank // world
    hi = a.gallery * (rank + 1) // world
    shard = M.synth_fill((hi - lo) * D, 5, synth.NORMAL, dev, offset=lo * D).view(hi - lo, D)
    gal = M.ShardedGallery(shard)
    del shard

    # One HIP stream by default: a step is embed THEN rank (split-bf16 MFMA loop).  With --streams 2 the rank of batch i is
    # enqueued on a second stream beside the embed of batch i+1; round 2 measured that this hides <= 1 % (the embed kernels
    # occupy the whole register file / LDS of every CU, so rank workgroups only run in the gaps) and that it makes kernel
    # durations erratic (k_dw3_lds 124-472 us), so it is no longer the default.  Every step's embed AND rank complete inside
    # the timed region either way.
    s_embed, s_rank = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def step():
        if a.streams == 1:
            with torch.cuda.stream(s_embed):
                return gal.search(model(x), TOPK)
        with torch.cuda.stream(s_embed):
            emb = model(x)
            done = torch.cuda.Event()
            done.record(s_embed)
        with torch.cuda.stream(s_rank):
            s_rank.wait_event(done)
            emb.record_stream(s_rank)
            return gal.search(emb, TOPK)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.synchronize()   # inputs / gallery were built on the default stream
    step()                     # priming call (one-time kernel attributes, RCCL communicator setup): untimed even with --warmup 0
    for _ in range(a.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / a.steps * 1e3
    images_per_s = a.batch * world * a.steps / dt
    assert out[1].shape == (a.batch * world, TOPK)
    if world > 1 and rank == 0:
        # self-check of the sharded path: the first 64 queries (rank 0's own embeddings) against the WHOLE gallery on this GPU
        with torch.cuda.stream(s_embed):
            q64 = model(x)[:64].clone()
        s_embed.synchronize()
        # the same representation the shards hold: rows normalised once (ShardedGallery / Gallery.add), scores = qn . gn
        full = M.l2_normalize_rows(M.synth_fill(a.gallery * D, 5, synth.NORMAL, dev).view(a.gallery, D))
        wv, wi = M.cosine_topk(q64, full, TOPK, gallery_is_normalized=True)
        del full
        torch.cuda.synchronize()
        if not (torch.equal(out[1][:64], wi) and torch.equal(out[0][:64], wv)):
            raise SystemExit("bench: sharded top-k differs from the unsharded result on rank 0's first 64 queries")

    result = None
    if rank == 0:
        # ---- side measurements on rank 0 (not part of `value`)
        def timeit(fn, n):
            fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n

        n_side = max(3, min(a.steps, 20))
        t_embed = timeit(lambda: model(x), n_side)
        q = M.synth_fill(a.batch * D, 13, synth.NORMAL, dev).view(a.batch, D)
        t_rank = timeit(lambda: gal.ops.local_topk(q, gal.local, TOPK, 0), n_side)
        t_rank1 = timeit(lambda: gal.ops.local_topk(q[:1], gal.local, TOPK, 0), n_side)   # the reference's per-query call shape
        # side metric: the same search against the shard held as the GEMM's bf16 planes (mi355_gallery_prepare: +6 B per element
        # of HBM, no per-call split of the gallery; results identical bit for bit).  `value` uses the fp32 rows.
        prep = M.PreparedGallery(gal.local)
        t_rank_prep = timeit(lambda: prep.search(q, TOPK), n_side)
        del prep

        # ---- roofline of the dominant KERNEL: hipEvents around every launch, on the launch stream.  A launch that runs
        # several ops of the plan (expand + depthwise, or a whole MBConv block) carries the layer-granular bytes of all of them.
        model.set_option("profile", 1)
        nprof = 3
        for _ in range(nprof):
            model(x)
        prof = model.profile_read()
        ops = model.profile_ops(a.batch)
        model.set_option("profile", 0)
        tr = model.traffic(a.batch)
        groups = []                                    # (label, kind, avg ms per launch, algorithmic bytes per launch, ops, key)
        for lab, kind, ms, by in ops:
            if ms > 0:
                groups.append([lab, kind, ms, by, 1, lab])
            elif groups:
                groups[-1][3] += by                    # an op executed inside the previous launch
                groups[-1][4] += 1
                groups[-1][5] += " + " + lab
        # the dominant KERNEL = the launches of identical layer shape (same ops, same shapes: one kernel instantiation) with
        # the largest total time per forward, as `rocprofv3 --stats` ranks kernel symbols; its roofline uses the mean launch
        by_key = {}
        for g_ in groups:
            e = by_key.setdefault(g_[5], {"first": g_, "ms": 0.0, "n": 0})
            e["ms"] += g_[2]
            e["n"] += 1
        dom = max(by_key.values(), key=lambda e: e["ms"])
        top = list(dom["first"])
        top[2] = dom["ms"] / dom["n"]
        kernel_of = {1: {"gemm": "k_gemm_bf16 / k_proj_lds / k_gemm_big / k_gemm_stream (1x1 conv)", "dw": "k_dwconv / k_dw3_lds", "stem": "k_stem",
                         "se": "k_se", "attn": "k_win_attn", "ln": "k_layernorm", "other": "other"},
                     2: "k_fused_late (1x1 expand + depthwise + SE squeeze on a whole-image tile, expanded tensor in LDS)",
                     4: "k_mbconv_block (whole MBConv block: expand, MFMA depthwise, SE, gated projection, residual; the expanded tensor stays in LDS, the depthwise output makes one L2 round trip)"}
        if top[4] == 2 and any(t in top[0] for t in ("@112x112", "@56x56", "@28x28")):
            kname = ("k_sweep_mbconv (row sweep: 1x1 expand on MFMA + depthwise on MFMA from an LDS row window + SE squeeze; the "
                     "expanded tensor never reaches HBM, so the layer-granular bytes below exceed what the launch moves)")
        else:
            kname = kernel_of[top[4]] if top[4] > 1 else kernel_of[1].get(top[1], top[1])
        achieved = top[3] / (top[2] * 1e-3) / 1e9
        # HBM bytes of that launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs, gfx950
        # x2 read correction) - cannot be collected from inside this process; null when the summary is from other code
        traffic, traffic_source = None, None
        pmc_path = os.path.join(ROOT, "profiles", PMC_EMBED)
        if a.model == "efficientnet_b3a" and a.batch == 256 and os.path.exists(pmc_path):
            with open(pmc_path) as f:
                pmc = json.load(f)
            if pmc.get("csrc_sha16") == csrc_sha16():
                ent = pmc.get("per_op", {}).get(top[0])
                if ent:
                    traffic = ent["hbm_bytes_per_launch"]
                    traffic_source = f"profiles/{PMC_EMBED} (rocprofv3 --pmc, separate passes, same kernel sources)"
        # the rank GEMM's HBM-side read bytes per launch, same provenance rule (tools/refresh_profiles.sh: bench_rank.py with
        # CASES=256x100000 under --pmc FETCH_SIZE).  RAW counter x 1024: the gallery arrives as 64-byte-per-row LDS-DMA pieces,
        # for which the gfx950 "x2 for wide coalesced reads" correction does not apply (x2 would exceed the HBM peak)
        rank_traffic, rank_traffic_source = None, None
        rk_path = os.path.join(ROOT, "profiles", PMC_RANK)
        if a.batch == 256 and (hi - lo) == 100_000 and os.path.exists(rk_path):
            with open(rk_path) as f:
                rk = json.load(f)
            if rk.get("csrc_sha16") == csrc_sha16():
                ent = rk.get("kernels", {}).get("void mi355::k_cos_gemm_split<2, 4>")
                if ent and "FETCH_SIZE" in ent:
                    rank_traffic = ent["FETCH_SIZE"] * 1024.0
                    rank_traffic_source = f"profiles/{PMC_RANK} (rocprofv3 --pmc FETCH_SIZE, raw counter x 1024 per launch of the main tile launch)"
        roofline = {"bound": "hbm", "kernel": f"{kname} @ {top[0]}", "achieved": achieved, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                    "avg_launch_ms": top[2], "launches_per_forward": dom["n"], "ms_per_forward": dom["ms"],
                    "ops_in_launch": top[5], "algorithmic_bytes_per_launch": top[3],
                    "bytes_model": "SURVEY 8d layer-granular bytes of every op the launch executes (fused intermediates included)",
                    "note": f"fused kernels are VALU / latency-bound, not HBM-bound (PMC per launch: profiles/{PMC_EMBED}; "
                            "whole-forward fraction: embed_roofline)",
                    "family_ms_per_forward": {k: v["ms"] / nprof for k, v in prof.items() if v["launches"]},
                    "family_GBps": {k: tr["bytes_by_kind"][k] / (v["ms"] / nprof * 1e-3) / 1e9
                                    for k, v in prof.items() if v["launches"] and tr["bytes_by_kind"].get(k)}}
        # ---- the reference's DEFAULT inference front end (conv_input=True, inference/inference.py:77,101-105) from uint8 images:
        # SquarePad + ToTensor + Normalize + conv_input + SiLU fused into the stem (side metric, not part of `value`)
        side = {}
        if a.model != "swin_base_patch4_window7_224":
            u8 = (M.synth_fill(a.batch * 224 * 224 * 3, 77, synth.UNIFORM, dev) * 255.0).to(torch.uint8).view(a.batch, 224, 224, 3)
            conv_in = M.models.ConvInput().to(dev) if hasattr(M.models, "ConvInput") else None
            if conv_in is not None:
                t_u8 = timeit(lambda: model.forward_uint8(u8, conv_input=conv_in), n_side)
                side["embed_uint8_conv_input_images_per_s_1gpu"] = a.batch / t_u8
            t_u8p = timeit(lambda: model.forward_uint8(u8), n_side)
            side["embed_uint8_images_per_s_1gpu"] = a.batch / t_u8p
        embed_gbs = (tr["act_bytes"] + tr["weight_bytes"]) / t_embed / 1e9
        result = {
            "metric": "images/sec embed + queries/sec top-k@100k-gallery, EffNet-B3a 224^2",
            "value": images_per_s, "unit": "images/s (each image embedded and ranked as a query)",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": f"{a.model} bf16 embed bs={a.batch}/GPU 224x224 + cosine top-{TOPK} of the "
                                   f"{a.batch * world} embeddings vs {a.gallery}x{D} fp32 gallery "
                                   f"(row-sharded over {world} GPU)", "batch_per_gpu": a.batch,
                       "gallery_rows": a.gallery, "dim": D, "k": TOPK, "microbatch": a.microbatch,
                       "parallelism": f"dp{world} + row-sharded gallery", "streams": "embed || rank (2 HIP streams)" if a.streams == 2 else "one HIP stream"},
            "embed_images_per_s_1gpu": a.batch / t_embed,
            "rank_queries_per_s_1gpu_shard": a.batch / t_rank,
            "rank_queries_per_s_1gpu_shard_prepared_gallery": a.batch / t_rank_prep,
            "embed_roofline": {"algorithmic_GBps": embed_gbs, "frac_of_8TBps": embed_gbs / HBM_PEAK_GBS,
                               "act_MB_per_img": tr["act_bytes"] / a.batch / 1e6,
                               "gflop_per_img": 2 * tr["macs"] / a.batch / 1e9},
            "rank_single_query": {"queries_per_s": 1.0 / t_rank1, "bound": "hbm",
                                  "gallery_stream_GBps": 4.0 * (hi - lo) * D / t_rank1 / 1e9, "peak_GBps": HBM_PEAK_GBS},
            # 2*Q*G*D useful flops; the default loop executes six bf16 products per fp32 product (three-way split, fp32
            # accumulation - rank.hip), so the matrix pipe does 6x that against the dense bf16 peak; the exact fp32 MFMA
            # loop (MI355_RANK_EXACT_F32=1) would be bounded by the 157.3 TF fp32 matrix peak
            "rank_roofline": (lambda tf, exact: {
                "bound": "mfma-f32" if exact else "mfma-bf16 (fp32 operands as 3 bf16 planes, 6 products, fp32 accumulate)",
                "tflops": tf, "executed_tflops": tf if exact else 6.0 * tf,
                "peak_tflops": 157.3 if exact else 2500.0, "frac": tf / 157.3 if exact else 6.0 * tf / 2500.0,
                "vs_fp32_mfma_peak": tf / 157.3,
                "algorithmic_bytes": 4.0 * (hi - lo) * D + 6.0 * a.batch * D, "traffic": rank_traffic,
                "traffic_source": rank_traffic_source})(2.0 * a.batch * (hi - lo) * D / t_rank / 1e12,
                                                  os.environ.get("MI355_RANK_EXACT_F32", "0") not in ("", "0")),
            "roofline": roofline,
        }
        result.update(side)
        if not a.no_cpu_baseline and world == 1:   # the host-core baseline is reported at N=1 only
            result["cpu_baseline"] = cpu_baseline(a.model, a.gallery)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
