"""Fold two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of tools/pmc_run.py into per-kernel-family HBM bytes.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/pmc_run.py efficientnet_b3a 256
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 tools/pmc_run.py efficientnet_b3a 256
    python tools/pmc_aggregate.py gpurun_out/pmc_fetch gpurun_out/pmc_write 3 > profiles/rNN_pmc_traffic_effnet_b256.json

The third argument is the number of forwards the workload ran (pmc_run.py: 3).  Correction (MI355X_MICROARCH.md, HBM
section): both counters are in KiB and gfx950's FETCH_SIZE reports half of a wide coalesced stream, so
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import csv
import glob
import json
import os
import sys


def family(name: str) -> str:
    if "k_gemm" in name:
        return "gemm"
    if "k_fused" in name:
        return "fused"
    if "k_dwconv" in name or "k_dw_tiled" in name:
        return "dw"
    if "k_se" in name:
        return "se"
    if "k_stem" in name:
        return "stem"
    if "k_win_attn" in name:
        return "attn"
    if "k_layernorm" in name or "k_ln_token_mean" in name or "k_patch_embed" in name:
        return "ln"
    return "other"


def read_pass(directory: str, counter: str):
    tot, launches = {}, {}
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {directory}")
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                fam = family(row["Kernel_Name"])
                tot[fam] = tot.get(fam, 0.0) + float(row["Counter_Value"])
                launches[fam] = launches.get(fam, 0) + 1
    return tot, launches


def main():
    fetch_dir, write_dir, forwards = sys.argv[1], sys.argv[2], float(sys.argv[3])
    fetch, launches = read_pass(fetch_dir, "FETCH_SIZE")
    write, _ = read_pass(write_dir, "WRITE_SIZE")
    fams = {}
    for fam in sorted(set(fetch) | set(write)):
        f_kb, w_kb = fetch.get(fam, 0.0) / forwards, write.get(fam, 0.0) / forwards
        n = launches.get(fam, 0) / forwards
        hbm = (2.0 * f_kb + w_kb) * 1024.0
        fams[fam] = {"launches_per_forward": n, "fetch_size_kb_raw": f_kb, "write_size_kb": w_kb,
                     "hbm_bytes_per_forward": hbm, "hbm_bytes_per_launch": hbm / n if n else 0.0}
    print(json.dumps({
        "workload": " ".join(sys.argv[4:]) or "tools/pmc_run.py, rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes",
        "forwards": forwards,
        "correction": "hbm = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE reports half of a wide coalesced "
                      "stream (MI355X_MICROARCH.md HBM section)",
        "families": fams}, indent=1))


if __name__ == "__main__":
    main()
