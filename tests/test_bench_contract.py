"""The committed bench lines (profiles/rNN_bench_n1.json, produced by `python bench.py` on an MI355X) carry every field
of the driver's contract, and bench.py still emits those keys (static check of the source: no GPU here)."""
import json
import os
import re

import pytest

from helpers import ROOT

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]
ROOFLINE = ["bound", "achieved", "peak", "unit", "frac", "traffic"]
CPU = ["value", "unit", "cores", "kind", "sample"]


@pytest.mark.parametrize("name", ["r01_bench_n1.json", "r02_bench_n1.json", "r03_bench_n1.json"])
def test_committed_bench_line_has_the_contract_fields(name):
    line = open(os.path.join(ROOT, "profiles", name)).read().strip().splitlines()[-1]
    d = json.loads(line)
    for k in REQUIRED:
        assert k in d, k
    for k in ROOFLINE:
        assert k in d["roofline"], k
    for k in CPU:
        assert k in d["cpu_baseline"], k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["roofline"]["bound"] in ("hbm", "mfma") and 0.0 < d["roofline"]["frac"] < 1.0
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-9
    assert d["cpu_baseline"]["kind"] in ("port", "reference") and d["cpu_baseline"]["cores"] >= 1
    assert "workload" in d["config"] and "model" not in d["config"]
    if not name.startswith("r01"):
        assert d["roofline"]["traffic"] and d["roofline"]["launches_per_forward"] >= 1   # PMC traffic of the same sources
    # value = images of all ranks / time of the timed steps
    assert abs(d["value"] - d["config"]["batch_per_gpu"] * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6


def test_bench_source_still_emits_the_contract_keys():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for k in REQUIRED:
        assert re.search(rf'"{k}"', src), k
    assert "/root/reference" not in src
