"""Host-side AddressSanitizer run of the C++ half of the library (race / memory checking).

`make -C imageretrievalresearch_amd/csrc asan` builds the same sources with -fsanitize=address on the HOST code (plan
builders arch_*.cpp, the executor's bookkeeping, argument checking, workspace carving); GPU ASan needs xnack+, which
the pool does not offer, so device code is not instrumented.  A child process preloads the ASan runtime, loads that
build through MI355_LIB_PATH and drives every entry point that works without a GPU; any heap / stack / global
overflow or use-after-free aborts the child with a report."""
import os
import shutil
import subprocess
import sys

import pytest

from helpers import ROOT

CSRC = os.path.join(ROOT, "imageretrievalresearch_amd", "csrc")
ASAN_LIB = os.path.join(ROOT, "imageretrievalresearch_amd", "libmi355_retrieval_asan.so")
HIPCC = "/opt/rocm/bin/hipcc"
CLANG = "/opt/rocm/lib/llvm/bin/clang"

CHILD = r"""
import ctypes as C
import imageretrievalresearch_amd as M
from imageretrievalresearch_amd import _lib
L = _lib.lib()
assert "asan" in _lib.LIB_PATH
n_models = 0
for name in ("efficientnet_b3a", "rexnet_100", "rexnet_130", "rexnet_150", "rexnet_200", "swin_base_patch4_window7_224"):
    for nc in (0, 7, 1000):
        m = M.create_model(name, num_classes=nc)          # plan builder + parameter table
        sd = m.state_dict()
        assert len(sd) > 100
        for B in (1, 3, 256):
            tr = m.traffic(B)                              # walks the whole plan (slot planner, fusion rules, byte model)
            assert tr["macs"] > 0 and tr["act_bytes"] > 0
        for key, val in (("microbatch", 64), ("fuse", 0), ("fuse", 1), ("fuse_block_min_batch", 1), ("profile", 1), ("profile", 0)):
            m.set_option(key, val)
        try:
            m.set_option("no_such_option", 1)
            raise SystemExit("unknown option accepted")
        except M.MI355Error:
            pass
        if nc == 7:                                        # head replacement re-creates the C handle
            import torch
            if hasattr(m, "classifier") and name.startswith("eff"):
                m.classifier = torch.nn.Linear(m.classifier.in_features, 11)
        del m
        n_models += 1
# argument checking / workspace carving (no HIP call is reached)
assert L.mi355_rank_topk(None, 1, None, 1, 8, 0, 1, 1e-6, 0, None, None, None, 0, None) != 0
for Q, G, D, k in ((1, 1, 1, 1), (5, 129, 33, 3), (256, 100000, 1536, 3), (2048, 125000, 1536, 3), (70000, 3000000, 64, 150), (3, 7, 5, 1024)):
    assert L.mi355_rank_workspace_bytes(Q, G, D, k) > 0
assert L.mi355_rank_workspace_bytes(0, 10, 8, 1) == 0
try:
    M.create_model("no_such_model")
    raise SystemExit("unknown model accepted")
except (M.MI355Error, AssertionError, ValueError, RuntimeError):
    pass
assert isinstance(L.mi355_last_error(), bytes)
print("ASAN_HOST_OK", n_models)
"""


@pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists(CLANG)), reason="needs the ROCm toolchain")
def test_host_code_is_clean_under_address_sanitizer():
    r = subprocess.run(["make", "-C", CSRC, "-j8", "asan"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and os.path.exists(ASAN_LIB), r.stdout[-2000:] + r.stderr[-2000:]
    rt = subprocess.run([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    assert os.path.exists(rt), rt
    env = dict(os.environ, LD_PRELOAD=rt, MI355_LIB_PATH=ASAN_LIB, PYTHONPATH=ROOT,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1:verify_asan_link_order=0")
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0 and "ASAN_HOST_OK" in r.stdout, (r.stdout[-3000:], r.stderr[-6000:])
    assert "AddressSanitizer" not in r.stderr, r.stderr[-6000:]
