"""C-ABI surface: the library loads without a GPU and exports exactly what include/*.h declares."""
import ctypes
import subprocess

from helpers import header_symbols
from imageretrievalresearch_amd import _lib


def test_header_matches_binding_table():
    assert header_symbols() == sorted(_lib.PROTOTYPES), "include/mi355_retrieval.h and _lib.PROTOTYPES differ"


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_symbols():
        assert hasattr(L, name), f"{name} declared in the header but not exported"
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln and "mi355_" in ln}
    assert set(header_symbols()) <= exported


def test_loads_and_reports_version_without_gpu():
    L = _lib.lib()
    assert L.mi355_abi_version() == 3
    assert L.mi355_device_count() >= 0
    assert isinstance(L.mi355_last_error(), bytes)


def test_argument_errors_do_not_need_a_gpu():
    L = _lib.lib()
    # bad arguments are rejected before any HIP call, with a message
    assert L.mi355_rank_topk(None, 1, None, 1, 8, 0, 1, 1e-6, 0, None, None, None, 0, None) != 0
    assert b"null" in L.mi355_last_error()
    # k <= 8 selects inside the GEMM epilogue: no Q x G score slab in the workspace (candidates are Q * ceil(G/128) * k * 8 B)
    assert L.mi355_rank_workspace_bytes(256, 100000, 1536, 3) < 16 * 2**20
    assert L.mi355_rank_workspace_bytes(2048, 125000, 1536, 3) < 64 * 2**20      # the 8-GPU shape (all-gathered queries)
    assert L.mi355_rank_workspace_bytes(256, 100000, 1536, 150) > 256 * 100000 * 4   # k > 8 keeps the slab path
    assert L.mi355_rank_workspace_bytes(0, 10, 8, 1) == 0


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under imageretrievalresearch_amd/ may import it (a product path that
    routes through the oracle would void every parity claim); bench.py may, inside its cpu_baseline leg only."""
    import ast
    import os
    from helpers import ROOT
    pkg = os.path.join(ROOT, "imageretrievalresearch_amd")
    for fn in sorted(os.listdir(pkg)):
        if not fn.endswith(".py"):
            continue
        tree = ast.parse(open(os.path.join(pkg, fn)).read())
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            assert not any(n == "oracle" or n.startswith("oracle.") for n in names), (fn, names)
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    for node in tree.body:                               # module level: no oracle import
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            names = [a.name for a in node.names] if isinstance(node, ast.Import) else [node.module or ""]
            assert not any(n.startswith("oracle") for n in names), names
    fn_imports = {f.name: [n for n in ast.walk(f) if isinstance(n, ast.ImportFrom) and (n.module or "").startswith("oracle")]
                  for f in tree.body if isinstance(f, ast.FunctionDef)}
    assert [k for k, v in fn_imports.items() if v] == ["cpu_baseline"], fn_imports
