"""The C-ABI library loads on a CPU-only host and exports every function include/*.h declares.
No compute call is made here; creating a context without a GPU must fail loudly (no CPU path)."""
import ctypes
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"^\s*(?:const\s+)?[A-Za-z_][\w\s\*]*?\b((?:socp_|hybr)\w+)\s*\(", text, flags=re.M):
            if not m.group(0).lstrip().startswith("typedef"):
                names.add(m.group(1))
    return sorted(names)


def test_headers_declare_the_expected_surface():
    names = declared_functions()
    for must in ("hybrd", "hybrj", "socp_ctx_create", "socp_integrate_batch", "socp_residual_batch",
                 "socp_fd_jacobian", "socp_fd_rows_dev", "socp_hybrd_batched", "socp_hybr_advance"):
        assert must in names, must


def test_library_exports_every_declared_symbol():
    from socp_amd import capi
    lib = ctypes.CDLL(capi.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing


def test_no_cpu_path():
    """Without a HIP device socp_ctx_create returns SOCP_ERR_NO_DEVICE with a message."""
    import torch
    from socp_amd import capi
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.SocpError) as e:
        capi.Context(capi.MODEL_GODDARD)
    assert e.value.code == capi.ERR_NO_DEVICE and "no CPU path" in str(e.value)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under socp_amd/ may reference it."""
    bad = []
    for path in glob.glob(os.path.join(ROOT, "socp_amd", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".py", ".cpp", ".hpp", ".hip", ".h", "Makefile")):
            # an import, include, link or path into oracle/ (a comment that says "the CPU oracle" is fine)
            if re.search(r"(import\s+oracle|from\s+oracle|oracle/|oracle\\.|socp_oracle|libsocp_ref|orc_[a-z])",
                         open(path, errors="ignore").read()):
                bad.append(os.path.relpath(path, ROOT))
    assert not bad, bad
