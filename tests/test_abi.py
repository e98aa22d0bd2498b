"""The C-ABI library loads and exports every symbol include/spmi.h declares, with the
argument checks that need no GPU."""
import ctypes
import os
import re

from spiral_amd import _native as nv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "spmi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(spmi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 17
    handle = ctypes.CDLL(nv.lib_path())
    for name in names:
        assert hasattr(handle, name), f"{name} declared in include/spmi.h but not exported"
    assert set(nv.EXPORTED_SYMBOLS) == set(names), "python binding and header disagree"


def test_version_and_argument_errors_without_a_gpu():
    lib = nv.lib()
    assert lib.spmi_version() >= 100
    assert lib.spmi_coo_to_csr_workspace_bytes(10_000_000, 1_000_000) > 3 * 4 * 10_000_000
    assert lib.spmi_degree_workspace_bytes(0, 1) > 0
    # negative sizes and null pointers are rejected before any HIP call
    rc = lib.spmi_coo_to_csr(None, None, -1, 5, 5, None, None, None, None, 0, None, None)
    assert rc == -1 and b"spmi_coo_to_csr" in lib.spmi_last_error()
    rc = lib.spmi_coo_to_csr(None, None, 1 << 31, 5, 5, None, None, None, None, 0, None, None)
    assert rc == -5
    rc = lib.spmi_gcn_spmm(None, None, 5, 5, 3, None, 4, 8, None, None, None, None, 0, 0.0, None, 8,
                           None, None, 0, 0, 1, 1, None, None)
    assert rc == -1
    rc = lib.spmi_gru_gate_fwd(None, None, None, None, None, None, 4, 8, None, 0, None, None)
    assert rc == -1
    rc = lib.spmi_norm_from_indptr(None, 4, 7, None, None)
    assert rc == -1
