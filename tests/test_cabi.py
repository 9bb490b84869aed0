"""The C-ABI library loads and exports every entry point include/gpsa_hip.h declares (no compute)."""
import os
import re

from spatial_alignment_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "gpsa_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gpsa_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gpsa_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)


def test_version_and_arch():
    lib = _lib.load()
    assert lib.gpsa_version() >= 100
    assert lib.gpsa_build_arch() == b"gfx950"


def test_workspace_queries_are_pure():
    lib = _lib.load()
    assert lib.gpsa_gemm_workspace(0, 200, 200, 1, 1) == 0
    assert lib.gpsa_gemm_workspace(1, 200, 200, 2, 4) == 2 * 4 * 200 * 200 * 8
    # 4 column blocks x 7 row chunks would leave the chip nearly empty: 8-row chunks (25 of them)
    assert lib.gpsa_kmat_bwd_workspace(0, 200, 1000, 2) == (4 * 400 + 25 * 2000 + 4 * 25 * 2) * 4
    assert lib.gpsa_kmat_bwd_workspace(0, 200, 100000, 2) == (391 * 400 + 7 * 200000 + 391 * 7 * 2) * 4
    assert lib.gpsa_quadform_workspace(0, 200, 1000, 50) >= 50 * 208 * 208 * 4


def test_library_is_stamped_with_its_sources():
    """the library carries the sha256 of the sources it was built from; load() refuses another one"""
    lib = _lib.load()
    stamp = lib.gpsa_source_hash().decode()
    assert stamp == "GPSA_SOURCE_HASH=" + _lib.source_hash()
    assert _lib.library_hash() == _lib.source_hash()
