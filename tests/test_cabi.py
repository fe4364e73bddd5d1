"""The C-ABI library loads, exports every symbol include/scanfold_hip.h declares, and refuses to run without a GPU."""
import os
import re

import pytest

from scanfold_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "scanfold_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sf_[a-z_0-9]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built_lib():
    import __graft_entry__
    __graft_entry__.build()
    return _lib.load_library()


def test_every_declared_symbol_is_exported_and_bound(built_lib):
    decl = declared_symbols()
    assert len(decl) >= 15
    assert sorted(_lib.EXPORTED_SYMBOLS) == decl
    for name in decl:
        assert getattr(built_lib, name) is not None


def test_error_strings(built_lib):
    assert built_lib.sf_strerror(0) == b"ok"
    for code in range(-8, 0):
        assert built_lib.sf_strerror(code) not in (b"ok", b"unknown status")
    assert built_lib.sf_strerror(-99) == b"unknown status"


def test_calls_before_init_fail_with_status_not_crash(built_lib):
    # no compute without sf_init; on a machine without a GPU sf_init itself must refuse
    import torch
    assert built_lib.sf_mfe_batch(None, 0, 120, None) in (-1, -2)
    assert built_lib.sf_prof_reset() in (-1, 0)
    if not torch.cuda.is_available():
        assert built_lib.sf_init(0) == -7  # SF_ERR_NO_DEVICE
        with pytest.raises(_lib.ScanFoldHipError):
            _lib.Engine(0)


def test_missing_library_is_loud(tmp_path):
    with pytest.raises(_lib.ScanFoldHipError):
        _lib.load_library(str(tmp_path / "nope.so"))


def test_product_never_references_oracle_or_emulation():
    pkg = os.path.join(ROOT, "scanfold_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "libsf_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, fn
                assert "libscanfold_emul" not in txt, fn
