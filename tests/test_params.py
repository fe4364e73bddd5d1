"""Parameter file parsing, blob layout and the table symmetries that need no oracle."""
import ctypes
import os
import subprocess

import numpy as np

from scanfold_amd import params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_blob_size_matches_c_struct(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "%s/include/sf_params_blob.h"\n'
                   'int main(){printf("%%zu", sizeof(sf_params_blob));return 0;}\n' % ROOT)
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", str(src), "-o", str(exe)])
    assert int(subprocess.check_output([str(exe)])) == params.BLOB_DTYPE.itemsize == len(params.default_params().blob())


def test_default_set_constants_and_symmetries():
    r = params.default_params().rec
    assert r["magic"] == params.MAGIC and r["temperature"] == 37.0
    assert r["MLclosing"] == 930 and r["MLbase"] == 0 and (r["MLintern"][1:7] == -90).all()
    assert r["TerminalAU"] == 50 and r["ninio"] == 60 and r["max_ninio"] == 300 and abs(r["lxc"] - 107.856) < 1e-9
    assert (r["hairpin"][:3] == params.INF).all() and r["bulge"][0] == params.INF
    assert (r["internal_loop"][:2] == params.INF).all()
    st = r["stack"][1:7, 1:7]
    assert (st == st.T).all() and st[0, 0] == -240 and st[1, 1] == -340 and st[0, 1] == -330
    i11 = r["int11"][1:7, 1:7]
    assert (i11 == i11.transpose(1, 0, 3, 2)).all()
    i22 = r["int22"][1:7, 1:7, 1:5, 1:5, 1:5, 1:5]
    assert (i22 == i22.transpose(1, 0, 4, 5, 2, 3)).all()
    for f in ("mismatchM", "mismatchExt", "dangle5", "dangle3"):
        assert (r[f][1:7] <= 0).all(), f
    assert r["n_tetra"] == 16 and r["n_tri"] == 2 and r["n_hexa"] == 4
    assert bytes(r["tetra_seq"][0]) .rstrip(b"\0") == b"CAACGG" and r["tetra_E"][0] == 550


def test_par_roundtrip_and_errors(tmp_path):
    text = open(params.DEFAULT_PAR).read()
    p = params.parse_par_text(text)
    assert p.blob() == params.default_params().blob()
    import pytest
    with pytest.raises(ValueError):
        params.parse_par_text("not a par file")
    with pytest.raises(ValueError):
        params.parse_par_text(text.replace("# stack\n", "# stackXX\n"))


def test_generator_script_reproduces_shipped_par(tmp_path):
    out = tmp_path / "gen.par"
    subprocess.check_call(["python", os.path.join(ROOT, "tools", "make_recon_par.py"), str(out)])
    assert out.read_text() == open(params.DEFAULT_PAR).read()


def test_random_params_respect_symmetries():
    r = params.random_params(3).rec
    assert (r["stack"] == r["stack"].T).all()
    assert (r["int11"] == r["int11"].transpose(1, 0, 3, 2)).all()
    assert (r["int22"] == r["int22"].transpose(1, 0, 4, 5, 2, 3)).all()
