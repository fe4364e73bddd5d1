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


def test_published_file_layout_enthalpies_def_tokens_and_temperature(tmp_path):
    """A file in the real rna_turner2004.par layout (enthalpy twins after every section, block and trailing comments, a
    comment spanning lines, dG/dH column pairs, '#END' without a blank, DEF tokens): parsed tables, the DEF fallback,
    and the ViennaRNA-style rescale dG(T) = dH - (dH - dG37)(T + K0)/(37 + K0), truncated towards zero."""
    import pytest
    from par_util import par_text, synthetic_enthalpies
    def same(a, b):  # field by field (the struct's tail padding is not part of the model)
        return all(np.array_equal(a.rec[f], b.rec[f]) for f in a.rec.dtype.names)
    base = params.default_params()
    dH = synthetic_enthalpies(base.rec, 4)
    p = params.parse_par_text(par_text(base.rec, dH), source="synthetic-published-layout.par")
    assert same(p, base) and p.dH is not None
    assert (p.dH["stack"][1:8, 1:8] == dH["stack"][1:8, 1:8]).all() and (p.dH["hairpin"] == dH["hairpin"]).all()
    for f in ("mismatchM", "mismatchH", "dangle5"):
        assert (p.dH[f][1:8] == dH[f][1:8]).all(), f
    assert (p.dH["int11"][1:8, 1:8] == dH["int11"][1:8, 1:8]).all() and (p.dH["int21"][1:8, 1:8] == dH["int21"][1:8, 1:8]).all()
    assert (p.dH["tetra_E"][:16] == dH["tetra_E"][:16]).all()
    assert (p.dH["int22"][1:7, 1:7, 1:5, 1:5, 1:5, 1:5] == dH["int22"][1:7, 1:7, 1:5, 1:5, 1:5, 1:5]).all()
    assert int(p.dH["MLclosing"]) == 3000 and int(p.dH["TerminalAU"]) == 370 and int(p.dH["ninio"]) == 320
    # DEF entries keep the shipped default value
    mask = np.zeros((7, 7), dtype=bool); mask[0, 1] = mask[3, 3] = True
    hp = np.zeros(31, dtype=bool); hp[9] = True
    changed = base.rec.copy()
    changed["stack"][1, 2] = 77; changed["stack"][4, 4] = 88; changed["hairpin"][9] = 999
    q = params.parse_par_text(par_text(changed, dH, {"stack": mask, "hairpin": hp}), source="with-DEF.par")
    # ... and the parser says which sections it filled (a published file with DEF entries becomes a MIXED table here:
    # --require-published-params refuses it)
    assert q.def_substituted == {"stack": int(mask.sum()), "hairpin": int(hp.sum())} and not p.def_substituted
    assert q.rec["stack"][1, 2] == base.rec["stack"][1, 2] and q.rec["stack"][4, 4] == base.rec["stack"][4, 4]
    assert q.rec["hairpin"][9] == base.rec["hairpin"][9]
    with pytest.raises(ValueError):
        params.parse_par_text(par_text(changed, dH, {"stack": mask}), source="no-defaults.par", defaults=None)
    # temperature rescale
    t25 = p.at_temperature(25)
    tempf = (25 + 273.15) / (37 + 273.15)
    assert t25.temperature == 25.0 and abs(float(t25.rec["lxc"]) - 107.856 * tempf) < 1e-9
    for f, idx in (("stack", (1, 1)), ("stack", (3, 4)), ("mismatchH", (2, 3, 1)), ("int11", (1, 5, 2, 3)),
                   ("hairpin", (5,)), ("bulge", (1,)), ("tetra_E", (0,)), ("dangle3", (6, 4))):
        g, h = int(base.rec[f][idx]), int(dH[f][idx])
        assert int(t25.rec[f][idx]) == int(h - (h - g) * tempf), (f, idx)  # int() truncates towards zero, as the C assignment
    assert (t25.rec["hairpin"][:3] == params.INF).all() and t25.rec["bulge"][0] == params.INF
    assert int(t25.rec["MLclosing"]) == int(3000 - (3000 - 930) * tempf)
    assert int(t25.rec["max_ninio"]) == 300
    assert same(p.at_temperature(37), base)
    assert same(t25.at_temperature(37), base)                # a rescaled set remembers its 37 C tables
    with pytest.raises(NotImplementedError):
        base.at_temperature(25)                                # the reconstructed set has no enthalpies


def test_provenance_warning_names_the_reconstructed_set(capsys):
    import io
    buf = io.StringIO()
    params._warned = False
    assert params.warn_if_reconstructed(params.default_params(), stream=buf) is True
    assert "RECONSTRUCTION" in buf.getvalue() and "rna_turner2004_recon.par" in buf.getvalue()
    buf2 = io.StringIO()
    assert params.warn_if_reconstructed(params.default_params(), stream=buf2) is True and buf2.getvalue() == ""  # once
    other = params.default_params(); other.source = "/some/where/rna_turner2004.par"
    assert params.warn_if_reconstructed(other) is False
