"""Structure extraction + motif refold bookkeeping (scanfold_amd/motifs.py) against tests/golden/motifs.json, which
tests/golden/make_golden_motifs.py produced by running the reference's own blocks (ScanFold.py:1582-1776) with canned
fold results.  The folds themselves are the engine's (tests/test_gpu_parity.py::test_motif_refolds_on_gpu)."""
import contextlib
import io
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden_motifs as gen  # noqa: E402  (only its canned_* helpers; main() is what needs the reference)

from scanfold_amd import motifs  # noqa: E402

G = json.load(open(os.path.join(HERE, "golden", "motifs.json")))


class CannedFolder:
    def constrained(self, frag, constraint):
        return gen.canned_fold(frag, constraint)

    def scramble(self, frag, r, shuffle_type):
        return [frag[k % len(frag):] + frag[:k % len(frag)] for k in range(1, r + 1)]

    def energies(self, seqlist):
        return gen.canned_energies(seqlist)


@pytest.mark.parametrize("k", range(len(G["cases"])))
def test_extraction_matches_reference(k):
    c = G["cases"][k]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        got = motifs.extract_structures(c["structure_line"], c["sequence"])
    assert buf.getvalue() == c["stdout"]
    assert [dict(count=m.structure_count, sequence=m.sequence, structure=m.structure, i=m.i, j=m.j) for m in got] == c["motifs"]


@pytest.mark.parametrize("k", [k for k, c in enumerate(G["cases"]) if "files" in c])
def test_refold_outputs_byte_for_byte(k, tmp_path):
    c = G["cases"][k]
    with contextlib.redirect_stdout(io.StringIO()):
        ms = motifs.extract_structures(c["structure_line"], c["sequence"])
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        recs = motifs.refold_motifs("rec|1", ms, "mono", "x.gff3", folder=CannedFolder())
        got = {fn: open(fn).read() for fn in sorted(os.listdir("."))}
    finally:
        os.chdir(cwd)
    assert got == c["files"]
    assert len(recs) == len(ms)


def test_unbalanced_line_fails_like_the_reference():
    with pytest.raises(IndexError):
        motifs.extract_structures("..((..((...))....\n", "ACGUACGUACGUACGUACGU")


def test_dbn2ct_rejects_length_mismatch(tmp_path):
    p = tmp_path / "m.dbn"
    p.write_text(">x\nACGU\n(..)..")
    with pytest.raises(TypeError):
        motifs.dbn2ct(str(p))
