"""The development tools that read or patch the kernel sources stay in step with them (no GPU, no compiler needed):
tools/isa_sections.py's `// @section` markers, the text anchors of tools/dev/abl.py's variant builds, tools/run_round.sh."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_section_markers_of_the_mfe_kernel_are_in_place():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_sections
    secs = isa_sections.load_sections(os.path.join(ROOT, "scanfold_amd", "csrc", "sf_mfe_fast.hip.h"))
    names = [n for _, _, n in secs]
    for want in ("cell_setup", "publish_terms", "generic_recurrence", "special_loops", "bulge_1xn", "multiloop_split_1cell",
                 "hairpin", "generic_minima", "multiloop_closing", "finish_publish", "multiloop_split_2cell", "traceback",
                 "exterior_sweep_native", "trailing_sweep", "kernel_prologue", "fold_prologue", "cell_list", "step_control",
                 "exchange", "fml_fixup", "fold_epilogue"):
        assert want in names, want
    # sections tile the file from the first marker on, in order
    for (a0, b0, _), (a1, _, _) in zip(secs, secs[1:]):
        assert a0 <= b0 and a1 == b0 + 1


def test_every_variant_patch_of_abl_py_still_finds_its_anchor():
    sys.path.insert(0, os.path.join(ROOT, "tools", "dev"))
    import abl
    missing = []
    for name, (patches, _flags) in abl.VARIANTS.items():
        texts = {}
        for rel, anchor, _repl in patches:
            if rel not in texts:
                texts[rel] = open(os.path.join(ROOT, rel)).read()
            n = texts[rel].count(anchor)
            if n != 1:
                missing.append((name, rel, n, anchor[:50]))
            else:
                texts[rel] = texts[rel].replace(anchor, _repl)  # later patches of a variant see the earlier ones applied
    assert not missing, missing


def test_run_round_script_parses_and_knows_its_modes():
    sh = os.path.join(ROOT, "tools", "run_round.sh")
    assert subprocess.run(["bash", "-n", sh]).returncode == 0
    p = subprocess.run(["bash", sh, "nonsense", "r00"], capture_output=True, text=True)
    assert p.returncode == 2 and "measure|final|sweeps|collect" in p.stdout
