#!/usr/bin/env python3
"""Generate tests/golden/reference_python_half.json by RUNNING the reference's pure-Python half.

Run in the build container only (needs /root/reference; the GPU box has neither it nor this need):
    python tests/golden/make_golden.py
What is captured (inputs + the reference's outputs, nothing else — no reference source text):
  * ScanFoldFunctions (imported with a stub `RNA` module): dinuclShuffle / scramble(di) / randomizer under
    fixed random.seed(), zscore_function, pvalue_function, get_gc_content, simple_transcribe,
    get_dinucleotide_counts;
  * ScanFold-Scan.py (cannot be imported: argparse + open() at import time): its inline functions are
    lifted with `ast` and executed; its header/row string expressions (lines 350, 442) and its window-loop
    condition (line 356) are compiled from the AST and evaluated on synthetic values.
The folding calls (RNA.*) cannot be captured: ViennaRNA is not installed.
"""
import ast
import json
import os
import random
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_python_half.json")


def load_sff():
    sys.dont_write_bytecode = True
    sys.modules.setdefault("RNA", types.ModuleType("RNA"))
    sys.path.insert(0, REF)
    import ScanFoldFunctions as sff
    return sff


def lift_scan():
    src = open(os.path.join(REF, "ScanFold-Scan.py")).read()
    tree = ast.parse(src)
    ns = {"np": np, "random": random}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name not in ("multiprocessing", "rna_folder", "energies"):
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, "ScanFold-Scan.py", "exec"), ns)
    exprs = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Expr) and isinstance(node.value, ast.Call):
            fn = node.value.func
            if isinstance(fn, ast.Attribute) and fn.attr == "write" and getattr(fn.value, "id", "") == "w":
                if node.lineno == 350:
                    exprs["header"] = compile(ast.Expression(node.value.args[0]), "hdr", "eval")
                if node.lineno == 442:
                    exprs["row"] = compile(ast.Expression(node.value.args[0]), "row", "eval")
        if isinstance(node, ast.While) and node.lineno == 356:
            exprs["while"] = compile(ast.Expression(node.test), "while", "eval")
    assert set(exprs) == {"header", "row", "while"}, exprs.keys()
    return ns, exprs


def rseq(rng, n, alphabet="ACGU"):
    return "".join(rng.choice(alphabet) for _ in range(n))


def main():
    sff = load_sff()
    scan, exprs = lift_scan()
    rng = random.Random(20261003)
    G = {"generated_by": "tests/golden/make_golden.py", "python": sys.version.split()[0], "numpy": np.__version__}

    # ---- shuffles under a fixed seed ----
    G["dinucl"] = []
    for k in range(24):
        n = rng.choice([12, 30, 76, 120, 200])
        s = rseq(rng, n) if k % 5 else rseq(rng, n, "ACGT")
        if k == 3:
            s = "A" * 20 + "C" * 20 + "G" * 20
        if k == 4:
            s = "AU" * 30
        s_in = s.replace("T", "U")  # dinuclShuffle itself needs RNA letters (a raw 'T' breaks upstream's walk)
        seed = 1000 + k
        random.seed(seed)
        a = sff.dinuclShuffle(s_in)
        st = random.getstate()[1][:3]
        random.seed(seed)
        b = scan["dinuclShuffle"](s_in)
        assert a == b
        G["dinucl"].append({"seed": seed, "s": s_in, "out": a, "state_after": list(st)})
    G["scramble_di"] = []
    for k in range(4):
        text = rseq(rng, 60, "ACGT")
        random.seed(50 + k)
        G["scramble_di"].append({"seed": 50 + k, "text": text, "r": 5, "out_sff": sff.scramble(text, 5, "di")})
        random.seed(50 + k)
        G["scramble_di"][-1]["out_scan_on_transcribed"] = scan["scramble"](text.replace("T", "U"), 5, "di")
    G["randomizer"] = []
    for k in range(6):
        frag = rseq(rng, rng.choice([10, 60, 120]))
        random.seed(7 + k)
        G["randomizer"].append({"seed": 7 + k, "frag": frag, "out": sff.randomizer(frag)})

    # ---- statistics ----
    G["zscore"] = []
    nprng = np.random.default_rng(5)
    for k in range(60):
        r = [1, 2, 5, 10, 30, 100][k % 6]
        vals = np.round(nprng.normal(-25, 4, r + 1), 1)
        E = [float(np.float32(v)) for v in vals]  # values shaped like (float)dcal/100
        if k % 13 == 0:
            E = [E[0]] * (r + 1)
        if r >= 2 and k % 7 == 0:
            E[2] = E[0]
        item = {"energy_list": E, "r": r}
        try:
            item["sff"] = sff.zscore_function(E, r)
        except Exception as e:  # statistics.StatisticsError for 1-2 points, as upstream
            item["sff_error"] = type(e).__name__
        z = scan["zscore_function"](E, r)
        item["scan"] = z if isinstance(z, str) else float(z)
        try:
            item["scan_rounded_str"] = str(round(z, 2))
        except Exception:
            item["scan_rounded_str"] = str(z)
        item["pvalue"] = sff.pvalue_function(E, r)
        item["pscore"] = scan["pscore_function"](E, r)
        item["pscore_rounded_str"] = str(round(scan["pscore_function"](E, r), 2))
        G["zscore"].append(item)

    # ---- small helpers ----
    G["helpers"] = []
    for frag in ["GGGAAACCC", "AUAUAU", "acgu", "GATTACA", "CCCC", "GGGG", "NNNN", "", "ACGUNNACGU"]:
        G["helpers"].append({"frag": frag, "gc": sff.get_gc_content(frag),
                             "transcribe": sff.simple_transcribe(frag),
                             "dicounts": sff.get_dinucleotide_counts(frag)})

    # ---- TSV strings from the reference's own expressions ----
    G["tsv"] = []
    for k in range(12):
        W = 120 if k % 2 else 30
        frag = rseq(rng, W)
        E = [float(np.float32(v)) for v in np.round(nprng.normal(-25, 4, 11), 1)]
        if k == 5:
            E = [E[0]] * 11
        z = scan["zscore_function"](E, 10)
        try:
            zscore = round(z, 2)
        except Exception:
            zscore = z
        env = dict(start_nucleotide=1 + 7 * k, end_nucleotide=7 * k + W, temperature=37,
                   MFE=round(E[0], 2), zscore=zscore, pscore=round(scan["pscore_function"](E, 10), 2),
                   ED=round(float(nprng.uniform(0, 40)), 2), frag=frag,
                   structure="." * W, centroid="(" + "." * (W - 2) + ")", read_name="rec%d" % k, str=str)
        if k == 7:
            env.update(MFE=int(0.0), zscore="#DIV/0", pscore=int(0.0), ED=int(0.0))
        row = eval(exprs["row"], env)
        hdr = eval(exprs["header"], env)
        item = {kk: (vv if not isinstance(vv, (np.floating,)) else float(vv)) for kk, vv in env.items()
                if kk not in ("str", "__builtins__")}
        item["energy_list"] = E
        item["row"] = row
        item["header"] = hdr
        G["tsv"].append(item)

    # ---- window loop ----
    G["windows"] = []
    for (L, W, step) in [(1000, 120, 40), (10000, 120, 10), (120, 120, 1), (121, 120, 5), (500, 120, 7), (130, 30, 1)]:
        starts = []
        env = {"i": 0, "length": L, "window_size": W}
        while eval(exprs["while"], env):
            starts.append(env["i"])
            env["i"] += step
        G["windows"].append({"L": L, "W": W, "step": step, "n": len(starts), "first": starts[:3], "last": starts[-1]})

    with open(OUT, "w") as f:
        json.dump(G, f, indent=1, sort_keys=True)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
