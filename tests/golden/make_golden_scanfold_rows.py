#!/usr/bin/env python3
"""Generate tests/golden/scanfold_py_rows.json: the scan-table header and row of ScanFold.py (lines 416 and 685),
evaluated by compiling the reference's own string expressions (lifted with `ast`; ScanFold.py cannot be imported: it
parses the command line and imports RNA / Bio at module level) on synthetic values, with the z-score, p-value and GC
content coming from the reference's ScanFoldFunctions (imported with a stub RNA module).
Run in the build container only:   python tests/golden/make_golden_scanfold_rows.py"""
import ast
import json
import os
import random
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scanfold_py_rows.json")


def main():
    sys.dont_write_bytecode = True
    sys.modules.setdefault("RNA", types.ModuleType("RNA"))
    sys.path.insert(0, REF)
    import ScanFoldFunctions as sff
    tree = ast.parse(open(os.path.join(REF, "ScanFold.py")).read())
    exprs = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Expr) and isinstance(node.value, ast.Call):
            fn = node.value.func
            if isinstance(fn, ast.Attribute) and fn.attr == "write" and getattr(fn.value, "id", "") == "w":
                if node.lineno == 416:
                    exprs["header"] = compile(ast.Expression(node.value.args[0]), "hdr", "eval")
                if node.lineno == 685:
                    exprs["row"] = compile(ast.Expression(node.value.args[0]), "row", "eval")
    assert set(exprs) == {"header", "row"}
    rng = random.Random(7)
    nprng = np.random.default_rng(7)
    items = []
    for k in range(16):
        W = 30 if k % 2 else 120
        frag = "".join(rng.choice("ACGU") for _ in range(W))
        if k == 3:
            frag = "AUAUAU" * 5
        r = [5, 10, 30, 100][k % 4]
        E = [float(np.float32(v)) for v in np.round(nprng.normal(-25, 4, r + 1), 1)]
        if k == 6:
            E = [E[0]] * (r + 1)
        try:
            zscore = round(sff.zscore_function(E, r), 2)
        except Exception:
            zscore = sff.zscore_function(E, r)
        env = dict(start_nucleotide=1 + 3 * k, end_nucleotide=3 * k + W, temperature=37, MFE=round(E[0], 2), zscore=zscore,
                   pvalue=round(sff.pvalue_function(E, r), 2), ED=round(float(nprng.uniform(0, 40)), 2), frag=frag,
                   structure="." * W, centroid="(" + "." * (W - 2) + ")", gc_content=sff.get_gc_content(frag),
                   read_name="rec%d" % k, str=str)
        item = {kk: vv for kk, vv in env.items() if kk not in ("str", "__builtins__")}
        item["energy_list"] = E
        item["r"] = r
        item["row"] = eval(exprs["row"], env)
        item["header"] = eval(exprs["header"], env)
        items.append(item)
    with open(OUT, "w") as f:
        json.dump({"generated_by": "tests/golden/make_golden_scanfold_rows.py", "items": items}, f, indent=1)
    print("wrote", OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    main()
