#!/usr/bin/env python3
"""Generate tests/golden/fold_cases.json by RUNNING the reference's Fold stage (ScanFold-Fold.py needs neither
ViennaRNA nor Biopython: it reads a Scan TSV and writes its consensus structure files).

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_fold.py
For every case: a Scan TSV (windows folded by the CPU oracle on the oracle's own shuffles — any TSV would do, the Fold
stage only reads it) is written to a scratch directory, `python /root/reference/ScanFold-Fold.py -i scan.tsv` runs
there, and the TSV plus every file the reference wrote is stored.  Only data: inputs and the reference's outputs.
The big per-nucleotide log is stored as its SHA-256 and its first 60 lines.
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/ScanFold-Fold.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fold_cases.json")


def make_tsv(L, W, step, r, seed, plant):
    from oracle import oracle
    from scanfold_amd import params, scan as sc
    oracle.build()
    oracle.set_params(params.default_params())
    rng = np.random.default_rng(seed)
    seq = "".join("ACGU"[k] for k in rng.choice(4, L, p=[0.3, 0.2, 0.2, 0.3]))
    if plant:
        stem = "GGCGCGGCACCGUCC"[:plant]
        comp = stem[::-1].translate(str.maketrans("ACGU", "UGCA"))
        hp = stem + "GAAA" + comp
        seq = seq[:L // 3] + hp + seq[L // 3 + len(hp):]
        seq = seq[:L]
    starts = sc.window_starts(len(seq), W, step)
    n = len(starts)
    rows = np.frombuffer(b"NACGU", dtype=np.uint8)[oracle.shuffle_windows(seq, W, step, 0, n, r, 1, seed)]
    res = oracle.scan_windows(rows, n, r)
    tsv = sc.rows_from_results(seq, starts, W, r, 37, res["energies"], res["structure"], res["centroid"], res["ens_div"])
    return sc.header_line("case%d" % seed) + "".join(tsv)


def main():
    cases = []
    for (L, W, step, r, seed, plant) in [(220, 40, 1, 15, 1, 12), (300, 60, 7, 12, 2, 15), (160, 30, 1, 10, 3, 0),
                                         (260, 50, 3, 20, 4, 14)]:
        tsv = make_tsv(L, W, step, r, seed, plant)
        with tempfile.TemporaryDirectory() as d:
            with open(os.path.join(d, "scan.tsv"), "w") as f:
                f.write(tsv)
            subprocess.run([sys.executable, REF, "-i", "scan.tsv"], cwd=d, check=True, capture_output=True)
            outputs = {}
            for name in sorted(os.listdir(d)):
                if name == "scan.tsv":
                    continue
                text = open(os.path.join(d, name)).read()
                if name.endswith(".log.txt"):
                    outputs[name] = {"sha256": hashlib.sha256(text.encode()).hexdigest(), "lines": text.count("\n"),
                                     "head": "".join(text.splitlines(True)[:60])}
                else:
                    outputs[name] = text
        # the competition-allowed mode (ScanFold-Fold.py -c 0, :1022-1038): DP files of the best partners instead of CT files
        with tempfile.TemporaryDirectory() as d:
            with open(os.path.join(d, "scan.tsv"), "w") as f:
                f.write(tsv)
            subprocess.run([sys.executable, REF, "-i", "scan.tsv", "-c", "0"], cwd=d, check=True, capture_output=True)
            outputs_c0 = {}
            for name in sorted(os.listdir(d)):
                if name == "scan.tsv" or name.endswith(".log.txt"):  # (the log is the same file as with -c 1)
                    continue
                outputs_c0[name] = open(os.path.join(d, name)).read()
        cases.append({"L": L, "W": W, "step": step, "r": r, "seed": seed, "tsv": tsv, "outputs": outputs,
                      "outputs_c0": outputs_c0})
        print("case", seed, "files:", len(outputs), "+", len(outputs_c0), "with -c 0")
    with open(OUT, "w") as f:
        json.dump({"generated_by": "tests/golden/make_golden_fold.py", "reference_cmd": "python ScanFold-Fold.py -i scan.tsv  (outputs_c0: ... -c 0)",
                   "cases": cases}, f, indent=0)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
