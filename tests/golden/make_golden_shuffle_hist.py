#!/usr/bin/env python3
"""Generate tests/golden/reference_dinucl_hist.json by RUNNING the reference's dinuclShuffle many times.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_shuffle_hist.py
For a few short inputs (n <= 10) the reference's own `dinuclShuffle` (ScanFoldFunctions.py:255-277, imported with
a stub `RNA` module) is called DRAWS times after one random.seed(); the histogram over the distinct outputs is
the expected distribution of a dinucleotide shuffle (Altschul-Erikson: uniform over the sequences with the same
dinucleotide counts and the same first / last character).  Only inputs and output counts are stored.
"""
import json
import os
import random
import sys
import types

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_dinucl_hist.json")
DRAWS = 30000
INPUTS = ["AACAGAUACA", "GAGAUAUGAG", "UGUAUGGUAU", "GCGAGUACCA", "UUAUAAAAGA", "AAAGCCCGUC"]


def main():
    sys.dont_write_bytecode = True
    sys.modules.setdefault("RNA", types.ModuleType("RNA"))
    sys.path.insert(0, REF)
    import ScanFoldFunctions as sff
    items = []
    for k, s in enumerate(INPUTS):
        random.seed(4000 + k)
        hist = {}
        for _ in range(DRAWS):
            o = sff.dinuclShuffle(s)
            hist[o] = hist.get(o, 0) + 1
        items.append({"s": s, "seed": 4000 + k, "draws": DRAWS, "hist": dict(sorted(hist.items()))})
        print(s, len(hist), "distinct outputs")
    with open(OUT, "w") as f:
        json.dump({"generated_by": "tests/golden/make_golden_shuffle_hist.py", "items": items}, f, indent=1)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
