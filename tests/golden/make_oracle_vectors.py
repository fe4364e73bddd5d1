#!/usr/bin/env python3
"""Freeze a few oracle outputs (default parameter set) as regression vectors: tests/golden/oracle_vectors.json.

These are NOT reference (ViennaRNA) outputs — none can be produced here; they pin the
oracle + shipped parameter file against accidental change, and give the GPU tests a committed fixture."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from scanfold_amd import params  # noqa: E402

oracle.set_params(params.default_params())
rng = np.random.default_rng(424242)
items = []
for W in (12, 30, 60, 120, 120, 120, 200):
    s = "".join("ACGU"[k] for k in rng.integers(0, 4, W))
    db, e = oracle.mfe(s)
    r = oracle.pf(s)
    items.append(dict(seq=s, mfe_dcal=e, structure=db, centroid=r["centroid"], ens_dG=r["dG"],
                      mean_bp_dist=r["mean_bp_dist"], centroid_dist=r["centroid_dist"]))
for s in ["GGGGAAAACCCC", "GGGGCUUCGGCCCC", "GGGAAAUCCCAAAGGGAAAUCCC", "A" * 40, "GC" * 30, "G" * 30 + "AAAA" + "C" * 30]:
    db, e = oracle.mfe(s)
    r = oracle.pf(s)
    items.append(dict(seq=s, mfe_dcal=e, structure=db, centroid=r["centroid"], ens_dG=r["dG"],
                      mean_bp_dist=r["mean_bp_dist"], centroid_dist=r["centroid_dist"]))
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_vectors.json")
json.dump(dict(generated_by="tests/golden/make_oracle_vectors.py", params="rna_turner2004_recon.par", items=items),
          open(out, "w"), indent=1)
print("wrote", out)
