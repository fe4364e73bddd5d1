#!/usr/bin/env python3
"""Fold stage at cfg3 scale on the GPU box: scan a 30 kb synthetic transcript (W=120, step 1, r=--r), then run the
Fold stage twice — pair tabulation on the host (numpy) and on the device (sf_tabulate_pairs) — check that every file
is byte-identical and print where the time goes.  Usage: python tools/gpu_fold_stage.py [--L 30000] [--r 20]"""
import argparse
import hashlib
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scanfold_amd import _lib, fold  # noqa: E402
from scanfold_amd import scanfold as sfd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=30000)
    ap.add_argument("--r", type=int, default=20)
    ap.add_argument("--W", type=int, default=120)
    a = ap.parse_args()
    rng = np.random.default_rng(11)
    seq = "".join("ACGU"[v] for v in rng.integers(0, 4, a.L))
    eng = _lib.get_engine()
    t0 = time.time()
    rows, table = sfd.scan_rows(seq, a.W, 1, a.r, "mono", 37, eng, 1)
    t_scan = time.time() - t0
    table.id = "rec"
    print("device", eng.device_name(), "| windows", len(table.starts), "| scan (r=%d) %.3f s" % (a.r, t_scan))
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for tag, e in (("host", None), ("device", eng), ("host", None), ("device", eng)):
            sub = os.path.join(d, tag)
            os.makedirs(sub, exist_ok=True)
            t0 = time.time()
            tab = fold.Tabulation(table) if e is None else fold.DeviceTabulation(table, e)
            t1 = time.time()
            g = tab.groups()
            t2 = time.time()
            tab.groups = lambda g=g: g
            res = fold.best_partners(tab, os.path.join(sub, "log.txt"))
            t3 = time.time()
            fold.compete(tab, res, os.path.join(sub, "final_partners.txt"))
            t4 = time.time()
            for name, f in (("no_filter", 10.0), ("-1", -1.0), ("-2", -2.0)):
                fold.write_ct(tab, res, os.path.join(sub, name + ".ct"), f, header_name=name)
            fold.write_bp(tab, res, os.path.join(sub, "x.bp"), "rec")
            t5 = time.time()
            out[tag] = {fn: hashlib.sha256(open(os.path.join(sub, fn), "rb").read()).hexdigest() for fn in sorted(os.listdir(sub))}
            print("%-6s tabulation: table arrays %.3f s, grouping + sums %.3f s (%d groups) | best partners + log %.3f s | "
                  "competition %.3f s | ct x3 + bp %.3f s | total %.3f s" % (tag, t1 - t0, t2 - t1, len(g[0]), t3 - t2, t4 - t3,
                                                                             t5 - t4, t5 - t0))
    print("files byte-identical between the two tabulations:", out["host"] == out["device"], sorted(out["host"]))
    return 0 if out["host"] == out["device"] else 1


if __name__ == "__main__":
    sys.exit(main())
