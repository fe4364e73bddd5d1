"""Report of the cycle stamps a "pfstamps" build (tools/dev/abl.py) appends to $SF_STAMP_OUT: per wave of sf_pf_lds_kernel, the
average cycles of a column up to a point inside the wave's own work, the end of that work, the exit of the first barrier, the end of
the post-barrier work and the exit of the second barrier (outside pass, by bucket of the column; inside pass), and the phases of a
fold (wave 0)."""
import sys
import numpy as np

tot = np.zeros((8, 64))
for path in sys.argv[1:]:
    rows = [ln.split() for ln in open(path) if ln.strip() and not ln.startswith("launch")]
    a = np.array(rows, dtype=np.float64).reshape(-1, 8, 64)
    tot += a.sum(axis=0)
folds = tot[0, 45]
print("folds stamped %d; inside columns per fold %.2f" % (folds, tot[0, 36] / folds))
for b in range(4):
    print("outside pass, columns %d..%d (%.1f per fold), cycles per column (team = wave / 2):" % (30 * b + 1, 30 * b + 30, tot[0, b * 8 + 5] / folds))
    print("  wave  mid-point   own work   barrier-1 exit   post work   barrier-2 exit")
    for w in range(8):
        n = max(tot[w, b * 8 + 5], 1)
        print("  %4d %10.0f %10.0f %16.0f %11.0f %16.0f" % ((w,) + tuple(tot[w, b * 8 + q] / n for q in range(5))))
print("inside pass, cycles per column:")
print("  wave   own work   barrier-1 exit   post work   barrier-2 exit")
for w in range(8):
    n = max(tot[w, 36], 1)
    print("  %4d %10.0f %16.0f %11.0f %16.0f" % ((w,) + tuple(tot[w, 32 + q] / n for q in range(4))))
ph = tot[0, 40:45] / folds
names = ("reload / shift", "inside columns", "park + exterior + tables", "outside columns", "reductions + outputs")
print("phases of a fold (cycles, wave 0): " + ", ".join("%s %.0f" % (n, v) for n, v in zip(names, ph)) + "; sum %.0f" % ph.sum())
if tot[0, 51] > 0:
    print("between the passes, cycles per fold from the end of the inside columns: exterior table built (barrier passed), own part done (waves 0-1: the sweeps; 2-7: park, tables, clearing), closing barrier passed")
    for w in range(8):
        n = max(tot[w, 51], 1)
        print("  wave %d %10.0f %10.0f %10.0f" % (w, tot[w, 48] / n, tot[w, 49] / n, tot[w, 50] / n))
