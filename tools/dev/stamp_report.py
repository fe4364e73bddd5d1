"""Per-step cycle profile of the MFE kernel from a "stamps" build (tools/dev/abl.py stamps):

    SF_STAMP_OUT=gpurun_out/stamps.txt SCANFOLD_LIB=tools/abl_stamps.so python tools/gpu_mfe_only.py 65536 120
    python tools/dev/stamp_report.py gpurun_out/stamps.txt [W]

Cycles (s_memtime ticks, averaged over the sampled folds) per step: each wave's own pre-barrier work, the main waves' finish
between the two barriers of a split step, the barrier waits, the fML fix-up — which wave a step waits for, and how long the
others idle."""
import sys

path = sys.argv[1]
W = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rows = [list(map(int, ln.split())) for ln in open(path) if ln.strip()]
print("%4s | %s | %7s %7s %7s %7s | %s" % ("d0", "pre-barrier work of waves 0..3", "exch", "finish", "endbar", "fixup", "step (wave 0)   waits of waves 0..3 at the first barrier"))
tot = {"pre": 0.0, "exch": 0.0, "fin": 0.0, "endbar": 0.0, "fix": 0.0, "step": 0.0}
for st in range(64):
    d0 = 2 * st
    w = rows[st * 4: st * 4 + 4]
    if not w or w[0][4] == 0:
        continue
    n = [max(x[4], 1) for x in w]
    pre = [w[k][0] / n[k] for k in range(4)]
    split = w[0][5] > 0
    t_ex = [w[k][5] / n[k] if split else pre[k] for k in range(4)]   # exit of the exchange barrier
    t_fin = [w[k][1] / n[k] for k in range(4)]
    t_eb = [w[k][2] / n[k] for k in range(4)]
    t_end = [w[k][3] / n[k] for k in range(4)]
    exch = t_ex[0] - pre[0]
    fin = t_fin[0] - t_ex[0]
    endbar = t_eb[0] - t_fin[0]
    fix = t_end[0] - t_eb[0]
    waits = [(t_ex[k] - pre[k]) if split else (t_eb[k] - t_fin[k]) for k in range(4)]
    extra = ""
    if w[0][6]:
        ta = w[0][6] / n[0]
        tb = w[0][7] / n[0] if w[0][7] else ta
        extra = "   w0: control %5.0f, dml2 %5.0f, cell %5.0f" % (ta, tb - ta, pre[0] - tb)
    print("%4d | %7.0f %7.0f %7.0f %7.0f | %7.0f %7.0f %7.0f %7.0f | %7.0f   %6.0f %6.0f %6.0f %6.0f%s"
          % (d0, pre[0], pre[1], pre[2], pre[3], exch, fin, endbar, fix, t_end[0], waits[0], waits[1], waits[2], waits[3], extra))
    tot["pre"] += max(pre); tot["exch"] += exch; tot["fin"] += fin; tot["endbar"] += endbar; tot["fix"] += fix; tot["step"] += t_end[0]
print("sum over steps (wave 0's clock): step %.0f; longest pre-barrier work %.0f, exchange-barrier wait of wave 0 %.0f, finish %.0f, "
      "end-barrier wait %.0f, fix-up %.0f" % (tot["step"], tot["pre"], tot["exch"], tot["fin"], tot["endbar"], tot["fix"]))
