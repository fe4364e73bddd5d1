"""ScanFold.py's scan stage + Fold stage on the HIP engine (the combined driver).

Restates the scan section of /root/reference/ScanFold.py (:420-757) — the ScanFoldFunctions flavour of the hot loop:
  z-score with the SAMPLE standard deviation and 0.0 when it vanishes      ScanFoldFunctions.py:741-751
  an eleventh column, the window's GC content                               ScanFold.py:433,685; SFF:1023-1036
  folds at md.temperature incl. the shuffles                                SFF:774-789  (ScanFold-Scan.py's differ, F8)
  md.max_bp_span = --span                                                   ScanFold.py:214-215
  --constraints: constrained native fold; --react: Deigan SHAPE term for the MFE only   ScanFold.py:508-544
and then feeds the windows straight into the Fold stage (scanfold_amd.fold; ScanFold.py:564-677,1036-1453) without a
TSV round trip; the pair tabulation of that stage runs on the GPU (sf_tabulate_pairs).  Output: `<name>.win_W.stp_S.rnd_R.shfl_T.out` (the scan table, ScanFold.py:381,685) and the Fold
stage's files with the prefix `<that>.ScanFold.`, then the dot-bracket files (makedbn) and the motif extraction /
refolds of ScanFold.py:1582-1776 (scanfold_amd.motifs: `<that>.ExtractedStructures.gff3`, `<that>_motif_<n>.dbn/.ct`).
ScanFold.py's per-record directories, IGV wig exports and the full-length global refold are not reproduced.
"""
import argparse
import os
import statistics
import sys

import numpy as np

from . import _lib, fold as foldmod
from . import functions as sff
from . import scan as scanmod
from . import motifs as motifmod
from . import writers


def header_line(read_name):
    return ("i\tj\tTemperature\tNative_dG\tZ-score\tP-score\tEnsembleDiversity\tSequence\tStructure\tCentroid\t"
            + read_name + "\n")


def zscores_rows_sff(E, randomizations):
    """zscore_function of ScanFoldFunctions.py:741-751 for every row of E (n, r+1), rounded as the caller does
    (round(z, 2) on a Python float).  The `statistics` module works in exact rational arithmetic, numpy in floating
    point; the two can differ in the last bits of z, which only matters when z sits on a rounding boundary — those
    rows (and the ones whose deviation vanishes) are recomputed with the reference's own calls."""
    E = np.asarray(E, dtype=np.float64)
    n = len(E)
    if E.shape[1] < 2:
        raise statistics.StatisticsError("variance requires at least two data points")
    sd = E.std(axis=1, ddof=1)
    if randomizations > 1:
        mean = E[:, 1:randomizations].mean(axis=1)
    else:
        raise statistics.StatisticsError("mean requires at least one data point")
    with np.errstate(divide="ignore", invalid="ignore"):
        z = (E[:, 0] - mean) / sd
    frac = np.abs(z * 100.0 - np.round(z * 100.0))
    exact_rows = np.nonzero(~np.isfinite(z) | (np.abs(frac - 0.5) < 1e-6) | (sd < 1e-9))[0]
    out = [round(v, 2) for v in z.tolist()]
    for k in exact_rows.tolist():
        out[k] = round(sff.zscore_function([float(v) for v in E[k]], randomizations), 2)
    return out


def gc_contents(tseq, starts, W):
    return [sff.get_gc_content(tseq[i:i + W]) for i in starts]


def scan_rows(seq, W, step, r, shuffle_type, temperature=37, engine=None, seed=0, constraints=None, reactivities=None,
              slope=0.8, intercept=-0.2, unbalanced="error"):
    """-> (rows, table): the TSV rows of ScanFold.py's `.out` file and the in-memory ScanTable of ALL windows."""
    from . import RNA
    eng = engine if engine is not None else _lib.get_engine()
    if shuffle_type not in ("di", "mono"):
        raise ValueError('Shuffle type not properly designated; please input "di" or "mono"')
    eng.set_temperature(int(temperature))
    tseq = scanmod.transcribe(seq)
    starts = scanmod.window_starts(len(seq), W, step)
    n_win = len(starts)
    kind = _lib.SHUFFLE_DI if shuffle_type == "di" else _lib.SHUFFLE_MONO
    plain = constraints is None and reactivities is None

    def work(w0, nw):
        if plain:
            return eng.scan(seq, W, step, w0, nw, r, kind, seed, raw=True)
        en = eng.scan(seq, W, step, w0, nw, r, kind, seed, _lib.SCAN_NO_PF | _lib.SCAN_NO_TRACE, raw=True)["energies"]
        wins = scanmod._window_rows(tseq, W, step, w0, nw)
        if constraints is not None:  # hc_add_from_db, then mfe and pf (ScanFold.py:508-520)
            cons = scanmod._window_rows(constraints, W, step, w0, nw)
            if unbalanced == "ignore":
                cons = scanmod._drop_unmatched_brackets(cons)
            fc = eng.fold_constrained(wins, cons)
            return dict(energies=en, native=fc["mfe"], structure=fc["structure"], centroid=fc["centroid"],
                        ens_div=fc["mean_bp_dist"])
        # SHAPE: pf first (unconstrained), then the Deigan term, then mfe (ScanFold.py:522-544)
        pf = eng.pf_batch(wins)
        sc = np.stack([RNA.deigan_pseudo_energies(reactivities[starts[w0 + t] + 1:starts[w0 + t] + W + 1], slope,
                                                  intercept, W) for t in range(nw)])
        fc = eng.fold_constrained(wins, None, sc, pf=False)
        return dict(energies=en, native=fc["mfe"], structure=fc["structure"], centroid=pf["centroid"],
                    ens_div=pf["mean_bp_dist"])

    rows, t_mfe, t_z, t_p, t_ed, t_struct = [], [], [], [], [], []
    for w0, res in scanmod._engine_chunks(work, 0, n_win):
        sub = starts[w0:w0 + len(res["ens_div"])]
        E = scanmod.dcal_to_float(res["energies"])
        nat = E[:, 0] if "native" not in res else scanmod.dcal_to_float(res["native"])
        mfe = [round(v, 2) for v in nat.tolist()]
        zs = zscores_rows_sff(E, r)
        ps = [round(v, 2) for v in scanmod.sfn.pscores_rows(E).tolist()]
        eds = [round(v, 2) for v in np.asarray(res["ens_div"], dtype=np.float64).tolist()]
        structs = scanmod._text_rows(res["structure"], W)
        cens = scanmod._text_rows(res["centroid"], W)
        gcs = gc_contents(tseq, sub, W)
        t = str(int(temperature))
        for k, i in enumerate(sub):
            frag = tseq[i:i + W]
            if frag == scanmod.ALL_N_120:  # ScanFold.py:486-492
                rows.append("%d\t%d\t%s\t0\t#DIV/0\t0\t0\t%s\t%s\t%s\t%s\n" % (i + 1, i + W, t, frag, scanmod.DOTS_120,
                                                                               scanmod.DOTS_120, str(gcs[k])))
                mfe[k], zs[k], ps[k], eds[k], structs[k] = 0.0, float("nan"), 0.0, 0.0, scanmod.DOTS_120
                continue
            rows.append("%d\t%d\t%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\n" % (i + 1, i + W, t, str(mfe[k]), str(zs[k]), str(ps[k]),
                                                                           str(eds[k]), frag, structs[k], cens[k], str(gcs[k])))
        t_mfe += mfe; t_z += zs; t_p += ps; t_ed += eds; t_struct += structs
    table = foldmod.ScanTable("", [i + 1 for i in starts], t_mfe, t_z, t_ed, [tseq[i:i + W] for i in starts], t_struct)
    table.pvalues = t_p  # the fourth metric list of ScanFold.py (:691): only the scan-pvalue wig track reads it
    return rows, table


def read_reactivities(path):
    """getShapeDataFromFile (ScanFold.py:218-262): 1-based list with -999.0 at index 0 and for missing positions / NA."""
    vec = [-999.0]
    count = 1
    with open(path, "r") as f:
        lines = f.read().splitlines()
    ncol = len(lines[0].split("\t"))
    if ncol not in (2, 3):
        raise TypeError("exceptions must derive from BaseException")  # upstream: raise("Trouble parsiging reactivity data")
    for line in lines:
        parts = line.split("\t")
        pos = int(parts[0])
        value = parts[2] if ncol == 3 else parts[1]
        if value == "NA":
            value = -999
        if pos != count:
            vec.extend([-999.0] * (pos - count))
            count = pos
        vec.append(float(value))
        count += 1
    return vec


def build_parser():
    p = argparse.ArgumentParser(description="ScanFold (scan + fold) on the MI355X HIP engine")
    p.add_argument('filename', type=str, help='input fasta')
    p.add_argument('--react', type=str, help='input SHAPE reactivity file')
    p.add_argument('-m', type=float, default=0.8, help='SHAPE slope value')
    p.add_argument('-b', type=float, default=-0.2, help='SHAPE intercept value')
    p.add_argument('--shapeD', action='store_true')
    p.add_argument('--shapeZ', action='store_true')
    p.add_argument('-f', type=int, default=-2, help='filter value')
    p.add_argument('-c', type=int, default=1, help='Competition (1 for disallow competition, 0 for allow; 1 by default)')
    p.add_argument('--name', type=str, default="UserInput", help='name of data being analyzed (chrom= of the wig tracks)')
    p.add_argument('--final_partners_wig', type=str, default="./IGV_BP_Zavg_metrics", help='final partners wig file path')
    p.add_argument('-s', type=int, default=1, help='step size')
    p.add_argument('-w', type=int, default=120, help='window size')
    p.add_argument('-r', type=int, default=100, help='randomizations')
    p.add_argument('-t', type=int, default=37, help='Folding temperature')
    p.add_argument('--type', type=str, default='mono', help='randomization type')
    p.add_argument('--print_random', action='store_true')
    p.add_argument('--algo', type=str, default='rnafold')
    p.add_argument('--constraints', type=str, help='optional | input constraint file')
    p.add_argument('--span', type=int, help='Max bp span')
    p.add_argument('--dont_fold', action='store_true', help='scan only')
    p.add_argument('--dont_extract', action='store_true', help='no motif extraction / refolds after the Fold stage')
    p.add_argument('--seed', type=int, default=0)
    p.add_argument('--params', type=str, default=None)
    p.add_argument('--constraint-unbalanced', choices=("error", "ignore"), default="error")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.algo != "rnafold":
        raise NameError("name 'RNAstructure' is not defined")  # upstream's rnastructure backend is never imported (ScanFold.py:40,439)
    if args.c not in (0, 1):
        raise ValueError("-c must be 1 (no competition allowed) or 0 (competition allowed)")
    from . import params as _params
    eng = _lib.get_engine()
    eng.set_max_bp_span(args.span or 0)
    if args.params:
        eng.load_params(_params.load_par(args.params))
    W, step, r = int(args.w), int(args.s), int(args.r)
    for read_name, seq in scanmod.read_fasta(args.filename):
        seq = scanmod.transcribe(seq)
        if "-" in seq:
            raise TypeError("exceptions must derive from BaseException")  # upstream: raise("Gaps found in sequence...")
        outname = read_name + ".win_" + str(W) + ".stp_" + str(step) + ".rnd_" + str(r) + ".shfl_" + str(args.type)
        print("Output name=" + str(outname))
        if len(seq) < W:
            print(read_name + " sequence is less than window size. Moving on to next entry.")
            continue
        cons = react = None
        if args.constraints is not None:
            print("Considering constraint input")
            cons = scanmod.read_constraints(args.constraints, len(seq))
        if args.react is not None:
            print("Considering SHAPE reactivity input")
            if args.shapeZ and not args.shapeD:
                raise TypeError("sc_add_SHAPE_zarringhalam() missing required arguments: b, default_value, shape_conversion")
            react = read_reactivities(args.react)
        rows, table = scan_rows(seq, W, step, r, args.type, args.t, eng, args.seed, cons, react, args.m, args.b,
                                args.constraint_unbalanced)
        with open(outname + ".out", "w") as w:
            w.write(header_line(read_name))
            w.writelines(rows)
        if not args.dont_fold:
            table.id = read_name
            # ScanFold.py tabulates every window (its inline loop has no dropped first row, unlike ScanFold-Fold.py)
            if args.c == 0:  # competition allowed: DP files + the track of the best partners, no CT / dbn / motifs (ScanFold.py:1454-1465)
                foldmod.fold(table, outname + ".ScanFold.", filt=int(args.f), bp_path=outname + ".ALL.bp", engine=eng,
                             competition=0)
            else:
                tab, res = foldmod.fold(table, outname + ".ScanFold.", filt=int(args.f), bp_path=outname + ".bp", engine=eng)
                for tag, label in (("no_filter", "NoFilter"), ("-1", "Zavg_-1"), ("-2", "Zavg_-2")):
                    writers.makedbn(outname + ".ScanFold." + tag, label)  # ScanFold.py:1487-1489
                # per-nucleotide mean z-score of the final partners, the track of the best partners (ScanFold.py:1491-1492)
                writers.write_wig_dict(res.fin_z.tolist(), args.final_partners_wig + "." + outname + ".wig", args.name, step)
                foldmod.write_bp(tab, res, outname + ".ALL.bp", tab.id, best=True)
            # the scanned sequence and the four per-window metric tracks (ScanFold.py:1494-1500)
            writers.write_fasta(seq, args.name + "." + outname + ".fa", args.name)
            zlist = [("#DIV/0" if z != z else z) for z in table.z.tolist()]  # (all-N windows, ScanFold.py:486-492)
            writers.write_wig(table.mfe.tolist(), step, args.name, outname + ".scan-MFE.wig")
            writers.write_wig(zlist, step, args.name, outname + ".scan-zscores.wig")
            writers.write_wig(table.pvalues, step, args.name, outname + ".scan-pvalue.wig")
            writers.write_wig(table.ed.tolist(), step, args.name, outname + ".scan-ED.wig")
            if args.c == 1 and not args.dont_extract:
                with open(outname + ".ScanFold.-2.dbn") as f:
                    line = f.readlines()[2]
                found = motifmod.extract_structures(line, seq)
                motifmod.refold_motifs(read_name, found, args.type, outname + ".ExtractedStructures.gff3",
                                       folder=motifmod.EngineFolder(args.t, args.algo), file_prefix=outname)
    return 0


if __name__ == "__main__":
    sys.exit(main())
