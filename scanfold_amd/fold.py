"""ScanFold-Fold: condense the scanning-window results into one consensus structure.

Restates /root/reference/ScanFold-Fold.py (the same procedure runs inline in ScanFold.py:564-677,1036-1453):
  Scan TSV parsing incl. the skipped first data row         :470-476,503-533
  per-window pair tabulation (unpaired nt = a pair with itself) :583-682
  best partner of every nucleotide by SumZ / #TotalWindows   :704-846      -> <in>.ScanFold.log.txt
  competition between partners, final pairs                   :848-1018    -> <in>.ScanFold.final_partners.txt
  CT files for the z-score filters, IGV .bp track             :293-333,385-451,1040-1053
The reference does this with per-nucleotide Python dictionaries of NucPair objects and rescans the whole dictionary
for every competitor look-up (quadratic in the transcript length: the 30 kb benchmark transcript takes hours).  Here
the windows' dot-bracket strings are parsed as one (windows x W) byte matrix, the (nucleotide, partner) groups come out
of one sort, their sums from numpy's own pairwise order applied to all groups of one length at once (group_sums) — so the
floating-point sums, means and the SumZ/#TotalWindows quotient are the reference's own numbers — and only the competition step walks
nucleotides one at a time, over precomputed neighbour lists.  Output files are byte-identical to the reference's
(tests/golden/fold_*: produced by running the reference script, see tests/golden/make_golden_fold.py).

`-c 0` (competition allowed, :1022-1038) writes the DP files of every nucleotide's best partner and best_bps_test.bp.
"""
import argparse
import os
import sys

import numpy as np


def transcribe(seq):
    return seq.replace("T", "U")


class ScanTable:
    """The rows of a Scan TSV the Fold stage uses: i, MFE, z-score, ED, sequence, structure — WITHOUT the first data
    row, which the reference drops (`f.readline()` for the header, then `f.readlines()[1:]`, ScanFold-Fold.py:471-475)."""

    def __init__(self, ident, starts, mfe, z, ed, seqs, structs):
        self.id = ident
        self.starts = np.asarray(starts, dtype=np.int64)
        self.mfe = np.asarray(mfe, dtype=np.float64)
        self.z = np.asarray(z, dtype=np.float64)
        self.ed = np.asarray(ed, dtype=np.float64)
        self.seqs = seqs
        self.structs = structs

    @classmethod
    def from_file(cls, path, ident=None):
        with open(path, "r") as f:
            top_row = f.readline().split("\t")
            lines = f.readlines()[1:]
        if ident is None:
            ident = str(top_row[-1]).strip()
        return cls.from_rows(lines, ident)

    @classmethod
    def from_rows(cls, lines, ident):
        """`lines`: data rows (tab separated, with or without the trailing newline) — already without the dropped one."""
        starts, mfe, z, ed, seqs, structs = [], [], [], [], [], []
        for row in lines:
            if not row.strip():
                continue
            data = row.rstrip("\n").split("\t")
            starts.append(int(data[0]))
            mfe.append(float(data[3]))
            z.append(float(data[4]))  # a "#DIV/0!" row makes the reference crash as well
            ed.append(float(data[6]))
            if "A" in data[8]:  # the reference's test `("A" or ...) in data[8]`: a ScanFold.py-style row with an extra column
                seqs.append(transcribe(data[8]))
                structs.append(data[9])
            else:
                seqs.append(transcribe(data[7]))
                structs.append(data[8])
        return cls(ident, starts, mfe, z, ed, seqs, structs)


def pair_partners(struct_matrix):
    """Partner (0-based position inside the window, -1 = unpaired) of every position of every window.
    struct_matrix: uint8 (n, W) of '.', '(' and ')'.  The reference numbers the nesting level of every bracket and pairs
    up, level by level, the positions of one level in ascending order (ScanFold-Fold.py:607-647); one sort does that for
    all windows at once."""
    n, W = struct_matrix.shape
    opn = struct_matrix == ord("(")
    cls = struct_matrix == ord(")")
    level = np.cumsum(opn, axis=1) - np.cumsum(cls, axis=1) + cls  # '(' and its ')' get the same level >= 1
    partner = np.full((n, W), -1, dtype=np.int64)
    rows, cols = np.nonzero(opn | cls)
    if len(rows) == 0:
        return partner
    lv = level[rows, cols]
    order = np.lexsort((cols, lv, rows))  # by window, level, position
    r, c = rows[order], cols[order]
    if len(r) % 2:
        raise ValueError("unbalanced structure string in the scan table")
    a, b = c[0::2], c[1::2]
    ra = r[0::2]
    if not np.array_equal(ra, r[1::2]) or not (opn[ra, a].all() and cls[ra, b].all()) or (level < 0).any():
        raise ValueError("unbalanced structure string in the scan table")
    partner[ra, a] = b
    partner[ra, b] = a
    return partner


class Tabulation:
    """Per nucleotide k and partner j (j == k: unpaired): the windows that support the pair, in window order.
    This class does the grouping on the host (numpy); DeviceTabulation hands it to the engine."""

    def __init__(self, table):
        n = len(table.starts)
        W = len(table.structs[0]) if n else 0
        if any(len(s) != W for s in table.structs) or any(len(s) != W for s in table.seqs):
            raise ValueError("windows of different lengths in one scan table")
        self._S = np.frombuffer("".join(table.structs).encode("ascii"), dtype=np.uint8).reshape(n, W)
        Q = np.frombuffer("".join(table.seqs).encode("ascii"), dtype=np.uint8).reshape(n, W)
        pos = np.arange(W)[None, :]
        k = (table.starts[:, None] + pos).ravel()                      # coordinate of the nucleotide
        # nucleotide of every coordinate: the LAST window that covers it wins, as in NucleotideDictionary (:108-159)
        self.start_coordinate = int(table.starts[0])
        self.end_coordinate = int(table.starts[-1] + W - 1)
        lo = int(k.min())
        self.lo = lo
        size = int(k.max()) - lo + 1
        nuc = np.zeros(size, dtype=np.uint8)
        nuc[k - lo] = Q.ravel()  # later windows overwrite earlier ones
        self.nuc = nuc
        self.present = np.zeros(size, dtype=bool)
        self.present[k - lo] = True
        self.table = table
        self.window_z = table.z
        self.id = table.id
        self.W = W
        self._k = k

    def groups(self):
        """-> gk, gj, gcount, gfirst, gsum_z, gsum_mfe, gsum_ed: one entry per (nucleotide, partner) group, ordered by
        nucleotide (the order inside one nucleotide is not relied upon); gfirst = first supporting window."""
        table, W = self.table, self.W
        n = len(table.starts)
        part = pair_partners(self._S)
        pos = np.arange(W)[None, :]
        k = self._k
        j = np.where(part >= 0, table.starts[:, None] + part, table.starts[:, None] + pos).ravel()
        win = np.repeat(np.arange(n), W)
        # NucleotideDictionary keys in insertion order: first coordinate = start of the first window, last = end of the last
        order = np.lexsort((win, j, k))  # groups (k, j) contiguous, windows ascending inside a group
        k, j, win = k[order], j[order], win[order]
        new_group = np.ones(len(k), dtype=bool)
        new_group[1:] = (k[1:] != k[:-1]) | (j[1:] != j[:-1])
        gstart = np.nonzero(new_group)[0]
        gcount = np.diff(np.append(gstart, len(k)))
        return (k[gstart], j[gstart], gcount, win[gstart],     # first window of the group = dictionary insertion order
                group_sums(table.z[win], gstart, gcount),      # np.sum of the group's values, window order
                group_sums(table.mfe[win], gstart, gcount), group_sums(table.ed[win], gstart, gcount))


class DeviceTabulation(Tabulation):
    """The same table with the grouping and the sums done by the HIP engine (sf_tabulate_pairs): bracket matching per
    window, one wave per nucleotide, numpy's summation order restated on the device — the arrays come back bit-equal
    to Tabulation.groups().  Needs window starts in ascending order (every scan table has them)."""

    def __init__(self, table, engine):
        super().__init__(table)
        self.engine = engine

    def groups(self):
        t = self.table
        g = self.engine.tabulate_pairs(self._S, t.starts, t.z, t.mfe, t.ed)
        return (g["k"].astype(np.int64), g["j"].astype(np.int64), g["windows"].astype(np.int64),
                g["first_window"].astype(np.int64), g["sum_z"], g["sum_mfe"], g["sum_ed"])


def _pairwise(M):
    """numpy's DOUBLE_pairwise_sum over the rows of M (g, n), column by column — same additions in the same order."""
    g, n = M.shape
    if n < 8:
        res = np.zeros(g)
        for c in range(n):
            res = res + M[:, c]
        return res
    if n <= 128:
        r = [M[:, c].copy() for c in range(8)]
        i = 8
        while i < n - (n % 8):
            for c in range(8):
                r[c] = r[c] + M[:, i + c]
            i += 8
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        while i < n:
            res = res + M[:, i]
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return _pairwise(M[:, :n2]) + _pairwise(M[:, n2:])


def group_sums(values, gstart, gcount):
    """np.sum(list_of_the_group) for every group of consecutive `values` — bit for bit: numpy sums a 1-D array with its
    pairwise scheme (np.add.reduceat associates differently, e.g. 3 x -0.1 -> -0.3 instead of -0.30000000000000004, which
    moves a rounded log entry from -0.01 to -0.0).  Groups of one length are summed together.  Checked against
    np.sum(list) for lengths 2..300 in tests/test_fold.py."""
    out = np.empty(len(gstart), dtype=np.float64)
    for n in np.unique(gcount).tolist():
        sel = np.nonzero(gcount == n)[0]
        M = values[gstart[sel][:, None] + np.arange(n)[None, :]]
        out[sel] = _pairwise(M)
    return out


def _round2(a):
    """str(round(x, 2)) for np.float64 values (np.float64.__round__ == np.round elementwise)."""
    return [str(v) for v in np.round(np.asarray(a, dtype=np.float64), 2).tolist()]


class FoldResult:
    pass


def best_partners(tab, log_path=None):
    """ScanFold-Fold.py:704-846.  Returns a FoldResult with, per nucleotide coordinate (ascending):
    coord, best partner, mean z / mfe / ed of that pair, SumZ/#TotalWindows of it."""
    gk, gj, gcount, gfirst, gsum_z, gsum_mfe, gsum_ed = tab.groups()
    gmean_z = gsum_z / gcount
    gmean_mfe = gsum_mfe / gcount
    gmean_ed = gsum_ed / gcount
    # nucleotides
    new_k = np.ones(len(gk), dtype=bool)
    new_k[1:] = gk[1:] != gk[:-1]
    kstart = np.nonzero(new_k)[0]
    coords = gk[kstart]
    total_windows = np.add.reduceat(gcount, kstart)
    kidx = np.cumsum(new_k) - 1                            # nucleotide index of every group
    gtw = gsum_z / total_windows[kidx]                     # SumZ / #TotalWindows
    # best partner: smallest quotient, ties -> the partner seen first (min() over a dict in insertion order)
    order = np.lexsort((gfirst, gtw, kidx))
    first_of_k = np.ones(len(order), dtype=bool)
    first_of_k[1:] = kidx[order][1:] != kidx[order][:-1]
    best = order[first_of_k]
    res = FoldResult()
    res.coords = coords
    res.best_j = gj[best]
    res.best_mean_z, res.best_mean_mfe, res.best_mean_ed = gmean_z[best], gmean_mfe[best], gmean_ed[best]
    res.best_tw_z = gtw[best]
    res.nuc = tab.nuc[coords - tab.lo]
    res.total_windows = total_windows
    if log_path is not None:
        # groups of one nucleotide in dictionary order (first window ascending)
        gorder = np.lexsort((gfirst, kidx))
        s_cnt = [str(int(v)) for v in gcount[gorder]]
        s_mfe, s_mz, s_med = _round2(gmean_mfe[gorder]), _round2(gmean_z[gorder]), _round2(gmean_ed[gorder])
        s_sum, s_tw = _round2(gsum_z[gorder]), _round2(gtw[gorder])
        gk_o, gj_o = gk[gorder], gj[gorder]
        gnuc = [chr(c) for c in tab.nuc[gj_o - tab.lo]]
        num_bp = np.add.reduceat((gk != gj).astype(np.int64), kstart)
        bounds = np.append(kstart, len(gk))
        out = []
        for t in range(len(coords)):
            kk = int(coords[t])
            out.append("\ni-nuc\tBP(j)\tNuc\t#BP_Win\tavgMFE\tavgZ\tavgED\tSumZ\tSumZ/#TotalWindows\tBPs= %d\n" % num_bp[t])
            out.append("nt-%d\t-\t%s\t%d\t-\t-\t-\t-\t-\n" % (kk, chr(res.nuc[t]), total_windows[t]))
            for g in range(bounds[t], bounds[t + 1]):
                jj = int(gj_o[g])
                out.append("%d\t%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\n" % (kk, "NoBP" if jj == kk else str(jj), gnuc[g], s_cnt[g],
                                                                   s_mfe[g], s_mz[g], s_med[g], s_sum[g], s_tw[g]))
        with open(log_path, "w") as f:
            f.write("".join(out))
    return res


def compete(tab, res, log_path=None):
    """ScanFold-Fold.py:848-1018 with competition == 1.  -> final partner per nucleotide:
    arrays fin_i, fin_j (the pair written for coordinate k; i == j: unpaired), fin_z, plus the i-nucleotide / j-nucleotide."""
    coords = res.coords
    n = len(coords)
    idx_of = {int(c): t for t, c in enumerate(coords)}
    start, end = tab.start_coordinate, tab.end_coordinate
    # subdict = best_total_window_mean_bps restricted to range(start, end): the LAST coordinate is left out (:879,900)
    in_sub = np.array([(start <= int(c) < end) for c in coords])
    bj = res.best_j
    # for a coordinate c: the entries of subdict (ascending key) whose i or j is c
    by_partner = {}
    for t in np.nonzero(in_sub)[0].tolist():
        by_partner.setdefault(int(bj[t]), []).append(t)

    def competing(c):
        lst = list(by_partner.get(c, ()))
        t = idx_of.get(c)
        if t is not None and in_sub[t] and int(bj[t]) != c:  # the entry of c itself (if its partner is c it is in the list already)
            lst.append(t)
            lst.sort()
        return lst

    twz = res.best_tw_z
    fin_i = np.empty(n, dtype=np.int64)
    fin_j = np.empty(n, dtype=np.int64)
    fin_z = np.empty(n, dtype=np.float64)
    lines = [COMPETE_HEADER]
    r_mfe, r_z, r_ed = _round2(res.best_mean_mfe), _round2(res.best_mean_z), _round2(res.best_mean_ed)
    r_tw = _round2(twz)
    cache = {}
    coords_l, bj_l, twz_l = coords.tolist(), bj.tolist(), twz.tolist()
    for t in range(n):
        kk = coords_l[t]
        vj = bj_l[t]
        merged = []
        for c0 in (kk, vj):
            for p in competing(c0):
                for c1 in (bj_l[p], coords_l[p]):
                    lst = cache.get(c1)
                    if lst is None:
                        lst = cache[c1] = competing(c1)
                    merged.extend(lst)
        if merged:
            b = min(merged, key=twz_l.__getitem__)  # the first of the smallest (min() over sums of one-element lists, :163-281)
            bi, bjj = coords_l[b], bj_l[b]
            bz_mfe, bz_z, bz_ed = r_mfe[b], r_tw[b], r_ed[b]
        else:
            b = None
            bi = bjj = kk
        if b is not None and kk != bi and kk != bjj:
            lines.append("nt-%d*:\t%d\t%d\t%s\t%s\t%s\n" % (kk, bi, bjj, bz_mfe, bz_z, bz_ed))
            fin_i[t] = fin_j[t] = kk
            fin_z[t] = res.best_mean_z[idx_of[bi]]
        else:
            lines.append("nt-%d:\t%d\t%d\t%s\t%s\t%s\n" % (kk, bi, bjj, r_mfe[t], r_z[t], r_ed[t]))
            fin_i[t], fin_j[t] = bi, bjj
            fin_z[t] = res.best_mean_z[idx_of[bi]]
    if log_path is not None:
        with open(log_path, "w") as f:
            f.write("".join(lines))
    res.fin_i, res.fin_j, res.fin_z = fin_i, fin_j, fin_z
    return res


def write_ct(tab, res, path, filt, header_name=None):
    """write_ct (ScanFold-Fold.py:293-333), forward strand.  One line per nucleotide: index, base, index-1, index+1,
    partner (0 when unpaired or when the pair's mean z-score is not below the filter), index."""
    coords = res.coords
    off = tab.start_coordinate - 1
    key = coords - off
    i_rel, j_rel = res.fin_i - off, res.fin_j - off
    paired = (res.fin_z < filt) & (res.fin_i != res.fin_j)
    partner = np.where(paired, np.where(key == i_rel, j_rel, i_rel), 0)
    # the base printed is the i-nucleotide when the key is i, else the j-nucleotide: both are the key's own nucleotide
    nuc = [chr(c) for c in tab.nuc[coords - tab.lo]]
    out = ["%d\t%s\n" % (len(coords), header_name if header_name is not None else path)]
    out += ["%d %s %d %d %d %d\n" % (kk, b, kk - 1, kk + 1, p, kk)
            for kk, b, p in zip(key.tolist(), nuc, partner.tolist())]
    with open(path, "w") as f:
        f.write("".join(out))


def write_dp(res, path, filt, minz):
    """write_dp (ScanFold-Fold.py:380-394; ScanFoldFunctions.py:564-578): IGV "dot plot" lines of the BEST partner of every
    nucleotide whose mean z-score is below the filter — i, j, (-1/minz * z) / minz — the competition-allowed mode's output
    (the three branches of the reference print the same three numbers: an unpaired nucleotide has j == k)."""
    out = ["%d\t%d\t%f\n" % (k, j, float((-1 / minz) * z) / minz)
           for k, j, z in zip(res.coords.tolist(), res.best_j.tolist(), res.best_mean_z.tolist()) if z < filt]
    with open(path, "w") as f:
        f.write("".join(out))


def write_bp(tab, res, path, ident, best=False):
    """write_bp (ScanFold-Fold.py:399-451): IGV arc track, one line per nucleotide, colour class by mean z-score.
    best: the track of the best partners before competition (`best_bps`, written by -c 0) instead of the final ones."""
    minz = min(tab.window_z.tolist())
    if best:
        fin_i, fin_j, fin_z = res.coords, res.best_j, res.best_mean_z
    else:
        fin_i, fin_j, fin_z = res.fin_i, res.fin_j, res.fin_z
    out = ["color:\t55\t129\t255\tLess than -2 %s\n" % str(minz), "color:\t89\t222\t111\t-1 to -2\n",
           "color:\t236\t236\t136\t0 to -1\n", "color:\t199\t199\t199\t0\n", "color:\t228\t228\t228\t0 to 1\n",
           "color:\t243\t243\t243\t1 to 2\n", "color:\t247\t247\t247\tGreater than 2\n"]
    z = fin_z
    score = np.full(len(z), 6)
    score[z <= 2] = 5
    score[z <= 1] = 4
    score[z == 0] = 3
    score[z < 0] = 2
    score[z < -1] = 1
    score[z < -2] = 0
    for t in range(len(z)):
        a, b = int(fin_i[t]), int(fin_j[t])
        if a == b:
            a = int(res.coords[t])
        out.append("%s\t%d\t%d\t%d\t%d\t%d\n" % (ident, a, a, b, b, score[t]))
    with open(path, "w") as f:
        f.write("".join(out))


COMPETE_HEADER = ("\ni\tbp(i)\tbp(j)\tavgMFE\tavgZ\tavgED\t*Indicates most favorable bp has more favorable partner or is "
                  "more likely to be unpaired (competing coordinates are reported)\n")


def fold(table, prefix, filt=-2, bp_path=None, write_log=True, engine=None, competition=1):
    """The whole stage for one scan table; files are named like the reference's (`prefix` = "<input>.ScanFold.").
    engine: a scanfold_amd Engine -> the pair tabulation runs on the GPU (DeviceTabulation); None -> numpy.
    competition: 1 (default) = competing partners resolved, CT files + final_partners_test.bp (ScanFold-Fold.py:860-1074);
    0 = competition allowed: DP files of every nucleotide's best partner + best_bps_test.bp (:1022-1038)."""
    tab = Tabulation(table) if engine is None else DeviceTabulation(table, engine)
    res = best_partners(tab, prefix + "log.txt" if write_log else None)
    z = tab.window_z
    meanz, stdz = float(np.mean(z)), float(np.std(z))
    one, two = float(meanz - stdz), float(meanz - 2 * stdz)
    if competition == 0:
        with open(prefix + "final_partners.txt", "w") as f:  # the reference opens the file and writes its header only
            f.write(COMPETE_HEADER)
        minz = min(z.tolist())
        for name, f in ((str(filt), filt), ("no_filter", 10.0), ("-1", -1.0), ("-2", -2.0),
                        ("mean_" + str(round(meanz, 2)), meanz), ("below_mean_" + str(round(one, 2)), one)):
            write_dp(res, prefix + name + ".dp", float(f), minz)
        write_bp(tab, res, bp_path or "best_bps_test.bp", tab.id, best=True)
        return tab, res
    bp_path = bp_path or "final_partners_test.bp"
    compete(tab, res, prefix + "final_partners.txt")
    base = os.path.basename(prefix)
    for name, f in ((str(filt), filt), ("no_filter", 10.0), ("-1", -1.0), ("-2", -2.0),
                    ("below_mean_" + str(round(meanz, 2)), meanz), ("1sd_below_mean_" + str(round(one, 2)), one),
                    ("2sd_below_mean_" + str(round(two, 2)), two)):
        write_ct(tab, res, prefix + name + ".ct", float(f), header_name=prefix + name + ".ct")
    write_bp(tab, res, bp_path, tab.id)
    return tab, res


def structure_string(tab, res, filt):
    """Dot-bracket of the final pairs below the filter (what `ct2dot` makes of the CT file)."""
    coords = res.coords
    db = ["."] * len(coords)
    pos = {int(c): t for t, c in enumerate(coords)}
    for t in range(len(coords)):
        a, b = int(res.fin_i[t]), int(res.fin_j[t])
        if a != b and res.fin_z[t] < filt and a in pos and b in pos:
            lo, hi = (a, b) if a < b else (b, a)
            db[pos[lo]], db[pos[hi]] = "(", ")"
    return "".join(db)


def main(argv=None):
    parser = argparse.ArgumentParser(description="ScanFold-Fold on vectorised tables")
    parser.add_argument('-i', '--input', type=str, required=True, help='input filename')
    parser.add_argument('-f', type=int, default=-2, help='filter value')
    parser.add_argument('-c', type=int, default=1, help='Competition (1 for disallow competition, 0 for allow; 1 by default)')
    parser.add_argument('-id', type=str, help='Accession number or ID of input sequence; creates properly named BP files.')
    args = parser.parse_args(argv)
    if args.c not in (0, 1):  # upstream runs neither branch and writes the two logs only
        raise ValueError("-c must be 1 (no competition allowed) or 0 (competition allowed)")
    table = ScanTable.from_file(args.input, args.id)
    print("Sequence length: " + str(len(set((table.starts[:, None] + np.arange(len(table.structs[0]))).ravel().tolist()))) + "nt")
    fold(table, str(args.input) + ".ScanFold.", filt=int(args.f), competition=int(args.c))
    print("ScanFold-Fold complete, find results in...")
    return 0


if __name__ == "__main__":
    sys.exit(main())
