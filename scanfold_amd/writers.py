"""File writers downstream of the scan table and the Fold stage.

reference (ScanFoldFunctions.py)            here
------------------------------------------  ---------------------------------------------------------------
write_wig(metric_list, step, name, path)    :626-642   same signature; IGV fixedStep track of a per-window metric
write_wig_dict(nuc_dict, path, name, step)  :616-624   write_wig_dict(zscores, path, name, step): per-nucleotide mean z-score
                                                       of the final partners
write_fasta(nuc_dict, path, name)           :597-605   write_fasta(sequence, path, name)
write_fai(nuc_dict, path, name)             :717-725   write_fai(sequence_length, path, name)
makedbn(ctfile, name)                       :67-138    same signature: <ctfile>.ct -> <ctfile>.dbn; a pair that
                                                       crosses an earlier one is written '<' '>' as upstream does
The per-nucleotide dictionaries of NucZscore objects the reference passes around are plain strings / lengths here.
Outputs are pinned byte for byte by tests/golden/writers.json (the reference's functions run on the same inputs by
tests/golden/make_golden_writers.py).  The CT / .bp / log writers of the Fold stage live in scanfold_amd/fold.py.
Motif extraction, the motif refolds and their gff3 / dbn2ct files: scanfold_amd/motifs.py.
write_dp (competition-allowed mode) lives in scanfold_amd/fold.py.  Not reproduced: the varna / PS plots of ScanFold.py:1484-1779.
"""


def write_wig(metric_list, step, name, outputfilename):
    out = ["%s %s %s %s %s\n" % ("fixedStep", "chrom=" + name, "start=1", "step=" + str(step), "span=" + str(step))]
    for metric in metric_list:
        if isinstance(metric, str):  # "#DIV/0!" and anything else that is not a number goes out as text
            out.append("%s\n" % metric)
        else:
            out.append("%f\n" % metric)
    with open(outputfilename, "w") as w:
        w.write("".join(out))


def write_wig_dict(zscores, outputfilename, name, step_size):
    """One %f line per nucleotide (the reference walks its final-partner dictionary in coordinate order)."""
    out = ["%s %s %s %s %s\n" % ("fixedStep", "chrom=" + name, "start=1", "step=" + str(step_size), "span=" + str(step_size))]
    out += ["%f\n" % z for z in zscores]
    with open(outputfilename, "w") as w:
        w.write("".join(out))


def write_fasta(sequence, outputfilename, name):
    with open(outputfilename, "w") as w:
        w.write(">" + name + "\n")
        w.write(str(sequence) + "\n")


def write_fai(sequence_length, filename, name):
    name = str(name)
    n = int(sequence_length)
    offset = len((">" + name + "\n").encode("utf-8"))
    with open(filename, "w") as w:
        w.write("%s\t%s\t%s\t%s\t%s\n" % (name, n, offset, n, n + 1))


def makedbn(ctfile, name):
    """<ctfile>.ct -> <ctfile>.dbn.  For a pair (i, j), i < j: '(' at i unless a line between i and j pairs with a
    position before i (a crossing pair), then '<'; the mirrored test for j gives ')' or '>'."""
    with open(ctfile + ".ct", "r") as f:
        data = f.readlines()
    rows = [ln.split() for ln in data]
    body = rows[1:]
    partner_of_line = [int(r[-2]) if r else 0 for r in rows]  # by file line (line 0 = header)
    first = [int(r[0]) if r else 0 for r in rows]
    sequence, dot = [], []
    for r in body:
        icoord, jcoord = int(r[0]), int(r[-2])
        if len(r[1]) > 1:
            continue
        sequence.append(r[1])
        if jcoord == 0:
            dot.append(".")
        elif icoord < jcoord:
            ch = None
            for ln in range(icoord, len(rows)):   # data[icoord:]: the file lines after nucleotide icoord's own
                kcoord, lcoord = first[ln], partner_of_line[ln]
                if kcoord == jcoord:
                    ch = "("
                    break
                if lcoord != 0 and lcoord < icoord:
                    ch = "<"
                    break
            if ch:
                dot.append(ch)
        elif icoord > jcoord:
            ch = None
            for ln in range(jcoord, len(rows)):
                kcoord, lcoord = first[ln], partner_of_line[ln]
                if kcoord == icoord:
                    ch = ")"
                    break
                if lcoord != 0 and lcoord < jcoord:
                    ch = ">"
                    break
            if ch:
                dot.append(ch)
    with open(ctfile + ".dbn", "w") as dbn:
        dbn.write(">%s\n%s\n%s\n" % (name, "".join(sequence), "".join(dot)))
