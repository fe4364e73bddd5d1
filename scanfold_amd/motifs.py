"""Structure extraction and motif refolds after the Fold stage (/root/reference/ScanFold.py:1582-1779).

Every top-level helix of the Zavg < -2 dot-bracket line becomes a motif: its sequence is refolded with its ScanFold
pairs as a hard constraint (MFE structure, MFE, ensemble diversity of the constrained ensemble) and gets a z-score and
p-value from 100 shuffles — the same engine calls as one scan window, so this is the hot path once more on a handful
of short sequences.  Written: `<name>_motif_<n>.dbn`, `<name>_motif_<n>.ct` (dbn2ct) and one gff3 line per motif.

reference                                                        here
---------------------------------------------------------------  -----------------------------------------------
ScanFold.py:1582-1717   bond-order walk + start / end lists      extract_structures()
ScanFold.py:1719-1776   refold loop, gff3 / dbn writer           refold_motifs()
ScanFoldFunctions.py:922-1016 dbn2ct, :869-920 findpair          dbn2ct()
Reproduced as they are upstream:
 * the start / end lists are paired by position in the list, not by bracket matching;
 * a start at the first nucleotide is lost (upstream looks its coordinate up in the 1-based nucleotide table with a
   0-based index, prints "<index> EXCEPT" and moves on), after which every later start meets the previous end;
 * '<' counts as a start where the nesting depth is 1 and '>' as an end where it is 0;
 * the motif's z-score uses the UNCONSTRAINED energy of the motif and the run's shuffle type (upstream defines
   sub_shuffle = "mono" and never uses it).
Not reproduced: the PostScript plot (RNA.PS_rna_plot_a) and the full-length "global refold" of the input sequence
(ScanFold.py:1510-1549: one O(L^3) fold of the whole transcript — a different kernel from the window scan).
"""
from . import RNA
from . import functions as sff


class ExtractedStructure:
    def __init__(self, structure_count, sequence, structure, i, j):
        self.structure_count = structure_count
        self.sequence = sequence
        self.structure = structure
        self.i = i
        self.j = j


_DEPTH_KEEPERS = ".<>{}"


def extract_structures(structure_line, sequence, verbose=True):
    """`structure_line`: line 3 of a makedbn file INCLUDING its newline (upstream walks `len(line) - 1` characters);
    `sequence`: the record's sequence.  -> [ExtractedStructure] with 0-based, inclusive i / j."""
    structure = list(structure_line)
    n = len(structure) - 1
    order, known = [0] * max(n, 0), [False] * max(n, 0)
    depth = 0
    for m in range(n):
        ch = structure[m]
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        elif ch not in _DEPTH_KEEPERS:
            continue
        if m >= len(sequence):
            raise IndexError("list index out of range")  # upstream: sequence[m]
        order[m], known[m] = depth, True
    starts, ends = [], []
    for j in range(n):
        if not known[j]:
            if verbose:
                print(j, "EXCEPT")
            continue
        ch = structure[j]
        if order[j] == 1 and ch in "(<{":
            if j == 0:  # nuc_dict[0] does not exist upstream
                if verbose:
                    print(j, "EXCEPT")
                continue
            starts.append(j)
        elif order[j] == 0 and ch in ")>}":
            ends.append(j)
    out = []
    for l, s in enumerate(starts):
        e = ends[l]  # IndexError when the brackets are unbalanced, as upstream
        keep = [k for k in range(max(s, 0), min(e, n - 1) + 1) if known[k]]
        out.append(ExtractedStructure(l, "".join(sequence[k] for k in keep), "".join(structure[k] for k in keep), s, e))
    return out


def dbn2ct(dbnfile):
    """<x>.dbn (header, sequence, structure) -> <x>.ct, the CT flavour of ScanFoldFunctions.dbn2ct: header
    "<length-1>\\tSequenceID", one line per '.', '(' or ')' character."""
    with open(dbnfile, "r") as f:
        lines = f.readlines()
    sequence, structure = lines[1].strip(), lines[2].strip()
    if len(sequence) != len(structure):
        raise TypeError("exceptions must derive from BaseException")  # upstream: raise("ERROR structure and sequence ...")
    partner, stack = {}, []
    for k, ch in enumerate(structure):
        if ch == "(":
            stack.append(k)
        elif ch == ")":
            if not stack:
                raise KeyError(k)  # upstream's findpair walks off the table
            o = stack.pop()
            partner[o], partner[k] = k, o
    if stack:
        raise KeyError(len(structure))
    out = [str(len(sequence) - 1) + "\tSequenceID\n"]
    depth = 0
    for k, ch in enumerate(structure):
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if depth < 0:
            continue
        if ch == ".":
            out.append("%d %s %d %d %d %d\n" % (k + 1, sequence[k], k, k + 2, 0, k + 1))
        elif ch in "()":
            out.append("%d %s %d %d %d %d\n" % (k + 1, sequence[k], k, k + 2, partner[k] + 1, k + 1))
    with open(dbnfile.replace(".dbn", ".ct"), "w") as ct:
        ct.write("".join(out))


class EngineFolder:
    """The two engine calls a motif needs, through the RNA facade / ScanFoldFunctions mirror."""

    def __init__(self, temperature=37, algo="rnafold"):
        self.temperature, self.algo = temperature, algo

    def constrained(self, frag, constraint):
        md = RNA.md()
        md.temperature = int(self.temperature)
        fc = RNA.fold_compound(str(frag), md)
        fc.hc_add_from_db(str(constraint))
        fc.pf()
        structure, mfe = fc.mfe()
        return structure, mfe, fc.mean_bp_distance()

    def scramble(self, frag, randomizations, shuffle_type):
        return sff.scramble(frag, randomizations, shuffle_type)

    def energies(self, seqlist):
        return sff.energies(seqlist, self.temperature, self.algo)


def refold_motifs(name, motifs, shuffle_type, gff_path, folder=None, randomizations=100, file_prefix=None):
    """ScanFold.py:1726-1776.  Returns the per-motif records; writes the gff3 and the motif dbn / ct files
    (`file_prefix` defaults to `name`, as upstream, which writes them into the working directory)."""
    folder = folder or EngineFolder()
    prefix = name if file_prefix is None else file_prefix
    records = []
    with open(gff_path, "w") as se:
        for num, es in enumerate(motifs, start=1):
            frag = es.sequence
            mfe_structure, mfe, ed = folder.constrained(frag, es.structure)
            mfe = round(mfe, 2)
            ed = round(ed, 2)
            seqlist = [frag] + list(folder.scramble(frag, randomizations, shuffle_type))
            energy_list = folder.energies(seqlist)
            try:
                zscore = round(sff.zscore_function(energy_list, randomizations), 2)
            except Exception:
                zscore = sff.zscore_function(energy_list, randomizations)
            pvalue = round(sff.pvalue_function(energy_list, randomizations), 2)
            dbn = "%s_motif_%d.dbn" % (prefix, num)
            with open(dbn, "w") as f:
                f.write(">%s_motif_%d_coordinates:%s-%s\n%s\n%s" % (name, num, es.i, es.j, frag, mfe_structure))
            dbn2ct(dbn)
            attributes = "motif_%d;sequence=%s;structure=%s;refoldedMFE=%s;MFE(kcal/mol)=%s;z-score=%s;ED=%s" % (
                num, es.sequence, str(es.structure), str(mfe_structure), str(mfe), str(zscore), str(ed))
            se.write("%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\n" % (str(name), ".", "RNA_sequence_secondary_structure",
                                                               str(int(es.i + 1)), str(int(es.j + 1)), ".", ".", ".",
                                                               attributes))
            records.append(dict(motif=num, i=es.i, j=es.j, sequence=frag, constraint=es.structure,
                                structure=mfe_structure, mfe=mfe, zscore=zscore, pvalue=pvalue, ed=ed))
    return records
