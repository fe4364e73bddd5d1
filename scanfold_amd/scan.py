"""ScanFold-Scan on the HIP engine: same command line, same output file, same TSV bytes.

Restates the driver half of /root/reference/ScanFold-Scan.py:
  flags and defaults                     :31-62   (-i -s -w -r -t -type -p --print_random -c)
  output file name                       :67
  record loop, length filter, header     :285-350
  window loop                            :355-358,449   i = 0; while i == 0 or i <= L - W
  T->U transcription of the window       :371
  all-N shortcut (literal 120 x 'N')     :374-380
  rounding and row formatting            :386,389,426-433,442
The per-window arithmetic (:382-423) is one `sf_scan` call per record for ALL windows: native MFE +
structure, partition function -> centroid / ensemble diversity, r shuffles, r+1 MFE folds.

Deliberate differences (documented in DESIGN.md): shuffles come from the device generator
(`--shuffle-backend python` restores the reference's `random`-module shuffles), constraints (-c) and
temperatures other than the parameter set's are rejected instead of silently half-applied
(SURVEY.md F8), and extra flags --seed/--params/--shuffle-backend/-o exist.
"""
import argparse
import sys

import numpy as np

from . import _lib
from . import scan_functions as sfn

ALL_N_120 = "N" * 120
DOTS_120 = "." * 120


def read_fasta(path):
    """(name, sequence) per record; name = first word of the header, as Bio.SeqIO's record.name."""
    records = []
    name, chunks = None, []
    with open(path, "r") as f:
        for line in f:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if name is not None:
                    records.append((name, "".join(chunks)))
                hdr = line[1:].split()
                name = hdr[0] if hdr else ""
                chunks = []
            elif name is not None:
                chunks.append("".join(line.split()))
    if name is not None:
        records.append((name, "".join(chunks)))
    return records


def transcribe(seq):
    return seq.replace("T", "U").replace("t", "u")


def window_starts(length, window_size, step_size):
    """0-based starts exactly as the reference's loop produces them (ScanFold-Scan.py:355-356,449)."""
    out = []
    i = 0
    while i == 0 or i <= (length - window_size):
        out.append(i)
        i += step_size
    return out


def dcal_to_float(dcal):
    """(float)energy/100. as a Python float, for every element of an int array."""
    return (np.asarray(dcal, dtype=np.float32) / np.float32(100.0)).astype(np.float64)


def header_line(read_name):
    return ("i\tj\tTemperature\tNative_dG\tZ-score\tP-score\tEnsembleDiversity\tSequence\tStructure\tCentroid\t"
            + read_name + "\n")


def format_row(start_nucleotide, end_nucleotide, temperature, MFE, zscore, pscore, ED, frag, structure, centroid):
    return (str(start_nucleotide) + "\t" + str(end_nucleotide) + "\t" + str(temperature) + "\t" + str(MFE) + "\t"
            + str(zscore) + "\t" + str(pscore) + "\t" + str(ED) + "\t" + str(frag) + "\t" + str(structure) + "\t"
            + str(centroid) + "\n")


def rows_from_results(seq, starts, W, r, temperature, energies_dcal, structures, centroids, ens_div):
    """TSV rows for the given windows from raw engine output; rounding exactly as the reference does it."""
    E = dcal_to_float(energies_dcal)  # (n, r+1) python-float values of the C floats
    z, sd0 = sfn.zscores_rows(E, r)
    p = sfn.pscores_rows(E)
    rows = []
    for k, i in enumerate(starts):
        frag = transcribe(seq[i:i + W])
        if frag == ALL_N_120:
            rows.append(format_row(i + 1, i + W, temperature, int(0.0), "#DIV/0", int(0.0), int(0.0), frag,
                                   DOTS_120, DOTS_120))
            continue
        MFE = round(float(E[k, 0]), 2)
        if sd0[k]:
            zscore = "#DIV/0!"
        else:
            zscore = round(np.float64(z[k]), 2)  # np.float64.__round__, as upstream's np.mean-derived value
        pscore = round(float(p[k]), 2)
        ED = round(float(ens_div[k]), 2)
        rows.append(format_row(i + 1, i + W, temperature, MFE, zscore, pscore, ED, frag, structures[k], centroids[k]))
    return rows


def scan_record(seq, W, step, r, shuffle_type, temperature=37, engine=None, seed=0, shuffle_backend="device",
                print_random=False):
    """All windows of one record -> list of TSV row strings."""
    eng = engine if engine is not None else _lib.get_engine()
    if float(int(temperature)) != eng.params.temperature:
        raise NotImplementedError("folding temperature %s C: parameter set valid at %s C only"
                                  % (temperature, eng.params.temperature))
    if shuffle_type not in ("di", "mono"):
        # upstream prints a message and goes on with zero shuffles (NaN z-scores); refuse instead
        raise ValueError('Shuffle type not properly designated; please input "di" or "mono"')
    starts = window_starts(len(seq), W, step)
    n_win = len(starts)
    kind = _lib.SHUFFLE_DI if shuffle_type == "di" else _lib.SHUFFLE_MONO
    if shuffle_backend == "device":
        res = eng.scan(seq, W, step, 0, n_win, r, kind, seed)
        energies_dcal = res["energies"]
    elif shuffle_backend == "python":
        # the reference's own generators on the host (random module), folds still batched on the device
        res = eng.scan(seq, W, step, 0, n_win, 0, kind, seed)
        rows_seq = []
        for i in starts:
            frag = transcribe(seq[i:i + W])
            rows_seq.append(frag)
            rows_seq.extend(sfn.scramble(frag, r, shuffle_type))
        energies_dcal = eng.mfe_batch(rows_seq).reshape(n_win, r + 1)
    else:
        raise ValueError("shuffle_backend must be 'device' or 'python'")
    if print_random:
        for k in range(n_win):
            print([float(v) for v in dcal_to_float(energies_dcal[k])])
    return rows_from_results(seq, starts, W, r, temperature, energies_dcal, res["structure"], res["centroid"],
                             res["ens_div"])


def build_parser():
    parser = argparse.ArgumentParser(description="ScanFold-Scan on the MI355X HIP engine")
    parser.add_argument('-i', '--filename', type=str, help='input filename')
    parser.add_argument('-s', type=int, default=10, help='step size')
    parser.add_argument('-w', type=int, default=120, help='window size')
    parser.add_argument('-r', type=int, default=50, help='randomizations')
    parser.add_argument('-t', type=int, default=37, help='Folding temperature')
    parser.add_argument('-type', type=str, default='mono', help='randomization type')
    parser.add_argument('-p', '--print_to_screen', action='store_true', help='print to screen option (default off)')
    parser.add_argument('--print_random', type=str, default='off', help='print to screen option (default off)')
    parser.add_argument('-c', '--constraints', type=str, help='optional | input constraint file')
    # additions
    parser.add_argument('--span', type=int, default=0, help='maximum base-pair span (ScanFold.py --span); 0 = none')
    parser.add_argument('--seed', type=int, default=0, help='seed of the device shuffle generator')
    parser.add_argument('--shuffle-backend', choices=("device", "python"), default="device")
    parser.add_argument('--params', type=str, default=None, help='ViennaRNA .par (v2.0) file to use')
    parser.add_argument('-o', '--output', type=str, default=None, help='output path (default: upstream naming)')
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    if not args.filename:
        raise SystemExit("-i/--filename is required")
    if args.constraints is not None:
        raise NotImplementedError("hard constraints (-c) are not supported by the HIP engine yet")
    window_size, step_size, randomizations = int(args.w), int(args.s), int(args.r)
    temperature, shuffle_type = int(args.t), str(args.type)
    out_path = args.output or (args.filename + ".forward.win_" + str(window_size) + ".stp_" + str(step_size)
                               + ".rnd_" + str(randomizations) + ".shfl_" + str(shuffle_type) + ".txt")
    eng = _lib.get_engine()
    eng.set_max_bp_span(args.span)
    if args.params:
        from . import params as _params
        eng.load_params(_params.load_par(args.params))
    if args.shuffle_backend == "python":
        import random
        random.seed(args.seed)
    with open(out_path, 'w') as w:
        for read_name, seq in read_fasta(args.filename):
            print("Scanning sequence " + str(read_name) + "\nSequence Length: " + str(len(seq)) + "nt long.")
            if len(seq) < window_size:
                continue
            w.write(header_line(read_name))
            rows = scan_record(seq, window_size, step_size, randomizations, shuffle_type, temperature, eng,
                               seed=args.seed, shuffle_backend=args.shuffle_backend,
                               print_random=(args.print_random == "on"))
            for row in rows:
                if args.print_to_screen:
                    f = row.rstrip("\n").split("\t")
                    print("\t".join(f[:7]) + "\n" + f[7] + "\n" + f[8] + "\n" + f[9] + "\n")
                w.write(row)
    return 0


if __name__ == "__main__":
    sys.exit(main())
