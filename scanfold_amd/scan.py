"""ScanFold-Scan on the HIP engine: same command line, same output file name, same TSV layout and rounding.

The numbers in the file are those of the loaded energy-parameter set.  The set shipped with this package is a
RECONSTRUCTION of Turner 2004 (scanfold_amd/params, tools/make_recon_par.py), not ViennaRNA's own table: a run
with it prints a warning, and `--require-published-params` refuses to run without `--params <rna_turner2004.par>`.

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
(`--shuffle-backend python` restores the reference's `random`-module shuffles); extra flags
--seed/--params/--shuffle-backend/--span/--gpus/--constraint-unbalanced/-o exist.  -t and -c behave as upstream
(SURVEY.md F8): they change the native window's fold only; -t needs a parameter set with enthalpy tables.
"""
import argparse
import os
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


def _text_rows(x, W):
    """structures / centroids as a list of W-char str: from a list of str, or a uint8 array (n, >= W)."""
    if isinstance(x, np.ndarray):
        big = np.ascontiguousarray(x[:, :W]).tobytes().decode("ascii")
        return [big[k * W:(k + 1) * W] for k in range(x.shape[0])]
    return x


def rows_from_results(seq, starts, W, r, temperature, energies_dcal, structures, centroids, ens_div,
                      native_dcal=None):
    """TSV rows for the given windows from raw engine output; rounding exactly as the reference does it
    (ScanFold-Scan.py:386,389,426-433,442).  No per-row numpy objects: the columns are rounded and converted to
    text as whole lists (np.round on a vector equals round() on each np.float64, str(float) equals
    str(np.float64) — tests/test_golden_host.py), then joined."""
    E = dcal_to_float(energies_dcal)  # (n, r+1) python-float values of the C floats
    z, sd0 = sfn.zscores_rows(E, r)
    p = sfn.pscores_rows(E)
    tseq = transcribe(seq)
    n = len(starts)
    t = str(temperature)
    # Native_dG column: the energy of the native fold as the reference computed it — with -t / -c that fold differs
    # from energy_list[0], which stays the plain 37 C RNA.fold of the window (ScanFold-Scan.py:244-246,382-398)
    nat = E[:, 0] if native_dcal is None else dcal_to_float(native_dcal)
    mfe_s = [str(round(v, 2)) for v in nat.tolist()]
    z_s = [str(v) for v in np.round(z, 2).tolist()]  # np.float64.__round__, as upstream's np.mean-derived value
    if sd0.any():
        for k in np.nonzero(sd0)[0].tolist():
            z_s[k] = "#DIV/0!"
    p_s = [str(round(v, 2)) for v in p.tolist()]
    ed_s = [str(round(v, 2)) for v in np.asarray(ens_div, dtype=np.float64).tolist()]
    structures = _text_rows(structures, W)
    centroids = _text_rows(centroids, W)
    rows = ["%d\t%d\t%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\n" % (i + 1, i + W, t, mfe_s[k], z_s[k], p_s[k], ed_s[k],
                                                            tseq[i:i + W], structures[k], centroids[k])
            for k, i in enumerate(starts)]
    if ALL_N_120 in tseq:  # the reference's literal 120 x 'N' shortcut (ScanFold-Scan.py:374-380)
        for k, i in enumerate(starts):
            if tseq[i:i + W] == ALL_N_120:
                rows[k] = format_row(i + 1, i + W, temperature, int(0.0), "#DIV/0", int(0.0), int(0.0), ALL_N_120,
                                     DOTS_120, DOTS_120)
    assert len(rows) == n
    return rows


CHUNK_WINDOWS = 4096  # windows per engine call: the host formats chunk k while the GPU computes chunk k+1
# what the last scan_record of this process spent where (seconds): `--timing` prints it per rank
LAST_STATS = {}


def _chunk_bounds(lo, hi, chunk=None):
    """[(first window, count)] of the engine calls for windows [lo, hi).  An explicit size (argument or SCANFOLD_CHUNK_WINDOWS)
    gives equal chunks.  The default schedule is tapered: what the pipeline cannot overlap is the host's work on the LAST chunk,
    and what the GPU loses per call (a partition-function run's first window is a full one, the tail of a launch) shrinks with
    the chunk — so the body goes in a few large calls (about 2 x CHUNK_WINDOWS each) and the end in a small one (CHUNK_WINDOWS / 8):
    cfg3 = 4 x 7 343 + 509 windows instead of 8 x 4 096."""
    n = hi - lo
    explicit = chunk or os.environ.get("SCANFOLD_CHUNK_WINDOWS")
    if explicit:
        c = int(explicit)
        if c < 1:
            raise ValueError("windows per engine call (chunk / SCANFOLD_CHUNK_WINDOWS) must be >= 1, got %d" % c)
        return [(w0, min(c, hi - w0)) for w0 in range(lo, hi, c)]
    if n <= CHUNK_WINDOWS:
        return [(lo, n)] if n > 0 else []
    tail = max(256, CHUNK_WINDOWS // 8)
    body = n - tail
    parts = -(-body // (2 * CHUNK_WINDOWS))
    size = -(-body // parts)
    bounds, w0 = [], lo
    while w0 < lo + body:
        nw = min(size, lo + body - w0)
        bounds.append((w0, nw))
        w0 += nw
    bounds.append((w0, hi - w0))
    return bounds


def _engine_chunks(work, lo, hi, chunk=None, threaded=True):
    """Yield (w0, work(w0, nw)) for consecutive chunks of windows [lo, hi).  `work` makes the engine calls; it runs
    on a helper thread (ctypes drops the GIL for the duration of a library call), so the caller's host work on
    chunk k overlaps chunk k+1 on the GPU.  Only that thread talks to the library while the generator is alive."""
    import queue
    import threading
    bounds = _chunk_bounds(lo, hi, chunk)
    if not threaded or len(bounds) <= 1:
        for w0, nw in bounds:
            yield w0, work(w0, nw)
        return
    q = queue.Queue(maxsize=2)
    stop = threading.Event()

    def put(item):
        # (bounded queue: give up when the consumer has gone away, so that the thread can end)
        while not stop.is_set():
            try:
                q.put(item, timeout=0.1)
                return True
            except queue.Full:
                pass
        return False

    def produce():
        try:
            for w0, nw in bounds:
                if stop.is_set() or not put((w0, work(w0, nw))):
                    return
        except BaseException as e:  # hand the error to the consumer
            put((None, e))
            return
        put((None, None))

    th = threading.Thread(target=produce, daemon=True)
    th.start()
    try:
        while True:
            w0, res = q.get()
            if w0 is None:
                if res is not None:
                    raise res
                return
            yield w0, res
    finally:
        # The consumer is done — normally, or because formatting raised / the generator was closed mid-way.  The library is
        # not re-entrant: nobody may call into it (e.g. to restore the temperature) while the helper thread is still inside
        # work().  Tell it to stop, make room in the queue, and wait for it.
        stop.set()
        while th.is_alive():
            try:
                q.get_nowait()
            except queue.Empty:
                pass
            th.join(timeout=0.05)


def read_constraints(path, seq_len):
    """The -c file as ScanFold-Scan.py:312-333 reads it: its THIRD line is the dot-bracket constraint of the whole
    record; upstream compares len(line incl. newline) - 1 with the sequence length and raises otherwise."""
    with open(path, "r") as f:
        constraints = f.readlines()[2]
    print("Constraint list is " + str(len(constraints) - 1) + "nt long.")
    if len(constraints) - 1 != seq_len:
        raise ValueError("Error detected. Sequence and Constraints must be same length.")
    return constraints[:seq_len]


def _window_rows(text, W, step, w0, nw):
    """uint8 (nw, W): windows w0 .. w0+nw-1 of `text`."""
    arr = np.frombuffer(text.encode("ascii"), dtype=np.uint8)
    view = np.lib.stride_tricks.sliding_window_view(arr, W)[::step]
    return np.ascontiguousarray(view[w0:w0 + nw])


def scan_record(seq, W, step, r, shuffle_type, temperature=37, engine=None, seed=0, shuffle_backend="device",
                print_random=False, chunk=None, constraints=None, unbalanced="error", windows=None):
    """All windows of one record (or the range windows = (lo, hi) of them: a rank's shard) -> list of TSV row strings.

    temperature / constraints follow ScanFold-Scan.py: the native window's MFE, structure, centroid and
    ensemble diversity honour -t and -c (fc = RNA.fold_compound(frag, md); fc.hc_add_from_db(...), :382-418), while
    the r shuffles AND the native energy that enters the z-score come from the global RNA.fold — 37 C, unconstrained
    (:244-246,419-433).  constraints: the record's dot-bracket line (read_constraints), sliced per window as :405-406.
    unbalanced: what to do when a window cuts through a bracket pair — "error" (ViennaRNA aborts there) or "ignore"
    (the unmatched brackets of that window become '.')."""
    eng = engine if engine is not None else _lib.get_engine()
    if shuffle_type not in ("di", "mono"):
        # upstream prints a message and goes on with zero shuffles (NaN z-scores); refuse instead
        raise ValueError('Shuffle type not properly designated; please input "di" or "mono"')
    if shuffle_backend not in ("device", "python"):
        raise ValueError("shuffle_backend must be 'device' or 'python'")
    temperature = int(temperature)
    plain = constraints is None and temperature == 37
    t_before = eng.params.temperature
    if not plain:
        eng.set_temperature(temperature)  # fails early for a parameter set without enthalpies
    starts = window_starts(len(seq), W, step)
    n_win = len(starts)
    win_lo, win_hi = (0, n_win) if windows is None else (max(0, int(windows[0])), min(n_win, int(windows[1])))
    kind = _lib.SHUFFLE_DI if shuffle_type == "di" else _lib.SHUFFLE_MONO
    r_dev = r if shuffle_backend == "device" else 0

    def work(w0, nw):
        if plain:
            eng.set_temperature(37)
            return eng.scan(seq, W, step, w0, nw, r_dev, kind, seed, raw=True)
        eng.set_temperature(37)
        out = dict(energies=eng.scan(seq, W, step, w0, nw, r_dev, kind, seed, _lib.SCAN_NO_PF | _lib.SCAN_NO_TRACE,
                                     raw=True)["energies"])
        eng.set_temperature(temperature)
        if constraints is None:
            nat = eng.scan(seq, W, step, w0, nw, 0, kind, seed, raw=True)
            out.update(native=nat["energies"][:, 0], structure=nat["structure"], centroid=nat["centroid"],
                       ens_div=nat["ens_div"])
        else:
            cons = _window_rows(constraints, W, step, w0, nw)
            if unbalanced == "ignore":
                cons = _drop_unmatched_brackets(cons)
            fc = eng.fold_constrained(_window_rows(transcribe(seq), W, step, w0, nw), cons)
            out.update(native=fc["mfe"], structure=fc["structure"], centroid=fc["centroid"], ens_div=fc["mean_bp_dist"])
        return out

    import time
    rows = []
    t_start = time.perf_counter()
    LAST_STATS.clear()
    LAST_STATS.update(windows=win_hi - win_lo, host_format_s=0.0, wait_engine_s=0.0)
    try:
        _scan_chunks(rows, work, win_lo, win_hi, chunk, shuffle_backend, starts, seq, W, r, shuffle_type, temperature,
                     eng, print_random)
    finally:
        # the engine is shared (functions.energies, the RNA facade): leave it at the model it had (-t / -c switch
        # between 37 C and T inside work())
        eng.set_temperature(t_before)
        LAST_STATS["total_s"] = time.perf_counter() - t_start
    return rows


def _scan_chunks(rows, work, win_lo, win_hi, chunk, shuffle_backend, starts, seq, W, r, shuffle_type, temperature, eng,
                 print_random):
    import time
    t_mark = time.perf_counter()
    for w0, res in _engine_chunks(work, win_lo, win_hi, chunk=chunk, threaded=(shuffle_backend == "device")):
        LAST_STATS["wait_engine_s"] += time.perf_counter() - t_mark
        t_mark = time.perf_counter()
        sub = starts[w0:w0 + len(res["ens_div"])]
        energies_dcal = res["energies"]
        if shuffle_backend == "python":
            # the reference's own generators on the host (random module), folds still batched on the device
            rows_seq = []
            for i in sub:
                frag = transcribe(seq[i:i + W])
                rows_seq.append(frag)
                rows_seq.extend(sfn.scramble(frag, r, shuffle_type))
            eng.set_temperature(37)
            energies_dcal = eng.mfe_batch(rows_seq).reshape(len(sub), r + 1)
        if print_random:
            for k in range(len(sub)):
                print([float(v) for v in dcal_to_float(energies_dcal[k])])
        rows.extend(rows_from_results(seq, sub, W, r, temperature, energies_dcal, res["structure"], res["centroid"],
                                      res["ens_div"], native_dcal=res.get("native")))
        LAST_STATS["host_format_s"] += time.perf_counter() - t_mark
        t_mark = time.perf_counter()


def _drop_unmatched_brackets(cons):
    """uint8 (n, W) constraint rows: brackets without a partner inside their window -> '.' (the `--constraint-unbalanced
    ignore` policy; upstream would abort inside ViennaRNA on such a window)."""
    cons = cons.copy()
    for row in cons:
        stack = []
        for k, ch in enumerate(row):
            if ch == 40:
                stack.append(k)
            elif ch == 41:
                if stack:
                    stack.pop()
                else:
                    row[k] = 46
        for k in stack:
            row[k] = 46
    return cons


def scan_record_sharded(seq, W, step, r, shuffle_type, temperature, eng, seed, rank, world, device=None, **kw):
    """One record over `world` ranks (one process per GPU).  Every rank scans ITS contiguous window range exactly as a
    single process would (scan_record: engine calls in chunks on a helper thread, z/p-scores and row formatting of
    chunk k while the GPU works on chunk k+1) — so the host work shards with the folds — and ONE all-gather of
    fixed-size row slots (scanfold_amd/dist.py: gather_rows) hands every rank all rows in window order; rank 0 writes
    them.  A window's shuffles depend on (seed, absolute window index) only, so the rows do not depend on the sharding.
    device: where the collective's tensors live (the rank's GPU for nccl = RCCL; None = host, gloo)."""
    from . import dist as sdist
    n_win = len(window_starts(len(seq), W, step))
    lo, hi = sdist.shard_range(n_win, rank, world)
    # A failure that only one rank sees (an unbalanced constraint in ITS windows, a row that does not fit its gather slot)
    # must not leave the others waiting in the collective: every rank reports how its shard went, one tiny all-reduce, and
    # either everybody gathers or everybody raises.
    rows, packed, err = None, None, None
    try:
        rows = scan_record(seq, W, step, r, shuffle_type, temperature, eng, seed=seed, windows=(lo, hi), **kw)
        if len(rows) != hi - lo:
            raise ValueError("rank %d formatted %d rows for the range [%d, %d)" % (rank, len(rows), lo, hi))
        # packed ONCE, here (raises if a row is too long for its slot); gather_rows ships this array
        packed = sdist.pack_rows(rows, sdist.shard_size(n_win, world), sdist.row_slot_width(W))
    except Exception as e:  # noqa: BLE001 — re-raised below, on every rank
        err = e
    failed = sdist.any_rank_failed(err is not None, world, device=device)
    if failed:
        if err is not None:
            raise err
        raise RuntimeError("rank %d: another rank failed while scanning its windows of this record (its own message "
                           "is on that rank's stderr); nothing was gathered" % rank)
    return sdist.gather_rows(rows, n_win, rank, world, W, device=device, want=(rank == 0), prepacked=packed)


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
    parser.add_argument('--constraint-unbalanced', choices=("error", "ignore"), default="error",
                        help='a window that cuts through a bracket pair of the -c line: error (ViennaRNA aborts '
                        'there, the default) or ignore (the unmatched brackets of that window become dots)')
    parser.add_argument('--require-published-params', action='store_true',
                        help='refuse to run on the reconstructed default parameter set (needs --params)')
    parser.add_argument('--gpus', type=int, default=1, help='shard the windows of every record over this many GPUs '
                        '(one process per GPU, one RCCL all-gather per record)')
    parser.add_argument('-o', '--output', type=str, default=None, help='output path (default: upstream naming)')
    parser.add_argument('--timing', action='store_true', help='print, per rank and record, where the time went')
    return parser


def _relaunch_sharded(args_list, gpus):
    """`--gpus N` from a plain `python -m scanfold_amd.scan`: start N ranks under torch.distributed.run as a CHILD
    process (never exec: nothing here has touched the GPU yet, and nothing may replace a process that has) and
    return its exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "scanfold_amd.scan"] + list(args_list)
    return subprocess.run(cmd).returncode


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = build_parser().parse_args(argv)
    if not args.filename:
        raise SystemExit("-i/--filename is required")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ:
        return _relaunch_sharded(argv, args.gpus)
    window_size, step_size, randomizations = int(args.w), int(args.s), int(args.r)
    temperature, shuffle_type = int(args.t), str(args.type)
    out_path = args.output or (args.filename + ".forward.win_" + str(window_size) + ".stp_" + str(step_size)
                               + ".rnd_" + str(randomizations) + ".shfl_" + str(shuffle_type) + ".txt")
    from . import params as _params
    if args.require_published_params and not args.params:
        raise SystemExit("--require-published-params: give --params <path to ViennaRNA's rna_turner2004.par>; the "
                         "shipped default (%s) is a reconstruction" % _params.DEFAULT_PAR)
    eng = _lib.get_engine()
    eng.set_max_bp_span(args.span)
    if args.params:
        user_set = _params.load_par(args.params)
        if args.require_published_params and user_set.def_substituted:
            raise SystemExit("--require-published-params: %s has DEF entries in %s; they would be filled from the "
                             "reconstructed default table" % (args.params, ", ".join(sorted(user_set.def_substituted))))
        eng.load_params(user_set)
    if rank == 0:
        _params.warn_if_reconstructed(eng.params)
    if world > 1:
        import torch
        import torch.distributed as dist
        if args.shuffle_backend != "device":
            raise SystemExit("--gpus > 1 needs the device shuffle generator")
        # SCANFOLD_DIST_BACKEND=gloo: the gather over host memory (several ranks on ONE GPU — RCCL refuses duplicate
        # devices —, or the CPU build of the engine in the no-GPU test suite)
        backend = os.environ.get("SCANFOLD_DIST_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(eng.device)
            dist_device = torch.device("cuda", eng.device)
            dist.init_process_group("nccl", device_id=dist_device)
        else:
            dist_device = None
            dist.init_process_group(backend)
    if args.shuffle_backend == "python":
        import random
        random.seed(args.seed)
    w = open(out_path, 'w') if rank == 0 else None
    try:
        for read_name, seq in read_fasta(args.filename):
            if rank == 0:
                print("Scanning sequence " + str(read_name) + "\nSequence Length: " + str(len(seq)) + "nt long.")
            if len(seq) < window_size:
                continue
            constraints = None
            if args.constraints is not None:
                if rank == 0:
                    print("Considering constraint input")
                constraints = read_constraints(args.constraints, len(seq))
            if world > 1:
                rows = scan_record_sharded(seq, window_size, step_size, randomizations, shuffle_type, temperature,
                                           eng, args.seed, rank, world, device=dist_device,
                                           print_random=(args.print_random == "on"), constraints=constraints,
                                           unbalanced=args.constraint_unbalanced)
            else:
                rows = scan_record(seq, window_size, step_size, randomizations, shuffle_type, temperature, eng,
                                   seed=args.seed, shuffle_backend=args.shuffle_backend,
                                   print_random=(args.print_random == "on"), constraints=constraints,
                                   unbalanced=args.constraint_unbalanced)
            if args.timing:
                print("scanfold_amd.scan timing rank %d/%d record %s: %s" % (rank, world, read_name, ", ".join(
                    "%s=%s" % (k, ("%.4f" % v) if isinstance(v, float) else v) for k, v in sorted(LAST_STATS.items()))),
                    file=sys.stderr, flush=True)
            if rank != 0:
                continue
            w.write(header_line(read_name))
            if args.print_to_screen:
                for row in rows:
                    f = row.rstrip("\n").split("\t")
                    print("\t".join(f[:7]) + "\n" + f[7] + "\n" + f[8] + "\n" + f[9] + "\n")
            w.writelines(rows)
    finally:
        if w is not None:
            w.close()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
