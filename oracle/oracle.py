"""ctypes wrapper over oracle/libsf_oracle.so.  TEST INFRASTRUCTURE ONLY (see sf_oracle.c header).

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by scanfold_amd.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libsf_oracle.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]  # sf_cpu_twin.c is #included by sf_oracle.c
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])


BASE_FLAGS = "-O3 -std=c11 -fPIC -fopenmp"  # oracle/Makefile CFLAGS (warnings aside)
NATIVE_FLAGS = "-O3 -march=native -std=c11 -fPIC -fopenmp"  # oracle/Makefile NATIVE_FLAGS (warnings aside)
_NATIVE_DIR = os.path.join(_HERE, "_native")
_NATIVE_LIB = os.path.join(_NATIVE_DIR, "libsf_oracle_native.so")
_native = None


def _cpu_stamp():
    """What `-march=native` depends on: the host's CPU model and ISA flags, and the compiler."""
    model = flags = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and not model:
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("flags") and not flags:
                flags = ln.split(":", 1)[1].strip()
            if model and flags:
                break
    except OSError:
        pass
    cc = subprocess.run(["gcc", "--version"], capture_output=True, text=True).stdout.splitlines()[:1]
    import hashlib
    return "%s | %s | %s | %s" % (model, hashlib.sha1(flags.encode()).hexdigest()[:12], cc[0] if cc else "gcc ?", NATIVE_FLAGS)


def build_native():
    """The same sources a second time, `-O3 -march=native`, ON THE MACHINE THAT WILL RUN IT (the object must not
    travel between hosts: oracle/_native/ is git- and gpurun-ignored and carries a stamp of the CPU it was built for).
    bench.py's cpu_baseline times this build beside the portable one.  -> path of the library."""
    stamp_path = os.path.join(_NATIVE_DIR, "built_for.txt")
    stamp = _cpu_stamp()
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
    fresh = (os.path.exists(_NATIVE_LIB) and os.path.exists(stamp_path) and open(stamp_path).read() == stamp
             and os.path.getmtime(_NATIVE_LIB) >= max(os.path.getmtime(f) for f in srcs))
    if not fresh:
        if os.path.exists(_NATIVE_LIB):
            os.remove(_NATIVE_LIB)
        subprocess.check_call(["make", "-C", _HERE, "-s", "native"])
        with open(stamp_path, "w") as f:
            f.write(stamp)
    return _NATIVE_LIB


def lib_native():
    """The -march=native build as a SECOND library instance (its own parameter tables: call set_params(p, L=lib_native()))."""
    global _native
    if _native is None:
        _native = _bind(ctypes.CDLL(build_native()))
    return _native


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = _bind(ctypes.CDLL(_LIB))
    return _lib


def _bind(L):
    L.sfo_set_params.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    L.sfo_set_params_exact.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_char_p]
    L.sfo_mfe.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_char_p]
    L.sfo_mfe_batch.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    L.sfo_eval.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.sfo_brute.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                            ctypes.POINTER(ctypes.c_double), ctypes.c_void_p,
                            ctypes.POINTER(ctypes.c_longlong)]
    L.sfo_pf.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_void_p,
                         ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    L.sfo_scan_windows.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    L.sfo_twin_scan_windows.argtypes = L.sfo_scan_windows.argtypes
    L.sfo_twin_mfe_batch.argtypes = L.sfo_mfe_batch.argtypes
    L.sfo_set_constraint.argtypes = [ctypes.c_char_p, ctypes.c_void_p]
    L.sfo_shuffle_windows.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p]
    return L


def set_params(paramset, L=None):
    L = L or lib()
    blob = paramset.blob()
    assert len(blob) == L.sfo_params_size(), (len(blob), L.sfo_params_size())
    b37, bdh = paramset.rescale_blobs()
    rc = L.sfo_set_params_exact(blob, len(blob), b37, bdh)
    if rc:
        raise RuntimeError("sfo_set_params rc=%d" % rc)


def set_max_bp_span(span):
    """RNA.md().max_bp_span: pairs (i, j) with j - i + 1 > span do not exist; <= 0 removes the limit."""
    lib().sfo_set_max_bp_span(int(span or 0))


_keep = []


def set_constraint(cons=None, sc_stack_dcal=None):
    """fc.hc_add_from_db(cons) / fc.sc_add_SHAPE_deigan pseudo-energies (int dcal per nucleotide, stacks only) for the
    following single-sequence calls (mfe, eval_structure, brute, pf); None, None clears.  See sf_oracle.c."""
    c = cons.encode() if isinstance(cons, str) else cons
    s = None if sc_stack_dcal is None else np.ascontiguousarray(sc_stack_dcal, dtype=np.int32)
    _keep[:] = [c, s]  # the C side keeps the pointers
    lib().sfo_set_constraint(c, None if s is None else s.ctypes.data)


def mfe(seq, structure=True):
    s = seq.encode()
    e = ctypes.c_int()
    buf = ctypes.create_string_buffer(len(s) + 1) if structure else None
    rc = lib().sfo_mfe(s, len(s), ctypes.byref(e), buf)
    if rc:
        raise RuntimeError("sfo_mfe rc=%d" % rc)
    return (buf.value.decode() if structure else None), e.value


def mfe_batch(seqs, nthreads=0):
    """seqs: list of equal-length str, or uint8 array (n, W) of ASCII. Returns int32 array (dcal/mol)."""
    if isinstance(seqs, np.ndarray):
        arr = np.ascontiguousarray(seqs, dtype=np.uint8)
    else:
        arr = np.frombuffer("".join(seqs).encode(), dtype=np.uint8).reshape(len(seqs), -1)
    n, W = arr.shape
    out = np.empty(n, dtype=np.int32)
    rc = lib().sfo_mfe_batch(arr.ctypes.data_as(ctypes.c_char_p), n, W, out.ctypes.data, nthreads)
    if rc:
        raise RuntimeError("sfo_mfe_batch rc=%d" % rc)
    return out


def eval_structure(seq, db):
    e = ctypes.c_int()
    rc = lib().sfo_eval(seq.encode(), db.encode(), len(seq), ctypes.byref(e))
    if rc:
        raise RuntimeError("sfo_eval rc=%d" % rc)
    return e.value


def brute(seq, want_bpp=False):
    n = len(seq)
    e = ctypes.c_int()
    Z = ctypes.c_double()
    cnt = ctypes.c_longlong()
    bpp = np.zeros((n + 1, n + 1)) if want_bpp else None
    rc = lib().sfo_brute(seq.encode(), n, ctypes.byref(e), ctypes.byref(Z),
                         bpp.ctypes.data if want_bpp else None, ctypes.byref(cnt))
    if rc:
        raise RuntimeError("sfo_brute rc=%d" % rc)
    return e.value, Z.value, bpp, cnt.value


def pf(seq, want_bpp=False):
    """-> dict(dG, centroid, centroid_dist, mean_bp_dist, bpp)"""
    n = len(seq)
    dG = ctypes.c_double()
    cd = ctypes.c_double()
    mbd = ctypes.c_double()
    cen = ctypes.create_string_buffer(n + 1)
    bpp = np.zeros((n + 1, n + 1)) if want_bpp else None
    rc = lib().sfo_pf(seq.encode(), n, ctypes.byref(dG), bpp.ctypes.data if want_bpp else None, cen,
                      ctypes.byref(cd), ctypes.byref(mbd))
    if rc:
        raise RuntimeError("sfo_pf rc=%d" % rc)
    return dict(dG=dG.value, centroid=cen.value.decode(), centroid_dist=cd.value, mean_bp_dist=mbd.value, bpp=bpp)


def scan_windows(rows, n_win, r, nthreads=0):
    """rows: uint8 ASCII (n_win*(r+1), W), row 0 of each window native.  Whole per-window job, one OpenMP thread
    per window.  -> dict(energies (n_win, r+1), structure, centroid, ens_div)"""
    arr = np.ascontiguousarray(rows, dtype=np.uint8)
    W = arr.shape[1]
    en = np.empty((n_win, r + 1), dtype=np.int32)
    db = np.zeros((n_win, W + 1), dtype=np.uint8)
    cen = np.zeros((n_win, W + 1), dtype=np.uint8)
    ed = np.zeros(n_win)
    rc = lib().sfo_scan_windows(arr.ctypes.data_as(ctypes.c_char_p), n_win, r, W, en.ctypes.data, db.ctypes.data,
                                cen.ctypes.data, ed.ctypes.data, nthreads)
    if rc:
        raise RuntimeError("sfo_scan_windows rc=%d" % rc)
    return dict(energies=en, structure=[bytes(x[:W]).decode() for x in db],
                centroid=[bytes(x[:W]).decode() for x in cen], ens_div=ed)


def shuffle_windows(transcript, W, step, win_begin, n_win, r, kind, seed):
    """The shuffle background on the product's Philox stream (sf_shuffle_oracle.c): uint8 codes
    (n_win*(r+1), W), same contract as Engine.shuffle_windows."""
    tr = transcript.encode("ascii") if isinstance(transcript, str) else bytes(transcript)
    out = np.empty((n_win * (r + 1), W), dtype=np.uint8)
    rc = lib().sfo_shuffle_windows(tr, len(tr), W, step, win_begin, n_win, r, kind, ctypes.c_uint64(seed),
                                   out.ctypes.data)
    if rc:
        raise RuntimeError("sfo_shuffle_windows rc=%d" % rc)
    return out


def twin_available():
    return hasattr(lib(), "sfo_twin_scan_windows")


def twin_scan_windows(rows, n_win, r, nthreads=0, L=None):
    """sf_cpu_twin.c: the same job as scan_windows on the fast CPU engine (bench.py's cpu_baseline); L = lib_native() runs
    the -march=native build."""
    arr = np.ascontiguousarray(rows, dtype=np.uint8)
    W = arr.shape[1]
    en = np.empty((n_win, r + 1), dtype=np.int32)
    db = np.zeros((n_win, W + 1), dtype=np.uint8)
    cen = np.zeros((n_win, W + 1), dtype=np.uint8)
    ed = np.zeros(n_win)
    rc = (L or lib()).sfo_twin_scan_windows(arr.ctypes.data_as(ctypes.c_char_p), n_win, r, W, en.ctypes.data, db.ctypes.data,
                                            cen.ctypes.data, ed.ctypes.data, nthreads)
    if rc:
        raise RuntimeError("sfo_twin_scan_windows rc=%d" % rc)
    return dict(energies=en, structure=[bytes(x[:W]).decode() for x in db],
                centroid=[bytes(x[:W]).decode() for x in cen], ens_div=ed)


def twin_mfe_batch(seqs, nthreads=0):
    arr = np.ascontiguousarray(seqs, dtype=np.uint8)
    n, W = arr.shape
    out = np.empty(n, dtype=np.int32)
    rc = lib().sfo_twin_mfe_batch(arr.ctypes.data_as(ctypes.c_char_p), n, W, out.ctypes.data, nthreads)
    if rc:
        raise RuntimeError("sfo_twin_mfe_batch rc=%d" % rc)
    return out
