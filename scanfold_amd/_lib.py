"""ctypes binding of libscanfold_hip.so (include/scanfold_hip.h) — the only compute path of this package.

There is no CPU fallback: if the shared library is missing or no GPU is usable, `get_engine()` raises.
(The reference's compute path is ViennaRNA on the CPU through a 12-process pool,
ScanFold-Scan.py:73-77,244-262; nothing of it is kept.)
"""
import ctypes
import os

import numpy as np

from . import params as _params

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libscanfold_hip.so")

SHUFFLE_MONO = 0
SHUFFLE_DI = 1
SCAN_NO_PF = 1
SCAN_NO_TRACE = 2

_c_u8p = ctypes.c_void_p
_EXPORTS = {
    "sf_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "sf_last_hip_error": (ctypes.c_char_p, []),
    "sf_init": (ctypes.c_int, [ctypes.c_int]),
    "sf_shutdown": (ctypes.c_int, []),
    "sf_device_name": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_size_t]),
    "sf_params_load": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double]),
    "sf_params_load_rescaled": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_char_p,
                                               ctypes.c_char_p]),
    "sf_mfe_batch": (ctypes.c_int, [_c_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "sf_mfe_batch_dev": (ctypes.c_int, [_c_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "sf_mfe_trace_batch": (ctypes.c_int, [_c_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "sf_pf_batch": (ctypes.c_int, [_c_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                   ctypes.c_void_p, ctypes.c_void_p]),
    "sf_fold_constrained": (ctypes.c_int, [_c_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "sf_shuffle_windows": (ctypes.c_int, [_c_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint64,
                                          ctypes.c_void_p]),
    "sf_scan": (ctypes.c_int, [_c_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                               ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint, ctypes.c_void_p,
                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "sf_scan_dev": (ctypes.c_int, [_c_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint, ctypes.c_void_p,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                   ctypes.c_void_p]),
    "sf_tabulate_pairs": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.POINTER(ctypes.c_int64)]),
    "sf_tabulate_fetch": (ctypes.c_int, [ctypes.c_void_p] * 7),
    "sf_last_status": (ctypes.c_int, []),
    "sf_prof_stop": (ctypes.c_int, []),
    "sf_set_kernel_mode": (ctypes.c_int, [ctypes.c_int]),
    "sf_set_max_bp_span": (ctypes.c_int, [ctypes.c_int]),
    "sf_prof_reset": (ctypes.c_int, []),
    "sf_prof_get": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64),
                                   ctypes.POINTER(ctypes.c_int64)]),
}
EXPORTED_SYMBOLS = tuple(_EXPORTS)


class ScanFoldHipError(RuntimeError):
    pass


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64 (same SONAME, different file name).  Two HIP runtimes in
    one process do not work ("No HIP GPUs are available" in whichever initialises second), so when torch is
    installed its copy is loaded first and this library binds to it by SONAME; torch later finds the same
    file already loaded.  Without torch the system runtime under /opt/rocm is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load_library(path=LIB_PATH):
    """dlopen the C-ABI library and declare every prototype; raises if it or a symbol is missing."""
    if not os.path.exists(path):
        raise ScanFoldHipError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
    if os.path.abspath(path) == os.path.abspath(LIB_PATH):
        _share_hip_runtime_with_torch()
    lib = ctypes.CDLL(path)
    for name, (res, args) in _EXPORTS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


def seqs_to_array(seqs, W=None):
    """list of equal-length str / bytes, or a uint8 (n, W) array -> contiguous uint8 (n, W)."""
    if isinstance(seqs, np.ndarray):
        arr = np.ascontiguousarray(seqs, dtype=np.uint8)
        if arr.ndim != 2:
            raise ValueError("sequence array must be 2-D (n, W)")
        return arr
    seqs = [s if isinstance(s, (bytes, bytearray)) else str(s).encode("ascii") for s in seqs]
    if not seqs:
        return np.zeros((0, W or 1), dtype=np.uint8)
    W = len(seqs[0])
    if any(len(s) != W for s in seqs):
        raise ValueError("all sequences of one batch must have the same length")
    return np.frombuffer(b"".join(seqs), dtype=np.uint8).reshape(len(seqs), W)


class Engine:
    """One process, one GPU.  Thin, stateful wrapper over the C ABI."""

    def __init__(self, device=0, paramset=None, lib_path=LIB_PATH):
        self.lib = load_library(lib_path)
        self._check(self.lib.sf_init(int(device)))
        self.device = int(device)
        self.params = None
        self.load_params(paramset if paramset is not None else _params.default_params())

    # -- plumbing --
    def _check(self, rc):
        if rc != 0:
            msg = self.lib.sf_strerror(rc).decode()
            if rc == -6:
                msg += ": " + self.lib.sf_last_hip_error().decode()
            raise ScanFoldHipError(msg)

    def device_name(self):
        buf = ctypes.create_string_buffer(256)
        self._check(self.lib.sf_device_name(buf, 256))
        return buf.value.decode()

    def load_params(self, paramset, temperature=None):
        """RNA.md() + RNA.fold_compound(seq, md): make `paramset` the folding model.  `temperature` other than the
        set's own rescales it first (ParamSet.at_temperature: needs the enthalpy tables of a real .par file)."""
        if temperature is not None and float(temperature) != paramset.temperature:
            paramset = paramset.at_temperature(temperature)
        self._load(paramset)
        self.params = paramset
        self._by_temp = {paramset.temperature: paramset}

    def _load(self, p):
        blob = p.blob()
        b37, bdh = p.rescale_blobs()
        self._check(self.lib.sf_params_load_rescaled(blob, len(blob), p.temperature, b37, bdh))

    def set_temperature(self, temperature):
        """md.temperature = T (ScanFold-Scan.py:70-71; ScanFoldFunctions.py:776-777): switch the resident model to the
        loaded set rescaled to T; a no-op when it is already there.  Raises NotImplementedError for a set without
        enthalpies (the reconstructed default) when T is not the set's temperature."""
        t = float(temperature)
        if t == self.params.temperature:
            return
        cache = self._by_temp
        if t not in cache:
            cache[t] = self.params.at_temperature(t)
        p = cache[t]
        self._load(p)
        self.params = p

    def shutdown(self):
        self._check(self.lib.sf_shutdown())

    # -- host-buffer entry points --
    def mfe_batch(self, seqs):
        arr = seqs_to_array(seqs)
        n, W = arr.shape
        out = np.empty(n, dtype=np.int32)
        self._check(self.lib.sf_mfe_batch(arr.ctypes.data, n, W, out.ctypes.data))
        return out

    def mfe_trace_batch(self, seqs):
        arr = seqs_to_array(seqs)
        n, W = arr.shape
        out = np.empty(n, dtype=np.int32)
        db = np.zeros((n, W + 1), dtype=np.uint8)
        self._check(self.lib.sf_mfe_trace_batch(arr.ctypes.data, n, W, out.ctypes.data, db.ctypes.data))
        return out, [bytes(row[:W]).decode() for row in db]

    def pf_batch(self, seqs):
        arr = seqs_to_array(seqs)
        n, W = arr.shape
        dG = np.empty(n)
        mbd = np.empty(n)
        cd = np.empty(n)
        cen = np.zeros((n, W + 1), dtype=np.uint8)
        self._check(self.lib.sf_pf_batch(arr.ctypes.data, n, W, dG.ctypes.data, mbd.ctypes.data, cen.ctypes.data,
                                         cd.ctypes.data))
        return dict(dG=dG, mean_bp_dist=mbd, centroid=[bytes(r[:W]).decode() for r in cen], centroid_dist=cd)

    def fold_constrained(self, seqs, cons=None, sc_stack_dcal=None, mfe=True, pf=True):
        """fc.hc_add_from_db / fc.sc_add_SHAPE_deigan + fc.mfe() / fc.pf() on n windows (ScanFold-Scan.py:405-418;
        ScanFold.py:508-544).  cons: list of W-char str or uint8 (n, W); sc_stack_dcal: int32 (n, W).
        -> dict(mfe, structure [str], dG, mean_bp_dist, centroid [str], centroid_dist) (the keys that were asked for)"""
        arr = seqs_to_array(seqs)
        n, W = arr.shape
        c = None if cons is None else seqs_to_array(cons)
        if c is not None and c.shape != (n, W):
            raise ValueError("constraint rows must match the sequence rows")
        s = None if sc_stack_dcal is None else np.ascontiguousarray(sc_stack_dcal, dtype=np.int32).reshape(n, W)
        e = np.zeros(n, dtype=np.int32)
        db = np.zeros((n, W + 1), dtype=np.uint8)
        dG, mbd, cd = np.zeros(n), np.zeros(n), np.zeros(n)
        cen = np.zeros((n, W + 1), dtype=np.uint8)
        flags = (0 if pf else 1) | (0 if mfe else 2)
        self._check(self.lib.sf_fold_constrained(arr.ctypes.data, n, W, None if c is None else c.ctypes.data,
                                                 None if s is None else s.ctypes.data, flags, e.ctypes.data,
                                                 db.ctypes.data, dG.ctypes.data, mbd.ctypes.data, cen.ctypes.data,
                                                 cd.ctypes.data))
        out = {}
        if mfe:
            out.update(mfe=e, structure=[bytes(r[:W]).decode() for r in db])
        if pf:
            out.update(dG=dG, mean_bp_dist=mbd, centroid=[bytes(r[:W]).decode() for r in cen], centroid_dist=cd)
        return out

    def shuffle_windows(self, transcript, W, step, win_begin, n_win, r, kind, seed):
        tr = np.frombuffer(transcript.encode("ascii") if isinstance(transcript, str) else bytes(transcript),
                           dtype=np.uint8)
        out = np.empty((n_win * (r + 1), W), dtype=np.uint8)
        self._check(self.lib.sf_shuffle_windows(tr.ctypes.data, len(tr), W, step, win_begin, n_win, r, kind,
                                                ctypes.c_uint64(seed), out.ctypes.data))
        return out

    def scan(self, transcript, W, step, win_begin, n_win, r, kind, seed, flags=0, raw=False):
        """-> dict(energies int32 (n_win, r+1), structure [str], centroid [str], ens_div, ens_dG);
        raw=True leaves structure / centroid as the uint8 arrays (n_win, W+1) the library filled."""
        tr = np.frombuffer(transcript.encode("ascii") if isinstance(transcript, str) else bytes(transcript),
                           dtype=np.uint8)
        en = np.empty((n_win, r + 1), dtype=np.int32)
        db = np.zeros((n_win, W + 1), dtype=np.uint8)
        cen = np.zeros((n_win, W + 1), dtype=np.uint8)
        div = np.zeros(n_win)
        dG = np.zeros(n_win)
        self._check(self.lib.sf_scan(tr.ctypes.data, len(tr), W, step, win_begin, n_win, r, kind,
                                     ctypes.c_uint64(seed), flags, en.ctypes.data, db.ctypes.data, cen.ctypes.data,
                                     div.ctypes.data, dG.ctypes.data))
        if raw:
            return dict(energies=en, structure=db, centroid=cen, ens_div=div, ens_dG=dG)
        return dict(energies=en, structure=[bytes(x[:W]).decode() for x in db],
                    centroid=[bytes(x[:W]).decode() for x in cen], ens_div=div, ens_dG=dG)

    def tabulate_pairs(self, structures, starts, z, mfe, ed, W=None, row_stride=None, on_device=False):
        """Base-pair tabulation of a scan table (ScanFold-Fold.py:583-682,704-760) -> dict of per-group arrays
        k, j (j == k: unpaired), windows, first_window, sum_z, sum_mfe, sum_ed, ordered by k then first window.
        structures: list of W-char str, a uint8 array (n, >= W) (e.g. the raw (n, W+1) table of scan(raw=True)), or —
        with on_device=True, W and row_stride given — the device pointer of the table sf_scan_dev wrote."""
        starts = np.ascontiguousarray(starts, dtype=np.int32)
        n = len(starts)
        z, mfe, ed = (np.ascontiguousarray(v, dtype=np.float64) for v in (z, mfe, ed))
        if not (len(z) == len(mfe) == len(ed) == n):
            raise ValueError("one z-score, MFE and ED per window")
        if on_device:
            ptr, keep = int(structures), None
        else:
            keep = seqs_to_array(structures) if not isinstance(structures, np.ndarray) else np.ascontiguousarray(structures, dtype=np.uint8)
            if keep.ndim != 2 or keep.shape[0] != n:
                raise ValueError("one structure row per window")
            row_stride = keep.shape[1]
            W = row_stride if W is None else W
            ptr = keep.ctypes.data
        ng = ctypes.c_int64(0)
        self._check(self.lib.sf_tabulate_pairs(ptr, int(row_stride), 1 if on_device else 0, n, int(W), starts.ctypes.data,
                                               z.ctypes.data, mfe.ctypes.data, ed.ctypes.data, ctypes.byref(ng)))
        g = int(ng.value)
        out = dict(k=np.empty(g, np.int32), j=np.empty(g, np.int32), windows=np.empty(g, np.int32),
                   first_window=np.empty(g, np.int32), sum_z=np.empty(g), sum_mfe=np.empty(g), sum_ed=np.empty(g))
        self._check(self.lib.sf_tabulate_fetch(*(out[key].ctypes.data for key in
                                                 ("k", "j", "windows", "first_window", "sum_z", "sum_mfe", "sum_ed"))))
        return out

    # -- device-buffer entry points (pointers are ints, e.g. torch.Tensor.data_ptr(); stream 0 = library stream) --
    def mfe_batch_dev(self, d_seqs, n, W, d_out, stream=0):
        self._check(self.lib.sf_mfe_batch_dev(d_seqs, n, W, d_out, stream))

    def scan_dev(self, d_transcript, L, W, step, win_begin, n_win, r, kind, seed, flags, d_energies, d_structure,
                 d_centroid, d_ens_div, d_ens_dG, stream=0):
        self._check(self.lib.sf_scan_dev(d_transcript, L, W, step, win_begin, n_win, r, kind, ctypes.c_uint64(seed),
                                         flags, d_energies, d_structure, d_centroid, d_ens_div, d_ens_dG, stream))

    def last_status(self):
        """0, or SF_ERR_INTERNAL (-8) if a traceback of an asynchronous call failed; waits for the device."""
        rc = self.lib.sf_last_status()
        if rc not in (0, -8):
            self._check(rc)
        return rc

    def prof_stop(self):
        self._check(self.lib.sf_prof_stop())

    def set_kernel_mode(self, mode):
        self._check(self.lib.sf_set_kernel_mode(int(mode)))

    def set_max_bp_span(self, span):
        """RNA.md().max_bp_span (ScanFold.py:214-215): pairs (i, j) with j - i + 1 > span do not exist; <= 0 = no limit."""
        self._check(self.lib.sf_set_max_bp_span(int(span or 0)))

    def prof_reset(self):
        self._check(self.lib.sf_prof_reset())

    def prof_get(self):
        ms = ctypes.c_double()
        nl = ctypes.c_int64()
        nf = ctypes.c_int64()
        self._check(self.lib.sf_prof_get(ctypes.byref(ms), ctypes.byref(nl), ctypes.byref(nf)))
        return ms.value, nl.value, nf.value


_engine = None


def get_engine(device=None):
    """Process-wide engine on LOCAL_RANK's GPU (or `device`)."""
    global _engine
    if _engine is None:
        if device is None:
            # SCANFOLD_DEVICE: several ranks on one GPU (the 2-rank test on a 1-GPU box); default: the rank's own GPU
            device = int(os.environ.get("SCANFOLD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        # SCANFOLD_LIB_PATH: another build of the same C ABI (build variants; tests/emul's CPU build in the no-GPU suite)
        _engine = Engine(device=device, lib_path=os.environ.get("SCANFOLD_LIB_PATH", LIB_PATH))
        _params.warn_if_reconstructed(_engine.params)  # the shipped table is not ViennaRNA's: say so once
    return _engine
