"""Multi-GPU sharding of the window scan: one process per GPU, windows split by contiguous range, ONE gather.

The reference has no distributed layer (its only parallelism is a 12-process pool on one host,
ScanFold-Scan.py:73-77).  Windows are independent and the per-window reduction (z/p over its r+1 energies,
ScanFold-Scan.py:426-433) stays inside a shard when sharding is by window, so the only exchange is a gather of
fixed-size per-window records at the end.  torch.distributed is used for exactly that:
backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.

Record layout (bytes, per window): int32 energies[r+1] | char structure[W+1] | char centroid[W+1] |
pad to 8 | float64 ens_div | float64 ens_dG.

Two users: bench.py keeps everything device-resident and gathers those records (pack_records on the device,
gather_records); the command lines (scan.py, scanfold.py --gpus N) let every rank turn ITS windows into TSV rows —
z-scores, p-scores and formatting are per window, so the host work shards as well as the folds do — and gather the
rows (gather_rows): still one collective per record, and rank 0 only writes.
"""
import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous [lo, hi) of rank's items; every rank but the last gets ceil(n/world)."""
    per = -(-n_items // world) if world > 0 else n_items
    lo = min(n_items, rank * per)
    hi = min(n_items, lo + per)
    return lo, hi


def shard_size(n_items, world):
    return -(-n_items // world)


def record_layout(W, r):
    e = 4 * (r + 1)
    s = W + 1
    txt_end = e + 2 * s
    dbl = (txt_end + 7) & ~7
    return dict(energies=(0, e), structure=(e, e + s), centroid=(e + s, txt_end), ens_div=(dbl, dbl + 8),
                ens_dG=(dbl + 8, dbl + 16), size=dbl + 16)


def pack_records(xp, W, r, energies, structure, centroid, ens_div, ens_dG, n_pad):
    """xp is `torch` or `numpy`; inputs are arrays/tensors of n rows; returns uint8 [n_pad, size]."""
    lay = record_layout(W, r)
    n = energies.shape[0]
    if xp.__name__ == "torch":
        rec = xp.zeros((n_pad, lay["size"]), dtype=xp.uint8, device=energies.device)
        as_u8 = lambda t, width: t.contiguous().view(xp.uint8).reshape(n, width)
    else:
        rec = xp.zeros((n_pad, lay["size"]), dtype=xp.uint8)
        as_u8 = lambda t, width: xp.ascontiguousarray(t).view(xp.uint8).reshape(n, width)
    if n == 0:  # an empty shard (more ranks than windows): nothing to copy, the padding is the record
        return rec
    for key, arr in (("energies", energies), ("structure", structure), ("centroid", centroid),
                     ("ens_div", ens_div), ("ens_dG", ens_dG)):
        lo, hi = lay[key]
        rec[:n, lo:hi] = as_u8(arr, hi - lo)
    return rec


def unpack_records(rec, W, r, n):
    """uint8 numpy [>=n, size] -> dict of numpy arrays for the first n windows."""
    lay = record_layout(W, r)
    rec = np.ascontiguousarray(rec[:n])
    get = lambda key: np.ascontiguousarray(rec[:, lay[key][0]:lay[key][1]])
    return dict(energies=get("energies").view(np.int32).reshape(n, r + 1),
                structure=get("structure"), centroid=get("centroid"),
                ens_div=get("ens_div").view(np.float64).reshape(n),
                ens_dG=get("ens_dG").view(np.float64).reshape(n))


def gather_records(local_rec, world, force=False):
    """One all-gather of the equal-sized shards; returns the [world*n_pad, size] tensor on every rank.
    With one rank nothing is exchanged unless `force` (bench.py's single-rank check of the RCCL path)."""
    import torch
    import torch.distributed as dist
    if world == 1 and not force:
        return local_rec
    if dist.get_backend() == "gloo" and local_rec.is_cuda:
        # gloo (the tests' stand-in for RCCL when several ranks share ONE GPU): the shard is staged through the host for
        # the collective only, exactly as gather_rows does; packing stays on the device
        local_rec = local_rec.cpu()
    out = torch.empty((world * local_rec.shape[0], local_rec.shape[1]), dtype=torch.uint8, device=local_rec.device)
    dist.all_gather_into_tensor(out, local_rec.contiguous())
    return out


def merge_shards(gathered, n_items, world, W, r):
    """Drop each shard's padding and restore window order -> dict of numpy arrays for all n_items windows."""
    per = shard_size(n_items, world)
    g = gathered.cpu().numpy() if hasattr(gathered, "cpu") else np.asarray(gathered)
    parts = []
    for rank in range(world):
        lo, hi = shard_range(n_items, rank, world)
        parts.append(g[rank * per: rank * per + (hi - lo)])
    rec = np.concatenate(parts, axis=0) if parts else g[:0]
    return unpack_records(rec, W, r, n_items)


def row_slot_width(W):
    """Bytes reserved per TSV row in the row gather: i, j, T, dG, z, p, ED (< 96 characters together with the tabs and
    the gc_content column of the ScanFold.py flavour) + sequence, structure, centroid."""
    return 3 * W + 128


def pack_rows(rows, n_pad, width):
    """list of row strings -> uint8 numpy [n_pad, width], zero-padded (a row never contains a zero byte)."""
    enc = [row.encode("ascii") for row in rows]
    if enc and max(map(len, enc)) > width:
        raise ValueError("a TSV row of %d bytes does not fit the %d-byte gather slot" % (max(map(len, enc)), width))
    blob = b"".join(b.ljust(width, b"\0") for b in enc) + bytes((n_pad - len(enc)) * width)
    return np.frombuffer(blob, dtype=np.uint8).reshape(n_pad, width).copy()


def unpack_rows(buf, n):
    """uint8 [>= n, width] -> list of n row strings (every row ends with its newline)."""
    text = np.ascontiguousarray(buf[:n]).tobytes().replace(b"\0", b"").decode("ascii")
    rows = text.splitlines(keepends=True)
    if len(rows) != n:
        raise ValueError("row gather: %d rows unpacked, %d expected" % (len(rows), n))
    return rows


def any_rank_failed(failed_here, world, device=None):
    """One int32 all-reduce (MAX) of "my shard failed": True on every rank if it did on any.  Called by every rank before
    the gather, so that a rank-local error cannot leave the others blocked in the collective."""
    if world == 1:
        return bool(failed_here)
    import torch
    import torch.distributed as dist
    flag = torch.tensor([1 if failed_here else 0], dtype=torch.int32)
    if device is not None:
        flag = flag.to(device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    return bool(int(flag.item()))


def gather_rows(rows, n_items, rank, world, W, device=None, want=True, prepacked=None):
    """Every rank hands in the TSV rows of ITS window range (formatted where they were computed); ONE all-gather of
    fixed-size row slots returns all n_items rows in window order (want=False: None — a rank that does not write).
    `device`: where the collective's tensors live (the rank's GPU under nccl/RCCL, None = host under gloo).
    `prepacked`: pack_rows(rows, shard_size(n_items, world), row_slot_width(W)) if the caller has made it already (the sharded
    scan packs before its status all-reduce, so that a row that does not fit fails on every rank, not inside the collective)."""
    import torch
    import torch.distributed as dist
    lo, hi = shard_range(n_items, rank, world)
    if len(rows) != hi - lo:
        raise ValueError("rank %d formatted %d rows for the range [%d, %d)" % (rank, len(rows), lo, hi))
    if world == 1:
        return list(rows)
    per, width = shard_size(n_items, world), row_slot_width(W)
    if prepacked is not None and prepacked.shape != (per, width):
        raise ValueError("prepacked rows have shape %r, the gather needs %r" % (prepacked.shape, (per, width)))
    local = torch.from_numpy(prepacked if prepacked is not None else pack_rows(rows, per, width))
    if device is not None:
        local = local.to(device)
    out = torch.empty((world * per, width), dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    if not want:
        return None
    g = out.cpu().numpy()
    merged = []
    for rk in range(world):
        a, b = shard_range(n_items, rk, world)
        merged.extend(unpack_rows(g[rk * per: rk * per + (b - a)], b - a))
    return merged


def scan_sharded(produce, n_win, W, r, rank, world, xp):
    """Run `produce(lo, hi)` -> (energies, structure, centroid, ens_div, ens_dG) on this rank's windows,
    gather every shard, and return the merged dict (same on all ranks)."""
    lo, hi = shard_range(n_win, rank, world)
    e, s, c, d, g = produce(lo, hi)
    rec = pack_records(xp, W, r, e, s, c, d, g, shard_size(n_win, world))
    if xp.__name__ == "torch":
        gathered = gather_records(rec, world)
    else:
        gathered = rec  # numpy path is single-process only
    return merge_shards(gathered, n_win, world, W, r)
