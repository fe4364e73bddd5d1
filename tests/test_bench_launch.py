"""bench.py's launch contract, as far as a machine without GPUs can show it.

`python bench.py --gpus N` must work when the driver calls it plainly: it starts `python -m torch.distributed.run
--nproc-per-node N bench.py ...` as a child, relays the JSON line and the exit code.  Here there is no GPU, so
every rank stops at device selection with a clear message — what matters is that N ranks were spawned and that
the failure is that one, not a usage error."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, env=None):
    e = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True,
                          timeout=600, env=e)


def test_gpus_2_spawns_two_ranks_and_fails_only_at_device_selection():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs present: the real run is the driver's job")
    out = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert out.returncode != 0
    assert "rank 0: --gpus 2 needs 2 GPUs" in out.stderr and "rank 1: --gpus 2 needs 2 GPUs" in out.stderr, out.stderr[-3000:]
    assert "GPU(s) visible" in out.stderr
    assert '"metric"' not in out.stdout


def test_world_size_mismatch_is_reported():
    out = run_bench("--gpus", "1", env={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert out.returncode == 2 and "WORLD_SIZE=2" in out.stderr
