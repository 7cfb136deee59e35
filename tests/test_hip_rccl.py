"""RCCL first contact on the one-GPU box (train.py:111-121 is DDP over NCCL in the reference; BASELINE configs[2] / [3] are 8-GPU runs).

No multi-GPU node has ever been available to this build, so the N > 1 paths are covered by world-size-2 gloo tests on CPU - and HERE the exact
RCCL code path runs with ONE rank: `bench.py --force-dist` initialises the `nccl` (= RCCL) process group and goes through its fence (barrier),
rank count (all-reduce of ones) and MAX-of-elapsed-time all-reduce; `tools/bench_train.py --force-dist` puts HipTrainer's data-parallel step
- parameter broadcast, staged backward, asynchronous all-reduce of every finished gradient chunk on RCCL's stream, division by the world size -
through the same group.  Each rank is a FRESH child process that initialises the GPU itself (never a re-exec of a process that touched it)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_rank0(cmd):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), OMP_NUM_THREADS='8')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    r = subprocess.run([sys.executable] + cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{') and '"metric"' in ln]
    assert lines, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads(lines[-1])


def test_bench_fence_and_reductions_run_through_rccl():
    d = run_rank0(['bench.py', '--gpus', '1', '--steps', '1', '--warmup', '1', '--batch', '8', '--timesteps', '40', '--force-dist',
                   '--no-extra-shapes', '--no-cpu-baseline', '--north-star-batch', '0'])
    assert d['n_gpus'] == 1 and d['value'] > 0 and d['collectives'].startswith('nccl')
    assert d['config']['chain_status']['nan_resets'] == 0


def test_training_step_chunked_allreduce_runs_through_rccl():
    d = run_rank0(['tools/bench_train.py', '--gpus', '1', '--batch', '8', '--steps', '3', '--warmup', '1', '--force-dist'])
    assert d['collectives'] == 'nccl' and d['n_gpus'] == 1 and d['value'] > 0
    ar = d['allreduce_ms']
    assert ar is not None and ar['overlapped_chunks_ms'] > 0 and ar['flat_after_backward_ms'] > 0
    import math
    assert math.isfinite(d['first_loss']) and math.isfinite(d['last_loss'])


def test_collectives_never_leave_an_event_on_a_stream_the_library_captures():
    """cmdgen_amd/collectives.py: a blocking RCCL collective on the chain's stream followed by a capture on that stream aborts the process when
    the backend's watchdog polls during the capture (profiles/r06_o_rccl_watchdog_capture.txt).  The fixed form - async_op=True, then wait -
    through 30 x [barrier -> sleep sweeping the watchdog's poll period -> capture of 100 evaluations]."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    r = subprocess.run([sys.executable, 'tools/rccl_capture_probe.py', 'async', '30', '100'], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0 and 'no abort' in r.stdout, r.stderr[-3000:]
