"""bench.py --gpus N must start its N ranks itself when it was not started by torch.distributed.run (the driver's
plain `python bench.py --gpus 8`), and report the number of ranks that really ran.  CPU test of the launch logic only
(--dry-run-launch: gloo rendezvous, no sampling, no GPU)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--dry-run-launch'] + extra, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout                 # exactly ONE JSON line, whatever the rank count
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks_and_prints_one_line():
    out = _run(['--gpus', '2', '--steps', '1', '--warmup', '0'])
    assert out['n_gpus'] == 2 and out['steps'] == 1


def test_single_rank_does_not_spawn():
    assert _run(['--gpus', '1'])['n_gpus'] == 1


def test_launched_by_torchrun_uses_the_given_ranks():
    """The driver's other form: python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr',
                        '127.0.0.1', '--master-port', str(29700 + os.getpid() % 200), os.path.join(ROOT, 'bench.py'),
                        '--gpus', '2', '--dry-run-launch'], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1 and json.loads(lines[0])['n_gpus'] == 2
