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


def test_strong_scaling_splits_one_global_batch():
    """--strong = BASELINE configs[2] verbatim: ONE batch of 512 pockets in contiguous blocks over the ranks (weak scaling keeps
    --batch pockets per rank)."""
    out = _run(['--gpus', '2', '--strong'])
    assert out['scaling'] == 'strong' and out['pocket_blocks'] == [[0, 256], [256, 512]]
    out = _run(['--gpus', '2', '--strong', '--global-batch', '7'])
    assert out['pocket_blocks'] == [[0, 4], [4, 7]]
    out = _run(['--gpus', '2'])
    assert out['scaling'] == 'weak' and out['pocket_blocks'] == [[0, 64], [64, 128]]


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


def test_roofline_flop_accounting_matches_the_kernel():
    """Per-launch FLOP of k_node = what a launch needs (P_c on moving rows only, no next-block P|Q in the last block):
    2.897 GFLOP at the headline shape - the figure SQ_INSTS_MFMA x 2048 FLOP gives (profiles/r01_e_pmc…) - not 14 H^2 N."""
    sys.path.insert(0, ROOT)
    import bench
    H, L, N, Nl = 256, 5, 64 * 59, 64 * 15
    got = bench.node_flop_per_launch(H, L, N, Nl)
    assert abs(got - 1414758 * 2048) / got < 2e-3
    assert got < 14 * H * H * N
    # whole job: L blocks of edge work + 12 H^2 per node + 2 H^2 per moving node, plus the embeddings
    f = bench.whole_job_flop(H, L, 33, 1000.0, 100.0, N, Nl)
    assert abs(f - (L * (2 * (H * H + H) * 1100.0 + 12 * H * H * N + 2 * H * H * Nl) + 4 * 33 * H * N)) < 1.0


def test_traffic_file_is_tied_to_the_kernel_sources():
    """profiles/kernel_traffic.json carries the hash of the kernel sources it was measured on; bench.py uses it only
    when that equals the hash of the sources it runs (a stale PMC number must not be pasted into a new line)."""
    sys.path.insert(0, ROOT)
    import bench
    sha = bench.kernel_source_sha()
    assert len(sha) == 16 and sha == bench.kernel_source_sha()
    tj = json.load(open(os.path.join(ROOT, 'profiles', 'kernel_traffic.json')))
    assert 'kernel_source_sha' in tj and set(tj['hbm_bytes_per_launch']) >= {'edge_msg', 'node', 'edge_coord'}


def test_scale_day_one_script_dry_run(tmp_path):
    """tools/scale_day_one.sh --dry-run-launch: the whole 8-GPU protocol (weak and strong scaling at N = 1, 2, 4, 8) through the real launch
    path on CPU over gloo - every line reports the rank count it COUNTED, the strong runs split one batch of 512 pockets."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['SCALE_OUT'] = str(tmp_path)
    r = subprocess.run(['bash', os.path.join(ROOT, 'tools', 'scale_day_one.sh'), '--dry-run-launch'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    for n in (1, 2, 4, 8):
        weak = json.loads(open(tmp_path / f'weak_n{n}.json').read().strip().splitlines()[-1])
        strong = json.loads(open(tmp_path / f'strong_n{n}.json').read().strip().splitlines()[-1])
        assert weak['n_gpus'] == n and weak['scaling'] == 'weak' and len(weak['pocket_blocks']) == n
        assert strong['n_gpus'] == n and strong['scaling'] == 'strong'
        assert strong['pocket_blocks'][0][0] == 0 and strong['pocket_blocks'][-1][1] == 512
