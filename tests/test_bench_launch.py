"""`python bench.py --gpus N` must start its own N ranks when no launcher did (reference: `accelerate launch`, README.md:344-358).
CPU-only: ALM_BENCH_LAUNCH_CHECK=1 makes every rank join a gloo group, all-reduce once and leave before anything needs a GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'ALM_BENCH_CHILD')}
    env.update(ALM_BENCH_LAUNCH_CHECK='1', **(extra_env or {}))
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *args], env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith('{')]


@pytest.mark.parametrize('n', [2, 3])
def test_gpus_n_without_world_size_launches_n_ranks_and_prints_one_line(n):
    r = _run(['--gpus', str(n), '--steps', '2', '--warmup', '1'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    assert lines[0] == {'launch_check': True, 'n_gpus': n, 'steps': 2, 'warmup': 1}
    assert f'launching {n} ranks myself' in r.stderr


def test_gpus_equals_form_and_single_gpu_default():
    r = _run(['--gpus=2'])
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_lines(r.stdout)[0]['n_gpus'] == 2
    r = _run([])
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_lines(r.stdout) == [{'launch_check': True, 'n_gpus': 1, 'steps': 20, 'warmup': 5}]
    assert 'launching' not in r.stderr


def test_under_a_launcher_it_does_not_launch_again():
    """the driver's own command line: torch.distributed.run owns the rendezvous, every rank measures in place"""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['ALM_BENCH_LAUNCH_CHECK'] = '1'
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1', '--master-port', '29533',
                        os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(_json_lines(r.stdout)) == 1
    assert 'launching 2 ranks myself' not in r.stderr
