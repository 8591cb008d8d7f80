"""`python bench.py --gpus N` must start N ranks itself (the reference's
distributed script spawns its own, experiments/cora_benchmark_graphsaint_distributed.py:130-142).
CPU rehearsal of the launch path (rendezvous + one gloo all-reduce, nothing measured) here; the
real two-rank step on a GPU is tests/test_gpu_parity.py::test_bench_launches_two_ranks."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(args, extra_env):
    env = dict(os.environ, **extra_env)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, timeout=300,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    return r


def test_gpus_flag_launches_ranks():
    r = _run(['--gpus', '2', '--workload', 'tiny'], {'AMPCONV_BENCH_LAUNCH_ONLY': '1'})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['ranks_seen'] == 2


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, AMPCONV_BENCH_LAUNCH_ONLY='1', WORLD_SIZE='1', RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env, timeout=120,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode != 0
    assert 'WORLD_SIZE' in (r.stderr + r.stdout)


def test_failed_rank_fails_the_launch():
    # no GPU here and no rehearsal switch: every rank stops at the GPU check, the parent must say so
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip('needs a machine without a GPU')
    r = _run(['--gpus', '2', '--workload', 'tiny'], {})
    assert r.returncode != 0
    assert 'ranks failed' in r.stderr


def test_one_rank_dying_after_the_rendezvous_stops_the_others():
    # rank 1 exits right after init_process_group; rank 0 then sits in an all-reduce that can never
    # complete.  The launcher must notice the first failure, stop the survivor and name the failed rank -- fast.
    import time
    t0 = time.time()
    r = _run(['--gpus', '2', '--workload', 'tiny'], {'AMPCONV_BENCH_LAUNCH_ONLY': '1', 'AMPCONV_BENCH_FAIL_RANK': '1'})
    assert r.returncode != 0
    assert 'ranks failed' in r.stderr and '(1, 3)' in r.stderr and 'first: rank 1' in r.stderr, r.stderr[-2000:]
    assert time.time() - t0 < 120
