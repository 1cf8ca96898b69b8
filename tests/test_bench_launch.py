"""`bench.py --gpus N` as the driver calls it (BASELINE.json: "reported at 1/2/4/8 MI355X"): without a launcher it must start the N
ranks itself (a child torch.distributed.run, no exec), under a launcher it must refuse a world size that differs from --gpus.
Runs on CPU through --dry-run (gloo rendezvous, no GPU call)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    env['OMP_NUM_THREADS'] = '1'
    return env


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith('{')]


def test_gpus_2_without_a_launcher_starts_two_ranks():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '3', '--warmup', '1', '--dry-run'], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1                                             # ONE line, from rank 0
    assert lines[0]['n_gpus'] == 2 and lines[0]['rccl_ranks'] == 2 and lines[0]['ranks_reported'] == [0, 1]
    assert lines[0]['steps'] == 3 and lines[0]['warmup'] == 1          # the arguments reached the ranks unchanged
    assert 'torch.distributed.run' in r.stderr and '--nproc-per-node=2' in r.stderr


def test_gpus_1_is_a_single_process_and_needs_no_launcher():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '1', '--dry-run'], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]['n_gpus'] == 1 and lines[0]['ranks_reported'] == [0]
    assert 'torch.distributed.run' not in r.stderr


def test_world_size_that_differs_from_gpus_is_refused():
    env = _env()
    env.update(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--dry-run'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE' in r.stderr
    assert not _json_lines(r.stdout)
