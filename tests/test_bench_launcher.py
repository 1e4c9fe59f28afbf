"""bench.py's rank plumbing on CPU: `--gpus N` without a rank environment starts N ranks itself (fresh children via
torch.distributed.run) and relays ONE JSON line whose n_gpus is N; a rank environment that disagrees with --gpus
fails loudly.  `--backend gloo --dry` swaps the kernels for the flat-gradient all-reduce alone (the product has no
CPU path), so the launcher, RANK/WORLD_SIZE/LOCAL_RANK, the bucketed collectives and the max-over-ranks timing run
here exactly as on the GPU node."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR',
                                                            'MASTER_PORT')}
    return env


def test_gpus2_launches_two_ranks_and_reports_n_gpus_2():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--backend', 'gloo', '--dry', '--steps', '2', '--warmup',
                        '1'], capture_output=True, text=True, timeout=600, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                      # ONE line on stdout, everything else went to stderr
    out = json.loads(lines[0])
    # dry_run() itself checks that the all-reduced gradients equal mean(1..world): the ranks were distinct
    assert out['n_gpus'] == 2 and out['dry'] is True and out['config']['parallelism'] == 'dp2'
    assert out['steps'] == 2 and out['warmup'] == 1 and out['scaling'] == 'weak'


def test_rank_environment_that_disagrees_with_gpus_fails_loudly():
    env = _env()
    env.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--backend', 'gloo', '--dry'], capture_output=True,
                       text=True, timeout=300, env=env)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in r.stderr


def test_single_rank_dry_line():
    r = subprocess.run([sys.executable, BENCH, '--backend', 'gloo', '--dry', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip())['n_gpus'] == 1
