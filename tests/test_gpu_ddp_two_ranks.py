"""SURVEY.md section 4's data-parallel invariant on the REAL backward: the gradients two ranks hold after one step on
different images equal the average of the two single-rank gradients, and both ranks take the same optimizer step.

Two child processes share the box's one GPU (RCCL refuses two ranks on one device, so the process group is `gloo`,
each bucket staged through the host -- torch's gloo path hangs on several in-flight device-tensor all-reduces between
queued kernels on this stack: scripts/gloo_cuda_probe.py, profiles/r03_gloo_cuda_probe.log): the same FlatGrads code
path -- completion-ordered flat buffer, buckets launched from inside the backward as blocks complete, wait before the
clip -- with the collective itself swapped.  The single-rank references are the
same worker run with world size 1 on each rank's images.  The 8-GPU RCCL run is the driver's."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, 'ddp_worker.py')


def _free_port():
    import socket
    with socket.socket() as s:                 # a leaked rank of an earlier run must not turn into a rendezvous hang
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _spawn(rank, world, port, out, amp=False):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONUNBUFFERED='1')
    log = open(os.path.join(str(out), f'rank{rank}.log'), 'w')
    return subprocess.Popen([sys.executable, WORKER, str(rank), str(world), str(port), str(out)] + (['bf16'] if amp else []),
                            env=env, stdout=log, stderr=subprocess.STDOUT), log


def _wait(procs, out):
    try:
        for p, _ in procs:
            p.wait(timeout=150)
    except subprocess.TimeoutExpired:
        for q, _ in procs:
            q.kill()
            q.wait()
    for r, (p, log) in enumerate(procs):
        log.close()
        text = open(log.name).read()
        assert p.returncode == 0, f'{log.name}:\n{text[-3000:]}'


@pytest.mark.parametrize('amp', [False, True])
def test_two_rank_gradients_are_the_average_of_the_single_rank_gradients(tmp_path, amp):
    """amp: the same under bf16 autocast.  The averaging itself is fp32 either way; what differs between the pair's run
    and the single-rank references is the order of the BatchNorm fp64 atomics, whose last-bit differences bf16 storage
    turns into ~1e-3 of gradient noise -- the tolerance on the average is wider there, bit-identity across the two
    ranks is not."""
    tol = 2e-2 if amp else 1e-4
    two, one = tmp_path / 'two', tmp_path / 'one'
    two.mkdir()
    one.mkdir()
    port = _free_port()
    _wait([_spawn(r, 2, port, two, amp) for r in (0, 1)], two)   # the pair, concurrently
    for r in (0, 1):                                             # each rank's images alone (world size 1)
        _wait([_spawn(r, 1, 0, one, amp)], one)
    d2 = [np.load(two / f'rank{r}.npz') for r in (0, 1)]
    d1 = [np.load(one / f'rank{r}.npz') for r in (0, 1)]
    assert int(d2[0]['buckets']) >= 5 and int(d1[0]['buckets']) == 0      # 124 MB of gradients in 4 MB buckets
    g0, g1 = d1[0]['flat'].astype(np.float64), d1[1]['flat'].astype(np.float64)
    assert np.linalg.norm(g0 - g1) > 0.05 * np.linalg.norm(g0)           # the ranks really saw different data
    for r in (0, 1):
        assert abs(float(d2[r]['loss']) - float(d1[r]['loss'])) <= (1e-3 if amp else 1e-5) * abs(float(d1[r]['loss']))   # forward is local
    mean = 0.5 * (g0 + g1)
    scale = np.linalg.norm(mean)
    assert scale > 0
    for r in (0, 1):
        assert np.linalg.norm(d2[r]['flat'] - mean) <= tol * scale, (r, np.linalg.norm(d2[r]['flat'] - mean) / scale,
                                                                      np.linalg.norm(d2[r]['flat'] - g0) / scale)
    assert np.array_equal(d2[0]['flat'], d2[1]['flat'])                   # bit-identical on both ranks
    assert float(d2[0]['total_norm']) == float(d2[1]['total_norm'])
    assert abs(float(d2[0]['total_norm']) - scale) <= tol * scale         # the clip saw the AVERAGED gradient
    for k in ('outc', 'inc'):
        assert np.array_equal(d2[0][k], d2[1][k])                         # same step on both ranks
        assert not np.array_equal(d2[0][k], d1[0][k])                     # ... and not rank 0's own
