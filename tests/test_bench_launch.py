"""``python bench.py --gpus N`` as the driver types it (no torchrun around it): for N > 1 the command re-executes itself under
``torch.distributed.run`` with one rank per GPU.  Here, without a GPU: the launcher path only (``--launch-probe``) -- two ranks come up with
WORLD_SIZE=2, rendezvous over gloo on 127.0.0.1, all-reduce a one, rank 0 prints one JSON line; nothing of the benchmark runs."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + extra, env=env, capture_output=True, text=True, timeout=600)


def test_gpus_2_without_a_torchrun_environment_launches_two_ranks():
    r = _run(['--gpus', '2', '--launch-probe'])
    assert r.returncode == 0, r.stderr[-2000:]
    line = r.stdout.strip().splitlines()[-1]
    out = json.loads(line)
    assert out['launch_probe'] and out['world'] == 2 and out['ranks_seen'] == 2, out


def test_gpus_1_stays_one_process():
    r = _run(['--gpus', '1', '--launch-probe'])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out['world'] == 1 and out['backend'] is None, out
