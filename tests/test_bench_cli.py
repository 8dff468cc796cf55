"""bench.py's launch contract: `--gpus N` starts N ranks itself (one process per GPU) before touching the GPU; under an existing
torch.distributed.run job it joins it; a `--gpus` that disagrees with WORLD_SIZE is an error, never a silent one-GPU line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_must_match_world_size():
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and 'WORLD_SIZE=1' in r.stderr and not r.stdout.strip()


def test_launcher_builds_one_rank_per_gpu(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}
    monkeypatch.setattr(bench.subprocess, 'call', lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '3'])
    assert bench.launch_ranks(4) == 0
    cmd = seen['cmd']
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node=4' in cmd and '127.0.0.1' in cmd
    assert cmd[-4:] == ['--gpus', '4', '--steps', '3'] and seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    a = bench.parse([])
    assert (a.gpus, a.scaling, a.candidates, a.workload, a.cpu_sample) == (1, 'strong', 64, 'adm64_eps_greedy', 64)
    assert a.dtype == 'f16x3'            # the metric of record is measured in the mode that reproduces the reference's selections


@pytest.mark.gpu
def test_two_ranks_started_by_the_bench_itself_shard_the_64_candidates():
    """`python bench.py --gpus 2` on the one-GPU test box (DTS_DIST_BACKEND=gloo: the two ranks share the card and stage the
    reward all-gather through the host; on a multi-GPU node the same command runs over RCCL): n_gpus = 2 and the 64 candidates of
    BASELINE config 3 are split 32 + 32, not 64 per GPU."""
    env = dict(os.environ, DTS_DIST_BACKEND='gloo')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--no-kernel-timing'],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(line) == 1, r.stdout
    j = json.loads(line[0])
    assert j['n_gpus'] == 2 and j['scaling'] == 'strong' and j['dtype'] == 'f16x3' and j['config']['parity_grade'] is True
    assert j['config']['candidates_total'] == 64 and j['config']['candidates_per_gpu'] == 32
    assert j['weak_value'] is not None and j['value'] > 0
