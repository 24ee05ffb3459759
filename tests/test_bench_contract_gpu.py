"""GPU: bench.py's contract with the driver -- one JSON line with the agreed keys, at N = 1 directly and under torchrun with one rank (RCCL
initialised, barriers, the optional gather) -- on a small batch so that the test takes seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline'}


def run(cmd):
    res = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [line for line in res.stdout.splitlines() if line.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def check(line, steps, warmup):
    assert KEYS <= set(line), KEYS - set(line)
    assert line['steps'] == steps and line['warmup'] == warmup and line['n_gpus'] == 1 and line['higher_is_better'] is True
    assert line['dtype'] == 'f64' and line['data'] == 'synthetic' and line['vs_baseline'] is None and 'workload' in line['config']
    assert line['value'] > 0 and line['ms_per_step'] > 0
    roof = line['roofline']
    assert {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'} <= set(roof) and roof['bound'] == 'hbm' and roof['peak'] == 8000.
    assert abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-12
    assert line['parity_spot_check_tilted_err'] < 1e-13


def test_single_process_line():
    line = run([sys.executable, 'bench.py', '--rows', '4000', '--steps', '3', '--warmup', '1', '--ramp-ms', '20', '--no-cpu-baseline', '--no-secondary'])
    check(line, 3, 1)
    assert line['scaling'] == 'weak' and line['config']['rccl_ranks'] == 0 and line['value_api'] > 0


def test_one_rank_under_torchrun():
    base = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1']
    line = run(base + ['--master-port', '29541', 'bench.py', '--gpus', '1', '--rows', '4000', '--steps', '3', '--warmup', '1', '--ramp-ms', '20',
                       '--no-cpu-baseline', '--no-secondary', '--gather'])
    check(line, 3, 1)
    assert line['config']['rccl_ranks'] == 1 and line['gather_ms'] > 0 and line['value_with_gather'] > 0
    split = run(base + ['--master-port', '29542', 'bench.py', '--gpus', '1', '--config', '5', '--rows', '200000', '--steps', '2', '--warmup', '1', '--gather'])
    assert split['scaling'] == 'strong' and split['config']['rccl_ranks'] == 1 and split['value'] > 0 and split['unit'] == 'samples/s'
    assert split['parity_spot_check']['max_rel_err'] < 1e-10 and split['ms_per_step_rank_min'] <= split['ms_per_step_rank_max'] <= split['ms_per_step'] * 1.01
    # config 4 under the launcher: both filters on the rank's block of cosmologies, each with its oracle spot check on the line
    filt = run(base + ['--master-port', '29543', 'bench.py', '--gpus', '1', '--config', '4', '--rows', '3000', '--steps', '2', '--warmup', '1', '--gather'])
    assert filt['scaling'] == 'strong' and filt['config']['rccl_ranks'] == 1 and filt['unit'] == 'filtered vectors/s' and filt['value'] > 0
    assert filt['config']['per_gpu'] == 3000 and len(filt['ms_per_step_by_rank']) == 1 and filt['gather_ms'] > 0
    for engine in ('wallish2018', 'brieden2022'):
        assert filt['parity_spot_check'][engine]['max_rel_err'] < 1e-9
    assert line['ms_per_step_rank_min'] <= line['ms_per_step_rank_max'] and len(line['ms_per_step_by_rank']) == 1
    # what a multi-GPU run prints by default (here forced on one rank): the gather timed, the strong splits of configs 4 and 5 on the same line
    multi = run(base + ['--master-port', '29544', 'bench.py', '--gpus', '1', '--rows', '3000', '--steps', '2', '--warmup', '1', '--ramp-ms', '20',
                        '--no-cpu-baseline', '--no-secondary', '--split-configs', '--gather'])
    check(multi, 2, 1)
    assert multi['gather_ms'] > 0 and set(multi['split_configs']) == {'config4', 'config5'}
    for name, unit in (('config4', 'filtered vectors/s'), ('config5', 'samples/s')):
        sub = multi['split_configs'][name]
        assert sub['unit'] == unit and sub['value'] > 0 and sub['scaling'] == 'strong' and sub['config']['rccl_ranks'] == 1 and sub['value_with_gather'] > 0


def test_self_launch():
    """`python bench.py --gpus N` without a launcher starts its own ranks: with --launcher also for N = 1 (RCCL then initialised); asking for more
    GPUs than the node has, or a launcher whose world size is not --gpus, exits non-zero instead of printing a line."""
    import torch
    small = ['--rows', '4000', '--steps', '3', '--warmup', '1', '--ramp-ms', '20', '--no-cpu-baseline', '--no-secondary']
    line = run([sys.executable, 'bench.py', '--gpus', '1', '--launcher'] + small)
    check(line, 3, 1)
    assert line['config']['rccl_ranks'] == 1
    too_many = torch.cuda.device_count() + 1
    res = subprocess.run([sys.executable, 'bench.py', '--gpus', str(too_many)] + small, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         universal_newlines=True, timeout=600)
    assert res.returncode != 0 and not any(ln.startswith('{') for ln in res.stdout.splitlines()) and 'GPU(s) visible' in res.stderr
    env = dict(os.environ, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    res = subprocess.run([sys.executable, 'bench.py', '--gpus', '2'] + small, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         universal_newlines=True, timeout=600)
    assert res.returncode != 0 and not any(ln.startswith('{') for ln in res.stdout.splitlines()) and 'WORLD_SIZE' in res.stderr
