"""The JSON line of bench.py as committed under profiles/ carries every field of the bench
contract (the driver parses it; a renamed key would go unnoticed until the end of a round)."""
import json
import os

import _cases


def test_committed_bench_line_has_the_contract_fields():
    path = os.path.join(_cases.ROOT, 'profiles', 'r02_bench.json')
    d = json.loads(open(path).read().strip().splitlines()[-1])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step',
                'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config',
                'roofline', 'cpu_baseline'):
        assert key in d, key
    assert d['metric'] == 'cluster-fits/sec' and d['unit'] == 'cluster-fits/s'
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['dtype'] == 'f64' and d['data'] == 'synthetic' and d['n_gpus'] == 1
    assert d['config']['workload'].startswith('cfg2')
    r = d['roofline']
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert key in r, key
    assert r['bound'] in ('hbm', 'mfma') and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    c = d['cpu_baseline']
    for key in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert key in c, key
    assert c['kind'] in ('reference', 'port')
    # what was timed was also checked
    assert d['status_equal_oracle'] is True and d['failed_clusters'] == 0
    assert d['parity_vs_scipy_slsqp_px']['rmse'] <= 1e-3      # north_star tolerance
    assert abs(d['value'] - d['config']['cluster_fits_per_gpu'] * 1e3 / d['ms_per_step']) < 1e-6 * d['value']


import pytest   # noqa: E402


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_rehearse_the_multi_gpu_path(tmp_path):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank),
    both ranks on the one GPU of the box, control plane over gloo: the inbox of rank 0 is mapped by
    the other process (ctr_ipc_open), every step's rows and step numbers arrive there, and the
    line carries the contract's fields for N = 2.  (Two processes time-share the GPU: the rate
    means nothing here.)"""
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(_cases.ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--single-device',
           '--frames', '32', '--steps', '6', '--warmup', '2', '--in-flight', '3', '--no-cpu-baseline']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=_cases.ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d['n_gpus'] == 2 and d['steps'] == 6 and d['scaling'] == 'weak'
    assert d['gather'] == 'step' and d['gather_transport'].startswith('ipc inbox')
    assert d['gather_checked'] is True
    assert d['failed_clusters'] == 0 and d['in_flight_results_identical'] is True
    assert abs(d['value'] - 2 * d['config']['cluster_fits_per_gpu'] * 1e3 / d['ms_per_step']) < 0.02 * d['value']
