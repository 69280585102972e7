"""The JSON line of bench.py as committed under profiles/ carries every field of the bench
contract (the driver parses it; a renamed key would go unnoticed until the end of a round)."""
import json
import os

import _cases


def test_committed_bench_line_has_the_contract_fields():
    path = os.path.join(_cases.ROOT, 'profiles', 'r01_v5_bench.json')
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
