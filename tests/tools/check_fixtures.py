"""Run every golden fixture through a backend and print the parity table.

    python tests/tools/check_fixtures.py engine     # HIP engine (needs the MI355X)
    python tests/tools/check_fixtures.py oracle     # C oracle (CPU)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import _cases  # noqa: E402


def main():
    backend = sys.argv[1] if len(sys.argv) > 1 else 'engine'
    names = sys.argv[2:] or _cases.case_names()
    orc = _cases.oracle_runner()
    for name in names:
        case = _cases.Case(name)
        res = case.run(None if backend == 'engine' else orc)
        line = '%-26s nan=%d' % (name, int(np.isnan(res['cost']).sum()))
        if backend == 'engine':
            ro = case.run(orc)
            rm, mx, oa, ob = _cases.compare(res, ro, case.pos_columns)
            line += ' | vs C-oracle rmse %.1e max %.1e status-eq %s cost-d %.1e' % (
                rm, mx, bool((oa == ob).all()),
                np.nanmax(np.abs(res['cost'].values - ro['cost'].values)) if oa.any() else 0)
        if not case.ref_aborts:
            for which in 'BA':
                rm, mx, oa, ob = _cases.compare(res, case.ref(which), case.pos_columns)
                line += ' | vs%s rmse %.1e max %.1e nan-eq %s' % (which, rm, mx, bool((oa == ob).all()))
        print(line, flush=True)


if __name__ == '__main__':
    main()
