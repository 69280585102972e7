"""Engine vs the reference's algorithm over the WHOLE bench workload (cfg 2, 256 frames, 41 033
clusters): tests/golden/cfg2_full_slsqp.npz holds oracle/ref_numpy.py's results (SciPy SLSQP with
the reference's default tol=1e-6 = run A, and converged, tol=1e-14 = run B; made by
tools/make_full_slsqp.py).  Prints one JSON object; `python tools/full_parity.py out.json` also
writes it (profiles/r03_parity_full_cfg2.json is that file, read by bench.py)."""
import json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import numpy as np

GOLD = os.path.join(ROOT, 'tests', 'golden', 'cfg2_full_slsqp.npz')


def compare(pos, cost, status, z, tag, truth=None, order=None):
    """Engine rows/clusters (pos [N,2], cost [C], status [C]) against run `tag` of the file."""
    n_per = z['n_per_cluster'].astype(int)
    off = np.concatenate([[0], np.cumsum(n_per)])
    rpos, rcost, rstat = z['pos_' + tag], z['cost_' + tag], z['status_' + tag]
    ok = (status == 0) & (rstat == 0)
    rows_ok = np.repeat(ok, n_per)
    d = pos - rpos
    dmax_c = np.maximum.reduceat(np.abs(d).max(1), off[:-1])
    rel = np.abs(cost - rcost) / np.abs(rcost)
    same = ok & (rel <= 1e-5)
    # two features that trade places at the same cost (the masks follow the labels, so this
    # happens only where the two masks hold the same pixels): compared under the best assignment
    d_lab = d.copy()
    n_swapped = 0
    import itertools
    for c in np.flatnonzero(same & (dmax_c > 1e-3) & (n_per <= 6)):
        sl = slice(off[c], off[c + 1])
        best = min(itertools.permutations(range(n_per[c])),
                   key=lambda pm: np.abs(pos[sl] - rpos[sl][list(pm)]).max())
        if list(best) != list(range(n_per[c])):
            n_swapped += 1
            d_lab[sl] = pos[sl] - rpos[sl][list(best)]
    out = {
        "clusters": int(len(cost)), "both_fit": int(ok.sum()),
        "failed_here_not_there": int(((status != 0) & (rstat == 0)).sum()),
        "failed_there_not_here": int(((status == 0) & (rstat != 0)).sum()),
        "rmse_unfiltered_px": float(np.sqrt(np.mean(d[rows_ok] ** 2))),
        "max_unfiltered_px": float(np.abs(d[rows_ok]).max()),
        "rmse_same_minimum_px": float(np.sqrt(np.mean(d[np.repeat(same, n_per)] ** 2))),
        "max_same_minimum_px": float(np.abs(d[np.repeat(same, n_per)]).max()),
        "clusters_same_cost_labels_swapped": n_swapped,
        "rmse_same_minimum_best_labels_px": float(np.sqrt(np.mean(d_lab[np.repeat(same, n_per)] ** 2))),
        "max_same_minimum_best_labels_px": float(np.abs(d_lab[np.repeat(same, n_per)]).max()),
        "clusters_cost_differs_1e-5": int((ok & ~same).sum()),
        "of_which_reference_lower": int((ok & ~same & (rcost < cost)).sum()),
        "of_which_engine_lower": int((ok & ~same & (cost < rcost)).sum()),
        "clusters_dpos_above_1e-3_px": int((ok & (dmax_c > 1e-3)).sum()),
    }
    offenders = []
    for c in np.flatnonzero(ok & (dmax_c > 1e-3)):
        sl = slice(off[c], off[c + 1])
        e = {"cluster": int(c), "features": int(n_per[c]), "max_dpos_px": float(dmax_c[c]),
             "cost_engine": float(cost[c]), "cost_reference": float(rcost[c])}
        if truth is not None:
            tr = truth[order[sl]]
            e["max_err_vs_truth_px"] = {"engine": float(np.abs(pos[sl] - tr).max()),
                                        "reference": float(np.abs(rpos[sl] - tr).max())}
        offenders.append(e)
    out["clusters_above_1e-3_px"] = offenders
    return out


def run(device=0):
    import clustertracking_amd as cta
    from clustertracking_amd import workloads, _lib
    z = np.load(GOLD)
    frames, f0, truth, opts = workloads.cfg2(int(z['n_frames']))
    prep = cta.prepare_batch(f0, cta.ArrayReader(frames), opts['diameter'])
    b = prep.batch
    assert (np.diff(b.feat_offset) == z['n_per_cluster']).all(), "cluster table differs from the file's"
    _lib.default_engine(device).refine_batch(prep.problem, b)
    res = {"workload": "cfg2: %d frames 512x512 u8, 200 features/frame" % int(z['n_frames']),
           "engine_failed_clusters": int((b.status != 0).sum()),
           "vs_reference_algorithm_default_tol_1e-6": compare(b.params_out[:, 2:4], b.cost, b.status, z, 'A', truth, prep.order),
           "vs_reference_algorithm_converged_tol_1e-14": compare(b.params_out[:, 2:4], b.cost, b.status, z, 'B', truth, prep.order),
           "reference_A_vs_B": compare(z['pos_A'], z['cost_A'], z['status_A'], z, 'B'),
           "note": "reference algorithm = oracle/ref_numpy.py (NumPy objective + SciPy SLSQP, the reference's call, "
                   "refine.py:373-375) on the same inputs; north_star tolerance 1e-3 px vs run A"}
    return res


if __name__ == '__main__':
    res = run()
    s = json.dumps(res, indent=1)
    print(s)
    if len(sys.argv) > 1:
        open(sys.argv[1], 'w').write(s + '\n')
