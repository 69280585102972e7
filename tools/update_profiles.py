"""Turn what tools/profile_r3.sh left under gpurun_out/<round>/ into the files of profiles/
(<round>_*: kernel stats with 8 batches in flight and with one, bench lines, counters per launch,
traffic_cfg2.json, valu_cfg2.json).

    python tools/update_profiles.py r02
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
os.chdir(ROOT)
RND = sys.argv[1] if len(sys.argv) > 1 else 'r03'
SRC = 'gpurun_out/%s' % RND


def latest(pat):
    return sorted(glob.glob(pat), key=os.path.getmtime)[-1]


def per_kernel(counter_dir):
    """{counter: {kernel: [values per launch]}} of one rocprofv3 --pmc pass"""
    out = {}
    for r in csv.DictReader(open(latest('%s/%s/*/*counter_collection.csv' % (SRC, counter_dir)))):
        out.setdefault(r['Counter_Name'], {}).setdefault(r['Kernel_Name'], []).append(float(r['Counter_Value']))
    return out


# ---- HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes, 8 batches in flight, 1 step) -------
tot = {}
rows_out = []
for cname, d in (('FETCH_SIZE', 'pmc_fetch'), ('WRITE_SIZE', 'pmc_write')):
    per = per_kernel(d)[cname]
    tot[cname] = per
    for k, v in per.items():
        rows_out.append((cname, k, len(v), sum(v) / len(v)))
with open('profiles/%s_pmc_per_launch.csv' % RND, 'w') as fo:
    fo.write('counter,kernel,launches,mean_value_KB\n')
    for o in rows_out:
        fo.write('%s,"%s",%d,%.3f\n' % o)


def step_total(counter):
    # kernels of one step with the bench's scheduling: the 8-wave block kernels only run in
    # the five "one batch alone" launches at the end
    skip = ('refine_block_kernel<2, true, 1, 8', 'refine_block_kernel<2, true, 2, 8')
    return sum(sum(v) / len(v) for k, v in tot[counter].items()
               if ('refine_' in k or 'front_load' in k) and not any(x in k for x in skip))


f, w = step_total('FETCH_SIZE'), step_total('WRITE_SIZE')
fm_f = [sum(v) / len(v) for k, v in tot['FETCH_SIZE'].items() if 'frame_max_kernel' in k][0]
fm_w = [sum(v) / len(v) for k, v in tot['WRITE_SIZE'].items() if 'frame_max_kernel' in k][0]
t = json.load(open('profiles/traffic_cfg2.json'))
t['profile'] = 'profiles/%s_pmc_per_launch.csv' % RND
t['source'] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (profiles/%s_pmc_per_launch.csv), "
               "per launch; tools/profile_r3.sh" % RND)
t['frame_max_kernel'].update(FETCH_SIZE_KB=fm_f, WRITE_SIZE_KB=fm_w, bytes_corrected=(2 * fm_f + fm_w) * 1024)
t['refine_kernels'].update(FETCH_SIZE_KB=f, WRITE_SIZE_KB=w, bytes_corrected=(2 * f + w) * 1024,
                           correction="x2 applied to FETCH_SIZE: the counter reports one 64-B unit per 128-B line filled, for "
                                      "streams and for 1-byte gathers alike (profiles/r03_fetch_calibration.json, tools/fetch_calib.hip); "
                                      "lines are filled once per XCD that touches them, so this is a count of line fills, not a bound")
json.dump(t, open('profiles/traffic_cfg2.json', 'w'), indent=1)

# ---- SQ / GRBM counters, one batch at a time (five passes) -----------------------------------------
sq = {}
for i in range(1, 7):
    if not glob.glob('%s/pmc_sq%d/*/*counter_collection.csv' % (SRC, i)):
        continue
    for cname, per in per_kernel('pmc_sq%d' % i).items():
        sq[cname] = {k: sum(v) / len(v) for k, v in per.items() if 'refine_' in k or 'frame_max_kernel' in k}
kernels = sorted({k for per in sq.values() for k in per})
with open('profiles/%s_pmc_sq_per_launch.csv' % RND, 'w') as fo:
    fo.write('Counter_Name,' + ','.join('"%s"' % k for k in kernels) + '\n')
    for cname in sorted(sq):
        fo.write(cname + ',' + ','.join('%.1f' % sq[cname].get(k, 0.) for k in kernels) + '\n')

bench1 = json.loads(open('%s/bench_inflight1.json' % SRC).read().strip().splitlines()[-1])
refine = [k for k in kernels if 'refine_' in k]
valu_insts = sum(sq['SQ_INSTS_VALU'].get(k, 0.) for k in refine)
active = sum(sq['SQ_ACTIVE_INST_VALU'].get(k, 0.) for k in refine)      # quad-cycles, summed over SIMDs
mfma_busy = sum(sq['SQ_VALU_MFMA_BUSY_CYCLES'].get(k, 0.) for k in refine)
step_s = bench1['roofline']['kernel_ms_one_batch_alone'] * 1e-3
clock = 2.4e9
simds = 256 * 4
valu = {
    "workload": "cfg2, 256 frames, one batch at a time (bench.py --in-flight 1), default scheduling",
    "source": "profiles/%s_pmc_sq_per_launch.csv (rocprofv3 --pmc, five passes; tools/profile_r3.sh)" % RND,
    "valu_wave_instructions_per_launch": valu_insts,
    "sq_active_inst_valu_quad_cycles": active,
    "refine_stage_s_one_batch_alone": step_s,
    "valu_busy_frac": active * 4 / (simds * step_s * clock),
    "valu_issue_bound_s": valu_insts * 4 / simds / clock,
    "mfma_busy_frac": mfma_busy / (simds * step_s * clock),
    "note": "valu_busy_frac = SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x refine-stage time x 2.4 GHz); "
            "valu_issue_bound_s = the time the VALU wave-instructions of one step need at one per 4 cycles "
            "on every SIMD (f64 FMA issues at half that rate)",
}
json.dump(valu, open('profiles/valu_cfg2.json', 'w'), indent=1)

# ---- kernel stats and bench lines ---------------------------------------------------------------------
shutil.copy(latest('%s/prof/*/*kernel_stats.csv' % SRC), 'profiles/%s_kernel_stats.csv' % RND)
shutil.copy(latest('%s/prof_inflight1/*/*kernel_stats.csv' % SRC), 'profiles/%s_kernel_stats_inflight1.csv' % RND)
for fn, o in (('bench', '%s_bench.json' % RND), ('bench_inflight1', '%s_bench_inflight1.json' % RND),
              ('bench_cfg5', '%s_bench_cfg5.json' % RND), ('bench_cfg3', '%s_bench_cfg3.json' % RND),
              ('bench_cfg3_inflight1', '%s_bench_cfg3_inflight1.json' % RND),
              ('bench_cfg3_192', '%s_bench_cfg3_192stacks.json' % RND)):
    path = '%s/%s.json' % (SRC, fn)
    if not os.path.exists(path):
        continue
    line = open(path).read().strip().splitlines()[-1]
    open('profiles/' + o, 'w').write(line + '\n')
    d = json.loads(line)
    print(fn, round(d['value'] / 1e6, 3), 'M fits/s', round(d['ms_per_step'], 3), 'ms/step; failed',
          d.get('failed_clusters'), 'in flight', d.get('batches_in_flight'), d.get('in_flight_results_identical'))
print('refine kernels: FETCH %.0f KB WRITE %.0f KB -> %.1f MB per step' % (f, w, (2 * f + w) * 1024 / 1e6))
print('VALU busy %.3f, issue bound %.3f ms of %.3f ms' % (valu['valu_busy_frac'], valu['valu_issue_bound_s'] * 1e3, step_s * 1e3))

# ---- cfg 3 at its stated density: the large-cluster kernel, one batch at a time ---------------------
if glob.glob('%s/cfg3_prof/*/*kernel_stats.csv' % SRC):
    shutil.copy(latest('%s/cfg3_prof/*/*kernel_stats.csv' % SRC), 'profiles/%s_cfg3_kernel_stats.csv' % RND)
    with open('profiles/%s_cfg3_pmc_per_launch.csv' % RND, 'w') as fo:
        fo.write('counter,kernel,launches,mean_value\n')
        for d in ('cfg3_pmc_fetch', 'cfg3_pmc_write', 'cfg3_pmc_sq1', 'cfg3_pmc_sq2', 'cfg3_pmc_wait'):
            if not glob.glob('%s/%s/*/*counter_collection.csv' % (SRC, d)):
                continue
            for cname, per in per_kernel(d).items():
                for k, v in per.items():
                    if 'refine_' in k or 'frame_max' in k:
                        fo.write('%s,"%s",%d,%.3f\n' % (cname, k, len(v), sum(v) / len(v)))
    # HBM traffic of the large-cluster kernel per launch (64 stacks, one batch at a time), for bench.py's cfg 3 line
    try:
        pf3, pw3 = per_kernel('cfg3_pmc_fetch')['FETCH_SIZE'], per_kernel('cfg3_pmc_write')['WRITE_SIZE']
        kf = [k for k in pf3 if 'refine_large_kernel' in k][0]
        f3, w3 = sum(pf3[kf]) / len(pf3[kf]), sum(pw3[kf]) / len(pw3[kf])
        json.dump({"workload": "cfg3, 64 stacks of 64x128x128 uint8, 500 features per stack, one batch at a time",
                   "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (profiles/%s_cfg3_pmc_per_launch.csv), per launch; tools/profile_r3.sh" % RND,
                   "refine_large_kernel": {"FETCH_SIZE_KB": f3, "WRITE_SIZE_KB": w3, "bytes_corrected": (2 * f3 + w3) * 1024,
                                           "correction": "x2 applied to FETCH_SIZE (one 64-B unit per 128-B line filled, profiles/r03_fetch_calibration.json)"},
                   "profile": "profiles/%s_cfg3_pmc_per_launch.csv" % RND}, open('profiles/traffic_cfg3.json', 'w'), indent=1)
    except (KeyError, IndexError) as e:
        print('no cfg3 traffic:', e)
    print('cfg3 profiles written')
# ---- what FETCH_SIZE counts (tools/fetch_calib.hip) ---------------------------------------------------
if glob.glob('%s/fetch_calib/*/*counter_collection.csv' % SRC):
    per = per_kernel('fetch_calib')['FETCH_SIZE']
    n = 64 << 20
    cal = {"buffer_bytes": n, "lines_128B": n // 128, "source": "tools/fetch_calib.hip under rocprofv3 --pmc FETCH_SIZE (tools/profile_r3.sh)"}
    for k, v in per.items():
        name = k.split('(')[0]
        cal[name] = {"FETCH_SIZE_KB_per_launch": sum(v) / len(v), "launches": len(v)}
    s16 = [v for k, v in cal.items() if k.startswith('stream16')][0]['FETCH_SIZE_KB_per_launch'] * 1024
    l1 = [v for k, v in cal.items() if k.startswith('line1')][0]['FETCH_SIZE_KB_per_launch'] * 1024
    w13 = [v for k, v in cal.items() if k.startswith('window13')][0]['FETCH_SIZE_KB_per_launch'] * 1024
    frames = n // (512 * 512)
    cal["reported_over_bytes_touched_stream16"] = s16 / n
    cal["reported_bytes_per_line_line1"] = l1 / (n // 128)
    cal["reported_bytes_per_distinct_line_window13"] = w13 / (frames * 31 * 13 * 4)
    json.dump(cal, open('profiles/%s_fetch_calibration.json' % RND, 'w'), indent=1)
    print('FETCH_SIZE calibration:', {k: v for k, v in cal.items() if k.startswith('reported')})

