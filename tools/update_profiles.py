"""Turn what tools/profile_all.sh left under gpurun_out/v5_* into the files of profiles/
(r01_v5_*: kernel stats, bench lines, counters per launch, traffic_cfg2.json)."""
import csv, glob, json, os, shutil

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
os.chdir(ROOT)


def latest(pat):
    return sorted(glob.glob(pat), key=os.path.getmtime)[-1]


rows_out, tot = [], {}
for cname, d in (('FETCH_SIZE', 'v5_pmc_fetch'), ('WRITE_SIZE', 'v5_pmc_write')):
    rows = list(csv.DictReader(open(latest('gpurun_out/%s/*/*counter_collection.csv' % d))))
    per = {}
    for r in rows:
        per.setdefault(r['Kernel_Name'], []).append(float(r['Counter_Value']))
    for k, v in per.items():
        rows_out.append((cname, k, len(v), sum(v) / len(v)))
    tot[cname] = per
with open('profiles/r01_v5_pmc_per_launch.csv', 'w') as fo:
    fo.write('counter,kernel,launches,mean_value_KB\n')
    for o in rows_out:
        fo.write('%s,"%s",%d,%.3f\n' % o)


def step_total(counter):
    # kernels of one step with the bench's scheduling: the 8-wave block kernels only run in
    # the five "one batch alone" launches at the end
    skip = ('refine_block_kernel<2, true, 1, 8>', 'refine_block_kernel<2, true, 2, 8>')
    return sum(sum(v) / len(v) for k, v in tot[counter].items()
               if ('refine_' in k or 'front_load' in k) and not any(x in k for x in skip))


f, w = step_total('FETCH_SIZE'), step_total('WRITE_SIZE')
fm_f = [sum(v) / len(v) for k, v in tot['FETCH_SIZE'].items() if 'frame_max_kernel' in k][0]
fm_w = [sum(v) / len(v) for k, v in tot['WRITE_SIZE'].items() if 'frame_max_kernel' in k][0]
t = json.load(open('profiles/traffic_cfg2.json'))
t['frame_max_kernel'].update(FETCH_SIZE_KB=fm_f, WRITE_SIZE_KB=fm_w, bytes_corrected=(2 * fm_f + fm_w) * 1024)
t['refine_kernels'].update(FETCH_SIZE_KB=f, WRITE_SIZE_KB=w, bytes_corrected=(2 * f + w) * 1024)
t['refine_kernels']['kernels'] = ("the kernels of one step with CTR_FLAG_THROUGHPUT (bench default): front_load, "
                                  "refine_small_kernel<2,1,true,8>, <2,2,true,64> and <2,2,true,16>, "
                                  "refine_block_kernel<2,true,1,2> and <2,true,2,2>")
json.dump(t, open('profiles/traffic_cfg2.json', 'w'), indent=1)
shutil.copy(latest('gpurun_out/v5_prof/*/*kernel_stats.csv'), 'profiles/r01_v5_kernel_stats.csv')
for fn, o in (('v5_bench', 'r01_v5_bench.json'), ('v5_bench_cfg5', 'r01_v5_bench_cfg5.json'),
              ('v5_bench_cfg3', 'r01_v5_bench_cfg3.json')):
    line = open('gpurun_out/%s.json' % fn).read().strip().splitlines()[-1]
    open('profiles/' + o, 'w').write(line + '\n')
    d = json.loads(line)
    print(fn, round(d['value'] / 1e6, 3), 'M fits/s', round(d['ms_per_step'], 3), 'ms/step; failed',
          d.get('failed_clusters'), 'in flight', d.get('batches_in_flight'), d.get('in_flight_results_identical'))
print('refine kernels: FETCH %.0f KB WRITE %.0f KB -> %.1f MB per step' % (f, w, (2 * f + w) * 1024 / 1e6))
