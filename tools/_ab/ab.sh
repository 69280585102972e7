for nfl in 8 12; do for sh in 0 1 2 3 4 5 6 7; do GPU_MAX_HW_QUEUES=20 python bench.py --steps 48 --warmup 12 --shard $sh --in-flight $nfl --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in flight $nfl shard $sh', round(d['value']/1e6,2), round(d['ms_per_step'],3), d['in_flight_results_identical'])"; done; done
