# Collects what profiles/ holds for a build (run on the GPU box: gpurun -- bash tools/profile_all.sh):
# bench line, rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE passes, cfg5 and cfg3 lines -> gpurun_out/v5_*
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $R/gpurun_out/v5_bench.json 2> $R/gpurun_out/v5_bench.err
echo bench done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/v5_prof -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/v5_prof.log 2>&1
echo stats done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/v5_pmc_fetch -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/v5_pmc_fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/v5_pmc_write -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/v5_pmc_write.log 2>&1
echo write done
python $R/bench.py --workload cfg5 --no-cpu-baseline > $R/gpurun_out/v5_bench_cfg5.json 2>/dev/null
python $R/bench.py --workload cfg3 --frames 64 --features 40 --no-cpu-baseline > $R/gpurun_out/v5_bench_cfg3.json 2>/dev/null
echo all done
