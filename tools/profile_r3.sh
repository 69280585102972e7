# Collects what profiles/ holds for this round's build (run on the GPU box:
#   gpurun --timeout 1200 -- bash tools/profile_r3.sh ; then python tools/update_profiles.py r03 here).
# The bench keeps 10 batches in flight and needs more hardware queues than ROCm's default of 4;
# under rocprofv3 the HIP runtime is initialised before bench.py can set the variable itself,
# so it is exported HERE, for every run alike.  `python bench.py` comes directly after `--`.
set -e
export GPU_MAX_HW_QUEUES=20
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench.json 2> $O/bench.err
echo bench done
python $R/bench.py --in-flight 1 --no-cpu-baseline > $O/bench_inflight1.json 2>> $O/bench.err
echo bench in-flight 1 done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --no-cpu-baseline > $O/prof.log 2>&1
echo stats done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_inflight1 -- python $R/bench.py --in-flight 1 --no-cpu-baseline > $O/prof_inflight1.log 2>&1
echo stats in-flight 1 done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_write.log 2>&1
echo write done
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_sq$i -- python $R/bench.py --steps 2 --warmup 1 --in-flight 1 --no-cpu-baseline > $O/pmc_sq$i.log 2>&1
  echo sq pass $i done
done
python $R/bench.py --workload cfg5 --no-cpu-baseline > $O/bench_cfg5.json 2>> $O/bench.err
echo cfg5 done
# cfg 3 at its stated density (64 stacks of 500 features: the large-cluster kernel), ten batches in
# flight and one; kernel stats and counters with one batch at a time (a step is seconds)
python $R/bench.py --workload cfg3 --frames 64 --steps 20 --warmup 5 > $O/bench_cfg3.json 2>> $O/bench.err
python $R/bench.py --workload cfg3 --frames 64 --steps 4 --warmup 1 --in-flight 1 --no-cpu-baseline > $O/bench_cfg3_inflight1.json 2>> $O/bench.err
# (the clusters of a batch last 0.1 - 3 s: with 64 stacks per batch and ten batches in flight compute units wait
#  for the batches' stragglers; 192 stacks per batch keep them busy -- more batches in flight would do the same
#  but oversubscribe the hardware queues)
python $R/bench.py --workload cfg3 --frames 192 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg3_192.json 2>> $O/bench.err
echo cfg3 bench done
C3="--workload cfg3 --frames 64 --steps 2 --warmup 1 --in-flight 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg3_prof -- python $R/bench.py $C3 > $O/cfg3_prof.log 2>&1
echo cfg3 stats done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cfg3_pmc_fetch -- python $R/bench.py $C3 > $O/cfg3_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cfg3_pmc_write -- python $R/bench.py $C3 > $O/cfg3_pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/cfg3_pmc_sq1 -- python $R/bench.py $C3 > $O/cfg3_pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/cfg3_pmc_sq2 -- python $R/bench.py $C3 > $O/cfg3_pmc_sq2.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/cfg3_pmc_wait -- python $R/bench.py $C3 > $O/cfg3_pmc_wait.log 2>&1
echo cfg3 counters done
# what FETCH_SIZE counts for the access shapes of the engine (tools/fetch_calib.hip)
hipcc --offload-arch=gfx950 -O2 $R/tools/fetch_calib.hip -o /tmp/fetch_calib > $O/fetch_calib_build.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_calib -- /tmp/fetch_calib > $O/fetch_calib.log 2>&1
echo all done
