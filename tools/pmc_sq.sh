# SQ / GRBM counters per kernel for one batch at a time (run on the GPU box): passes of a few counters each
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/v5_pmc_sq$i -- python $R/bench.py --steps 2 --warmup 1 --in-flight 1 --no-cpu-baseline > $R/gpurun_out/v5_pmc_sq$i.log 2>&1
  echo pass $i done
done
