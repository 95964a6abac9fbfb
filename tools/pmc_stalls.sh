#!/bin/bash
# Where do the wavefront cycles of the bench kernels go?  SQ activity / wait / FIFO-full counters, one rocprofv3 pass per group.
# usage (GPU box, repo root): tools/pmc_stalls.sh <outdir under gpurun_out>
REPO=$(pwd); OUT=$REPO/gpurun_out/${1:-stalls}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
            "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/p$i -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-power > $OUT/p$i.log 2>&1
done
cd $REPO
(for j in 1 2 3 4; do for k in closed_loop replay_tuned; do python3 tools/pmc_summary.py $OUT/p$j $k; done; done) > $OUT/stalls_summary.txt
cat $OUT/stalls_summary.txt
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
