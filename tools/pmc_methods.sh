#!/bin/bash
# SQ activity / wait counters of the closed-loop kernels of every estimator (tools/time_methods.py), optionally on an experiment build:
# usage (GPU box, repo root): [UVS_LIB_PATH=...] tools/pmc_methods.sh <outdir under gpurun_out> [methods] ["extra time_methods.py flags"]
REPO=$(pwd); OUT=$REPO/gpurun_out/${1:-methods_pmc}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/p$i -- python3 $REPO/tools/time_methods.py --reps 2 --methods ${2:-GMCKF,KF,IMCCKF,MCKF} ${3:-} > $OUT/p$i.log 2>&1
done
cd $REPO
(for j in 1 2; do python3 tools/pmc_summary.py $OUT/p$j closed_loop_tuned; done) > $OUT/summary.txt
cat $OUT/summary.txt
rm -rf $OUT/p1 $OUT/p2
