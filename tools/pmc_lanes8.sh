# PMC counters of the (8,6) closed loop at 8 192 trials on 4 and 8 lanes per filter (tools/time_shards.py; python3 itself after `--`).
REPO=$(pwd); OUT=$REPO/gpurun_out/l8pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/p$i -- python3 $REPO/tools/time_shards.py --lanes 4,8 --trials 8192 --reps 3 > $OUT/log$i.txt 2>&1
done
cd $REPO
(for n in 1 2; do python3 tools/pmc_summary.py $OUT/p$n closed_loop; done) > $OUT/summary.txt
cat $OUT/summary.txt
rm -rf $OUT/p1 $OUT/p2
