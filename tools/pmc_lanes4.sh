REPO=$(pwd); OUT=$REPO/gpurun_out/l4pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/p -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-power --no-replay --no-side --lanes 4 > $OUT/log.txt 2>&1
cd $REPO
python3 tools/pmc_summary.py $OUT/p closed_loop > $OUT/summary.txt
cat $OUT/summary.txt
tail -c 600 $OUT/log.txt
rm -rf $OUT/p
