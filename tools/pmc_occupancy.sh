REPO=$(pwd); OUT=$REPO/gpurun_out/occlevel; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LEVEL_WAVES SQ_WAVES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p -- python3 $REPO/tools/time_methods.py --reps 2 --methods GMCKF,KF > $OUT/log.txt 2>&1
cd $REPO
python3 tools/pmc_summary.py $OUT/p closed_loop_tuned > $OUT/summary.txt
cat $OUT/summary.txt | cut -c20-
tail -3 $OUT/log.txt
rm -rf $OUT/p
