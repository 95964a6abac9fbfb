#!/bin/bash
# Collect the rocprofv3 evidence for profiles/<round>/ on the GPU box.  usage: tools/profile_round.sh <outdir under gpurun_out>
# Kernel-trace statistics and PMC counters are separate passes (never --pmc together with other trace domains).
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/${1:-prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/stats_bench.log 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$tag -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_$tag.log 2>&1
done
cd $REPO
f=$(ls $OUT/stats/*/*kernel_stats.csv | head -1)
cp $f $OUT/kernel_stats.csv
(for tag in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do python3 tools/pmc_summary.py $OUT/pmc_$tag closed_loop; python3 tools/pmc_summary.py $OUT/pmc_$tag replay_tuned; python3 tools/pmc_summary.py $OUT/pmc_$tag replay_rows; python3 tools/pmc_summary.py $OUT/pmc_$tag noise_kernel; done) > $OUT/pmc_summary.txt
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
head -8 $OUT/kernel_stats.csv | cut -c1-200
cat $OUT/pmc_summary.txt
tail -c 600 $OUT/bench_line.json
# keep only the summaries in the merged output
rm -rf $OUT/stats $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_INSTS_VALU
