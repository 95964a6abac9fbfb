#!/bin/bash
# Collect the rocprofv3 evidence for profiles/<round>/ on the GPU box.  usage: tools/profile_round.sh <outdir under gpurun_out>
# Kernel-trace statistics and PMC counters are separate passes (never --pmc together with other trace domains); the program after
# `--` is python3 itself.  The traced command is the driver's: bench.py --steps 20 --warmup 5 (minus the host CPU baseline).
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/${1:-prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# which kernels these counts belong to: the library's version string carries the fingerprint of its kernel sources (csrc/src_hash.py)
python3 -c "import sys; sys.path.insert(0, '$REPO'); import uvs_amd; print(uvs_amd.lib().uvs_version().decode())" > $OUT/library_version.txt
# the un-profiled line first: a profiled run clocks 2-3 % lower (MI355X_MICROARCH.md, DVFS give-back)
UVS_BENCH_FULL_JSON=$OUT/bench_full.json python3 $REPO/bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_line.err
echo "bench line done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-power > $OUT/stats_bench.json 2> $OUT/stats_bench.err
echo "kernel trace done"
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
            "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$i -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-power > $OUT/pmc_$i.log 2>&1
  echo "pmc pass $i done"
done
cd $REPO
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_all_dispatches.csv
# headline kernel: 5 warm-up dispatches, then the 20 timed ones; what follows at the same grid belongs to the side objects (other estimators use other instantiations; the end-to-end sweep re-launches this one)
python3 tools/trace_summary.py $OUT/stats --skip "closed_loop_tuned_kernel<8, 6, 2, 5, 2, 2, true, false, false, false, false, false> grid=131072:5:20" closed_loop_wide_kernel:2 replay_tuned_kernel:1 replay_rows_kernel:1 replay_f32_kernel:1 > $OUT/kernel_trace_summary.csv
(for n in 1 2 3 4 5 6; do for k in closed_loop_tuned closed_loop_wide replay_tuned replay_rows replay_f32 noise_kernel; do python3 tools/pmc_summary.py --halves '8, 6, 2, 3, ' $OUT/pmc_$n $k; done; done) > $OUT/pmc_summary.txt
cat $OUT/kernel_trace_summary.csv
tail -c 1500 $OUT/bench_line.json
head -2 $(ls $OUT/stats/*/*kernel_trace.csv | head -1) | cut -c1-600
head -2 $(ls $OUT/pmc_1/*/*counter_collection.csv | head -1) | cut -c1-600
# keep only the summaries in the merged output
rm -rf $OUT/stats $OUT/pmc_1 $OUT/pmc_2 $OUT/pmc_3 $OUT/pmc_4 $OUT/pmc_5 $OUT/pmc_6
