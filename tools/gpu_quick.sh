#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_quick.sh "<lanes list>" [pytest -k filter]
# Runs the closed-loop parity tests, then times bench.py for each lanes-per-filter value.
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q -x -k "${2:-closed_loop or replay or experiment or math}" 2>&1 | tail -4)
for L in $1; do
  (timeout 300 python bench.py --steps 5 --warmup 1 --lanes $L --no-cpu-baseline) 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('L', d['config']['lanes_per_filter'], 'ms %.3f' % d['roofline']['avg_kernel_ms'], 'Gupd/s %.3f' % (d['value']/1e9), 'GB/s %.0f' % d['roofline']['achieved'], 'frac %.3f' % d['roofline']['frac'])"
done
