#!/usr/bin/env python3
"""Where the microseconds of a drop-in step go (engine.FilterBank.step_host).  A real BASELINE config-2 servo trial is driven from the host (the
package's numpy plant, the reference's noise stream, commands fed back), as bench.py's `drop_in` object does; printed: the step_host call as the Experiment loop
sees it and the round-5 tensor route on the same trial; T = 1 and T = 64 copies of the trial.
usage (GPU box): python tools/time_step_route.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import uvs_amd  # noqa: E402
from uvs_amd import engine  # noqa: E402
import bench  # noqa: E402

d = bench.drop_in_step_latency(torch, uvs_amd, engine, torch.device('cuda'))
for key, v in d['detail'].items():
    print(f'{key:14s} median {v["median_us"]:6.1f} us   mean {v["mean_us"]:6.1f}   p95 {v["p95_us"]:6.1f}   (the trial converged to {v["final_feature_error_px"]:.1f} px)')
