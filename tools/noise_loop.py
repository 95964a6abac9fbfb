"""Keeps the alpha-stable noise generator running back to back (tools/power_probe.sh samples clock and power meanwhile)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import uvs_amd
from uvs_amd import noise_device as nd, engine
T = 65536
seeds = torch.arange(123456, 123456 + T, dtype=torch.int64, device='cuda')
out = engine.alloc_stream(T, 299, 8, 'kct', 'cuda')
open('gpurun_out/.probe_started', 'w').write('1')
for i in range(40000):
    nd.generate(uvs_amd.NoiseType.ALPHA_STABLE, dict(alpha=1.5, beta=0, gamma=1, delta=0), seeds, 8, 299, out=out)
    if i % 200 == 199:
        torch.cuda.synchronize()
