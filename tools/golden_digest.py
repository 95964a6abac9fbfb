"""Bit-level digests of closed-loop outputs: a REGRESSION pin of the library against itself, not parity with the reference.

A kernel rewrite that must leave every output bit where it was (the MCKF iterating branch) is checked against digests taken
from the library BEFORE the rewrite: `python tools/golden_digest.py --write` on a GPU box writes tests/golden/digests.json,
tests/test_gpu_digest.py recomputes them.  A digest is the wrap-around int64 sum of (bit pattern x position-dependent odd weight) over
the LOGGED rows of a stream (rows at and after a FAILed trial's k_done are unspecified and carry weight 0), computed on the device.
Inputs (noise, q_start) are digested too, so a changed generator shows up as such and not as a changed estimator."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PATH = os.path.join(ROOT, 'tests', 'golden', 'digests.json')

# (name, method, alpha, trials, lanes, forced segments (bits 8-15 of fp.reserved), fpi_threshold)
CASES = [
    ('mckf_a1p0_t4099', 'MCKF', 1.0, 4099, 0, 0, 0.1),
    ('mckf_a1p0_t4099_thr0p02', 'MCKF', 1.0, 4099, 0, 0, 0.02),
    ('mckf_a1p2_t4099_lanes2', 'MCKF', 1.2, 4099, 2, 0, 0.1),
    ('mckf_a1p0_t40000_seg', 'MCKF', 1.0, 40000, 0, 0, 0.1),
    ('mckf_a1p5_t65536', 'MCKF', 1.5, 65536, 0, 0, 0.1),
    ('mckf_a1p0_t65536', 'MCKF', 1.0, 65536, 0, 0, 0.1),
    ('gmckf_a1p5_t4099', 'GMCKF', 1.5, 4099, 0, 0, 0.1),
    ('gmckf_a1p5_t65536', 'GMCKF', 1.5, 65536, 0, 0, 0.1),
    ('kf_a1p5_t4099', 'KF', 1.5, 4099, 0, 0, 0.1),
    ('imcckf_a1p5_t4099', 'IMCCKF', 1.5, 4099, 0, 0, 0.1),
]


def digest(torch, tensor, k_done=None, step_dim=None, trial_dim=None):
    """Wrap-around int64 checksum of a tensor's bit patterns; rows with step >= k_done[trial] count as zero."""
    t = tensor.contiguous()
    bits = t.view(torch.int64) if t.dtype == torch.float64 else t.to(torch.int64)
    n = bits.numel()
    w = (torch.arange(n, dtype=torch.int64, device=t.device) * -7046029254386353131 + 1442695040888963407) | 1
    w = w.view(bits.shape)
    if k_done is not None:
        shape_k = [1] * bits.dim()
        shape_k[step_dim] = bits.shape[step_dim]
        shape_t = [1] * bits.dim()
        shape_t[trial_dim] = bits.shape[trial_dim]
        keep = torch.arange(bits.shape[step_dim], device=t.device).view(shape_k) < k_done.to(torch.int64).view(shape_t)
        w = w * keep
    return int((bits * w).sum().item())


def run_case(uvs, torch, bench, case):
    name, method, alpha, T, lanes, seg, thr = case
    cfg = bench.config2()
    cfg['experiments']['epoch'] = T
    cfg['estimator']['method'] = method
    cfg['estimator']['estimator_params']['fpi_threshold'] = thr
    cfg['noise']['noise_params']['alpha'] = alpha
    res = uvs.batch.run_batch(cfg, cells=[alpha], epoch=T, want=('x', 'err', 'q'), lanes=lanes)
    torch.cuda.synchronize()
    kd = res.k_done
    out = {'noise': digest(torch, res.noise), 'k_done': digest(torch, kd), 'status': digest(torch, res.status),
           'failed': int((res.status != 0).sum().item()), 'updates': int(kd.sum().item())}
    for key in ('x', 'err', 'q'):
        out[key] = digest(torch, res.streams[key], kd, 0, 2)                 # [step][component][trial]
    ok = res.status == 0
    out['stats'] = digest(torch, res.stats * ok[:, None])                    # (a FAILed trial's statistics are discarded by the reference)
    return name, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--write', action='store_true')
    ap.add_argument('--only', default=None)
    args = ap.parse_args()
    import torch
    import bench
    import uvs_amd
    uvs_amd.lib()
    got = {}
    for case in CASES:
        if args.only and args.only not in case[0]:
            continue
        name, d = run_case(uvs_amd, torch, bench, case)
        got[name] = d
        print(name, json.dumps(d), flush=True)
    if args.write:
        doc = {'note': 'regression pin of libuvs_rmckf against itself (tools/golden_digest.py), taken from the round-4 kernels', 'cases': got}
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        for path in (PATH, os.path.join(ROOT, 'gpurun_out', 'digests.json')):     # (only gpurun_out/ travels back from the GPU box)
            json.dump(doc, open(path, 'w'), indent=1)
        return
    want = json.load(open(PATH))['cases']
    bad = [(n, k) for n, d in got.items() for k in d if want.get(n, {}).get(k) != d[k]]
    print('MISMATCH' if bad else 'all digests match', bad)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
