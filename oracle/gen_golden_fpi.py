"""Generate tests/golden/fpi_*.npz: MCKF trials of the UNMODIFIED reference whose fixed-point iteration really iterates.

BUILD-CONTAINER ONLY (imports /root/reference through gen_golden.py; only the .npz vectors travel).

The two MCKF fixtures of gen_golden.py (closed_mckf_a1p5 / _a2p0) converge in ONE pass on every step at the reference's
shipped threshold 0.1, so they never reach the second and later passes of experiment.py:215-245.  The runs below do:

  * tight thresholds (1e-3 ... 1e-6): several passes per step (Cholesky factor, Cx != I, gain recomputed, stop test :244);
  * ``fpi_epoch_max`` 3 / 4: "reached max epoch" skips the correction of a step (:246-250) for some steps and not others;
  * Cauchy-like noise (alpha = 1.0 / 1.2): an innovation beyond 38.6 sigma drives a weight Cy to exactly 0 --
    ``inv(Cy)`` raises and the step keeps only the prediction (:225-236);
  * a weight Cy that is subnormal but NOT zero (innovation between 37.7 and 38.6 sigma): ``inv(Cy)`` returns inf without
    raising, ``Br @ Cy_inv @ Br.T`` is 0 * inf = NaN, K and X turn NaN and ``pinv`` raises in the control law: the trial
    ends with ExperimentStatus.FAIL at that step (:232, :312-316).  About 7 % of the reference's own alpha = 1.0 trials
    end this way at its shipped configuration -- including the very first one (seed 123456, fpi_default_a1p0_seed0).

Per step the fixtures also record ``fpi_epochs`` (the reference's ``epoch`` counter after the while loop) and ``fpi_skip``
(``skip_correction``), captured by the same ``sys.settrace`` hook at experiment.py:302.

    python oracle/gen_golden_fpi.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G                                                # noqa: E402  (stubs cv2 / zmq, imports the reference)


def save(name, alpha, seed, **mp):
    M, NT = G.E.Method, G.NoiseType
    out, rec = G.save_closed(name, M.MCKF, NT.ALPHA_STABLE, dict(alpha=alpha, beta=0, gamma=1, delta=0), seed, x_stride=4, prefix='fpi_', **mp)
    epochs = np.array([int(r['epoch']) for r in rec])                 # one entry per executed filter update (k, or k + 1 when the last one FAILed)
    skip = np.array([bool(r['skip_correction']) for r in rec])
    path = os.path.join(G.OUT, f'fpi_{name}.npz')
    z = dict(np.load(path))
    # the whole noise stream of the trial (the logs stop at the FAILing step, whose own sample they do not hold)
    prof = G.NoiseProfiler(num_features=8, noise_type=NT.ALPHA_STABLE, seed=seed, noise_hold=False, noise_hold_cnt=10,
                           noise_params=dict(alpha=alpha, beta=0, gamma=1, delta=0))
    noise_full = np.stack([prof.getNoise().copy() for _ in range(299)])
    assert np.array_equal(noise_full[:len(z['noise'])], z['noise'])
    z.update(fpi_epochs=epochs, fpi_skip=skip, noise_full=noise_full)
    np.savez_compressed(path, **z)
    print(f'    epochs: {np.bincount(epochs)[:8]} multi-pass steps {int((epochs >= 2).sum())}, skipped corrections {int(skip.sum())}, status {out[0].name}, k {len(out[1])}')


def main():
    import logging
    logging.disable(logging.CRITICAL)                                 # "Cy is singular" / "Reached max epoch" / "Experiment failed" are expected here
    save('mckf_a1p5_thr1em6', 1.5, 123456, fpi_threshold=1e-6)                       # 298 multi-pass steps, non-chaotic
    save('mckf_a1p5_anneal_thr1em4', 1.5, 123456, fpi_threshold=1e-4, annealing=True)
    save('mckf_a1p0_thr1em3', 1.0, 123457, fpi_threshold=1e-3)                       # + 10 steps whose Cy underflows to 0
    save('mckf_a1p2_cap3', 1.2, 123458, fpi_threshold=1e-3, fpi_epoch_max=3)         # epoch cap reached on most steps
    save('mckf_a1p2_cap4', 1.2, 123458, fpi_threshold=1e-3, fpi_epoch_max=4)         # ... on some steps
    save('mckf_a1p2_thr1em6_fail', 1.2, 123458, fpi_threshold=1e-6)                  # subnormal Cy at step 157: FAIL
    save('mckf_a1p0_bw1_fail', 1.0, 123457, fpi_threshold=1e-3, kernel_bw=1)         # sigma = 1: many zero weights, FAIL at step 87
    save('default_a1p0_seed0', 1.0, 123456)                                          # the reference's shipped estimator parameters: FAIL at step 110
    save('default_a1p0_seed38', 1.0, 123494)                                         # ... and another seed of the same sweep cell


if __name__ == '__main__':
    main()
