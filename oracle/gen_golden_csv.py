"""Generate tests/golden/results_*.csv by running the reference's UNMODIFIED main.py in the build container.

BUILD-CONTAINER ONLY (imports /root/reference, read-only; only the CSV text it writes travels).  main.py is a script: it reads
``config.json`` from the working directory, sweeps 12 cells x ``epoch`` trials and appends each trial's DataFrame to
``results/data/<timestamp>/results.csv`` (main.py:12-13, 104-196).  It is executed here with ``runpy`` in a scratch directory with

  * ``cv2`` and the ZMQ client stubbed (they are only used by the detector bodies / the simulator RPC),
  * ``ur10_simulation.UR10Simulation`` replaced by the kinematic pinhole plant of gen_golden.py (a subclass that keeps the
    reference's own fkine / jacobian / dh), whose ``computePose`` returns the camera position and the angles the reference's
    ``utils.quat2euler`` extracts from the rotation's scalar-first quaternion (ur10_simulation.py:151-163),
  * ``experiment.detect4Circles`` bound to the plant's projection.

Everything else -- config parsing, sweep, seed schedule, q_start jitter, Experiment.run(), the 41-column DataFrame and
``to_csv`` -- is the reference's code.  The fixtures pin the on-disk format of ``batch.write_results_csv`` (column order, status
strings, the ``rho`` column carrying alpha for ALPHA_STABLE, ``kernel_bw`` = annealed bandwidth for MCKF and -1 otherwise).

    python oracle/gen_golden_csv.py
"""
import glob
import json
import os
import runpy
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G                                                # noqa: E402  (stubs cv2 / zmq, imports the reference)

OUT = G.OUT
T_MAX = 0.5                                                           # 9 rows per trial keeps the fixture small


def quat_from_rotation(R):
    """Scalar-first unit quaternion of a rotation matrix (w > 0 branch is enough for the camera poses visited here)."""
    w = 0.5 * np.sqrt(max(0.0, 1.0 + R[0, 0] + R[1, 1] + R[2, 2]))
    assert w > 1e-3
    return np.array([w, (R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w)])


class CsvPlant(G.RefPlant):
    def __init__(self, logger=None, visualization=False):
        super().__init__()

    def computePose(self, recalculate_fkine=False):
        import utils                                                  # reference
        T = self.fkine(recalculate=True)
        return np.r_[T[:3, 3], utils.quat2euler(quat_from_rotation(T[:3, :3]))]


def run_main(config):
    work = tempfile.mkdtemp(prefix='uvs_main_')
    cwd = os.getcwd()
    try:
        os.makedirs(os.path.join(work, 'results', 'data'))
        with open(os.path.join(work, 'config.json'), 'w', encoding='utf-8') as fh:
            json.dump(config, fh)
        os.chdir(work)
        G.U.UR10Simulation = CsvPlant                                 # main.py:3 binds the name at import
        G._Clock.t = 0.0
        runpy.run_path(os.path.join(G.REF, 'main.py'), run_name='__main__')
        (path,) = glob.glob(os.path.join(work, 'results', 'data', '*', 'results.csv'))
        return open(path, encoding='utf-8').read()
    finally:
        os.chdir(cwd)
        shutil.rmtree(work, ignore_errors=True)


def main():
    base = json.load(open(os.path.join(OUT, 'config_reference.json')))
    base.pop('_provenance', None)
    base['log_level'] = 'CRITICAL'
    base['experiments'].update(epoch=1, t_max=T_MAX)
    for name, method, anneal in (('results_mckf_anneal', 'MCKF', True), ('results_gmckf', 'GMCKF', False)):
        cfg = json.loads(json.dumps(base))
        cfg['estimator']['method'] = method
        cfg['estimator']['estimator_params']['annealing'] = anneal
        text = run_main(cfg)
        with open(os.path.join(OUT, name + '.csv'), 'w', encoding='utf-8') as fh:
            fh.write(text)
        with open(os.path.join(OUT, name + '.config.json'), 'w', encoding='utf-8') as fh:
            json.dump(dict(cfg, _provenance={'note': 'config the reference main.py was run with to produce ' + name + '.csv (oracle/gen_golden_csv.py)'}),
                      fh, indent=1, sort_keys=True)
        print(name, len(text.splitlines()), 'lines')


if __name__ == '__main__':
    main()
