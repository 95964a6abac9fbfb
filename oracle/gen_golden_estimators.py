"""More closed-loop fixtures for the estimators of SURVEY 8f rank 2 -- KF, IMCC-KF and MCKF under the noise types, the outlier hold and the
bandwidth annealing the first generator exercises for RMCKF only.

BUILD-CONTAINER ONLY (imports /root/reference through gen_golden.py; only the .npz vectors travel).  Every run is the UNMODIFIED reference's
``Experiment.run()`` (experiment.py:48-359) on the plant of SURVEY Appendix A, recorded as in gen_golden.py.

    python oracle/gen_golden_estimators.py      # writes tests/golden/closed_{kf,imcckf,mckf}_*.npz (the eighteen listed in main)
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import numpy as np                                                    # noqa: E402
import gen_golden as G                                                # noqa: E402


def main():
    M, NT = G.E.Method, G.NoiseType
    MIX = dict(std=1.0, mean=50.0, rho=0.1)
    AS = lambda a: dict(alpha=a, beta=0, gamma=1, delta=0)             # noqa: E731
    G.save_closed('kf_white', M.KF, NT.WHITE_NOISE, dict(std=1.0), 223456, x_stride=8)
    G.save_closed('kf_mix', M.KF, NT.GAUSSIAN_MIXTURE, dict(std=1.0, mean=50.0, rho=0.02), 223457, x_stride=8)    # (rho = 0.1 or the hold make the plain KF diverge: a chaotic run pins nothing)
    G.save_closed('kf_bimodal', M.KF, NT.GAUSSIAN_BIMODAL, MIX, 223458, x_stride=8)
    G.save_closed('imcckf_white', M.IMCCKF, NT.WHITE_NOISE, dict(std=1.0), 223459, x_stride=8)
    G.save_closed('imcckf_mix_anneal', M.IMCCKF, NT.GAUSSIAN_MIXTURE, MIX, 223460, annealing=True, x_stride=8)
    G.save_closed('imcckf_bimodal', M.IMCCKF, NT.GAUSSIAN_BIMODAL, MIX, 223461, x_stride=8)
    G.save_closed('imcckf_a1p5_hold', M.IMCCKF, NT.ALPHA_STABLE, AS(1.5), 223465, hold=True, x_stride=8)
    G.save_closed('mckf_white', M.MCKF, NT.WHITE_NOISE, dict(std=1.0), 223462, x_stride=8)
    G.save_closed('mckf_mix_anneal', M.MCKF, NT.GAUSSIAN_MIXTURE, MIX, 223463, annealing=True, x_stride=8)
    G.save_closed('mckf_a1p5_hold', M.MCKF, NT.ALPHA_STABLE, AS(1.5), 223464, hold=True, x_stride=8)
    # other clocks and gains than config.json's 0.05 s / 15 s / 0.2: the loop count of experiment.py:150-152 (t accumulated in floating point),
    # the annealing span (experiment.py:267-271) and the control gain
    for name, meth, dt, t_max, gain, kw in (('gmckf_dt0p02_t6_gain0p5_anneal', M.GMCKF, 0.02, 6, 0.5, dict(annealing=True)),
                                            ('gmckf_dt0p1_t20_gain0p1', M.GMCKF, 0.1, 20, 0.1, {}),
                                            ('mckf_dt0p02_t4_gain0p4', M.MCKF, 0.02, 4, 0.4, {}),
                                            ('kf_dt0p03_t7_gain0p3', M.KF, 0.03, 7, 0.3, {})):
        G.DT, G.T_MAX, G.GAIN = dt, t_max, gain
        G.save_closed(name, meth, NT.ALPHA_STABLE, AS(1.5), 323456, x_stride=8, **kw)
    G.DT, G.T_MAX, G.GAIN = 0.05, 15, 0.2
    # another servo target in the SAME scene: the discs stay where gen_golden.py puts them (goal pose -> DESIRED), desired_f moves.  RefPlant reads the
    # module's DESIRED when it is built, so it is built under the scene's value and the run is made under the target's; the fixture carries both
    scene = G.DESIRED.copy()
    plant_init = G.RefPlant.__init__

    def scene_init(self):
        keep, G.DESIRED = G.DESIRED, scene
        try:
            plant_init(self)
        finally:
            G.DESIRED = keep
    G.RefPlant.__init__ = scene_init
    G.DESIRED = scene + np.array([12.0, -8.0, 12.0, -8.0, 12.0, -8.0, 12.0, -8.0])
    G.save_closed('gmckf_target_shift', M.GMCKF, NT.ALPHA_STABLE, AS(1.5), 423456, x_stride=8, extra=dict(scene_desired=scene))
    G.save_closed('kf_target_shift', M.KF, NT.ALPHA_STABLE, AS(2.0), 423459, x_stride=8, extra=dict(scene_desired=scene))
    G.DESIRED = scene
    G.RefPlant.__init__ = plant_init
    # other kernel bandwidths for the estimators with ONE weight per filter / per state entry
    G.save_closed('imcckf_sigma30_anneal', M.IMCCKF, NT.ALPHA_STABLE, AS(1.5), 423457, kernel_bw=30, annealing=True, x_stride=8)
    G.save_closed('mckf_sigma30', M.MCKF, NT.ALPHA_STABLE, AS(1.2), 423458, kernel_bw=30, x_stride=8)


if __name__ == '__main__':
    main()
