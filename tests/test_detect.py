"""Circle detectors of the live-simulator route (utils.py:11-166 of the reference) against outputs of the reference itself
(tests/golden/detect_circles.npz, written by oracle/gen_golden_detect.py).  Integer thresholding, fp64 centre of mass: bit-exact."""
import os

import numpy as np
import pytest

import uvs_amd
from oracle.plant_ref import render_discs

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'detect_circles.npz'))
ORDER = ('red', 'green', 'blue', 'pink')


def _scene(i):
    centres = {c: (G['cu'][i, j], G['cv'][i, j]) for j, c in enumerate(ORDER)}
    radii = {c: G['radius'][i, j] for j, c in enumerate(ORDER)}
    return render_discs(centres, radii, soften=bool(G['soften'][i]))


@pytest.mark.parametrize('i', range(len(G['f4'])))
def test_centre_of_mass_detectors_match_reference(i):
    img = _scene(i)
    assert np.array_equal(uvs_amd.utils.detect4Circles(img), G['f4'][i])
    assert np.array_equal(uvs_amd.utils.detectRGBCircles(img), G['f3'][i])
    assert np.array_equal(uvs_amd.utils.detectGreenCircle(img), G['f1'][i])
    # the hook Experiment.run() calls (experiment.py:2) takes camera frames as well as synthetic feature carriers
    assert np.array_equal(uvs_amd.experiment.detect4Circles(img), G['f4'][i])


def test_detector_recovers_disc_centres():
    i = 0
    f = uvs_amd.utils.detect4Circles(_scene(i)).reshape(4, 2)
    assert np.abs(f[:, 0] - G['cu'][i]).max() < 0.6 and np.abs(f[:, 1] - G['cv'][i]).max() < 0.6


def test_missing_circle_gives_nan_like_reference():
    img = render_discs({'red': (50, 50), 'green': (100, 100), 'blue': (150, 150)}, 8.0)
    with np.errstate(all='ignore'):
        f = uvs_amd.utils.detect4Circles(img)
    assert np.all(np.isfinite(f[:6])) and np.all(np.isnan(f[6:]))          # 0/0, as utils.py:142-144 produces


def test_detector_method_errors():
    img = _scene(1)
    with pytest.raises(NotImplementedError):
        uvs_amd.utils.detect4Circles(img, uvs_amd.utils.HOUGH_CIRCLES)
    with pytest.raises(Exception, match='Unknown method'):
        uvs_amd.utils.detect4Circles(img, 7)


def test_quat2euler_matches_reference():
    for h, e in zip(G['quat'], G['euler']):
        assert np.array_equal(np.array(uvs_amd.utils.quat2euler(h)), e)
