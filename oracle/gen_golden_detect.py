"""Generate tests/golden/detect_circles.npz by running the reference's centre-of-mass detectors (utils.py:11-166).

BUILD-CONTAINER ONLY (imports /root/reference).  OpenCV is not installed here; the detectors use it for exactly one call on
this route, ``cv2.flip(image, 0)`` (vertical flip, utils.py:13,55,134), which the stub below provides as ``image[::-1]``.  The
fixture stores scene parameters (disc centres / radii, rendered by oracle/plant_ref.render_discs) and the reference's outputs.

    python oracle/gen_golden_detect.py
"""
import os
import sys
import types

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
_cv2 = types.ModuleType('cv2')
_cv2.flip = lambda image, code: np.ascontiguousarray(image[::-1]) if code == 0 else (_ for _ in ()).throw(NotImplementedError(code))
sys.modules['cv2'] = _cv2

import utils as RU                                                    # noqa: E402  (reference)
from oracle.plant_ref import render_discs                             # noqa: E402


def main():
    rng = np.random.default_rng(20240611)
    scenes = []
    for i in range(12):
        centres = {c: tuple(rng.uniform(30, 226, 2)) for c in ('red', 'green', 'blue', 'pink')}
        radii = {c: float(rng.uniform(4, 14)) for c in centres}
        scenes.append((centres, radii, bool(i % 2)))
    # a disc cut by the frame edge, and one-pixel discs
    scenes.append(({'red': (2.0, 100.0), 'green': (128.0, 253.5), 'blue': (250.0, 3.0), 'pink': (128.0, 128.0)}, 9.0, False))
    scenes.append(({'red': (10.0, 10.0), 'green': (245.0, 10.0), 'blue': (10.0, 245.0), 'pink': (245.0, 245.0)}, 0.4, False))
    cu, cv, rr, soft, f4, f3, f1 = [], [], [], [], [], [], []
    for centres, radii, soften in scenes:
        img = render_discs(centres, radii, soften=soften)
        order = ('red', 'green', 'blue', 'pink')
        cu.append([centres[c][0] for c in order])
        cv.append([centres[c][1] for c in order])
        rr.append([radii[c] if isinstance(radii, dict) else radii for c in order])
        soft.append(soften)
        f4.append(RU.detect4Circles(img))
        f3.append(RU.detectRGBCircles(img))
        f1.append(RU.detectGreenCircle(img))
    q = rng.standard_normal((6, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    euler = np.array([RU.quat2euler(h) for h in q])
    out = os.path.join(ROOT, 'tests', 'golden', 'detect_circles.npz')
    np.savez_compressed(out, cu=np.array(cu), cv=np.array(cv), radius=np.array(rr), soften=np.array(soft), f4=np.array(f4),
                        f3=np.array(f3), f1=np.array(f1), quat=q, euler=euler)
    print('wrote', out, np.array(f4).shape)


if __name__ == '__main__':
    main()
