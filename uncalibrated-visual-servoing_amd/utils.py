"""Host-side helpers with the reference's names (utils.py of the reference).

``gaussianKernel`` (utils.py:171-172) is on the RMCKF path; the centre-of-mass circle detectors
(utils.py:11-166) are the perception front-end of the live-simulator route, restated in numpy
without OpenCV (their Hough variant is not built).
"""
import numpy as np


def gaussianKernel(e, bw):
    """Correntropy (Gaussian) kernel exp(-e^2 / (2 bw^2)); same evaluation order as the reference."""
    return np.exp(-0.5 * e ** 2 / bw ** 2)


# ---------------------------------------------------------------------------------------------------------------------
# Perception front-end of the live-simulator route: colour-threshold + centre-of-mass circle
# detectors with the reference's names and return conventions (utils.py:11-166).  Host-side numpy: one 256x256 image per
# step at batch size 1 is not device work.  The Hough variant needs OpenCV, which this image does not have.
CENTER_OF_MASS = 0
HOUGH_CIRCLES = 1

_THRESHOLD = 250                                            # utils.py:15
_GRID = np.linspace(0, 1, 256)                              # utils.py:7-9: pixel coordinates normalised to [0, 1]
_X, _Y = np.meshgrid(_GRID, _GRID)
# colour -> (channels that must exceed the threshold, channels that must stay below it); image is RGB
_COLOURS = {'red': ((0,), (1, 2)), 'green': ((1,), (0, 2)), 'blue': ((2,), (0, 1)), 'pink': ((0, 2), (1,))}


def _centre_of_mass(image, colour):
    """(u, v) of one colour's blob: 255 * sum(grid * mask) / sum(mask), mask in {0, 255} as uint8 like the reference
    (utils.py:18-27), on the vertically flipped image (cv2.flip(image, 0), utils.py:13)."""
    flipped = np.asarray(image)[::-1]
    above, below = _COLOURS[colour]
    mask = np.ones(flipped.shape[:2], bool)
    for ch in above:
        mask &= flipped[:, :, ch] > _THRESHOLD
    for ch in below:
        mask &= flipped[:, :, ch] < _THRESHOLD
    weight = 255 * mask.astype(np.uint8)
    total = np.sum(weight)
    return 255 * np.sum(_X * weight) / total, 255 * np.sum(_Y * weight) / total


def _detect(image, colours, method):
    if method == HOUGH_CIRCLES:
        raise NotImplementedError('HOUGH_CIRCLES needs OpenCV (cv2.HoughCircles); only CENTER_OF_MASS is built')
    if method != CENTER_OF_MASS:
        raise Exception('Unknown method')                   # utils.py:47-48
    f = np.zeros(2 * len(colours))
    for i, colour in enumerate(colours):
        f[2 * i], f[2 * i + 1] = _centre_of_mass(image, colour)
    return f


def detectGreenCircle(image, method=CENTER_OF_MASS):
    """utils.py:11-51: f = [u, v] of the green circle."""
    return _detect(image, ('green',), method)


def detectRGBCircles(image, method=CENTER_OF_MASS):
    """utils.py:53-124: f = [u_r, v_r, u_g, v_g, u_b, v_b]."""
    return _detect(image, ('red', 'green', 'blue'), method)


def detect4Circles(image, method=CENTER_OF_MASS):
    """utils.py:126-166: red, green, blue and pink circles, f in R^8."""
    return _detect(image, ('red', 'green', 'blue', 'pink'), method)


def quat2euler(h):
    """utils.py:174-179 (quaternion scalar-first)."""
    roll = np.arctan2(2 * (h[0] * h[1] + h[2] * h[3]), 1 - 2 * (h[1] ** 2 + h[2] ** 2))
    pitch = np.arcsin(2 * (h[0] * h[2] - h[3] * h[1]))
    yaw = np.arctan2(2 * (h[0] * h[3] + h[1] * h[2]), 1 - 2 * (h[2] ** 2 + h[3] ** 2))
    return (roll, pitch, yaw)
