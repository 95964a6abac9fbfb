"""Host-side helpers with the reference's names (utils.py of the reference).

Only ``gaussianKernel`` (utils.py:171-172) is on the RMCKF path; the OpenCV circle detectors
(utils.py:11-166) are the perception front-end and out of scope.
"""
import numpy as np


def gaussianKernel(e, bw):
    """Correntropy (Gaussian) kernel exp(-e^2 / (2 bw^2)); same evaluation order as the reference."""
    return np.exp(-0.5 * e ** 2 / bw ** 2)
