"""uncalibrated-visual-servoing_amd: batched RMCKF Jacobian estimator for MI355X behind the reference's Python API.

Modules keep the reference's names: ``experiment`` (Experiment, Method, ExperimentStatus), ``noise`` (NoiseProfiler,
NoiseType), ``utils`` (gaussianKernel).  ``batch`` is the Monte-Carlo driver (main.py of the reference), ``engine`` the
thin typed layer over the C ABI in ``include/uvs_rmckf.h``, ``plant`` the synthetic robot.
"""
from . import _lib, utils, noise, pcg, plant, engine, noise_device, experiment, stats, batch, dist  # noqa: F401
from ._lib import UvsError, UvsLibraryError, build, lib  # noqa: F401
from .experiment import Experiment, ExperimentStatus, Method  # noqa: F401
from .noise import NoiseProfiler, NoiseType, noise_batch  # noqa: F401
from .plant import LinearPlant, SyntheticPlant, SyntheticRobot  # noqa: F401
from .utils import gaussianKernel  # noqa: F401

__version__ = "0.6.0"
