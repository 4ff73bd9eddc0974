"""muscle_synergies_amd -- MI355X-native NMF engine behind the ``find_synergies`` call boundary of
elvis-sik/muscle_synergies.

Public names mirror the part of the reference's API that sits on the hot path
(``src/muscle_synergies/__init__.py:5-23``): ``find_synergies``, ``vaf``, ``SynergyRunResult`` and the
preprocessing functions that build the matrix (``zero_center``, ``rms``, ``linear_envelope``,
``digital_filter``, ``time_normalize``, ``normalize``) -- plus the estimator (``HipNMF``) and the batched /
multi-GPU entry points the reference does not have.

Importing this package never loads the HIP library; the first compute call does, and fails loudly if
``libhip_nmf.so`` is missing or no GPU is visible (there is no CPU fallback for ``solver='mu'``).
"""

from .analysis import SynergyRunResult, find_synergies, find_synergies_batched, vaf
from .engine import (BatchedResult, HostBatch, RankSweepResult, RestartResult, fit_batched, fit_batched_multi_gpu, fit_ragged,
                     fit_restarts, random_init_batched, random_init_device, rank_sweep_batched, rank_sweep_native)
from .hip_nmf import HipNMF
from .preprocess import (digital_filter, emg_envelope_batched, linear_envelope, linear_envelope_batched, normalize, set_filter_mode,
                         rms, sosfilt_batched, time_normalize, zero_center)
from .segments import find_synergies_segments, segment_frames
from .tsharded import fit_tsharded_devices
from ._lib import HipNmfError

__version__ = "0.1.0"

__all__ = [
    "find_synergies",
    "find_synergies_batched",
    "find_synergies_segments",
    "segment_frames",
    "vaf",
    "SynergyRunResult",
    "HipNMF",
    "fit_batched",
    "fit_batched_multi_gpu",
    "fit_tsharded_devices",
    "fit_ragged",
    "fit_restarts",
    "RestartResult",
    "BatchedResult",
    "HostBatch",
    "rank_sweep_batched",
    "rank_sweep_native",
    "random_init_device",
    "random_init_batched",
    "RankSweepResult",
    "HipNmfError",
    # the preprocessing functions of the reference that build the matrix (src/muscle_synergies/__init__.py:11-18)
    "zero_center",
    "linear_envelope",
    "digital_filter",
    "rms",
    "normalize",
    "time_normalize",
    "emg_envelope_batched",
    "linear_envelope_batched",
    "set_filter_mode",
    "sosfilt_batched",
]
