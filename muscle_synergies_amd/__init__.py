"""muscle_synergies_amd -- MI355X-native NMF engine behind the ``find_synergies`` call boundary of
elvis-sik/muscle_synergies.

Public names mirror the part of the reference's API that sits on the hot path
(``src/muscle_synergies/__init__.py:5-23``): ``find_synergies``, ``vaf``, ``SynergyRunResult`` -- plus the
estimator (``HipNMF``) and the batched / multi-GPU entry points the reference does not have.

Importing this package never loads the HIP library; the first compute call does, and fails loudly if
``libhip_nmf.so`` is missing or no GPU is visible (there is no CPU fallback for ``solver='mu'``).
"""

from .analysis import SynergyRunResult, find_synergies, find_synergies_batched, vaf
from .engine import (BatchedResult, RankSweepResult, fit_batched, fit_batched_multi_gpu, fit_ragged,
                     random_init_batched, rank_sweep_batched)
from .hip_nmf import HipNMF
from ._lib import HipNmfError

__version__ = "0.1.0"

__all__ = [
    "find_synergies",
    "find_synergies_batched",
    "vaf",
    "SynergyRunResult",
    "HipNMF",
    "fit_batched",
    "fit_batched_multi_gpu",
    "fit_ragged",
    "BatchedResult",
    "rank_sweep_batched",
    "random_init_batched",
    "RankSweepResult",
    "HipNmfError",
]
