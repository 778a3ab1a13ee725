"""oarfish_amd -- MI355X-native EM quantification engine for oarfish's hot path.

Only what the path needs: ``csrc/`` (HIP kernels + the C ABI of
``include/oarfish_em.h``), the ctypes loader, and a host-side mirror of the
reference interface (``InMemoryAlignmentStore``, ``EMInfo``, ``em``, ``em_par``,
``bootstrap``; src/em.rs, src/util/oarfish_types.rs of COMBINE-lab/oarfish).
There is no CPU fallback: without ``liboarfish_em.so`` imports of the compute
path fail, and without a HIP device every compute call raises ``OemError``.
"""
from ._lib import OemError, device_count  # noqa: F401
from .types import (AlignmentFilters, DeviceStore, EMInfo, InMemoryAlignmentStore,  # noqa: F401
                    RunInfo, TranscriptInfo)
from .em import bootstrap, em, em_cells, em_par  # noqa: F401

__all__ = [
    "AlignmentFilters", "DeviceStore", "EMInfo", "InMemoryAlignmentStore", "RunInfo",
    "TranscriptInfo", "bootstrap", "em", "em_cells", "em_par", "OemError", "device_count",
]
