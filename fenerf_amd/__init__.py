"""fenerf_amd -- MI355X-native volumetric rendering core behind FENeRF's generator API.

The hot path (FiLM-SIREN evaluation, compositing, hierarchical resampling) lives in
csrc/ as hand-written HIP for gfx950 and is reached through the C-ABI in
include/fenerf.h (loaded lazily by fenerf_amd._lib; it raises if the shared
library is missing -- there is no CPU fallback).
"""
__version__ = "0.1.0"
