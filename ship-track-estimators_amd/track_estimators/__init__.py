"""
track_estimators — MI355X-native drop-in for the UKF + URTSS path of NOC-OI/ship-track-estimators.

Import paths mirror the reference package (``track_estimators.kalman_filters.unscented`` ...); the filter arithmetic
runs in hand-written HIP kernels (csrc/) reached through the C ABI of include/ste.h.
"""
__version__ = "0.1.0+mi355x"

__all__ = ["__version__", "kalman_filters", "utils", "constants", "ship_track", "batch", "synthetic"]
