"""Physical and numerical constants shared by the host package (the kernels carry their own copies in csrc/ste_math.h)."""

# Sphere used by the great-circle process model and by the haversine helpers, in km.  Same value as the reference's
# constants.py:1 -- it is the WGS84 equatorial radius.
EARTH_RADIUS = 6378.137

# WGS84 ellipsoid, for the inverse-geodesic fallback in utils.py (metres, flattening).
WGS84_A = 6378137.0
WGS84_F = 1.0 / 298.257223563

# np.radians / np.degrees multiply by these doubles.
DEG2RAD = 0.017453292519943295
RAD2DEG = 57.29577951308232
