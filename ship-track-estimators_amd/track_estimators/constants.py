EARTH_RADIUS = 6378.137  # Radius of the earth in km (reference constants.py:1)
