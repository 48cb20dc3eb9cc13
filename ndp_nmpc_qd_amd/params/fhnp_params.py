"""Airframe constants the hot path reads (reference params/fhnp_params.py:9,12,19)."""
mass = 1.4844  # kg
gravity = 9.81  # m/s^2
c_max = gravity / 0.36  # max collective acceleration, m/s^2
