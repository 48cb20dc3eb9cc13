"""Downwash observer gate radius (reference params/downwash_params.py:10)."""
r_horiz = 1.0  # m
