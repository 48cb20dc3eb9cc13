"""MI355X-native batched NMPC + NN-downwash control step (drop-in for ndp_nmpc's controller API)."""
__version__ = "0.1.0"
