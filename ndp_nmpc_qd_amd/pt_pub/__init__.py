"""Batched counterpart of the reference's pt_pub package (ndp_nmpc/scripts/pt_pub/__init__.py:9-10):
the polynomial optimiser (host, numpy -- it runs once per trajectory) and the reference-window publisher whose
per-tick work runs on the MI355X."""
from .polym_optimizer import MinMethod, PolymOptimizer  # noqa: F401
from .pt_publisher import BatchedNMPCRefPublisher, TrajCoefficients  # noqa: F401
