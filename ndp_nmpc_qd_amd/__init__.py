"""MI355X-native batched NMPC + NN-downwash control step (drop-in for ndp_nmpc's controller API).

    from ndp_nmpc_qd_amd.nmpc_ctl import NMPCBodyRateController
    from ndp_nmpc_qd_amd.ndp_nmpc_ctl import NDPNMPCBodyRateController
    from ndp_nmpc_qd_amd.dnwash_nn_est import DownwashNN
    from ndp_nmpc_qd_amd import BatchedNMPC            # B instances per call
"""
__version__ = "0.1.0"

from .batched import BatchedNMPC, NdpError  # noqa: E402,F401
