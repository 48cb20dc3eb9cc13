"""Builds the gfx950 shared library in-tree (hipcc cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libndp_nmpc_hip.so")
SOURCES = ["ndp_hip.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp")) + [os.path.join("..", "..", "include", "ndp_nmpc.h")]   # every header ndp_hip.hip can include


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> ndp_nmpc_qd_amd/libndp_nmpc_hip.so"""
    if not (force or _stale()):
        return LIB
    # -amdgpu-mfma-vgpr-form: MFMA results go straight to VGPRs.  With the default AGPR form every accumulator that is
    # live across a basic block or feeds VALU/LDS is copied through v_accvgpr_read/write behind full-latency s_nops,
    # which serialised the matrix pipe against the VALU in the Riccati sweep.
    # -amdgpu-schedule-relaxed-occupancy: the scheduler does not trade instruction order for an occupancy target these kernels cannot
    # reach anyway (one wave per SIMD by LDS): MLP tile 8.45 k -> 8.30 k cycles, headline +0.6 % (round 5, A/B on one box).
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form",
           "-mllvm", "-amdgpu-schedule-relaxed-occupancy=true",
           "-fPIC", "-shared", "-o", LIB] + os.environ.get("NDP_EXTRA_HIPCC_FLAGS", "").split() \
          + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
