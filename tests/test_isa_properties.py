"""Properties of the SHIPPED gfx950 binary that no run-time test can see (no GPU needed: the code object is taken out of
ndp_nmpc_qd_amd/libndp_nmpc_hip.so and read with llvm-readelf / llvm-objdump)."""
import os

import pytest

from ndp_nmpc_qd_amd import build, isa_inspect as I


@pytest.fixture(scope="module")
def co():
    return I.CodeObject(build.build())


def test_prefetch_epoch_store_waits_for_the_force_rows(co):
    """mlp_stream_kernel publishes a tile's epoch word only after the tile's force rows have COMPLETED --
    an explicit s_waitcnt vmcnt(0) between the row stores and the epoch store (a workgroup-scope release fence emits none on gfx950)."""
    sym = [n for n in co.kernels() if "mlp_stream_kernel" in n]
    assert len(sym) == 1
    ok, detail = I.epoch_store_is_ordered(co.disassemble(sym[0]))
    assert ok, detail


def test_reference_configuration_kernels_have_no_scratch(co):
    """Every N = 20 instantiation (the metric's configuration: in place fused / unfused, work-list producer / consumer) and the
    N = 40 producer keep their whole state in registers + LDS."""
    k = co.kernels()
    names = [I.rti_kernel_name(3, 4, True, 20), I.rti_kernel_name(3, 4, False, 20), I.rti_kernel_name(3, 4, True, 20, qmode=1),
             I.rti_kernel_name(3, 4, False, 20, qmode=1), I.rti_kernel_name(3, 4, False, 20, qmode=2),
             I.rti_kernel_name(5, 2, False, 40, nrc=2, qmode=1)]
    for n in names:
        assert n in k, n
        assert k[n]["scratch"] == 0, (n, k[n])


def test_lds_budget_of_the_reference_configuration():
    """4 instances per workgroup at N = 20 fit the CU's 160 KB; 2 at N = 40."""
    from ndp_nmpc_qd_amd import _lib
    lib = _lib.load()
    assert 4 * 8 * lib.ndp_debug_lds_doubles(20) <= 160 * 1024
    assert 2 * 8 * lib.ndp_debug_lds_doubles(40) <= 160 * 1024
