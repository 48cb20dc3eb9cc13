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
             I.rti_kernel_name(5, 2, False, 40, nrc=2, qmode=1),
             # the one-launch control tick (ndp_tick): in place and work-list producer, with and without the fused downwash
             I.rti_kernel_name(3, 4, True, 20, tick=True), I.rti_kernel_name(3, 4, False, 20, tick=True),
             I.rti_kernel_name(3, 4, True, 20, qmode=1, tick=True), I.rti_kernel_name(3, 4, False, 20, qmode=1, tick=True)]
    for n in names:
        assert n in k, n
        assert k[n]["scratch"] == 0, (n, k[n])


def test_config5_kernels_have_no_scratch(co):
    """BASELINE config 5's shape (N = 40, 2 RTI iterations): the in-place kernel and the work list's consumer -- the instantiations
    that carry the interior-point loop -- spilled 944 / 912 B per lane through round 3.  What it took: per-iteration
    lane ids for the input / cost / linearisation / write-back index arithmetic (the compiler had hoisted it out of the iteration
    loop and parked it), inputs requested inside the iteration loop, constraint slots with 3 instead of 8 offsets (cost block and
    constants area re-laid out), the corrector's products instead of the predictor's four step arrays, bounds re-read from LDS,
    -Lam^-1 of four stages packed into one register, and its storage as a compile-time choice instead of a null test."""
    k = co.kernels()
    for qmode in (0, 1, 2):
        n = I.rti_kernel_name(5, 2, False, 40, nrc=2, qmode=qmode)
        assert n in k, n
        assert k[n]["scratch"] == 0 and k[n]["spill"] <= 8, (n, k[n])


def test_no_rti_kernel_instantiation_uses_scratch_memory(co):
    """All of them: run-time horizons (which spilled 320-376 B per lane through round 3), the fp32 / bf16 study kernels, every
    compile-time shape."""
    k = {n: v for n, v in co.kernels().items() if "rti_kernel" in n}
    assert len(k) >= 20
    bad = {n: v for n, v in k.items() if v["scratch"] != 0}
    assert not bad, bad


def test_late_force_step_and_downwash_launch_share_a_simd(co):
    """The downwash-one-tick-ahead forms (bench: downwash_forms.prefetch, exchange.peer_ahead) run mlp_stream_kernel of tick t+1 BESIDE the
    control step of tick t: a wave of each per SIMD, whose 512 registers (allocated in units of 8) they share.  The late-force step is
    therefore the lean instantiation (QMODE 3: no stiff sweeps -- they cost 80 registers); in round 4 the step grew to 372 registers
    unnoticed and that form fell from 47 to 23 M solves/s with timed-out force waits."""
    k = co.kernels()
    lean = k[I.rti_kernel_name(3, 4, False, 20, qmode=3)]
    mlp = [v for n, v in k.items() if "mlp_stream_kernel" in n][0]
    up8 = lambda r: (r + 7) // 8 * 8                                            # noqa: E731
    assert up8(lean["vgpr"]) + up8(mlp["vgpr"]) <= 512, (lean, mlp)
    assert lean["scratch"] == 0 and mlp["scratch"] == 0 and mlp["lds"] == 0    # (the control step's workgroups hold the whole LDS)


def test_headline_hot_path_has_no_spill_traffic(co):
    """Up to the end of the first Riccati sweep the headline kernel moves nothing to or from accumulation registers or lane-spill slots:
    the cold branches (interior-point loop, stiff sweeps, debug hooks) are marked NDP_RARELY, so the register allocator -- which weighs
    uses by block frequency -- parks values there and not in the straight-line path."""
    d = co.disassemble(I.rti_kernel_name(3, 4, True, 20))
    idx = [i for i, l in enumerate(d) if "mfma" in l]
    clusters, s, p = [], idx[0], idx[0]
    for i in idx[1:]:
        if i - p > 150:
            clusters.append((s, p)); s = i
        p = i
    clusters.append((s, p))
    end = clusters[1][1]                      # cluster 0: the downwash tile, cluster 1: the first backward + forward sweep
    assert 4000 < end < 6500, clusters
    hot = [l.split()[0] for l in d[:end] if l.strip()]
    bad = [o for o in hot if o in ("v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_readlane_b32", "v_writelane_b32") or o.startswith("scratch_")]
    # What is allowed: a handful of scalar values parked in lanes ONCE in front of the sweep (rounds 4-5: 4; round 6, with the active
    # set's loop state: 6) -- and NOTHING read back, no accumulation-register traffic, no scratch inside the hot region
    assert len(bad) <= 8 and all(o == "v_writelane_b32" for o in bad), (len(bad), bad[:8])


def test_lds_budget_of_the_reference_configuration():
    """4 instances per workgroup at N = 20 fit the CU's 160 KB; 2 at N = 40."""
    from ndp_nmpc_qd_amd import _lib
    lib = _lib.load()
    assert 4 * 8 * lib.ndp_debug_lds_doubles(20) <= 160 * 1024
    assert 2 * 8 * lib.ndp_debug_lds_doubles(40) <= 160 * 1024
