"""bench.py's evidence plumbing, checked off-GPU: which committed profile annotates the default run, and when it is refused.

Round 2 shipped a bench line whose `kernel_us_rocprof` / `frac_rocprof` / clock came from the wrong CSV (a glob picked the
interior-point-always trace, which holds the same kernel instantiation): nothing ran `committed_profile` without a GPU."""
import json
import os
import shutil

import bench


def test_committed_profile_is_the_default_configurations():
    prof = bench.committed_profile(True)
    assert prof["tag"] in bench.PROFILE_TAGS
    assert 15.0 < prof["kernel_us"] < 30.0, prof                      # rti_kernel, batch 1024, early-exit QP: ~20-23 us, not the 61 us of ipm_always
    clock_ghz = prof["wave_cycles_per_simd"] / (prof["kernel_us"] * 1e-6) / 1e9
    assert 2.0 < clock_ghz < 2.5, clock_ghz                           # MI355X shader clock (2.4 GHz peak)
    abytes = bench.algorithmic_bytes_per_solve(20, True) * 1024
    assert abytes <= prof["traffic"] < 2.0 * abytes, (prof["traffic"], abytes)   # PMC traffic: no spill, nothing missing
    f_qp, _ = bench.algorithmic_flops_per_solve(20, 1.0, True)
    frac = f_qp * 1024 / (prof["kernel_us"] * 1e-6) / 1e12 / bench.F64_MFMA_PEAK_TFLOPS
    assert 0.10 < frac < 0.30, frac


def test_other_configurations_get_no_profile():
    prof = bench.committed_profile(False)
    assert prof["kernel_us"] is None and prof["traffic"] is None and prof["wave_cycles_per_simd"] is None


def test_profile_is_read_by_name_not_by_glob(tmp_path):
    """A second trace of the same kernel instantiation in profiles/ (e.g. interior point always) must not be picked up."""
    pdir = tmp_path / "profiles"
    pdir.mkdir()
    tag = bench.committed_profile(True)["tag"]
    src = os.path.join(bench.ROOT, "profiles")
    shutil.copy(os.path.join(src, f"{tag}_kernel_stats_fused_b1024.csv"), pdir)
    shutil.copy(os.path.join(src, f"{tag}_pmc_rti_kernel.json"), pdir)
    want = bench.committed_profile(True, root=str(tmp_path))
    with open(pdir / f"{tag}_kernel_stats_zz_other_run.csv", "w") as fh:      # sorts last, same kernel name, 61 us
        fh.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n'
                 '"void ndp::rti_kernel<3, 4, true, 20, 0, 1, 0>(ndp::KernArgs)",484,29732120,61430.0,99.7,60000,63000,500.0\n')
    assert bench.committed_profile(True, root=str(tmp_path)) == want


def test_a_profile_that_disagrees_with_the_live_duration_is_refused():
    assert bench.profile_agrees(23.06, 22.4)
    assert bench.profile_agrees(23.06, 26.3)
    assert not bench.profile_agrees(61.43, 26.3)          # round 2's wrong CSV against that run's HIP-event duration
    assert not bench.profile_agrees(None, 26.3)
    assert not bench.profile_agrees(23.06, 0.0)


def test_the_rounds_committed_bench_line_is_self_consistent():
    """The bench line kept under profiles/ for the newest tag: rocprof-derived fields agree with the CSV beside it."""
    tag = bench.committed_profile(True)["tag"]
    path = os.path.join(bench.ROOT, "profiles", f"{tag}_bench_b1024_fused.json")
    with open(path) as fh:
        line = json.loads(fh.read())
    roof = line["roofline"]
    assert abs(roof["kernel_us_rocprof"] - bench.committed_profile(True)["kernel_us"]) < 0.01
    assert roof["frac_rocprof"] is None or 0.10 < roof["frac_rocprof"] < 0.30
