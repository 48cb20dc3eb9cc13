"""`bench.py --gpus N` without a torch.distributed environment launches its own N ranks (bench.launch_ranks): a child
`python -m torch.distributed.run`, rank 0's JSON line relayed, the child's exit code returned.  Driven here with a stub rank
script on the CPU (two children), and once end to end: the real bench.py on a box without a GPU must fail in its RANKS, not
in the launcher, and hand the failure back."""
import io
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

STUB = r'''
import json, os, sys
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["MASTER_ADDR"] == "127.0.0.1"
print(f"noise from rank {rank} of {world}", flush=True)
print("{not json", flush=True)
if "--fail" in sys.argv and rank == 1:
    sys.exit(7)
if "--silent" in sys.argv:
    sys.exit(0)
if rank == 0:
    print(json.dumps({"metric": "stub", "value": 1.0, "n_gpus": world, "argv": sys.argv[1:],
                      "ipc": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}), flush=True)
    print(json.dumps({"metric": "a second line must not reach stdout"}), flush=True)
'''


@pytest.fixture()
def stub(tmp_path):
    p = tmp_path / "stub_rank.py"
    p.write_text(STUB)
    return str(p)


def _env_without_dist(monkeypatch):
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)


def test_two_ranks_json_line_relayed(stub, monkeypatch):
    _env_without_dist(monkeypatch)
    out, err = io.StringIO(), io.StringIO()
    rc = bench.launch_ranks(2, stub, ["--gpus", "2", "--steps", "5"], out=out, err=err)
    assert rc == 0, err.getvalue()
    lines = [ln for ln in out.getvalue().splitlines() if ln.strip()]
    assert len(lines) == 1                                  # ONE JSON line on stdout, nothing else
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["argv"] == ["--gpus", "2", "--steps", "5"]      # the same arguments reach the ranks
    assert rec["ipc"] == "0"                                # dmabuf IPC is set for the children
    e = err.getvalue()
    assert "noise from rank 0 of 2" in e and "noise from rank 1 of 2" in e and "a second line" in e


def test_failing_rank_sets_the_exit_code(stub, monkeypatch):
    _env_without_dist(monkeypatch)
    out, err = io.StringIO(), io.StringIO()
    rc = bench.launch_ranks(2, stub, ["--fail"], out=out, err=err)
    assert rc != 0


def test_no_json_line_is_an_error(stub, monkeypatch):
    _env_without_dist(monkeypatch)
    out, err = io.StringIO(), io.StringIO()
    assert bench.launch_ranks(2, stub, ["--silent"], out=out, err=err) == 3
    assert out.getvalue() == ""


def test_bench_gpus_2_launches_itself(monkeypatch):
    """The driver's command shape with N = 2 and no torchrun around it.  There is no GPU here: both ranks must start, say so, and
    the parent must return their failure -- not die in front of the launch with "launch with torch.distributed.run"."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CUDA_VISIBLE_DEVICES"] = ""
    env["HIP_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "launching 2 ranks" in r.stderr
    assert "needs an MI355X" in r.stderr
    assert r.stdout.strip() == ""


def test_world_size_mismatch_is_still_refused(monkeypatch):
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr


def test_launcher_refuses_under_a_profiler_preload(stub, monkeypatch):
    """rocprofv3's preloaded library initialises the GPU in the parent: starting ranks from there is an exec chain out of a
    GPU-initialised process.  Refused with a message and a non-zero code; nothing is launched."""
    _env_without_dist(monkeypatch)
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    out, err = io.StringIO(), io.StringIO()
    assert bench.launch_ranks(2, stub, ["--gpus", "2"], out=out, err=err) == 6
    assert out.getvalue() == "" and "profiler preload" in err.getvalue() and "launching" not in err.getvalue()


def test_the_line_is_compacted_and_the_legend_covers_its_blocks():
    line = {"value": 47190568.11407378, "ms_per_step": 0.02169925137422979, "n": 3, "ok": True, "none": None, "nan": float("nan"),
            "nested": {"a": [1.23456789e-7, 2, "text"], "b": (0.1 + 0.2,)}}
    c = bench.compact(line)
    assert c["value"] == 47191000.0 and c["ms_per_step"] == 0.021699 and c["n"] == 3 and c["ok"] is True and c["none"] is None
    assert c["nan"] != c["nan"] and c["nested"]["a"] == [1.2346e-07, 2, "text"] and c["nested"]["b"] == [0.3]
    assert len(json.dumps(c, separators=(",", ":"))) < len(json.dumps(line))
    for block in ("roofline", "forms", "scaling_baseline", "configs", "ipm_always", "mixed", "constrained", "repeat", "host", "tick", "cpu_baseline", "config1", "rows"):
        assert block in bench.LEGEND and len(bench.LEGEND[block]) > 40


def test_the_committed_line_fits_the_drivers_record():
    """the driver keeps the tail of stdout (~8 KB); the line must come through whole."""
    tag = bench.committed_profile(True)["tag"]
    with open(os.path.join(ROOT, "profiles", f"{tag}_bench_b1024_fused.json")) as fh:
        raw = fh.read().strip()
    assert len(raw) <= 6144, len(raw)
    d = json.loads(raw)
    for block in ("roofline", "cpu_baseline", "tick", "rows", "config1", "configs", "host", "mixed", "ipm_always", "scaling_baseline"):
        assert block in d, block
    assert d["tick"]["launches_per_tick"] == 1 and d["tick"]["parity"] <= 1e-5 and d["tick"]["value_host_inclusive_x0_only"] > 3.0e7
    assert list(d).index("configs") < list(d).index("mixed")
    # the line says which reading of the metric `value` is and carries SURVEY 8d's host-inclusive readings beside it;
    # the timed region is backed by repeated passes over the same steps; the host block says what it ran on
    assert d["metric_variant"] == "device_resident"
    assert d["value_host_inclusive"] == d["host"]["two"]["value"] and d["value_tick"] == d["tick"]["value_host_inclusive_x0_only"]
    rp = d["repeat"]
    assert rp["n"] >= 10 and rp["ms_per_step"][0] <= rp["ms_per_step"][1] <= rp["ms_per_step"][2] and abs(rp["ms_per_step"][1] / d["ms_per_step"] - 1) < 0.1
    assert d["host"]["pack_threads"] >= 0 and 1 <= d["host"]["usable_cores"] <= d["host"]["hw_threads"]
    assert d["cpu_baseline"]["per_core"] * d["cpu_baseline"]["cores"] == pytest.approx(d["cpu_baseline"]["value"], rel=1e-3) and 0.1 < d["cpu_baseline"]["mlp_share"] < 0.7
    # the constrained workloads in the default mode against the interior-point loop on the same inputs
    assert d["mixed"]["b1024"]["value"] >= 1.6e7 and d["mixed"]["b1024"]["value"] > 2 * d["mixed"]["b1024"]["value_as_off"] and d["mixed"]["b1024"]["bad"] == 0
    assert d["constrained"]["value"] >= 2.5e7 and d["constrained"]["value_as_off"] < d["constrained"]["value"] and d["constrained"]["bad"] == 0
    # row +: the condensed study's legs
    c5 = d["configs"]["config5"]["nominal"]
    assert c5["condensed_fp32"]["qps_kept_condensed"] == 2.0 and 1e-6 < c5["condensed_fp32"]["err"] < 1e-2 and c5["condensed_fp32"]["value"] < c5["fp64_in_place"]["value"]
    assert c5["condensed_bf16"]["bad"] == 0
