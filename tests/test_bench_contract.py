"""The driver's contract for `bench.py` (one JSON line on stdout; see the task statement and DESIGN.md section 6): run it as
the driver does -- a child process, default workload, few steps -- and check the line's shape and internal consistency."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract(gpu):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-seconds", "2",
                        "--variant-steps", "2"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly ONE line on stdout"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "ray-steps/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = forward ray-steps of all steps / wall time of the timed region
    assert abs(d["value"] - d["config"]["fwd_ray_steps_global"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    ro = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in ro, k
    assert ro["bound"] in ("hbm", "mfma") and ro["unit"] == "GB/s" and ro["peak"] == 8000.0
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) <= 1e-9
    # achieved = algorithmic bytes per launch / the kernel's average duration measured in this run
    want = ro["algorithmic_bytes_per_ray_step"] * ro["ray_steps_per_launch"] / (ro["avg_kernel_ms"] * 1e-3) / 1e9
    assert abs(ro["achieved"] - want) <= 1e-6 * want
    # the committed PMC summary should belong to the library that ran (profiles of the final library): reported, not fatal --
    # a stale summary makes `traffic` describe other kernels, which the line itself says (pmc_stale)
    assert isinstance(ro["pmc_stale"], bool)
    if ro["pmc_stale"]:
        import warnings
        warnings.warn(f"profiles/*_pmc.json is stale: recorded with {ro['pmc_lib_version']}, library is {d['lib_version']}")
    assert ro["traffic"] is not None and ro["traffic"] > 0
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0
    assert d["parity_check"]["ok"] is True
    v = d["variants"]
    for name in ("cube6_rotated", "plane_shifted", "tomo_weak"):
        # a cross-check, not the parity test (that is tests/test_baseline_configs.py, <= 2e-5 against the oracle for the same
        # three workloads): the one-atomic-per-tap kernel sums in fp32 in whatever order its atomics land, and on the dense
        # shifted source -- 16 rays per cell column -- that noise alone measures 1.7e-5 ... 2.2e-5 from run to run
        assert v[name]["grad_rel_l2_vs_direct_atomics"] <= 5e-5 and v[name]["n_failed"] == 0
        assert v[name]["adj_ns_ratio_to_headline"] > 0
        assert v[name]["adjoint_kernel"]["kernel"] in ("box", "ring", "ring_sparse", "ring_direct")
    # which adjoint kernel the device-side classification chose (a drifting threshold / sort key would show here)
    assert d["config"]["adjoint_kernel"]["kernel"] == "box" and v["cube6_rotated"]["adjoint_kernel"]["kernel"] == "ring_sparse"
    # the line states itself that SURVEY's byte model is exceeded and which bound physically applies
    assert d["whole_step_algorithmic_over_peak"] > 0
    for k in ("physical_bound", "physical_frac", "valu_issue_frac", "clock_ghz", "clock_source"):
        assert k in ro, k
    ph = d["phase_ms"]
    assert ph["sort_avg"] + (ph["pair_copy"] or 0.0) + ph["trace"] + ph["backtrace"] <= d["ms_per_step"] * 1.02


def test_bench_refuses_to_report_a_value_when_parity_fails(gpu):
    """A throughput whose kernels disagree with the oracle is not a result: with the adjoint ablated (development
    switch: no gradient emission) the run must print no `value` and exit non-zero."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-seconds", "2",
                        "--no-variants", "--experiment", "1"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 3, (r.returncode, r.stderr[-1500:])
    d = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert d["value"] is None and d["parity_check"]["ok"] is False and "error" in d
