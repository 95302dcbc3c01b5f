"""bench.py's own logic that needs no GPU: what its exit status says, and when its launcher refuses a job."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _line(**extra):
    base = {"metric": "m", "value": 1.0, "roofline": {"frac": 0.6},
            "parity_spot_check": {"ok": True, "max_rel": 3e-7},
            "pghi_invert": {"noise_clips_1024": {"seconds": 0.7}, "noise_spot_check": {"ok": True},
                            "tonal_spot_check": {"ok": True}}}
    base.update(extra)
    return base


def test_exit_status_follows_the_spot_checks():
    """a failed verification of a timed result must fail the run (status 5), a side measurement that
    raised gives 4, and a spot check that could not even run (its `*_error` key) is not a pass either."""
    assert bench.verification_failures(_line()) == [] and bench.exit_status(_line()) == 0
    bad = _line(parity_spot_check={"ok": False, "max_rel": 0.5})
    assert bench.verification_failures(bad) == ["parity_spot_check"] and bench.exit_status(bad) == 5
    nested = _line()
    nested["pghi_invert"]["tonal_spot_check"] = {"ok": False, "pop_order_identical": False}
    assert bench.verification_failures(nested) == ["pghi_invert.tonal_spot_check"] and bench.exit_status(nested) == 5
    missing_ok = _line(parity_spot_check={"max_rel": 1e-7})           # no verdict is not a pass
    assert bench.exit_status(missing_ok) == 5
    raised = _line(other_sizes_error="RuntimeError('x')")
    assert bench.exit_status(raised) == 4
    both = _line(other_sizes_error="x", parity_spot_check={"ok": False})
    assert bench.exit_status(both) == 5
    not_run = _line()
    del not_run["parity_spot_check"]
    not_run["parity_spot_check_error"] = "RuntimeError('oracle blew up')"
    assert bench.exit_status(not_run) == 4


def test_launcher_refuses_only_a_definite_undercount(monkeypatch, tmp_path):
    """the parent of `--gpus N` may refuse only when it KNOWS there are too few devices -- a visibility mask
    says so, or a readable KFD topology does.  Here (no /sys/class/kfd) the census is 'unknown' without a mask."""
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    n, definite = bench.visible_gpu_census()
    if not os.path.isdir("/sys/class/kfd/kfd/topology/nodes"):
        assert (n, definite) == (0, False)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpu_census() == (0, True)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    n2, d2 = bench.visible_gpu_census()
    assert d2 and n2 <= 3
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "0")
    assert bench.visible_gpu_census()[0] <= 1


def test_dry_plan_prints_the_multi_gpu_sheet_without_torch():
    """`bench.py --gpus 8 --dry-plan` runs in a process that never imports torch and prints the rank ->
    device map, the clip ranges of configs[3] (8192 / 8), the wire bytes of the three figures and the timeout."""
    import json
    import subprocess
    import sys
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '8', '--dry-plan']\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n    assert e.code in (0, None)\n"
            "assert 'torch' not in sys.modules\n" % os.path.join(ROOT, "bench.py"))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=60)
    assert r.returncode == 0, r.stderr[-1500:]
    plan = json.loads(r.stdout.strip().splitlines()[-1])
    assert plan["dry_plan"] and plan["n_gpus"] == 8
    assert plan["rank_to_device"]["7"] == "cuda:7" and plan["clip_ranges"]["7"] == [7168, 8192]
    per_rank = 1024 * 690 * (128 + 40)
    wire = plan["config4"]["wire_bytes_sent_per_rank_per_step"]
    assert wire == {"with_allgather_fp32": 4 * per_rank, "with_allgather_bf16_wire": 2 * per_rank,
                    "with_allgather_mfcc40_only_fp32": 4 * 1024 * 690 * 40}
    assert plan["config4"]["gathered_bytes_per_rank_per_step"]["with_allgather_fp32"] == 8 * 4 * per_rank
    assert plan["timeout_s"] == 180 and "torch.distributed.run" in plan["commands"]["driver"]
    assert plan["hooks_armed"] == []
