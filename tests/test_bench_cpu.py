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
