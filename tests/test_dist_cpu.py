"""world_size-2 gloo test of the clip-sharding + all-gather path (CPU, no kernels:
the per-clip function is a stand-in; the HIP transforms have no CPU route)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_clips, q):
    sys.path.insert(0, ROOT)
    from acids_transforms_amd.dist import shard_batch, shard_bounds, all_gather_features, sharded_apply
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        x = torch.randn(n_clips, 50)                       # replicated batch

        def fn(t):                                          # any per-clip (row-independent) function
            return torch.stack([t.cumsum(-1), t * t], 1)

        full = fn(x)
        mine = shard_batch(x)
        lo, hi = shard_bounds(n_clips, rank, world)
        assert mine.shape[0] == hi - lo and torch.equal(mine, x[lo:hi])
        out = sharded_apply(fn, x, gather=True)
        assert out.shape == full.shape and torch.equal(out, full)
        local = sharded_apply(fn, x, gather=False)
        assert torch.equal(local, full[lo:hi])
        out2, work = all_gather_features(local, n_clips, async_op=True)
        if work is not None:
            work.wait()
        assert torch.equal(out2, full)
        q.put((rank, "ok"))
    except Exception as e:                                  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [8, 7])
def test_shard_and_all_gather_gloo(n_clips):
    world = 2
    port = 29500 + (os.getpid() % 2000) + n_clips
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clips, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def _subgroup_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from acids_transforms_amd.dist import shard_bounds, sharded_apply
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        members = [1, 2]
        g = dist.new_group(members)                       # every rank takes part in the creation
        if rank in members:
            torch.manual_seed(0)
            x = torch.randn(5, 12)                         # ragged over two ranks: 3 + 2 clips
            out = sharded_apply(lambda t: t * 2 + 1, x, gather=True, group=g)
            assert torch.equal(out, x * 2 + 1)
            lo, hi = shard_bounds(5, members.index(rank), 2)
            assert torch.equal(sharded_apply(lambda t: t * 2 + 1, x, gather=False, group=g), x[lo:hi] * 2 + 1)
        q.put((rank, "ok"))
    except Exception as e:                                  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_sharding_follows_the_group_not_the_world():
    """sharded_apply over a sub-group shards by the group's rank / size (it used the default group's)."""
    world = 3
    port = 31500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_subgroup_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok"), (2, "ok")], res


def _run_bench(argv, env_extra):
    import subprocess
    env = dict(os.environ, **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        if k not in env_extra:
            env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True,
                          timeout=300)


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-side launcher checks (the GPU box runs the real thing)")
def test_bench_launcher_never_reports_a_smaller_job():
    """`bench.py --gpus N` either runs N ranks or fails: fewer devices than ranks, a launcher WORLD_SIZE that
    disagrees with --gpus, and a failing rank all end non-zero."""
    import json
    r = _run_bench(["--gpus", "2"], {})
    assert r.returncode != 0 and "error" in json.loads(r.stdout.strip().splitlines()[-1])
    r = _run_bench(["--gpus", "2"], {"WORLD_SIZE": "4", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stdout
    # rehearsal mode skips the device-count check, so both ranks really start (and fail: no device here);
    # the parent must relay that failure
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"ACIDS_BENCH_REHEARSAL": "1"})
    assert r.returncode != 0 and r.stdout.count("no ROCm device") >= 1      # (the parent stops the other rank at the first failure)


def test_bench_launcher_parent_never_imports_torch(tmp_path):
    """The parent of `bench.py --gpus N` starts its ranks BEFORE torch is imported and counts devices from sysfs:
    with a `torch` on the path that raises on import (so neither torch.cuda nor hipGetDeviceCount can be reached), the
    parent still gets as far as its own verdict (here: too few devices -> the JSON error and exit status 3); the same
    poisoned torch makes a plain N = 1 run die at once, which shows that the trap is armed."""
    import json
    import subprocess
    (tmp_path / "torch").mkdir()
    (tmp_path / "torch" / "__init__.py").write_text("raise RuntimeError('torch imported in the launcher parent')\n")
    env = dict(os.environ, PYTHONPATH=str(tmp_path) + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env["HIP_VISIBLE_DEVICES"] = ""            # no devices for this test, whatever the box has
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert "torch imported in the launcher parent" not in r.stderr, r.stderr
    assert r.returncode == 3 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus_visible"] == 0
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=env, capture_output=True,
                        text=True, timeout=120)
    assert r1.returncode != 0 and "torch imported in the launcher parent" in r1.stderr


def test_visible_gpu_count_follows_the_runtime_masks(monkeypatch):
    import bench
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    total, known = bench.visible_gpu_census()
    assert total >= 0

    def cap(k):            # a mask is an upper bound; the topology, where readable, lowers it
        return min(total, k) if known else k

    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpu_census() == (0, True)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,-1,2")
    assert bench.visible_gpu_census() == (cap(2), True)          # the list ends at the first negative entry
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "0")
    assert bench.visible_gpu_count() == cap(1)


def test_shard_bounds_cover_everything():
    from acids_transforms_amd.dist import shard_bounds
    for n in [0, 1, 7, 8, 1024, 8191]:
        for w in [1, 2, 3, 8]:
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
