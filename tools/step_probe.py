"""Step-level diagnosis of the bench step -- run ON THE GPU BOX.

    python tools/step_probe.py [sections] [--libs name=path,...] [--out gpurun_out/r05_step_probe.json]

sections (comma list, default all):
  ptrs    per-step, per-kernel HIP-event times of K consecutive bench steps together with the data_ptr() of X / feat / y:
          does a kernel's duration alternate with period 2, and does it follow the allocator's blocks?
  hold    the same with the three outputs allocated ONCE and every step writing the same buffers (no allocator in the loop)
  power   sclk / socket power sampled from sysfs by a side thread (10 ms) across idle -> fresh process steps -> settled steps,
          then each kernel of the step alone in a loop: average power, ms per launch and JOULES PER FRAME per kernel
  pattern the access-pattern copy kernels of tools/ubench/pattern_lib.hip (no arithmetic): ms and watts
  energy  instruction-energy microbenchmarks (tools/ubench/pattern_lib.hip ek_run): nJ per wave-instruction by kind
  ab      same-process alternating A/B of the WHOLE step over library builds (--libs a=tools/ab/libacids_a.so,...; the
          in-tree library is always "0"): per round and library, settle launches then 20 timed steps (wall + events)

The library handle is swapped under acids_transforms_amd._lib (every loaded build gets its own at_init); all tensors,
modules and allocator state are shared, so the only thing that changes between A and B is the kernels.
"""
import argparse
import ctypes
import glob
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import acids_transforms_amd as A  # noqa: E402
from acids_transforms_amd import _lib as L  # noqa: E402
from acids_transforms_amd import ops  # noqa: E402

SR, N_FFT, HOP = 44100, 1024, 256
CLIP = 4 * SR
T = 1 + CLIP // HOP
BYTES = {"fused": 5640, "istft": 5128, "plain": 5128, "featonly": 1536}


# ------------------------------------------------------------------------------------------------ telemetry
def device_hwmon(index=0):
    """hwmon directory of HIP device `index`: PCI bus id from the runtime -> /sys/bus/pci/devices/<bdf>/hwmon/hwmon*."""
    hip = ctypes.CDLL("libamdhip64.so")
    buf = ctypes.create_string_buffer(64)
    if hip.hipDeviceGetPCIBusId(buf, 64, index) != 0:
        return None, None
    bdf = buf.value.decode().lower()
    hits = glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % bdf)
    return (hits[0] if hits else None), bdf


class Sampler(threading.Thread):
    """(t, watts, sclk MHz) every `period` seconds from the device's hwmon files (plain reads; no SMI library)."""

    def __init__(self, hwmon, period=0.01):
        super().__init__(daemon=True)
        self.period = period
        self.fp = open(os.path.join(hwmon, "power1_input")) if hwmon else None
        self.ff = open(os.path.join(hwmon, "freq1_input")) if hwmon else None
        self.rows = []
        self.stop_flag = False

    def read(self):
        self.fp.seek(0)
        self.ff.seek(0)
        return time.perf_counter(), int(self.fp.read()) * 1e-6, int(self.ff.read()) * 1e-6

    def run(self):
        if self.fp is None:
            return
        while not self.stop_flag:
            try:
                self.rows.append(self.read())
            except (OSError, ValueError):
                pass
            time.sleep(self.period)

    def window(self, t0, t1):
        rows = [r for r in self.rows if t0 <= r[0] <= t1]
        if not rows:
            return {"samples": 0}
        w = [r[1] for r in rows]
        f = [r[2] for r in rows]
        return {"samples": len(rows), "watts_mean": sum(w) / len(w), "watts_max": max(w), "watts_min": min(w),
                "sclk_mean": sum(f) / len(f), "sclk_min": min(f), "sclk_max": max(f)}


# ------------------------------------------------------------------------------------------------ library builds
class Libs:
    def __init__(self, spec):
        self.handles = {"0": L.lib()}
        for kv in filter(None, (spec or "").split(",")):
            name, path = kv.split("=")
            h = ctypes.CDLL(os.path.abspath(path))
            for fn, argtypes in L._SIGNATURES.items():
                f = getattr(h, fn)
                f.argtypes = argtypes
                f.restype = L._RESTYPES.get(fn, L.c_int)
            assert h.at_abi_version() == L.ABI_VERSION, path
            L.check(h.at_init(torch.cuda.current_device()), "at_init " + name)
            self.handles[name] = h

    def use(self, name):
        L._lib = self.handles[name]


def ev():
    return torch.cuda.Event(enable_timing=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sections", nargs="?", default="ptrs,hold,power,pattern,energy,ab")
    ap.add_argument("--libs", default="")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--out", default="gpurun_out/r05_step_probe.json")
    ap.add_argument("--unfused", action="store_true", help="the step as three kernels: plain forward, stand-alone projection, inverse")
    ap.add_argument("--env", default="", help="ab: per-library environment, e.g. dev:ACIDS_FWD_FPR=173 (dev builds read it per launch)")
    args = ap.parse_args()
    sections = args.sections.split(",")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    hwmon, bdf = device_hwmon(0)
    sampler = Sampler(hwmon)
    t_proc0 = time.perf_counter()
    sampler.start()
    time.sleep(0.5)                                   # idle window before anything is launched
    t_idle1 = time.perf_counter()

    B = args.batch
    frames = B * T
    gen = torch.Generator(device=dev).manual_seed(1234)
    x = torch.randn(B, CLIP, device=dev, generator=gen) * 0.1
    stft = A.STFT(sr=SR, n_fft=N_FFT, hop_length=HOP).to(dev)
    mag = A.Magnitude(sr=SR, n_fft=N_FFT, n_mels=128, mode="unipolar", contrast="log1p").to(dev)
    mag.scale_data(stft(x[:8]))
    libs = Libs(args.libs)
    res = {"hwmon": hwmon, "pci": bdf, "batch": B, "frames": frames}
    held = {}

    def step(record=False, ptrs=None):
        held.clear()                                   # bench.py's order: release, then allocate
        e = [ev() for _ in range(3)] if record else None
        if record:
            e[0].record()
        if args.unfused:
            X = stft(x)
            feat = mag(X)
        else:
            X, feat = mag.forward_fused(stft, x, return_spectrum=True)
        if record:
            e[1].record()
        y = stft.invert(X)
        if record:
            e[2].record()
        if ptrs is not None:
            ptrs.append((X.data_ptr(), feat.data_ptr(), y.data_ptr()))
        held["X"], held["feat"], held["y"] = X, feat, y
        return e

    def settle(fn, max_steps=600):
        marks, means, n = [ev()], [], 0
        marks[0].record()
        while n < max_steps:
            for _ in range(10):
                fn()
            n += 10
            marks.append(ev())
            marks[-1].record()
            if len(marks) >= 3:
                marks[-2].synchronize()
                means.append(marks[-3].elapsed_time(marks[-2]) / 10.0)
            if n >= 40 and len(means) >= 3 and max(means[-3:]) <= 1.015 * min(means[-3:]):
                break
        torch.cuda.synchronize()
        return n, means

    def timed(fn, k):
        """K steps: wall ms per step and per-kernel event times."""
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        evs = [fn(True) for _ in range(k)]
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / k * 1e3
        fwd = [e[0].elapsed_time(e[1]) for e in evs]
        inv = [e[1].elapsed_time(e[2]) for e in evs]
        return wall, fwd, inv

    def even_odd(v):
        a, b = v[0::2], v[1::2]
        return {"even": sum(a) / len(a), "odd": sum(b) / len(b), "rel": (sum(a) / len(a)) / (sum(b) / len(b)) - 1.0}

    # -------------------------------------------------------------------------------- fresh process + settle (power trace)
    t_fresh0 = time.perf_counter()
    for _ in range(25):
        step()
    torch.cuda.synchronize()
    t_fresh1 = time.perf_counter()
    n_settle, means = settle(step)
    t_settled = time.perf_counter()
    res["settle"] = {"steps": n_settle, "ten_step_means_ms": [round(m, 4) for m in means]}

    if "ptrs" in sections:
        ptrs = []
        wall, fwd, inv = timed(lambda rec: step(rec, ptrs), args.steps)
        uniq = [len(set(p[i] for p in ptrs)) for i in range(3)]
        res["ptrs"] = {"wall_ms_per_step": wall, "fwd_ms": [round(v, 4) for v in fwd], "inv_ms": [round(v, 4) for v in inv],
                       "distinct_ptrs_X_feat_y": uniq, "ptrs_first4": [[hex(q) for q in p] for p in ptrs[:4]],
                       "fwd_even_odd": even_odd(fwd), "inv_even_odd": even_odd(inv),
                       "step_even_odd": even_odd([a + b for a, b in zip(fwd, inv)])}
        print("ptrs   wall %.4f ms/step  fwd %.4f  inv %.4f  distinct ptrs %s" % (wall, sum(fwd) / len(fwd), sum(inv) / len(inv), uniq))
        print("       even/odd: fwd %+.2f %%  inv %+.2f %%  step %+.2f %%" % tuple(
            100 * res["ptrs"][k]["rel"] for k in ("fwd_even_odd", "inv_even_odd", "step_even_odd")), flush=True)

    if "hold" in sections:
        # the kernels into buffers allocated once (C ABI called directly through ops' own helpers would re-allocate: use
        # the step with outputs held, i.e. the allocator must hand out a SECOND set -- and compare with a fixed set)
        held.clear()
        torch.cuda.synchronize()
        Xb = torch.empty((B, T, 513), dtype=torch.complex64, device=dev)
        fb = torch.empty((B, T, 128), dtype=torch.float32, device=dev)
        yb = torch.empty((B, HOP * (T - 1)), dtype=torch.float32, device=dev)
        band = mag._banded()
        lane_filter, lane_start, weights = band.on(dev)
        off, sc = mag._affine()
        win = stft.window[:1024]
        iw = stft.inv_window[:1024]
        env16 = stft._env16 if stft._env16.numel() else ops.istft_envelope_table(iw, N_FFT, HOP)

        def fixed_step(record=False):
            e = [ev() for _ in range(3)] if record else None
            if record:
                e[0].record()
            L.check(L.lib().at_stft_mel_forward(L.ptr(x), B, CLIP, CLIP, T, N_FFT, HOP, L.ptr(win), L.ptr(lane_filter), L.ptr(lane_start),
                                                L.ptr(weights), band.N, band.n_passes, band.pass_len.ctypes.data, ops.contrast_code("log1p"), 0,
                                                L.ptr(off), L.ptr(sc), mag._eps, L.ptr(Xb), L.ptr(None), L.ptr(fb), 0,
                                                L.stream_ptr()), "fwd")
            if record:
                e[1].record()
            L.check(L.lib().at_istft(L.ptr(Xb), L.ptr(None), L.ptr(None), B, T, N_FFT, HOP, L.ptr(iw), L.ptr(env16), L.ptr(yb),
                                     L.ptr(None), 0, L.stream_ptr()), "inv")
            if record:
                e[2].record()
            return e

        settle(fixed_step, 100)
        wall, fwd, inv = timed(fixed_step, args.steps)
        res["hold"] = {"wall_ms_per_step": wall, "fwd_ms": [round(v, 4) for v in fwd], "inv_ms": [round(v, 4) for v in inv],
                       "fwd_even_odd": even_odd(fwd), "inv_even_odd": even_odd(inv),
                       "step_even_odd": even_odd([a + b for a, b in zip(fwd, inv)])}
        print("hold   wall %.4f ms/step  fwd %.4f  inv %.4f (fixed buffers, C ABI called directly)" % (
            wall, sum(fwd) / len(fwd), sum(inv) / len(inv)))
        print("       even/odd: fwd %+.2f %%  inv %+.2f %%  step %+.2f %%" % tuple(
            100 * res["hold"][k]["rel"] for k in ("fwd_even_odd", "inv_even_odd", "step_even_odd")), flush=True)
        del Xb, fb, yb

    if "power" in sections:
        pw = {"idle": sampler.window(t_proc0, t_idle1), "fresh_25_steps": sampler.window(t_fresh0, t_fresh1),
              "settling": sampler.window(t_fresh1, t_settled)}
        # settled step, 1.5 s
        t0 = time.perf_counter()
        n = 0
        e0, e1 = ev(), ev()
        e0.record()
        while time.perf_counter() - t0 < 1.5:
            for _ in range(20):
                step()
            n += 20
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        w = sampler.window(t0 + 0.2, t1)
        ms = (t1 - t0) / n * 1e3
        w.update({"ms_per_step": ms, "joules_per_step": w.get("watts_mean", 0) * ms * 1e-3,
                  "microjoules_per_frame": w.get("watts_mean", 0) * ms * 1e-3 / frames * 1e6})
        pw["step_settled"] = w
        X = stft(x)
        off, sc = mag._affine()
        kernels = {
            "fused": lambda: mag.forward_fused(stft, x, return_spectrum=True),
            "istft": lambda: stft.invert(X),
            "plain": lambda: stft(x),
            "featonly": lambda: ops.stft_mel_forward(x, stft.window[:N_FFT], mag._banded(), "log1p", off, sc, mag._eps,
                                                     want_spectrum=False),
            "copy_": None,
        }
        a = torch.empty(1 << 29, device=dev)
        b = torch.empty(1 << 29, device=dev)
        kernels["copy_"] = lambda: b.copy_(a)
        # the reference's default 513-filter bank: spectrum + features (generic epilogue) and features only (packed epilogue)
        mag513 = A.Magnitude(sr=SR, n_fft=N_FFT, mode="unipolar", contrast="log1p").to(dev)
        mag513.scale_data(X[:8])
        off5, sc5 = mag513._affine()
        kernels["fused513"] = lambda: mag513.forward_fused(stft, x, return_spectrum=True)
        kernels["featonly513"] = lambda: ops.stft_mel_forward(x, stft.window[:N_FFT], mag513._banded(), "log1p", off5, sc5, mag513._eps,
                                                                want_spectrum=False)
        BYTES.update({"fused513": 1024 + 4104 + 2052, "featonly513": 1024 + 2052})
        idle_w = pw["idle"].get("watts_mean", 0.0)
        for name, fn in kernels.items():
            for _ in range(30):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < 1.2:
                for _ in range(40):
                    fn()
                n += 40
                torch.cuda.synchronize()
            t1 = time.perf_counter()
            w = sampler.window(t0 + 0.2, t1)
            ms = (t1 - t0) / n * 1e3
            w["ms_per_launch"] = ms
            w["joules_per_launch"] = w.get("watts_mean", 0) * ms * 1e-3
            if name in BYTES:
                w["microjoules_per_frame"] = w["joules_per_launch"] / frames * 1e6
                w["microjoules_per_frame_above_idle"] = (w.get("watts_mean", 0) - idle_w) * ms * 1e-3 / frames * 1e6
                w["frac_of_8TBps"] = frames * BYTES[name] / (ms * 1e-3) / 8e12
            else:
                w["TBps"] = 2 * 4 * (1 << 29) / (ms * 1e-3) / 1e12
            pw[name] = w
            print("power  %-9s %.4f ms  %6.0f W (min %4.0f max %4.0f)  sclk %4.0f MHz (min %4.0f)  %s" % (
                name, ms, w.get("watts_mean", 0), w.get("watts_min", 0), w.get("watts_max", 0), w.get("sclk_mean", 0),
                w.get("sclk_min", 0), ("%.3f uJ/frame" % w["microjoules_per_frame"]) if name in BYTES else ""), flush=True)
        del X, a, b
        pw["trace_first_4s"] = [[round(r[0] - t_proc0, 3), round(r[1], 1), round(r[2])] for r in sampler.rows
                                if r[0] - t_proc0 < 4.0][::2]
        res["power"] = pw
        for k in ("idle", "fresh_25_steps", "settling", "step_settled"):
            print("power  %-15s %s" % (k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in pw[k].items()}), flush=True)

    if "ab" in sections and len(libs.handles) > 1:
        envs = {}
        for item in filter(None, args.env.split(";")):
            name, kv = item.split(":", 1)
            envs.setdefault(name, []).append(kv.split("=", 1))
        ab = {name: {"wall": [], "fwd": [], "inv": []} for name in libs.handles}
        for r in range(args.rounds):
            for name in libs.handles:
                libs.use(name)
                for k, v in envs.get(name, []):
                    os.environ[k] = v
                for _ in range(30):           # a change of kernels is a change of power draw: let the controller settle
                    step()
                wall, fwd, inv = timed(lambda rec: step(rec), 20)
                for k, v in envs.get(name, []):
                    os.environ.pop(k, None)
                ab[name]["wall"].append(round(wall, 4))
                ab[name]["fwd"].append(round(sum(fwd) / len(fwd), 4))
                ab[name]["inv"].append(round(sum(inv) / len(inv), 4))
                print("ab     round %d  %-8s wall %.4f ms/step  fwd %.4f  inv %.4f" % (r, name, wall, ab[name]["fwd"][-1], ab[name]["inv"][-1]),
                      flush=True)
        libs.use("0")
        for name, v in ab.items():
            v["wall_mean"] = sum(v["wall"]) / len(v["wall"])
            v["fwd_mean"] = sum(v["fwd"]) / len(v["fwd"])
            v["inv_mean"] = sum(v["inv"]) / len(v["inv"])
        res["ab"] = ab
        print("ab     means: " + "  ".join("%s %.4f (fwd %.4f inv %.4f)" % (n, v["wall_mean"], v["fwd_mean"], v["inv_mean"])
                                             for n, v in ab.items()), flush=True)

    def pattern_lib():
        h = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "libpattern.so"))
        V, I64, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
        h.pat_fwd.argtypes = [V, V, V, I64, I, I, V]
        h.pat_inv.argtypes = [V, V, I64, I, I, V]
        h.ek_run.argtypes = [I, I, I, V, V]
        h.pat_fwd_flags.argtypes = [V, V, V, I64, I, I, I, V]
        h.pat_inv_flags.argtypes = [V, V, I64, I, I, I, V]
        h.pat_fwd513.argtypes = [V, V, V, I64, I, I, I, V]
        return h

    def powered_loop(fn, seconds=1.2, chunk=20, warm=10):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(chunk):
                fn()
            n += chunk
            torch.cuda.synchronize()
        t1 = time.perf_counter()
        w = sampler.window(t0 + 0.25, t1)
        w["ms_per_launch"] = (t1 - t0) / n * 1e3
        return w

    if "pattern" in sections:
        # the access-pattern copy kernels (no arithmetic) on this box, in this process: time AND power
        pl = pattern_lib()
        Xb = torch.empty((B, T, 513), dtype=torch.complex64, device=dev)
        fb = torch.empty((B, T, 128), dtype=torch.float32, device=dev)
        yb = torch.empty((B, HOP * (T - 1) + 1024), dtype=torch.float32, device=dev)
        xb = torch.empty((B * T * HOP + 1024,), dtype=torch.float32, device=dev).normal_()
        pat = {}
        for name, G, wpb in (("D_G8_wpb4", 8, 4), ("D_G16_wpb4", 16, 4), ("D_G32_wpb4", 32, 4), ("D_G173_wpb8", 173, 8)):
            f = powered_loop(lambda: pl.pat_fwd(L.ptr(xb), L.ptr(Xb), L.ptr(fb), frames, G, wpb, L.stream_ptr()))
            i = powered_loop(lambda: pl.pat_inv(L.ptr(Xb), L.ptr(yb), frames, G, wpb, L.stream_ptr()))
            pat[name] = {"fwd_feat": f, "inv": i}
            print("pattern %-12s fwd+feat %.4f ms %5.0f W sclk %4.0f | inv %.4f ms %5.0f W sclk %4.0f" % (
                name, f["ms_per_launch"], f.get("watts_mean", 0), f.get("sclk_mean", 0), i["ms_per_launch"], i.get("watts_mean", 0),
                i.get("sclk_mean", 0)), flush=True)
        res["pattern"] = pat
        del Xb, fb, yb, xb

    if "featonly" in sections and "dev" in libs.handles:
        # the literal configs[1] forward (features only) in its two builds -- three waves per SIMD with twiddles and window in
        # registers (product) against four waves per SIMD reading them from LDS (dev switch ACIDS_FWD_NOHYB): time, watts, clock.
        # If both sit at the cap, more latency hiding cannot buy time, only fewer Joules can.
        off, sc = mag._affine()
        fo = {}
        call = lambda: ops.stft_mel_forward(x, stft.window[:N_FFT], mag._banded(), "log1p", off, sc, mag._eps, want_spectrum=False)  # noqa: E731
        libs.use("dev")
        for rnd in range(3):
            for name, env in (("3 waves/SIMD, registers", None), ("4 waves/SIMD, LDS tables", "1")):
                if env:
                    os.environ["ACIDS_FWD_NOHYB"] = env
                w = powered_loop(call, seconds=1.0, chunk=20, warm=30)
                os.environ.pop("ACIDS_FWD_NOHYB", None)
                w["microjoules_per_frame"] = w.get("watts_mean", 0) * w["ms_per_launch"] * 1e-3 / frames * 1e6
                fo.setdefault(name, []).append(w)
                print("featonly round %d %-26s %.4f ms %5.0f W sclk %4.0f MHz  %.3f uJ/frame" % (
                    rnd, name, w["ms_per_launch"], w.get("watts_mean", 0), w.get("sclk_mean", 0), w["microjoules_per_frame"]), flush=True)
        libs.use("0")
        res["featonly"] = fo

    if "phase" in sections:
        # the phase side with time, watts and clock: which of these kernels sit on the power cap too
        ph = {}
        Xs = stft(x)
        x_odd = x[:, :CLIP - HOP].contiguous()                  # T = 689: fint_central's odd-T form
        Xo = stft(x_odd)
        inst = ops.phase_scan(Xs, "forward")
        insto = ops.phase_scan(Xo, "forward")
        pol = A.Polar().to(dev)
        pol.scale_data(Xs[:8])
        comp = stft + pol
        pif = A.PolarIF().to(dev)
        pif.scale_data(Xs[:8])
        Yp, Yi = pol(Xs), pif(Xs)
        cases = [
            ("if_forward (scan)", lambda: ops.phase_scan(Xs, "forward"), 12 * 513, T),
            ("if_central (scan)", lambda: ops.phase_scan(Xs, "central"), 12 * 513, T),
            ("if_invert forward T=690", lambda: ops.phase_integrate(inst, "forward"), 8 * 513, T),
            ("if_invert central T=690", lambda: ops.phase_integrate(inst, "central"), 8 * 513, T),
            ("if_invert central T=689", lambda: ops.phase_integrate(insto, "central"), 8 * 513, T - 1),
            ("if_invert backward T=690", lambda: ops.phase_integrate(inst, "backward"), 8 * 513, T),
            ("Polar.forward", lambda: pol(Xs), 16 * 513, T),
            ("Polar.invert", lambda: pol.invert(Yp), 16 * 513, T),
            ("STFT+Polar one kernel", lambda: comp(x), 1024 + 8 * 513, T),
            ("PolarIF.forward", lambda: pif(Xs), 16 * 513, T),
            ("PolarIF.invert", lambda: pif.invert(Yi), 16 * 513, T),
        ]
        for name, fn, bpf, tt in cases:
            w = powered_loop(fn, seconds=1.0, chunk=10, warm=10)
            w["frac_of_8TBps"] = B * tt * bpf / (w["ms_per_launch"] * 1e-3) / 8e12
            ph[name] = w
            print("phase  %-26s %.4f ms  %.3f of 8 TB/s  %5.0f W  sclk %4.0f MHz" % (
                name, w["ms_per_launch"], w["frac_of_8TBps"], w.get("watts_mean", 0), w.get("sclk_mean", 0)), flush=True)
        res["phase"] = ph
        del Xs, Xo, inst, insto, Yp, Yi, x_odd

    if "pat513" in sections:
        # the reference-default chain STFT() + Magnitude(): spectrum + 513 features per frame (7180 B): what do these three
        # streams cost with no arithmetic, feature rows written where they lie or as one stream of 1-KB blocks?
        pl = pattern_lib()
        Xb = torch.empty((B, T, 513), dtype=torch.complex64, device=dev)
        fb = torch.empty((B * T * 513 + 1024,), dtype=torch.float32, device=dev)
        xb = torch.empty((B * T * HOP + 1024,), dtype=torch.float32, device=dev).normal_()
        mag513 = A.Magnitude(sr=SR, n_fft=N_FFT, mode="unipolar", contrast="log1p").to(dev)
        mag513.scale_data(stft(x[:8]))
        out = {}
        for G in (173, 16):
            for style, name in ((1, "rows"), (2, "1-KB blocks")):
                w = powered_loop(lambda: pl.pat_fwd513(L.ptr(xb), L.ptr(Xb), L.ptr(fb), frames, G, 8, style, L.stream_ptr()))
                out["G%d_%s" % (G, name)] = w
                print("pat513  G=%-3d %-12s %.4f ms  %.3f of 8 TB/s  %5.0f W  sclk %4.0f" % (
                    G, name, w["ms_per_launch"], frames * 7180 / (w["ms_per_launch"] * 1e-3) / 8e12, w.get("watts_mean", 0), w.get("sclk_mean", 0)), flush=True)
        w = powered_loop(lambda: mag513.forward_fused(stft, x, return_spectrum=True))
        print("pat513  product fused513     %.4f ms  %.3f of 8 TB/s  %5.0f W  sclk %4.0f" % (
            w["ms_per_launch"], frames * 7180 / (w["ms_per_launch"] * 1e-3) / 8e12, w.get("watts_mean", 0), w.get("sclk_mean", 0)), flush=True)
        out["product"] = w
        res["pat513"] = out
        del Xb, fb, xb

    if "memflavour" in sections:
        # Joules of the step's memory streams by store / load flavour (under a power cap the cheapest stream wins, not the
        # fastest one at full clock): non-temporal vs plain stores, 8- vs 16-byte stores, non-temporal vs plain loads
        pl = pattern_lib()
        Xb = torch.empty((B, T, 513), dtype=torch.complex64, device=dev)
        fb = torch.empty((B, T, 128), dtype=torch.float32, device=dev)
        yb = torch.empty((B, HOP * (T - 1) + 1024), dtype=torch.float32, device=dev)
        xb = torch.empty((B * T * HOP + 1024,), dtype=torch.float32, device=dev).normal_()
        sink = torch.zeros(16, device=dev)
        base = powered_loop(lambda: pl.ek_run(0, 1024, 500, L.ptr(sink), L.stream_ptr()), seconds=0.8, chunk=5, warm=3).get("watts_mean", 0.0)
        mf = {"sleeping_grid_watts": base}
        names = {0: "nt stores", 1: "plain stores", 2: "nt stores, alt", 3: "plain stores, alt"}
        for G in (173, 16):
            for flags in (0, 1, 2, 3):
                f = powered_loop(lambda: pl.pat_fwd_flags(L.ptr(xb), L.ptr(Xb), L.ptr(fb), frames, G, 4, flags, L.stream_ptr()))
                i = powered_loop(lambda: pl.pat_inv_flags(L.ptr(Xb), L.ptr(yb), frames, G, 4, flags, L.stream_ptr()))
                for w in (f, i):
                    w["joules_per_launch_above_sleep"] = (w.get("watts_mean", 0) - base) * w["ms_per_launch"] * 1e-3
                mf["G%d_flags%d" % (G, flags)] = {"fwd_feat": f, "inv": i}
                print("memflavour G=%-3d %-18s fwd(alt = 16-B stores) %.4f ms %5.0f W %.3f J | inv(alt = plain loads) %.4f ms %5.0f W %.3f J" % (
                    G, names[flags], f["ms_per_launch"], f.get("watts_mean", 0), f["joules_per_launch_above_sleep"],
                    i["ms_per_launch"], i.get("watts_mean", 0), i["joules_per_launch_above_sleep"]), flush=True)
        res["memflavour"] = mf
        del Xb, fb, yb, xb

    if "energy" in sections:
        # Joules per wave-instruction: a chip-filling grid (4 waves per SIMD) of one instruction kind, socket power sampled
        pl = pattern_lib()
        sink = torch.zeros(16, device=dev)
        kinds = ["sleep", "v_pk_fma_f32", "v_fma_f32", "v_mov_b32", "v_cndmask_b32", "ds_read_b64", "ds_write_b64", "ds_bpermute_b32",
                 "ds_read_b128", "v_sqrt_f32", "v_pk_add_f32(op_sel)", "s_add_u32"]
        blocks = 256 * 4            # 4 workgroups of 4 waves per CU = 4 waves per SIMD
        en = {}
        idle_w = sampler.window(t_proc0, t_idle1).get("watts_mean", 0.0)
        for kind, name in enumerate(kinds):
            iters = 20000 if kind else 500
            w = powered_loop(lambda: pl.ek_run(kind, blocks, iters, L.ptr(sink), L.stream_ptr()), seconds=1.0, chunk=5, warm=3)
            n_instr = blocks * 4 * iters * (8 if kind == 0 else 64)      # wave-instructions per launch
            w["wave_instr_per_launch"] = n_instr
            w["wave_instr_per_s"] = n_instr / (w["ms_per_launch"] * 1e-3)
            w["instr_per_clk_per_simd"] = w["wave_instr_per_s"] / (1024 * w.get("sclk_mean", 1) * 1e6) if w.get("sclk_mean") else None
            en[name] = w
        base = en["sleep"].get("watts_mean", idle_w)
        for name, w in en.items():
            w["nanojoules_per_wave_instr_above_sleep"] = (w.get("watts_mean", 0) - base) * w["ms_per_launch"] * 1e-3 / w["wave_instr_per_launch"] * 1e9
            print("energy  %-22s %.3f ms %5.0f W sclk %4.0f MHz  %.3f instr/clk/SIMD  %.3f nJ per wave-instruction above the sleeping grid" % (
                name, w["ms_per_launch"], w.get("watts_mean", 0), w.get("sclk_mean", 0), w["instr_per_clk_per_simd"] or 0,
                w["nanojoules_per_wave_instr_above_sleep"]), flush=True)
        res["energy"] = en

    sampler.stop_flag = True
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
