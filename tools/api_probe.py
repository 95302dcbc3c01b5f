"""Dev timing: the public transforms at sizes other than the headline one, to spot paths that fall off the
bandwidth-bound kernels.  Prints ms per call over a 1024-clip x 4-s batch and the effective TB/s of the bytes the op
has to move (input + output once)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import acids_transforms_amd as A

dev = torch.device("cuda:0")
B, L = 1024, 176400
x = torch.randn(B, L, device=dev) * 0.1


def timeit(fn, n=4, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def nbytes(*ts):
    return sum(t.numel() * t.element_size() for t in ts)


def report(name, fn, *io):
    t = timeit(fn)
    print("%-58s %9.3f ms  %6.2f TB/s" % (name, t, nbytes(*io) / t / 1e9), flush=True)


which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["stft", "mel", "phase", "dgt"]

if "stft" in which:
    for n_fft, hop in [(400, 160), (1000, 250), (1024, 64), (1024, 100), (1024, 441), (1536, 384), (2000, 500),
                       (2048, 256), (4096, 1024), (4096, 512), (8192, 2048)]:
        st = A.STFT(n_fft=n_fft, hop_length=hop).to(dev)
        xs = x[:256] if hop < 128 else x
        X = st(xs)
        report("STFT(%d, hop %d) forward [%d clips]" % (n_fft, hop, xs.shape[0]), lambda: st(xs), xs, X)
        y = st.invert(X)
        report("STFT(%d, hop %d) invert" % (n_fft, hop), lambda: st.invert(X), X, y)
        del X, y

if "mel" in which:
    for n_fft, hop, n_mels in [(2048, 512, 128), (2048, 512, 80), (4096, 1024, 128), (512, 128, 64), (400, 160, 40)]:
        ms = A.MFCC(n_fft=n_fft, hop_length=hop, n_mels=n_mels).to(dev)
        y = ms(x)
        report("MFCC(%d, hop %d, %d mels) [mel spectrogram]" % (n_fft, hop, n_mels), lambda: ms(x), x, y)
        mf = A.MFCC(n_fft=n_fft, hop_length=hop, n_mels=n_mels, n_mfcc=min(40, n_mels)).to(dev)
        y = mf(x)
        report("MFCC(%d, hop %d, %d mels, %d coeffs)" % (n_fft, hop, n_mels, min(40, n_mels)), lambda: mf(x), x, y)

if "phase" in which:
    for n_fft in (512, 2048, 4096):
        st = A.STFT(n_fft=n_fft, hop_length=n_fft // 4).to(dev)
        X = st(x)
        for name, tr in [("Phase", A.Phase()), ("Phase(unwrap)", A.Phase(unwrap=True)), ("IF", A.IF()),
                         ("Real", A.Real()), ("Imaginary", A.Imaginary()),
                         ("Polar", A.Polar(magnitude_args={"mode": "bipolar", "n_fft": n_fft})),
                         ("PolarIF", A.PolarIF(magnitude_args={"mode": "bipolar", "n_fft": n_fft})),
                         ("Cartesian", A.Cartesian())]:
            tr = tr.to(dev)
            try:
                tr.scale_data(X)
                y = tr(X)
            except Exception as exc:       # a size the transform does not take
                print("%-58s %s" % ("%s @ n_fft %d" % (name, n_fft), type(exc).__name__ + ": " + str(exc)[:80]))
                continue
            report("%s @ n_fft %d forward" % (name, n_fft), lambda: tr(X), X, y)
            Xi = tr.invert(y)
            report("%s @ n_fft %d invert" % (name, n_fft), lambda: tr.invert(y), y, Xi)
            del y, Xi
        del X

if "dgt" in which:
    for n_fft, hop in [(512, 128), (2048, 512), (4096, 1024)]:
        dg = A.DGT(n_fft=n_fft, hop_length=hop).to(dev)
        X = dg(x)
        report("DGT(%d, hop %d) forward" % (n_fft, hop), lambda: dg(x), x, X)
        y = dg.invert(X)
        report("DGT(%d, hop %d) invert" % (n_fft, hop), lambda: dg.invert(X), X, y)
        xs = x[:64]
        Xs = dg(xs)
        mag = Xs.abs()
        for mode in ("pghi", "griffin_lim"):
            try:
                t = timeit(lambda: dg.invert(mag, inversion_mode=mode), n=1, warm=1)
                print("%-58s %9.3f ms" % ("DGT(%d, hop %d).invert(|X|, %s) [64 clips]" % (n_fft, hop, mode), t), flush=True)
            except Exception as exc:
                print("%-58s %s" % ("DGT(%d).invert %s" % (n_fft, mode), type(exc).__name__ + ": " + str(exc)[:80]))
        del X, y, Xs, mag
