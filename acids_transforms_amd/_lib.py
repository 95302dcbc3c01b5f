"""ctypes binding of libacids_hip.so (the C ABI declared in include/acids_hip.h).

There is no CPU fallback: if the library is missing, or a transform is handed
a tensor that is not on a ROCm device, the call fails loudly.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("ACIDS_HIP_LIB") or os.path.join(_HERE, "libacids_hip.so")   # override: kernel A/B experiments

ABI_VERSION = 4            # what this binding was written against (include/acids_hip.h, at_abi_version())
# return codes (include/acids_hip.h)
AT_OK, AT_EINVAL, AT_EUNSUPPORTED, AT_ENOTINIT, AT_EWORKSPACE, AT_ELAUNCH = 0, -1, -2, -3, -4, -5

c_f = ctypes.c_void_p      # device pointers travel as void*
c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_sz = ctypes.c_size_t
c_flt = ctypes.c_float

# name -> argtypes (restype is int unless listed in _RESTYPES)
_SIGNATURES = {
    "at_abi_version": [],
    "at_error_string": [c_int],
    "at_init": [c_int],
    "at_set_variant": [c_int, c_int],
    "at_get_variant": [c_int],
    "at_stft_forward": [c_f, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_f, c_f, c_f, c_f],
    "at_stft_mel_forward": [c_f, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_f, c_f, c_f, c_f, c_int, c_int,
                            c_f, c_int, c_int, c_f, c_f, c_flt, c_f, c_f, c_f, c_int, c_f],
    "at_stft_polar_forward": [c_f, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_f, c_f, c_f, c_f, c_int, c_int, c_f,
                              c_int, c_f, c_f, c_flt, c_f, c_f, c_f, c_f],
    "at_istft_envelope_table": [c_f, c_int, c_int, c_f, c_f],
    "at_istft_workspace_bytes": [c_i64, c_i64, c_int, c_int],
    "at_istft": [c_f, c_f, c_f, c_i64, c_i64, c_int, c_int, c_f, c_f, c_f, c_f, c_sz, c_f],
    "at_irfft_frames": [c_f, c_f, c_f, c_i64, c_int, c_f, c_f, c_f],
    "at_angle": [c_f, c_i64, c_f, c_f],
    "at_phase_scan": [c_f, c_f, c_i64, c_i64, c_i64, c_int, c_int, c_f, c_f, c_f, c_f, c_f],
    "at_phase_integrate": [c_f, c_i64, c_i64, c_i64, c_int, c_int, c_f, c_f, c_f, c_f],
    "at_polar_to_complex": [c_f, c_f, c_i64, c_f, c_f],
    "at_phase_scan_strided": [c_f, c_f, c_i64, c_i64, c_i64, c_int, c_int, c_f, c_f, c_f, c_f, c_i64, c_f],
    "at_polarif_forward": [c_f, c_i64, c_i64, c_i64, c_int, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i64, c_int, c_f, c_f, c_flt,
                           c_f, c_f],
    "at_phase_integrate_polar": [c_f, c_i64, c_i64, c_i64, c_i64, c_int, c_f, c_f, c_f, c_f, c_f],
    "at_cartesian_pack": [c_f, c_i64, c_int, c_f, c_f, c_f, c_f, c_f, c_f],
    "at_cartesian_unpack": [c_f, c_i64, c_int, c_f, c_f, c_f, c_f, c_f, c_f],
    "at_resample_sinc": [c_f, c_i64, c_i64, c_int, c_int, c_int, c_f, c_i64, c_f, c_f],
    "at_sinebank_workspace_bytes": [c_i64, c_int, c_i64, c_int],
    "at_sinebank_offline": [c_f, c_i64, c_i64, c_int, c_f, c_f, c_f, c_i64, c_int, c_f, c_f, c_f, c_f, c_f, c_sz, c_f],
    "at_sinebank_realtime": [c_f, c_i64, c_int, c_int, c_int, c_f, c_f, c_f, c_f, c_f, c_f],
    "at_mel_project": [c_f, c_int, c_i64, c_i64, c_int, c_f, c_int, c_int, c_int, c_int, c_f, c_f, c_flt, c_f, c_i64,
                       c_i64, c_f],
    "at_mel_project_banded": [c_f, c_int, c_i64, c_i64, c_int, c_f, c_f, c_f, c_int, c_int, c_f, c_int, c_int, c_f, c_f,
                              c_flt, c_f, c_i64, c_i64, c_f, c_i64, c_f, c_f, c_f, c_f],
    "at_mel_bf16_bank_bytes": [c_int, c_int],
    "at_mel_bf16_pack_bank": [c_f, c_int, c_int, c_int, c_f, c_f],
    "at_mel_project_bf16": [c_f, c_int, c_i64, c_i64, c_int, c_f, c_int, c_int, c_f, c_f, c_flt, c_f, c_i64, c_f],
    "at_project_small": [c_f, c_i64, c_int, c_f, c_int, c_f, c_f, c_f, c_i64, c_f],
    "at_mag_pointwise": [c_f, c_int, c_i64, c_int, c_int, c_f, c_f, c_flt, c_f, c_f],
    "at_stats_workspace_bytes": [],
    "at_stats": [c_f, c_int, c_i64, c_int, c_flt, c_f, c_f, c_sz, c_f],
    "at_affine": [c_f, c_i64, c_f, c_f, c_int, c_f, c_f],
    "at_pghi_gradients": [c_f, c_i64, c_int, c_int, c_flt, c_int, c_int, c_flt, c_f, c_f, c_f, c_f],
    "at_pghi_offline_workspace_bytes": [c_i64, c_int, c_int],
    "at_pghi_offline": [c_f, c_i64, c_int, c_int, c_flt, c_int, c_int, c_flt, c_flt, c_f, c_f, c_sz, c_f, c_f, c_f],
    "at_pghi_integrate": [c_f, c_f, c_f, c_i64, c_int, c_int, c_flt, c_flt, c_f, c_f, c_sz, c_f, c_f, c_f],
    "at_pghi_rt_workspace_bytes": [c_int, c_int, c_int],
    "at_pghi_realtime": [c_f, c_f, c_f, c_f, c_int, c_int, c_int, c_flt, c_int, c_int, c_flt, c_flt, c_f, c_f, c_f,
                         c_f, c_sz, c_f],
    "at_pghi_realtime_seeded": [c_f, c_f, c_f, c_f, c_int, c_int, c_int, c_flt, c_int, c_int, c_flt, c_flt, c_f, c_f, c_sz,
                                c_f],
    "at_rt_update_buffers": [c_f, c_f, c_int, c_int, c_int, c_f, c_f, c_f, c_f],
    "at_griffinlim_update": [c_f, c_f, c_f, c_flt, c_i64, c_f, c_f],
    "at_istft_griffinlim": [c_f, c_f, c_f, c_flt, c_i64, c_i64, c_int, c_int, c_f, c_f, c_f, c_f],
    "at_scale_complex": [c_f, c_f, c_i64, c_f, c_f],
    "at_oadd_forward": [c_f, c_f, c_int, c_i64, c_int, c_i64, c_f, c_f, c_f],
    "at_oadd_invert": [c_f, c_f, c_int, c_int, c_int, c_int, c_int, c_f, c_f, c_f, c_f],
    "at_oadd_push": [c_f, c_int, c_i64, c_int, c_i64, c_f, c_f],
    "at_mulaw_encode": [c_f, c_i64, c_int, c_f, c_f],
    "at_mulaw_decode": [c_f, c_f, c_i64, c_int, c_f, c_f],
    "at_onehot": [c_f, c_i64, c_int, c_i64, c_f, c_f],
    "at_argmax_last": [c_f, c_f, c_i64, c_int, c_f, c_f],
}
_RESTYPES = {"at_error_string": ctypes.c_char_p, "at_istft_workspace_bytes": c_sz, "at_stats_workspace_bytes": c_sz,
             "at_pghi_offline_workspace_bytes": c_sz, "at_mel_bf16_bank_bytes": c_sz, "at_pghi_rt_workspace_bytes": c_sz,
             "at_sinebank_workspace_bytes": c_sz}


class AcidsHipError(RuntimeError):
    pass


def build(verbose=False):
    """Compile every HIP source for gfx950 into libacids_hip.so (in-tree)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"]
    if not verbose:
        cmd.append("-s")
    subprocess.check_call(cmd)
    return _SO


_lib = None
_inited = set()


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise AcidsHipError(
                "libacids_hip.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C acids_transforms_amd/csrc`. There is no CPU fallback." % _SO)
        L = ctypes.CDLL(_SO)
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(L, name)   # AttributeError if the .so is stale: fail loudly
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, c_int)
        if L.at_abi_version() != ABI_VERSION:
            raise AcidsHipError("%s exports ABI version %d, this binding needs %d: rebuild it (make -C acids_transforms_amd/csrc)"
                                % (_SO, L.at_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


# kernel variants (include/acids_hip.h AT_VARIANT_*): same results, different kernels; for tests and A/B runs
VARIANTS = {"epilogue": 0, "frame_kernels": 1, "small_projection": 2, "scan_layout": 3, "pghi_kernel": 4, "istft_runs": 5}


class variant:
    """`with variant("epilogue", 1): ...` -- force a kernel variant for the calls inside, restore it afterwards.
    Process-wide (the table lives in the library), so not for concurrent use from several threads."""

    def __init__(self, name, value):
        self.which, self.value = VARIANTS[name], int(value)

    def __enter__(self):
        self.prev = lib().at_get_variant(self.which)
        check(lib().at_set_variant(self.which, self.value), "at_set_variant")
        return self

    def __exit__(self, *exc):
        check(lib().at_set_variant(self.which, self.prev), "at_set_variant")
        return False


def exported_symbols():
    return sorted(_SIGNATURES)


def check(rc, what):
    if rc != 0:
        raise AcidsHipError("%s failed: %s (%d)" % (what, lib().at_error_string(rc).decode(), rc))


def require_device(*tensors):
    """Product path guard: HIP kernels only, never a silent CPU route; all operands on ONE device, and that
    device must be the current one (ops.py enters it around every op: the kernels are launched on the current
    stream of the current device and capi.hip picks its per-device tables with hipGetDevice())."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise AcidsHipError(
                "acids_transforms_amd runs on MI355X only: got a %s tensor. Move inputs (and the module, "
                "`.to('cuda')`) to the ROCm device; there is no CPU fallback." % t.device)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise AcidsHipError("operands live on different devices (%s and %s): move them to one device"
                                % (dev, t.device))
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx != torch.cuda.current_device():
        raise AcidsHipError("operands are on cuda:%d but the current device is cuda:%d (call through "
                            "acids_transforms_amd.ops, which enters the operands' device)" % (idx, torch.cuda.current_device()))
    if idx not in _inited:
        with torch.cuda.device(idx):
            check(lib().at_init(idx), "at_init")
        _inited.add(idx)
    return idx


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """The current HIP stream of the current device as a void*.  torch's raw-stream accessor when it exists (0.3 us;
    `torch.cuda.current_stream()` builds a Stream object first: 8 us of the 16 a call through ops.py costs)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def first_device_index(args, kwargs):
    """Index of the first ROCm tensor among the arguments (None if there is none)."""
    for a in list(args) + list(kwargs.values()):
        if isinstance(a, torch.Tensor) and a.is_cuda:
            return a.device.index
    return None


def device_scoped(fn):
    """Run `fn` with the device of its first ROCm tensor argument as the current device, so that
    `stream_ptr()` and the library's hipGetDevice() both see the operands' device (a module on cuda:1 used
    while cuda:0 is current would otherwise launch on device 0 with device-1 pointers)."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        idx = first_device_index(args, kwargs)
        if idx is None or idx == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(idx):
            return fn(*args, **kwargs)
    return wrapped
