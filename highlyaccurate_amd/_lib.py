"""ctypes binding of libhla.so (the C ABI declared in include/hla.h).

The product path has no fallback: if the HIP library cannot be loaded, or a tensor is not
on a HIP device, calls raise.  PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('HLA_LIB') or os.path.join(HERE, 'libhla.so')   # HLA_LIB: experiment builds

HLA_F32, HLA_BF16, HLA_F16, HLA_F16X3 = 0, 1, 2, 3
HLA_VGG_WANT_CONF, HLA_VGG_DEFER_NORM, HLA_VGG_SAVE_FOR_BACKWARD, HLA_VGG_FEAT16 = 1, 2, 4, 8
HLA_VGG_BWD_SCALE_INVARIANT = 1
HLA_VGG_BWD_DENSE = 2
HLA_VGG_BWD_WGRAD_TWO_PHASE = 4
HLA_VGG_BWD_WGRAD0_UNFUSED = 8
ABI_VERSION = 21


class HlaError(RuntimeError):
    pass


class VggParams(C.Structure):
    _fields_ = [('w', C.c_void_p * 17), ('b', C.c_void_p * 7)]


class S2GLevel(C.Structure):
    _fields_ = [('sat_feat', C.c_void_p), ('grd_feat', C.c_void_p), ('grd_conf', C.c_void_p), ('xyz', C.c_void_p),
                ('sat_inv_norm', C.c_void_p), ('grd_inv_norm', C.c_void_p),
                ('A', C.c_int), ('h', C.c_int), ('w', C.c_int), ('C', C.c_int), ('row0', C.c_int), ('grd_row_skip', C.c_int),
                ('meter_per_pixel', C.c_double), ('centre', C.c_double), ('feat_dtype', C.c_int)]


class S2GConfig(C.Structure):
    _fields_ = [('ford', C.c_int), ('n_levels', C.c_int), ('n_iters', C.c_int), ('level_first', C.c_int),
                ('using_weight', C.c_int), ('use_hessian', C.c_int), ('dof', C.c_int),
                ('shift_range_lat', C.c_double), ('shift_range_lon', C.c_double), ('rotation_range', C.c_double),
                ('damping', C.c_double * 3), ('keep', C.c_void_p), ('keep_stride', C.c_size_t),
                ('optimizer', C.c_int), ('beta1', C.c_double), ('beta2', C.c_double), ('count_in_view', C.c_int),
                ('grd_grad_overwrite', C.c_int), ('deterministic', C.c_int)]


class FillRegion(C.Structure):
    _fields_ = [('ptr', C.c_void_p), ('chunk_bytes', C.c_size_t), ('stride_bytes', C.c_size_t), ('n_chunks', C.c_int)]


class VggGrads(C.Structure):
    _fields_ = [('dw', C.c_void_p * 17), ('db', C.c_void_p * 7)]


class S2GLevelGrad(C.Structure):
    _fields_ = [('d_sat_feat', C.c_void_p), ('d_grd_feat', C.c_void_p), ('d_grd_conf', C.c_void_p)]


class ProfRecord(C.Structure):
    _fields_ = [('kernel_id', C.c_int), ('ms', C.c_float), ('flops', C.c_double), ('bytes', C.c_double)]


HLA_POSE_LOSS_F32, HLA_POSE_LOSS_F64 = 0, 1


class PoseLossArgs(C.Structure):
    _fields_ = [('x', C.c_void_p * 3), ('x_stride', (C.c_longlong * 3) * 3), ('gt', C.c_void_p * 3), ('gt_stride', C.c_longlong * 3),
                ('coe', C.c_double * 3), ('B', C.c_int), ('N', C.c_int), ('L', C.c_int), ('gt_dtype', C.c_int)]


_lib = None


def _check_binary(lib: C.CDLL, path: str) -> None:
    """Refuse a library that does not match this binding: ABI version, then the size of every struct that crosses the
    boundary (ctypes structs are positional, so a mismatch would corrupt memory silently)."""
    lib.hla_abi_version.restype = C.c_int
    have = lib.hla_abi_version()
    if have != ABI_VERSION:
        raise HlaError(f'{path}: ABI version {have}, this binding needs {ABI_VERSION}; rebuild it '
                       f'(python -m highlyaccurate_amd.build --force)')
    try:
        fn = lib.hla_sizeof_struct
    except AttributeError:
        raise HlaError(f'{path}: no hla_sizeof_struct export; rebuild it') from None
    fn.restype, fn.argtypes = C.c_size_t, [C.c_int]
    for sid, cls in enumerate((VggParams, VggGrads, S2GLevel, S2GConfig, S2GLevelGrad, ProfRecord, PoseLossArgs, FillRegion)):
        if fn(sid) != C.sizeof(cls):
            raise HlaError(f'{path}: sizeof({cls.__name__}) is {fn(sid)} in the library and {C.sizeof(cls)} in the binding')


def load() -> C.CDLL:
    """Load libhla.so.  The binary must have been built from the sources next to it (content hash baked in at build
    time): a missing or stale one is rebuilt with hipcc when a compiler is present and refused otherwise.  Then the ABI
    version and struct sizes are compared with this binding's."""
    global _lib
    if _lib is not None:
        return _lib
    from . import build as _b
    try:
        want = _b.source_hash()
    except _b.SourcesMissing as e:
        # a prebuilt library deployed without its sources: nothing to compare its content with (and nothing to rebuild from);
        # the ABI-version and struct-size checks below still apply
        if not os.path.exists(LIB_PATH):
            raise HlaError(f'{LIB_PATH} is missing and so are its sources: {e}') from None
        want = _b.lib_hash(LIB_PATH)
    if _b.lib_hash(LIB_PATH) != want and not (os.environ.get('HLA_LIB') and os.environ.get('HLA_ALLOW_STALE') == '1'):
        # (HLA_LIB + HLA_ALLOW_STALE=1: A/B timing of an older experiment build against the current one, tools/ab_libs.py)
        if os.environ.get('HLA_LIB'):
            raise HlaError(f'{LIB_PATH} (HLA_LIB) was not built from the sources in {_b.CSRC}; rebuild it')
        if not _b.have_compiler():
            what = 'is missing' if not os.path.exists(LIB_PATH) else 'is stale (built from different sources)'
            raise HlaError(f'{LIB_PATH} {what} and there is no hipcc to rebuild it; highlyaccurate_amd has no fallback path')
        _b.build()          # takes a file lock and re-checks: concurrent ranks build once
    lib = C.CDLL(LIB_PATH)
    _check_binary(lib, LIB_PATH)
    vp, i, sz = C.c_void_p, C.c_int, C.c_size_t
    lib.hla_last_error.restype = C.c_char_p
    lib.hla_source_hash.restype = C.c_char_p
    lib.hla_vgg_workspace_bytes.restype = sz
    lib.hla_vgg_workspace_bytes.argtypes = [i, i, i, i, i]
    lib.hla_vgg_forward.restype = i
    lib.hla_vgg_forward.argtypes = [vp, sz, C.POINTER(VggParams), vp, C.POINTER(vp), C.POINTER(vp), vp, vp, sz,
                                    i, i, i, i, i, i, i, vp]
    lib.hla_vgg_packed_weight_bytes.restype = sz
    lib.hla_vgg_packed_weight_bytes.argtypes = [i]
    lib.hla_vgg_pack_weights.restype = i
    lib.hla_vgg_pack_weights.argtypes = [C.POINTER(VggParams), vp, i, vp]
    lib.hla_vgg_packed_weight_T_bytes.restype = sz
    lib.hla_vgg_packed_weight_T_bytes.argtypes = [i]
    lib.hla_vgg_pack_weights_T.restype = i
    lib.hla_vgg_pack_weights_T.argtypes = [C.POINTER(VggParams), vp, i, vp]
    lib.hla_vgg_bwd_workspace_bytes.restype = sz
    lib.hla_vgg_bwd_workspace_bytes.argtypes = [i, i, i, i, i]
    lib.hla_vgg_backward_live_tiles.restype = i
    lib.hla_vgg_backward_live_tiles.argtypes = [vp, i, i, i, i, i, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    lib.hla_vgg_backward.restype = i
    lib.hla_vgg_backward.argtypes = [vp, sz, C.POINTER(VggParams), vp, vp, C.POINTER(vp), vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(VggGrads),
                                     vp, sz, i, i, i, i, i, i, i, vp]
    lib.hla_resize_bilinear.restype = i
    lib.hla_resize_bilinear.argtypes = [vp, vp, vp, i, vp, vp, i, vp, vp, i, i, i, i, i, vp]
    lib.hla_sat_tile.restype = i
    lib.hla_sat_tile.argtypes = [vp, vp, vp, i, i, i, vp]
    lib.hla_grid_sample.restype = i
    lib.hla_grid_sample.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, i, i, vp]
    lib.hla_s2g_workspace_bytes.restype = sz
    lib.hla_s2g_workspace_bytes.argtypes = [C.POINTER(S2GConfig), C.POINTER(S2GLevel), i]
    lib.hla_g2s_workspace_bytes.restype = sz
    lib.hla_g2s_workspace_bytes.argtypes = [C.POINTER(S2GConfig), C.POINTER(S2GLevel), i]
    lib.hla_g2s_lm_solve.restype = i
    lib.hla_g2s_lm_solve.argtypes = [C.POINTER(S2GConfig), C.POINTER(S2GLevel), vp, i, i, vp, vp, vp, vp, sz, i, vp]
    lib.hla_g2s_bwd_workspace_bytes.restype = sz
    lib.hla_g2s_bwd_workspace_bytes.argtypes = [C.POINTER(S2GConfig), C.POINTER(S2GLevel), i]
    lib.hla_g2s_lm_solve_bwd.restype = i
    lib.hla_g2s_lm_solve_bwd.argtypes = [C.POINTER(S2GConfig), C.POINTER(S2GLevel), C.POINTER(S2GLevelGrad), vp, i, i, vp, vp,
                                         vp, vp, vp, vp, sz, i, vp]
    lib.hla_s2g_lm_solve.restype = i
    lib.hla_s2g_lm_solve.argtypes = [C.POINTER(S2GConfig), C.POINTER(S2GLevel), vp, vp, vp, vp, vp, vp, vp, sz, i, vp]
    lib.hla_s2g_bwd_workspace_bytes.restype = sz
    lib.hla_s2g_bwd_workspace_bytes.argtypes = [C.POINTER(S2GConfig), C.POINTER(S2GLevel), i]
    lib.hla_s2g_lm_solve_bwd.restype = i
    lib.hla_s2g_lm_solve_bwd.argtypes = [C.POINTER(S2GConfig), C.POINTER(S2GLevel), C.POINTER(S2GLevelGrad), vp, vp, vp, vp,
                                         vp, vp, vp, vp, sz, i, vp]
    lib.hla_zero_fill.restype = i
    lib.hla_zero_fill.argtypes = [C.POINTER(FillRegion), i, i, vp]
    lib.hla_pose_loss.restype = i
    lib.hla_pose_loss.argtypes = [C.POINTER(PoseLossArgs), vp, vp]
    lib.hla_pose_loss_bwd.restype = i
    lib.hla_pose_loss_bwd.argtypes = [C.POINTER(PoseLossArgs), C.POINTER(vp), C.POINTER(vp), C.POINTER((C.c_longlong * 3) * 3), vp]
    lib.hla_prof_enable.restype = i
    lib.hla_prof_enable.argtypes = [i]
    lib.hla_prof_kernel_name.restype = C.c_char_p
    lib.hla_prof_kernel_name.argtypes = [i]
    lib.hla_prof_mfma_peak.restype = i
    lib.hla_prof_mfma_peak.argtypes = [i, i, C.c_float, C.POINTER(C.c_float), vp]
    lib.hla_prof_fetch.restype = i
    lib.hla_prof_fetch.argtypes = [C.POINTER(ProfRecord), i, C.POINTER(i)]
    _lib = lib
    return lib


def zero_fill(regions, max_blocks: int = 0, stream=None) -> None:
    """``hla_zero_fill``: regions = [(tensor_or_ptr, chunk_bytes, stride_bytes, n_chunks)] (<= 16), cleared by ONE launch on the
    current stream (``max_blocks`` > 0: a background fill with that many workgroups per region).  A tensor stands for its data pointer (it must stay alive until the launch has run, as for any kernel)."""
    if not regions:
        return
    arr = (FillRegion * len(regions))()
    for k, (t, cb, sb, n) in enumerate(regions):
        arr[k].ptr = t.data_ptr() if hasattr(t, 'data_ptr') else int(t)
        arr[k].chunk_bytes, arr[k].stride_bytes, arr[k].n_chunks = int(cb), int(sb), int(n)
    check(load().hla_zero_fill(arr, len(regions), int(max_blocks), stream_ptr() if stream is None else stream), 'hla_zero_fill')


def prof_enable(on: bool) -> None:
    load().hla_prof_enable(1 if on else 0)


def prof_fetch(max_records: int = 1 << 16):
    """Synchronise and return [(kernel_name, ms, flops, bytes)] for every launch since the last fetch."""
    lib = load()
    buf = (ProfRecord * max_records)()
    n = C.c_int(0)
    lib.hla_prof_fetch(buf, max_records, C.byref(n))
    return [(lib.hla_prof_kernel_name(buf[k].kernel_id).decode(), buf[k].ms, buf[k].flops, buf[k].bytes)
            for k in range(n.value)]


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise HlaError(f'{what} failed (status {rc}): {load().hla_last_error().decode()}')


def stream_ptr() -> C.c_void_p:
    """The CURRENT device's current stream.  The C side launches on whatever device is current, so every Python entry point
    that reaches it is wrapped in ``on_device`` (below), which makes the tensors' device current first."""
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def on_device(pick):
    """Decorator: run ``fn`` with the device of ``pick(*args, **kwargs)`` (a tensor) current, so that ``stream_ptr()``, the
    workspaces allocated inside and the kernels all belong to the device the data lives on -- a model moved to 'cuda:1'
    without ``torch.cuda.set_device(1)`` would otherwise launch on device 0's stream against device-1 pointers."""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrapper(*a, **k):
            t = pick(*a, **k)
            if t is None or not t.is_cuda:
                return fn(*a, **k)            # fn raises its own "no CPU path" error
            if torch.cuda.current_device() == t.device.index:
                return fn(*a, **k)
            with torch.cuda.device(t.device):
                return fn(*a, **k)
        return wrapper
    return deco


def same_device(*named) -> None:
    """named: (name, tensor-or-None) pairs; all tensors must live on one HIP device."""
    devs = {(n, str(t.device)) for n, t in named if t is not None}
    if len({d for _, d in devs}) > 1:
        raise ValueError('all inputs and the model must be on one device, got ' + ', '.join(f'{n}: {d}' for n, d in sorted(devs)))


def require_gpu(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise HlaError(f'{name} must live on the HIP device (got {t.device}); highlyaccurate_amd has no CPU path')


def ptr(t) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def mfma_sustained_tflops(dtype_code: int, data: int, ms_target: float = 8.0) -> float:
    """TFLOP/s the matrix pipe of the current device sustains on register-resident operands (include/hla.h, hla_prof_mfma_peak):
    data 0 = zeros, 1 = random, 2 = random with half the elements zero."""
    out = C.c_float(0.0)
    check(load().hla_prof_mfma_peak(dtype_code, data, ms_target, C.byref(out), stream_ptr()), 'hla_prof_mfma_peak')
    return float(out.value)
