"""VGGUnet -- same module surface as the reference's ``VGG.py:13-203`` (parameter names, ctor
argument, ``forward(x) -> ([feat_l], [conf_l])``), computed by libhla's MFMA convolution kernels.

Differences a caller can observe:
  * returned maps are logically [B,C,H,W] but stored channels-last (NHWC); values match the reference
  * ``precision='fp32'`` (default) runs exact-fp32 MFMA; ``'fp16x3'`` (split fp16: hi+lo operands, three fp16 MFMAs per
    product) gives fp32-class results -- it passes the same parity gates -- at about three times the speed;
    ``'bf16'`` / ``'fp16'`` are the reduced-precision throughput modes
  * pretrained torchvision weights are not downloaded here: load a state dict (keys are identical)
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib

_W_ORDER = ['conv0', 'conv2', 'conv5', 'conv7', 'conv10', 'conv12', 'conv14',
            'conv_dec1.1', 'conv_dec1.3', 'conv_dec2.1', 'conv_dec2.3', 'conv_dec3.1', 'conv_dec3.3',
            'conf0.1', 'conf1.1', 'conf2.1', 'conf3.1']
_LEVEL_SEL = {-1: [0], -2: [1], -3: [2], 2: [1, 2], 3: [0, 1, 2], 4: [0, 1, 2, 3]}
_CH = (256, 128, 64, 16)


def _dtype_code(precision: str) -> int:
    if precision == 'bf16':
        return _lib.HLA_BF16
    if precision == 'fp16':
        return _lib.HLA_F16
    if precision == 'fp32':
        return _lib.HLA_F32
    if precision == 'fp16x3':
        return _lib.HLA_F16X3
    raise ValueError(f"precision must be 'fp32', 'fp16x3', 'bf16' or 'fp16', got {precision!r}")


def _param_table(module: 'VGGUnet'):
    sd = dict(module.named_parameters())
    prm = _lib.VggParams()
    keep, versions = [], []
    pad = {11: (64, 128), 12: (64, 64), 16: (1, 64)}       # level 4: conv_dec3.1/3 and conf3 run zero-padded to 64 channels
    for i, name in enumerate(_W_ORDER):
        w = sd[name + '.weight']
        versions.append((w.data_ptr(), w._version))
        w = w.detach().contiguous().float()
        if i in pad:
            if module.level != 4:
                continue
            wp = torch.zeros(pad[i][0], pad[i][1], 3, 3, device=w.device, dtype=torch.float32)
            wp[:w.shape[0], :w.shape[1]] = w
            w = wp
        keep.append(w)
        prm.w[i] = w.data_ptr()
        if i < 7:
            b = sd[name + '.bias']
            if i == 0:          # conv0's bias is part of the packed fragments (include/hla.h): a change of it alone repacks
                versions.append((b.data_ptr(), b._version))
            b = b.detach().contiguous().float()
            keep.append(b)
            prm.b[i] = b.data_ptr()
    return prm, keep, tuple(versions)


def _packed_weights(module: 'VGGUnet', prm, versions, dt: int, device):
    """MFMA-fragment-ordered copy of the conv weights, rebuilt only when a parameter changed
    (tensor version counters / storage pointers), e.g. after an optimizer step or load_state_dict."""
    key = (dt, str(device), versions)
    cache = module.__dict__.setdefault('_hla_packed', {})
    if cache.get('key') != key:
        lib = _lib.load()
        buf = cache.get('buf')
        nbytes = lib.hla_vgg_packed_weight_bytes(dt)
        if buf is None or buf.numel() != nbytes or buf.device != device:
            buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _lib.check(lib.hla_vgg_pack_weights(C.byref(prm), _lib.ptr(buf), dt, _lib.stream_ptr()), 'hla_vgg_pack_weights')
        cache['key'], cache['buf'] = key, buf
    return cache['buf']


def _image_window(x: torch.Tensor):
    """(x, x_plane) for the C side: an fp32 NCHW image whose rows are W apart and whose samples are 3 planes apart is passed as
    it lies in memory -- in particular ``img[:, :, r0:, :]``, a window of rows of a taller image (x_plane = the full plane), is
    NOT copied (mode='test' crops the ground image that way on every call: 66 MB per step at B = 32).  Anything else is made dense."""
    x = x.float()
    B, _, H, W = x.shape
    sb, sc, sh, sw = x.stride()
    if not (sw == 1 and sh == W and sc >= H * W and (B == 1 or sb == 3 * sc)):      # (a size-1 batch dimension may carry any stride)
        x = x.contiguous()
        sc = H * W
    return x, sc


@_lib.on_device(lambda module, x, *a, **k: x)
def vgg_forward_nhwc(module: 'VGGUnet', x: torch.Tensor, want_conf: bool = True, defer_norm: bool = False,
                     save_for_backward: bool = False, first_row8: int = 0, feat16: bool = False):
    """Run the three-level extractor.  Returns (feats, confs, inv_norm): lists of NHWC fp32 tensors
    [B,h,w,C] and [B,h,w] (or None), and inv_norm [3,B] fp64 = 1/max(||map||, 1e-12).
    With ``defer_norm`` the maps are left un-normalised (the LM loop folds inv_norm into its sums);
    otherwise they are L2-normalised per sample like the reference's (VGG.py:172-175).
    ``first_row8`` = f > 0 promises that only rows f / 2f / 4f.. of the three maps will be read (include/hla.h): the layers
    skip the rows nothing depends on, the rows above stay unwritten and inv_norm covers the computed rows only.
    ``feat16`` (bf16 / fp16 precision, ``defer_norm`` only, inference only): the raw maps are returned in the 16-bit activation
    type instead of fp32 -- half the bytes through the HBM-bound LM loop, whose arithmetic stays fp32 / fp64."""
    _lib.require_gpu(x, 'VGGUnet input')
    if x.dim() != 4 or x.shape[1] != 3:
        raise ValueError(f'expected [B,3,H,W], got {tuple(x.shape)}')
    lib = _lib.load()
    x, x_plane = _image_window(x)
    B, _, H, W = x.shape
    _lib.same_device(('input', x), ('parameters', module.conv0.weight))
    dt = _dtype_code(module.precision)
    prm, keep, versions = _param_table(module)
    packed = _packed_weights(module, prm, versions, dt, x.device)
    L = 4 if module.level == 4 else 3
    fdt = torch.float32
    if feat16:
        if not (defer_norm and not save_for_backward and L == 3 and dt in (_lib.HLA_BF16, _lib.HLA_F16)):
            raise ValueError("feat16 needs precision 'bf16' / 'fp16', defer_norm=True, level 3 and no save_for_backward")
        fdt = torch.float16         # (also in bf16 mode: the raw maps are written as fp16, saturating)
    # level 4: x24 is stored with 64 channels, the 16 real ones first, zeros behind them (see include/hla.h)
    feats = [torch.empty(B, H >> (3 - l), W >> (3 - l), 64 if l == 3 else _CH[l], device=x.device, dtype=fdt)
             for l in range(L)]
    confs = [torch.empty(B, H >> (3 - l), W >> (3 - l), device=x.device, dtype=torch.float32) if want_conf else None
             for l in range(L)]
    inv_norm = torch.empty(L, B, device=x.device, dtype=torch.float64)
    fp = (C.c_void_p * 4)(*([f.data_ptr() for f in feats] + [0] * (4 - L)))
    cp = (C.c_void_p * 4)(*([(c.data_ptr() if c is not None else 0) for c in confs] + [0] * (4 - L)))
    nbytes = lib.hla_vgg_workspace_bytes(B, H, W, L, dt)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    flags = (_lib.HLA_VGG_WANT_CONF if want_conf else 0) | (_lib.HLA_VGG_DEFER_NORM if defer_norm else 0) | \
            (_lib.HLA_VGG_FEAT16 if feat16 else 0)
    if save_for_backward:
        if not defer_norm:
            raise ValueError('save_for_backward needs defer_norm=True (the backward works on the raw maps)')
        flags |= _lib.HLA_VGG_SAVE_FOR_BACKWARD
    rc = lib.hla_vgg_forward(_lib.ptr(x), x_plane, C.byref(prm), _lib.ptr(packed), fp, cp, _lib.ptr(inv_norm), _lib.ptr(ws), nbytes,
                             B, H, W, L, dt, flags, int(first_row8), _lib.stream_ptr())
    _lib.check(rc, 'hla_vgg_forward')
    # ws is only used by work already enqueued on this stream; the caching allocator keeps the block
    # stream-ordered, so dropping the Python reference here is safe.
    if save_for_backward:
        # The backward reads x again (conv0's weight gradient) and x may be a window of the caller's own image storage: like
        # autograd's saved tensors, remember its version counter -- an in-place change before backward() raises there instead of
        # silently producing the gradient of another image.
        return feats, confs, inv_norm, dict(x=x, x_version=x._version, x_plane=x_plane, ws=ws, feats=feats, inv_norm=inv_norm, dt=dt, L=L)
    return feats, confs, inv_norm


@_lib.on_device(lambda module, ctx, *a, **k: ctx['x'])
def vgg_backward_nhwc(module: 'VGGUnet', ctx: dict, d_feats, confs=None, d_confs=None, scale_invariant: bool = False,
                      first_row8: int = 0, flat: bool = False, dense: bool = False, stats: dict = None,
                      wgrad_two_phase: int = 0):
    """Backward of ``vgg_forward_nhwc(..., defer_norm=True, save_for_backward=True)``.
    d_feats[l]: NHWC fp32 gradient w.r.t. the L2-normalised map l.  Returns {parameter name: gradient} for the
    22 tensors that receive one at level 3 (conv0..conv14 weights+biases, conv_dec1/2 weights), plus the conf head weights
    when ``confs`` / ``d_confs`` (the forward's confidence maps and their [B,h,w] gradients) are given, plus conv_dec3.* at
    level 4 (d_feats[3] is [B,H,W,64]: the gradient w.r.t. the zero-padded x24).
    ``flat=True``: the 18 weight / bias gradients of conv0..conv_dec2.3 are VIEWS of one contiguous fp32 buffer that the
    wgrad kernels write into directly, and ``(grads, flat_buffer)`` is returned: a data-parallel caller all-reduces that one
    buffer in place (parallel.GradSync) -- no gather copy before and no scatter copy after the collective.
    With ``scale_invariant`` the call skips every tile whose gradient is zero because the d_feats are (include/hla.h,
    HLA_VGG_BWD_SCALE_INVARIANT); ``dense=True`` (``args.bwd_trim = 0`` on the models) visits all of them (A/B, tests).
    ``stats`` (a dict, diagnostics: costs a device synchronisation) receives 'live_tiles' / 'total_tiles' per sample, summed over
    the data- and weight-gradient launches; both 0 when the call took the dense walk.
    ``wgrad_two_phase`` (``args.wgrad_two_phase`` on the models; A/B and tests): bit 0 HLA_VGG_BWD_WGRAD_TWO_PHASE, bit 1
    HLA_VGG_BWD_WGRAD0_UNFUSED (conv0's weight gradient from a stored map of conv2's data gradient, as before round 6)."""
    lib = _lib.load()
    x, dt = ctx['x'], ctx['dt']
    if x._version != ctx.get('x_version', x._version):
        raise RuntimeError('the input image saved for the backward of VGGUnet has been modified by an inplace operation '
                           f"(version {x._version}, expected {ctx['x_version']}): conv0's weight gradient reads it again")
    B, _, H, W = x.shape
    prm, keep, versions = _param_table(module)
    cache = module.__dict__.setdefault('_hla_packed_T', {})
    key = (dt, str(x.device), versions)
    if cache.get('key') != key:
        buf = torch.empty(lib.hla_vgg_packed_weight_T_bytes(dt), dtype=torch.uint8, device=x.device)
        _lib.check(lib.hla_vgg_pack_weights_T(C.byref(prm), _lib.ptr(buf), dt, _lib.stream_ptr()), 'hla_vgg_pack_weights_T')
        cache['key'], cache['buf'] = key, buf
    sd = dict(module.named_parameters())
    grads, gs = {}, _lib.VggGrads()
    shapes = [(name + '.weight', tuple(sd[name + '.weight'].shape)) for name in _W_ORDER[:11]] + \
             [(name + '.bias', tuple(sd[name + '.bias'].shape)) for name in _W_ORDER[:7]]
    sizes = [int(torch.Size(shp).numel()) for _, shp in shapes]
    # one allocation either way; every view starts on a 16-B boundary (all sizes are multiples of 4 elements)
    flat_buf = torch.empty(sum(sizes), device=x.device, dtype=torch.float32)
    o = 0
    for (key, shp), n in zip(shapes, sizes):
        grads[key] = flat_buf[o:o + n].view(shp)
        o += n
    for i, name in enumerate(_W_ORDER[:11]):
        gs.dw[i] = grads[name + '.weight'].data_ptr()
        if i < 7:
            gs.db[i] = grads[name + '.bias'].data_ptr()
    L = ctx.get('L', 3)
    padded = {}                         # level 4: gradients of the zero-padded weights; their leading blocks are the real ones
    if L == 4:
        for i, name, shape in ((11, 'conv_dec3.1', (64, 128, 3, 3)), (12, 'conv_dec3.3', (64, 64, 3, 3))):
            g = torch.empty(shape, device=x.device, dtype=torch.float32)
            padded[name + '.weight'] = g
            gs.dw[i] = g.data_ptr()
    cp = dcp = None
    if d_confs is not None:
        dcs = [d.contiguous().float() for d in d_confs]
        cp = (C.c_void_p * 4)(*([c.data_ptr() for c in confs] + [0] * (4 - L)))
        dcp = (C.c_void_p * 4)(*([d.data_ptr() for d in dcs] + [0] * (4 - L)))
        for l in range(L):
            name = f'conf{l}.1.weight'
            if l == 3:
                g = torch.empty(1, 64, 3, 3, device=x.device, dtype=torch.float32)
                padded[name] = g
            else:
                g = torch.empty_like(sd[name], dtype=torch.float32, memory_format=torch.contiguous_format)
                grads[name] = g
            gs.dw[13 + l] = g.data_ptr()
    dfs = [d.contiguous().float() for d in d_feats]
    fp = (C.c_void_p * 4)(*([f.data_ptr() for f in ctx['feats']] + [0] * (4 - L)))
    dp = (C.c_void_p * 4)(*([d.data_ptr() for d in dfs] + [0] * (4 - L)))
    nbytes = lib.hla_vgg_bwd_workspace_bytes(B, H, W, L, dt)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    rc = lib.hla_vgg_backward(_lib.ptr(x), ctx.get('x_plane', 0), C.byref(prm), _lib.ptr(cache['buf']), _lib.ptr(ctx['ws']), fp, _lib.ptr(ctx['inv_norm']),
                              dp, cp, dcp, C.byref(gs), _lib.ptr(ws), nbytes, B, H, W, L, dt,
                              (_lib.HLA_VGG_BWD_SCALE_INVARIANT if scale_invariant else 0)
                              | (_lib.HLA_VGG_BWD_DENSE if dense else 0)
                              | (_lib.HLA_VGG_BWD_WGRAD_TWO_PHASE if int(wgrad_two_phase) & 1 else 0)
                              | (_lib.HLA_VGG_BWD_WGRAD0_UNFUSED if int(wgrad_two_phase) & 2 else 0),
                              int(first_row8), _lib.stream_ptr())
    _lib.check(rc, 'hla_vgg_backward')
    if stats is not None:
        torch.cuda.current_stream().synchronize()
        live, total = C.c_longlong(0), C.c_longlong(0)
        _lib.check(lib.hla_vgg_backward_live_tiles(_lib.ptr(ws), B, H, W, L, dt, C.byref(live), C.byref(total)),
                   'hla_vgg_backward_live_tiles')
        stats['live_tiles'], stats['total_tiles'] = live.value, total.value
    for name, g in padded.items():
        co, ci = sd[name].shape[:2]
        grads[name] = g[:co, :ci].contiguous()
    return (grads, flat_buf) if flat else grads


class VGGUnet(nn.Module):
    def __init__(self, level, estimate_depth=0, precision: str = 'fp32'):
        super().__init__()
        if estimate_depth:
            raise NotImplementedError('estimate_depth=1 (Ford height heads, VGG.py:85-118) is out of scope')
        if level not in _LEVEL_SEL:
            raise NotImplementedError(f'VGGUnet level {level}: the reference defines -1, -2, -3, 2, 3, 4')
        self.level = level
        _dtype_code(precision)          # validate now, not at the first forward
        self.precision = precision

        def c(ci, co, bias):
            return nn.Conv2d(ci, co, kernel_size=(3, 3), stride=(1, 1), padding=1, bias=bias)

        self.conv0, self.conv2 = c(3, 64, True), c(64, 64, True)
        self.conv5, self.conv7 = c(64, 128, True), c(128, 128, True)
        self.conv10, self.conv12, self.conv14 = c(128, 256, True), c(256, 256, True), c(256, 256, True)
        self.conv_dec1 = nn.Sequential(nn.ReLU(inplace=True), c(384, 128, False), nn.ReLU(inplace=True), c(128, 128, False))
        self.conv_dec2 = nn.Sequential(nn.ReLU(inplace=True), c(192, 64, False), nn.ReLU(inplace=True), c(64, 64, False))
        self.conv_dec3 = nn.Sequential(nn.ReLU(inplace=True), c(128, 32, False), nn.ReLU(inplace=True), c(32, 16, False))
        self.conf0 = nn.Sequential(nn.ReLU(), c(256, 1, False), nn.Sigmoid())
        self.conf1 = nn.Sequential(nn.ReLU(), c(128, 1, False), nn.Sigmoid())
        self.conf2 = nn.Sequential(nn.ReLU(), c(64, 1, False), nn.Sigmoid())
        self.conf3 = nn.Sequential(nn.ReLU(), c(16, 1, False), nn.Sigmoid())

    def forward(self, x):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # an ordinary differentiable module, like the reference's (VGG.py:121-203): the backward is hla_vgg_backward
            names = [n for n, _ in self.named_parameters()]
            out = _VggFn.apply(self, names, x, *[p for _, p in self.named_parameters()])
            L = len(out) // 2
            feats, confs = list(out[:L]), list(out[L:])
        else:
            feats, confs, _ = vgg_forward_nhwc(self, x, want_conf=True)
        sel = _LEVEL_SEL[self.level]
        # NCHW-shaped views over the NHWC storage
        view = lambda i: (feats[i][..., :_CH[i]] if i == 3 else feats[i]).permute(0, 3, 1, 2)   # x24: 16 real of 64 stored channels
        return [view(i) for i in sel], [confs[i].unsqueeze(1) for i in sel]


class _VggFn(torch.autograd.Function):
    """Stand-alone VGGUnet under autograd: forward = the training-mode extractor (activations and pool argmax kept), outputs
    the L2-normalised maps and the confidence maps; backward = hla_vgg_backward (MFMA dgrad / wgrad kernels) -> parameter
    gradients.  The input image gets no gradient (the reference never asks for one: its images come from the DataLoader)."""

    @staticmethod
    def forward(ctx, module, names, x, *params):
        feats, confs, inv, c = vgg_forward_nhwc(module, x, want_conf=True, defer_norm=True, save_for_backward=True)
        B = x.shape[0]
        # the un-deferred path scales in-kernel with one fp64 multiply per element (scale_kernel): the same arithmetic here
        normed = [(f.double() * inv[l].view(B, 1, 1, 1)).float() for l, f in enumerate(feats)]
        ctx.module, ctx.names, ctx.saved, ctx.n_levels = module, names, c, len(confs)
        confs = tuple(confs)
        # the confidence maps are OUTPUTS: kept through save_for_backward (which knows how to hold an output without the
        # output -> grad_fn -> ctx -> output cycle a plain attribute would make), not as ctx.confs
        ctx.save_for_backward(*confs)
        return tuple(normed) + confs

    @staticmethod
    def backward(ctx, *grads):
        c = ctx.saved
        if c is None:
            raise RuntimeError('VGGUnet: backward through the same forward twice (the saved activations were released after the '
                               'first backward; retain_graph is not supported by the HIP backward)')
        confs = list(ctx.saved_tensors)
        L = ctx.n_levels
        d_feats = [g if g is not None else torch.zeros_like(c['feats'][l]) for l, g in enumerate(grads[:L])]
        d_confs = [g if g is not None else torch.zeros_like(confs[l]) for l, g in enumerate(grads[L:])]
        g = vgg_backward_nhwc(ctx.module, c, d_feats, confs, d_confs, scale_invariant=False)
        ctx.saved = None
        return (None, None, None) + tuple(g.get(n) for n in ctx.names)


def L2_norm(x):
    """VGG.py:511-514 (host-side helper kept for API parity; the fused path normalises in-kernel)."""
    B = x.shape[0]
    n = x.reshape(B, -1).norm(dim=1).clamp_min(1e-12)
    return x / n.view(B, 1, 1, 1)
