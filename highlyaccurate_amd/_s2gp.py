"""Shared host logic of the satellite->ground LM localisation models.

Mirrors the orchestration of ``models_kitti.py:1141-1316`` / ``models_ford.py:652-866`` but the whole
N_iters x levels loop is one C-ABI call (``hla_s2g_lm_solve``): no per-step host sync, no per-step
H2D copies (the reference does ~60 syncs and ~100 small uploads per forward).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _lib, utils
from .VGG import VGGUnet, vgg_forward_nhwc, vgg_backward_nhwc

KITTI_K = [[582.9802, 0.0, 496.2420], [0.0, 482.7076, 125.0034], [0.0, 0.0, 1.0]]      # models_kitti.py:657-660
FORD_K_FL = [945.391406, 0.0, 855.502825, 0.0, 945.668274, 566.372868, 0.0, 0.0, 1.0]   # models_ford.py:116
FORD_H_FL, FORD_W_FL = 860, 1656                                                        # models_ford.py:119-120


_FEAT_DTYPES = {torch.float32: _lib.HLA_F32, torch.bfloat16: _lib.HLA_BF16, torch.float16: _lib.HLA_F16}

# Phase marks of a training step (bench.py's train.step_breakdown): when set, ``PHASE_HOOK(name)`` is called on the host right
# after the launches of phase ``name`` were enqueued on the CURRENT stream (the hook records an event there).  None: no cost.
PHASE_HOOK = None


def _phase(name: str):
    h = PHASE_HOOK
    if h is not None:
        h(name)


def ground_plane_table(K_ori, grd_H, grd_W, ori_H, ori_W):
    """Back-project the pixel grid onto the ground plane y = camera height
    (models_kitti.py:655-682 / models_ford.py:132-155).  Same fp32 op sequence as the reference so the
    table is bit-identical.  Returns xyz [h,w,3] fp32 (CPU)."""
    K = torch.tensor(K_ori, dtype=torch.float32).reshape(1, 3, 3)
    Ks = K.clone()
    Ks[:, :1, :] = K[:, :1, :] * grd_W / ori_W
    Ks[:, 1:2, :] = K[:, 1:2, :] * grd_H / ori_H
    Kinv = torch.inverse(Ks)
    v, u = torch.meshgrid(torch.arange(0, grd_H, dtype=torch.float32),
                          torch.arange(0, grd_W, dtype=torch.float32), indexing='ij')
    uv1 = torch.stack([u, v, torch.ones_like(u)], dim=-1).unsqueeze(0)
    xyz_w = torch.sum(Kinv[:, None, None, :, :] * uv1[:, :, :, None, :], dim=-1)
    y = xyz_w[..., 1:2]
    w = utils.Camera_height / torch.where(torch.abs(y) > utils.EPS, y, utils.EPS * torch.ones_like(y))
    return (xyz_w * w)[0].contiguous()


def ford_K_network_input():
    """models_ford.py:116-130: K_FL rescaled from the 860x1656 sensor to the 256x1024 network input."""
    K = torch.tensor(FORD_K_FL, dtype=torch.float32).reshape(3, 3)
    out = torch.zeros_like(K)
    out[0] = K[0] / FORD_W_FL * 1024
    out[1] = K[1] / FORD_H_FL * 256
    out[2] = K[2]
    return out.tolist()


def _loss_tensor_ops(shift_lats, shift_lons, thetas, gt_shift_lat, gt_shift_lon, gt_theta, coe_shift_lat, coe_shift_lon, coe_theta):
    """models_ford.py:1073-1092 as the reference writes it (tensor ops on whatever device / dtype the inputs have)."""
    d_lat = torch.abs(shift_lats - gt_shift_lat[:, None, None]).mean(dim=0)
    d_lon = torch.abs(shift_lons - gt_shift_lon[:, None, None]).mean(dim=0)
    d_th = torch.abs(thetas - gt_theta[:, None, None]).mean(dim=0)
    losses = coe_shift_lat * d_lat + coe_shift_lon * d_lon + coe_theta * d_th
    return (losses.mean(), losses[0] - losses[-1], d_lat[0] - d_lat[-1], d_lon[0] - d_lon[-1], d_th[0] - d_th[-1],
            losses[-1], d_lat[-1], d_lon[-1], d_th[-1])


def _pose_loss_args(xs, gts, coes):
    a = _lib.PoseLossArgs()
    B, N, L = xs[0].shape
    for k in range(3):
        a.x[k], a.gt[k], a.gt_stride[k], a.coe[k] = xs[k].data_ptr(), gts[k].data_ptr(), gts[k].stride(0), float(coes[k])
        for j in range(3):
            a.x_stride[k][j] = xs[k].stride(j)
    a.B, a.N, a.L = B, N, L
    a.gt_dtype = _lib.HLA_POSE_LOSS_F64 if gts[0].dtype == torch.float64 else _lib.HLA_POSE_LOSS_F32
    return a


class _PoseLossFn(torch.autograd.Function):
    """loss_func method 0 as ONE launch forward (hla_pose_loss) and ONE backward (hla_pose_loss_bwd) instead of ~25 + ~40
    element-wise launches of 5-7 us each (0.55 ms of a training step between the LM loop and its backward).
    xs = (trace [B,N,L,3],) with ``cols`` = the trace columns holding (lat, lon, theta), or three [B,N,L] tensors (cols None)."""

    @staticmethod
    def forward(ctx, cols, coes, gt_lat, gt_lon, gt_th, *xs):
        views = [xs[0][..., c] for c in cols] if cols is not None else list(xs)
        gts = (gt_lat, gt_lon, gt_th)
        L = views[0].shape[2]
        out = torch.empty(1 + 8 * L, dtype=gt_lat.dtype, device=views[0].device)
        args = _pose_loss_args(views, gts, coes)
        _lib.check(_lib.load().hla_pose_loss(C.byref(args), _lib.ptr(out), _lib.stream_ptr()), 'hla_pose_loss')
        ctx.save_for_backward(*xs, *gts)
        ctx.cols, ctx.coes = cols, coes
        ctx.set_materialize_grads(False)
        return (out[0],) + tuple(out[1 + j * L:1 + (j + 1) * L] for j in range(8))

    @staticmethod
    @torch.autograd.function.once_differentiable       # (the gradient is piecewise constant: its own derivative is zero a.e.)
    def backward(ctx, *g):
        cols, saved = ctx.cols, ctx.saved_tensors
        nx = 1 if cols is not None else 3
        xs, gts = saved[:nx], saved[nx:]
        if cols is not None:
            d = torch.empty_like(xs[0]) if xs[0].shape[-1] == 3 and len(set(cols)) == 3 else torch.zeros_like(xs[0])
            views, dviews, grads = [xs[0][..., c] for c in cols], [d[..., c] for c in cols], (d,)
        else:
            views = list(xs)
            grads = tuple(torch.empty(x.shape, dtype=torch.float32, device=x.device) for x in xs)
            dviews = list(grads)
        args = _pose_loss_args(views, gts, ctx.coes)
        vp = C.c_void_p
        gp, keep = (vp * 9)(), []
        for j, t in enumerate(g):
            if t is not None:
                t = t.to(gts[0].dtype).contiguous()
                keep.append(t)
                gp[j] = t.data_ptr()
        dp = (vp * 3)(*[v.data_ptr() for v in dviews])
        ds = ((C.c_longlong * 3) * 3)()
        for k in range(3):
            for j in range(3):
                ds[k][j] = dviews[k].stride(j)
        _lib.check(_lib.load().hla_pose_loss_bwd(C.byref(args), gp, dp, C.byref(ds), _lib.stream_ptr()), 'hla_pose_loss_bwd')
        return (None,) * 5 + grads


def _pose_loss(cols, xs, gts, coes):
    """The nine tensors of loss_func method 0.  Device fp32 poses with fp32 / fp64 ground truth on the same device go through
    libhla (one launch each way; the batch means are summed in fp64 and rounded once, i.e. within 1 ulp of torch's fp32
    reduction, then the reference's operation order in its result type; a second backward through the gradient raises
    (once_differentiable) where the reference would return zeros); anything else (CPU tensors, other dtypes, ground truth that requires grad, tensor-valued coefficients) is evaluated
    with the reference's own tensor ops on the tensors' device -- that is the reference function, not a fallback of a kernel."""
    views = [xs[0][..., c] for c in cols] if cols is not None else list(xs)
    fused = (all(isinstance(c, (int, float)) for c in coes)
             and all(v.is_cuda and v.dtype == torch.float32 and v.dim() == 3 and v.shape == views[0].shape for v in views)
             and all(t.is_cuda and t.device == views[0].device and t.dtype == gts[0].dtype and t.dim() == 1
                     and t.shape[0] == views[0].shape[0] and not t.requires_grad for t in gts)
             and gts[0].dtype in (torch.float32, torch.float64) and 0 < views[0].shape[1] * views[0].shape[2] <= 512
             and views[0].shape[0] > 0)
    if not fused:
        return _loss_tensor_ops(*views, *gts, *coes)
    with torch.cuda.device(views[0].device):
        return _PoseLossFn.apply(cols, tuple(float(c) for c in coes), *gts, *xs)


def loss_func(loss_method, ref_feat_list, pred_feat_dict, gt_feat_dict, shift_lats, shift_lons, thetas,
              gt_shift_lat, gt_shift_lon, gt_theta, pred_uv_dict, gt_uv_dict,
              coe_shift_lat=100, coe_shift_lon=100, coe_theta=100, coe_L1=100, coe_L2=100, coe_L3=100, coe_L4=100):
    """models_ford.py:1041-1093, method 0 (the only valid one per models_ford.py:1040).  Same positional
    signature as the reference; the feature/uv dictionaries are unused by method 0 and may be None."""
    if loss_method != 0:
        raise NotImplementedError('only loss_method=0 is supported (the reference marks 1-3 as failed trials)')
    out = _pose_loss(None, (shift_lats, shift_lons, thetas), (gt_shift_lat, gt_shift_lon, gt_theta),
                     (coe_shift_lat, coe_shift_lon, coe_theta))
    return (*out, None, None, None, None)


def loss_from_trace(loss_method, trace, cols, gt_shift_lat, gt_shift_lon, gt_theta, coe_shift_lat, coe_shift_lon, coe_theta):
    """``loss_func`` for the models' own call (models_kitti.py:1304-1310, models_ford.py:848-854): the three pose tensors are
    columns ``cols`` = (lat, lon, theta) of the LM loop's trace [B,N,L,3], so the gradient comes back as ONE d_trace tensor
    instead of three column gradients that autograd would scatter into zero-filled copies and add up."""
    if loss_method != 0:
        raise NotImplementedError('only loss_method=0 is supported (the reference marks 1-3 as failed trials)')
    out = _pose_loss(tuple(cols), (trace,), (gt_shift_lat, gt_shift_lon, gt_theta), (coe_shift_lat, coe_shift_lon, coe_theta))
    return (*out, None, None, None, None)


class S2GPBase(nn.Module):
    ford = False

    # Fields of ``args`` beyond the reference's argparse namespace (all optional; a reference Namespace works unchanged):
    #   precision          'fp32' (default, exact-fp32 MFMA) | 'fp16x3' | 'bf16' | 'fp16'                       DESIGN.md 3.8
    #   lm_feat16          1: bf16 / fp16 inference hands the LM loop fp16 feature maps; 0: fp32 maps           DESIGN.md 3.3
    #   ground_crop        1: mode='test' skips the ground-image rows (and, layer by layer, the feature rows) that cannot
    #                      reach the rows the LM loop reads; 0: whole image.  Computed rows are bit-identical  DESIGN.md 3.5
    #   train_ground_crop  0; 1: the same for training (changes the returned confidence maps above the crop)   DESIGN.md 3.5
    #   bwd_trim           1: the backward skips rows / tiles whose gradient is exactly zero; 0: dense walk     DESIGN.md 6
    #   bwd_two_streams    1: the two extractors' backward passes run on two streams (single-GPU training)       DESIGN.md 6
    #   bwd_prefill        0: the LM backward's gradient buffers are cleared by one launch in front of it; 1: allocated and cleared
    #                      during the forward, on a side stream; n > 1: as a background fill of n workgroups per buffer
    #                      (an experiment kept as a switch: measured neutral, EXPERIMENTS.md round 6)
    #   wgrad_two_phase    0; 1: weight gradients on the two-phase kernels (A/B and tests: HLA_VGG_BWD_WGRAD_TWO_PHASE); 2: only conv0's
    #                      from a stored map of conv2's data gradient instead of inside that kernel's epilogue (..._WGRAD0_UNFUSED)
    #   strict_errors      0; 1: reproduce jacobian.py:172's AssertionError (costs a host sync per forward)     DESIGN.md 1
    #   small_batch_two_streams  4: inference batches up to this size run the two extractors on two streams        DESIGN.md 5
    #   fwd_two_streams    unset / None / 0: off.  -1 or 1: inference at ANY batch with the satellite extractor on a side stream
    #                      (-1: a high-priority one, 1: equal priority) -- an experiment kept as a switch: 0.3-0.6 % SLOWER at B = 32
    #                      (EXPERIMENTS.md round 5)
    #   deterministic_backward  0; 1: lm_bwd_accum accumulates d(loss)/d(sat map) in a fixed order instead of with fp32 atomics:
    #                      the same batch twice gives bitwise equal parameter gradients
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.level = args.level
        self.N_iters = args.N_iters
        self.using_weight = args.using_weight
        self.loss_method = args.loss_method
        # level 2 = [x18, x21] with the H/4 and H/2 ground-plane tables: coherent in the Ford class only (models_ford.py:59-65;
        # the KITTI class pairs x18 with its H/8 table at level 2 and fails on the shapes)
        if args.level not in ((2, 3, 4) if self.ford else (3, 4)):
            raise NotImplementedError('args.level must be 3 (x15, x18, x21) or 4 (+ x24)' +
                                      (', or 2 (x18, x21)' if self.ford else "; the reference's KITTI model is shape-inconsistent at level 2"))
        if getattr(args, 'proj', 'geo') != 'geo':
            raise NotImplementedError("only proj='geo' is in scope")
        opt = getattr(args, 'Optimizer', 'LM')
        # KITTI (models_kitti.py:1176-1283): LM, SGD, ADAM, NN.  Ford (models_ford.py:751-788): LM, GN, NN and an SGD_update
        # (609-634, a sign-of-residual step of 0.001) that indexes its [B,3] update with three subscripts and raises as shipped.
        allowed = ('LM', 'GN') if self.ford else ('LM', 'SGD', 'ADAM')
        if opt not in allowed:
            raise NotImplementedError(f"{type(self).__name__}: Optimizer must be one of {allowed} ('NN' needs the NNrefine "
                                      f"network: out of scope; the reference's Ford SGD_update raises as shipped)")
        if getattr(args, 'estimate_depth', 0):
            raise NotImplementedError('estimate_depth (Ford height heads, VGG.py:85-118) is out of scope')
        # args.use_gt_depth only takes effect when a gt_depth tensor is passed to forward (models_kitti.py:741); neither
        # driver ever passes one (train_kitti.py:357,49), so the flag is accepted and forward() rejects an actual depth map
        precision = getattr(args, 'precision', 'fp32')
        self.SatFeatureNet = VGGUnet(self.level, precision=precision)
        self.GrdFeatureNet = VGGUnet(self.level, precision=precision)
        if self.ford or args.rotation_range > 0:
            self.damping = nn.Parameter(torch.zeros(size=(1, 3), dtype=torch.float32))
        else:
            self.damping = nn.Parameter(torch.zeros(size=(), dtype=torch.float32))
        self.meters_per_pixel = [utils.get_meter_per_pixel() * (2 ** (3 - l)) for l in range(4)]
        self._tables = {}
        self.last_trace = None       # [B, N_iters, Level, 3] (shift_u, shift_v, theta) of the last forward
        self.last_normal_eq = None
        self.last_keep = None
        self.keep_normal_eq = False

    # -- geometry tables ---------------------------------------------------------------------
    def xyz_tables(self, grd_H: int, grd_W: int, device):
        """Per-level ground-plane tables.  K is given for a 256x1024 image (models_kitti.py:657-667); the
        table of level l is grd_img2cam(H/2^(3-l), W/2^(3-l), 256, 1024): identical to the reference for its
        own 256x1024 input, and the same camera resampled for any other input size (BASELINE config 5)."""
        key = (grd_H, grd_W, str(device))
        if key not in self._tables:
            K = ford_K_network_input() if self.ford else KITTI_K
            self._tables[key] = [ground_plane_table(K, grd_H / 2 ** (3 - l), grd_W / 2 ** (3 - l), 256, 1024).to(device)
                                 for l in range(4)]
        # level 2 (Ford, models_ford.py:59-65): grd_img2cam(H / 2^(2 - l)) for l = 0, 1 = the H/4 and H/2 tables
        return self._tables[key][1:3] if self.level == 2 else self._tables[key]

    def _levels(self, maps):
        """The extractor always computes x15, x18, x21 (x15 feeds the decoder); level 2 uses the last two (VGG.py:183-184,198-199)."""
        if self.level != 2 or maps is None:
            return maps
        return maps[1:]

    # -- LM options -> C structs -------------------------------------------------------------
    def _config(self, n_levels: int, level_first: int) -> _lib.S2GConfig:
        a = self.args
        cfg = _lib.S2GConfig()
        cfg.ford = 1 if self.ford else 0
        cfg.n_levels, cfg.n_iters, cfg.level_first = n_levels, self.N_iters, 1 if level_first else 0
        cfg.optimizer = {'LM': 0, 'SGD': 1, 'ADAM': 2, 'GN': 3}[getattr(a, 'Optimizer', 'LM')]
        # SGD_update / ADAM_update (models_kitti.py:1056-1116) read neither the confidence maps nor args.dropout;
        # GN_update (models_ford.py:534-598) reads the confidence maps but not args.dropout
        cfg.using_weight = 1 if (self.using_weight and cfg.optimizer in (0, 3)) else 0
        cfg.use_hessian = 1 if getattr(a, 'use_hessian', 0) else 0
        if self.ford:
            cfg.dof = 3
        elif a.rotation_range == 0:
            cfg.dof = 2
        elif a.shift_range_lat == 0 and a.shift_range_lon == 0:
            cfg.dof = 1
        else:
            cfg.dof = 3
        cfg.shift_range_lat, cfg.shift_range_lon = float(a.shift_range_lat), float(a.shift_range_lon)
        cfg.rotation_range = float(a.rotation_range)
        cfg.beta1, cfg.beta2 = float(getattr(a, 'beta1', 0.9)), float(getattr(a, 'beta2', 0.999))
        if getattr(a, 'train_damping', 0):
            lam = (10.0 ** (-6 + torch.sigmoid(self.damping.detach().double()) * 11.0)).reshape(-1).tolist()
        else:
            lam = [float(a.damping)] * 3
        lam = (lam + lam + lam)[:3]
        if not self.ford and cfg.dof == 1 and getattr(a, 'train_damping', 0):
            lam = [lam[0]] * 3
        for i in range(3):
            cfg.damping[i] = lam[i]
        return cfg

    def _draw_reinit(self, n_steps: int, B: int, device):
        """Reproduce the reference's global-RNG consumption: two Uniform(-1,1).sample([B,1]) draws per
        LM step (models_kitti.py:1028-1029), in step order, from torch's CPU generator."""
        return draw_reinit(n_steps, B, device)

    def _draw_dropout(self, lv, level_first, device):
        """args.dropout > 0 (models_kitti.py:968-974, models_ford.py:406-412): every LM step keeps a random half of the
        pixels, ``np.random.permutation(H*W)[:H*W//2]`` from numpy's GLOBAL generator, one draw per step in execution
        order.  Returns a uint8 [steps, stride] keep mask on the device (or None)."""
        if not getattr(self.args, 'dropout', 0) or getattr(self.args, 'Optimizer', 'LM') != 'LM':
            return None
        L, N = len(lv), self.N_iters
        order = [l for l in range(L) for _ in range(N)] if level_first else [l for _ in range(N) for l in range(L)]
        npix = [(lv[l].h - lv[l].row0) * lv[l].w for l in range(L)]
        keep = np.zeros((len(order), max(npix)), np.uint8)
        for k, l in enumerate(order):
            inds = np.random.permutation(np.arange(npix[l]))[: npix[l] // 2]
            keep[k, inds] = 1
        return torch.from_numpy(keep).to(device)

    def _lm_structs(self, sat_feats, grd_feats, grd_confs, grd_hw, extra, level_first, sat_inv_norm, grd_inv_norm):
        dev = sat_feats[0].device
        L = len(sat_feats)
        tables = self.xyz_tables(grd_hw[0], grd_hw[1], dev)
        cfg = self._config(L, level_first)
        lv = (_lib.S2GLevel * L)()
        for l in range(L):
            s, g = sat_feats[l], grd_feats[l]
            A, w, Cn = s.shape[1], g.shape[2], g.shape[3]
            h = tables[l].shape[0]                     # the level's full map height; g may hold only its last rows
            skip = h - g.shape[1]
            if not (s.shape[2] == A and s.shape[3] == Cn and tuple(tables[l].shape) == (h, w, 3) and 0 <= skip <= h // 2
                    and g.shape[0] == s.shape[0] and s.is_contiguous() and g.is_contiguous()
                    and s.dtype == g.dtype and s.dtype in _FEAT_DTYPES):
                raise ValueError(f'level {l}: inconsistent feature maps sat {tuple(s.shape)} / grd {tuple(g.shape)} '
                                 f'for a {tuple(grd_hw)} ground image')
            lv[l].sat_feat, lv[l].grd_feat, lv[l].feat_dtype = s.data_ptr(), g.data_ptr(), _FEAT_DTYPES[s.dtype]
            lv[l].grd_conf = grd_confs[l].data_ptr() if (self.using_weight and grd_confs[l] is not None) else 0
            lv[l].xyz = tables[l].data_ptr()
            lv[l].sat_inv_norm = sat_inv_norm[l].data_ptr() if sat_inv_norm is not None else 0
            lv[l].grd_inv_norm = grd_inv_norm[l].data_ptr() if grd_inv_norm is not None else 0
            lv[l].A, lv[l].h, lv[l].w, lv[l].C, lv[l].row0, lv[l].grd_row_skip = A, h, w, Cn, h // 2, skip
            if self.ford:
                lv[l].meter_per_pixel = float(extra['side_m']) / A          # models_ford.py:230
                lv[l].centre = float(A // 2)                                # models_ford.py:231
            else:
                lv[l].meter_per_pixel = utils.get_meter_per_pixel() * utils.get_process_satmap_sidelength() / A
                lv[l].centre = A / 2.0                                      # models_kitti.py:765-767
        R_FL = extra['R_FL'].to(dev).float().contiguous() if self.ford else None
        T_FL = extra['T_FL'].to(dev).float().contiguous() if self.ford else None
        return cfg, lv, R_FL, T_FL

    @_lib.on_device(lambda self, sat_feats, *a, **k: sat_feats[0])
    def lm_solve(self, sat_feats, grd_feats, grd_confs, grd_hw, extra=None, level_first=0, init_pose=None,
                 sat_inv_norm=None, grd_inv_norm=None, keep_normal_eq=None):
        """sat_feats/grd_feats: NHWC fp32 lists (L2-normalised, or raw together with their [L,B] fp64
        inverse norms); returns trace [B,N_iters,L,3] = (shift_u, shift_v, theta)."""
        lib = _lib.load()
        dev = sat_feats[0].device
        B, L = sat_feats[0].shape[0], len(sat_feats)
        cfg, lv, R_FL, T_FL = self._lm_structs(sat_feats, grd_feats, grd_confs, grd_hw, extra, level_first,
                                               sat_inv_norm, grd_inv_norm)
        steps = L * self.N_iters
        reinit = (self.ford or cfg.dof == 3) and cfg.optimizer in (0, 3)
        rand_uv = self._draw_reinit(steps, B, dev) if reinit else None
        self.last_keep = self._draw_dropout(lv, level_first, dev)
        if self.last_keep is not None:
            cfg.keep, cfg.keep_stride = self.last_keep.data_ptr(), self.last_keep.shape[1]
        trace = torch.empty(B, self.N_iters, L, 3, device=dev, dtype=torch.float32)
        strict = bool(getattr(self.args, 'strict_errors', 0))
        want_neq = strict or cfg.optimizer == 3 or (self.keep_normal_eq if keep_normal_eq is None else keep_normal_eq)
        cfg.count_in_view = 1 if strict else 0
        neq = torch.empty(steps, B, 16, device=dev, dtype=torch.float64) if want_neq else None
        nbytes = lib.hla_s2g_workspace_bytes(C.byref(cfg), lv, B)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        p0 = init_pose.to(dev).float().contiguous() if init_pose is not None else None
        rc = lib.hla_s2g_lm_solve(C.byref(cfg), lv, _lib.ptr(R_FL), _lib.ptr(T_FL), _lib.ptr(p0), _lib.ptr(rand_uv),
                                  _lib.ptr(trace), _lib.ptr(neq), _lib.ptr(ws), nbytes, B, _lib.stream_ptr())
        _lib.check(rc, 'hla_s2g_lm_solve')
        # Error behaviour of the reference, which costs a host sync and is therefore only reproduced (a) under the ablation
        # flags that can make H + damping*D exactly singular (use_hessian / zero damping) and (b) when strict error checking
        # is asked for (args.strict_errors).  With the default flags the forward has no host sync;
        # a step whose pixels all fall outside the satellite map then leaves the pose unchanged (J = 0, r = -g).
        risky = cfg.optimizer == 3 or (cfg.optimizer == 0 and (cfg.use_hessian or min(cfg.damping[i] for i in range(3)) <= 0.0))
        if risky or strict:
            raise_like_reference(trace, neq[:, :, 14] if strict else None, level_first,
                                 gn_norm2=neq[:, :, 0] if cfg.optimizer == 3 else None)
        # a detached alias: under autograd `trace` becomes the Function's output (grad_fn -> ctx), and ctx/model must not hold it
        # or every step's ctx (8.5 GB of saved workspaces at B = 32) lives in a reference cycle until the cyclic GC runs
        self.last_trace, self.last_normal_eq = trace.detach(), neq
        return trace

    def lm_grad_buffers(self, sat_feats, grd_feats, grd_confs, row0s, row_skips, grd_first_row8=0, overwrite=True, deterministic=False,
                        max_blocks=0):
        """The buffers ``hla_s2g_lm_solve_bwd`` accumulates into, cleared where they have to be by ONE launch (``hla_zero_fill``) on
        the current stream: (d_sat[l], d_grd[l], d_conf[l] or None, d_lambda[4]).
        * d_sat: zero-filled (the scatter adds); with ``deterministic`` not at all (the closing pass writes every element).
        * d_grd: the loop only touches rows h_l/2.. and WRITES them on each level's first visit (cfg.grd_grad_overwrite): no
          zero-fill and no read-modify-write of half a map there.  What still has to be zero is what the consumer reads above
          them: with hla_vgg_backward(first_row8 = f) two rows (it never reads d_grd[l] above row f * 2^l - 2), else the top half.
        * d_conf: zero-filled (added to).  d_lambda: written by the call."""
        L = len(sat_feats)
        regions = []
        d_sat = [torch.empty_like(t) for t in sat_feats]
        if not deterministic:
            regions += [(d, d.numel() * 4, d.numel() * 4, 1) for d in d_sat]
        d_grd = []
        for l, t in enumerate(grd_feats):
            d = torch.empty_like(t)
            lo, hi = (max(0, (grd_first_row8 << l) - 2) if grd_first_row8 else 0), row0s[l] - row_skips[l]
            if not overwrite:
                hi = t.shape[1]
            if hi > lo:
                row = t.shape[2] * t.shape[3] * 4
                regions.append((d.data_ptr() + lo * row, (hi - lo) * row, t.shape[1] * row, t.shape[0]))
            d_grd.append(d)
        d_conf = [torch.empty_like(grd_confs[l]) if (self.using_weight and grd_confs[l] is not None) else None for l in range(L)]
        regions += [(d, d.numel() * 4, d.numel() * 4, 1) for d in d_conf if d is not None]
        d_lambda = torch.empty(4, device=sat_feats[0].device, dtype=torch.float64)
        keep = d_sat + d_grd                      # (the regions of d_grd are raw pointers into tensors that are alive here)
        for k0 in range(0, len(regions), 16):
            _lib.zero_fill(regions[k0:k0 + 16], max_blocks)
        del keep
        return d_sat, d_grd, d_conf, d_lambda

    @_lib.on_device(lambda self, sat_feats, *a, **k: sat_feats[0])
    def lm_backward(self, sat_feats, grd_feats, grd_confs, grd_hw, trace, normal_eq, d_trace, extra=None, level_first=0,
                    init_pose=None, sat_inv_norm=None, grd_inv_norm=None, keep=None, grd_first_row8=0, overwrite=True, bufs=None):
        """Backward of ``lm_solve``: d(loss)/d(trace) [B,N,L,3] -> (d_sat[l], d_grd[l], d_conf[l] or None, d_lambda[4]).
        ``keep``: the forward's dropout mask (``self.last_keep``), if args.dropout.
        Map gradients are NHWC fp32 and taken w.r.t. the L2-normalised maps (inv_norm * stored map).
        ``bufs``: what ``lm_grad_buffers`` returned for the same maps and flags (training allocates and clears them while the
        forward's convolutions run); None: made here.  ``args.deterministic_backward``: hla_s2g_config.deterministic."""
        lib = _lib.load()
        dev = sat_feats[0].device
        B, L = sat_feats[0].shape[0], len(sat_feats)
        cfg, lv, R_FL, T_FL = self._lm_structs(sat_feats, grd_feats, grd_confs, grd_hw, extra, level_first,
                                               sat_inv_norm, grd_inv_norm)
        if keep is not None:
            cfg.keep, cfg.keep_stride = keep.data_ptr(), keep.shape[1]
        det = bool(getattr(self.args, 'deterministic_backward', 0))
        cfg.deterministic = 1 if det else 0                  # (sizes the workspace: 8 B per satellite-map element)
        cfg.grd_grad_overwrite = 1 if overwrite else 0       # (0: zero-filled buffers, every step adds -- kept for callers of the C ABI)
        if bufs is None:
            bufs = self.lm_grad_buffers(sat_feats, grd_feats, grd_confs, [lv[l].row0 for l in range(L)],
                                        [lv[l].grd_row_skip for l in range(L)], grd_first_row8, overwrite, det)
        d_sat, d_grd, d_conf, d_lambda = bufs
        gr = (_lib.S2GLevelGrad * L)()
        for l in range(L):
            gr[l].d_sat_feat, gr[l].d_grd_feat = d_sat[l].data_ptr(), d_grd[l].data_ptr()
            gr[l].d_grd_conf = d_conf[l].data_ptr() if d_conf[l] is not None else 0
        dtr = d_trace.to(dev).float().contiguous()
        nbytes = lib.hla_s2g_bwd_workspace_bytes(C.byref(cfg), lv, B)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        p0 = init_pose.to(dev).float().contiguous() if init_pose is not None else None
        # deterministic mode: 40 bits below the first visit's bound, 2^10 of room for the later visits' bounds; a batch that needs more
        # (the check costs a host sync per step: the price of this opt-in mode) is run again with 30 bits / 2^20 -- for given inputs
        # the same attempts fail and succeed, so the result stays a function of the inputs alone.  (The second attempt accumulates
        # into fresh fixed-point sums -- the call clears them -- and, with grd_grad_overwrite, rewrites d_grd; d_conf is re-cleared.)
        for attempt, bits in enumerate((1, 30) if det else (0,)):
            cfg.deterministic = bits
            if attempt:
                for t in d_conf:
                    if t is not None:
                        t.zero_()
                if not overwrite:
                    for t in d_grd:
                        t.zero_()
            rc = lib.hla_s2g_lm_solve_bwd(C.byref(cfg), lv, gr, _lib.ptr(R_FL), _lib.ptr(T_FL), _lib.ptr(p0), _lib.ptr(trace),
                                          _lib.ptr(normal_eq), _lib.ptr(dtr), _lib.ptr(d_lambda), _lib.ptr(ws), nbytes, B,
                                          _lib.stream_ptr())
            _lib.check(rc, 'hla_s2g_lm_solve_bwd')
            if not det or float(d_lambda[3]) == 0.0:
                break
        else:
            raise RuntimeError(f'deterministic_backward: the fixed-point range of d(loss)/d(sat map) was exceeded in {int(d_lambda[3])} '
                               '(step, sample) pairs even with 2^20 of head room -- the adjoints of the LM chain grow by more than that '
                               'from its last step to an earlier one; the result is not valid (run without args.deterministic_backward)')
        return d_sat, d_grd, d_conf, d_lambda[:3]

    def _features(self, sat_map, grd_img, want_conf, return_confs):
        """Both extractors of the inference path: (sat_feats, sat_inv, grd_feats, grd_confs, grd_inv), normalisation deferred."""
        # Reduced-precision inference modes: the LM loop reads fp16 feature maps (written saturating by the three feature
        # layers' epilogues; also in bf16 mode: bf16's 8 significand bits moved the worst golden seed's pose 23x, fp16's 11
        # move it 1.6x).  With the gather loop written on channel PAIRS (lm_solve.hip) the accumulate kernels are VALU-bound on
        # 16-bit maps and 30 % faster than on fp32 ones: +7 % pairs/s.  args.lm_feat16 = 0 keeps fp32 maps; the fp32-class modes
        # and every training path always do.
        f16 = (self.SatFeatureNet.precision in ('bf16', 'fp16') and self.level == 3 and bool(getattr(self.args, 'lm_feat16', 1)))
        # A small batch cannot fill the chip (a B = 1 conv launch is 8-256 workgroups on 256 CUs) and the forward is then a chain of
        # ~70 dependent launches: the two extractors are independent, so the satellite branch runs on a side stream next to the
        # ground branch's (B <= args.small_batch_two_streams, default 4: B = 1 0.700 -> 0.615 ms, B = 4 1.195 -> 1.122; at B = 32, where both are dense, the same split measured
        # 2 % slower: DESIGN 3.1).
        small = sat_map.shape[0] <= int(getattr(self.args, 'small_batch_two_streams', 4))
        # args.fwd_two_streams (round 5 experiment, default off = None / 0): the same split at ANY batch, the satellite branch on a side
        # stream: -1 = a high-priority one (its chain owns the chip and the ground branch's launches fill the gaps its launch
        # boundaries leave), 1 = equal priority
        prio = getattr(self.args, 'fwd_two_streams', None)
        prio = None if not prio else (-1 if int(prio) < 0 else 0)
        small = small or prio is not None
        if small:
            cur = torch.cuda.current_stream()
            side = _side_stream(sat_map.device, 0 if prio is None else int(prio))
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                sat_feats, _, sat_inv = vgg_forward_nhwc(self.SatFeatureNet, sat_map, want_conf=False, defer_norm=True, feat16=f16)
            sat_map.record_stream(side)
        else:
            sat_feats, _, sat_inv = vgg_forward_nhwc(self.SatFeatureNet, sat_map, want_conf=False, defer_norm=True, feat16=f16)
        grd_in = grd_img
        # (only LM_update renormalises the ground features; SGD / ADAM see the whole-map L2_norm scale, so they need every row)
        dead_ok = (not return_confs and self.level == 3 and getattr(self.args, 'Optimizer', 'LM') == 'LM'
                   and bool(getattr(self.args, 'ground_crop', 1)))
        skip = dead_ground_rows(grd_img.shape[-2]) if dead_ok else 0
        if skip:
            grd_in = grd_img[:, :, skip:, :]         # a row window, passed as it lies in memory (hla_vgg_forward's x_plane): no copy
        # ... and inside the extractor every layer only computes the rows the LM loop's rows depend on
        f8 = ((grd_img.shape[-2] // 8) // 2 - skip // 8) if dead_ok else 0
        f8 = f8 if f8 >= 4 else 0
        grd_feats, grd_confs, grd_inv = vgg_forward_nhwc(self.GrdFeatureNet, grd_in, want_conf=want_conf, defer_norm=True,
                                                         first_row8=f8, feat16=f16)
        if small:
            cur.wait_stream(side)
            for t in list(sat_feats) + [sat_inv]:        # allocated on the side stream, consumed (LM loop) on this one
                t.record_stream(cur)
        L = self._levels
        return L(sat_feats), L(sat_inv), L(grd_feats), L(grd_confs), L(grd_inv)

    @_lib.on_device(lambda self, sat_map, *a, **k: sat_map)
    def localise(self, sat_map, grd_img, want_conf, extra, level_first, init_pose, return_confs=True):
        """Both feature pyramids (normalisation deferred into the LM sums) + the whole LM loop.
        return_confs=False (mode='test'): the caller does not need full-size confidence maps, so the ground extractor
        only runs on the image rows that can influence the bottom half of its maps (``dead_ground_rows``).
        Under autograd (training) the same kernels run inside one autograd.Function whose backward is the HIP
        backward pass (hla_s2g_lm_solve_bwd + hla_vgg_backward for both extractors)."""
        if sat_map.dim() != 4 or grd_img.dim() != 4 or sat_map.shape[0] != grd_img.shape[0] or sat_map.shape[1] != 3 \
                or grd_img.shape[1] != 3 or sat_map.shape[2] != sat_map.shape[3]:
            raise ValueError(f'expected sat_map [B,3,A,A] and grd_img [B,3,H,W] with one B, got {tuple(sat_map.shape)} '
                             f'and {tuple(grd_img.shape)}')
        _lib.same_device(('sat_map', sat_map), ('grd_img', grd_img), ('parameters', self.damping))
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            names = [n for n, _ in self.named_parameters()]
            params = [p for _, p in self.named_parameters()]
            out = _LocaliseFn.apply(self, names, sat_map, grd_img, want_conf, extra, level_first, init_pose, *params)
            return out[0], list(out[1:]) if want_conf else [None] * self.level
        # (the two dense extractor passes on two streams measured 2 % slower than back to back: DESIGN.md 3.1)
        sat_feats, sat_inv, grd_feats, grd_confs, grd_inv = self._features(sat_map, grd_img, want_conf, return_confs)
        trace = self.lm_solve(sat_feats, grd_feats, grd_confs, grd_img.shape[-2:], extra, level_first, init_pose,
                              sat_inv, grd_inv)
        return trace, grd_confs


def draw_reinit(n_steps: int, B: int, device):
    """[n_steps, 2, B] = the (rand_u, rand_v) of every LM step, VALUE FOR VALUE what the reference's 2 * n_steps calls
    ``Uniform(-1, 1).sample([B, 1])`` (models_kitti.py:1028-1029) draw from torch's global CPU generator, which is left in the
    same state: ``Uniform.sample`` is ``low + torch.rand(shape) * (high - low)`` and the CPU generator fills a tensor serially,
    so one ``torch.rand`` of all of them is the same stream (``tests/test_host_logic_cpu.py`` pins both).  One call, one pinned
    staging buffer from the caching host allocator and one asynchronous copy, instead of 30 draws + a blocking upload per forward."""
    r = torch.rand(n_steps, 2, B).mul_(2.0).add_(-1.0)         # -1 + rand * 2, the same two fp32 roundings
    if torch.device(device).type == 'cuda':
        return r.pin_memory().to(device, non_blocking=True)
    return r.to(device)


def raise_like_reference(trace, in_view, level_first, gn_norm2=None):
    """The reference's two run-time errors, in the order it would hit them.  trace [B,N,L,3]; in_view [steps,B] (or None):
    per sample, the number of pixels sampled inside the map in that step (any quantity that is zero iff there is none).
    * jacobian.py:172 `assert mask.sum() > 0` fires in step k when NO pixel of the whole batch is in view (checked before
      that step's solve);
    * torch.inverse (models_kitti.py:372,1012, models_ford.py:446,578) raises in step k when any sample's matrix is exactly
      singular -- here that step's pose comes out non-finite.
    Steps after the first failure are garbage (the reference never ran them), so only the first one counts.
    gn_norm2 [steps,B] (GN_update only): ||s||^2.  GN_update divides by the UNclamped norm (models_ford.py:551-553), so a
    sample with no pixel in the compared rows gets a NaN matrix, which torch.inverse accepts: its pose is NaN from then on
    without an error of its own (the next step's assertion fires if the rest of the batch is out of view as well)."""
    B, N, L, _ = trace.shape
    steps = N * L
    tr = (trace.permute(0, 2, 1, 3) if level_first else trace).reshape(B, steps, 3)
    bad = ~torch.isfinite(tr).all(-1)                                # [B,steps]
    if gn_norm2 is not None:
        first = torch.where(bad.any(1), bad.float().argmax(1), torch.full((B,), steps - 1, device=bad.device))
        poisoned = bad.any(1) & (gn_norm2.t().gather(1, first[:, None])[:, 0] == 0)
        bad = bad & ~poisoned[:, None]
    bad = bad.any(0)                                                 # [steps]
    k_s = int(torch.nonzero(bad)[0]) if bool(bad.any()) else steps
    k_a = steps
    if in_view is not None:
        empty = in_view.sum(1) == 0
        k_a = int(torch.nonzero(empty)[0]) if bool(empty.any()) else steps
    if k_a < steps and k_a <= k_s:
        raise AssertionError(f'grid_sample: no pixel of the batch is sampled inside the map in LM step {k_a} (jacobian.py:172)')
    if k_s < steps:
        raise RuntimeError(f'linalg.inv: the normal matrix of LM step {k_s} is singular '
                           '(use_hessian / zero damping / Gauss-Newton with no Jacobian support in one pose component)')


_SIDE_STREAMS = {}


def _side_stream(device, priority: int = 0) -> 'torch.cuda.Stream':
    """One extra stream per device (and priority), kept OUTSIDE the modules (a Stream inside a module's __dict__ would break pickling / deepcopy)."""
    idx = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    key = (idx, priority)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=idx, priority=priority)
    return st


def dead_ground_rows(H: int) -> int:
    """Rows at the top of the ground image that cannot influence anything the LM loop reads.
    The loop only reads rows h_l/2.. of each ground map (models_kitti.py:1194-1199, models_ford.py:739-743) and the
    per-sample L2_norm scale cancels in LM_update's own renormalisation (models_kitti.py:982-990).  Walking the
    receptive field back through VGGUnet.forward (x21 <- dec2.3 <- dec2.1 <- {up(x18), x3}, x18 <- dec1.3 <- dec1.1 <-
    {up(x15), x8}, x15 <- pool(conv14 <- conv12 <- conv10 <- x8), x8 <- pool(conv7 <- conv5 <- x3),
    x3 <- pool(conv2 <- conv0 <- image)) the first input row needed is H/2 - 34; rounding down to a multiple of 8 keeps
    the three 2x2 pools aligned.  The rows that ARE computed are bit-identical to the full-image run."""
    return max(0, ((H // 2 - 34) // 8) * 8)


def _bwd_first_row8(model, grd_hw, x21_rows: int) -> int:
    """hla_vgg_backward's first_row8 for the ground branch: the ground maps' gradient lives in rows h_l/2.. (all the LM loop reads),
    so the backward skips the rows above its support (level 3, LM_update, args.bwd_trim)."""
    inv = getattr(model.args, 'Optimizer', 'LM') == 'LM'
    f8 = (grd_hw[0] // 8) // 2 - (grd_hw[0] - x21_rows * 2) // 8
    return f8 if (inv and f8 >= 4 and model.level == 3 and bool(getattr(model.args, 'bwd_trim', 1))) else 0


class _LocaliseFn(torch.autograd.Function):
    """forward: trace [B,N,L,3] (+ the three ground confidence maps); backward: parameter gradients from HIP kernels."""

    @staticmethod
    def forward(ctx, model, names, sat_map, grd_img, want_conf, extra, level_first, init_pose, *params):
        sat_feats, _, sat_inv, cs = vgg_forward_nhwc(model.SatFeatureNet, sat_map, want_conf=False, defer_norm=True,
                                                     save_for_backward=True)
        _phase('fwd_sat')
        # args.train_ground_crop (an extension, default 0): train on the image rows that can reach the loss only.  The loss
        # sees the ground branch through rows h_l/2.. of its maps, so the gradient of every other row is exactly zero and the
        # rows above `dead_ground_rows` influence neither the loss nor any gradient (the L2_norm scale cancels in
        # LM_update, forward and backward).  What changes is the 14th element of the train-mode tuple: the returned confidence
        # maps are then only computed from the crop on (valid from row h_l/2, zero above the crop).
        skip = 0
        if getattr(model.args, 'train_ground_crop', 0) and model.level == 3 and getattr(model.args, 'Optimizer', 'LM') == 'LM':
            skip = dead_ground_rows(grd_img.shape[-2])
        grd_in = grd_img[:, :, skip:, :] if skip else grd_img      # (a view: the extractor takes the window's plane stride)
        grd_feats, grd_confs, grd_inv, cg = vgg_forward_nhwc(model.GrdFeatureNet, grd_in, want_conf=want_conf,
                                                             defer_norm=True, save_for_backward=True)
        _phase('fwd_grd')
        L = model._levels
        ctx.conf0 = grd_confs[0] if (model.level == 2 and grd_confs is not None) else None      # (the head's backward needs its own map)
        sat_feats, sat_inv, grd_feats, grd_confs, grd_inv = L(sat_feats), L(sat_inv), L(grd_feats), L(grd_confs), L(grd_inv)
        # args.bwd_prefill (round 6 experiment, default 0): allocate and clear the buffers the LM backward accumulates into NOW, on
        # the side stream (1.4 GB of zero-fill at B = 32), so that it runs under the extractors' convolutions queued above instead
        # of in front of the LM backward.  Measured neutral (same-box A/B, bf16 23.20 / 23.39 / 23.31 ms for off / on / a 16-workgroup
        # background fill; fp16x3 53.51 / 53.53 / 53.65): the fill takes from the 2-stage layers it runs under what it saves
        # (conv5 281 -> 439 us under a 204-us fill).  Default: ONE hla_zero_fill launch in front of the LM backward.
        ctx.bufs = None
        prefill = int(getattr(model.args, 'bwd_prefill', 0))
        if prefill:
            f8 = _bwd_first_row8(model, tuple(grd_img.shape[-2:]), grd_feats[-1].shape[1])
            tabs = model.xyz_tables(grd_img.shape[-2], grd_img.shape[-1], sat_map.device)
            row0s = [t.shape[0] // 2 for t in tabs]
            skips = [t.shape[0] - g.shape[1] for t, g in zip(tabs, grd_feats)]
            side = _side_stream(sat_map.device)
            with torch.cuda.stream(side):
                ctx.bufs = model.lm_grad_buffers(sat_feats, grd_feats, grd_confs, row0s, skips, f8, True,
                                                 bool(getattr(model.args, 'deterministic_backward', 0)),
                                                 max_blocks=prefill if prefill > 1 else 0)
        trace = model.lm_solve(sat_feats, grd_feats, grd_confs, grd_img.shape[-2:], extra, level_first, init_pose,
                               sat_inv, grd_inv, keep_normal_eq=True)
        _phase('lm_fwd')
        ctx.model, ctx.names, ctx.extra, ctx.level_first, ctx.init_pose = model, names, extra, level_first, init_pose
        ctx.state = (sat_feats, grd_feats, grd_confs, tuple(grd_img.shape[-2:]), trace.detach(), model.last_normal_eq, sat_inv, grd_inv, cs, cg,
                     model.last_keep)
        out_confs = ()
        if want_conf:
            out_confs = tuple(grd_confs)
            if skip:                                    # full-size maps for the caller, zero above the crop
                out_confs = tuple(torch.nn.functional.pad(c, (0, 0, skip >> (3 - l), 0)) for l, c in enumerate(grd_confs))
            ctx.mark_non_differentiable(*out_confs)     # loss_method 0 does not read them; their LM-weight role is in backward()
        # (no zero tensors for the outputs that got no gradient: autograd would fill three [B,h,w] maps per step just to hand
        #  them to a backward that ignores them)
        ctx.set_materialize_grads(False)
        return (trace,) + out_confs

    @staticmethod
    def backward(ctx, d_trace, *unused):
        model = ctx.model
        _phase('loss')
        if ctx.state is None:
            raise RuntimeError('backward through the same forward twice: the saved activations (GBs at B = 32) are released after '
                               'the first backward; retain_graph is not supported by the HIP backward')
        sat_feats, grd_feats, grd_confs, grd_hw, trace, neq, sat_inv, grd_inv, cs, cg, keep = ctx.state
        if d_trace is None:         # (only the confidence maps were used downstream: they are non-differentiable outputs)
            d_trace = torch.zeros_like(trace)
        # LM_update renormalises both maps, so d_feat is orthogonal to feat: HLA_VGG_BWD_SCALE_INVARIANT (include/hla.h)
        inv = getattr(model.args, 'Optimizer', 'LM') == 'LM'
        trim = bool(getattr(model.args, 'bwd_trim', 1))
        # the ground maps' gradient lives in rows h_l/2.. (all the LM loop reads): the backward skips the rows above its support
        f8 = _bwd_first_row8(model, grd_hw, grd_feats[-1].shape[1])       # (grd_feats[-1]: the H/2 map)
        bufs, ctx.bufs = ctx.bufs, None
        if bufs is not None:          # cleared on the side stream during the forward: long done, but the order must be stated
            cur = torch.cuda.current_stream()
            cur.wait_stream(_side_stream(sat_feats[0].device))
            for t in list(bufs[0]) + list(bufs[1]) + [c for c in bufs[2] if c is not None] + [bufs[3]]:
                t.record_stream(cur)
        d_sat, d_grd, d_conf, d_lam = model.lm_backward(sat_feats, grd_feats, grd_confs, grd_hw, trace, neq, d_trace, ctx.extra,
                                                   ctx.level_first, ctx.init_pose, sat_inv, grd_inv, keep, grd_first_row8=f8, bufs=bufs)
        _phase('lm_bwd')
        if model.level == 2:        # x15 takes no part in the loop: its gradient (and its confidence map's) is zero
            zf = lambda cx: torch.zeros_like(cx['feats'][0], dtype=torch.float32)
            d_sat, d_grd = [zf(cs)] + list(d_sat), [zf(cg)] + list(d_grd)
            if grd_confs is not None and all(c is not None for c in d_conf):
                grd_confs, d_conf = [ctx.conf0] + list(grd_confs), [torch.zeros_like(ctx.conf0)] + list(d_conf)
        sync = getattr(model, 'grad_sync', None)        # optional: overlap the sat-branch all-reduce with the grd backward
        # LM_update renormalises both projected maps (models_kitti.py:982-990), so the loss does not depend on the per-sample
        # scale of either extractor's output: d_feat is orthogonal to feat and the L2_norm backward needs no (x . dy) pass
        # (model.bwd_stats = {}: diagnostics, filled with the satellite branch's live / total backward tiles; costs a device sync)
        # The two extractors' backward passes are independent.  After the data-dependent trimming the satellite branch's launches
        # are sparse (0.42 of the tiles: many of them fill the chip for a round or two only), so it runs on a side stream next to
        # the ground branch's: 24.78 -> 24.43 ms per step (same-box A/B, four alternating pairs, the faster one every time).  (The
        # same split of the two FORWARD passes, both dense, measured 2 % slower.)  Not with a gradient all-reduce installed: there
        # the satellite bucket is already in flight under the ground branch's backward.  args.bwd_two_streams = 0 switches it off.
        # Memory: the satellite branch's backward workspace and gradient buffer then come from the side stream's pool of the
        # caching allocator, so the freed block cannot be reused for the ground branch's workspace on the main stream: peak
        # training memory is one backward workspace higher (GB-class at B = 32 in fp32) for that 1.4 %.
        two = bool(getattr(model.args, 'bwd_two_streams', 1)) and sync is None
        tp = int(getattr(model.args, 'wgrad_two_phase', 0))       # (A/B and tests: bit 0 HLA_VGG_BWD_WGRAD_TWO_PHASE, bit 1 ..._WGRAD0_UNFUSED)
        if two:
            cur = torch.cuda.current_stream()
            side = _side_stream(d_sat[0].device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                g_sat, flat_sat = vgg_backward_nhwc(model.SatFeatureNet, cs, d_sat, scale_invariant=inv, flat=True, dense=not trim,
                                                    stats=getattr(model, 'bwd_stats', None), wgrad_two_phase=tp)
            for t in d_sat:
                t.record_stream(side)
        else:
            g_sat, flat_sat = vgg_backward_nhwc(model.SatFeatureNet, cs, d_sat, scale_invariant=inv, flat=True, dense=not trim,
                                                stats=getattr(model, 'bwd_stats', None), wgrad_two_phase=tp)
        h1 = sync.start({'SatFeatureNet.' + k: v for k, v in g_sat.items()}, flat_sat) if sync else None
        use_w = model.using_weight and all(c is not None for c in d_conf)
        g_grd, flat_grd = vgg_backward_nhwc(model.GrdFeatureNet, cg, d_grd, grd_confs if use_w else None, d_conf if use_w else None,
                                            scale_invariant=inv, first_row8=f8, flat=True, dense=not trim, wgrad_two_phase=tp)
        h2 = sync.start({'GrdFeatureNet.' + k: v for k, v in g_grd.items()}, flat_grd) if sync else None
        if two:
            torch.cuda.current_stream().wait_stream(side)
            for t in [flat_sat] + list(g_sat.values()):       # allocated on the side stream, consumed (optimizer) on this one
                t.record_stream(torch.cuda.current_stream())
        if sync:
            sync.finish(h1)
            sync.finish(h2)
        grads = {'SatFeatureNet.' + k: v for k, v in g_sat.items()}
        grads.update({'GrdFeatureNet.' + k: v for k, v in g_grd.items()})
        if getattr(model.args, 'train_damping', 0):
            d = model.damping.detach().double()
            sg = torch.sigmoid(d)
            lam = 10.0 ** (-6 + sg * 11.0)
            dlam_dd = lam * np.log(10.0) * 11.0 * sg * (1 - sg)
            if d.dim() == 0:
                n = 2 if model.args.rotation_range == 0 else 1
                grads['damping'] = (d_lam[:n].sum() * dlam_dd).float()
            else:
                grads['damping'] = (d_lam.view(1, 3) * dlam_dd).float()
            if sync:
                sync.finish(sync.start({'damping': grads['damping']}))
        ctx.state = None            # release the saved workspaces now, not when the loss tensor dies
        _phase('vgg_bwd')
        return (None,) * 8 + tuple(grads.get(n) for n in ctx.names)


