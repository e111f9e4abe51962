"""``LM_G2SP`` -- the ground->satellite variant of the KITTI model (``models_kitti.py:22-499``, ``proj='geo'``): the
ground feature map is projected onto the satellite plane with the per-sample camera intrinsics and the LM update
runs on the satellite grid.  Same module surface as the reference (ctor argument, ``forward(sat_map, grd_img_left,
left_camera_k, gt_shift_u, gt_shift_v, gt_heading, mode, ...)``, state-dict keys).  Under autograd the forward runs
inside one ``torch.autograd.Function`` whose backward is ``hla_g2s_lm_solve_bwd`` + ``hla_vgg_backward`` x 2."""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from . import _lib, utils
from ._s2gp import loss_from_trace, loss_func, raise_like_reference  # noqa: F401
from .VGG import VGGUnet, vgg_backward_nhwc, vgg_forward_nhwc


class LM_G2SP(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.level = args.level
        self.N_iters = args.N_iters
        self.using_weight = args.using_weight
        self.loss_method = args.loss_method
        if args.level not in (3, 4):
            raise NotImplementedError('args.level must be 3 (x15, x18, x21) or 4 (+ x24)')
        if getattr(args, 'proj', 'geo') != 'geo':
            raise NotImplementedError("only proj='geo' is built (proj='nn' needs VGGUnet_G2S, VGG.py:206-350)")
        precision = getattr(args, 'precision', 'fp32')
        self.SatFeatureNet = VGGUnet(self.level, precision=precision)
        self.GrdFeatureNet = VGGUnet(self.level, precision=precision)
        self.damping = nn.Parameter(args.damping * torch.ones(size=(1, 3), dtype=torch.float32))   # models_kitti.py:41
        self.meters_per_pixel = [utils.get_meter_per_pixel() * (2 ** (3 - l)) for l in range(4)]
        self.last_trace = None
        self.last_normal_eq = None
        self.keep_normal_eq = False

    def _structs(self, sat_feats, grd_feats, grd_confs, camera_k, sat_inv_norm, grd_inv_norm):
        dev = sat_feats[0].device
        a = self.args
        B, L = sat_feats[0].shape[0], len(sat_feats)
        cfg = _lib.S2GConfig()
        cfg.ford, cfg.n_levels, cfg.n_iters, cfg.level_first = 0, L, self.N_iters, 0
        cfg.using_weight, cfg.use_hessian, cfg.dof = (1 if self.using_weight else 0), 0, 3
        cfg.shift_range_lat, cfg.shift_range_lon = float(a.shift_range_lat), float(a.shift_range_lon)
        cfg.rotation_range = float(a.rotation_range)
        lam = self.damping.detach().double().reshape(-1).tolist() if getattr(a, 'train_damping', 0) else [float(a.damping)] * 3
        for i in range(3):
            cfg.damping[i] = lam[i]
        lv = (_lib.S2GLevel * L)()
        for l in range(L):
            s, g = sat_feats[l], grd_feats[l]
            A, Cn = s.shape[1], s.shape[3]
            if not (s.shape[2] == A and g.shape[0] == B and g.shape[3] == Cn and s.is_contiguous() and g.is_contiguous()
                    and s.dtype == torch.float32 and g.dtype == torch.float32):
                raise ValueError(f'level {l}: inconsistent feature maps sat {tuple(s.shape)} / grd {tuple(g.shape)}')
            lv[l].sat_feat, lv[l].grd_feat = s.data_ptr(), g.data_ptr()
            lv[l].grd_conf = grd_confs[l].data_ptr() if (self.using_weight and grd_confs[l] is not None) else 0
            lv[l].sat_inv_norm = sat_inv_norm[l].data_ptr() if sat_inv_norm is not None else 0
            lv[l].grd_inv_norm = grd_inv_norm[l].data_ptr() if grd_inv_norm is not None else 0
            lv[l].A, lv[l].h, lv[l].w, lv[l].C, lv[l].row0, lv[l].grd_row_skip = A, g.shape[1], g.shape[2], Cn, 0, 0
            lv[l].meter_per_pixel = utils.get_meter_per_pixel() * utils.get_process_satmap_sidelength() / A   # 71-72
            lv[l].centre = float(A // 2)
        K = camera_k.to(dev).float().contiguous()
        if tuple(K.shape) != (B, 3, 3):
            raise ValueError(f'left_camera_k must be [B,3,3], got {tuple(K.shape)}')
        return cfg, lv, K

    @_lib.on_device(lambda self, sat_feats, *a, **k: sat_feats[0])
    def lm_solve(self, sat_feats, grd_feats, grd_confs, camera_k, ori_hw, init_pose=None, sat_inv_norm=None,
                 grd_inv_norm=None, keep_normal_eq=None):
        """NHWC fp32 feature lists (raw + [L,B] fp64 inverse norms, or already normalised) -> trace [B,N_iters,L,3]."""
        lib = _lib.load()
        dev = sat_feats[0].device
        B, L = sat_feats[0].shape[0], len(sat_feats)
        cfg, lv, K = self._structs(sat_feats, grd_feats, grd_confs, camera_k, sat_inv_norm, grd_inv_norm)
        trace = torch.empty(B, self.N_iters, L, 3, device=dev, dtype=torch.float32)
        strict = bool(getattr(self.args, 'strict_errors', 0))
        want_neq = strict or (self.keep_normal_eq if keep_normal_eq is None else keep_normal_eq)
        neq = torch.empty(L * self.N_iters, B, 16, device=dev, dtype=torch.float64) if want_neq else None
        nbytes = lib.hla_g2s_workspace_bytes(C.byref(cfg), lv, B)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        p0 = init_pose.to(dev).float().contiguous() if init_pose is not None else None
        rc = lib.hla_g2s_lm_solve(C.byref(cfg), lv, _lib.ptr(K), int(ori_hw[0]), int(ori_hw[1]), _lib.ptr(p0), _lib.ptr(trace),
                                  _lib.ptr(neq), _lib.ptr(ws), nbytes, B, _lib.stream_ptr())
        _lib.check(rc, 'hla_g2s_lm_solve')
        # the reference's run-time errors (see _s2gp.raise_like_reference).  This direction has no norms among its sums: a
        # sample with no satellite pixel projecting into the ground image has H = 0 exactly
        risky = cfg.use_hessian or min(cfg.damping[i] for i in range(3)) <= 0.0
        if risky or strict:
            raise_like_reference(trace, neq[:, :, 2:8].abs().sum(-1) if strict else None, 0)
        # a detached alias: under autograd `trace` becomes the Function's output (grad_fn -> ctx), and ctx/model must not hold it
        # or every step's ctx (8.5 GB of saved workspaces at B = 32) lives in a reference cycle until the cyclic GC runs
        self.last_trace, self.last_normal_eq = trace.detach(), neq
        return trace

    @_lib.on_device(lambda self, sat_feats, *a, **k: sat_feats[0])
    def lm_backward(self, sat_feats, grd_feats, grd_confs, camera_k, ori_hw, trace, normal_eq, d_trace, init_pose=None,
                    sat_inv_norm=None, grd_inv_norm=None):
        """d(loss)/d(trace) -> (d_sat[l], d_grd[l], d_conf[l] or None, d_lambda[3]); gradients w.r.t. the normalised maps."""
        lib = _lib.load()
        dev = sat_feats[0].device
        B, L = sat_feats[0].shape[0], len(sat_feats)
        cfg, lv, K = self._structs(sat_feats, grd_feats, grd_confs, camera_k, sat_inv_norm, grd_inv_norm)
        d_sat = [torch.zeros_like(f) for f in sat_feats]
        d_grd = [torch.zeros_like(f) for f in grd_feats]
        d_conf = [torch.zeros_like(grd_confs[l]) if (self.using_weight and grd_confs[l] is not None) else None for l in range(L)]
        gr = (_lib.S2GLevelGrad * L)()
        for l in range(L):
            gr[l].d_sat_feat, gr[l].d_grd_feat = d_sat[l].data_ptr(), d_grd[l].data_ptr()
            gr[l].d_grd_conf = d_conf[l].data_ptr() if d_conf[l] is not None else 0
        d_lambda = torch.zeros(3, device=dev, dtype=torch.float64)
        dtr = d_trace.contiguous().float()
        nbytes = lib.hla_g2s_bwd_workspace_bytes(C.byref(cfg), lv, B)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        p0 = init_pose.to(dev).float().contiguous() if init_pose is not None else None
        rc = lib.hla_g2s_lm_solve_bwd(C.byref(cfg), lv, gr, _lib.ptr(K), int(ori_hw[0]), int(ori_hw[1]), _lib.ptr(p0),
                                      _lib.ptr(trace), _lib.ptr(normal_eq), _lib.ptr(dtr), _lib.ptr(d_lambda), _lib.ptr(ws),
                                      nbytes, B, _lib.stream_ptr())
        _lib.check(rc, 'hla_g2s_lm_solve_bwd')
        return d_sat, d_grd, d_conf, d_lambda

    @_lib.on_device(lambda self, sat_map, *a, **k: sat_map)
    def forward(self, sat_map, grd_img_left, left_camera_k, gt_shift_u=None, gt_shift_v=None, gt_heading=None,
                mode='train', file_name=None, gt_depth=None, init_pose=None):
        """mode='test' -> (shift_lat[B], shift_lon[B], theta[B]) (models_kitti.py:498-499);
        mode='train' -> the 14-tuple (486-496); under autograd its loss back-propagates through the HIP backward (_G2sFn)."""
        if sat_map.dim() != 4 or grd_img_left.dim() != 4 or sat_map.shape[0] != grd_img_left.shape[0] \
                or sat_map.shape[2] != sat_map.shape[3]:
            raise ValueError(f'expected sat_map [B,3,A,A] and grd_img [B,3,H,W], got {tuple(sat_map.shape)} and '
                             f'{tuple(grd_img_left.shape)}')
        want_conf = bool(self.using_weight) or mode == 'train'
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            names = [n for n, _ in self.named_parameters()]
            params = [p for _, p in self.named_parameters()]
            out = _G2sFn.apply(self, names, sat_map, grd_img_left, left_camera_k, want_conf, init_pose, *params)
            trace, grd_confs = out[0], (list(out[1:]) if want_conf else [None] * self.level)
        else:
            sat_feats, _, sat_inv = vgg_forward_nhwc(self.SatFeatureNet, sat_map, want_conf=False, defer_norm=True)
            grd_feats, grd_confs, grd_inv = vgg_forward_nhwc(self.GrdFeatureNet, grd_img_left, want_conf=want_conf, defer_norm=True)
            trace = self.lm_solve(sat_feats, grd_feats, grd_confs, left_camera_k, grd_img_left.shape[-2:], init_pose, sat_inv,
                                  grd_inv)
        shift_lons, shift_lats, thetas = trace[..., 0], trace[..., 1], trace[..., 2]        # models_kitti.py:470-472
        if mode == 'train':
            a = self.args
            out = loss_from_trace(self.loss_method, trace, (1, 0, 2), gt_shift_v[:, 0], gt_shift_u[:, 0], gt_heading[:, 0],
                                  a.coe_shift_lat, a.coe_shift_lon, a.coe_heading)
            return (*out, [c.unsqueeze(1) for c in grd_confs])
        return shift_lats[:, -1, -1], shift_lons[:, -1, -1], thetas[:, -1, -1]


class _G2sFn(torch.autograd.Function):
    """forward: trace [B,N,L,3] (+ the three ground confidence maps); backward: parameter gradients from HIP kernels."""

    @staticmethod
    def forward(ctx, model, names, sat_map, grd_img, camera_k, want_conf, init_pose, *params):
        sat_feats, _, sat_inv, cs = vgg_forward_nhwc(model.SatFeatureNet, sat_map, want_conf=False, defer_norm=True,
                                                     save_for_backward=True)
        grd_feats, grd_confs, grd_inv, cg = vgg_forward_nhwc(model.GrdFeatureNet, grd_img, want_conf=want_conf,
                                                             defer_norm=True, save_for_backward=True)
        trace = model.lm_solve(sat_feats, grd_feats, grd_confs, camera_k, grd_img.shape[-2:], init_pose, sat_inv, grd_inv,
                               keep_normal_eq=True)
        ctx.model, ctx.names, ctx.init_pose = model, names, init_pose
        ctx.state = (sat_feats, grd_feats, grd_confs, camera_k, tuple(grd_img.shape[-2:]), trace.detach(), model.last_normal_eq,
                     sat_inv, grd_inv, cs, cg)
        outs = (trace,) + (tuple(grd_confs) if want_conf else ())
        if want_conf:
            ctx.mark_non_differentiable(*grd_confs)
        return outs

    @staticmethod
    def backward(ctx, d_trace, *unused):
        model = ctx.model
        if ctx.state is None:
            raise RuntimeError('backward through the same forward twice: the saved activations are released after the first '
                               'backward; retain_graph is not supported by the HIP backward')
        sat_feats, grd_feats, grd_confs, camera_k, ori_hw, trace, neq, sat_inv, grd_inv, cs, cg = ctx.state
        d_sat, d_grd, d_conf, d_lam = model.lm_backward(sat_feats, grd_feats, grd_confs, camera_k, ori_hw, trace, neq, d_trace,
                                                        ctx.init_pose, sat_inv, grd_inv)
        sync = getattr(model, 'grad_sync', None)
        g_sat, flat_sat = vgg_backward_nhwc(model.SatFeatureNet, cs, d_sat, flat=True)
        h1 = sync.start({'SatFeatureNet.' + k: v for k, v in g_sat.items()}, flat_sat) if sync else None
        use_w = model.using_weight and all(c is not None for c in d_conf)
        g_grd, flat_grd = vgg_backward_nhwc(model.GrdFeatureNet, cg, d_grd, grd_confs if use_w else None, d_conf if use_w else None,
                                            flat=True)
        h2 = sync.start({'GrdFeatureNet.' + k: v for k, v in g_grd.items()}, flat_grd) if sync else None
        if sync:
            sync.finish(h1)
            sync.finish(h2)
        grads = {'SatFeatureNet.' + k: v for k, v in g_sat.items()}
        grads.update({'GrdFeatureNet.' + k: v for k, v in g_grd.items()})
        if getattr(model.args, 'train_damping', 0):            # lambda is the parameter itself (models_kitti.py:357-358)
            grads['damping'] = d_lam.view(1, 3).float()
            if sync:
                sync.finish(sync.start({'damping': grads['damping']}))
        ctx.state = None            # release the saved workspaces now, not when the loss tensor dies
        return (None,) * 7 + tuple(grads.get(n) for n in ctx.names)
