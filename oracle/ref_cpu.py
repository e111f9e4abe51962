"""CPU oracle for the HighlyAccurate per-pair localisation hot path.

TEST INFRASTRUCTURE ONLY.  This file is a from-scratch restatement, in plain
PyTorch tensor ops, of the algorithm the reference implements in
``VGG.py``, ``jacobian.py``, ``models_kitti.py`` and ``models_ford.py``
(``/root/reference``).  It exists so that the HIP path can be checked against
something that runs on any CPU (the reference's Python files never travel to
the GPU box).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package
``highlyaccurate_amd`` never does.

Pinning: ``oracle/make_golden.py`` imports the *real* reference in the build
container (torchvision shimmed) and records its outputs for seeded inputs under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this restatement
against those vectors, in fp32 and fp64.

Everything is dtype-generic: feed ``.double()`` parameters/inputs to get the
fp64 oracle that the parity gates compare against.

Reference citations (file:line, relative to /root/reference):
  * VGGUnet                 VGG.py:13-203, L2_norm VGG.py:511-514
  * grid_sample             jacobian.py:138-205
  * LM_S2GP                 models_kitti.py:598-1316 (+ level-first 1318-1492)
  * LM_S2GP_Ford            models_ford.py:21-466, 652-1036
  * loss_func (method 0)    models_ford.py:1041-1093
  * constants               utils.py:5-32
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# constants (utils.py:5-32)
# ----------------------------------------------------------------------------
CAMERA_HEIGHT = 1.65
SATMAP_PROCESS_SIDELENGTH = 512
EPS = 1e-7


def meter_per_pixel() -> float:
    """utils.py:28-32 with the default arguments: lat 49.015, zoom 18, scale 1."""
    m = 156543.03392 * np.cos(49.015 * np.pi / 180.0) / (2 ** 18)
    m /= 2
    m /= 1.0
    return m


def default_args(**kw) -> SimpleNamespace:
    """The argparse defaults of train_kitti.py:428-481 that the model reads."""
    d = dict(level=3, N_iters=5, using_weight=0, loss_method=0, proj='geo', Optimizer='LM',
             rotation_range=10.0, shift_range_lat=20.0, shift_range_lon=20.0,
             damping=0.1, train_damping=0, dropout=0, use_hessian=0, use_gt_depth=0, visualize=0,
             coe_shift_lat=100.0, coe_shift_lon=100.0, coe_heading=100.0,
             coe_L1=100.0, coe_L2=100.0, coe_L3=100.0, coe_L4=100.0,
             beta1=0.9, beta2=0.999, estimate_depth=0, level_first=0)
    d.update(kw)
    return SimpleNamespace(**d)


# ----------------------------------------------------------------------------
# synthetic, portable parameter / input generators (numpy legacy RandomState is
# bit-stable across numpy versions, so fixtures only need to hold outputs)
# ----------------------------------------------------------------------------
VGG_LAYOUT = [  # (state-dict name, Cout, Cin, has_bias)   -- VGG.py:23-81
    ('conv0', 64, 3, True), ('conv2', 64, 64, True), ('conv5', 128, 64, True), ('conv7', 128, 128, True),
    ('conv10', 256, 128, True), ('conv12', 256, 256, True), ('conv14', 256, 256, True),
    ('conv_dec1.1', 128, 384, False), ('conv_dec1.3', 128, 128, False),
    ('conv_dec2.1', 64, 192, False), ('conv_dec2.3', 64, 64, False),
    ('conv_dec3.1', 32, 128, False), ('conv_dec3.3', 16, 32, False),
    ('conf0.1', 1, 256, False), ('conf1.1', 1, 128, False), ('conf2.1', 1, 64, False), ('conf3.1', 1, 16, False),
]


def synth_vgg_state(rs: np.random.RandomState, bias_scale: float = 0.0) -> dict:
    """Kaiming-normal(fan_out, relu) weights as torchvision's non-pretrained VGG init
    (SURVEY 8(d) 'Synthetic inputs').  ``bias_scale`` > 0 gives non-zero biases so the
    bias path is exercised."""
    sd = {}
    for name, co, ci, has_b in VGG_LAYOUT:
        std = math.sqrt(2.0 / (co * 9))
        sd[name + '.weight'] = torch.from_numpy((rs.standard_normal((co, ci, 3, 3)) * std).astype(np.float32))
        if has_b:
            sd[name + '.bias'] = torch.from_numpy((rs.standard_normal((co,)) * bias_scale).astype(np.float32))
    return sd


def synth_model_state(seed: int, bias_scale: float = 0.0, rotation_range: float = 10.0) -> dict:
    rs = np.random.RandomState(seed)
    sd = {'damping': torch.zeros(1, 3) if rotation_range > 0 else torch.zeros(())}
    for net in ('SatFeatureNet', 'GrdFeatureNet'):
        for k, v in synth_vgg_state(rs, bias_scale).items():
            sd[net + '.' + k] = v
    return sd


def synth_images(seed: int, B: int, grd_hw=(256, 1024), sat_a=512):
    """iid U[0,1) 'white noise' images and U(-1,1) ground-truth poses."""
    rs = np.random.RandomState(seed)
    sat = torch.from_numpy(rs.random_sample((B, 3, sat_a, sat_a)).astype(np.float32))
    grd = torch.from_numpy(rs.random_sample((B, 3, grd_hw[0], grd_hw[1])).astype(np.float32))
    gt = torch.from_numpy(rs.uniform(-1, 1, size=(3, B, 1)).astype(np.float32))
    return sat, grd, gt[0], gt[1], gt[2]


# ----------------------------------------------------------------------------
# VGG16 U-Net feature extractor (VGG.py:13-203)
# ----------------------------------------------------------------------------
class VGGUnet(nn.Module):
    def __init__(self, level: int):
        super().__init__()
        self.level = level

        def c(ci, co, bias):
            return nn.Conv2d(ci, co, 3, 1, 1, bias=bias)

        self.conv0, self.conv2 = c(3, 64, True), c(64, 64, True)
        self.conv5, self.conv7 = c(64, 128, True), c(128, 128, True)
        self.conv10, self.conv12, self.conv14 = c(128, 256, True), c(256, 256, True), c(256, 256, True)
        # index 1 and 3 hold the convs so that the state-dict keys match VGG.py:31-56
        self.conv_dec1 = nn.Sequential(nn.ReLU(), c(384, 128, False), nn.ReLU(), c(128, 128, False))
        self.conv_dec2 = nn.Sequential(nn.ReLU(), c(192, 64, False), nn.ReLU(), c(64, 64, False))
        self.conv_dec3 = nn.Sequential(nn.ReLU(), c(128, 32, False), nn.ReLU(), c(32, 16, False))
        self.conf0 = nn.Sequential(nn.ReLU(), c(256, 1, False), nn.Sigmoid())
        self.conf1 = nn.Sequential(nn.ReLU(), c(128, 1, False), nn.Sigmoid())
        self.conf2 = nn.Sequential(nn.ReLU(), c(64, 1, False), nn.Sigmoid())
        self.conf3 = nn.Sequential(nn.ReLU(), c(16, 1, False), nn.Sigmoid())

    def raw_maps(self, x):
        """Pre-normalisation maps x15,x18,x21,x24 and the skip tensors (VGG.py:121-155)."""
        r = F.relu
        x2 = self.conv2(r(self.conv0(x)))
        x3 = F.max_pool2d(x2, 2)
        x7 = self.conv7(r(self.conv5(r(x3))))
        x8 = F.max_pool2d(x7, 2)
        x14 = self.conv14(r(self.conv12(r(self.conv10(r(x8))))))
        x15 = F.max_pool2d(x14, 2)
        up = lambda t, like: F.interpolate(t, like.shape[2:], mode='nearest')
        x18 = self.conv_dec1(torch.cat([up(x15, x8), x8], 1))
        x21 = self.conv_dec2(torch.cat([up(x18, x3), x3], 1))
        x24 = self.conv_dec3(torch.cat([up(x21, x2), x2], 1))
        return x15, x18, x21, x24

    def forward(self, x):
        x15, x18, x21, x24 = self.raw_maps(x)
        # confidence = sigmoid(-sigmoid(conv(relu(.))))   VGG.py:160-163
        confs = [torch.sigmoid(-m(t)) for m, t in
                 ((self.conf0, x15), (self.conf1, x18), (self.conf2, x21), (self.conf3, x24))]
        feats = [l2_norm_map(t) for t in (x15, x18, x21, x24)]
        sel = {-1: [0], -2: [1], -3: [2], 2: [1, 2], 3: [0, 1, 2], 4: [0, 1, 2, 3]}[self.level]
        return [feats[i] for i in sel], [confs[i] for i in sel]


def l2_norm_map(x):
    """VGG.py:511-514: x / max(||x||_2 over C*H*W, 1e-12), per sample."""
    B = x.shape[0]
    n = x.reshape(B, -1).norm(dim=1).clamp_min(1e-12)
    return x / n.view(B, 1, 1, 1)


# ----------------------------------------------------------------------------
# bilinear sampler with analytic Jacobian (jacobian.py:138-205)
# ----------------------------------------------------------------------------
def grid_sample(image, optical, jac=None):
    """image [N,C,IH,IW]; optical [N,H,W,2] pixel coords (x,y); jac [M,N,H,W,2] or None.
    Returns (out [N,C,H,W], jac_out [M,N,C,H,W] or None)."""
    N, C, IH, IW = image.shape
    _, H, W, _ = optical.shape
    ix = optical[..., 0].reshape(N, 1, H * W)
    iy = optical[..., 1].reshape(N, 1, H * W)
    with torch.no_grad():
        x0 = torch.floor(ix)
        y0 = torch.floor(iy)
        x0c, x1c = x0.clamp(0, IW - 1), (x0 + 1).clamp(0, IW - 1)
        y0c, y1c = y0.clamp(0, IH - 1), (y0 + 1).clamp(0, IH - 1)
    inb = ((ix >= 0) & (ix <= IW - 1) & (iy >= 0) & (iy <= IH - 1)).to(image.dtype)
    assert inb.sum() > 0  # jacobian.py:172
    # weights use the *clamped* corner coordinates (jacobian.py:174-177)
    wx0, wx1 = (x1c - ix), (ix - x0c)
    wy0, wy1 = (y1c - iy), (iy - y0c)
    flat = image.reshape(N, C, IH * IW)

    def tap(yy, xx):
        idx = (yy * IW + xx).long().expand(N, C, H * W)
        return torch.gather(flat, 2, idx)

    v_nw, v_ne, v_sw, v_se = tap(y0c, x0c), tap(y0c, x1c), tap(y1c, x0c), tap(y1c, x1c)
    out = (v_nw * (wx0 * wy0 * inb) + v_ne * (wx1 * wy0 * inb) +
           v_sw * (wx0 * wy1 * inb) + v_se * (wx1 * wy1 * inb))
    out = out.reshape(N, C, H, W)
    if jac is None:
        return out, None
    d_dx = (v_nw * (-wy0 * inb) + v_ne * (wy0 * inb) + v_sw * (-wy1 * inb) + v_se * (wy1 * inb))
    d_dy = (v_nw * (-wx0 * inb) + v_ne * (-wx1 * inb) + v_sw * (wx0 * inb) + v_se * (wx1 * inb))
    d_dx = d_dx.reshape(1, N, C, H, W)
    d_dy = d_dy.reshape(1, N, C, H, W)
    jac_out = d_dx * jac[:, :, None, :, :, 0] + d_dy * jac[:, :, None, :, :, 1]
    return out, jac_out


# ----------------------------------------------------------------------------
# ground-plane back-projection tables
# ----------------------------------------------------------------------------
KITTI_K = [[582.9802, 0.0, 496.2420], [0.0, 482.7076, 125.0034], [0.0, 0.0, 1.0]]   # models_kitti.py:657-660
FORD_K_FL = [945.391406, 0.0, 855.502825, 0.0, 945.668274, 566.372868, 0.0, 0.0, 1.0]  # models_ford.py:116
FORD_HW_FL = (860, 1656)                                                              # models_ford.py:119-120


def ground_points(K_ori, grd_H, grd_W, ori_H, ori_W):
    """models_kitti.py:655-682 / models_ford.py:132-155.  fp32 on purpose (the reference
    builds these tables in fp32 in the ctor and they stay fp32 even in an fp64 run).
    Returns xyz_grd [1,h,w,3], mask [1,h,w]."""
    K = torch.tensor(K_ori, dtype=torch.float32).reshape(1, 3, 3).clone()
    Ks = K.clone()
    Ks[:, :1, :] = K[:, :1, :] * grd_W / ori_W
    Ks[:, 1:2, :] = K[:, 1:2, :] * grd_H / ori_H
    Kinv = torch.inverse(Ks)
    v, u = torch.meshgrid(torch.arange(0, grd_H, dtype=torch.float32),
                          torch.arange(0, grd_W, dtype=torch.float32), indexing='ij')
    uv1 = torch.stack([u, v, torch.ones_like(u)], dim=-1).unsqueeze(0)
    xyz_w = torch.sum(Kinv[:, None, None, :, :] * uv1[:, :, :, None, :], dim=-1)
    y = xyz_w[..., 1:2]
    w = CAMERA_HEIGHT / torch.where(torch.abs(y) > EPS, y, EPS * torch.ones_like(y))
    xyz_grd = xyz_w * w
    mask = (xyz_grd[..., -1] > 0).float()
    return xyz_grd, mask


def ford_K_256x1024():
    """models_ford.py:116-130: K_FL rows rescaled from 860x1656 to the 256x1024 network input."""
    K = torch.tensor(FORD_K_FL, dtype=torch.float32).reshape(3, 3)
    H_FL, W_FL = FORD_HW_FL
    out = torch.zeros_like(K)
    out[0] = K[0] / W_FL * 1024
    out[1] = K[1] / H_FL * 256
    out[2] = K[2]
    return out.tolist()


# ----------------------------------------------------------------------------
# pose -> satellite pixel coordinates + d(uv)/d(pose)
# ----------------------------------------------------------------------------
def kitti_pose_to_uv(args, xyz_grd, shift_u, shift_v, heading, A, require_jac=True):
    """models_kitti.py:700-801.  xyz_grd [1,h,w,3] (fp32 table); pose tensors [B,1].
    Returns uv [B,h,w,2] and (jac_u, jac_v, jac_theta) each [B,h,w,2]."""
    dt = shift_u.dtype
    B = heading.shape[0]
    k = args.rotation_range / 180 * np.pi
    th = heading * k
    su = shift_u * args.shift_range_lon
    sv = shift_v * args.shift_range_lat
    c, s = torch.cos(th), torch.sin(th)
    z, o = torch.zeros_like(c), torch.ones_like(c)
    R = torch.cat([c, z, -s, z, o, z, s, z, c], -1).view(B, 3, 3)
    T0 = torch.cat([sv, CAMERA_HEIGHT * o, -su], -1)                       # [B,3]
    T = -(R * T0[:, None, :]).sum(-1)
    X = xyz_grd.to(dt).expand(B, -1, -1, -1)
    xyz = (R[:, None, None] * X[:, :, :, None, :]).sum(-1) + T[:, None, None, :]
    mpp = meter_per_pixel() * SATMAP_PROCESS_SIDELENGTH / A
    uv = torch.stack([xyz[..., 2], xyz[..., 0]], -1) / mpp + A / 2
    if not require_jac:
        return uv, None
    dR = k * torch.cat([-s, z, -c, z, z, z, c, z, -s], -1).view(B, 3, 3)
    h, w = X.shape[1:3]
    e_u = torch.tensor([0.0, 0.0, -1.0], dtype=dt) * args.shift_range_lon
    e_v = torch.tensor([1.0, 0.0, 0.0], dtype=dt) * args.shift_range_lat
    d_u = -(R * e_u.view(1, 1, 3)).sum(-1)[:, None, None, :].expand(B, h, w, 3)
    d_v = -(R * e_v.view(1, 1, 3)).sum(-1)[:, None, None, :].expand(B, h, w, 3)
    d_t = (dR[:, None, None] * X[:, :, :, None, :]).sum(-1) - (dR * T0[:, None, :]).sum(-1)[:, None, None, :]
    pick = lambda d: torch.stack([d[..., 2], d[..., 0]], -1) / mpp
    return uv, (pick(d_u), pick(d_v), pick(d_t))


def ford_pose_to_uv(args, xyz_grd, R_FL, T_FL, shift_u, shift_v, theta, side_m, A, require_jac=True):
    """models_ford.py:173-264."""
    dt = shift_u.dtype
    B = shift_u.shape[0]
    Xc = xyz_grd.to(dt).expand(B, -1, -1, -1)
    Xb = (R_FL.to(dt)[:, None, None] * Xc[:, :, :, None, :]).sum(-1) + T_FL.to(dt)[:, None, None, :]
    su_m = args.shift_range_lat * shift_u
    sv_m = args.shift_range_lon * shift_v
    Tw = torch.cat([sv_m, -su_m, torch.zeros_like(sv_m)], -1)
    k = args.rotation_range / 180 * np.pi
    yaw = theta * k
    c, s = torch.cos(yaw), torch.sin(yaw)
    z, o = torch.zeros_like(c), torch.ones_like(c)
    Rw = torch.cat([c, s, z, -s, c, z, z, z, o], -1).view(B, 3, 3)
    P = Xb + Tw[:, None, None, :]
    Xw = (Rw[:, None, None] * P[:, :, :, None, :]).sum(-1)
    Rs = torch.tensor([0, 1, 0, -1, 0, 0, 0, 0, 1], dtype=dt).view(1, 3, 3).expand(B, 3, 3)
    Xs = (Rs[:, None, None] * Xw[:, :, :, None, :]).sum(-1)
    mpp = side_m / A
    uv = Xs[..., :2] / mpp + A // 2
    if not require_jac:
        return uv, None
    h, w = Xc.shape[1:3]
    dRw = k * torch.cat([-s, c, z, -c, -s, z, z, z, z], -1).view(B, 3, 3)
    e_u = args.shift_range_lat * torch.tensor([0.0, -1.0, 0.0], dtype=dt)
    e_v = args.shift_range_lon * torch.tensor([1.0, 0.0, 0.0], dtype=dt)
    dXw_t = (dRw[:, None, None] * P[:, :, :, None, :]).sum(-1)
    dXw_u = (Rw * e_u.view(1, 1, 3)).sum(-1)
    dXw_v = (Rw * e_v.view(1, 1, 3)).sum(-1)
    dXs_t = (Rs[:, None, None] * dXw_t[:, :, :, None, :]).sum(-1)
    dXs_u = (Rs * dXw_u[:, None, :]).sum(-1)[:, None, None, :].expand(B, h, w, 3)
    dXs_v = (Rs * dXw_v[:, None, :]).sum(-1)[:, None, None, :].expand(B, h, w, 3)
    return uv, (dXs_u[..., :2] / mpp, dXs_v[..., :2] / mpp, dXs_t[..., :2] / mpp)


# ----------------------------------------------------------------------------
# one damped Gauss-Newton / LM step (models_kitti.py:939-1041, models_ford.py:380-466)
# ----------------------------------------------------------------------------
def lm_update(args, damping_param, shift_u, shift_v, theta, sat_feat_proj, grd_feat, grd_conf, dfeat_dpose,
              using_weight, ford=False, rand_uv=None, gauss_newton=False):
    """All map tensors are already restricted to the rows that take part
    (bottom half for proj=='geo').  ``rand_uv`` optionally supplies the (rand_u, rand_v)
    pair instead of drawing from the global torch CPU generator."""
    dt = sat_feat_proj.dtype
    if not ford:
        if args.rotation_range == 0:
            dfeat_dpose = dfeat_dpose[:2]
        elif args.shift_range_lat == 0 and args.shift_range_lon == 0:
            dfeat_dpose = dfeat_dpose[2:]
    N, B, C, H, W = dfeat_dpose.shape
    if args.train_damping:
        lam = 10.0 ** (-6 + torch.sigmoid(damping_param) * 11.0)
    else:
        lam = args.damping * torch.ones(1, 3 if ford else N, dtype=torch.float32)
    lam = lam.to(dt)
    if args.dropout > 0 and not gauss_newton:
        inds = np.random.permutation(np.arange(H * W))[: H * W // 2]
        J = dfeat_dpose.reshape(N, B, C, -1)[:, :, :, inds].reshape(N, B, -1)
        s = sat_feat_proj.reshape(B, C, -1)[:, :, inds].reshape(B, -1)
        g = grd_feat.reshape(B, C, -1)[:, :, inds].reshape(B, -1)
        gc = grd_conf.reshape(B, -1)[:, inds]
    else:
        J = dfeat_dpose.reshape(N, B, -1)
        s = sat_feat_proj.reshape(B, -1)
        g = grd_feat.reshape(B, -1)
        gc = grd_conf.reshape(B, -1)
    if gauss_newton:
        # GN_update (models_ford.py:534-598): torch.norm without a clamp, the ground map is NOT renormalised, no damping,
        # no dropout (args.dropout is not read there)
        ns = s.norm(dim=-1)
        s = s / ns[:, None]
        J = J / ns[None, :, None]
    else:
        ns = s.norm(dim=-1).clamp_min(1e-6)
        s = s / ns[:, None]
        J = J / ns[None, :, None]
        ng = g.norm(dim=-1).clamp_min(1e-6)
        g = g / ng[:, None]
    r = s - g
    Jb = J.permute(1, 2, 0)                                   # [B,D,N]
    if using_weight:
        wgt = gc[:, None, :].expand(B, C, gc.shape[-1]).reshape(B, -1)
        JtW = Jb.transpose(1, 2) * wgt[:, None, :]
    else:
        JtW = Jb.transpose(1, 2)
    Hm = JtW @ Jb
    if args.use_hessian:
        D = torch.diag_embed(torch.diagonal(Hm, dim1=1, dim2=2))
    else:
        D = torch.eye(N, dtype=dt).expand(B, N, N)
    if gauss_newton:
        delta = -torch.inverse(Hm) @ JtW @ r.reshape(B, -1, 1)
    else:
        delta = -torch.inverse(Hm + lam * D) @ JtW @ r.reshape(B, -1, 1)
    if (not ford) and args.rotation_range == 0:
        return shift_u + delta[:, 0:1, 0], shift_v + delta[:, 1:2, 0], theta
    if (not ford) and args.shift_range_lat == 0 and args.shift_range_lon == 0:
        return shift_u, shift_v, theta + delta[:, 0:1, 0]
    su = shift_u + delta[:, 0:1, 0]
    sv = shift_v + delta[:, 1:2, 0]
    th = theta + delta[:, 2:3, 0]
    if rand_uv is None:
        # same generator consumption as two Uniform(-1,1).sample([B,1]) calls
        ru = (torch.rand(B, 1) * 2 - 1).to(dt)
        rv = (torch.rand(B, 1) * 2 - 1).to(dt)
    else:
        ru, rv = rand_uv
    su = torch.where((su > -2.5) & (su < 2.5), su, ru)
    sv = torch.where((sv > -2.5) & (sv < 2.5), sv, rv)
    return su, sv, th


# ----------------------------------------------------------------------------
# loss_func, method 0 (models_ford.py:1069-1093)
# ----------------------------------------------------------------------------
def loss_func(shift_lats, shift_lons, thetas, gt_lat, gt_lon, gt_theta, coe_lat=100, coe_lon=100, coe_theta=100):
    d_lat = (shift_lats - gt_lat[:, None, None]).abs().mean(0)
    d_lon = (shift_lons - gt_lon[:, None, None]).abs().mean(0)
    d_th = (thetas - gt_theta[:, None, None]).abs().mean(0)
    losses = coe_lat * d_lat + coe_lon * d_lon + coe_theta * d_th
    return (losses.mean(), losses[0] - losses[-1], d_lat[0] - d_lat[-1], d_lon[0] - d_lon[-1],
            d_th[0] - d_th[-1], losses[-1], d_lat[-1], d_lon[-1], d_th[-1], None, None, None, None)


# ----------------------------------------------------------------------------
# the two models
# ----------------------------------------------------------------------------
class _S2GPBase(nn.Module):
    ford = False

    def __init__(self, args, grd_hw=(256, 1024)):
        super().__init__()
        self.args = args
        self.level = args.level
        self.N_iters = args.N_iters
        self.using_weight = args.using_weight
        self.SatFeatureNet = VGGUnet(self.level)
        self.GrdFeatureNet = VGGUnet(self.level)
        if self.ford or args.rotation_range > 0:
            self.damping = nn.Parameter(torch.zeros(1, 3))
        else:
            self.damping = nn.Parameter(torch.zeros(()))
        oh, ow = grd_hw
        K = ford_K_256x1024() if self.ford else KITTI_K
        # K is expressed for a 256x1024 image (models_kitti.py:657-667).  For the reference's own input size
        # this is exactly grd_img2cam(h_l, w_l, 256, 1024); for any other size the same call rescales the
        # intrinsics to the level grid, i.e. the image is treated as a resampled view of the same camera.
        self.xyz_grds = [ground_points(K, oh / 2 ** (3 - l), ow / 2 ** (3 - l), 256, 1024) for l in range(4)]
        self.trace = None

    # level index into xyz_grds for feature-list position `pos`
    def _table(self, pos):
        # level 2 (Ford only, models_ford.py:59-65): positions 0, 1 = [x18, x21] use grd_img2cam(H / 2^(2 - pos)) = the H/4, H/2 tables
        return self.xyz_grds[pos + 1 if self.level == 2 else pos]

    def _pose_to_uv(self, pos, A, su, sv, th, extra, require_jac=True):
        xyz, mask = self._table(pos)
        if self.ford:
            R_FL, T_FL, side_m = extra
            uv, jac = ford_pose_to_uv(self.args, xyz, R_FL, T_FL, su, sv, th, side_m, A, require_jac)
        else:
            uv, jac = kitti_pose_to_uv(self.args, xyz, su, sv, th, A, require_jac)
        return uv, jac, mask.to(su.dtype)

    def project_map_to_grd(self, sat_f, sat_c, su, sv, th, pos, extra=None, require_jac=True):
        """models_kitti.py:803-937 / models_ford.py:266-378."""
        A = sat_f.shape[-1]
        uv, jac, mask = self._pose_to_uv(pos, A, su, sv, th, extra, require_jac)
        B = uv.shape[0]
        mask = mask.expand(B, -1, -1)
        jac_t = torch.stack(jac, 0) if require_jac else None
        f, new_jac = grid_sample(sat_f, uv, jac_t)
        f = f * mask[:, None]
        if require_jac:
            new_jac = new_jac * mask[None, :, None]
        c = None
        if sat_c is not None:
            c, _ = grid_sample(sat_c, uv)
            c = c * mask[:, None]
        return f, c, new_jac, uv * mask[..., None], mask

    def _step(self, pos, sat_feat, sat_conf, grd_feat, grd_conf, su, sv, th, extra):
        h = grd_feat.shape[-2]
        f, c, jac, _, mask = self.project_map_to_grd(sat_feat, sat_conf, su, sv, th, pos, extra)
        g = grd_feat * mask[:, None]
        gc = grd_conf * mask[:, None]
        if self.args.proj == 'geo':
            f, g, gc, jac = f[:, :, h // 2:], g[:, :, h // 2:], gc[:, :, h // 2:], jac[:, :, :, h // 2:]
        opt = getattr(self.args, 'Optimizer', 'LM')
        if opt == 'LM':
            return lm_update(self.args, self.damping, su, sv, th, f, g, gc, jac, self.using_weight, ford=self.ford)
        if opt == 'GN':                 # Ford only (models_ford.py:775-781)
            assert self.ford
            return lm_update(self.args, self.damping, su, sv, th, f, g, gc, jac, self.using_weight, ford=True, gauss_newton=True)
        # the reference's ablation updaters (models_kitti.py:1056-1125): gradient of sum r^2 on the raw maps, step 0.01
        B = f.shape[0]
        delta = (2 * (f - g)[None] * jac).reshape(3, B, -1).sum(-1).transpose(0, 1)          # [B,3]
        if opt == 'ADAM':
            b1, b2, t = self.args.beta1, self.args.beta2, self._adam_t
            if t == 0:
                self._adam_m = self._adam_v = 0
            self._adam_m = b1 * self._adam_m + (1 - b1) * delta
            self._adam_v = b2 * self._adam_v + (1 - b2) * delta * delta
            delta = (self._adam_m / (1 - b1 ** (t + 1))) / ((self._adam_v / (1 - b2 ** (t + 1))) ** 0.5 + 1e-8)
            self._adam_t = t + 1
        return su - 0.01 * delta[:, 0:1], sv - 0.01 * delta[:, 1:2], th - 0.01 * delta[:, 2:3]

    def solve(self, sat_feats, sat_confs, grd_feats, grd_confs, extra=None, level_first=0):
        """The 15/30-step loop (models_kitti.py:1176-1283 iter-first; 1352-1459 level-first).
        Returns (us, vs, thetas) each [B, N_iters, Level] (iter-first stacking)."""
        B = sat_feats[0].shape[0]
        dt = sat_feats[0].dtype
        self._adam_t = 0
        su = torch.zeros(B, 1, dtype=dt)
        sv = torch.zeros(B, 1, dtype=dt)
        th = torch.zeros(B, 1, dtype=dt)
        L = len(sat_feats)
        us = [[None] * L for _ in range(self.N_iters)]
        vs = [[None] * L for _ in range(self.N_iters)]
        ts = [[None] * L for _ in range(self.N_iters)]
        order = ([(i, l) for l in range(L) for i in range(self.N_iters)] if level_first
                 else [(i, l) for i in range(self.N_iters) for l in range(L)])
        for i, l in order:
            su, sv, th = self._step(l, sat_feats[l], sat_confs[l], grd_feats[l], grd_confs[l], su, sv, th, extra)
            us[i][l], vs[i][l], ts[i][l] = su[:, 0], sv[:, 0], th[:, 0]
            su, sv, th = su.clone(), sv.clone(), th.clone()
        stack = lambda x: torch.stack([torch.stack(r, 1) for r in x], 1)
        return stack(us), stack(vs), stack(ts)


class LM_S2GP(_S2GPBase):
    """KITTI model, models_kitti.py:598-1316."""
    ford = False

    def forward(self, sat_map, grd_img_left, gt_shiftu=None, gt_shiftv=None, gt_heading=None, mode='train',
                file_name=None, gt_depth=None, loop=0, level_first=0):
        sat_feats, sat_confs = self.SatFeatureNet(sat_map)
        grd_feats, grd_confs = self.GrdFeatureNet(grd_img_left)
        us, vs, ts = self.solve(sat_feats, sat_confs, grd_feats, grd_confs, None, level_first)
        shift_lats, shift_lons, thetas = vs, us, ts                      # models_kitti.py:1281-1283
        self.trace = (shift_lats, shift_lons, thetas)
        if mode == 'train':
            coe_h = 0 if self.args.rotation_range == 0 else self.args.coe_heading
            out = loss_func(shift_lats, shift_lons, thetas, gt_shiftv[:, 0], gt_shiftu[:, 0], gt_heading[:, 0],
                            self.args.coe_shift_lat, self.args.coe_shift_lon, coe_h)
            return (*out, grd_confs)
        return shift_lats[:, -1, -1], shift_lons[:, -1, -1], thetas[:, -1, -1]


# ----------------------------------------------------------------------------
# ground -> satellite direction (LM_G2SP, models_kitti.py:22-499), proj == 'geo'
# ----------------------------------------------------------------------------
def g2s_pose_to_uv(args, A, shift_u, shift_v, heading, camera_k, grd_h, grd_w, ori_h, ori_w, require_jac=True):
    """get_warp_sat2real (models_kitti.py:53-84) + seq_warp_real2camera (86-161): for every pixel of an A x A
    satellite map, the ground-image coordinates (at the grd_h x grd_w feature resolution) it projects to, and their
    derivatives w.r.t. (shift_u, shift_v, heading).  Returns uv [B,A,A,2], (du, dv, dtheta) each [B,A,A,2], mask [B,A,A,1]."""
    dt = shift_u.dtype
    B = shift_u.shape[0]
    i = torch.arange(A)
    ii, jj = torch.meshgrid(i, i, indexing='ij')
    uvc = torch.stack([jj, ii], -1).to(dt) - (A // 2)                      # (u, v) from the centre, 66-68
    mpp = meter_per_pixel() * SATMAP_PROCESS_SIDELENGTH / A                # 71-72
    X, Z = mpp * uvc[..., 1], mpp * uvc[..., 0]                            # R = [[0,1],[1,0]]: v -> X, u -> Z   (73-78)
    XYZ1 = torch.stack([X, torch.zeros_like(X), Z, torch.ones_like(X)], -1)    # [A,A,4]
    su_m = args.shift_range_lon * shift_u
    sv_m = args.shift_range_lat * shift_v
    k = args.rotation_range / 180 * np.pi
    ang = heading * k
    c, s = torch.cos(-ang), torch.sin(-ang)
    z, o = torch.zeros_like(c), torch.ones_like(c)
    R = torch.cat([c, z, -s, z, o, z, s, z, c], -1).view(B, 3, 3)          # 99
    T = torch.cat([sv_m, CAMERA_HEIGHT * o, -su_m], -1).unsqueeze(-1)      # 105-106
    K = camera_k.to(dt).clone()
    K[:, :1, :] = camera_k[:, :1, :].to(dt) * grd_w / ori_w               # 111-113
    K[:, 1:2, :] = camera_k[:, 1:2, :].to(dt) * grd_h / ori_h
    P = K @ torch.cat([R, T], -1)                                          # [B,3,4]
    uv1 = (P[:, None, None] * XYZ1[None, :, :, None, :]).sum(-1)           # [B,A,A,3]
    last = torch.maximum(uv1[..., 2:], torch.full_like(uv1[..., 2:], 1e-6))
    uv = uv1[..., :2] / last
    mask = last > 1e-6
    if not require_jac:
        return uv, None, mask
    dT_dx = args.shift_range_lon * torch.tensor([0.0, 0.0, -1.0], dtype=dt).view(1, 3, 1).expand(B, 3, 1)
    dT_dy = args.shift_range_lat * torch.tensor([1.0, 0.0, 0.0], dtype=dt).view(1, 3, 1).expand(B, 3, 1)
    dR = k * torch.cat([s, z, c, z, z, z, -c, z, s], -1).view(B, 3, 3)     # 130
    Z3, Z1 = torch.zeros(B, 3, 3, dtype=dt), torch.zeros(B, 3, 1, dtype=dt)
    jac = []
    for dP in (K @ torch.cat([Z3, dT_dx], -1), K @ torch.cat([Z3, dT_dy], -1), K @ torch.cat([dR, Z1], -1)):
        d1 = (dP[:, None, None] * XYZ1[None, :, :, None, :]).sum(-1)
        d = d1[..., :2] / last - uv1[..., :2] * d1[..., 2:] / last ** 2    # 143-145
        jac.append(torch.where(mask, d, torch.zeros_like(d)))
    return uv, tuple(jac), mask


def lm_update_g2s(args, damping_param, shift_u, shift_v, heading, grd_feat_proj, grd_conf_proj, sat_feat, dfeat_dpose,
                  using_weight):
    """LM_G2SP.LM_update (models_kitti.py:333-379): no renormalisation, no re-initialisation, always 3-DoF,
    lambda = the `damping` parameter itself when train_damping else args.damping."""
    N, B, C, H, W = dfeat_dpose.shape
    dt = sat_feat.dtype
    r = (grd_feat_proj - sat_feat).reshape(B, -1, 1)
    lam = (damping_param if args.train_damping else args.damping * torch.ones(1, 3)).to(dt)
    Jb = dfeat_dpose.flatten(2).permute(1, 2, 0)                           # [B,D,3]
    JtW = Jb.transpose(1, 2)
    if using_weight:
        JtW = JtW * grd_conf_proj.expand(B, C, H, W).reshape(B, 1, -1)
    Hm = JtW @ Jb
    delta = -torch.inverse(Hm + lam * torch.eye(3, dtype=dt)) @ JtW @ r
    return shift_u + delta[:, 0:1, 0], shift_v + delta[:, 1:2, 0], heading + delta[:, 2:, 0]


class LM_G2SP(nn.Module):
    """KITTI model, ground -> satellite projection (models_kitti.py:22-499), proj == 'geo'."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.level = args.level
        self.N_iters = args.N_iters
        self.using_weight = args.using_weight
        self.SatFeatureNet = VGGUnet(self.level)
        self.GrdFeatureNet = VGGUnet(self.level)
        self.damping = nn.Parameter(args.damping * torch.ones(1, 3))       # models_kitti.py:41
        self.trace = None

    def project_grd_to_map(self, grd_f, grd_c, su, sv, th, camera_k, A, ori_h, ori_w):
        """models_kitti.py:163-303: features are NOT multiplied by the z>0 mask (only the Jacobian is)."""
        h, w = grd_f.shape[-2:]
        uv, jac, _ = g2s_pose_to_uv(self.args, A, su, sv, th, camera_k, h, w, ori_h, ori_w)
        f, new_jac = grid_sample(grd_f, uv, torch.stack(jac, 0))
        c = grid_sample(grd_c, uv)[0] if grd_c is not None else None
        return f, c, new_jac

    def forward(self, sat_map, grd_img_left, left_camera_k, gt_shift_u=None, gt_shift_v=None, gt_heading=None,
                mode='train', file_name=None, gt_depth=None):
        B, _, ori_h, ori_w = grd_img_left.shape
        sat_feats, _ = self.SatFeatureNet(sat_map)
        grd_feats, grd_confs = self.GrdFeatureNet(grd_img_left)
        dt = sat_map.dtype
        su, sv, th = (torch.zeros(B, 1, dtype=dt) for _ in range(3))
        us, vs, ts = [], [], []
        for _ in range(self.N_iters):
            u_, v_, t_ = [], [], []
            for l in range(len(sat_feats)):
                A = sat_feats[l].shape[-1]
                f, c, jac = self.project_grd_to_map(grd_feats[l], grd_confs[l], su, sv, th, left_camera_k, A, ori_h, ori_w)
                su, sv, th = lm_update_g2s(self.args, self.damping, su, sv, th, f, c, sat_feats[l], jac, self.using_weight)
                u_.append(su[:, 0]); v_.append(sv[:, 0]); t_.append(th[:, 0])
            us.append(torch.stack(u_, 1)); vs.append(torch.stack(v_, 1)); ts.append(torch.stack(t_, 1))
        shift_lats, shift_lons, thetas = torch.stack(vs, 1), torch.stack(us, 1), torch.stack(ts, 1)   # 470-472
        self.trace = (shift_lats, shift_lons, thetas)
        if mode == 'train':
            out = loss_func(shift_lats, shift_lons, thetas, gt_shift_v[:, 0], gt_shift_u[:, 0], gt_heading[:, 0],
                            self.args.coe_shift_lat, self.args.coe_shift_lon, self.args.coe_heading)
            return (*out, grd_confs)
        return shift_lats[:, -1, -1], shift_lons[:, -1, -1], thetas[:, -1, -1]


class LM_S2GP_Ford(_S2GPBase):
    """Ford model, models_ford.py:21-1036 (estimate_depth=0 path)."""
    ford = True

    def _step(self, pos, sat_feat, sat_conf, grd_feat, grd_conf, su, sv, th, extra):
        # identical to the base step; sat_conf_proj = 1/(1+.) (models_ford.py:716) is never
        # consumed by LM_update, so it is not restated.
        return super()._step(pos, sat_feat, sat_conf, grd_feat, grd_conf, su, sv, th, extra)

    def forward(self, sat_map, grd_img_left, satmap_sidelength_meters, R_FL, T_FL,
                gt_shift_u=None, gt_shift_v=None, gt_theta=None, mode='train',
                file_name=None, level_first=0, loop=0):
        sat_feats, sat_confs = self.SatFeatureNet(sat_map)
        grd_feats, grd_confs = self.GrdFeatureNet(grd_img_left)
        extra = (R_FL, T_FL, satmap_sidelength_meters)
        us, vs, ts = self.solve(sat_feats, sat_confs, grd_feats, grd_confs, extra, level_first)
        shift_lats, shift_lons, thetas = us, vs, ts                      # models_ford.py:837-839
        self.trace = (shift_lats, shift_lons, thetas)
        if mode == 'train':
            coe_h = 0 if self.args.rotation_range == 0 else self.args.coe_heading
            out = loss_func(shift_lats, shift_lons, thetas, gt_shift_u, gt_shift_v, gt_theta,
                            self.args.coe_shift_lat, self.args.coe_shift_lon, coe_h)
            return (*out, grd_confs)
        return us[:, -1, -1], vs[:, -1, -1], ts[:, -1, -1]


def build(kind: str, args, seed: int, dtype=torch.float32, bias_scale=0.0, grd_hw=(256, 1024)):
    """Construct an oracle model with the portable synthetic weights."""
    cls = {'kitti': LM_S2GP, 'ford': LM_S2GP_Ford}[kind]
    net = cls(args, grd_hw=grd_hw)
    sd = synth_model_state(seed, bias_scale, rotation_range=(10.0 if kind == 'ford' else args.rotation_range))
    net.load_state_dict(sd)
    return net.to(dtype)
