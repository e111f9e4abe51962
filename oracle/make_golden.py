#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (/root/reference).

Build-container only: the reference's Python never travels to the GPU box, so
this script is run here, once, and its outputs (data only: seeds -> numbers) are
committed.  Inputs and weights are regenerated on both sides from
``numpy.random.RandomState(seed)`` (see oracle/ref_cpu.py ``synth_*``), so the
fixtures hold expected outputs plus the tiny explicit inputs of the
known-answer tests.

The reference imports ``torchvision`` (absent in this image) and asks for
pretrained VGG-16 weights (no network); both are satisfied by a shim that
provides a randomly initialised ``vgg16().features`` whose weights are then
overwritten by ``load_state_dict``.

Usage:  python oracle/make_golden.py [--only kat|e2e|ford|train|screen]
"""
import argparse
import os
import sys
import types
import time

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True  # /root/reference is read-only

from oracle import ref_cpu as O  # noqa: E402


def install_torchvision_shim():
    tv = types.ModuleType('torchvision')
    models = types.ModuleType('torchvision.models')
    transforms = types.ModuleType('torchvision.transforms')
    tfun = types.ModuleType('torchvision.transforms.functional')
    tutils = types.ModuleType('torchvision.utils')

    def vgg16(pretrained=False, **kw):
        cfg = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M']
        layers, cin = [], 3
        for v in cfg:
            if v == 'M':
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        return types.SimpleNamespace(features=nn.Sequential(*layers))

    models.vgg16 = vgg16
    tfun.center_crop = lambda img, size: img
    transforms.functional = tfun
    transforms.ToPILImage = lambda *a, **k: None
    tv.models, tv.transforms, tv.utils = models, transforms, tutils
    for name, mod in (('torchvision', tv), ('torchvision.models', models), ('torchvision.transforms', transforms),
                      ('torchvision.transforms.functional', tfun), ('torchvision.utils', tutils)):
        sys.modules[name] = mod


def import_reference():
    install_torchvision_shim()
    sys.path.insert(0, '/root/reference')
    import models_kitti  # noqa
    import models_ford   # noqa
    import jacobian      # noqa
    import VGG           # noqa
    return models_kitti, models_ford, jacobian, VGG


def ref_model(mod, cls, args, seed, dtype, bias_scale=0.0):
    net = getattr(mod, cls)(args)
    torch.autograd.set_detect_anomaly(False)
    rot = 10.0 if cls.endswith('Ford') else args.rotation_range
    net.load_state_dict(O.synth_model_state(seed, bias_scale, rotation_range=rot))
    return net.to(dtype)


SAMPLE_N = 64


def sample_idx(numel, salt):
    return np.random.RandomState(1000 + salt).randint(0, numel, size=SAMPLE_N)


def feat_stats(feats, B):
    """Per level: per-sample [sum, sumsq] and SAMPLE_N sampled values (NCHW flat order per sample)."""
    out = []
    for l, f in enumerate(feats):
        f = f.detach().double().reshape(B, -1)
        idx = sample_idx(f.shape[1], l)
        out.append(np.concatenate([f.sum(1, keepdim=True).numpy(), (f * f).sum(1, keepdim=True).numpy(),
                                   f[:, idx].numpy()], 1))
    return out


# ----------------------------------------------------------------------------
def gen_kat(mk, mf, jac, VGG):
    out = {}
    rs = np.random.RandomState(7)
    # (1) grid_sample: random in/out-of-bounds coords + exact-edge cases (jacobian.py:216-225)
    img = rs.standard_normal((2, 4, 8, 10)).astype(np.float32)
    uv = np.stack([rs.uniform(-1.5, 10.5, size=(2, 6, 7)), rs.uniform(-1.5, 8.5, size=(2, 6, 7))], -1).astype(np.float32)
    uv[0, 0, 0] = [9.0, 3.25]     # x exactly IW-1 -> 0
    uv[0, 0, 1] = [4.5, 7.0]      # y exactly IH-1 -> 0
    uv[0, 0, 2] = [-0.5, 2.0]     # outside -> 0
    uv[0, 0, 3] = [0.0, 0.0]      # exact corner
    uv[0, 0, 4] = [3.0, 5.0]      # integer coords
    uv[0, 0, 5] = [8.999, 6.999]
    jc = rs.standard_normal((3, 2, 6, 7, 2)).astype(np.float32)
    o, j = jac.grid_sample(torch.from_numpy(img), torch.from_numpy(uv), torch.from_numpy(jc))
    out.update(gs_img=img, gs_uv=uv, gs_jac=jc, gs_out=o.numpy(), gs_jac_out=j.numpy())
    # identity with F.grid_sample(align_corners=True) for in-bounds coords is asserted in the test itself.

    # (2) geometry: KITTI + Ford pose->uv and analytic Jacobians at level 0 (32x128)
    args = O.default_args()
    netk = mk.LM_S2GP(args)
    torch.autograd.set_detect_anomaly(False)
    pose = torch.tensor([[0.31], [-0.72]]), torch.tensor([[-0.55], [0.18]]), torch.tensor([[0.83], [-0.4]])
    for level, A in ((0, 64), (2, 256)):
        uvk, mask, ju, jv, jt = netk.grd2cam2world2sat(pose[0], pose[1], pose[2], level, A, require_jac=True)
        st = 1 if level == 0 else 8      # level 2 is stored on a stride-8 pixel lattice to keep the fixture small
        out[f'kitti_uv_l{level}'] = uvk.detach().numpy()[:, ::st, ::st]
        out[f'kitti_jac_l{level}'] = torch.stack([ju, jv, jt]).detach().numpy()[:, :, ::st, ::st]
        out[f'kitti_mask_l{level}'] = mask.numpy()[:, ::st, ::st]
        out[f'kitti_xyz_l{level}'] = netk.xyz_grds[level][0].detach().numpy()[:, ::st, ::st]
    netf = mf.LM_S2GP_Ford(args)
    torch.autograd.set_detect_anomaly(False)
    R_FL = torch.tensor([[[0., 0., 1.], [1., 0., 0.], [0., 1., 0.]]]).repeat(2, 1, 1)
    T_FL = torch.tensor([[1.7, 0.3, -1.2]]).repeat(2, 1)
    for level, A in ((0, 64), (2, 256)):
        uvf, mask, ju, jv, jt = netf.cam2body2world2sat(R_FL, T_FL, pose[0], pose[1], pose[2], level, 112.64, A,
                                                        require_jac=True)
        st = 1 if level == 0 else 8
        out[f'ford_uv_l{level}'] = uvf.detach().numpy()[:, ::st, ::st]
        out[f'ford_jac_l{level}'] = torch.stack([ju, jv, jt]).detach().numpy()[:, :, ::st, ::st]
        out[f'ford_mask_l{level}'] = mask.numpy()[:, ::st, ::st]
        out[f'ford_xyz_l{level}'] = netf.xyz_grds[level][0].detach().numpy()[:, ::st, ::st]
    out.update(geo_pose=torch.stack(pose).numpy(), ford_R=R_FL.numpy(), ford_T=T_FL.numpy())

    # (3) LM_update on small random tensors, option sweep
    B, C, H, W = 2, 8, 8, 16
    sp = rs.standard_normal((B, C, H, W)).astype(np.float32)
    gf = rs.standard_normal((B, C, H, W)).astype(np.float32)
    gc = rs.uniform(0.27, 0.5, size=(B, 1, H, W)).astype(np.float32)
    dj = rs.standard_normal((3, B, C, H, W)).astype(np.float32)
    p0 = rs.uniform(-0.5, 0.5, size=(3, B, 1)).astype(np.float32)
    out.update(lm_sat=sp, lm_grd=gf, lm_conf=gc, lm_jac=dj, lm_pose=p0)
    combos = [dict(), dict(using_weight=1), dict(use_hessian=1), dict(train_damping=1),
              dict(rotation_range=0.0), dict(shift_range_lat=0.0, shift_range_lon=0.0), dict(damping=1e-3)]
    for i, kw in enumerate(combos):
        a = O.default_args(**kw)
        net = mk.LM_S2GP(a)
        torch.autograd.set_detect_anomaly(False)
        if kw.get('train_damping'):
            with torch.no_grad():
                net.damping.copy_(torch.tensor([[0.3, -0.2, 0.1]]))
        torch.manual_seed(5)
        r = net.LM_update(torch.from_numpy(p0[0]), torch.from_numpy(p0[1]), torch.from_numpy(p0[2]),
                          torch.from_numpy(sp), torch.from_numpy(gc), torch.from_numpy(gf), torch.from_numpy(gc),
                          torch.from_numpy(dj))
        out[f'lm_out_{i}'] = torch.stack([x.detach() for x in r]).numpy()
    out['lm_combos'] = np.array([repr(c) for c in combos])
    # out-of-range re-initialisation branch: huge Jacobian-free step via tiny damping & big residual
    a = O.default_args(damping=1e-9)
    net = mk.LM_S2GP(a)
    torch.autograd.set_detect_anomaly(False)
    torch.manual_seed(11)
    r = net.LM_update(torch.from_numpy(p0[0]), torch.from_numpy(p0[1]), torch.from_numpy(p0[2]),
                      torch.from_numpy(sp), torch.from_numpy(gc), torch.from_numpy(gf), torch.from_numpy(gc),
                      torch.from_numpy(dj * 1e-3))
    out['lm_out_reinit'] = torch.stack([x.detach() for x in r]).numpy()

    # (4) VGGUnet on a small image, level 4 (all maps + confidences), non-zero biases
    vrs = np.random.RandomState(21)
    vsd = O.synth_vgg_state(vrs, bias_scale=0.05)
    vnet = VGG.VGGUnet(4)
    vnet.load_state_dict(vsd)
    x = torch.from_numpy(vrs.random_sample((2, 3, 32, 64)).astype(np.float32))
    with torch.no_grad():
        f32, c32 = vnet(x)
        f64, c64 = vnet.double()(x.double())
    for l in range(4):
        out[f'vgg_feat32_l{l}'] = f32[l].numpy()
        out[f'vgg_feat64_l{l}'] = f64[l].numpy()
        out[f'vgg_conf64_l{l}'] = c64[l].numpy()
    np.savez_compressed(os.path.join(GOLD, 'kat_small.npz'), **out)
    print('kat_small.npz written,', len(out), 'arrays')


def run_e2e(mod, cls, args, seed, B, dtype, extra=None, level_first=0, bias_scale=0.0, grd_hw=(256, 1024), sat_a=512,
            hook='LM_update'):
    net = ref_model(mod, cls, args, seed, dtype, bias_scale)
    if tuple(grd_hw) != (256, 1024):
        # the reference hard-codes its ground-plane tables for a 256x1024 input (models_kitti.py:622); rebuild them with
        # its OWN grd_img2cam for the actual level sizes, K still calibrated at 256x1024
        net.xyz_grds = [tuple(t.to(dtype) if t.is_floating_point() else t for t in
                              net.grd_img2cam(grd_hw[0] / 2 ** (3 - l), grd_hw[1] / 2 ** (3 - l), 256, 1024)) for l in range(4)]
    sat, grd, gu, gv, gh = O.synth_images(seed + 100, B, grd_hw=grd_hw, sat_a=sat_a)
    sat, grd = sat.to(dtype), grd.to(dtype)
    torch.manual_seed(seed)
    np.random.seed(seed)                  # args.dropout (models_kitti.py:969) draws from numpy's global generator
    # capture per-step poses through the train-mode return path: run test mode and
    # recover the full [B,N,L] traces by re-running the loop pieces is intrusive; instead
    # monkey-patch loss_func-free access: call train mode with gt to get nothing extra, so
    # we wrap LM_update to log.
    log = []
    orig = getattr(net, hook)

    def wrap(*a, **k):
        r = orig(*a, **k)
        log.append(torch.stack([x.detach()[:, 0] for x in r[:3]], -1))  # [B,3] = (u, v, theta)
        return r
    setattr(net, hook, wrap)
    with torch.no_grad():
        sf, _ = net.SatFeatureNet(sat)
        gf, _ = net.GrdFeatureNet(grd)
        if extra is None:
            res = net(sat, grd, mode='test', level_first=level_first)
        else:
            res = net(sat, grd, extra[2], extra[0].to(dtype), extra[1].to(dtype), mode='test', level_first=level_first)
    trace = torch.stack(log, 1).double().numpy()          # [B, steps, 3] in execution order
    final = torch.stack([r.detach() for r in res], -1).double().numpy()
    return trace, final, feat_stats(sf, B), feat_stats(gf, B)


def gen_screen(mk):
    """Conditioning screen (SURVEY B-5): |fp32 - fp64| of the final pose per seed."""
    args = O.default_args()
    for seed in range(1, 13):
        t0 = time.time()
        t32, f32, _, _ = run_e2e(mk, 'LM_S2GP', args, seed, 1, torch.float32)
        t64, f64, _, _ = run_e2e(mk, 'LM_S2GP', args, seed, 1, torch.float64)
        print(f'seed {seed}: final64 {f64[0]}  |32-64| final {np.abs(f32 - f64).max():.2e} '
              f'trace {np.abs(t32 - t64).max():.2e}  ({time.time() - t0:.1f}s)', flush=True)


def gen_e2e(mk, seeds, B=2):
    args = O.default_args()
    out = {'seeds': np.array(seeds), 'B': np.array(B)}
    for seed in seeds:
        t32, f32, _, _ = run_e2e(mk, 'LM_S2GP', args, seed, B, torch.float32)
        t64, f64, sf, gf = run_e2e(mk, 'LM_S2GP', args, seed, B, torch.float64)
        out[f'trace32_{seed}'], out[f'trace64_{seed}'] = t32, t64
        out[f'final32_{seed}'], out[f'final64_{seed}'] = f32, f64
        for l in range(3):
            out[f'satfeat64_{seed}_l{l}'] = sf[l]
            out[f'grdfeat64_{seed}_l{l}'] = gf[l]
        print(f'kitti seed {seed}: gap {np.abs(t32 - t64).max():.2e} final {f64.tolist()}', flush=True)
    # level-first ordering and the option flags, one seed, B=1
    seed = seeds[0]
    for tag, kw, lf in (('levelfirst', {}, 1), ('weight', dict(using_weight=1), 0),
                        ('hess', dict(use_hessian=1, damping=0.5), 0), ('rot0', dict(rotation_range=0.0), 0),
                        ('dropout', dict(dropout=1), 0)):
        a = O.default_args(**kw)
        t64, f64, _, _ = run_e2e(mk, 'LM_S2GP', a, seed, 1, torch.float64, level_first=lf)
        t32, f32, _, _ = run_e2e(mk, 'LM_S2GP', a, seed, 1, torch.float32, level_first=lf)
        out[f'trace64_{tag}'], out[f'trace32_{tag}'] = t64, t32
        print(f'kitti {tag}: gap {np.abs(t32 - t64).max():.2e}', flush=True)
    np.savez_compressed(os.path.join(GOLD, 'e2e_kitti.npz'), **out)


def gen_optim(mk, seed=1, B=1):
    """The reference's ablation updaters Optimizer='SGD' / 'ADAM' (models_kitti.py:1056-1125), 15 steps each."""
    out = {'seed': np.array(seed), 'B': np.array(B)}
    for opt in ('SGD', 'ADAM'):
        args = O.default_args(Optimizer=opt)
        for dtype, tag in ((torch.float32, '32'), (torch.float64, '64')):
            t, f, _, _ = run_e2e(mk, 'LM_S2GP', args, seed, B, dtype, hook=opt + '_update')
            out[f'trace{tag}_{opt}'] = t
        print(f"{opt}: gap {np.abs(out[f'trace32_{opt}'] - out[f'trace64_{opt}']).max():.2e} last {out[f'trace64_{opt}'][0, -1].tolist()}", flush=True)
    np.savez_compressed(os.path.join(GOLD, 'e2e_kitti_optim.npz'), **out)


def gen_level4(mk, seed=1, B=1):
    """args.level = 4: the LM loop also runs on the full-resolution 16-channel map x24 (20 steps)."""
    args = O.default_args(level=4)
    out = {'seed': np.array(seed), 'B': np.array(B)}
    for dtype, tag in ((torch.float32, '32'), (torch.float64, '64')):
        t, f, _, _ = run_e2e(mk, 'LM_S2GP', args, seed, B, dtype)
        out['trace' + tag], out['final' + tag] = t, f
    print(f"level4: gap {np.abs(out['trace32'] - out['trace64']).max():.2e} final {out['final64'].tolist()}", flush=True)
    np.savez_compressed(os.path.join(GOLD, 'e2e_kitti_level4.npz'), **out)


def gen_hires(mk, seed=1, B=1):
    """BASELINE config 5: grd 512x2048, sat 1024x1024, 10 LM iterations."""
    args = O.default_args(N_iters=10)
    out = {'seed': np.array(seed), 'B': np.array(B)}
    for dtype, tag in ((torch.float32, '32'), (torch.float64, '64')):
        t, f, sf, gf = run_e2e(mk, 'LM_S2GP', args, seed, B, dtype, grd_hw=(512, 2048), sat_a=1024)
        out['trace' + tag], out['final' + tag] = t, f
        if tag == '64':
            for l in range(3):
                out[f'satfeat64_l{l}'], out[f'grdfeat64_l{l}'] = sf[l], gf[l]
    print(f"hires: gap {np.abs(out['trace32'] - out['trace64']).max():.2e} final {out['final64'].tolist()}", flush=True)
    np.savez_compressed(os.path.join(GOLD, 'e2e_kitti_hires.npz'), **out)


def gen_g2s(mk, seeds, B=1):
    """LM_G2SP (ground -> satellite direction), full KITTI shape.  The class calls .cuda() on a few
    helper tensors (models_kitti.py:59,68,73); here those calls are made no-ops so that the unmodified code runs on the
    CPU.  It mixes hard-coded float32 tensors into its matmuls (124-133), so it can only run in fp32."""
    torch.Tensor.cuda = lambda self, *a, **k: self
    args = O.default_args()
    out = {'seeds': np.array(seeds), 'B': np.array(B), 'K': np.array(O.KITTI_K)}
    for seed in seeds:
        net = mk.LM_G2SP(args)
        torch.autograd.set_detect_anomaly(False)
        sd = O.synth_model_state(seed)
        sd['damping'] = args.damping * torch.ones(1, 3)
        net.load_state_dict(sd)
        sat, grd, gu, gv, gh = O.synth_images(seed + 100, B)
        K = torch.tensor([O.KITTI_K], dtype=torch.float32).repeat(B, 1, 1)
        log = []
        orig = net.LM_update

        def wrap(*a, **k):
            r = orig(*a, **k)
            log.append(torch.stack([x.detach()[:, 0] for x in r], -1))
            return r
        net.LM_update = wrap
        with torch.no_grad():
            res = net(sat, grd, K, mode='test')
        out[f'trace32_{seed}'] = torch.stack(log, 1).double().numpy()          # [B, steps, 3] = (u, v, theta)
        out[f'final32_{seed}'] = torch.stack([r.detach() for r in res], -1).double().numpy()
        # the same with confidence weighting
        net.using_weight = 1
        log.clear()
        with torch.no_grad():
            net(sat, grd, K, mode='test')
        out[f'trace32w_{seed}'] = torch.stack(log, 1).double().numpy()
        net.using_weight = 0
        # train-mode tuple
        res = net(sat, grd, K, gu, gv, gh, mode='train')
        out[f'tuple32_{seed}'] = np.stack([np.atleast_1d(r.detach().double().numpy()) if r.dim() else
                                           np.full(3, float(r)) for r in res[:9]])
        if seed == seeds[0]:          # gradient samples from the reference's own autograd (fp32), using_weight=1 + train_damping=1
            net.using_weight = 1
            net.args.train_damping = 1
            net.zero_grad()
            res = net(sat, grd, K, gu, gv, gh, mode='train')
            res[0].backward()
            out['wtuple32'] = np.stack([np.atleast_1d(r.detach().double().numpy()) if r.dim() else
                                        np.full(3, float(r)) for r in res[:9]])
            sdp = dict(net.named_parameters())
            for k in GRAD_KEYS + CONF_KEYS:
                gq = sdp[k].grad.double().reshape(-1)
                out[f'grad32_{k}'] = np.concatenate([[gq.abs().sum().item(), (gq * gq).sum().item()], gq[sample_idx(gq.numel(), 77)].numpy()])
            out['nograd_32'] = np.array([k for k, p in sdp.items() if p.grad is None])
            net.using_weight = 0
            net.args.train_damping = 0
            # the reference class cannot run in fp64 (see the docstring); the fp64 column therefore comes from the
            # RESTATEMENT (oracle/ref_cpu.py, equal to the reference in fp32 to 1e-6 on these very samples) and only
            # serves to measure how much of |hip - reference_fp32| is the reference's own fp32 rounding
            oa = O.default_args(using_weight=1, train_damping=1)
            on = O.LM_G2SP(oa)
            on.load_state_dict(sd)
            on = on.double()
            ro = on(sat.double(), grd.double(), K, gu.double(), gv.double(), gh.double(), mode='train')
            ro[0].backward()
            sdo = dict(on.named_parameters())
            for k in GRAD_KEYS + CONF_KEYS:
                gq = sdo[k].grad.double().reshape(-1)
                out[f'ograd64_{k}'] = np.concatenate([[gq.abs().sum().item(), (gq * gq).sum().item()], gq[sample_idx(gq.numel(), 77)].numpy()])
        print(f'g2s seed {seed}: final {out[f"final32_{seed}"].tolist()} range {np.abs(out[f"trace32_{seed}"]).max():.3f}', flush=True)
    np.savez_compressed(os.path.join(GOLD, 'e2e_kitti_g2s.npz'), **out)


def ford_extra(B):
    R_FL = torch.tensor([[[0., 0., 1.], [1., 0., 0.], [0., 1., 0.]]]).repeat(B, 1, 1)
    T_FL = torch.tensor([[1.7, 0.3, -1.2]]).repeat(B, 1)
    return R_FL, T_FL, 112.64


def gen_ford(mf, seeds, B=1):
    args = O.default_args(N_iters=10)
    out = {'seeds': np.array(seeds), 'B': np.array(B)}
    for seed in seeds:
        t32, f32, _, _ = run_e2e(mf, 'LM_S2GP_Ford', args, seed, B, torch.float32, extra=ford_extra(B))
        t64, f64, _, _ = run_e2e(mf, 'LM_S2GP_Ford', args, seed, B, torch.float64, extra=ford_extra(B))
        out[f'trace32_{seed}'], out[f'trace64_{seed}'] = t32, t64
        out[f'final32_{seed}'], out[f'final64_{seed}'] = f32, f64
        print(f'ford seed {seed}: gap {np.abs(t32 - t64).max():.2e} final {f64.tolist()}', flush=True)
    # level-first ordering (models_ford.py:868-1026) and confidence weighting, first seed
    seed = seeds[0]
    for tag, kw, lf in (('levelfirst', {}, 1), ('weight', dict(using_weight=1), 0), ('dropout', dict(dropout=1), 0),
                        ('level4', dict(level=4, N_iters=5), 0)):
        a = O.default_args(**{'N_iters': 10, **kw})
        t64, _, _, _ = run_e2e(mf, 'LM_S2GP_Ford', a, seed, B, torch.float64, extra=ford_extra(B), level_first=lf)
        t32, _, _, _ = run_e2e(mf, 'LM_S2GP_Ford', a, seed, B, torch.float32, extra=ford_extra(B), level_first=lf)
        out[f'trace64_{tag}'], out[f'trace32_{tag}'] = t64, t32
        print(f'ford {tag}: gap {np.abs(t32 - t64).max():.2e}', flush=True)
    np.savez_compressed(os.path.join(GOLD, 'e2e_ford.npz'), **out)


def gen_ford_level2(mf, seed, B=1):
    """LM_S2GP_Ford(level=2) (models_ford.py:59-65: two ground-plane tables, H/4 and H/2; VGG.py:183-184,198-199 returns
    [x18, x21]): 5 iterations x 2 levels, iteration-first and level-first, fp32 and fp64.  (The KITTI class indexes its tables
    wrongly at level 2 -- SURVEY Appendix A-3 -- so only the Ford model has this mode.)"""
    out = {'seed': np.array(seed), 'B': np.array(B)}
    for tag, lf in (('iterfirst', 0), ('levelfirst', 1)):
        a = O.default_args(N_iters=5, level=2)
        t64, f64, _, _ = run_e2e(mf, 'LM_S2GP_Ford', a, seed, B, torch.float64, extra=ford_extra(B), level_first=lf)
        t32, f32, _, _ = run_e2e(mf, 'LM_S2GP_Ford', a, seed, B, torch.float32, extra=ford_extra(B), level_first=lf)
        out[f'trace64_{tag}'], out[f'trace32_{tag}'], out[f'final64_{tag}'] = t64, t32, f64
        print(f'ford level 2 {tag}: steps {t64.shape[1]} gap {np.abs(t32 - t64).max():.2e} final {f64.tolist()}', flush=True)
    np.savez_compressed(os.path.join(GOLD, 'e2e_ford_l2.npz'), **out)


def gen_ford_gn(mf, seed, B=1):
    """Optimizer='GN' (GN_update, models_ford.py:534-598, dispatched at 775-781): undamped Gauss-Newton without
    renormalising the ground map.  Traces with and without confidence weighting, fp32 and fp64."""
    out = {'seed': np.array(seed), 'B': np.array(B)}
    for tag, kw in (('plain', {}), ('weight', dict(using_weight=1))):
        a = O.default_args(N_iters=5, Optimizer='GN', **kw)
        t64, _, _, _ = run_e2e(mf, 'LM_S2GP_Ford', a, seed, B, torch.float64, extra=ford_extra(B), hook='GN_update')
        t32, _, _, _ = run_e2e(mf, 'LM_S2GP_Ford', a, seed, B, torch.float32, extra=ford_extra(B), hook='GN_update')
        out[f'trace64_{tag}'], out[f'trace32_{tag}'] = t64, t32
        print(f'ford GN {tag}: gap {np.abs(t32 - t64).max():.2e} final {t64[:, -1].tolist()}', flush=True)
    np.savez_compressed(os.path.join(GOLD, 'e2e_ford_gn.npz'), **out)


GRAD_KEYS = ['SatFeatureNet.conv0.weight', 'SatFeatureNet.conv14.weight', 'SatFeatureNet.conv_dec2.3.weight',
             'GrdFeatureNet.conv0.weight', 'GrdFeatureNet.conv14.weight', 'GrdFeatureNet.conv_dec1.1.weight',
             'GrdFeatureNet.conv2.bias']


CONF_KEYS = ['GrdFeatureNet.conf0.1.weight', 'GrdFeatureNet.conf1.1.weight', 'GrdFeatureNet.conf2.1.weight', 'damping']


def gen_train(mk, seed, B=1, weighted=False):
    """Train-mode 14-tuple + gradient samples, fp64.  weighted: using_weight=1 + train_damping=1
    (gradients then also reach the ground branch's confidence heads and the damping parameter)."""
    args = O.default_args(using_weight=1, train_damping=1) if weighted else O.default_args()
    out = {'seed': np.array(seed), 'B': np.array(B)}
    for dtype, tag in ((torch.float64, '64'), (torch.float32, '32')):
        net = ref_model(mk, 'LM_S2GP', args, seed, dtype)
        sat, grd, gu, gv, gh = O.synth_images(seed + 100, B)
        torch.manual_seed(seed)
        res = net(sat.to(dtype), grd.to(dtype), gu.to(dtype), gv.to(dtype), gh.to(dtype), mode='train')
        res[0].backward()
        out['tuple' + tag] = np.stack([np.atleast_1d(r.detach().double().numpy()) if r.dim() else
                                       np.full(3, float(r)) for r in res[:9]])
        sd = dict(net.named_parameters())
        for k in GRAD_KEYS + (CONF_KEYS if weighted else []):
            g = sd[k].grad.double().reshape(-1)
            idx = sample_idx(g.numel(), 77)
            out[f'grad{tag}_{k}'] = np.concatenate([[g.abs().sum().item(), (g * g).sum().item()], g[idx].numpy()])
        out['nograd_' + tag] = np.array([k for k, p in sd.items() if p.grad is None])
        print('train', tag, 'loss', float(res[0]), flush=True)
    np.savez_compressed(os.path.join(GOLD, 'train_kitti_w.npz' if weighted else 'train_kitti.npz'), **out)


def gen_loss(mf):
    """loss_func method 0 of the REAL reference (models_ford.py:1041-1093): values of the nine tensors and, per case, the
    gradient of a random linear functional of ALL nine w.r.t. the three pose tensors (autograd).  Cases: KITTI-like fp32 ground
    truth, Ford-like fp64 ground truth (type promotion: fp64 results), N = 1 (losses[0] is losses[-1]), an exact zero residual."""
    out = {}
    cases = [(32, 5, 3, torch.float32, (100.0, 100.0, 100.0)), (3, 10, 3, torch.float64, (100.0, 50.0, 0.0)),
             (4, 1, 2, torch.float32, (1.0, 2.0, 3.0)), (5, 2, 4, torch.float32, (100.0, 100.0, 10.0))]
    for ci, (B, N, L, gdt, coe) in enumerate(cases):
        g = torch.Generator().manual_seed(100 + ci)
        xs = [torch.randn(B, N, L, generator=g) for _ in range(3)]
        gts = [torch.randn(B, generator=g).to(gdt) for _ in range(3)]
        if ci == 3:
            xs[0][1, 0, 2] = gts[0][1]                # |x - gt| = 0: sign(0) = 0 in abs' backward
        ws = [torch.randn((), generator=g)] + [torch.randn(L, generator=g) for _ in range(8)]
        xr = [x.clone().requires_grad_(True) for x in xs]
        res = mf.loss_func(0, None, None, None, xr[0], xr[1], xr[2], gts[0], gts[1], gts[2], None, None, coe[0], coe[1], coe[2])
        assert all(r is None for r in res[9:]) and len(res) == 13
        f = sum((r.double() * w.double()).sum() for r, w in zip(res[:9], ws))
        gr = torch.autograd.grad(f, xr)
        pre = f'c{ci}_'
        out[pre + 'shape'] = np.array([B, N, L]); out[pre + 'coe'] = np.array(coe)
        for k in range(3):
            out[pre + f'x{k}'] = xs[k].numpy(); out[pre + f'gt{k}'] = gts[k].numpy(); out[pre + f'dx{k}'] = gr[k].numpy()
        for j in range(9):
            out[pre + f'out{j}'] = res[j].detach().numpy(); out[pre + f'w{j}'] = ws[j].numpy()
        # the training step's own use: d(loss)/d(x) alone
        xr = [x.clone().requires_grad_(True) for x in xs]
        res = mf.loss_func(0, None, None, None, xr[0], xr[1], xr[2], gts[0], gts[1], gts[2], None, None, coe[0], coe[1], coe[2])
        res[0].backward()
        for k in range(3):
            out[pre + f'dloss{k}'] = xr[k].grad.numpy()
    out['n_cases'] = np.array(len(cases))
    np.savez_compressed(os.path.join(GOLD, 'loss_kat.npz'), **out)
    print('loss_kat.npz written')


def gen_manifest(mk, mf):
    """State-dict manifest of the REAL reference classes (key -> shape, in state_dict order) for the configurations a
    checkpoint can come from: what `torch.save(net.state_dict())` (train_kitti.py:167-170,409-414) writes and
    `load_state_dict` (546-554) expects.  tests/test_oracle_golden.py holds the product classes to it on CPU."""
    import json
    torch.Tensor.cuda = lambda self, *a, **k: self          # LM_G2SP calls .cuda() on helper tensors
    out = {}
    for tag, mod, cls, kw in (('LM_S2GP', mk, 'LM_S2GP', {}), ('LM_S2GP level4', mk, 'LM_S2GP', dict(level=4)),
                              ('LM_S2GP rot0', mk, 'LM_S2GP', dict(rotation_range=0.0)),
                              ('LM_G2SP', mk, 'LM_G2SP', {}), ('LM_S2GP_Ford', mf, 'LM_S2GP_Ford', {}),
                              ('LM_S2GP_Ford level4', mf, 'LM_S2GP_Ford', dict(level=4))):
        net = getattr(mod, cls)(O.default_args(**kw))
        torch.autograd.set_detect_anomaly(False)
        sd = net.state_dict()
        out[tag] = {'class': cls, 'args': kw, 'n_tensors': len(sd), 'n_params': int(sum(v.numel() for v in sd.values())),
                    'state_dict': [[k, list(v.shape), str(v.dtype)] for k, v in sd.items()]}
        print(tag, out[tag]['n_tensors'], out[tag]['n_params'])
    json.dump(out, open(os.path.join(GOLD, 'state_dict_manifest.json'), 'w'), indent=0)


def gen_results():
    """The files the REAL reference's test1() leaves on disk (train_kitti.py:34-170): Test1_results.mat and the block it
    appends to Test1_results.txt, for prescribed predictions.  train_kitti.py is imported with its data loader and
    tensorboard shimmed out, `load_test1_data` returns three synthetic batches and the "network" returns prescribed poses."""
    import tempfile
    import types
    import scipy.io as scio
    for name in ('torch.utils.tensorboard', 'dataLoader', 'dataLoader.KITTI_dataset'):
        m = types.ModuleType(name)
        sys.modules.setdefault(name, m)
    sys.modules['torch.utils.tensorboard'].SummaryWriter = object
    for fn in ('load_train_data', 'load_test1_data', 'load_test2_data'):
        setattr(sys.modules['dataLoader.KITTI_dataset'], fn, None)
    import train_kitti as tk
    rs = np.random.RandomState(11)
    N, bs = 24, 8
    gt = rs.uniform(-1, 1, (N, 3)).astype(np.float32)                       # (u, v, heading), normalised
    pred = (gt + rs.standard_normal((N, 3)) * np.array([0.04, 0.08, 0.15])).astype(np.float32)
    batches = []
    for i in range(0, N, bs):
        g = torch.from_numpy(gt[i:i + bs])
        batches.append((torch.zeros(bs, 3, 8, 8), torch.zeros(bs, 3, 3), torch.full((bs, 3, 4, 4), float(i)),
                        g[:, 0:1].clone(), g[:, 1:2].clone(), g[:, 2:3].clone(), ['f'] * bs))

    class FakeNet(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(()))

        def forward(self, sat_map, grd, mode='test', **kw):
            i = int(grd[0, 0, 0, 0].item())
            out = torch.from_numpy(pred[i:i + bs]) + self.p
            return out[:, 1], out[:, 0], out[:, 2]                           # (shifts_lat, shifts_lon, theta)

    tk.load_test1_data = lambda *a, **k: batches
    tk.device, tk.mini_batch = torch.device('cpu'), bs
    args = O.default_args(direction='S2GP')
    tmp = tempfile.mkdtemp()
    result = tk.test1(FakeNet(), args, tmp, 1e9, 7)
    mat = scio.loadmat(os.path.join(tmp, 'Test1_results.mat'))
    txt = open(os.path.join(tmp, 'Test1_results.txt')).read()
    print(txt, result)
    np.savez_compressed(os.path.join(GOLD, 'results_kitti.npz'), gt=gt, pred=pred, result=np.float64(result), epoch=7,
                        txt=np.frombuffer(txt.encode(), dtype=np.uint8),
                        **{'mat_' + k: mat[k] for k in ('gt_shifts', 'gt_headings', 'pred_shifts', 'pred_headings')})


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='all')
    ap.add_argument('--seeds', default='')
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    mk, mf, jac, VGG = import_reference()
    seeds = [int(s) for s in a.seeds.split(',')] if a.seeds else None
    if a.only == 'screen':
        gen_screen(mk)
    if a.only in ('all', 'manifest'):
        gen_manifest(mk, mf)
    if a.only in ('all', 'loss'):
        gen_loss(mf)
    if a.only in ('all', 'results'):
        gen_results()
    if a.only in ('all', 'kat'):
        gen_kat(mk, mf, jac, VGG)
    if a.only in ('all', 'e2e'):
        gen_e2e(mk, seeds or [1, 2, 3])
    if a.only in ('all', 'ford'):
        gen_ford(mf, (seeds or [1])[:2])
    if a.only in ('all', 'fordl2'):
        gen_ford_level2(mf, (seeds or [1])[0])
    if a.only in ('all', 'fordgn'):
        gen_ford_gn(mf, (seeds or [1])[0])
    if a.only in ('all', 'train'):
        gen_train(mk, (seeds or [1])[0])
    if a.only in ('all', 'g2s'):
        gen_g2s(mk, seeds or [1, 2])
    if a.only in ('all', 'optim'):
        gen_optim(mk, (seeds or [1])[0])
    if a.only in ('all', 'level4'):
        gen_level4(mk, (seeds or [1])[0])
    if a.only in ('all', 'hires'):
        gen_hires(mk, (seeds or [1])[0])
    if a.only in ('all', 'trainw'):
        gen_train(mk, (seeds or [1])[0], weighted=True)
