"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the committed
golden vectors recorded from the real reference.  Run with `pytest -m gpu` on an MI355X.

Tolerances (stated per SURVEY 8(c)):
  * features (fp32 mode): 1e-5 relative to the map's max-abs, against the fp64 oracle / fp64 golden
  * pose (fp32 mode): |hip - ref_fp64| <= max(tol, 2*|ref_fp32 - ref_fp64|), tol = 5e-6 normalised for the
    shifts (= 1e-4 m at 20 m range) and 5.7e-4 normalised for yaw (= 1e-4 rad at 10 deg)
  * bf16 mode is the throughput mode; its pose deviation is reported and bounded loosely
"""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

TOL_SHIFT, TOL_YAW = 5e-6, 5.7e-4
FORD_BF16_LIMITS = (1.2e-2, 2.1e-2)     # 3x the deviation measured on MI355X: shift 3.94e-3, yaw 6.82e-3 (normalised; 30 LM steps, 2 seeds)


def _dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_library_loaded_and_exports():
    from highlyaccurate_amd import _lib
    lib = _lib.load()
    assert lib.hla_abi_version() == _lib.ABI_VERSION
    for sym in ('hla_vgg_forward', 'hla_vgg_pack_weights', 'hla_s2g_lm_solve', 'hla_grid_sample', 'hla_vgg_workspace_bytes',
                'hla_s2g_workspace_bytes', 'hla_last_error'):
        assert hasattr(lib, sym)


def test_graft_entry_smoke():
    """The driver's round-end smoke(): one small forward in all four arithmetic modes against the fp64 oracle."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__
    __graft_entry__.smoke()


def test_cpu_tensor_is_rejected():
    from highlyaccurate_amd import _lib
    from highlyaccurate_amd.VGG import VGGUnet
    with pytest.raises(_lib.HlaError):
        VGGUnet(3)(torch.zeros(1, 3, 32, 64))


def test_grid_sample_kat(kat):
    from highlyaccurate_amd.jacobian import grid_sample
    d = _dev()
    out, jac = grid_sample(T(kat['gs_img']).to(d), T(kat['gs_uv']).to(d), T(kat['gs_jac']).to(d))
    np.testing.assert_allclose(out.cpu().numpy(), kat['gs_out'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(jac.cpu().numpy(), kat['gs_jac_out'], rtol=0, atol=2e-6)
    o2, j2 = grid_sample(T(kat['gs_img']).to(d), T(kat['gs_uv']).to(d))
    assert j2 is None and torch.equal(o2, out)


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize('precision,tol', [('fp32', 1e-5), ('fp16x3', 1e-5), ('bf16', 3e-2), ('fp16', 4e-3)])
def test_vgg_small_vs_golden(kat, precision, tol):
    """VGGUnet on [2,3,32,64] with non-zero biases against the reference's fp64 maps."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.VGG import VGGUnet
    d = _dev()
    rs = np.random.RandomState(21)
    sd = O.synth_vgg_state(rs, bias_scale=0.05)
    x = T(rs.random_sample((2, 3, 32, 64)).astype(np.float32))
    net = VGGUnet(3, precision=precision)
    net.load_state_dict(sd)
    net = net.to(d)
    with torch.no_grad():
        feats, confs = net(x.to(d))
    for l in range(3):
        f = feats[l].cpu().numpy()
        assert f.shape == kat[f'vgg_feat64_l{l}'].shape
        e = _rel(f, kat[f'vgg_feat64_l{l}'])
        ec = _rel(confs[l].cpu().numpy(), kat[f'vgg_conf64_l{l}'])
        print(f'vgg small {precision} level {l}: feat rel {e:.2e} conf rel {ec:.2e}')
        assert e < tol, (precision, l, e)
        assert ec < max(tol, 2e-6), (precision, l, ec)


@pytest.mark.parametrize('shape', [(1, 40, 72), (2, 56, 200), (3, 88, 104), (1, 8, 8), (2, 264, 40)])
@pytest.mark.parametrize('level', [3, 4])
def test_vgg_16bit_modes_on_ragged_shapes_vs_fp32_mode(shape, level):
    """The 16-bit types have code of their own (conv0's bias inside the matrix product, accumulators that start at the bias, the
    batched activation epilogue) that the randomised fp32-class harness never runs.  Image sizes that are multiples of 8 but not
    of the 8 x 32-pixel conv tiles, non-zero biases: every map of the bf16 / fp16 modes against the library's own exact-fp32
    mode (itself gated against the reference) within the modes' rounding -- a boundary bug shows up as O(1), not as 1e-2."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.VGG import VGGUnet
    d = _dev()
    B, H, W = shape
    rs = np.random.RandomState(1000 * H + W)
    sd = O.synth_vgg_state(rs, bias_scale=0.05)
    x = T(rs.random_sample((B, 3, H, W)).astype(np.float32)).to(d)
    outs = {}
    for precision in ('fp32', 'bf16', 'fp16'):
        net = VGGUnet(level, precision=precision)
        net.load_state_dict(sd)
        net = net.to(d)
        with torch.no_grad():
            f, c = net(x)
        outs[precision] = ([t.float().cpu().numpy() for t in f], [t.float().cpu().numpy() for t in c])
    for precision, tol in (('bf16', 4e-2), ('fp16', 6e-3)):
        for l in range(len(outs['fp32'][0])):
            ref, got = outs['fp32'][0][l], outs[precision][0][l]
            assert got.shape == ref.shape and np.isfinite(got).all()
            e = np.abs(got - ref).max() / np.abs(ref).max()
            ec = np.abs(outs[precision][1][l] - outs['fp32'][1][l]).max()
            assert e < tol and ec < tol, (precision, l, e, ec)


@pytest.mark.parametrize('precision', ['bf16', 'fp16', 'fp32'])
def test_conv0_bias_change_alone_repacks(precision):
    """conv0's bias is part of the packed weight fragments (it rides in the padded k slots of the fused conv0 + conv2 kernel for
    the 16-bit types; include/hla.h): an in-place change of that bias ALONE must invalidate the module's packed-weight cache --
    the maps change, and they equal, bit for bit, those of a fresh module built from the same state dict."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.VGG import VGGUnet
    d = _dev()
    rs = np.random.RandomState(33)
    sd = O.synth_vgg_state(rs, bias_scale=0.05)
    x = T(rs.random_sample((2, 3, 32, 64)).astype(np.float32)).to(d)
    net = VGGUnet(3, precision=precision)
    net.load_state_dict(sd)
    net = net.to(d)
    with torch.no_grad():
        f0 = [f.clone() for f in net(x)[0]]
        net.conv0.bias.add_(0.25 * torch.sign(net.conv0.bias) + 0.1)
        f1 = [f.clone() for f in net(x)[0]]
    assert any(not torch.equal(a, b) for a, b in zip(f0, f1)), 'the changed bias did not reach the kernel'
    fresh = VGGUnet(3, precision=precision)
    fresh.load_state_dict(net.state_dict())
    fresh = fresh.to(d)
    with torch.no_grad():
        f2 = fresh(x)[0]
    for a, b in zip(f1, f2):
        assert torch.equal(a, b)


@pytest.mark.parametrize('precision,tol', [('fp32', 1e-5), ('fp16x3', 1e-5), ('bf16', 3e-2)])
def test_vgg_level4_vs_golden(kat, precision, tol):
    """VGGUnet(level=4): the fourth map x24 (conv_dec3 on cat(up(x21), x2), 16 channels at full resolution) and conf3
    against the reference's fp64 maps; the first three maps must be bit-identical to the level-3 run."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.VGG import VGGUnet
    d = _dev()
    rs = np.random.RandomState(21)
    sd = O.synth_vgg_state(rs, bias_scale=0.05)
    x = T(rs.random_sample((2, 3, 32, 64)).astype(np.float32))
    net = VGGUnet(4, precision=precision)
    net.load_state_dict(sd)
    net = net.to(d)
    with torch.no_grad():
        feats, confs = net(x.to(d))
    assert len(feats) == 4 and tuple(feats[3].shape) == (2, 16, 32, 64) and tuple(confs[3].shape) == (2, 1, 32, 64)
    for l in range(4):
        e = _rel(feats[l].cpu().numpy(), kat[f'vgg_feat64_l{l}'])
        ec = _rel(confs[l].cpu().numpy(), kat[f'vgg_conf64_l{l}'])
        print(f'vgg level4 {precision} map {l}: feat rel {e:.2e} conf rel {ec:.2e}')
        assert e < tol and ec < max(tol, 2e-6), (precision, l, e, ec)
    net3 = VGGUnet(3, precision=precision)
    net3.load_state_dict(sd)
    with torch.no_grad():
        f3, _ = net3.to(d)(x.to(d))
    if precision == 'fp32':
        for l in range(3):
            assert torch.allclose(f3[l], feats[l], rtol=0, atol=0) or _rel(f3[l].cpu().numpy(), feats[l].cpu().numpy()) < 1e-6


def _oracle_small(args, seed, B, grd_hw, sat_a, dtype=torch.float64, ford=False):
    """Random NHWC feature pyramids + the oracle's solve on them."""
    from oracle import ref_cpu as O
    rs = np.random.RandomState(seed)
    Cs = (256, 128, 64)
    sat, grd, conf = [], [], []
    for l in range(3):
        A = sat_a >> (2 - l)
        h, w = grd_hw[0] >> (3 - l), grd_hw[1] >> (3 - l)
        sat.append(T(rs.standard_normal((B, Cs[l], A, A)).astype(np.float32)))
        grd.append(T(rs.standard_normal((B, Cs[l], h, w)).astype(np.float32)))
        conf.append(T(rs.uniform(0.27, 0.5, size=(B, 1, h, w)).astype(np.float32)))
    cls = O.LM_S2GP_Ford if ford else O.LM_S2GP
    net = cls(args, grd_hw=grd_hw).to(dtype)
    return net, sat, grd, conf


def _oracle_normal_eq(onet, sat, grd, conf, pose, level, using_weight, extra=None):
    """The 14 sums of one step, from the oracle's own projection (fp64): S, G, H(6), U(3), V(3)."""
    su, sv, th = pose
    f, _, jac, _, mask = onet.project_map_to_grd(sat[level].double(), None, su, sv, th, level, extra)
    h = grd[level].shape[-2]
    g = grd[level].double() * mask[:, None]
    w = conf[level].double() * mask[:, None] if using_weight else torch.ones_like(g[:, :1])
    f, g, w, jac = f[:, :, h // 2:], g[:, :, h // 2:], w[:, :, h // 2:], jac[:, :, :, h // 2:]
    B = f.shape[0]
    s_, g_, J = f.reshape(B, -1), g.reshape(B, -1), jac.reshape(3, B, -1)
    W = w.expand(-1, f.shape[1], -1, -1).reshape(B, -1)
    out = [(s_ * s_).sum(1), (g_ * g_).sum(1)]
    for p, q in ((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)):
        out.append((W * J[p] * J[q]).sum(1))
    out += [(W * J[p] * s_).sum(1) for p in range(3)] + [(W * J[p] * g_).sum(1) for p in range(3)]
    return torch.stack(out, 1).numpy()


@pytest.mark.parametrize('kw', [dict(), dict(using_weight=1), dict(use_hessian=1, damping=0.5),
                                dict(rotation_range=0.0), dict(shift_range_lat=0.0, shift_range_lon=0.0),
                                dict(level_first=1), dict(N_iters=2, damping=10.0), dict(dropout=1),
                                dict(dropout=1, using_weight=1, level_first=1)])
def test_lm_solve_small_vs_oracle(kw):
    """The fused projection+Jacobian+normal-equation+solve loop on random feature pyramids:
    first-step normal equations (tight) and the whole pose trace."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_S2GP
    d = _dev()
    kw = dict(kw)
    lf = kw.pop('level_first', 0)
    args = O.default_args(**{'N_iters': 3, 'damping': 1.0, **kw})
    B, grd_hw, sat_a = 3, (64, 256), 128
    onet, sat, grd, conf = _oracle_small(args, 3, B, grd_hw, sat_a)
    p0 = T(np.random.RandomState(4).uniform(-0.3, 0.3, size=(B, 3)).astype(np.float32))
    net = LM_S2GP(args).to(d)
    net.keep_normal_eq = True
    nh = lambda t: t.permute(0, 2, 3, 1).contiguous().to(d)
    torch.manual_seed(0)
    np.random.seed(0)                      # args.dropout draws its pixel subsets from numpy's global generator
    trace = net.lm_solve([nh(s) for s in sat], [nh(g) for g in grd], [c[:, 0].contiguous().to(d) for c in conf],
                         grd_hw, None, lf, init_pose=p0).cpu().numpy()
    neq = net.last_normal_eq[0, :, :14].cpu().numpy()
    pose = [p0[:, i:i + 1].double() for i in range(3)]
    if not args.dropout:
        ref_neq = _oracle_normal_eq(onet, sat, grd, conf, pose, 0, args.using_weight)
        e_neq = np.abs(neq - ref_neq).max(0) / np.abs(ref_neq).max(0).clip(1e-30)
        print('normal-eq rel err per sum:', np.array2string(e_neq, precision=1))
        assert e_neq.max() < 2e-6, e_neq
    # whole trace: restart the oracle from the same pose by running its loop manually
    torch.manual_seed(0)
    np.random.seed(0)
    su, sv, th = pose
    L, N = 3, args.N_iters
    order = [(i, l) for l in range(L) for i in range(N)] if lf else [(i, l) for i in range(N) for l in range(L)]
    ref = np.zeros((B, N, L, 3))
    for i, l in order:
        su, sv, th = onet._step(l, sat[l].double(), None, grd[l].double(), conf[l].double(), su, sv, th, None)
        ref[:, i, l] = torch.cat([su, sv, th], 1).numpy()
    err = np.abs(trace - ref).max()
    print('lm small', kw, 'lf', lf, 'trace max err', err, 'ref range', np.abs(ref).max())
    assert np.isfinite(trace).all()
    assert err < 1e-4 * max(1.0, np.abs(ref).max()), (kw, err)


def test_lm_solve_ford_small_vs_oracle():
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_ford import LM_S2GP_Ford
    d = _dev()
    args = O.default_args(N_iters=3, damping=1.0)
    B, grd_hw, sat_a = 2, (64, 256), 128
    onet, sat, grd, conf = _oracle_small(args, 5, B, grd_hw, sat_a, ford=True)
    R_FL = torch.tensor([[[0., 0., 1.], [1., 0., 0.], [0., 1., 0.]]]).repeat(B, 1, 1)
    T_FL = torch.tensor([[1.7, 0.3, -1.2]]).repeat(B, 1)
    torch.manual_seed(0)
    us, vs, ts = onet.solve([s.double() for s in sat], [None] * 3, [g.double() for g in grd],
                            [c.double() for c in conf], (R_FL.double(), T_FL.double(), 112.64), 0)
    ref = torch.stack([us, vs, ts], -1).numpy()
    net = LM_S2GP_Ford(args).to(d)
    nh = lambda t: t.permute(0, 2, 3, 1).contiguous().to(d)
    torch.manual_seed(0)
    trace = net.lm_solve([nh(s) for s in sat], [nh(g) for g in grd], [None] * 3, grd_hw,
                         dict(R_FL=R_FL, T_FL=T_FL, side_m=112.64), 0).cpu().numpy()
    err = np.abs(trace - ref).max()
    print('lm ford small max err', err, 'ref range', np.abs(ref).max())
    assert err < 2e-5 * max(1.0, np.abs(ref).max())


# reduced-precision pose deviation limits (shift, yaw; normalised units) = 3x the deviation measured on MI355X: on golden seed 0
# of e2e_kitti.npz ('bf16', 'fp16'), on the B = 32 bench batch of seed 2 (worst of the four checked samples: 5.26e-4 / 3.20e-3), on
# BASELINE configs[4] in its own dtype (1.98e-5 / 2.37e-4) and on configs[3] (Ford, 30 steps) in ITS secondary dtype
REDUCED_LIMITS = {'bf16': (2.2e-3, 1.1e-2), 'fp16': (1.2e-3, 2.0e-3), 'bf16 B32': (1.6e-3, 9.6e-3), 'fp16 hires': (6.0e-5, 7.2e-4),
                  'bf16 ford': FORD_BF16_LIMITS}


def _pose_gate(got, g64, g32, what):
    """|hip - ref64| <= max(tol, 2*|ref32 - ref64|), componentwise; last axis = (u, v, theta)."""
    tol = np.array([TOL_SHIFT, TOL_SHIFT, TOL_YAW])
    allow = np.maximum(tol, 2 * np.abs(g32 - g64))
    err = np.abs(got - g64)
    worst = (err / allow).max()
    print(f'{what}: max err {err.max():.2e} (ref fp32-fp64 gap {np.abs(g32 - g64).max():.2e}), worst ratio {worst:.2f}')
    assert worst <= 1.0, (what, err.max())


def _run_kitti(seed, B, precision='fp32', level_first=0, grd_hw=(256, 1024), sat_a=512, **kw):
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_S2GP
    d = _dev()
    args = O.default_args(precision=precision, **kw)
    net = LM_S2GP(args)
    net.load_state_dict(O.synth_model_state(seed, rotation_range=args.rotation_range))
    net = net.to(d)
    sat, grd, gu, gv, gh = O.synth_images(seed + 100, B, grd_hw=grd_hw, sat_a=sat_a)
    torch.manual_seed(seed)
    np.random.seed(seed)
    with torch.no_grad():
        res = net(sat.to(d), grd.to(d), mode='test', level_first=level_first)
    return net, res


def _exec_order(trace, level_first):
    """[B,N,L,3] -> [B,steps,3] in execution order (what make_golden.py logged)."""
    B, N, L, _ = trace.shape
    t = trace.permute(0, 2, 1, 3) if level_first else trace
    return t.reshape(B, N * L, 3)


@pytest.mark.parametrize('precision', ['fp32', 'fp16x3'])
def test_e2e_kitti_full_shape_vs_golden(precision):
    """Full KITTI shapes, B=2: pose trace of all 15 steps + feature samples vs the reference.  Both fp32-class modes
    (exact-fp32 MFMA and split fp16) must pass the SAME gate."""
    g = load_golden('e2e_kitti.npz')
    B = int(g['B'])
    from make_idx import sample_idx
    for seed in g['seeds']:
        seed = int(seed)
        net, res = _run_kitti(seed, B, precision=precision)
        trace = _exec_order(net.last_trace, 0).cpu().numpy().astype(np.float64)
        _pose_gate(trace, g[f'trace64_{seed}'], g[f'trace32_{seed}'], f'kitti {precision} seed {seed}')
        final = torch.stack(res, -1).cpu().numpy()
        np.testing.assert_allclose(final, g[f'final64_{seed}'], atol=2e-3)   # ordering check (lat, lon, theta)
        np.testing.assert_array_equal(final[:, [1, 0, 2]], trace[:, -1].astype(np.float32))


@pytest.mark.parametrize('precision', ['fp32', 'fp16x3'])
def test_e2e_kitti_features_vs_golden(precision):
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_S2GP
    from make_idx import sample_idx
    g = load_golden('e2e_kitti.npz')
    B = int(g['B'])
    seed = int(g['seeds'][0])
    d = _dev()
    net = LM_S2GP(O.default_args(precision=precision))
    net.load_state_dict(O.synth_model_state(seed))
    net = net.to(d)
    sat, grd, *_ = O.synth_images(seed + 100, B)
    for name, mod, img in (('sat', net.SatFeatureNet, sat), ('grd', net.GrdFeatureNet, grd)):
        with torch.no_grad():
            feats, _ = mod(img.to(d))
        for l in range(3):
            ref = g[f'{name}feat64_{seed}_l{l}']
            f = feats[l].contiguous().reshape(B, -1).double().cpu()       # NCHW flat order
            idx = sample_idx(f.shape[1], l)
            got = np.concatenate([f.sum(1, keepdim=True).numpy(), (f * f).sum(1, keepdim=True).numpy(), f[:, idx].numpy()], 1)
            scale = np.abs(ref[:, 2:]).max()
            e = np.abs(got[:, 2:] - ref[:, 2:]).max() / scale
            print(f'{precision} {name} level {l}: sampled rel err {e:.2e}, sumsq {got[:, 1]}, sum err {np.abs(got[:, 0] - ref[:, 0]).max():.2e}')
            assert e < 1e-5
            np.testing.assert_allclose(got[:, 1], 1.0, atol=1e-6)


@pytest.mark.parametrize('tag,kw,lf', [('levelfirst', {}, 1), ('weight', dict(using_weight=1), 0),
                                       ('hess', dict(use_hessian=1, damping=0.5), 0),
                                       ('rot0', dict(rotation_range=0.0), 0), ('dropout', dict(dropout=1), 0)])
def test_e2e_kitti_variants_vs_golden(tag, kw, lf):
    g = load_golden('e2e_kitti.npz')
    seed = int(g['seeds'][0])
    net, _ = _run_kitti(seed, 1, level_first=lf, **kw)
    trace = _exec_order(net.last_trace, lf).cpu().numpy().astype(np.float64)
    _pose_gate(trace, g[f'trace64_{tag}'], g[f'trace32_{tag}'], f'kitti {tag}')


@pytest.mark.parametrize('precision', ['fp32', 'fp16x3'])
def test_e2e_ford_full_shape_vs_golden(precision):
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_ford import LM_S2GP_Ford
    g = load_golden('e2e_ford.npz')
    B = int(g['B'])
    d = _dev()
    for seed in g['seeds']:
        seed = int(seed)
        args = O.default_args(N_iters=10, precision=precision)
        net = LM_S2GP_Ford(args)
        net.load_state_dict(O.synth_model_state(seed))
        net = net.to(d)
        sat, grd, *_ = O.synth_images(seed + 100, B)
        R_FL = torch.tensor([[[0., 0., 1.], [1., 0., 0.], [0., 1., 0.]]]).repeat(B, 1, 1)
        T_FL = torch.tensor([[1.7, 0.3, -1.2]]).repeat(B, 1)
        torch.manual_seed(seed)
        with torch.no_grad():
            res = net(sat.to(d), grd.to(d), 112.64, R_FL.to(d), T_FL.to(d), mode='test')
        trace = _exec_order(net.last_trace, 0).cpu().numpy().astype(np.float64)
        _pose_gate(trace, g[f'trace64_{seed}'], g[f'trace32_{seed}'], f'ford {precision} seed {seed}')
        np.testing.assert_array_equal(torch.stack(res, -1).cpu().numpy(), trace[:, -1].astype(np.float32))


def test_e2e_ford_bf16_pose_deviation_bounded():
    """BASELINE configs[3] (Ford shapes, 10 LM iterations = 30 steps) in the dtype bench.py's `secondary` leg runs it in."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_ford import LM_S2GP_Ford
    g = load_golden('e2e_ford.npz')
    B, d = int(g['B']), _dev()
    worst_s = worst_y = 0.0
    for seed in (int(s) for s in g['seeds']):
        net = LM_S2GP_Ford(O.default_args(N_iters=10, precision='bf16'))
        net.load_state_dict(O.synth_model_state(seed))
        net = net.to(d)
        sat, grd, *_ = O.synth_images(seed + 100, B)
        R_FL = torch.tensor([[[0., 0., 1.], [1., 0., 0.], [0., 1., 0.]]]).repeat(B, 1, 1)
        T_FL = torch.tensor([[1.7, 0.3, -1.2]]).repeat(B, 1)
        torch.manual_seed(seed)
        with torch.no_grad():
            net(sat.to(d), grd.to(d), 112.64, R_FL.to(d), T_FL.to(d), mode='test')
        trace = _exec_order(net.last_trace, 0).cpu().numpy().astype(np.float64)
        err = np.abs(trace - g[f'trace64_{seed}'])
        assert np.isfinite(trace).all()
        worst_s, worst_y = max(worst_s, err[..., :2].max()), max(worst_y, err[..., 2].max())
    print(f'ford bf16 pose deviation vs fp64 reference (30 steps, 2 seeds): shift {worst_s:.3e} yaw {worst_y:.3e} (normalised)')
    lim_s, lim_y = REDUCED_LIMITS['bf16 ford']
    assert worst_s < lim_s and worst_y < lim_y, (worst_s, worst_y)


@pytest.mark.parametrize('tag,kw,lf', [('levelfirst', {}, 1), ('weight', dict(using_weight=1), 0), ('dropout', dict(dropout=1), 0),
                                       ('level4', dict(level=4, N_iters=5), 0)])
def test_e2e_ford_variants_vs_golden(tag, kw, lf):
    """Ford level-first ordering (models_ford.py:868-1026) and confidence weighting, 30 steps, full shape."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_ford import LM_S2GP_Ford
    g = load_golden('e2e_ford.npz')
    seed, B = int(g['seeds'][0]), int(g['B'])
    d = _dev()
    net = LM_S2GP_Ford(O.default_args(**{'N_iters': 10, **kw}))
    net.load_state_dict(O.synth_model_state(seed))
    net = net.to(d)
    sat, grd, *_ = O.synth_images(seed + 100, B)
    R_FL = torch.tensor([[[0., 0., 1.], [1., 0., 0.], [0., 1., 0.]]]).repeat(B, 1, 1)
    T_FL = torch.tensor([[1.7, 0.3, -1.2]]).repeat(B, 1)
    torch.manual_seed(seed)
    np.random.seed(seed)
    with torch.no_grad():
        net(sat.to(d), grd.to(d), 112.64, R_FL.to(d), T_FL.to(d), mode='test', level_first=lf)
    trace = _exec_order(net.last_trace, lf).cpu().numpy().astype(np.float64)
    _pose_gate(trace, g[f'trace64_{tag}'], g[f'trace32_{tag}'], f'ford {tag}')


@pytest.mark.parametrize('precision', ['fp32', 'fp16x3'])
@pytest.mark.parametrize('tag,lf', [('iterfirst', 0), ('levelfirst', 1)])
def test_e2e_ford_level2_vs_golden(tag, lf, precision):
    """LM_S2GP_Ford(level=2) -- the two-level pyramid [x18, x21] with the H/4 and H/2 tables (models_ford.py:59-65; VGG.py:198-199):
    10-step traces of both loop orders against the REAL reference's runs, in both fp32-class modes; the KITTI class rejects it
    (its level-2 table indexing is shape-inconsistent in the reference)."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_ford import LM_S2GP_Ford
    from highlyaccurate_amd.models_kitti import LM_S2GP
    g = load_golden('e2e_ford_l2.npz')
    seed, B = int(g['seed']), int(g['B'])
    d = _dev()
    net = LM_S2GP_Ford(O.default_args(N_iters=5, level=2, precision=precision))
    net.load_state_dict(O.synth_model_state(seed))
    net = net.to(d)
    sat, grd, *_ = O.synth_images(seed + 100, B)
    R_FL = torch.tensor([[[0., 0., 1.], [1., 0., 0.], [0., 1., 0.]]]).repeat(B, 1, 1)
    T_FL = torch.tensor([[1.7, 0.3, -1.2]]).repeat(B, 1)
    torch.manual_seed(seed)
    with torch.no_grad():
        res = net(sat.to(d), grd.to(d), 112.64, R_FL.to(d), T_FL.to(d), mode='test', level_first=lf)
    assert tuple(net.last_trace.shape) == (B, 5, 2, 3)
    trace = _exec_order(net.last_trace, lf).cpu().numpy().astype(np.float64)
    _pose_gate(trace, g[f'trace64_{tag}'], g[f'trace32_{tag}'], f'ford level 2 {tag} {precision}')
    np.testing.assert_array_equal(torch.stack(res, -1).cpu().numpy(), trace[:, -1].astype(np.float32))
    with pytest.raises(NotImplementedError):
        LM_S2GP(O.default_args(level=2))


def test_ford_level2_train_step_vs_oracle_autograd():
    """The same model in mode='train' on a small image: loss and every parameter gradient against the fp64 oracle's autograd
    (x15 takes no part in the loop: conv14 still trains through the decoder, conf0 and x15's own gradient are zero)."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_ford import LM_S2GP_Ford
    d = _dev()
    seed, B, hw, A = 7, 2, (64, 256), 128
    args = O.default_args(N_iters=2, level=2, using_weight=1, train_damping=1)
    sd = O.synth_model_state(seed)
    onet = O.LM_S2GP_Ford(args, grd_hw=hw)
    onet.load_state_dict(sd)
    onet = onet.double()
    sat, grd, gu, gv, gt = O.synth_images(seed + 100, B, grd_hw=hw, sat_a=A)
    R_FL = torch.tensor([[[0., 0., 1.], [1., 0., 0.], [0., 1., 0.]]]).repeat(B, 1, 1)
    T_FL = torch.tensor([[1.7, 0.3, -1.2]]).repeat(B, 1)
    gts = [x.double().reshape(-1) for x in (gu, gv, gt)]
    torch.manual_seed(seed)
    ro = onet(sat.double(), grd.double(), 0.22 * A, R_FL.double(), T_FL.double(), *gts, mode='train')
    ro[0].backward()
    net = LM_S2GP_Ford(args)
    net.load_state_dict(sd)
    net = net.to(d)
    torch.manual_seed(seed)
    r = net(sat.to(d), grd.to(d), 0.22 * A, R_FL.to(d), T_FL.to(d), *[x.to(d) for x in gts], mode='train')
    r[0].backward()
    assert len(r) == 14 and len(r[13]) == 2 and tuple(r[13][0].shape) == (B, 1, hw[0] // 4, hw[1] // 4)
    assert abs(float(r[0].detach()) - float(ro[0].detach())) < 1e-5 * abs(float(ro[0].detach()))
    worst, nchk = 0.0, 0
    for (n, p), (_, po) in zip(net.named_parameters(), onet.named_parameters()):
        if po.grad is None or float(po.grad.norm()) < 1e-12:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        a, b = p.grad.double().cpu().flatten(), po.grad.flatten()
        e = float((a - b).norm() / b.norm())
        worst, nchk = max(worst, e), nchk + 1
        assert e < 5e-3, (n, e)          # (max-pool flip noise bounds the encoder tensors, as in the level-3 tests)
    print(f'ford level 2 train step: {nchk} gradients, worst rel L2 {worst:.2e}')
    assert nchk >= 36


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
def test_reduced_precision_pose_deviation_reported(precision):
    g = load_golden('e2e_kitti.npz')
    seed, B = int(g['seeds'][0]), int(g['B'])
    net, _ = _run_kitti(seed, B, precision=precision)
    trace = _exec_order(net.last_trace, 0).cpu().numpy().astype(np.float64)
    err = np.abs(trace - g[f'trace64_{seed}'])
    print(f'{precision} mode pose deviation vs fp64 reference: shift {err[..., :2].max():.3e} yaw {err[..., 2].max():.3e} (normalised)')
    # bounds = 3x what this seed measures on MI355X (bf16: 7.2e-4 shift / 3.7e-3 yaw; fp16: 3.9e-4 / 6.7e-4): a regression of
    # the reduced-precision paths by more than that fails, not only one by two orders of magnitude
    lim_s, lim_y = REDUCED_LIMITS[precision]
    assert np.isfinite(trace).all() and err[..., :2].max() < lim_s and err[..., 2].max() < lim_y, (err[..., :2].max(), err[..., 2].max())


def test_lm_feat16_default_and_opt_out():
    """The reduced-precision inference modes hand the LM loop fp16 feature maps (HLA_VGG_FEAT16 + hla_s2g_level.feat_dtype);
    args.lm_feat16 = 0 keeps fp32 maps.  Both stay within the reduced-precision limits on the golden seed and really are
    different data paths; the fp32-class modes never take the 16-bit path."""
    g = load_golden('e2e_kitti.npz')
    seed, B = int(g['seeds'][0]), int(g['B'])
    for precision in ('bf16', 'fp16'):
        traces = {}
        for f16 in (1, 0):
            net, _ = _run_kitti(seed, B, precision=precision, lm_feat16=f16)
            trace = _exec_order(net.last_trace, 0).cpu().numpy().astype(np.float64)
            err = np.abs(trace - g[f'trace64_{seed}'])
            print(f'{precision}, lm_feat16={f16}: shift {err[..., :2].max():.3e} yaw {err[..., 2].max():.3e}')
            lim_s, lim_y = REDUCED_LIMITS[precision]
            assert np.isfinite(trace).all() and err[..., :2].max() < lim_s and err[..., 2].max() < lim_y
            traces[f16] = net.last_trace.clone()
        assert not torch.equal(traces[0], traces[1])
    for precision in ('fp32', 'fp16x3'):                                # the parity modes: lm_feat16 is ignored
        a, _ = _run_kitti(seed, 1, precision=precision, lm_feat16=1)
        b, _ = _run_kitti(seed, 1, precision=precision, lm_feat16=0)
        assert torch.equal(a.last_trace, b.last_trace)
    with pytest.raises(ValueError):                                     # fp32-class modes keep fp32 maps: asking for more is an error
        from oracle import ref_cpu as O
        from highlyaccurate_amd.VGG import VGGUnet, vgg_forward_nhwc
        vgg_forward_nhwc(VGGUnet(3, precision='fp16x3').to(_dev()), torch.rand(1, 3, 32, 64, device=_dev()), defer_norm=True, feat16=True)


@pytest.mark.parametrize('opt', ['SGD', 'ADAM'])
def test_e2e_ablation_optimisers_vs_golden(opt):
    """Optimizer='SGD' / 'ADAM', the reference's ablation updaters (forward only): 15-step trace vs the reference."""
    g = load_golden('e2e_kitti_optim.npz')
    seed, B = int(g['seed']), int(g['B'])
    net, _ = _run_kitti(seed, B, Optimizer=opt)
    trace = _exec_order(net.last_trace, 0).cpu().numpy().astype(np.float64)
    if opt == 'SGD':
        _pose_gate(trace, g[f'trace64_{opt}'], g[f'trace32_{opt}'], f'kitti {opt}')
    else:
        # ADAM's normalised steps amplify rounding ~10x per step (tests/test_oracle_golden.py): gate the first steps tightly
        # and the whole trace against the reference's own fp32-vs-fp64 gap (6.7e-4)
        err = np.abs(trace - g['trace64_ADAM'])
        gap = np.abs(g['trace32_ADAM'] - g['trace64_ADAM']).max()
        print(f'kitti ADAM: max err per step {np.array2string(err.max((0, 2)), precision=1)} (reference fp32-fp64 gap {gap:.2e})')
        assert err[:, :3].max() < 1e-5 and err.max() < 0.5 * gap


def test_e2e_kitti_level4_vs_golden():
    """args.level = 4 (inference): 4 levels x 5 iterations = 20 LM steps, the last level on the full-resolution x24 map."""
    g = load_golden('e2e_kitti_level4.npz')
    seed, B = int(g['seed']), int(g['B'])
    net, res = _run_kitti(seed, B, level=4)
    trace = _exec_order(net.last_trace, 0).cpu().numpy().astype(np.float64)
    assert trace.shape == g['trace64'].shape == (B, 20, 3)
    _pose_gate(trace, g['trace64'], g['trace32'], 'kitti level 4')


def test_vgg_backward_level4_vs_oracle_autograd():
    """VGGUnet(level=4) backward: gradients arriving at x24 (16 real channels of the padded map) and conf3 flow through the
    zero-padded conv_dec3 layers, the full-resolution skip connection into conv2 (unpool + skip merge) and into x21."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.VGG import VGGUnet, vgg_forward_nhwc, vgg_backward_nhwc
    d = _dev()
    rs = np.random.RandomState(41)
    sd = O.synth_vgg_state(rs, bias_scale=0.05)
    x = T(rs.random_sample((2, 3, 32, 64)).astype(np.float32))
    onet = O.VGGUnet(4)
    onet.load_state_dict(sd)
    onet = onet.double()
    feats64, confs64 = onet(x.double())
    ups = [T(rs.standard_normal(tuple(f.shape))) for f in feats64]
    cups = [T(rs.standard_normal(tuple(c.shape))) * 30.0 for c in confs64]
    loss = sum((u * f).sum() for u, f in zip(ups, feats64)) + sum((u * c).sum() for u, c in zip(cups, confs64))
    loss.backward()
    ref = {k: p.grad for k, p in onet.named_parameters()}
    net = VGGUnet(4, precision='fp32')
    net.load_state_dict(sd)
    net = net.to(d)
    feats, confs, inv, ctx = vgg_forward_nhwc(net, x.to(d), want_conf=True, defer_norm=True, save_for_backward=True)
    dfe = [u.permute(0, 2, 3, 1).contiguous().float().to(d) for u in ups]
    dfe[3] = torch.cat([dfe[3], torch.randn(2, 32, 64, 48, device=d)], -1).contiguous()     # the padded channels' gradient is ignored
    grads = vgg_backward_nhwc(net, ctx, dfe, confs, [u[:, 0].contiguous().float().to(d) for u in cups])
    assert set(grads) == set(k for k, v in ref.items() if v is not None)
    for k, g in grads.items():
        r = ref[k].numpy()
        assert tuple(g.shape) == r.shape, k
        e = np.abs(g.cpu().double().numpy() - r).max() / max(np.abs(r).max(), 1e-30)
        print(f'vgg bwd level4 {k:24s} rel err max {e:.2e} (max |ref| {np.abs(r).max():.2e})')
        assert e < 2e-4, (k, e)


def test_level4_train_step_vs_oracle_autograd_small():
    """LM_S2GP with args.level = 4, mode='train' under autograd on a reduced shape: loss and every parameter gradient."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_S2GP
    d = _dev()
    args = O.default_args(level=4, N_iters=2, using_weight=1)
    B, grd_hw, sat_a = 2, (64, 256), 128
    sd = O.synth_model_state(4, bias_scale=0.02)
    sat, grd, gu, gv, gh = O.synth_images(9, B, grd_hw=grd_hw, sat_a=sat_a)
    on = O.LM_S2GP(args, grd_hw=grd_hw)
    on.load_state_dict(sd)
    on = on.double()
    torch.manual_seed(0)
    ro = on(sat.double(), grd.double(), gu.double(), gv.double(), gh.double(), mode='train')
    ro[0].backward()
    ref = {k: p.grad for k, p in on.named_parameters()}
    net = LM_S2GP(args)
    net.load_state_dict(sd)
    net = net.to(d).train()
    torch.manual_seed(0)
    r = net(sat.to(d), grd.to(d), gu.to(d), gv.to(d), gh.to(d), mode='train')
    assert abs(float(r[0].detach()) - float(ro[0])) < 1e-4 * abs(float(ro[0])) and len(r[13]) == 4
    r[0].backward()
    for k, p in net.named_parameters():
        assert (p.grad is None) == (ref[k] is None), k
        if p.grad is None:
            continue
        rr = ref[k].numpy()
        e = np.abs(p.grad.cpu().double().numpy() - rr).max() / max(np.abs(rr).max(), 1e-30)
        print(f'level4 train grad {k:36s} rel err {e:.2e} (max |ref| {np.abs(rr).max():.2e})')
        assert e < 5e-3, (k, e)


def test_e2e_hires_config5_vs_golden():
    """BASELINE config 5: grd 512x2048, sat 1024x1024, 10 LM iterations (30 steps).  The reference hard-codes its
    ground-plane tables for 256x1024 (models_kitti.py:622); the golden was recorded with its own grd_img2cam re-run for
    the actual level sizes (oracle/make_golden.py gen_hires).  fp32 mode is gated, fp16 (the config's dtype) reported."""
    from make_idx import sample_idx
    g = load_golden('e2e_kitti_hires.npz')
    seed, B = int(g['seed']), int(g['B'])
    net, res = _run_kitti(seed, B, grd_hw=(512, 2048), sat_a=1024, N_iters=10)
    trace = _exec_order(net.last_trace, 0).cpu().numpy().astype(np.float64)
    assert trace.shape == g['trace64'].shape == (B, 30, 3)
    _pose_gate(trace, g['trace64'], g['trace32'], 'kitti hires')
    d = _dev()
    from oracle import ref_cpu as O
    sat, grd, *_ = O.synth_images(seed + 100, B, grd_hw=(512, 2048), sat_a=1024)
    for name, mod, img in (('sat', net.SatFeatureNet, sat), ('grd', net.GrdFeatureNet, grd)):
        with torch.no_grad():
            feats, _ = mod(img.to(d))
        for l in range(3):
            ref = g[f'{name}feat64_l{l}']
            f = feats[l].contiguous().reshape(B, -1).double().cpu()
            got = f[:, sample_idx(f.shape[1], l)].numpy()
            e = np.abs(got - ref[:, 2:]).max() / np.abs(ref[:, 2:]).max()
            print(f'hires {name} level {l}: sampled rel err {e:.2e}')
            assert e < 1e-5
    net16, _ = _run_kitti(seed, B, precision='fp16', grd_hw=(512, 2048), sat_a=1024, N_iters=10)
    t16 = _exec_order(net16.last_trace, 0).cpu().numpy().astype(np.float64)
    err = np.abs(t16 - g['trace64'])
    print(f'hires fp16 pose deviation vs fp64 reference: shift {err[..., :2].max():.3e} yaw {err[..., 2].max():.3e}')
    # configs[4]'s own dtype, bounded at 3x what it measures on MI355X (shift 1.98e-5, yaw 2.37e-4 normalised = 4e-4 m / 4e-5 rad)
    lim_s, lim_y = REDUCED_LIMITS['fp16 hires']
    assert np.isfinite(t16).all() and err[..., :2].max() < lim_s and err[..., 2].max() < lim_y


def test_determinism_and_batch_independence():
    """Same input twice -> bitwise identical; a sample's pose does not depend on its batch mates
    (the path shards over the batch with no exchange)."""
    from oracle import ref_cpu as O
    n1, r1 = _run_kitti(2, 3)
    t1 = n1.last_trace.clone()
    n2, r2 = _run_kitti(2, 3)
    assert torch.equal(t1, n2.last_trace)
    d = _dev()
    sat, grd, *_ = O.synth_images(102, 3)
    with torch.no_grad():
        n1(sat[1:2].to(d), grd[1:2].to(d), mode='test')
    assert torch.equal(t1[1:2], n1.last_trace)
    # B >= 8 switches the LM kernels to the XCD-affine block map (a sample's tiles stay on one XCD; B = 11 leaves idle
    # blocks in the last group of 8): still bitwise equal to the same samples run alone, in forward AND in training mode
    sat, grd, gu, gv, gh = O.synth_images(77, 11, grd_hw=(64, 256), sat_a=128)
    with torch.no_grad():
        n1(sat.to(d), grd.to(d), mode='test')
        big = n1.last_trace.clone()
        n1(sat[9:11].to(d), grd[9:11].to(d), mode='test')
    assert torch.equal(big[9:11], n1.last_trace)
    n1.zero_grad()
    n1(sat.to(d), grd.to(d), gu.to(d), gv.to(d), gh.to(d), mode='train')[0].backward()
    assert all(torch.isfinite(p.grad).all() for p in n1.parameters() if p.grad is not None)
    # Full KITTI shape, B = 9 (one stream, XCD-affine LM block map) against the same samples alone (B <= 4: the two extractors on
    # two streams, the plain block map): bitwise equal
    sat, grd, *_ = O.synth_images(31, 9)
    with torch.no_grad():
        n1(sat.to(d), grd.to(d), mode='test')
        big = n1.last_trace.clone()
        for k in (0, 8):
            n1(sat[k:k + 1].to(d), grd[k:k + 1].to(d), mode='test')
            assert torch.equal(big[k:k + 1], n1.last_trace), k


@pytest.mark.parametrize('form', ['three_tensors', 'trace_columns'])
def test_pose_loss_kernels_vs_reference_golden(form):
    """loss_func method 0 on the device = hla_pose_loss / hla_pose_loss_bwd (one launch each way) against vectors recorded from
    the REAL reference's loss_func and its autograd: nine tensors (fp32; fp64 with the Ford loader's fp64 ground truth), the
    gradient of a random functional of all nine, d(loss)/d(pose); N = 1 and an exactly-zero residual included.  `trace_columns`
    is the models' own call (the poses are columns of the LM trace, one d_trace comes back)."""
    from highlyaccurate_amd import _s2gp
    from highlyaccurate_amd.models_kitti import loss_func
    g = load_golden('loss_kat.npz')
    d = _dev()
    for ci in range(int(g['n_cases'])):
        pre = f'c{ci}_'
        xs = [T(g[pre + f'x{k}']).to(d) for k in range(3)]
        gts = [T(g[pre + f'gt{k}']).to(d) for k in range(3)]
        coe = [float(c) for c in g[pre + 'coe']]
        for functional in (True, False):
            if form == 'three_tensors':
                leaves = [x.clone().requires_grad_(True) for x in xs]
                res = loss_func(0, None, None, None, *leaves, *gts, None, None, *coe)
            else:
                cols = (1, 0, 2)
                tr = torch.empty(*xs[0].shape, 3, device=d)
                for k in range(3):
                    tr[..., cols[k]] = xs[k]
                leaves = [tr.requires_grad_(True)]
                res = _s2gp.loss_from_trace(0, leaves[0], cols, *gts, *coe)
            assert len(res) == 13 and all(r is None for r in res[9:])
            assert res[0].grad_fn is not None and 'PoseLossFn' in type(res[0].grad_fn).__name__      # the HIP form, not tensor ops
            for j in range(9):
                ref = g[pre + f'out{j}']
                assert res[j].dtype == T(ref).dtype and tuple(res[j].shape) == ref.shape
                # (fp32: the batch means are summed in another order than torch's reduction -- an ulp of values up to ~300; the
                #  differences losses[0] - losses[-1] inherit that absolutely)
                f32 = ref.dtype == np.float32
                np.testing.assert_allclose(res[j].detach().cpu().numpy(), ref, rtol=2e-6 if f32 else 1e-12, atol=4e-5 if f32 else 1e-12)
            if functional:
                f = sum((r.double() * T(g[pre + f'w{j}']).to(d).double()).sum() for j, r in enumerate(res[:9]))
                gr = torch.autograd.grad(f, leaves)
                key = 'dx'
            else:
                res[0].backward()
                gr = [l.grad for l in leaves]
                key = 'dloss'
            for k in range(3):
                got = gr[k] if form == 'three_tensors' else gr[0][..., (1, 0, 2)[k]]
                np.testing.assert_allclose(got.cpu().numpy(), g[pre + f'{key}{k}'], rtol=2e-6, atol=1e-8)
    # a CPU call runs the reference's tensor ops on the CPU tensors (no device work, no kernel): same values
    xs = [T(g['c0_x%d' % k]) for k in range(3)]
    gts = [T(g['c0_gt%d' % k]) for k in range(3)]
    res = loss_func(0, None, None, None, *xs, *gts, None, None, 100, 100, 100)
    np.testing.assert_allclose(float(res[0]), float(g['c0_out0']), rtol=1e-6)


def test_pose_loss_dispatch_and_error_codes():
    """What does NOT go through the kernels (more than 512 (iteration, level) pairs, fp16 poses, ground truth that requires grad) is
    evaluated with the reference's tensor ops -- same values; the C entry refuses bad sizes with an error code and a message."""
    import ctypes as C
    from highlyaccurate_amd import _lib, _s2gp
    from highlyaccurate_amd.models_kitti import loss_func
    d = _dev()
    g = torch.Generator().manual_seed(3)
    for shape, dt, gt_grad in (((2, 100, 6), torch.float32, False), ((3, 4, 3), torch.float16, False), ((3, 4, 3), torch.float32, True)):
        xs = [torch.randn(*shape, generator=g).to(d).to(dt).requires_grad_(True) for _ in range(3)]
        gts = [torch.randn(shape[0], generator=g).to(d).requires_grad_(gt_grad) for _ in range(3)]
        res = loss_func(0, None, None, None, *xs, *gts, None, None, 100, 50, 10)
        assert 'PoseLossFn' not in type(res[0].grad_fn).__name__
        ref = _s2gp._loss_tensor_ops(*xs, *gts, 100, 50, 10)
        for a, b in zip(res[:9], ref):
            assert torch.equal(a, b)
    a = _lib.PoseLossArgs()
    a.B, a.N, a.L = 2, 100, 6
    out = torch.empty(49, device=d)
    rc = _lib.load().hla_pose_loss(C.byref(a), _lib.ptr(out), _lib.stream_ptr())
    assert rc != 0 and b'N * L' in _lib.load().hla_last_error()


def test_train_mode_forward_values_vs_golden():
    """mode='train' 14-tuple values (no autograd yet) against the reference's fp64 tuple."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_S2GP
    g = load_golden('train_kitti.npz')
    seed, B = int(g['seed']), int(g['B'])
    d = _dev()
    net = LM_S2GP(O.default_args())
    net.load_state_dict(O.synth_model_state(seed))
    net = net.to(d)
    sat, grd, gu, gv, gh = O.synth_images(seed + 100, B)
    torch.manual_seed(seed)
    with torch.no_grad():
        res = net(sat.to(d), grd.to(d), gu.to(d), gv.to(d), gh.to(d), mode='train')
    assert len(res) == 14 and res[9] is None and res[12] is None and len(res[13]) == 3
    ref = g['tuple64']
    assert abs(float(res[0]) - ref[0][0]) < 1e-3 * abs(ref[0][0])
    for i in range(1, 9):
        np.testing.assert_allclose(res[i].cpu().numpy(), ref[i], rtol=1e-3, atol=2e-3)
    assert tuple(res[13][0].shape) == (B, 1, 32, 128)


@pytest.mark.parametrize('precision', ['fp32', 'fp16x3'])
def test_train_step_gradients_vs_reference_golden(precision):
    """mode='train' under autograd: loss.backward() runs the HIP backward (LM loop + both VGGs).  Gradients are checked
    against samples recorded from the REAL reference's autograd (fp64 run), full KITTI shape, B=1.  fp16x3: the split-fp16
    forward saves fp32 activations and the backward runs split-fp16 data- and weight-gradient kernels on them (three fp16 MFMAs
    per product) -- training in the matched-accuracy mode must meet the same gradient gates as the exact-fp32 mode."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_S2GP
    from make_idx import sample_idx
    g = load_golden('train_kitti.npz')
    seed, B = int(g['seed']), int(g['B'])
    d = _dev()
    net = LM_S2GP(O.default_args(precision=precision))
    net.load_state_dict(O.synth_model_state(seed))
    net = net.to(d).train()
    sat, grd, gu, gv, gh = O.synth_images(seed + 100, B)
    torch.manual_seed(seed)
    res = net(sat.to(d), grd.to(d), gu.to(d), gv.to(d), gh.to(d), mode='train')
    assert abs(float(res[0].detach()) - g['tuple64'][0][0]) < 1e-3 * abs(g['tuple64'][0][0])
    res[0].backward()
    named = dict(net.named_parameters())
    nograd = set(str(k) for k in g['nograd_64'])
    for k, p in named.items():
        assert (p.grad is None) == (k in nograd), k          # same 13 parameters without gradient as the reference (B-8)
    keys = [k[len('grad64_'):] for k in g.files if k.startswith('grad64_')]
    worst = 0.0
    for k in keys:
        ref = g['grad64_' + k]
        gr = named[k].grad.double().reshape(-1).cpu()
        idx = sample_idx(gr.numel(), 77)
        got = np.concatenate([[gr.abs().sum().item(), (gr * gr).sum().item()], gr[idx].numpy()])
        gap = np.abs(g['grad32_' + k][2:] - ref[2:]).max()        # the reference's own fp32-vs-fp64 gradient gap
        scale = np.abs(ref[2:]).max()
        e = np.abs(got[2:] - ref[2:]).max()
        print(f'train grad [{precision}] {k:36s} max err {e:.2e} (ref fp32 gap {gap:.2e}, scale {scale:.2e}); l1 {got[0]:.4e} vs {ref[0]:.4e}')
        # Gradients that pass through a max-pool carry "flip noise": one fp32 near-tie (relative gap < 1e-6, measured
        # with tests/diag/diag_argmax.py: exactly 1 window per map differs from the fp64 run) reroutes one gradient element,
        # which moves a weight gradient by ~1/sqrt(#pixels) ~ 1e-3 relative.  The reference's own fp32-vs-fp64 gap shows
        # the same effect.  conv_dec2.* sit above every pool in the backward order and must be tight.
        rel_tol = 2e-4 if 'conv_dec2' in k else 5e-3
        assert e <= max(rel_tol * scale, 3 * gap), (k, e, gap, scale)
        # L1 norms: 1e-3 relative (fp32 rounding flips a few max-pool near-ties differently than the fp64 run does)
        assert abs(got[0] - ref[0]) <= max(2e-3 * ref[0], 3 * abs(g['grad32_' + k][0] - ref[0]))
        worst = max(worst, e / scale)
    print('train grads worst rel err', worst)
    # a second step after an optimizer update must re-pack the weights (version counters) and still run
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    opt.step(); opt.zero_grad()
    torch.manual_seed(seed)
    res2 = net(sat.to(d), grd.to(d), gu.to(d), gv.to(d), gh.to(d), mode='train')
    res2[0].backward()
    assert torch.isfinite(res2[0]) and float(res2[0]) != float(res[0])


def test_test_mode_outputs_are_differentiable():
    """train_kitti.py:63-64 calls .backward() on the test-mode outputs."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_S2GP
    d = _dev()
    net = LM_S2GP(O.default_args(N_iters=1)).to(d)
    sat, grd, gu, *_ = O.synth_images(3, 1, grd_hw=(64, 256), sat_a=128)
    lat, lon, th = net(sat.to(d), grd.to(d), mode='test')
    torch.mean(lat - gu.to(d)[:, 0]).backward()
    assert net.SatFeatureNet.conv0.weight.grad is not None


def test_full_bench_config_runs_and_is_consistent():
    """BASELINE config 2 shape (B=32, bf16): finite, deterministic, and each sample equals its B=1 run."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_S2GP
    d = _dev()
    net = LM_S2GP(O.default_args(precision='bf16'))
    net.load_state_dict(O.synth_model_state(1))
    net = net.to(d)
    rs = np.random.RandomState(9)
    gen = torch.Generator(device='cpu').manual_seed(9)
    sat = torch.rand(32, 3, 512, 512, generator=gen).to(d)
    grd = torch.rand(32, 3, 256, 1024, generator=gen).to(d)
    with torch.no_grad():
        torch.manual_seed(1)
        net(sat, grd, mode='test')
        t = net.last_trace.clone()
        torch.manual_seed(1)
        net(sat[5:6], grd[5:6], mode='test')
        t5 = net.last_trace.clone()
    assert torch.isfinite(t).all()
    assert torch.equal(t[5:6], t5)


_ORACLE_B32 = {}


@pytest.mark.parametrize('precision', ['fp32', 'fp16x3', 'bf16'])
def test_bench_batch_per_sample_vs_oracle(precision):
    """BASELINE configs[1] batch size (B = 32, what bench.py times): four samples of the batch are compared ONE BY ONE with
    the oracle run on that sample alone (fp64, and fp32 for the reference's own rounding gap).  The two fp32-class modes
    pass the parity gate; bf16 stays within its measured-deviation limits."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_S2GP
    d = _dev()
    seed = 2
    args = O.default_args(precision=precision)
    net = LM_S2GP(args)
    sd = O.synth_model_state(seed)
    net.load_state_dict(sd)
    net = net.to(d)
    sat, grd, *_ = O.synth_images(seed + 300, 32)
    torch.manual_seed(seed)
    with torch.no_grad():
        net(sat.to(d), grd.to(d), mode='test')
    trace = net.last_trace.cpu().numpy().astype(np.float64)              # [32,5,3,(u,v,theta)]
    assert np.isfinite(trace).all()
    for k in (0, 7, 19, 31):
        if (seed, k) not in _ORACLE_B32:                 # the oracle runs are shared by the three precisions
            o32 = O.LM_S2GP(O.default_args())
            o32.load_state_dict(sd)
            o64 = O.LM_S2GP(O.default_args())
            o64.load_state_dict(sd)
            o64 = o64.double()
            with torch.no_grad():
                torch.manual_seed(seed)
                o64(sat[k:k + 1].double(), grd[k:k + 1].double(), mode='test')
                torch.manual_seed(seed)
                o32(sat[k:k + 1], grd[k:k + 1], mode='test')
            _ORACLE_B32[(seed, k)] = (torch.stack([o64.trace[1], o64.trace[0], o64.trace[2]], -1)[0].numpy(),   # (lon, lat, theta) = (u, v, theta)
                                      torch.stack([o32.trace[1], o32.trace[0], o32.trace[2]], -1)[0].double().numpy())
        t64, t32 = _ORACLE_B32[(seed, k)]
        if precision in ('fp32', 'fp16x3'):
            _pose_gate(trace[k], t64, t32, f'B=32 {precision} sample {k}')
        else:
            err = np.abs(trace[k] - t64)
            print(f'B=32 {precision} sample {k}: shift {err[..., :2].max():.3e} yaw {err[..., 2].max():.3e}')
            lim_s, lim_y = REDUCED_LIMITS[precision + ' B32']        # recorded for this very batch: no extra slack
            assert err[..., :2].max() < lim_s and err[..., 2].max() < lim_y


def test_split_fp16_scaling_is_robust_and_sample_local():
    """fp16x3 feeds the matrix cores (hi, lo) fp16 pairs of s*x with a power-of-two scale s per tensor and SAMPLE.
    (1) Dynamic range: weights scaled by 2^-9 / 2^+7 layer by layer and inputs far from [0,1] (x 3e-4 and x 5e3 in one
        batch) must still give fp32-class maps -- a fixed or batch-wide scale would push small samples into the fp16
        subnormals or overflow the large one.
    (2) The scale is per sample: a sample's result is bitwise independent of its batch mates.
    (3) An all-zero image gives finite maps (the bias path) and does not disturb its batch mates."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.VGG import VGGUnet
    d = _dev()
    rs = np.random.RandomState(33)
    sd = O.synth_vgg_state(rs, bias_scale=0.05)
    names = ['conv0', 'conv2', 'conv5', 'conv7', 'conv10', 'conv12', 'conv14', 'conv_dec1.1', 'conv_dec1.3', 'conv_dec2.1', 'conv_dec2.3']
    for i, n in enumerate(names):                     # alternate tiny / large layers; biases scaled with the running gain
        sd[n + '.weight'] = sd[n + '.weight'] * (2.0 ** -9 if i % 2 == 0 else 2.0 ** 7)
    x = rs.random_sample((4, 3, 40, 72)).astype(np.float32)
    x[0] *= 3e-4
    x[1] *= 5e3
    x[3] = 0.0
    x = T(x)
    onet = O.VGGUnet(3)
    onet.load_state_dict(sd)
    with torch.no_grad():
        ref, _ = onet.double()(x.double())            # L2-normalised maps, like the module's
    net = VGGUnet(3, precision='fp16x3')
    net.load_state_dict(sd)
    net = net.to(d)
    with torch.no_grad():
        feats, confs = net(x.to(d))
        for l in range(3):
            f = feats[l].cpu().double()
            assert torch.isfinite(f).all()
            for b in range(4):
                e = (f[b] - ref[l][b]).abs().max() / ref[l][b].abs().max().clamp_min(1e-300)
                print(f'split robustness: sample {b} level {l} rel err {e:.2e}')
                assert e < 1e-5, (b, l, float(e))
        f1, _ = net(x[1:2].to(d))
        f2, _ = net(x[2:4].to(d))
    for l in range(3):
        assert torch.equal(f1[l][0], feats[l][1]) and torch.equal(f2[l], feats[l][2:4])


@pytest.mark.parametrize('kw', [dict(), dict(using_weight=1), dict(use_hessian=1, damping=0.5),
                                dict(rotation_range=0.0), dict(level_first=1), dict(train_damping=1),
                                dict(train_damping=1, use_hessian=1), dict(dropout=1, using_weight=1),
                                dict(deterministic_backward=1), dict(deterministic_backward=1, using_weight=1, train_damping=1),
                                dict(deterministic_backward=1, level_first=1)])
def test_lm_backward_small_vs_oracle_autograd(kw):
    """hla_s2g_lm_solve_bwd against torch autograd through the fp64 oracle's unrolled loop
    (gather values, bilinear weights, norms, J^T W J, inverse, pose->uv chain of later steps)."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_S2GP
    d = _dev()
    kw = dict(kw)
    lf = kw.pop('level_first', 0)
    args = O.default_args(**{'N_iters': 2, 'damping': 1.0, **kw})
    B, grd_hw, sat_a = 2, (64, 256), 128
    onet, sat, grd, conf = _oracle_small(args, 7, B, grd_hw, sat_a)
    dparam = torch.tensor([[0.55, 0.6, 0.5]])          # lambda = 10^(-6+11*sigmoid(.)) ~ 1 .. 3
    if args.train_damping:
        with torch.no_grad():
            onet.damping.copy_(dparam.double())
    p0 = T(np.random.RandomState(5).uniform(-0.2, 0.2, size=(B, 3)).astype(np.float32))
    L, N = 3, args.N_iters
    coef = T(np.random.RandomState(6).standard_normal((B, N, L, 3)))         # loss = sum(coef * trace)
    # oracle: leaves in fp64
    sat64 = [s.double().requires_grad_(True) for s in sat]
    grd64 = [g.double().requires_grad_(True) for g in grd]
    conf64 = [c.double().requires_grad_(True) for c in conf]
    su, sv, th = [p0[:, i:i + 1].double() for i in range(3)]
    order = [(i, l) for l in range(L) for i in range(N)] if lf else [(i, l) for i in range(N) for l in range(L)]
    torch.manual_seed(0)
    np.random.seed(0)
    loss = 0
    for i, l in order:
        su, sv, th = onet._step(l, sat64[l], None, grd64[l], conf64[l], su, sv, th, None)
        loss = loss + (coef[:, i, l] * torch.cat([su, sv, th], 1)).sum()
    loss.backward()
    net = LM_S2GP(args).to(d)
    if args.train_damping:
        with torch.no_grad():
            net.damping.copy_(dparam.to(d))
    nh = lambda t: t.permute(0, 2, 3, 1).contiguous().to(d)
    feats = ([nh(s) for s in sat], [nh(g) for g in grd], [c[:, 0].contiguous().to(d) for c in conf])
    torch.manual_seed(0)
    np.random.seed(0)
    trace = net.lm_solve(*feats, grd_hw, None, lf, init_pose=p0, keep_normal_eq=True)
    d_sat, d_grd, d_conf, d_lam = net.lm_backward(*feats, grd_hw, trace, net.last_normal_eq, coef.float(), None, lf, init_pose=p0,
                                                  keep=net.last_keep)
    # hla_s2g_config.grd_grad_overwrite = 0 (zero-filled ground gradient buffers, every step adds) gives the same d_grd bit for bit
    # (the ground side has no atomics: a pixel's sum is first-visit value + the later steps' terms in the same order either way)
    acc = net.lm_backward(*feats, grd_hw, trace, net.last_normal_eq, coef.float(), None, lf, init_pose=p0, keep=net.last_keep,
                          overwrite=False)
    for l in range(L):
        assert torch.equal(acc[1][l][:, grd_hw[0] >> (4 - l):], d_grd[l][:, grd_hw[0] >> (4 - l):]), l
        assert not bool(d_grd[l][:, :grd_hw[0] >> (4 - l)].any())          # rows above h_l / 2: zero
    if getattr(args, 'deterministic_backward', 0):
        # hla_s2g_config.deterministic: d(loss)/d(sat map) accumulated in 64-bit fixed point -- the two calls above (and a third)
        # give the same bits whatever order the atomics ran in; d_lambda is summed in a fixed order in either mode
        again = net.lm_backward(*feats, grd_hw, trace, net.last_normal_eq, coef.float(), None, lf, init_pose=p0, keep=net.last_keep)
        for l in range(L):
            assert torch.equal(again[0][l], d_sat[l]) and torch.equal(acc[0][l], d_sat[l]), l
            assert bool(d_sat[l].any())
        assert torch.equal(again[3], d_lam)
    for l in range(L):
        for name, got, ref in (('sat', d_sat[l], sat64[l].grad), ('grd', d_grd[l], grd64[l].grad)):
            got = got.permute(0, 3, 1, 2).cpu().double().numpy()
            e = np.abs(got - ref.numpy()).max() / max(np.abs(ref.numpy()).max(), 1e-30)
            print(f'lm bwd {kw} lf{lf} level {l} d_{name}: rel err {e:.2e} (max |ref| {np.abs(ref.numpy()).max():.2e})')
            assert e < 2e-4, (kw, l, name, e)
        if args.using_weight:
            got = d_conf[l].cpu().double().numpy()
            ref = conf64[l].grad[:, 0].numpy()
            e = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)
            print(f'lm bwd level {l} d_conf: rel err {e:.2e}')
            assert e < 2e-4
    if args.train_damping:                      # d(loss)/d(damping parameter) through lambda = 10^(-6+11*sigmoid(d))
        dd = dparam.double()
        sg = torch.sigmoid(dd)
        lam = 10.0 ** (-6 + sg * 11.0)
        got = (d_lam.cpu().view(1, 3) * lam * np.log(10.0) * 11.0 * sg * (1 - sg)).numpy()
        ref = onet.damping.grad.numpy()
        e = np.abs(got - ref).max() / np.abs(ref).max()
        print(f'lm bwd d_damping: rel err {e:.2e}', got, ref)
        assert e < 1e-5


def test_e2e_ford_gauss_newton_vs_golden():
    """Optimizer='GN' (GN_update, models_ford.py:534-598: LM_update without damping and without renormalising the ground
    map; dispatched in the iteration-first loop, 775-781) against the REAL reference's traces."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_ford import LM_S2GP_Ford
    d = _dev()
    g = load_golden('e2e_ford_gn.npz')
    seed, B = int(g['seed']), int(g['B'])
    sat, grd, *_ = O.synth_images(seed + 100, B)
    R_FL = torch.tensor([[[0., 0., 1.], [1., 0., 0.], [0., 1., 0.]]]).repeat(B, 1, 1)
    T_FL = torch.tensor([[1.7, 0.3, -1.2]]).repeat(B, 1)
    for tag, kw in (('plain', {}), ('weight', dict(using_weight=1))):
        net = LM_S2GP_Ford(O.default_args(N_iters=5, Optimizer='GN', **kw))
        net.load_state_dict(O.synth_model_state(seed))
        net = net.to(d)
        torch.manual_seed(seed)
        with torch.no_grad():
            net(sat.to(d), grd.to(d), 112.64, R_FL.to(d), T_FL.to(d), mode='test')
        trace = _exec_order(net.last_trace, 0).cpu().numpy().astype(np.float64)
        _pose_gate(trace, g[f'trace64_{tag}'], g[f'trace32_{tag}'], f'ford GN {tag}')


def test_ford_gauss_newton_train_step_vs_oracle_autograd_small():
    """Optimizer='GN' under autograd (using_weight=1): loss and parameter gradients against the fp64 oracle."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_ford import LM_S2GP_Ford
    d = _dev()
    args = O.default_args(N_iters=2, Optimizer='GN', using_weight=1)
    B, grd_hw, sat_a = 2, (64, 256), 128
    sd = O.synth_model_state(4, bias_scale=0.02)
    sat, grd, gu, gv, gh = O.synth_images(9, B, grd_hw=grd_hw, sat_a=sat_a)
    R_FL = torch.tensor([[[0., 0., 1.], [1., 0., 0.], [0., 1., 0.]]]).repeat(B, 1, 1)
    T_FL = torch.tensor([[1.7, 0.3, -1.2]]).repeat(B, 1)
    gts = [gu[:, 0].double(), gv[:, 0].double(), gh[:, 0].double()]
    on = O.LM_S2GP_Ford(args, grd_hw=grd_hw)
    on.load_state_dict(sd)
    on = on.double()
    torch.manual_seed(0)
    ro = on(sat.double(), grd.double(), 28.16, R_FL.double(), T_FL.double(), *gts, mode='train')
    ro[0].backward()
    net = LM_S2GP_Ford(args)
    net.load_state_dict(sd)
    net = net.to(d).train()
    torch.manual_seed(0)
    r = net(sat.to(d), grd.to(d), 28.16, R_FL.to(d), T_FL.to(d), *[x.to(d) for x in gts], mode='train')
    r[0].backward()
    lerr = abs(float(r[0].detach()) - float(ro[0].detach())) / abs(float(ro[0].detach()))
    ref = dict(on.named_parameters())
    worst = 1.0
    for k, p in net.named_parameters():
        if ref[k].grad is None or float(ref[k].grad.norm()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        a, b = p.grad.double().cpu().flatten(), ref[k].grad.flatten()
        worst = min(worst, float(a @ b / (a.norm() * b.norm())) if a.numel() > 1 else 1.0 - abs(float(a - b)) / abs(float(b)))
        assert abs(float(a.norm() / b.norm()) - 1) < 2e-2, (k, float(a.norm()), float(b.norm()))
    print(f'ford GN train step: loss rel err {lerr:.2e}, worst gradient cosine {worst:.6f}')
    assert lerr < 1e-4 and worst > 0.999


@pytest.mark.parametrize('opt,lf', [('SGD', 0), ('ADAM', 0)])   # the reference has them in the iteration-first loop only
def test_ablation_optimisers_backward_vs_oracle_autograd(opt, lf):
    """Optimizer='SGD' / 'ADAM' (models_kitti.py:1056-1116) under autograd: hla_s2g_lm_solve_bwd's branch for the
    ablation updaters (ADAM: moment recurrence re-run from the saved sums, moment adjoints carried backwards) against
    torch autograd through the fp64 oracle's unrolled loop."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_S2GP
    d = _dev()
    args = O.default_args(N_iters=2, Optimizer=opt)
    B, grd_hw, sat_a = 2, (64, 256), 128
    onet, sat, grd, conf = _oracle_small(args, 7, B, grd_hw, sat_a)
    # these updaters work on the whole-map-normalised maps without renormalising: give the random maps unit-ish norm
    sat = [s / s[0].numel() ** 0.5 for s in sat]
    grd = [g / g[0].numel() ** 0.5 for g in grd]
    p0 = T(np.random.RandomState(5).uniform(-0.2, 0.2, size=(B, 3)).astype(np.float32))
    L, N = 3, args.N_iters
    coef = T(np.random.RandomState(6).standard_normal((B, N, L, 3)))
    sat64 = [s.double().requires_grad_(True) for s in sat]
    grd64 = [g.double().requires_grad_(True) for g in grd]
    su, sv, th = [p0[:, i:i + 1].double() for i in range(3)]
    order = [(i, l) for l in range(L) for i in range(N)] if lf else [(i, l) for i in range(N) for l in range(L)]
    onet._adam_t = 0
    loss = 0
    for i, l in order:
        su, sv, th = onet._step(l, sat64[l], None, grd64[l], conf[l].double(), su, sv, th, None)
        loss = loss + (coef[:, i, l] * torch.cat([su, sv, th], 1)).sum()
    loss.backward()
    net = LM_S2GP(args).to(d)
    nh = lambda t: t.permute(0, 2, 3, 1).contiguous().to(d)
    feats = ([nh(s) for s in sat], [nh(g) for g in grd], [c[:, 0].contiguous().to(d) for c in conf])
    trace = net.lm_solve(*feats, grd_hw, None, lf, init_pose=p0, keep_normal_eq=True)
    ref_trace = torch.stack([su, sv, th], -1)[:, 0].detach().numpy()
    slot = (N - 1, L - 1)
    # (ADAM's first steps are sign(g)-like: rounding differences in the fp32 pose grow along the loop)
    assert np.abs(trace[:, slot[0], slot[1]].cpu().double().numpy() - ref_trace).max() < (1e-5 if opt == 'SGD' else 1e-4)
    d_sat, d_grd, _, _ = net.lm_backward(*feats, grd_hw, trace, net.last_normal_eq, coef.float(), None, lf, init_pose=p0)
    for l in range(L):
        for name, got, ref in (('sat', d_sat[l], sat64[l].grad), ('grd', d_grd[l], grd64[l].grad)):
            got = got.permute(0, 3, 1, 2).cpu().double().numpy()
            e = np.abs(got - ref.numpy()).max() / max(np.abs(ref.numpy()).max(), 1e-30)
            print(f'{opt} lf{lf} bwd level {l} d_{name}: rel err {e:.2e} (max |ref| {np.abs(ref.numpy()).max():.2e})')
            # ADAM: the oracle's own fp32-vs-fp64 gradients differ by 2.6e-2 here (pose 2e-4): sign(g)-like first steps
            assert e < (2e-4 if opt == 'SGD' else 5e-2), (opt, l, name, e)


@pytest.mark.parametrize('opt', ['SGD', 'ADAM'])
def test_ablation_optimisers_train_step_vs_oracle_autograd_small(opt):
    """mode='train' + backward with the ablation updaters on a reduced shape: loss and parameter gradients."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_S2GP
    d = _dev()
    args = O.default_args(N_iters=2, Optimizer=opt, using_weight=1)      # using_weight is ignored by SGD_update / ADAM_update
    B, grd_hw, sat_a = 2, (64, 256), 128
    sd = O.synth_model_state(4, bias_scale=0.02)
    sat, grd, gu, gv, gh = O.synth_images(9, B, grd_hw=grd_hw, sat_a=sat_a)
    on = O.LM_S2GP(args, grd_hw=grd_hw)
    on.load_state_dict(sd)
    on = on.double()
    ro = on(sat.double(), grd.double(), gu.double(), gv.double(), gh.double(), mode='train')
    ro[0].backward()
    net = LM_S2GP(args)
    net.load_state_dict(sd)
    net = net.to(d).train()
    r = net(sat.to(d), grd.to(d), gu.to(d), gv.to(d), gh.to(d), mode='train')
    r[0].backward()
    lerr = abs(float(r[0].detach()) - float(ro[0].detach())) / abs(float(ro[0].detach()))
    print(f'{opt} train step: loss rel err {lerr:.2e}')
    assert lerr < (1e-4 if opt == 'SGD' else 2e-3)        # ADAM: chaotic, see test_e2e_ablation_optimisers_vs_golden
    ref = dict(on.named_parameters())
    worst = 1.0
    for k, p in net.named_parameters():
        if ref[k].grad is None or float(ref[k].grad.norm()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        a, b = p.grad.double().cpu().flatten(), ref[k].grad.flatten()
        cos = float(a @ b / (a.norm() * b.norm()))
        worst = min(worst, cos)
        assert abs(float(a.norm() / b.norm()) - 1) < (2e-2 if opt == 'SGD' else 0.5), (k, float(a.norm()), float(b.norm()))
    print(f'{opt} train step: worst gradient cosine {worst:.6f}')
    assert worst > (0.999 if opt == 'SGD' else 0.9)


# bf16: the one-hop gradient (conv_dec2.3) agrees to 0.5 %; deeper layers differ by up to ~17 % in relative L2 because
# bf16 rounding of the FORWARD flips max-pool argmax / ReLU signs of near-ties (a flipped argmax moves a gradient
# element to a neighbouring pixel).  That is a property of bf16 training, not of the backward kernels, whose logic
# the fp32 run pins to 2e-6; the bf16 bound below only guards against gross breakage.
@pytest.mark.parametrize('precision,tol', [('fp32', 2e-4), ('bf16', 0.3), ('fp16', 0.1)])
def test_vgg_backward_small_vs_oracle_autograd(precision, tol):
    """hla_vgg_backward (L2-norm bwd, dgrad convs with fused ReLU mask / fan-in / unpool / upsample-sum, MFMA wgrad,
    bias grads) against torch autograd through the fp64 oracle VGGUnet."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.VGG import VGGUnet, vgg_forward_nhwc, vgg_backward_nhwc
    d = _dev()
    rs = np.random.RandomState(31)
    sd = O.synth_vgg_state(rs, bias_scale=0.05)
    x = T(rs.random_sample((2, 3, 32, 64)).astype(np.float32))
    onet = O.VGGUnet(3)
    onet.load_state_dict(sd)
    onet = onet.double()
    feats64, _ = onet(x.double())
    ups = [T(rs.standard_normal(tuple(f.shape))) for f in feats64]          # d(loss)/d(normalised map), NCHW
    loss = sum((u * f).sum() for u, f in zip(ups, feats64))
    loss.backward()
    ref = {k: p.grad for k, p in onet.named_parameters()}
    net = VGGUnet(3, precision=precision)
    net.load_state_dict(sd)
    net = net.to(d)
    feats, _, inv, ctx = vgg_forward_nhwc(net, x.to(d), want_conf=False, defer_norm=True, save_for_backward=True)
    grads = vgg_backward_nhwc(net, ctx, [u.permute(0, 2, 3, 1).contiguous().float().to(d) for u in ups])
    worst, errs = 0.0, {}
    for k, g in grads.items():
        r = ref[k].numpy()
        e = np.abs(g.cpu().double().numpy() - r).max() / max(np.abs(r).max(), 1e-30)
        worst = max(worst, e)
        rl2 = np.linalg.norm(g.cpu().double().numpy() - r) / max(np.linalg.norm(r), 1e-30)
        print(f'vgg bwd {precision} {k:24s} rel err max {e:.2e} l2 {rl2:.2e} (max |ref| {np.abs(r).max():.2e})')
        errs[k] = rl2 if precision != 'fp32' else e
    assert max(errs.values()) < tol, (precision, errs)
    if precision != 'fp32':
        assert errs['conv_dec2.3.weight'] < 2e-2
    for k in ('conv_dec3.1.weight', 'conf0.1.weight'):
        assert ref[k] is None and k not in grads        # the reference leaves these without a gradient too (SURVEY B-8)
    print(f'vgg bwd {precision}: worst rel err {worst:.2e}')


def test_vgg_backward_confidence_heads_vs_oracle_autograd():
    """Gradients arriving at the confidence maps (using_weight=1, models_kitti.py:994-996) flow through
    sigmoid(-sigmoid(conv(relu(x)))) (VGG.py:62-76,160-162) into the head weights and into the trunk."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.VGG import VGGUnet, vgg_forward_nhwc, vgg_backward_nhwc
    d = _dev()
    rs = np.random.RandomState(37)
    sd = O.synth_vgg_state(rs, bias_scale=0.05)
    x = T(rs.random_sample((2, 3, 32, 64)).astype(np.float32))
    onet = O.VGGUnet(3)
    onet.load_state_dict(sd)
    onet = onet.double()
    feats64, confs64 = onet(x.double())
    ups = [T(rs.standard_normal(tuple(f.shape))) for f in feats64]
    cups = [T(rs.standard_normal(tuple(c.shape))) * 30.0 for c in confs64[:3]]   # comparable in size to the feature term
    loss = sum((u * f).sum() for u, f in zip(ups, feats64)) + sum((u * c).sum() for u, c in zip(cups, confs64))
    loss.backward()
    ref = {k: p.grad for k, p in onet.named_parameters()}
    net = VGGUnet(3, precision='fp32')
    net.load_state_dict(sd)
    net = net.to(d)
    feats, confs, inv, ctx = vgg_forward_nhwc(net, x.to(d), want_conf=True, defer_norm=True, save_for_backward=True)
    grads = vgg_backward_nhwc(net, ctx, [u.permute(0, 2, 3, 1).contiguous().float().to(d) for u in ups], confs,
                              [u[:, 0].contiguous().float().to(d) for u in cups])
    for k in ('conf0.1.weight', 'conf1.1.weight', 'conf2.1.weight'):
        assert k in grads
    for k, g in grads.items():
        r = ref[k].numpy()
        e = np.abs(g.cpu().double().numpy() - r).max() / max(np.abs(r).max(), 1e-30)
        print(f'vgg bwd conf {k:24s} rel err max {e:.2e} (max |ref| {np.abs(r).max():.2e})')
        assert e < 2e-4, (k, e)
    # the heads must matter in this test: without d_conf the trunk gradient differs visibly
    g0 = vgg_backward_nhwc(net, ctx, [u.permute(0, 2, 3, 1).contiguous().float().to(d) for u in ups])
    assert (g0['conv14.weight'] - grads['conv14.weight']).abs().max() > 1e-3 * grads['conv14.weight'].abs().max()


def test_train_step_using_weight_vs_reference_golden():
    """using_weight=1 + train_damping=1, mode='train': the confidence maps weight the LM sums, so loss.backward() also
    reaches GrdFeatureNet.conf{0,1,2} and `damping`.  Samples recorded from the REAL reference's autograd (fp64)."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_S2GP
    from make_idx import sample_idx
    g = load_golden('train_kitti_w.npz')
    seed, B = int(g['seed']), int(g['B'])
    d = _dev()
    net = LM_S2GP(O.default_args(using_weight=1, train_damping=1))
    net.load_state_dict(O.synth_model_state(seed))
    net = net.to(d).train()
    sat, grd, gu, gv, gh = O.synth_images(seed + 100, B)
    torch.manual_seed(seed)
    res = net(sat.to(d), grd.to(d), gu.to(d), gv.to(d), gh.to(d), mode='train')
    assert abs(float(res[0]) - g['tuple64'][0][0]) < 1e-3 * abs(g['tuple64'][0][0])
    res[0].backward()
    named = dict(net.named_parameters())
    nograd = set(str(k) for k in g['nograd_64'])
    for k, p in named.items():
        assert (p.grad is None) == (k in nograd), k
    for k in [k[len('grad64_'):] for k in g.files if k.startswith('grad64_')]:
        ref = g['grad64_' + k]
        gr = named[k].grad.double().reshape(-1).cpu()
        idx = sample_idx(gr.numel(), 77)
        got = np.concatenate([[gr.abs().sum().item(), (gr * gr).sum().item()], gr[idx].numpy()])
        gap = np.abs(g['grad32_' + k][2:] - ref[2:]).max()
        scale = np.abs(ref[2:]).max()
        e = np.abs(got[2:] - ref[2:]).max()
        print(f'train(w) grad {k:36s} max err {e:.2e} (ref fp32 gap {gap:.2e}, scale {scale:.2e}); l1 {got[0]:.4e} vs {ref[0]:.4e}')
        rel_tol = 2e-4 if ('conv_dec2' in k or 'conf2' in k or k == 'damping') else 5e-3   # see the flip-noise note above
        assert e <= max(rel_tol * scale, 3 * gap), (k, e, gap, scale)


def test_reference_call_pattern_harness_runs():
    """tools/train_harness.py = the reference's train/test call pattern (train_kitti.py) on synthetic batches."""
    import importlib.util, os
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'train_harness.py')
    spec = importlib.util.spec_from_file_location('train_harness', p)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    log = m.main(['--batch_size', '2', '--iters_per_epoch', '2', '--N_iters', '2', '--grd_h', '64', '--grd_w', '256',
                  '--sat_a', '128', '--precision', 'fp32', '--train_damping', '1'])
    assert len(log) == 2 and all(np.isfinite(log))


def test_checkpoint_round_trip_resume_and_result_files(tmp_path):
    """the harness writes what the reference's driver writes (model_<epoch>.pth after every epoch, Test1_results.mat
    / .txt, Model_best.pth); a checkpoint loaded in a NEW process reproduces the saved model's pose trace bit for bit;
    --resume continues from model_<N-1>.pth; --test 1 evaluates model_1.pth."""
    import importlib.util, os, subprocess, sys
    import scipy.io as scio
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('train_harness', os.path.join(root, 'tools', 'train_harness.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    sp = str(tmp_path / 'ckpt')
    common = ['--batch_size', '2', '--iters_per_epoch', '2', '--N_iters', '2', '--grd_h', '64', '--grd_w', '256', '--sat_a', '128',
              '--precision', 'fp32', '--save_path', sp]
    log = m.main(common + ['--epochs', '2'])
    assert len(log) == 4 and all(np.isfinite(log))
    for f in ('model_0.pth', 'model_1.pth', 'Test1_results.mat', 'Test1_results.txt'):
        assert os.path.exists(os.path.join(sp, f)), f
    mat = scio.loadmat(os.path.join(sp, 'Test1_results.mat'))
    assert mat['pred_shifts'].shape == (4, 2) and mat['gt_headings'].shape == (4, 1)       # 2 test batches of 2
    txt = open(os.path.join(sp, 'Test1_results.txt')).read()
    assert txt.count('EPOCH:') == 2 and 'lat within 5 & angle within 5 (pred, init):' in txt
    # the checkpoint in this process ...
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_S2GP
    d = _dev()
    sd = torch.load(os.path.join(sp, 'model_1.pth'), map_location='cpu')
    assert len(sd) == 49
    net = LM_S2GP(O.default_args(N_iters=2))
    net.load_state_dict(sd)                                  # strict
    net = net.to(d)
    sat, grd, *_ = O.synth_images(5, 2, grd_hw=(64, 256), sat_a=128)
    torch.manual_seed(0)
    with torch.no_grad():
        net(sat.to(d), grd.to(d), mode='test')
    here = net.last_trace.cpu().numpy()
    # ... and in a new process
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); from oracle import ref_cpu as O; "
            "from highlyaccurate_amd.models_kitti import LM_S2GP; net = LM_S2GP(O.default_args(N_iters=2)); "
            "net.load_state_dict(torch.load(%r, map_location='cpu')); net = net.to('cuda:0'); "
            "sat, grd, *_ = O.synth_images(5, 2, grd_hw=(64, 256), sat_a=128); torch.manual_seed(0); "
            "torch.no_grad().__enter__(); net(sat.cuda(), grd.cuda(), mode='test'); np.save(%r, net.last_trace.cpu().numpy())"
            % (root, os.path.join(sp, 'model_1.pth'), str(tmp_path / 'there.npy')))
    subprocess.run([sys.executable, '-c', code], check=True, timeout=600)
    np.testing.assert_array_equal(np.load(str(tmp_path / 'there.npy')), here)
    # --resume 2: loads model_1.pth, runs epoch 2 only
    log = m.main(common + ['--epochs', '3', '--resume', '2'])
    assert len(log) == 2 and os.path.exists(os.path.join(sp, 'model_2.pth'))
    assert open(os.path.join(sp, 'Test1_results.txt')).read().count('EPOCH:') == 3
    # --test 1: evaluates model_1.pth, appends one more block, trains nothing
    assert m.main(common + ['--test', '1']) == []
    assert open(os.path.join(sp, 'Test1_results.txt')).read().count('EPOCH:') == 4


@pytest.mark.parametrize('level', [3, 4])
def test_standalone_vggunet_is_differentiable_like_the_reference(level):
    """VGGUnet used on its own (VGG.py:121-203 is an ordinary autograd module): a loss on its returned maps and confidence maps
    back-propagates to every parameter that reaches them; gradients against the fp64 oracle's autograd."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.VGG import VGGUnet
    d = _dev()
    rs = np.random.RandomState(41)
    sd = O.synth_vgg_state(rs, bias_scale=0.05)
    x = T(rs.random_sample((2, 3, 32, 64)).astype(np.float32))
    onet = O.VGGUnet(level)
    onet.load_state_dict(sd)
    onet = onet.double()
    f64, c64 = onet(x.double())
    uf = [T(rs.standard_normal(tuple(f.shape))) for f in f64]
    uc = [T(rs.standard_normal(tuple(c.shape))) for c in c64]
    (sum((u * f).sum() for u, f in zip(uf, f64)) + sum((u * c).sum() for u, c in zip(uc, c64))).backward()
    ref = {k: p.grad for k, p in onet.named_parameters() if p.grad is not None}
    net = VGGUnet(level)
    net.load_state_dict(sd)
    net = net.to(d)
    feats, confs = net(x.to(d))
    assert all(f.grad_fn is not None for f in feats) and all(c.grad_fn is not None for c in confs)
    (sum((u.float().to(d) * f).sum() for u, f in zip(uf, feats)) + sum((u.float().to(d) * c).sum() for u, c in zip(uc, confs))).backward()
    got = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    assert set(got) == set(ref), set(got) ^ set(ref)
    worst = 0.0
    for k in ref:
        r = ref[k].numpy()
        e = np.abs(got[k].cpu().double().numpy() - r).max() / max(np.abs(r).max(), 1e-30)
        worst = max(worst, e)
        assert e < 2e-4, (k, e)
    print(f'standalone VGGUnet level {level}: {len(ref)} gradients, worst rel err {worst:.2e}')
    with torch.no_grad():                                    # and without autograd it is the plain forward, same values
        f2, c2 = net(x.to(d))
    assert all(torch.equal(a, b) for a, b in zip(f2, feats)) and all(torch.equal(a, b) for a, b in zip(c2, confs))


def test_input_modified_before_backward_raises_like_autograd():
    """The backward reads the input image again (conv0's weight gradient) and keeps no copy of it -- the image may be a row window
    of the caller's own storage.  Like a tensor autograd saved, an in-place change between forward and backward must raise
    instead of silently giving the gradient of another image."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.VGG import VGGUnet
    d = _dev()
    rs = np.random.RandomState(43)
    net = VGGUnet(3)
    net.load_state_dict(O.synth_vgg_state(rs, bias_scale=0.05))
    net = net.to(d)
    img = T(rs.random_sample((2, 3, 48, 64)).astype(np.float32)).to(d)
    feats, _ = net(img[:, :, 16:, :])                         # a row window: passed to the kernels without a copy
    img.mul_(0.5)
    with pytest.raises(RuntimeError, match='modified by an inplace operation'):
        sum(f.sum() for f in feats).backward()
    feats, _ = net(img[:, :, 16:, :])                         # untouched: fine
    sum(f.sum() for f in feats).backward()
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)


@pytest.mark.parametrize('kw', [dict(), dict(using_weight=1, train_damping=1)])
def test_two_rank_real_model_gradients_match_full_batch(kw, tmp_path):
    """the batch shards over ranks and the only exchange is the gradient all-reduce.  Two processes (gloo; they
    share this box's one GPU) run LM_S2GP train steps on the two halves of a B = 4 batch with `net.grad_sync` installed, i.e.
    through the model's own backward: bucket per branch, flat gradient buffers reduced in place, the satellite bucket in flight
    during the ground branch's backward.  Every parameter gradient on BOTH ranks must equal the single-process B = 4 gradient
    (to the order of the weight-gradient partial sums), and the parameters that get no gradient stay grad-less on both."""
    import json, os, socket, subprocess, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import dp_worker as W
    from highlyaccurate_amd.models_kitti import LM_S2GP
    d = _dev()
    args, sd, batch = W.build_case(kw)
    net = LM_S2GP(args)
    net.load_state_dict(sd)
    net = net.to(d).train()
    full, loss_full = W.run(net, batch, d)
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, W.__file__, str(tmp_path), json.dumps(kw)], env=env))
    for p in procs:
        assert p.wait(timeout=900) == 0
    ranks = [torch.load(str(tmp_path / f'rank{r}.pt')) for r in range(2)]
    n_live = sum(g is not None for g in full.values())
    assert n_live == (40 if kw else 36), n_live                      # SURVEY B-8: 36 live tensors by default, +3 heads +damping
    assert abs(0.5 * (ranks[0]['loss'] + ranks[1]['loss']) - loss_full) < 1e-4 * max(1.0, abs(loss_full))
    worst = 0.0
    for r in range(2):
        assert ranks[r]['collectives'] >= 2 and ranks[r]['bytes'] >= 19_000_000     # two flat buckets (+ staged extras)
        for k, g in full.items():
            gr = ranks[r]['grads'][k]
            if g is None:
                assert gr is None, (r, k)                             # grad-less on every rank, never a zero tensor
                continue
            e = float((gr - g).abs().max() / g.abs().max().clamp_min(1e-30))
            worst = max(worst, e)
            assert e < 5e-4, (r, k, e)
        for k in full:
            if full[k] is not None:
                assert torch.equal(ranks[0]['grads'][k], ranks[1]['grads'][k]), k    # the all-reduce leaves identical replicas
    print(f'2-rank vs full-batch gradients {kw}: worst rel err {worst:.2e} over {n_live} tensors')


@pytest.mark.parametrize('level', [3, 4])
def test_ford_train_step_vs_oracle_autograd_small(level):
    """Ford model, mode='train' under autograd on a reduced shape: loss and a few gradients vs the fp64 oracle."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_ford import LM_S2GP_Ford
    d = _dev()
    args = O.default_args(N_iters=2, level=level)
    B, grd_hw, sat_a = 2, (64, 256), 128
    sd = O.synth_model_state(4, bias_scale=0.02)
    sat, grd, gu, gv, gh = O.synth_images(9, B, grd_hw=grd_hw, sat_a=sat_a)
    R_FL = torch.tensor([[[0., 0., 1.], [1., 0., 0.], [0., 1., 0.]]]).repeat(B, 1, 1)
    T_FL = torch.tensor([[1.7, 0.3, -1.2]]).repeat(B, 1)
    gts = [gu[:, 0].double(), gv[:, 0].double(), gh[:, 0].double()]            # Ford passes [B] float64 (Ford_dataset.py:211)
    on = O.LM_S2GP_Ford(args, grd_hw=grd_hw)
    on.load_state_dict(sd)
    on = on.double()
    torch.manual_seed(0)
    ro = on(sat.double(), grd.double(), 28.16, R_FL.double(), T_FL.double(), *gts, mode='train')
    ro[0].backward()
    net = LM_S2GP_Ford(args)
    net.load_state_dict(sd)
    net = net.to(d).train()
    torch.manual_seed(0)
    r = net(sat.to(d), grd.to(d), 28.16, R_FL.to(d), T_FL.to(d), *[g.to(d) for g in gts], mode='train')
    r[0].backward()
    assert abs(float(r[0]) - float(ro[0])) < 1e-4 * abs(float(ro[0]))
    ref = dict(on.named_parameters())
    for k in ('SatFeatureNet.conv_dec2.3.weight', 'GrdFeatureNet.conv_dec2.3.weight', 'SatFeatureNet.conv0.weight', 'GrdFeatureNet.conv5.weight'):
        a, b = dict(net.named_parameters())[k].grad.double().cpu().numpy(), ref[k].grad.numpy()
        e = np.linalg.norm(a - b) / np.linalg.norm(b)
        print(f'ford train grad {k}: rel-l2 {e:.2e}')
        assert e < (2e-4 if 'dec2' in k else 2e-2)


def test_dead_ground_rows_elimination_is_exact():
    """mode='test' extracts ground features only for image rows >= dead_ground_rows(H) (88 at H=256): every row of the
    three maps that the LM loop reads (h_l/2..) must be BIT-identical to the full-image run, and the pose trace equal up
    to the rounding of the (mathematically cancelling) per-sample L2_norm scale."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd._s2gp import dead_ground_rows
    from highlyaccurate_amd.VGG import VGGUnet, vgg_forward_nhwc
    assert dead_ground_rows(256) == 88 and dead_ground_rows(512) == 216 and dead_ground_rows(64) == 0
    d = _dev()
    for precision in ('fp32', 'bf16'):
        net = VGGUnet(3, precision=precision)
        net.load_state_dict(O.synth_vgg_state(np.random.RandomState(5), bias_scale=0.05))
        net = net.to(d)
        x = torch.rand(2, 3, 256, 1024, device=d)
        full, cfull, _ = vgg_forward_nhwc(net, x, want_conf=True, defer_norm=True)
        crop, ccrop, _ = vgg_forward_nhwc(net, x[:, :, 88:], want_conf=True, defer_norm=True)       # a row window: no copy (x_plane)
        dense, cdense, _ = vgg_forward_nhwc(net, x[:, :, 88:].contiguous(), want_conf=True, defer_norm=True)
        assert all(torch.equal(a, b) for a, b in zip(crop + ccrop, dense + cdense)), precision
        for l in range(3):
            h = full[l].shape[1]
            skip = 88 >> (3 - l)
            assert crop[l].shape[1] == h - skip
            assert torch.equal(full[l][:, h // 2:], crop[l][:, h // 2 - skip:]), (precision, l)
            assert torch.equal(cfull[l][:, h // 2:], ccrop[l][:, h // 2 - skip:]), (precision, l)
        # per-layer trimming (first_row8 = f: only rows f / 2f / 4f.. of the three maps are promised to be read): those rows
        # are still bit-identical, whatever lies above them
        f8 = 16 - 11
        for wc in (False, True):
            trim, ctrim, _ = vgg_forward_nhwc(net, x[:, :, 88:], want_conf=wc, defer_norm=True, first_row8=f8)
            for l in range(3):
                r = f8 << l
                assert torch.equal(trim[l][:, r:], crop[l][:, r:]), (precision, l, wc)
                assert not wc or torch.equal(ctrim[l][:, r:], ccrop[l][:, r:]), (precision, l)
                assert r == full[l].shape[1] // 2 - (88 >> (3 - l))
    g = load_golden('e2e_kitti.npz')
    seed, B = int(g['seeds'][0]), int(g['B'])
    for kw in (dict(), dict(using_weight=1)):
        net0, _ = _run_kitti(seed, B, ground_crop=0, **kw)
        net1, _ = _run_kitti(seed, B, **kw)              # args.ground_crop defaults to 1
        dev = (net0.last_trace - net1.last_trace).abs().max().item()
        print(f'dead-row elimination {kw}: max pose deviation {dev:.2e}')
        assert dev < 2e-6


@pytest.mark.parametrize('precision,shape', [('fp32', (3, 96, 160)), ('bf16', (3, 96, 160)), ('bf16', (1, 1024, 64))])
def test_backward_dynamic_trimming_is_exact(precision, shape):
    """hla_vgg_backward finds, per map row, the column interval of the non-zero incoming gradient and skips every tile of every
    dgrad / wgrad launch that it proves zero (vgg_backward.hip, bwd_fan_kernel).  Against the dense walk (HLA_VGG_BWD_DENSE) the
    gradients may differ only by the summation order of the weight-gradient partials -- for boxes at odd offsets, one per level,
    a single texel, per-sample different footprints, a sparse wedge, and no gradient at all."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.VGG import VGGUnet, vgg_forward_nhwc, vgg_backward_nhwc
    d = _dev()
    rs = np.random.RandomState(77)
    sd = O.synth_vgg_state(rs, bias_scale=0.05)
    B, H, W = shape         # (3, 96, 160): maps 12x20, 24x40, 48x80, several 32-px column tiles; (1, 1024, 64): the row limit
    x = T(rs.random_sample((B, 3, H, W)).astype(np.float32)).to(d)
    net = VGGUnet(3, precision=precision)
    net.load_state_dict(sd)
    net = net.to(d)
    feats, _, inv, ctx = vgg_forward_nhwc(net, x, want_conf=False, defer_norm=True, save_for_backward=True)
    shapes = [(B, H // 8, W // 8, 256), (B, H // 4, W // 4, 128), (B, H // 2, W // 2, 64)]

    def boxes(spec):
        out = []
        for (b, h, w, c), per_level in zip(shapes, spec):
            g = torch.zeros(b, h, w, c)
            for bi, (y0, y1, x0, x1) in per_level:
                g[bi, y0:y1, x0:x1] = T(rs.standard_normal((y1 - y0, x1 - x0, c)).astype(np.float32))
            out.append(g.to(d))
        return out

    cases = {
        'right half': [[(0, (1, 11, 11, 19)), (1, (2, 9, 12, 20))], [(0, (3, 22, 23, 39)), (2, (5, 20, 25, 40))],
                       [(0, (7, 45, 47, 79)), (1, (9, 40, 50, 80))]],
        'coarse level only': [[(1, (5, 6, 3, 4))], [], []],
        'fine level only, one texel': [[], [], [(2, (17, 18, 33, 34))]],
        'disjoint boxes per level': [[(0, (0, 2, 0, 3))], [(1, (20, 24, 36, 40))], [(2, (21, 27, 1, 9))]],
        'nothing': [[], [], []],
        'everything': [[(0, (0, 12, 0, 20))], [(1, (0, 24, 0, 40))], [(2, (0, 48, 0, 80))]],
    }
    def fan():          # the shape the LM loop produces: a wedge opening towards +x from the map centre, per sample slightly rotated
        out = []
        for (b, h, w, c) in shapes:
            g = torch.zeros(b, h, w, c)
            ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
            for bi in range(b):
                m = ((ys - h / 2 - 0.1 * bi * (xs - w / 2)).abs() < 0.45 * (xs - w / 2 - 1)) & (((xs + 3 * ys).long() % 4) == 0)
                g[bi][m] = T(rs.standard_normal((int(m.sum()), c)).astype(np.float32))
            out.append(g.to(d))
        return out

    if H != 96:
        cases = {'nothing': [[], [], []]}
    # live-tile fractions as measured, + 0.03: they pin how TIGHT the propagated footprint is (a regression that marks
    # everything live would still be exact)
    LIMITS = {('right half', 96): 0.97, ('coarse level only', 96): 0.21, ('fine level only, one texel', 96): 0.47,
              ('disjoint boxes per level', 96): 0.62, ('fan', 96): 0.90, ('fan', 1024): 0.11}
    for tag, spec in list(cases.items()) + [('fan', None)]:
        dfe = fan() if spec is None else boxes(spec)
        st_d, st_t = {}, {}
        g_dense = vgg_backward_nhwc(net, ctx, dfe, scale_invariant=True, dense=True, stats=st_d)
        g_trim = vgg_backward_nhwc(net, ctx, dfe, scale_invariant=True, stats=st_t)
        # the trimming must actually have happened (and by how much): hla_vgg_backward_live_tiles
        assert st_d == {'live_tiles': 0, 'total_tiles': 0} and st_t['total_tiles'] > 0, (tag, st_d, st_t)
        frac = st_t['live_tiles'] / st_t['total_tiles']
        limit = {'nothing': 0.0, 'everything': 1.0}.get(tag, LIMITS.get((tag, H), 1.0))
        assert frac <= limit and (tag != 'everything' or frac == 1.0), (tag, frac)
        worst, wk = 0.0, ''
        for k in g_dense:
            a, b = g_dense[k].double(), g_trim[k].double()
            assert torch.isfinite(b).all(), (tag, k)
            if tag == 'nothing':
                assert float(b.abs().max()) == 0.0 and float(a.abs().max()) == 0.0, (tag, k)
                continue
            e = float((a - b).norm() / max(float(a.norm()), 1e-30))
            if e > worst:
                worst, wk = e, k
        print(f'dynamic trimming {precision} {H}x{W} [{tag}]: live tiles {frac:.2f}, worst gradient rel-l2 deviation {worst:.2e} ({wk})')
        assert worst < 2e-5, (tag, wk, worst)


@pytest.mark.parametrize('kw', [dict(), dict(using_weight=1, train_damping=1), dict(train_ground_crop=1),
                                dict(precision='fp16x3')])      # (fp16x3: the wave-specialised weight-gradient kernels on ODD first rows)
def test_backward_row_trimming_is_exact(kw):
    """The ground branch's gradient lives in rows h_l/2.. of its three maps; hla_vgg_backward(first_row8) skips, layer by
    layer, the rows above the support of every activation gradient (exact zeros).  Gradients must equal the untrimmed
    backward up to the summation order of the weight-gradient partials."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_S2GP
    d = _dev()
    seed, B = 2, 2
    sat, grd, gu, gv, gh = O.synth_images(seed + 100, B)
    res = {}
    for trim in ('0', '1'):
        net = LM_S2GP(O.default_args(bwd_trim=int(trim), **kw))     # (0: no row trimming and no data-dependent trimming either)
        sd = O.synth_model_state(seed, bias_scale=0.02)
        if kw.get('train_damping'):
            sd['damping'] = torch.tensor([[0.1, -0.2, 0.15]])
        net.load_state_dict(sd)
        net = net.to(d).train()
        torch.manual_seed(0)
        r = net(sat.to(d), grd.to(d), gu.to(d), gv.to(d), gh.to(d), mode='train')
        r[0].backward()
        res[trim] = {k: p.grad.double().cpu() for k, p in net.named_parameters() if p.grad is not None}
    assert set(res['0']) == set(res['1'])
    worst, wk = 0.0, ''
    for k in res['0']:
        e = float((res['0'][k] - res['1'][k]).norm() / max(float(res['0'][k].norm()), 1e-30))
        if e > worst:
            worst, wk = e, k
    print(f'backward row trimming {kw}: worst gradient rel-l2 deviation {worst:.2e} ({wk})')
    assert worst < 2e-5


@pytest.mark.parametrize('kw', [dict(), dict(using_weight=1, train_damping=1)])
def test_training_ground_crop_is_exact(kw):
    """args.train_ground_crop=1 (an extension): forward(train) + backward on the ground-image rows that can reach the loss.
    Loss and every parameter gradient must equal the full-image run up to the rounding of the (cancelling) L2_norm scale;
    the returned confidence maps keep their shape and agree from row h_l/2 on."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_S2GP
    d = _dev()
    seed, B = 1, 2
    sat, grd, gu, gv, gh = O.synth_images(seed + 100, B)
    res = {}
    for crop in (0, 1):
        args = O.default_args(train_ground_crop=crop, **kw)
        net = LM_S2GP(args)
        sd = O.synth_model_state(seed, bias_scale=0.02)
        if kw.get('train_damping'):
            sd['damping'] = torch.tensor([[0.1, -0.2, 0.15]])
        net.load_state_dict(sd)
        net = net.to(d).train()
        torch.manual_seed(0)
        r = net(sat.to(d), grd.to(d), gu.to(d), gv.to(d), gh.to(d), mode='train')
        r[0].backward()
        res[crop] = (float(r[0].detach()), {k: p.grad.double().cpu() for k, p in net.named_parameters() if p.grad is not None},
                     [c.detach().cpu() for c in r[13]])
    (l0, g0, c0), (l1, g1, c1) = res[0], res[1]
    assert abs(l0 - l1) < 1e-6 * abs(l0)
    assert set(g0) == set(g1)
    worst = 0.0
    for k in g0:
        e = float((g0[k] - g1[k]).norm() / max(float(g0[k].norm()), 1e-30))
        worst = max(worst, e)
    print(f'train_ground_crop {kw}: loss {l0:.6f} / {l1:.6f}, worst gradient rel-l2 deviation {worst:.2e}')
    assert worst < 2e-5
    for l in range(3):
        assert c0[l].shape == c1[l].shape
        h = c0[l].shape[-2]
        assert torch.equal(c0[l][..., h // 2:, :], c1[l][..., h // 2:, :])


@pytest.mark.parametrize('B,grd_hw,sat_a', [(3, (72, 264), 136), (1, (64, 256), 128), (5, (88, 200), 104)])
def test_e2e_ragged_shapes_vs_oracle(B, grd_hw, sat_a):
    """Sizes that are multiples of 8 but not of the 8x32 conv tile / the LM pixel tile, odd batch sizes, B = 1:
    whole forward (fp32 mode) against the fp64 oracle."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_S2GP
    d = _dev()
    args = O.default_args(N_iters=2)
    sd = O.synth_model_state(3, bias_scale=0.02)
    sat, grd, *_ = O.synth_images(11, B, grd_hw=grd_hw, sat_a=sat_a)
    onet = O.LM_S2GP(args, grd_hw=grd_hw)
    onet.load_state_dict(sd)
    onet = onet.double()
    torch.manual_seed(0)
    with torch.no_grad():
        ref = torch.stack(onet(sat.double(), grd.double(), mode='test'), -1).numpy()
    net = LM_S2GP(args)
    net.load_state_dict(sd)
    net = net.to(d)
    torch.manual_seed(0)
    with torch.no_grad():
        res = torch.stack(net(sat.to(d), grd.to(d), mode='test'), -1).cpu().numpy()
    err = np.abs(res - ref).max()
    print(f'ragged B={B} grd {grd_hw} sat {sat_a}: max pose err {err:.2e} (range {np.abs(ref).max():.2e})')
    assert np.isfinite(res).all() and err < 2e-4


def test_error_behaviour_matches_the_reference():
    """(1) use_hessian=1 with a pose component that has no Jacobian support: H + damping*diag(H) is singular and the
    reference raises from torch.inverse (models_ford.py:446); the HIP path must raise too, not return a NaN pose.
    (2) No pixel of the batch projecting inside the satellite map: the reference asserts (jacobian.py:172); the HIP path
    reproduces that under args.strict_errors = 1 (it costs a host sync) and otherwise leaves the pose unchanged."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_ford import LM_S2GP_Ford
    d = _dev()
    R_FL = torch.tensor([[[0., 0., 1.], [1., 0., 0.], [0., 1., 0.]]])
    T_FL = torch.tensor([[1.7, 0.3, -1.2]])
    # (1) a naturally singular case found by tests/diag/fuzz_e2e.py (seed 59)
    args = O.default_args(N_iters=2, using_weight=1, use_hessian=1)
    grd_hw, sat_a = (88, 152), 104
    sd = O.synth_model_state(59, bias_scale=0.02)
    sat, grd, *_ = O.synth_images(1059, 1, grd_hw=grd_hw, sat_a=sat_a)
    on = O.LM_S2GP_Ford(args, grd_hw=grd_hw)
    on.load_state_dict(sd)
    with pytest.raises(RuntimeError, match='singular'), torch.no_grad():      # torch.linalg.LinAlgError is a RuntimeError
        on.double()(sat.double(), grd.double(), 0.22 * sat_a, R_FL.double(), T_FL.double(), mode='test')
    net = LM_S2GP_Ford(args)
    net.load_state_dict(sd)
    net = net.to(d)
    with pytest.raises(RuntimeError, match='singular'), torch.no_grad():
        net(sat.to(d), grd.to(d), 0.22 * sat_a, R_FL.to(d), T_FL.to(d), mode='test')
    # (2) a 1 mm satellite footprint: every projected point falls outside the map
    args = O.default_args(N_iters=1)
    on = O.LM_S2GP_Ford(args, grd_hw=grd_hw)
    on.load_state_dict(sd)
    with pytest.raises(AssertionError), torch.no_grad():
        on.double()(sat.double(), grd.double(), 1e-3, R_FL.double(), T_FL.double(), mode='test')
    net = LM_S2GP_Ford(args)
    net.load_state_dict(sd)
    net = net.to(d)
    with torch.no_grad():
        out = net(sat.to(d), grd.to(d), 1e-3, R_FL.to(d), T_FL.to(d), mode='test')
    assert all(float(o.abs().max()) == 0.0 for o in out)          # J = 0: the pose stays at its initial value
    net.args.strict_errors = 1
    with pytest.raises(AssertionError, match='jacobian.py:172'), torch.no_grad():
        net(sat.to(d), grd.to(d), 1e-3, R_FL.to(d), T_FL.to(d), mode='test')


def test_bad_arguments_raise():
    """Error behaviour at the boundary: every misuse is a Python exception carrying hla_last_error(), never a crash."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd import _lib
    from highlyaccurate_amd.VGG import VGGUnet
    from highlyaccurate_amd.models_kitti import LM_S2GP
    d = _dev()
    net = VGGUnet(3).to(d)
    with pytest.raises(_lib.HlaError, match='multiples of 8'):
        net(torch.zeros(1, 3, 36, 64, device=d))
    with pytest.raises(ValueError):
        VGGUnet(3, precision='int8')
    with pytest.raises(NotImplementedError):
        VGGUnet(5)
    with pytest.raises(NotImplementedError):
        LM_S2GP(O.default_args(Optimizer='NN'))
    m = LM_S2GP(O.default_args(N_iters=1)).to(d)
    with pytest.raises(Exception):                       # sat / grd batch mismatch
        with torch.no_grad():
            m(torch.zeros(2, 3, 128, 128, device=d), torch.zeros(1, 3, 64, 256, device=d), mode='test')
    lib = _lib.load()
    assert lib.hla_vgg_forward(None, 0, None, None, None, None, None, None, 0, 1, 8, 8, 3, 0, 0, 0, None) != 0
    assert b'null' in lib.hla_last_error()


def test_g2s_e2e_vs_reference_golden_and_oracle():
    """LM_G2SP (ground -> satellite direction), full KITTI shape, fp32 mode.  The reference itself only
    runs in fp32, so the gate is |hip - oracle_fp64| <= max(tol, 2*|ref_fp32 - oracle_fp64|) per component, and
    additionally |hip - ref_fp32| is reported."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_G2SP
    g = load_golden('e2e_kitti_g2s.npz')
    B = int(g['B'])
    d = _dev()
    for si, seed in enumerate(int(s) for s in g['seeds']):
        for uw, key in ((0, f'trace32_{seed}'), (1, f'trace32w_{seed}'))[:2 if si == 0 else 1]:
            args = O.default_args(using_weight=uw)
            sd = O.synth_model_state(seed)
            sd['damping'] = args.damping * torch.ones(1, 3)
            onet = O.LM_G2SP(args)
            onet.load_state_dict(sd)
            onet = onet.double()
            sat, grd, gu, gv, gh = O.synth_images(seed + 100, B)
            K = torch.tensor([O.KITTI_K], dtype=torch.float32).repeat(B, 1, 1)
            with torch.no_grad():
                onet(sat.double(), grd.double(), K, mode='test')
            lat, lon, th = onet.trace
            o64 = torch.stack([lon, lat, th], -1).reshape(B, -1, 3).numpy()
            net = LM_G2SP(args)
            net.load_state_dict(sd)
            net = net.to(d)
            with torch.no_grad():
                res = net(sat.to(d), grd.to(d), K.to(d), mode='test')
            got = net.last_trace.reshape(B, -1, 3).cpu().numpy().astype(np.float64)
            _pose_gate(got, o64, g[key], f'g2s seed {seed} using_weight={uw}')
            print(f'   |hip - reference_fp32| max {np.abs(got - g[key]).max():.2e}')
            np.testing.assert_array_equal(torch.stack(res, -1).cpu().numpy()[:, [1, 0, 2]], got[:, -1].astype(np.float32))
    # train-mode tuple (values), bf16 mode finite
    with torch.no_grad():
        tup = net(sat.to(d), grd.to(d), K.to(d), gu.to(d), gv.to(d), gh.to(d), mode='train')
    assert len(tup) == 14 and tup[13][2].shape == (B, 1, 128, 512)
    nb = LM_G2SP(O.default_args(precision='bf16')).to(d)
    with torch.no_grad():
        r = nb(sat.to(d), grd.to(d), K.to(d), mode='test')
    assert all(torch.isfinite(x).all() for x in r)


@pytest.mark.parametrize('kw', [dict(), dict(using_weight=1), dict(train_damping=1, using_weight=1)])
def test_g2s_lm_backward_small_vs_oracle_autograd(kw):
    """hla_g2s_lm_solve_bwd on random feature pyramids against torch autograd through the fp64 oracle's
    project_grd_to_map + LM_update chain (perspective projection, projected-confidence weights, lambda = parameter)."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_G2SP
    d = _dev()
    args = O.default_args(N_iters=2, damping=0.5, **kw)
    B, grd_hw, sat_a = 2, (64, 256), 128
    rs = np.random.RandomState(17)
    Cs, L = (256, 128, 64), 3
    sat = [T(rs.standard_normal((B, Cs[l], sat_a >> (2 - l), sat_a >> (2 - l))).astype(np.float32) * 0.02) for l in range(L)]
    grd = [T(rs.standard_normal((B, Cs[l], grd_hw[0] >> (3 - l), grd_hw[1] >> (3 - l))).astype(np.float32) * 0.02) for l in range(L)]
    conf = [T(rs.uniform(0.27, 0.5, size=(B, 1, grd_hw[0] >> (3 - l), grd_hw[1] >> (3 - l))).astype(np.float32)) for l in range(L)]
    K = (torch.tensor([O.KITTI_K]) * torch.tensor([[grd_hw[1] / 1024.0], [grd_hw[0] / 256.0], [1.0]])).float().repeat(B, 1, 1)
    p0 = T(rs.uniform(-0.2, 0.2, size=(B, 3)).astype(np.float32))
    coef = T(rs.standard_normal((B, args.N_iters, L, 3)))
    onet = O.LM_G2SP(args).double()
    dpar = torch.tensor([[0.4, 0.7, 0.55]])
    with torch.no_grad():
        onet.damping.copy_(dpar.double())
    sat64 = [s.double().requires_grad_(True) for s in sat]
    grd64 = [g.double().requires_grad_(True) for g in grd]
    conf64 = [c.double().requires_grad_(True) for c in conf]
    su, sv, th = (p0[:, i:i + 1].double() for i in range(3))
    loss = 0.0
    for it in range(args.N_iters):
        for l in range(L):
            f, c, jac = onet.project_grd_to_map(grd64[l], conf64[l], su, sv, th, K, sat64[l].shape[-1], *grd_hw)
            su, sv, th = O.lm_update_g2s(args, onet.damping, su, sv, th, f, c, sat64[l], jac, args.using_weight)
            loss = loss + (coef[:, it, l] * torch.cat([su, sv, th], 1)).sum()
    loss.backward()
    net = LM_G2SP(args).to(d)
    with torch.no_grad():
        net.damping.copy_(dpar.to(d))
    nh = lambda t: t.permute(0, 2, 3, 1).contiguous().to(d)
    feats = ([nh(s) for s in sat], [nh(g) for g in grd], [c[:, 0].contiguous().to(d) for c in conf])
    trace = net.lm_solve(*feats, K.to(d), grd_hw, init_pose=p0, keep_normal_eq=True)
    ref_last = torch.cat([su, sv, th], 1).detach().numpy()
    assert np.abs(trace[:, -1, -1].cpu().numpy() - ref_last).max() < 1e-5
    d_sat, d_grd, d_conf, d_lam = net.lm_backward(*feats, K.to(d), grd_hw, trace, net.last_normal_eq, coef.float().to(d), init_pose=p0)
    for l in range(L):
        for name, got, ref in (('sat', d_sat[l], sat64[l].grad), ('grd', d_grd[l], grd64[l].grad)):
            got = got.permute(0, 3, 1, 2).cpu().double().numpy()
            e = np.abs(got - ref.numpy()).max() / max(np.abs(ref.numpy()).max(), 1e-30)
            print(f'g2s lm bwd {kw} level {l} d_{name}: rel err {e:.2e} (max |ref| {np.abs(ref.numpy()).max():.2e})')
            assert e < 2e-4, (kw, l, name, e)
        if args.using_weight:
            got = d_conf[l].cpu().double().numpy()
            ref = conf64[l].grad[:, 0].numpy()
            e = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)
            print(f'g2s lm bwd level {l} d_conf: rel err {e:.2e}')
            assert e < 2e-4
    if args.train_damping:
        ref = onet.damping.grad.numpy()
        e = np.abs(d_lam.cpu().view(1, 3).numpy() - ref).max() / np.abs(ref).max()
        print(f'g2s lm bwd d_damping: rel err {e:.2e}')
        assert e < 1e-5


@pytest.mark.parametrize('level', [3, 4])
def test_g2s_train_step_vs_oracle_autograd_small(level):
    """LM_G2SP mode='train' under autograd on a reduced shape: loss and parameter gradients vs the fp64 oracle."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_G2SP
    d = _dev()
    args = O.default_args(level=level, N_iters=2, using_weight=1, train_damping=1)
    B, grd_hw, sat_a = 2, (64, 256), 128
    sd = O.synth_model_state(4, bias_scale=0.02)
    sd['damping'] = torch.tensor([[0.1, 0.2, 0.15]])
    sat, grd, gu, gv, gh = O.synth_images(9, B, grd_hw=grd_hw, sat_a=sat_a)
    K = (torch.tensor([O.KITTI_K]) * torch.tensor([[grd_hw[1] / 1024.0], [grd_hw[0] / 256.0], [1.0]])).float().repeat(B, 1, 1)
    on = O.LM_G2SP(args)
    on.load_state_dict(sd)
    on = on.double()
    ro = on(sat.double(), grd.double(), K, gu.double(), gv.double(), gh.double(), mode='train')
    ro[0].backward()
    ref = {k: p.grad for k, p in on.named_parameters()}
    net = LM_G2SP(args)
    net.load_state_dict(sd)
    net = net.to(d).train()
    r = net(sat.to(d), grd.to(d), K.to(d), gu.to(d), gv.to(d), gh.to(d), mode='train')
    assert abs(float(r[0]) - float(ro[0])) < 1e-4 * abs(float(ro[0]))
    r[0].backward()
    worst = 0.0
    for k, p in net.named_parameters():
        assert (p.grad is None) == (ref[k] is None), k
        if p.grad is None:
            continue
        rr = ref[k].numpy()
        e = np.abs(p.grad.cpu().double().numpy() - rr).max() / max(np.abs(rr).max(), 1e-30)
        worst = max(worst, e)
        print(f'g2s train grad {k:36s} rel err {e:.2e} (max |ref| {np.abs(rr).max():.2e})')
        assert e < 5e-3, (k, e)
    print('g2s train grads worst rel err', worst)


def test_g2s_train_step_vs_reference_autograd_golden():
    """Full KITTI shape, LM_G2SP mode='train' with using_weight=1 + train_damping=1: loss.backward() through the HIP
    backward against gradient samples from the REAL reference's autograd (which only exists in fp32)."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_G2SP
    from make_idx import sample_idx
    g = load_golden('e2e_kitti_g2s.npz')
    seed, B = int(g['seeds'][0]), int(g['B'])
    d = _dev()
    args = O.default_args(using_weight=1, train_damping=1)
    sd = O.synth_model_state(seed)
    sd['damping'] = args.damping * torch.ones(1, 3)
    net = LM_G2SP(args)
    net.load_state_dict(sd)
    net = net.to(d).train()
    sat, grd, gu, gv, gh = O.synth_images(seed + 100, B)
    K = torch.tensor([O.KITTI_K], dtype=torch.float32).repeat(B, 1, 1)
    res = net(sat.to(d), grd.to(d), K.to(d), gu.to(d), gv.to(d), gh.to(d), mode='train')
    assert abs(float(res[0].detach()) - g['wtuple32'][0][0]) < 1e-3 * abs(g['wtuple32'][0][0])
    res[0].backward()
    named = dict(net.named_parameters())
    nograd = set(str(k) for k in g['nograd_32'])
    for k, p in named.items():
        assert (p.grad is None) == (k in nograd), k
    for k in [k[len('grad32_'):] for k in g.files if k.startswith('grad32_')]:
        ref = g['grad32_' + k]
        gr = named[k].grad.double().reshape(-1).cpu()
        got = gr[sample_idx(gr.numel(), 77)].numpy()
        o64 = g['ograd64_' + k][2:]                       # fp64 restatement: how much is the reference's own fp32 rounding?
        scale = np.abs(ref[2:]).max()
        e, e64, gap = np.abs(got - ref[2:]).max(), np.abs(got - o64).max(), np.abs(ref[2:] - o64).max()
        print(f'g2s train grad {k:36s} |hip-ref32| {e / scale:.2e}  |hip-oracle64| {e64 / scale:.2e}  |ref32-oracle64| '
              f'{gap / scale:.2e} (scale {scale:.2e})')
        # Every stage is exact for the inputs it sees (tests/diag/diag_g2s_e2e.py: LM backward 2e-7 against the oracle evaluated
        # on the HIP feature maps; VGG backward 5e-7 given the oracle's upstream gradients, tests/diag/diag_g2s_vgg.py), but this
        # direction has no renormalisation and its Jacobian is built from differences of neighbouring texels of nearly
        # constant level-2 maps, so the 2e-6 (of max-abs) feature deviation of the fp32-MFMA extractor is amplified to a few
        # 1e-3 in the parameter gradients (the reference's own fp32-vs-fp64 gap on the same quantity is 1e-4).
        assert e64 <= max(1.5e-2 * scale, 3 * gap), (k, e64, gap, scale)


def test_training_steps_do_not_accumulate_memory():
    """Regression: the autograd context must not sit in a reference cycle (ctx -> saved trace -> grad_fn -> ctx); every step's
    saved workspaces (8.5 GB at the bench shape) would then live until the cyclic GC happens to run."""
    import gc
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_G2SP, LM_S2GP
    d = _dev()
    sat, grd, gu, gv, gh = O.synth_images(3, 2, grd_hw=(64, 256), sat_a=128)
    K = (torch.tensor([O.KITTI_K]) * torch.tensor([[0.25], [0.25], [1.0]])).float().repeat(2, 1, 1).to(d)
    gc.collect()
    gc.disable()
    try:
        for cls, extra in ((LM_S2GP, ()), (LM_G2SP, (K,))):
            net = cls(O.default_args(N_iters=2)).to(d).train()
            used = []
            for _ in range(5):
                net.zero_grad(set_to_none=True)
                r = net(sat.to(d), grd.to(d), *extra, gu.to(d), gv.to(d), gh.to(d), mode='train')
                r[0].backward()
                del r
                torch.cuda.synchronize()
                used.append(torch.cuda.memory_allocated())
            assert used[-1] <= used[1] + (1 << 20), (cls.__name__, used)
    finally:
        gc.enable()


@pytest.mark.multirank
def test_bench_gpus_flag_spawns_its_ranks_and_reports_them():
    """`python bench.py --gpus 2` with NO launcher and no WORLD_SIZE must spawn two ranks itself and say so in its one JSON line
    (round 2's bench parsed --gpus and ignored it).  On this one-GPU box the two ranks share the device (rehearsal: gloo instead
    of RCCL, flagged in the line), which exercises everything but the transport: rank spawn, rendezvous on 127.0.0.1, the
    collective census, barriers + MAX-over-ranks timing, the batch-sharded inference leg, the data-parallel training leg with
    the in-place all-reduce of the two flat gradient buffers (19.78 MB of live fp32 gradients, SURVEY B-8), rank-0-only output."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT',
                                                            'HLA_BENCH_REHEARSE')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--train-steps', '2', '--batch', '8', '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]                      # ONE line, from rank 0
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['collective_ranks_seen'] == 2 and j['self_launched'] is True
    assert j['rehearsal'] == (torch.cuda.device_count() < 2) and j['n_gpus_physical'] == min(2, torch.cuda.device_count())
    assert j['collective_backend'].startswith('gloo' if j['rehearsal'] else 'nccl')
    assert j['config']['global_batch'] == 16 and j['config']['pairs_per_gpu'] == 8 and j['value'] > 0 and j['scaling'] == 'weak'
    t = j['train']
    assert 'error' not in t, t
    assert abs(t['allreduce_bytes_per_step'] - 19.78e6) < 0.02e6, t['allreduce_bytes_per_step']     # 4 945 536 fp32, once per step
    assert t['loss_finite'] and t['value'] > 0 and t['single_rank_value'] > 0 and 0 < t['scaling_eff'] < 1.5
    assert 'roofline' in j and 'scale_reads' in j


@pytest.mark.multirank
def test_bench_eight_rank_rehearsal_of_configs2_shard_arithmetic():
    """BASELINE configs[2] is 8 ranks; no 8-GPU node has been available to any round.  What CAN run here is its rank count: a plain
    `python bench.py --gpus 8 --batch 4` spawns eight ranks that share this box's GPU (rehearsal: gloo, flagged) -- the census
    must see 8 ranks, the global batch is 8 x 4 = 32 pairs, every training step all-reduces the 19.78 MB of live gradients once,
    rank 0 prints ONE line.  Control flow and shard arithmetic at the real rank count; not a measurement of anything."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT',
                                                            'HLA_BENCH_REHEARSE')}
    env['OMP_NUM_THREADS'] = '4'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '2', '--warmup', '1',
                        '--train-steps', '1', '--train-precision', 'bf16', '--batch', '4', '--no-cpu-baseline', '--no-kernel-timing'],
                       env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    ndev = torch.cuda.device_count()
    assert j['n_gpus'] == 8 and j['collective_ranks_seen'] == 8 and j['self_launched'] is True
    assert j['rehearsal'] == (ndev < 8) and j['n_gpus_physical'] == min(8, ndev)
    assert j['config']['global_batch'] == 32 and j['config']['pairs_per_gpu'] == 4 and j['value'] > 0 and j['scaling'] == 'weak'
    t = j['train']
    assert 'error' not in t, t
    assert abs(t['allreduce_bytes_per_step'] - 19.78e6) < 0.02e6, t['allreduce_bytes_per_step']
    assert t['loss_finite'] and t['value'] > 0 and t['single_rank_value'] > 0


@pytest.mark.multirank
def test_bench_single_rank_through_rccl():
    """The N > 1 code path on the REAL transport, as far as a one-GPU box can take it: HLA_BENCH_FORCE_DIST=1 runs the N = 1 job
    through a one-rank RCCL process group, so every collective call of the multi-GPU path -- the census all-reduce, the barriers,
    MAX over ranks on a device tensor, the asynchronous in-place all-reduce of the two flat gradient buffers from inside the
    backward, wait + divide -- executes on RCCL (the two-rank test above has to fall back to gloo when the ranks share a GPU:
    RCCL refuses two ranks on one device).  A CPU tensor in a collective, a wait on the wrong stream or a buffer the backend
    cannot take would fail here and not there."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT',
                                                            'HLA_BENCH_REHEARSE')}
    env['HLA_BENCH_FORCE_DIST'] = '1'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1',
                        '--train-steps', '2', '--batch', '8', '--no-cpu-baseline', '--no-extra-legs'],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j['n_gpus'] == 1 and j['collective_ranks_seen'] == 1 and j['collective_backend'].startswith('nccl') and not j['rehearsal']
    # round 5: clock / power of the timed region and the training step's roofline are IN the line (VERDICT r04 #1a, #5)
    rf = j['roofline']
    assert rf['telemetry']['samples'] > 0 and 500 < rf['clock_mhz_mean'] < 2600 and 100 < rf['power_w_mean'] < 1600, rf.get('telemetry')
    assert 'mfma_sustained_clock_mhz_mean' in rf and rf['mfma_sustained']['telemetry']['random']['samples'] > 0
    t = j['train']
    assert 'error' not in t, t
    assert abs(t['allreduce_bytes_per_step'] - 19.78e6) < 0.02e6, t['allreduce_bytes_per_step']
    assert t['loss_finite'] and t['value'] > 0
    assert t['roofline']['bound'] == 'mfma' and 0.05 < t['roofline']['frac'] < 1.0 and 300 < t['gflop_per_pair_executed'] < 1000, t.get('roofline')
    # round 6: the overlap evidence of the gradient all-reduce is in the line whenever the collectives run
    sb = t['step_breakdown']
    assert 0.0 <= sb['allreduce_exposed_ms'] < 5.0 and len(sb['allreduce_buckets']) == 2, sb
    assert all(abs(b['bytes'] - 9.89e6) < 0.02e6 for b in sb['allreduce_buckets']), sb['allreduce_buckets']
    assert sb['allreduce_buckets'][0]['issued_ms_before_backward_end'] > sb['allreduce_buckets'][1]['issued_ms_before_backward_end'] >= 0.0, sb


def test_grad_sync_over_rccl_leaves_one_rank_gradients_unchanged():
    """GradSync on RCCL (one-rank group, force=True) inside the model's own backward: the flat buffers go through
    ncclAllReduce in place and come back divided by 1 -- every parameter gradient must equal the run without a process group
    (the all-reduce of one rank is the identity; 1e-5 relative covers the LM backward's fp32 atomics, whose order differs from
    run to run, and the single-stream instead of two-stream backward), and the grad-less parameters stay grad-less.  The exchange is asynchronous on RCCL's own stream: a missing wait or a wait on the wrong stream shows up as a
    stale or half-divided buffer here."""
    import torch.distributed as dist
    from highlyaccurate_amd.parallel import GradSync
    import socket
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_S2GP
    dev = _dev()
    net = LM_S2GP(O.default_args(precision='fp32'))
    net.load_state_dict(O.synth_model_state(3))
    net = net.to(dev)
    rs = np.random.RandomState(5)
    B = 2
    sat = torch.from_numpy(rs.random_sample((B, 3, 512, 512)).astype(np.float32)).to(dev)
    grd = torch.from_numpy(rs.random_sample((B, 3, 256, 1024)).astype(np.float32)).to(dev)
    gt = [torch.from_numpy(rs.uniform(-1, 1, (B, 1)).astype(np.float32)).to(dev) for _ in range(3)]

    def grads():
        net.zero_grad(set_to_none=True)
        torch.manual_seed(0)
        r = net(sat, grd, gt[0], gt[1], gt[2], mode='train')
        r[0].backward()
        torch.cuda.synchronize()
        return {n: (None if p.grad is None else p.grad.clone()) for n, p in net.named_parameters()}

    base = grads()
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=dev)
    try:
        net.grad_sync = GradSync(force=True)
        got = grads()
        assert net.grad_sync.collectives >= 2 and abs(net.grad_sync.bytes_reduced - 19.78e6) < 0.02e6
    finally:
        net.grad_sync = None
        dist.destroy_process_group()
    for n, g in base.items():
        assert (g is None) == (got[n] is None), n
        if g is not None:
            rel = float((g - got[n]).double().norm() / g.double().norm().clamp_min(1e-30))
            assert rel < 1e-5, (n, rel)


# A fixed slice of the randomised harness (tests/diag/fuzz_e2e.py: random batch / image sizes -- mostly not multiples of the conv
# and LM tiles --, model family, level, flags, iteration counts; forward poses on even seeds, training loss + every parameter
# gradient on odd ones, against the fp64 oracle): LM updater only, including the reference's two run-time errors (three cases
# where no pixel of the batch is in view -> AssertionError, jacobian.py:172; two singular normal matrices -> LinAlgError) and the
# largest displacements the harness has produced (2090: |shift| 2.8, past the re-initialisation threshold; 2462: 1.9).
_FUZZ_SLICE = [2050, 2085, 2090, 2096, 2110, 2120, 2452, 2453, 2454, 2456, 2457, 2459, 2460, 2461, 2462, 2463, 2464, 2465, 2466,
               2467, 2468, 2469, 2470, 2471, 2473, 2474, 2475, 2477, 2478, 2302]


@pytest.mark.parametrize('chunk', [0, 1, 2])
def test_fuzz_slice_vs_oracle(chunk):
    import importlib.util, os
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'diag', 'fuzz_e2e.py')
    spec = importlib.util.spec_from_file_location('fuzz_e2e', p)
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    bad = [s for s in _FUZZ_SLICE[chunk::3] if not fz.one_case(s)]
    assert not bad, f'fuzz seeds beyond their bound: {bad}'


def test_fuzz_16bit_kernels_slice_vs_fp32_mode():
    """A fixed slice of tests/diag/fuzz_16bit.py: the 16-bit kernels' own code paths (LDS-DMA halo loader -- source-side swizzle,
    range-checked zero border --, bias-initialised accumulators, batched epilogue; in the backward the same loader under the
    data-dependent tile lists) on random ragged shapes, levels 3 / 4, forward maps and parameter gradients against the library's
    exact-fp32 mode.  (The harness's 120 cases of round 4: profiles/r04_fuzz_16bit_kernels_120_cases.txt.)"""
    import importlib.util, os
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'diag', 'fuzz_16bit.py')
    spec = importlib.util.spec_from_file_location('fuzz_16bit', p)
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    bad = [s for s in range(5000, 5016) if not fz.one_case(s)]
    assert not bad, f'16-bit fuzz seeds beyond their bound: {bad}'


@pytest.mark.parametrize('precision', ['fp32', 'fp16x3'])
def test_fuzz_ill_conditioned_seed_2515_is_excused_by_the_reference_itself(precision):
    """the one recorded failure of the split-fp16 backward's fuzz run (seed 2515: LM_G2SP, use_hessian + train_damping;
    the `damping` gradient came out with cosine -1) was explained as ill-conditioning in prose only.  The harness now checks the
    explanation: a tensor beyond the cosine gate is excused only if the REFERENCE's own fp32 autograd is beyond the gate on the
    same tensor (against its fp64 autograd), and every other tensor still has to pass.  Both fp32-class modes."""
    import importlib.util, os
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'diag', 'fuzz_e2e.py')
    spec = importlib.util.spec_from_file_location('fuzz_e2e', p)
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    assert fz.one_case(2515, precision=precision)


@pytest.mark.parametrize('precision', ['fp32', 'fp16x3', 'bf16', 'fp16'])
def test_train_step_gradient_fidelity_by_precision(precision):
    """The number bench.py reports as train.by_precision.<mode>.gradients, gated: worst per-tensor relative L2 error and cosine
    of the full-shape training-step gradients against the REAL reference's autograd (tests/golden/train_kitti.npz, fp64 run).
    fp32 / fp16x3 (the matched-accuracy training mode: split-fp16 dgrad + wgrad kernels): within 3x the reference's OWN
    fp32-vs-fp64 gap (max-pool flip noise, see test_train_step_gradients_vs_reference_golden) -- bench.py's train.value is the
    fp16x3 step for that reason.  bf16 / fp16, the reduced-precision steps (reported as parity_grade: false): 3x their measured
    deviation (bf16 0.237 / cosine 0.9726, fp16 0.0774 / 0.99700 on MI355X), so that a regression of a factor of three fails.
    Their error is the BACKWARD's own rounding and no format choice repairs it: behind an EXACT forward + LM loop (fp16x3), the
    extractors' backward alone in fp16 / bf16 gives the same 0.077 / 0.23 (tools/probes/mixed_bwd_fidelity.py; gradient scaling by
    2^0..2^16 changes nothing: no underflow) -- these weight gradients are sums that cancel to 1/20..1/300 of their terms, so 8 or
    11 significand bits in either operand of the 16-bit dgrad / wgrad products are amplified by that factor."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    r = bench.gradient_fidelity(precision, _dev())
    print(f'gradient fidelity [{precision}]: {r}')
    if precision == 'bf16':
        assert r['worst_rel_l2'] < 3 * 0.237 and r['worst_cosine'] > 1 - 3 * (1 - 0.9726) and r['loss_rel_err'] < 3 * 3.2e-4
    elif precision == 'fp16':
        assert r['worst_rel_l2'] < 3 * 0.0774 and r['worst_cosine'] > 1 - 3 * (1 - 0.99700) and r['loss_rel_err'] < 3 * 5.9e-5
    else:
        assert r['worst_rel_l2'] < 3 * r['reference_fp32_vs_fp64_worst_rel_l2'] and r['worst_cosine'] > 0.99998
        assert r['loss_rel_err'] < 1e-6


def test_fp32_class_modes_need_fp32_lm_maps():
    """VERDICT r02 asked to let the fp32-class modes' LM loop read 16-bit maps "or at least prove fp32 maps are needed".  The proof:
    the extractor of the matched-accuracy mode (fp16x3) hands the loop fp32 maps and the 15-step trace passes the parity gate;
    the SAME maps rounded to fp16 (what HLA_VGG_FEAT16 would store: 11 significand bits) through the SAME loop miss it by two
    orders of magnitude -- the loop amplifies a 2^-12 relative perturbation of the features to ~1e-4 in the pose, against a
    tolerance of 5e-6.  Half the bytes are not worth a mode that no longer reproduces the reference."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_S2GP
    g = load_golden('e2e_kitti.npz')
    seed, B = int(g['seeds'][0]), int(g['B'])
    d = _dev()
    net = LM_S2GP(O.default_args(precision='fp16x3'))
    net.load_state_dict(O.synth_model_state(seed))
    net = net.to(d)
    sat, grd, *_ = O.synth_images(seed + 100, B)
    with torch.no_grad():
        sf, si, gf, gc, gi = net._features(sat.to(d), grd.to(d), False, False)
        torch.manual_seed(seed)
        t32 = net.lm_solve(sf, gf, gc, grd.shape[-2:], None, 0, None, si, gi).clone()
        torch.manual_seed(seed)
        t16 = net.lm_solve([f.half() for f in sf], [f.half() for f in gf], gc, grd.shape[-2:], None, 0, None, si, gi).clone()
    ref = g[f'trace64_{seed}']
    e32 = np.abs(_exec_order(t32, 0).cpu().numpy().astype(np.float64) - ref)
    e16 = np.abs(_exec_order(t16, 0).cpu().numpy().astype(np.float64) - ref)
    print(f'fp16x3 extractor, LM loop on fp32 maps: shift {e32[..., :2].max():.2e} yaw {e32[..., 2].max():.2e}; '
          f'on the same maps rounded to fp16: shift {e16[..., :2].max():.2e} yaw {e16[..., 2].max():.2e} (tolerance {TOL_SHIFT:.0e} / {TOL_YAW:.1e})')
    _pose_gate(_exec_order(t32, 0).cpu().numpy().astype(np.float64), ref, g[f'trace32_{seed}'], 'fp16x3 + fp32 maps')
    assert e16[..., :2].max() > 10 * max(TOL_SHIFT, e32[..., :2].max())          # the 16-bit maps are NOT good enough


@pytest.mark.parametrize('precision', ['fp16x3', 'bf16'])
def test_wave_specialised_wgrad_matches_the_two_phase_kernels(precision, monkeypatch):
    """Round 5's weight-gradient kernels (wgrad_split_ws_kernel / wgrad_ws_kernel: 4 matrix + 4 loader waves per CU, 256 resident
    workgroups) against the round-4 kernels they replace (args.wgrad_two_phase = 1 -> HLA_VGG_BWD_WGRAD_TWO_PHASE: wgrad_split_kernel / wgrad_dma_kernel /
    wgrad_kernel, 512 workgroups): the same products in a different split-K grouping, so every weight and bias gradient of a
    full-shape training step must agree to the order of the partial sums: 1e-5 of the tensor's norm in split mode, whose products
    are fp32-class; for bf16 2e-3 or four times what the SAME kernels differ by from one run to the next (the LM backward's
    atomics move bf16-rounded gradient maps by an ulp here and there, and the bias gradients sum everything).  Ragged batch
    (B = 3), full KITTI shape: plain, up-sampled, concatenated and un-pooled operands, and the ground branch's ODD first rows --
    this test is what caught the wave-specialised loaders splitting an odd tile origin through the upsample shift."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_S2GP
    d = _dev()
    net = LM_S2GP(O.default_args(precision=precision))
    net.load_state_dict(O.synth_model_state(5))
    net = net.to(d).train()
    net.args.bwd_two_streams = 0
    B = 3
    sat, grd, gu, gv, gh = O.synth_images(105, B)
    sat, grd, gt = sat.to(d), grd.to(d), [gu.to(d), gv.to(d), gh.to(d)]

    def grads():
        net.zero_grad(set_to_none=True)
        torch.manual_seed(3)
        r = net(sat, grd, *gt, mode='train')
        r[0].backward()
        return {n: p.grad.detach().double().clone() for n, p in net.named_parameters() if p.grad is not None}

    net.args.wgrad_two_phase = 0
    g_ws, g_ws2 = grads(), grads()          # twice: the step's own run-to-run noise (LM-backward atomics), per tensor
    net.args.wgrad_two_phase = 1
    g_old = grads()
    rel = lambda a, b: float((a - b).norm() / max(float(b.norm()), 1e-30))
    tol = 1e-5 if precision == 'fp16x3' else 2e-3
    worst, wn, wnoise, bad = 0.0, '', 0.0, []
    for n in g_ws:
        e, noise = rel(g_ws[n], g_old[n]), rel(g_ws[n], g_ws2[n])
        if e > worst:
            worst, wn, wnoise = e, n, noise
        if e > max(tol, 4 * noise):
            bad.append((n, e, noise))
    print(f'ws vs two-phase wgrad [{precision}]: worst relative L2 {worst:.2e} ({wn}; the same kernels twice: {wnoise:.2e}), {len(g_ws)} tensors')
    assert len(g_ws) >= 36 and not bad, bad


@pytest.mark.parametrize('precision', ['fp32', 'fp16x3', 'bf16'])
def test_fused_conv0_weight_gradient_matches_the_stored_map_kernel(precision):
    """Round 6: conv0's weight / bias gradient is formed inside the epilogue of conv2's un-pooling data gradient
    (conv_epilogue_wg0: the 64-channel full-resolution gradient map is never written) instead of by wgrad0_kernel from the stored
    map (args.wgrad_two_phase = 2 -> HLA_VGG_BWD_WGRAD0_UNFUSED).  The same products -- exact fp32: the same MFMAs; split mode:
    (hi, lo) fp16 pairs of both operands instead of the fp32 MFMA, i.e. 2^-22 per product; bf16: the map rounded to bf16 in both --
    in another summation grouping, so the two agree to the order of the partial sums.  Ragged batch (B = 3), full KITTI shape:
    the satellite branch's data-dependent tile list and the ground branch's ODD first row both go through the fused epilogue.
    Everything else of the step must not move at all beyond the LM backward's atomics."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_S2GP
    d = _dev()
    net = LM_S2GP(O.default_args(precision=precision))
    net.load_state_dict(O.synth_model_state(5))
    net = net.to(d).train()
    net.args.bwd_two_streams = 0
    B = 3
    sat, grd, gu, gv, gh = O.synth_images(105, B)
    sat, grd, gt = sat.to(d), grd.to(d), [gu.to(d), gv.to(d), gh.to(d)]

    def grads():
        net.zero_grad(set_to_none=True)
        torch.manual_seed(3)
        r = net(sat, grd, *gt, mode='train')
        r[0].backward()
        return {n: p.grad.detach().double().clone() for n, p in net.named_parameters() if p.grad is not None}

    net.args.wgrad_two_phase = 0
    g_f, g_f2 = grads(), grads()
    net.args.wgrad_two_phase = 2
    g_u = grads()
    net.args.wgrad_two_phase = 0
    rel = lambda a, b: float((a - b).norm() / max(float(b.norm()), 1e-30))
    tol = {'fp32': 1e-5, 'fp16x3': 3e-5, 'bf16': 2e-3}[precision]
    bad, seen = [], 0
    for n in g_f:
        e, noise = rel(g_f[n], g_u[n]), rel(g_f[n], g_f2[n])
        if 'conv0.' in n:
            seen += 1
            print(f'fused vs stored-map conv0 gradient [{precision}] {n}: {e:.2e} (the fused kernels twice: {noise:.2e})')
            assert float(g_f[n].norm()) > 0
        if e > max(tol, 4 * noise):
            bad.append((n, e, noise))
    assert seen == 4 and not bad, bad


@pytest.mark.parametrize('precision', ['fp16x3', 'bf16'])
def test_deterministic_backward_gives_bitwise_equal_gradients(precision):
    """args.deterministic_backward = 1 (hla_s2g_config.deterministic, VERDICT r05 #2c): the same batch twice gives BITWISE equal
    parameter gradients -- the LM backward's scatter into d(loss)/d(sat map) is the one place of the training step whose summation
    order varies from run to run (fp32 atomics), and everything behind it (the satellite extractor's whole backward, incl. its
    data-dependent tile lists) inherits that.  With the flag the scatter adds integers of a per-sample power-of-two quantum.
    Also: the deterministic gradients agree with the default mode's to the atomics' own run-to-run noise, and the default mode
    is NOT bitwise reproducible on this input (if it ever becomes so this test says so instead of silently testing nothing).
    Ragged batch (B = 3), full KITTI shape, both streams of the backward in use."""
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_S2GP
    d = _dev()
    net = LM_S2GP(O.default_args(precision=precision))
    net.load_state_dict(O.synth_model_state(5))
    net = net.to(d).train()
    B = 3
    sat, grd, gu, gv, gh = O.synth_images(105, B)
    sat, grd, gt = sat.to(d), grd.to(d), [gu.to(d), gv.to(d), gh.to(d)]

    def grads():
        net.zero_grad(set_to_none=True)
        torch.manual_seed(3)
        r = net(sat, grd, *gt, mode='train')
        r[0].backward()
        torch.cuda.synchronize()
        return {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}

    net.args.deterministic_backward = 1
    g1, g2, g3 = grads(), grads(), grads()
    assert len(g1) >= 36
    for n in g1:
        assert torch.equal(g1[n], g2[n]) and torch.equal(g1[n], g3[n]), n
    net.args.deterministic_backward = 0
    a1, a2 = grads(), grads()
    rel = lambda a, b: float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-30))
    worst = max(rel(g1[n], a1[n]) for n in g1)
    noise = max(rel(a1[n], a2[n]) for n in g1)
    same = all(torch.equal(a1[n], a2[n]) for n in a1)
    print(f'deterministic backward [{precision}]: 3 runs bitwise equal over {len(g1)} tensors; vs the atomics mode: worst relative L2 '
          f'{worst:.2e} (the atomics mode against itself: {noise:.2e}, bitwise equal: {same})')
    tol = 1e-5 if precision == 'fp16x3' else 2e-3
    assert worst < max(tol, 4 * noise)


def test_zero_fill_clears_strided_regions_in_one_launch():
    """hla_zero_fill (include/hla.h): up to 16 strided regions cleared by one launch; everything outside them untouched."""
    from highlyaccurate_amd import _lib
    d = _dev()
    a = torch.full((5, 7, 64), 3.0, device=d)
    b = torch.full((1000003 * 4,), 2.0, device=d)
    c = torch.full((4, 4), 1.0, device=d, dtype=torch.float64)
    row = 64 * 4
    _lib.zero_fill([(a.data_ptr() + 2 * row, 3 * row, 7 * row, 5), (b, b.numel() * 4, b.numel() * 4, 1), (c.data_ptr() + 32, 32, 32, 1)])
    torch.cuda.synchronize()
    ref = torch.full((5, 7, 64), 3.0)
    ref[:, 2:5] = 0
    assert torch.equal(a.cpu(), ref) and not bool(b.any())
    rc = torch.ones(4, 4, dtype=torch.float64)
    rc[1] = 0
    assert torch.equal(c.cpu(), rc)
    # a region that is only 4-byte granular (odd map sizes: a [B, h, w] confidence gradient of 3 x 27 x 5 floats) takes word stores
    d2 = torch.full((3 * 27 * 5 + 2,), 5.0, device=d)
    _lib.zero_fill([(d2.data_ptr() + 4, 3 * 27 * 5 * 4, 3 * 27 * 5 * 4, 1)])
    torch.cuda.synchronize()
    assert float(d2[0]) == 5.0 and float(d2[-1]) == 5.0 and float(d2[1:-1].abs().max()) == 0.0
    with pytest.raises(_lib.HlaError):
        _lib.zero_fill([(a.data_ptr() + 2, 16, 16, 1)])           # misaligned
    b.fill_(2.0)
    _lib.zero_fill([(b, b.numel() * 4, b.numel() * 4, 1)], max_blocks=3)      # a background fill: same result
    torch.cuda.synchronize()
    assert not bool(b.any())


def test_pose_loss_backward_propagates_nan_like_torch():
    """A diverged (NaN) pose gives a NaN loss AND NaN gradients, like torch's abs / sgn backward: the fused backward
    used to hand finite zeros to the optimizer there."""
    from highlyaccurate_amd._s2gp import loss_func
    d = _dev()
    x = [torch.rand(4, 2, 3, device=d).requires_grad_(True) for _ in range(3)]
    gt = [torch.rand(4, device=d) for _ in range(3)]
    with torch.no_grad():
        x[1][2, 1, 0] = float('nan')
    out = loss_func(0, None, None, None, x[0], x[1], x[2], gt[0], gt[1], gt[2], None, None)
    assert torch.isnan(out[0])
    out[0].backward()
    assert torch.isnan(x[1].grad[2, 1, 0]) and torch.isfinite(x[0].grad).all() and torch.isfinite(x[2].grad).all()
    assert int(torch.isnan(x[1].grad).sum()) == 1
