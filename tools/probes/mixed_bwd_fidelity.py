"""How good are the training-step gradients when ONLY the extractors' backward runs in a 16-bit mode, behind an fp32-class
forward?  (The forward must be fp32-class: the LM loop amplifies feature rounding a few hundred times.  The backward is a linear
map of d_feat.)  Forward + LM loop + LM backward in fp16x3; then hla_vgg_backward in fp16 / bf16 on the saved activations of a
16-bit forward of the SAME weights, with the incoming gradient scaled by 2^k; compared with the reference's fp64 autograd
(tests/golden/train_kitti.npz).     gpurun -- 'python tools/probes/mixed_bwd_fidelity.py'"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from highlyaccurate_amd import synthetic
from highlyaccurate_amd.VGG import vgg_forward_nhwc, vgg_backward_nhwc
from highlyaccurate_amd._s2gp import loss_func

dev = torch.device('cuda:0')
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'train_kitti.npz'), allow_pickle=False)
seed, B = int(g['seed']), int(g['B'])
keys = [k[len('grad64_'):] for k in g.files if k.startswith('grad64_')]
state = synthetic.model_state(seed)
sat, grd, gu, gv, gh = (t.to(dev) for t in synthetic.images(seed + 100, B))

net = bench.build_net('kitti', 'fp16x3', 5, dev, state=state).train()
torch.manual_seed(seed)
sf, _, si, cs = vgg_forward_nhwc(net.SatFeatureNet, sat, want_conf=False, defer_norm=True, save_for_backward=True)
gf, gc, gi, cg = vgg_forward_nhwc(net.GrdFeatureNet, grd, want_conf=True, defer_norm=True, save_for_backward=True)
trace = net.lm_solve(sf, gf, gc, grd.shape[-2:], None, 0, None, si, gi, keep_normal_eq=True)
tr = trace.detach().clone().requires_grad_(True)
a = net.args
out = loss_func(0, None, None, None, tr[..., 1], tr[..., 0], tr[..., 2], gv[:, 0], gu[:, 0], gh[:, 0], None, None,
                a.coe_shift_lat, a.coe_shift_lon, a.coe_heading, a.coe_L1, a.coe_L2, a.coe_L3, a.coe_L4)
out[0].backward()
d_sat, d_grd, d_conf, d_lam = net.lm_backward(sf, gf, gc, grd.shape[-2:], trace, net.last_normal_eq, tr.grad, None, 0, None, si, gi, None)
print('loss', float(out[0]), 'ref', float(g['tuple64'][0][0]), ' |d_sat| max per level', [float(d.abs().max()) for d in d_sat],
      ' |d_grd| max', [float(d.abs().max()) for d in d_grd])


def report(tag, grads):
    worst = 0.0
    for k in keys:
        br, name = k.split('.', 1)
        gr = grads[br][name].double().reshape(-1).cpu().numpy()
        ref = g['grad64_' + k][2:]
        got = gr[synthetic.fixture_sample_idx(gr.size, 77)]
        l2 = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        cos = np.dot(got, ref) / (np.linalg.norm(got) * np.linalg.norm(ref))
        worst = max(worst, l2)
        print(f'   {tag:28s} {k:40s} rel-L2 {l2:.3e}  cos {cos:.7f}')
    print(f'   {tag}: worst rel-L2 {worst:.3e}')


# reference point: the fp16x3 backward itself
report('fp16x3 backward', {'SatFeatureNet': vgg_backward_nhwc(net.SatFeatureNet, cs, d_sat, scale_invariant=True),
                           'GrdFeatureNet': vgg_backward_nhwc(net.GrdFeatureNet, cg, d_grd, scale_invariant=True)})
for prec in ('fp16', 'bf16'):
    n16 = bench.build_net('kitti', prec, 5, dev, state=state).train()
    _, _, _, cs16 = vgg_forward_nhwc(n16.SatFeatureNet, sat, want_conf=False, defer_norm=True, save_for_backward=True)
    _, _, _, cg16 = vgg_forward_nhwc(n16.GrdFeatureNet, grd, want_conf=True, defer_norm=True, save_for_backward=True)
    for k in ((0, 10, 16, 20) if prec == 'fp16' else (0,)):
        S = float(2 ** k)
        gs = vgg_backward_nhwc(n16.SatFeatureNet, dict(cs16), [d * S for d in d_sat], scale_invariant=True)
        gg = vgg_backward_nhwc(n16.GrdFeatureNet, dict(cg16), [d * S for d in d_grd], scale_invariant=True)
        report(f'{prec} backward, scale 2^{k}', {'SatFeatureNet': {n: v / S for n, v in gs.items()}, 'GrdFeatureNet': {n: v / S for n, v in gg.items()}})
        # the 16-bit forward's saved activations are consumed by the backward (released): run it again for the next scale
        _, _, _, cs16 = vgg_forward_nhwc(n16.SatFeatureNet, sat, want_conf=False, defer_norm=True, save_for_backward=True)
        _, _, _, cg16 = vgg_forward_nhwc(n16.GrdFeatureNet, grd, want_conf=True, defer_norm=True, save_for_backward=True)
