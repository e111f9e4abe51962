"""Diagnostic: G2S LM backward with raw maps + deferred inverse norms (A) vs explicitly normalised maps (B)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ref_cpu as O
from highlyaccurate_amd.models_kitti import LM_G2SP
from highlyaccurate_amd.VGG import vgg_forward_nhwc

d = torch.device('cuda:0')
seed, B = 1, 1
args = O.default_args(using_weight=1, train_damping=1)
sd = O.synth_model_state(seed); sd['damping'] = args.damping * torch.ones(1, 3)
net = LM_G2SP(args); net.load_state_dict(sd); net = net.to(d)
sat, grd, gu, gv, gh = O.synth_images(seed + 100, B)
K = torch.tensor([O.KITTI_K], dtype=torch.float32).repeat(B, 1, 1).to(d)
with torch.no_grad():
    sf, _, sinv = vgg_forward_nhwc(net.SatFeatureNet, sat.to(d), want_conf=False, defer_norm=True)
    gf, gc, ginv = vgg_forward_nhwc(net.GrdFeatureNet, grd.to(d), want_conf=True, defer_norm=True)
dtr = torch.randn(B, 5, 3, 3, device=d)
trA = net.lm_solve(sf, gf, gc, K, (256, 1024), None, sinv, ginv, keep_normal_eq=True)
A = net.lm_backward(sf, gf, gc, K, (256, 1024), trA, net.last_normal_eq, dtr, None, sinv, ginv)
sfn = [(f.double() * sinv[l].view(B, 1, 1, 1)).float() for l, f in enumerate(sf)]
gfn = [(f.double() * ginv[l].view(B, 1, 1, 1)).float() for l, f in enumerate(gf)]
trB = net.lm_solve(sfn, gfn, gc, K, (256, 1024), None, None, None, keep_normal_eq=True)
Bb = net.lm_backward(sfn, gfn, gc, K, (256, 1024), trB, net.last_normal_eq, dtr)
print('trace A vs B', (trA - trB).abs().max().item())
for l in range(3):
    for i, name in enumerate(('d_sat', 'd_grd', 'd_conf')):
        a, b = A[i][l].double(), Bb[i][l].double()
        print(f'level {l} {name}: A vs B rel l2 {((a - b).norm() / b.norm()).item():.2e} max {((a - b).abs().max() / b.abs().max()).item():.2e}')
print('d_lam', A[3].cpu().numpy(), Bb[3].cpu().numpy())
