"""Diagnostic: HIP LM_G2SP end-to-end (full KITTI shape): d(loss)/d(normalised feature maps) vs the oracle's (autograd hooks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ref_cpu as O
from highlyaccurate_amd.models_kitti import LM_G2SP

d = torch.device('cuda:0')
seed, B = 1, 1
args = O.default_args(using_weight=1, train_damping=1)
sd = O.synth_model_state(seed); sd['damping'] = args.damping * torch.ones(1, 3)
on = O.LM_G2SP(args); on.load_state_dict(sd); on = on.double()
sat, grd, gu, gv, gh = O.synth_images(seed + 100, B)
K = torch.tensor([O.KITTI_K], dtype=torch.float32).repeat(B, 1, 1)
caps = {}
def hook(name):
    def f(mod, inp, out):
        for l, t in enumerate(out[0]):
            t.retain_grad(); caps[(name, 'f', l)] = t
        for l, t in enumerate(out[1][:3]):
            t.retain_grad(); caps[(name, 'c', l)] = t
    return f
on.SatFeatureNet.register_forward_hook(hook('sat')); on.GrdFeatureNet.register_forward_hook(hook('grd'))
res = on(sat.double(), grd.double(), K, gu.double(), gv.double(), gh.double(), mode='train')
res[0].backward()
net = LM_G2SP(args); net.load_state_dict(sd); net = net.to(d).train()
stash = {}
orig = net.lm_backward
def wrap(*a, **k):
    r = orig(*a, **k); stash['r'] = r; stash['d_trace'] = a[7]; return r
net.lm_backward = wrap
r = net(sat.to(d), grd.to(d), K.to(d), gu.to(d), gv.to(d), gh.to(d), mode='train')
r[0].backward()
print('loss', float(r[0]), float(res[0]))
lat, lon, th = on.trace
otr = torch.stack([lon, lat, th], -1)
print('trace err', (net.last_trace.cpu().double() - otr.detach()).abs().max().item())
d_sat, d_grd, d_conf, d_lam = stash['r']
for l in range(3):
    for name, got, ref in (('sat', d_sat[l].permute(0, 3, 1, 2), caps[('sat', 'f', l)].grad), ('grd', d_grd[l].permute(0, 3, 1, 2), caps[('grd', 'f', l)].grad),
                           ('conf', d_conf[l][:, None], caps[('grd', 'c', l)].grad)):
        got = got.cpu().double().numpy(); ref = ref.numpy()
        print(f'level {l} d_{name}: rel err max {np.abs(got - ref).max() / np.abs(ref).max():.2e}  l2 {np.linalg.norm(got - ref) / np.linalg.norm(ref):.2e}')
print('d_lam', d_lam.cpu().numpy(), on.damping.grad.numpy())
gt = torch.cat([gu, gv, gh], 1).double()                      # (u, v, theta)
od = (otr.detach() - gt[:, None, None, :])
hd = (net.last_trace.cpu().double() - gt[:, None, None, :])
print('min |pose - gt| oracle', od.abs().min().item(), ' sign mismatches', int((torch.sign(od) != torch.sign(hd)).sum()))
dt = stash['d_trace'].cpu().double()
exp = torch.sign(od) * 100.0 / (B * 15)
print('d_trace max dev from oracle sign pattern', (dt - exp).abs().max().item(), 'd_trace absmax', dt.abs().max().item())
# ---- the oracle's LM chain evaluated ON THE HIP FEATURE MAPS (ground truth for exactly the inputs the HIP backward saw)
from highlyaccurate_amd.VGG import vgg_forward_nhwc
with torch.no_grad():
    sfh, _, sinv = vgg_forward_nhwc(net.SatFeatureNet, sat.to(d), want_conf=False, defer_norm=True)
    gfh, gch, ginv = vgg_forward_nhwc(net.GrdFeatureNet, grd.to(d), want_conf=True, defer_norm=True)
tonchw = lambda f, inv: (f.double() * inv.view(B, 1, 1, 1)).permute(0, 3, 1, 2).cpu().contiguous().requires_grad_(True)
sfo = [tonchw(f, sinv[l]) for l, f in enumerate(sfh)]; gfo = [tonchw(f, ginv[l]) for l, f in enumerate(gfh)]
gco = [c.double()[:, None].cpu().contiguous().requires_grad_(True) for c in gch]
on.zero_grad()
su, sv, th = (torch.zeros(B, 1, dtype=torch.float64) for _ in range(3))
tr = []
for it in range(args.N_iters):
    for l in range(3):
        f, c, jac = on.project_grd_to_map(gfo[l], gco[l], su, sv, th, K, sfo[l].shape[-1], 256, 1024)
        su, sv, th = O.lm_update_g2s(args, on.damping, su, sv, th, f, c, sfo[l], jac, 1)
        tr.append(torch.cat([su, sv, th], 1))
tr = torch.stack(tr, 1).reshape(B, args.N_iters, 3, 3)
(tr * dt).sum().backward()
for l in range(3):
    for name, got, ref in (('sat', d_sat[l].permute(0, 3, 1, 2), sfo[l].grad), ('grd', d_grd[l].permute(0, 3, 1, 2), gfo[l].grad)):
        got = got.cpu().double().numpy(); ref = ref.numpy()
        print(f'[oracle on HIP maps] level {l} d_{name}: rel err max {np.abs(got - ref).max() / np.abs(ref).max():.2e}  l2 {np.linalg.norm(got - ref) / np.linalg.norm(ref):.2e}')
print('[oracle on HIP maps] d_lam', on.damping.grad.numpy())
