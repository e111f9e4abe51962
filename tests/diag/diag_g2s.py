"""Diagnostic: LM_G2SP LM-loop backward at FULL KITTI shape, HIP vs oracle autograd, on the oracle's own feature maps."""
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
with torch.no_grad():
    sf, _ = on.SatFeatureNet(sat.double()); gf, gc = on.GrdFeatureNet(grd.double())
sf = [f.detach().requires_grad_(True) for f in sf]; gf = [f.detach().requires_grad_(True) for f in gf]
gc = [c.detach().requires_grad_(True) for c in gc[:3]]
su, sv, th = (torch.zeros(B, 1, dtype=torch.float64) for _ in range(3))
us, vs, ts = [], [], []
for it in range(args.N_iters):
    for l in range(3):
        f, c, jac = on.project_grd_to_map(gf[l], gc[l], su, sv, th, K, sf[l].shape[-1], 256, 1024)
        su, sv, th = O.lm_update_g2s(args, on.damping, su, sv, th, f, c, sf[l], jac, 1)
        us.append(su); vs.append(sv); ts.append(th)
tr = torch.stack([torch.cat([u, v, t], 1) for u, v, t in zip(us, vs, ts)], 1).reshape(B, args.N_iters, 3, 3)
loss = 100 * ((tr[..., 0] - gu.double()[:, :, None]).abs().mean(0) + (tr[..., 1] - gv.double()[:, :, None]).abs().mean(0)
              + (tr[..., 2] - gh.double()[:, :, None]).abs().mean(0)).mean()
loss.backward()
net = LM_G2SP(args); net.load_state_dict(sd); net = net.to(d)
nh = lambda t: t.detach().float().permute(0, 2, 3, 1).contiguous().to(d)
feats = ([nh(s) for s in sf], [nh(g) for g in gf], [c.detach().float()[:, 0].contiguous().to(d) for c in gc])
trace = net.lm_solve(*feats, K.to(d), (256, 1024), keep_normal_eq=True)
print('trace err', (trace.cpu().double() - tr.detach()).abs().max().item())
trg = trace.detach().clone().requires_grad_(True)
l2 = 100 * ((trg[..., 0] - gu.to(d)[:, :, None]).abs().mean(0) + (trg[..., 1] - gv.to(d)[:, :, None]).abs().mean(0)
            + (trg[..., 2] - gh.to(d)[:, :, None]).abs().mean(0)).mean()
l2.backward()
d_sat, d_grd, d_conf, d_lam = net.lm_backward(*feats, K.to(d), (256, 1024), trace, net.last_normal_eq, trg.grad)
for l in range(3):
    for name, got, ref in (('sat', d_sat[l].permute(0, 3, 1, 2), sf[l].grad), ('grd', d_grd[l].permute(0, 3, 1, 2), gf[l].grad),
                           ('conf', d_conf[l][:, None], gc[l].grad)):
        got = got.cpu().double().numpy(); ref = ref.numpy()
        print(f'level {l} d_{name}: rel err max {np.abs(got - ref).max() / np.abs(ref).max():.2e}  l2 {np.linalg.norm(got - ref) / np.linalg.norm(ref):.2e}')
print('d_lam', d_lam.cpu().numpy(), on.damping.grad.numpy())
