"""Diagnostic: where do the 1e-3-level G2S full-shape gradient deviations come from?  Feed the ORACLE's exact d(loss)/d(feature
map) into (a) the HIP VGG backward and (b) torch-CPU fp32 autograd of the oracle extractor, compare both with fp64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ref_cpu as O
from highlyaccurate_amd.VGG import VGGUnet, vgg_forward_nhwc, vgg_backward_nhwc

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
            t.retain_grad(); caps[(name, l)] = t
    return f
on.SatFeatureNet.register_forward_hook(hook('sat'))
res = on(sat.double(), grd.double(), K, gu.double(), gv.double(), gh.double(), mode='train')
res[0].backward()
ref = {k: p.grad for k, p in on.named_parameters()}
ups = [caps[('sat', l)].grad for l in range(3)]
# (a) HIP
net = VGGUnet(3); net.load_state_dict({k[len('SatFeatureNet.'):]: v for k, v in sd.items() if k.startswith('SatFeatureNet.')}); net = net.to(d)
feats, _, inv, ctx = vgg_forward_nhwc(net, sat.to(d), want_conf=False, defer_norm=True, save_for_backward=True)
gh_ = vgg_backward_nhwc(net, ctx, [u.float().permute(0, 2, 3, 1).contiguous().to(d) for u in ups])
# (b) torch fp32 CPU
o32 = O.VGGUnet(3); o32.load_state_dict({k[len('SatFeatureNet.'):]: v for k, v in sd.items() if k.startswith('SatFeatureNet.')})
f32, _ = o32(sat)
sum((u.float() * f).sum() for u, f in zip(ups, f32)).backward()
g32 = {k: p.grad for k, p in o32.named_parameters()}
for k in ('conv_dec2.3.weight', 'conv_dec2.1.weight', 'conv_dec1.3.weight', 'conv14.weight', 'conv0.weight'):
    r = ref['SatFeatureNet.' + k].numpy(); s = np.abs(r).max()
    print(f'{k:22s} hip-vs-64 {np.abs(gh_[k].cpu().double().numpy() - r).max() / s:.2e}   torch32-vs-64 {np.abs(g32[k].double().numpy() - r).max() / s:.2e}   scale {s:.2e}')
