"""Diagnostic: full-map comparison of the HIP extractor's outputs with the fp64 oracle at full KITTI shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ref_cpu as O
from highlyaccurate_amd.VGG import VGGUnet

d = torch.device('cuda:0')
sd = O.synth_model_state(1)
sat, grd, *_ = O.synth_images(101, 1)
for name, img in (('Sat', sat), ('Grd', grd)):
    st = {k[len(name + 'FeatureNet.'):]: v for k, v in sd.items() if k.startswith(name + 'FeatureNet.')}
    on = O.VGGUnet(3); on.load_state_dict(st); on = on.double()
    with torch.no_grad():
        of, oc = on(img.double())
    net = VGGUnet(3); net.load_state_dict(st); net = net.to(d)
    with torch.no_grad():
        hf, hc = net(img.to(d))
    for l in range(3):
        a, b = hf[l].cpu().double().numpy(), of[l].numpy()
        e = np.abs(a - b)
        idx = np.unravel_index(e.argmax(), e.shape)
        print(f'{name} level {l}: max abs err {e.max():.2e} at {idx} (max |ref| {np.abs(b).max():.2e}); count > 1e-5*max: {(e > 1e-5 * np.abs(b).max()).sum()}')
        c = np.abs(hc[l].cpu().double().numpy() - oc[l].numpy())
        print(f'      conf max err {c.max():.2e}')
