"""VGG backward alone at a mid-size non-square shape vs the fp64 oracle (multi-tile in x and y at every level)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ref_cpu as O
from highlyaccurate_amd.VGG import VGGUnet, vgg_forward_nhwc, vgg_backward_nhwc
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 256)
d = torch.device('cuda:0')
rs = np.random.RandomState(31)
sd = O.synth_vgg_state(rs, bias_scale=0.05)
x = torch.from_numpy(rs.random_sample((2, 3, H, W)).astype(np.float32))
onet = O.VGGUnet(3); onet.load_state_dict(sd); onet = onet.double()
feats64, _ = onet(x.double())
ups = [torch.from_numpy(rs.standard_normal(tuple(f.shape))) for f in feats64]
only = sys.argv[3] if len(sys.argv) > 3 else ''
if only:
    for i, u in enumerate(ups):
        if str(i) not in only: u.zero_()
sum((u * f).sum() for u, f in zip(ups, feats64)).backward()
ref = {k: p.grad for k, p in onet.named_parameters()}
net = VGGUnet(3, precision='fp32'); net.load_state_dict(sd); net = net.to(d)
feats, _, inv, ctx = vgg_forward_nhwc(net, x.to(d), want_conf=False, defer_norm=True, save_for_backward=True)
grads = vgg_backward_nhwc(net, ctx, [u.permute(0, 2, 3, 1).contiguous().float().to(d) for u in ups])
for k, g in grads.items():
    r = ref[k].numpy(); a = g.cpu().double().numpy()
    print(f'{k:24s} rel-l2 {np.linalg.norm(a-r)/max(np.linalg.norm(r),1e-30):.2e}')
