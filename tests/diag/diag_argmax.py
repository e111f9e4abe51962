"""Compare the max-pool argmax maps saved by the HIP forward with torch's (fp64 oracle) on the same input."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn.functional as F
from oracle import ref_cpu as O
from highlyaccurate_amd.VGG import VGGUnet, vgg_forward_nhwc
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 256)
B = 2
d = torch.device('cuda:0')
rs = np.random.RandomState(31)
sd = O.synth_vgg_state(rs, bias_scale=0.05)
x = torch.from_numpy(rs.random_sample((B, 3, H, W)).astype(np.float32))
on = O.VGGUnet(3); on.load_state_dict(sd); on = on.double()
r = F.relu
with torch.no_grad():
    xd = x.double()
    x2 = on.conv2(r(on.conv0(xd))); x3 = F.max_pool2d(x2, 2)
    x7 = on.conv7(r(on.conv5(r(x3)))); x8 = F.max_pool2d(x7, 2)
    x14 = on.conv14(r(on.conv12(r(on.conv10(r(x8))))))
def argpos(t):
    Bq, C, h, w = t.shape
    v = t.reshape(Bq, C, h // 2, 2, w // 2, 2).permute(0, 2, 4, 1, 3, 5).reshape(Bq, h // 2, w // 2, C, 4)
    srt = v.sort(-1, descending=True).values
    return v.argmax(-1).numpy(), ((srt[..., 0] - srt[..., 1]) / srt[..., 0].abs().clamp_min(1e-30)).numpy()
net = VGGUnet(3, precision='fp32'); net.load_state_dict(sd); net = net.to(d)
feats, _, inv, ctx = vgg_forward_nhwc(net, x.to(d), want_conf=False, defer_norm=True, save_for_backward=True)
ws = ctx['ws'].cpu().numpy()
al = lambda n: (n + 255) // 256 * 256
P, es, o = B * H * W, 4, 0
sizes = [P//4*64*es, P//4*128*es, P//16*128*es, P//16*256*es, P//16*256*es, P//64*256*es, P//16*128*es, P//16*128*es, P//4*64*es, P//4*64*es]
for s in sizes: o += al(s)
tiles = lambda h, w: ((h + 7)//8) * ((w + 31)//32)
for n in (tiles(H//4, W//4)*2, tiles(H//4, W//4), tiles(H//2, W//2)): o += al(B*n*8)
o += al(3*B*8)
o_a0 = o; o += al(P*64*es)
for name, t, n, shp in (('idx3', x2, P//4*64, (B, H//2, W//2, 64)), ('idx8', x7, P//16*128, (B, H//4, W//4, 128)), ('idx15', x14, P//64*256, (B, H//8, W//8, 256))):
    idx = ws[o:o+n].reshape(shp); o += al(n)
    ref, gap = argpos(t)
    mism = idx != ref
    print(f'{name}: {mism.sum()} mismatches of {mism.size}; rel gap at mismatches: {np.sort(gap[mism])[:8]}; idx range {idx.min()}..{idx.max()}')
    if mism.sum():
        w = np.argwhere(mism)[:6]
        print('   first mismatches (b,y,x,c):', w.tolist(), 'hip', idx[mism][:6].tolist(), 'ref', ref[mism][:6].tolist())
