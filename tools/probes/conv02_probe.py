"""Where the TRAINING variant of the fused conv0 + conv2 kernel spends its extra time: the extractor's forward with
save_for_backward, timed per launch, under the library's timing-only switches.
    for a in 0 1 2 3; do HLA_ABL_C02=$a python tools/probes/conv02_probe.py fp16x3; done    # bit 0: no relu(conv0) copy, bit 1: no argmax"""
import os, sys, torch
sys.path.insert(0, '/root/repo')
from highlyaccurate_amd.VGG import VGGUnet, vgg_forward_nhwc
from highlyaccurate_amd import _lib
prec = sys.argv[1] if len(sys.argv) > 1 else 'fp16x3'
d = torch.device('cuda:0')
net = VGGUnet(3, precision=prec).to(d)
x = torch.rand(32, 3, 512, 512, device=d)
for train in (False, True):
    for _ in range(3):
        vgg_forward_nhwc(net, x, want_conf=False, defer_norm=True, save_for_backward=train)
    torch.cuda.synchronize()
    _lib.prof_enable(True); _lib.prof_fetch()
    for _ in range(4):
        vgg_forward_nhwc(net, x, want_conf=False, defer_norm=True, save_for_backward=train)
    recs = _lib.prof_fetch(); _lib.prof_enable(False)
    agg = {}
    for n, ms, fl, by in recs:
        a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += ms
    print(prec, 'abl', os.environ.get('HLA_ABL_C02', '0'), 'train' if train else 'infer', {n: round(v[1] / v[0] * 1e3, 1) for n, v in agg.items() if n.startswith('conv')})
