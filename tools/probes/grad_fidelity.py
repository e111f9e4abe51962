"""Per-tensor gradient fidelity of one training step against the reference's fp64 autograd (tests/golden/train_kitti.npz), by
arithmetic mode:   gpurun -- 'python tools/probes/grad_fidelity.py bf16 fp16 fp16x3'"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from highlyaccurate_amd import synthetic

dev = torch.device('cuda:0')
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'train_kitti.npz'), allow_pickle=False)
seed, B = int(g['seed']), int(g['B'])
keys = [k[len('grad64_'):] for k in g.files if k.startswith('grad64_')]
for prec in sys.argv[1:] or ['bf16', 'fp16', 'fp16x3']:
    extra = {}
    if ':' in prec:            # e.g. bf16:bwd_f16=1
        prec, kv = prec.split(':', 1)
        extra = {k: int(v) for k, v in (e.split('=') for e in kv.split(','))}
    net = bench.build_net('kitti', prec, 5, dev, state=synthetic.model_state(seed)).train()
    for k, v in extra.items():
        setattr(net.args, k, v)
    sat, grd, gu, gv, gh = synthetic.images(seed + 100, B)
    torch.manual_seed(seed)
    r = net(sat.to(dev), grd.to(dev), gu.to(dev), gv.to(dev), gh.to(dev), mode='train')
    r[0].backward()
    named = dict(net.named_parameters())
    print(f'== {prec} {extra}: loss rel err {abs(float(r[0].detach()) - float(g["tuple64"][0][0])) / abs(float(g["tuple64"][0][0])):.2e}')
    for k in keys:
        ref, r32 = g['grad64_' + k][2:], g['grad32_' + k][2:]
        gr = named[k].grad.double().reshape(-1).cpu().numpy()
        got = gr[synthetic.fixture_sample_idx(gr.size, 77)]
        l2 = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        cos = np.dot(got, ref) / (np.linalg.norm(got) * np.linalg.norm(ref))
        l232 = np.linalg.norm(r32 - ref) / np.linalg.norm(ref)
        print(f'   {k:40s} rel-L2 {l2:.3e}  cos {cos:.6f}  |ref| {np.linalg.norm(ref):.2e}   (reference fp32 vs fp64: {l232:.1e})')
