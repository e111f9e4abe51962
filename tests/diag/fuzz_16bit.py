"""Randomised check of the 16-bit convolution kernels' OWN code paths (LDS-DMA halo loader with its source-side swizzle and
range-checked zero border, bias-initialised accumulators, conv0's bias inside the matrix product, the batched epilogue, and in the
backward the same loader under the data-dependent tile lists) against the library's exact-fp32 mode, which is itself gated against
the reference: random batch / image sizes (multiples of 8, mostly not of the 8 x 32 tiles), levels 3 and 4, forward maps and the
parameter gradients of a random linear functional of them.  A boundary or addressing bug shows up as O(1), rounding as 1e-2.

    python tests/diag/fuzz_16bit.py [n_cases] [first_seed]
"""
import os, sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ref_cpu as O
from highlyaccurate_amd.VGG import VGGUnet

d = torch.device('cuda:0')
LIM = {'bf16': (4e-2, 0.97), 'fp16': (6e-3, 0.995)}          # max relative map error, minimum gradient cosine


def one_case(seed):
    rs = np.random.RandomState(seed)
    B = int(rs.randint(1, 5))
    H, W = int(rs.randint(1, 25)) * 8, int(rs.randint(1, 41)) * 8
    level = int(rs.choice([3, 3, 4]))
    sd = O.synth_vgg_state(rs, bias_scale=0.05)
    x = torch.from_numpy(rs.random_sample((B, 3, H, W)).astype(np.float32)).to(d)
    sparse = rs.randint(3) == 0          # a gradient that lives in a corner of the maps: the backward's tile lists get short
    res = {}
    for prec in ('fp32', 'bf16', 'fp16'):
        net = VGGUnet(level, precision=prec)
        net.load_state_dict(sd)
        net = net.to(d)
        feats, confs = net(x)
        us = []
        g = torch.Generator().manual_seed(seed)
        for f in feats:
            u = torch.randn(tuple(f.shape), generator=g)
            if sparse:
                m = torch.zeros_like(u)
                m[..., : max(1, u.shape[-2] // 3), u.shape[-1] // 2:] = 1
                u = u * m
            us.append(u.to(d))
        (sum((u * f).sum() for u, f in zip(us, feats)) + sum(c.sum() for c in confs)).backward()
        res[prec] = ([f.detach().float().cpu().numpy() for f in feats], {k: p.grad.detach().double().cpu().flatten() for k, p in net.named_parameters() if p.grad is not None})
    ok, notes = True, []
    for prec in ('bf16', 'fp16'):
        tol, cmin = LIM[prec]
        e = max(float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)) for a, b in zip(res[prec][0], res['fp32'][0]))
        fin = all(np.isfinite(a).all() for a in res[prec][0])
        worst, wn = 1.0, ''
        for k, gr in res['fp32'][1].items():
            if float(gr.norm()) < 1e-12:
                continue
            a = res[prec][1][k]
            c = float((a @ gr) / (a.norm() * gr.norm() + 1e-300))
            if c < worst:
                worst, wn = c, k
        good = fin and e < tol and worst > cmin
        ok = ok and good
        notes.append(f'{prec}: map err {e:.2e} grad cos {worst:.5f} ({wn})' + ('' if good else '  <-- FAIL'))
    print(f"{'ok  ' if ok else 'FAIL'} seed {seed} B{B} {H}x{W} level {level}{' sparse' if sparse else ''}: " + '; '.join(notes), flush=True)
    return ok


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = [s for s in range(s0, s0 + n) if not one_case(s)]
    print('failed seeds:', bad)
    sys.exit(1 if bad else 0)
