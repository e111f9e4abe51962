"""Which part of the GROUND maps receives gradient in a training step (bench configuration)?  (cf. dsat_rows.py)"""
import sys, torch
sys.path.insert(0, '/root/repo')
from highlyaccurate_amd import synthetic
from highlyaccurate_amd.models_kitti import LM_S2GP
d = torch.device('cuda:0')
net = LM_S2GP(synthetic.reference_args(precision='bf16'))
net.load_state_dict(synthetic.model_state(1))
net = net.to(d).train()
B = 32
sat, grd, gu, gv, gh = [t.to(d) for t in synthetic.images(3, B)]
orig = net.lm_backward
def spy(*a, **k):
    out = orig(*a, **k)
    for l, dg in enumerate(out[1]):
        nz = (dg != 0).any(dim=3)                     # [B,h,w]
        h, w = nz.shape[1:]
        u = nz.any(0)
        rows = u.any(1).nonzero().flatten()
        print(f'level {l} {h}x{w}: rows with gradient [{int(rows.min())}, {int(rows.max())}] (bottom half starts at {h // 2}); '
              f'texels touched {float(nz.float().mean()):.3f} (of the whole map); per-sample mean {float(nz.float().mean((1, 2)).mean()):.3f}')
        for r in range(h // 2, h, max(1, h // 16)):
            c = u[r].nonzero().flatten()
            print(f'   row {r}: ' + (f'cols [{int(c.min())}, {int(c.max())}] count {len(c)}' if len(c) else 'empty'))
    return out
net.lm_backward = spy
r = net(sat, grd, gu, gv, gh, mode='train')
r[0].backward()
