"""Which rows / columns of the satellite maps receive any gradient in a training step (bench configuration)?"""
import sys, torch
sys.path.insert(0, '/root/repo')
from highlyaccurate_amd import synthetic
from highlyaccurate_amd.models_kitti import LM_S2GP
d = torch.device('cuda:0')
net = LM_S2GP(synthetic.reference_args(precision='bf16')).to(d).train()
for m in net.modules():
    if isinstance(m, torch.nn.Conv2d):
        torch.nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        if m.bias is not None: torch.nn.init.zeros_(m.bias)
B = 32
torch.manual_seed(1234)
sat, grd = torch.rand(B, 3, 512, 512, device=d), torch.rand(B, 3, 256, 1024, device=d)
gt = [torch.rand(B, 1, device=d) * 2 - 1 for _ in range(3)]
orig = net.lm_backward
def spy(*a, **k):
    out = orig(*a, **k)
    for l, ds in enumerate(out[0]):
        nz = (ds != 0).any(dim=3)                     # [B,A,A]
        rows = nz.any(dim=2); cols = nz.any(dim=1)    # [B,A]
        A = ds.shape[1]
        r_any = rows.any(0).nonzero().flatten(); c_any = cols.any(0).nonzero().flatten()
        per = [(int(rows[b].nonzero().min()), int(rows[b].nonzero().max())) for b in range(B)]
        print(f'level {l} A={A}: rows with gradient over the batch [{int(r_any.min())}, {int(r_any.max())}], cols [{int(c_any.min())}, {int(c_any.max())}]; '
              f'per-sample last row: min {min(p[1] for p in per)} max {max(p[1] for p in per)}; texels touched {float(nz.float().mean()):.3f}')
        # live 8x32 tiles of this map under three descriptions of the batch-union support, after a 3-px dilation
        u = nz.any(0).float()[None, None]
        u = torch.nn.functional.max_pool2d(u, 7, 1, 3)[0, 0] > 0
        t = u.view(A // 8, 8, A // 32, 32).any(3).any(1)                 # exact tile bitmap [A/8, A/32]
        ys, xs = u.any(1).nonzero().flatten(), u.any(0).nonzero().flatten()
        box = ((int(ys.max()) // 8 - int(ys.min()) // 8 + 1) * (int(xs.max()) // 32 - int(xs.min()) // 32 + 1))
        band = 0
        for ty in range(A // 8):
            c = u[ty * 8:ty * 8 + 8].any(0).nonzero().flatten()
            if len(c): band += int(c.max()) // 32 - int(c.min()) // 32 + 1
        print(f'   tiles 8x32: all {t.numel()}, box {box}, per-band interval {band}, exact bitmap {int(t.sum())}')
    return out
net.lm_backward = spy
r = net(sat, grd, gt[0], gt[1], gt[2], mode='train')
r[0].backward()
print('trace final poses range', float(net.last_trace.abs().max()))
