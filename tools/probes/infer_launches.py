"""Per-launch conv timing of one inference step, in launch order (HLA_LIB selects an experiment build)."""
import sys, torch
sys.path.insert(0, '/root/repo')
from types import SimpleNamespace
from highlyaccurate_amd.models_kitti import LM_S2GP
from highlyaccurate_amd import _lib
d = torch.device('cuda:0')
args = SimpleNamespace(level=3, N_iters=5, using_weight=0, loss_method=0, proj='geo', Optimizer='LM', rotation_range=10.0, shift_range_lat=20.0, shift_range_lon=20.0, damping=0.1, train_damping=0, dropout=0, use_hessian=0, use_gt_depth=0, visualize=0, coe_shift_lat=100.0, coe_shift_lon=100.0, coe_heading=100.0, coe_L1=100.0, coe_L2=100.0, coe_L3=100.0, coe_L4=100.0, estimate_depth=0, precision='bf16')
net = LM_S2GP(args).to(d).eval()
B = 32
sat, grd = torch.rand(B, 3, 512, 512, device=d), torch.rand(B, 3, 256, 1024, device=d)
with torch.no_grad():
    for _ in range(5): net(sat, grd, mode='test')
    torch.cuda.synchronize()
    _lib.prof_enable(True); _lib.prof_fetch()
    for _ in range(4): net(sat, grd, mode='test')
    recs = _lib.prof_fetch()
_lib.prof_enable(False)
n = len(recs) // 4
for k in range(n):
    nm, ms, fl, by = recs[k]
    if not nm.startswith('conv'): continue
    avg = sum(recs[k + i * n][1] for i in range(4)) / 4
    print(f'{nm:34s} {avg*1e3:8.1f} us  {fl/avg/1e9:7.1f} TF')
print('total ms/step', sum(r[1] for r in recs) / 4)
