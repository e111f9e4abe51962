"""Per-launch conv timing of one inference step, in launch order (HLA_LIB selects an experiment build).
    python tools/probes/infer_launches.py [hires]      # hires: BASELINE configs[4] (1024^2 / 512x2048, fp16, 10 iterations, B = 8)"""
import sys, torch
sys.path.insert(0, '/root/repo')
from types import SimpleNamespace
from highlyaccurate_amd.models_kitti import LM_S2GP
from highlyaccurate_amd import _lib
d = torch.device('cuda:0')
args = SimpleNamespace(level=3, N_iters=5, using_weight=0, loss_method=0, proj='geo', Optimizer='LM', rotation_range=10.0, shift_range_lat=20.0, shift_range_lon=20.0, damping=0.1, train_damping=0, dropout=0, use_hessian=0, use_gt_depth=0, visualize=0, coe_shift_lat=100.0, coe_shift_lon=100.0, coe_heading=100.0, coe_L1=100.0, coe_L2=100.0, coe_L3=100.0, coe_L4=100.0, estimate_depth=0, precision='bf16')
hires = len(sys.argv) > 1 and sys.argv[1] == 'hires'
if hires:
    args.precision, args.N_iters = 'fp16', 10
net = LM_S2GP(args).to(d).eval()
B = 8 if hires else 32
sat, grd = (torch.rand(B, 3, 1024, 1024, device=d), torch.rand(B, 3, 512, 2048, device=d)) if hires else \
           (torch.rand(B, 3, 512, 512, device=d), torch.rand(B, 3, 256, 1024, device=d))
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
agg = {}
for nm, ms, fl, by in recs:
    a = agg.setdefault(nm, [0, 0.0]); a[0] += 1; a[1] += ms
for nm, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]): print(f'{nm:34s} x{c // 4:3d} {ms / 4:7.3f} ms/step')
