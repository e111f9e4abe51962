"""A few training steps of a non-default configuration (for `rocprofv3 --kernel-trace --stats`): finds kernels that are slow
outside the paths bench.py exercises.   python tools/probes/train_variants_rocprof.py {kitti_w|level4|g2sp_w|ford_w}"""
import sys, torch
sys.path.insert(0, '/root/repo')
from types import SimpleNamespace
from highlyaccurate_amd.models_kitti import LM_S2GP, LM_G2SP
from highlyaccurate_amd.models_ford import LM_S2GP_Ford
which = sys.argv[1] if len(sys.argv) > 1 else 'kitti_w'
d = torch.device('cuda:0')
kw = dict(level=3, N_iters=5, using_weight=1, loss_method=0, proj='geo', Optimizer='LM', rotation_range=10.0, shift_range_lat=20.0, shift_range_lon=20.0, damping=0.1, train_damping=1, dropout=0, use_hessian=0, use_gt_depth=0, visualize=0, coe_shift_lat=100.0, coe_shift_lon=100.0, coe_heading=100.0, coe_L1=100.0, coe_L2=100.0, coe_L3=100.0, coe_L4=100.0, estimate_depth=0, precision='bf16')
if which == 'level4': kw['level'] = 4
B = 16 if which == 'level4' else 32
args = SimpleNamespace(**kw)
net = {'kitti_w': LM_S2GP, 'level4': LM_S2GP, 'g2sp_w': LM_G2SP, 'ford_w': LM_S2GP_Ford}[which](args).to(d).train()
sat, grd = torch.rand(B, 3, 512, 512, device=d), torch.rand(B, 3, 256, 1024, device=d)
gt = [torch.rand(B, 1, device=d) * 2 - 1 for _ in range(3)]
extra = ()
if which == 'g2sp_w':
    extra = (torch.tensor([[[582.9802, 0., 496.2420], [0., 482.7076, 125.0034], [0., 0., 1.]]], device=d).repeat(B, 1, 1),)
if which == 'ford_w':
    extra = (112.64, torch.tensor([[[0., 0., 1.], [1., 0., 0.], [0., 1., 0.]]], device=d).repeat(B, 1, 1), torch.tensor([[1.7, 0.3, -1.2]], device=d).repeat(B, 1))
    gt = [g.reshape(-1).double() for g in gt]
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
import time
for i in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    r = net(sat, grd, *extra, *gt, mode='train')
    r[0].backward()
    opt.step()
    torch.cuda.synchronize()
    print(which, 'step', i, f'{(time.perf_counter() - t0) * 1e3:.1f} ms', flush=True)
