"""Training step with the two extractors' backward passes on two streams (args.bwd_two_streams = 1, the default) against one stream.
    python tools/probes/train_streams_ab.py [fp16x3|bf16]"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from types import SimpleNamespace
from highlyaccurate_amd.models_kitti import LM_S2GP
prec = sys.argv[1] if len(sys.argv) > 1 else 'fp16x3'
d = torch.device('cuda:0')
args = SimpleNamespace(level=3, N_iters=5, using_weight=0, loss_method=0, proj='geo', Optimizer='LM', rotation_range=10.0, shift_range_lat=20.0, shift_range_lon=20.0, damping=0.1, train_damping=0, dropout=0, use_hessian=0, use_gt_depth=0, visualize=0, coe_shift_lat=100.0, coe_shift_lon=100.0, coe_heading=100.0, coe_L1=100.0, coe_L2=100.0, coe_L3=100.0, coe_L4=100.0, estimate_depth=0, precision=prec)
net = LM_S2GP(args).to(d).train()
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
B = 32
sat, grd = torch.rand(B, 3, 512, 512, device=d), torch.rand(B, 3, 256, 1024, device=d)
gt = [torch.rand(B, 1, device=d) * 2 - 1 for _ in range(3)]
def step():
    opt.zero_grad(set_to_none=True)
    r = net(sat, grd, gt[0], gt[1], gt[2], mode='train')
    r[0].backward()
    opt.step()
for two in (1, 0, 1, 0):
    args.bwd_two_streams = two
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): step()
    torch.cuda.synchronize()
    print(prec, 'two_streams', two, round((time.perf_counter() - t0) / 8 * 1e3, 3), 'ms/step')
