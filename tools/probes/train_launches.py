"""Per-launch kernel timing of one training step (forward + backward), in launch order.
    python tools/probes/train_launches.py [bf16|fp16|fp32|fp16x3] [two_streams: 1|0]"""
import sys, torch
sys.path.insert(0, '/root/repo')
from types import SimpleNamespace
from highlyaccurate_amd.models_kitti import LM_S2GP
from highlyaccurate_amd import _lib
d = torch.device('cuda:0')
args = SimpleNamespace(level=3, N_iters=5, using_weight=0, loss_method=0, proj='geo', Optimizer='LM', rotation_range=10.0, shift_range_lat=20.0, shift_range_lon=20.0, damping=0.1, train_damping=0, dropout=0, use_hessian=0, use_gt_depth=0, visualize=0, coe_shift_lat=100.0, coe_shift_lon=100.0, coe_heading=100.0, coe_L1=100.0, coe_L2=100.0, coe_L3=100.0, coe_L4=100.0, estimate_depth=0, precision=(sys.argv[1] if len(sys.argv) > 1 else 'bf16'))
args.bwd_two_streams = int(sys.argv[2]) if len(sys.argv) > 2 else 1      # 0: one stream -> every launch owns the chip (exclusive timings)
net = LM_S2GP(args).to(d).train()
B = 32
sat, grd = torch.rand(B, 3, 512, 512, device=d), torch.rand(B, 3, 256, 1024, device=d)
gt = [torch.rand(B, 1, device=d) * 2 - 1 for _ in range(3)]
def step():
    net.zero_grad(set_to_none=True)
    r = net(sat, grd, gt[0], gt[1], gt[2], mode='train')
    r[0].backward()
    torch.cuda.synchronize()
for _ in range(3): step()
_lib.prof_enable(True); _lib.prof_fetch()
step()
recs = _lib.prof_fetch()
_lib.prof_enable(False)
tot = sum(r[1] for r in recs)
agg = {}
for n, ms, fl, by in recs:
    if ms > 0.1 or n.startswith('conv') or n.startswith('wgrad') or n.startswith('lm_bwd'):
        print(f'{n:34s} {ms*1e3:8.1f} us  {fl/ms/1e9 if fl else 0:7.1f} TF  {by/ms/1e6 if by else 0:7.1f} GB/s')
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += ms
print('total kernel ms', tot)
for n, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]): print(f'{n:34s} x{c:3d} {ms:7.3f} ms {ms/tot*100:5.1f} %')
