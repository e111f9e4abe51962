"""Inference step with the two extractors on one stream (default) against two streams with / without a priority for the satellite chain.
    python tools/probes/fwd_streams_ab.py [bf16|fp16x3]"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from types import SimpleNamespace
from highlyaccurate_amd.models_kitti import LM_S2GP
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
d = torch.device('cuda:0')
print('priority range', torch.cuda.Stream.priority_range())
args = SimpleNamespace(level=3, N_iters=5, using_weight=0, loss_method=0, proj='geo', Optimizer='LM', rotation_range=10.0, shift_range_lat=20.0, shift_range_lon=20.0, damping=0.1, train_damping=0, dropout=0, use_hessian=0, use_gt_depth=0, visualize=0, coe_shift_lat=100.0, coe_shift_lon=100.0, coe_heading=100.0, coe_L1=100.0, coe_L2=100.0, coe_L3=100.0, coe_L4=100.0, estimate_depth=0, precision=prec)
net = LM_S2GP(args).to(d).eval()
B = 32
sat, grd = torch.rand(B, 3, 512, 512, device=d), torch.rand(B, 3, 256, 1024, device=d)
lo, hi = torch.cuda.Stream.priority_range()
for rep in range(2):
    for mode in (None, 0, hi, lo):
        if mode is None:
            if hasattr(args, 'fwd_two_streams'): del args.fwd_two_streams
        else:
            args.fwd_two_streams = mode
        with torch.no_grad():
            for _ in range(8): net(sat, grd, mode='test')
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(40): net(sat, grd, mode='test')
            torch.cuda.synchronize()
        print(prec, 'fwd_two_streams', mode, round(B * 40 / (time.perf_counter() - t0), 1), 'pairs/s', flush=True)
