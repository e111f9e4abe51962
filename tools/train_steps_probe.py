import sys, time, torch
sys.path.insert(0, '/root/repo')
from types import SimpleNamespace
from highlyaccurate_amd.models_kitti import LM_S2GP
d = torch.device('cuda:0')
args = SimpleNamespace(level=3, N_iters=5, using_weight=0, loss_method=0, proj='geo', Optimizer='LM', rotation_range=10.0, shift_range_lat=20.0, shift_range_lon=20.0, damping=0.1, train_damping=0, dropout=0, use_hessian=0, use_gt_depth=0, visualize=0, coe_shift_lat=100.0, coe_shift_lon=100.0, coe_heading=100.0, coe_L1=100.0, coe_L2=100.0, coe_L3=100.0, coe_L4=100.0, estimate_depth=0, precision='bf16')
net = LM_S2GP(args).to(d).train()
B = 32
sat, grd = torch.rand(B, 3, 512, 512, device=d), torch.rand(B, 3, 256, 1024, device=d)
gt = [torch.rand(B, 1, device=d) * 2 - 1 for _ in range(3)]
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
# inference steps first, like bench.py
with torch.no_grad():
    for _ in range(5): net(sat, grd, mode='test')
torch.cuda.synchronize()
print('after inference loop: peak GB', torch.cuda.max_memory_allocated() / 2**30, 'current', torch.cuda.memory_allocated() / 2**30)
ts = []
for i in range(14):
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    r = net(sat, grd, gt[0], gt[1], gt[2], mode='train')
    r[0].backward()
    opt.step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
    if i < 3: print(f'after train step {i}: peak GB', torch.cuda.max_memory_allocated() / 2**30, 'current', torch.cuda.memory_allocated() / 2**30)
print(' '.join(f'{t:.1f}' for t in ts))
print('reserved GB', torch.cuda.memory_reserved() / 2**30, 'alloc retries', torch.cuda.memory_stats().get('num_alloc_retries'), 'segments', torch.cuda.memory_stats().get('segment.all.allocated'))
st = torch.cuda.memory_stats()
print('peak allocated GB', st['allocated_bytes.all.peak'] / 2**30, 'current', st['allocated_bytes.all.current'] / 2**30, 'inactive split GB', st['inactive_split_bytes.all.current'] / 2**30)
snap = torch.cuda.memory_snapshot()
big = sorted(((s['total_size'], s['allocated_size']) for s in snap), reverse=True)[:12]
print('largest segments (total, allocated) GB:', [(round(a / 2**30, 2), round(b / 2**30, 2)) for a, b in big])
