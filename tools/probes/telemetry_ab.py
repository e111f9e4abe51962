"""Does the telemetry side thread (hwmon sysfs reads at 100 Hz) perturb what it measures?  Same-process A/B:
inference (B = 32, bf16) and the training step (fp16x3), alternating without / with the sampler.
    gpurun -- 'python tools/probes/telemetry_ab.py'"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from types import SimpleNamespace
from highlyaccurate_amd.models_kitti import LM_S2GP
d = torch.device('cuda:0')
def mk(prec):
    a = SimpleNamespace(level=3, N_iters=5, using_weight=0, loss_method=0, proj='geo', Optimizer='LM', rotation_range=10.0, shift_range_lat=20.0, shift_range_lon=20.0, damping=0.1, train_damping=0, dropout=0, use_hessian=0, use_gt_depth=0, visualize=0, coe_shift_lat=100.0, coe_shift_lon=100.0, coe_heading=100.0, coe_L1=100.0, coe_L2=100.0, coe_L3=100.0, coe_L4=100.0, estimate_depth=0, precision=prec)
    return LM_S2GP(a).to(d)
B = 32
sat, grd = torch.rand(B, 3, 512, 512, device=d), torch.rand(B, 3, 256, 1024, device=d)
net = mk('bf16').eval()
for rep in range(3):
    for on in (0, 1):
        tele = bench.Telemetry(0) if on else None
        dt, _ = bench.timed_infer(net, sat, grd, (), 50, 5, None, tele)
        print(f'inference bf16  telemetry={on}  {B * 50 / dt:8.1f} pairs/s  {dt / 50 * 1e3:.3f} ms', (tele.summary().get('samples') if tele else ''), flush=True)
del net
net = mk('fp16x3').train()
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
gt = [torch.rand(B, 1, device=d) * 2 - 1 for _ in range(3)]
def tstep():
    opt.zero_grad(set_to_none=True)
    r = net(sat, grd, gt[0], gt[1], gt[2], mode='train')
    r[0].backward(); opt.step()
for _ in range(4): tstep()
torch.cuda.synchronize()
for rep in range(3):
    for on in (0, 1, 2):
        tele = bench.Telemetry(0, hz=100.0 if on == 1 else 10.0) if on else None
        if tele: tele.__enter__()
        t0 = time.perf_counter()
        for _ in range(6): tstep()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if tele: tele.__exit__()
        print(f'train fp16x3  telemetry={("off","100Hz","10Hz")[on]}  {dt / 6 * 1e3:.3f} ms per step', (tele.summary().get('samples') if tele else ''), flush=True)
