"""Experiment: capture LM_S2GP.forward(mode='test') in a HIP graph (torch.cuda.CUDAGraph) and compare replay vs eager."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from highlyaccurate_amd.models_kitti import LM_S2GP

d = torch.device('cuda:0')
B = int(os.environ.get("GRAPH_B", "32"))
args = SimpleNamespace(level=3, N_iters=5, using_weight=0, loss_method=0, proj='geo', Optimizer='LM', rotation_range=10.0,
                       shift_range_lat=20.0, shift_range_lon=20.0, damping=0.1, train_damping=0, dropout=0, use_hessian=0,
                       use_gt_depth=0, visualize=0, coe_shift_lat=100.0, coe_shift_lon=100.0, coe_heading=100.0, coe_L1=100.0,
                       coe_L2=100.0, coe_L3=100.0, coe_L4=100.0, estimate_depth=0, precision='bf16')
net = LM_S2GP(args).to(d).eval()
sat, grd = torch.rand(B, 3, 512, 512, device=d), torch.rand(B, 3, 256, 1024, device=d)
fixed = net._draw_reinit(15, B, d)
net._draw_reinit = lambda n, b, dev: fixed        # static device buffer instead of CPU draws + H2D inside the capture


def step():
    with torch.no_grad():
        return net(sat, grd, mode='test')


for _ in range(3):
    out = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    out = step()
torch.cuda.synchronize()
te = (time.perf_counter() - t0) / 20
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    gout = step()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
tg = (time.perf_counter() - t0) / 20
print(f'eager {te * 1e3:.3f} ms/step  graph replay {tg * 1e3:.3f} ms/step  ({B / te:.0f} vs {B / tg:.0f} pairs/s); '
      f'outputs equal: {all(torch.equal(a, b) for a, b in zip(out, gout))}')
