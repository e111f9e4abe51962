"""Which host-side op launches each device kernel of ONE training step (torch.profiler, CPU + device activities, with stacks).
    python tools/probes/step_ops.py [bf16|fp16x3] [B]
Prints, in launch order, every kernel that does not come from libhla (those have no aten op around them) with the aten op and the
innermost repo frame that issued it -- how the ~25 torch fill / copy launches between the LM loop and its backward were named."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from torch.profiler import profile, ProfilerActivity
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device('cuda:0')
net = bench.build_net('kitti', prec, 5, dev).train()
sat, grd, extra = bench.make_inputs('kitti', B, (256, 1024), 512, dev, 0)
gt = [torch.rand(B, 1, device=dev) * 2 - 1 for _ in range(3)]
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
def step():
    opt.zero_grad(set_to_none=True)
    r = net(sat, grd, gt[0], gt[1], gt[2], mode='train')
    r[0].backward()
    opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith('aten::')]
# leaf aten ops only (those that launch): keep ops with no aten child
evs.sort(key=lambda e: e.time_range.start)
seen = 0
for e in evs:
    kids = [k for k in e.cpu_children if k.name.startswith('aten::')]
    if kids:
        continue
    if not e.kernels:
        continue
    st = [s for s in (e.stack or []) if '/repo/' in s or 'bench.py' in s]
    print(f"{e.time_range.start - evs[0].time_range.start:9.0f} us  {e.name:<28s} {'; '.join(k.name[:40] for k in e.kernels)[:60]:<60s}  {st[0] if st else ''}")
    seen += 1
print('aten leaf ops that launched kernels:', seen)
