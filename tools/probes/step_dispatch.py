"""Every aten op a training step dispatches between the LM loop and the extractors' backward, with the repo frame that issued it
(TorchDispatchMode: sees the ops autograd and custom Functions issue as well).
    python tools/probes/step_dispatch.py [bf16|fp16x3] [B]"""
import sys, os, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from torch.utils._python_dispatch import TorchDispatchMode
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
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
for _ in range(2):
    step()
log = []
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        fr = [f for f in traceback.extract_stack() if ('/highlyaccurate_amd/' in f.filename or f.filename.endswith('bench.py') or 'step_dispatch' in f.filename)]
        where = f'{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}' if fr else '(autograd engine / torch)'
        log.append((str(func), where))
        return func(*args, **(kwargs or {}))
with Log():
    opt.zero_grad(set_to_none=True)
    r = net(sat, grd, gt[0], gt[1], gt[2], mode='train')
    r[0].backward()
torch.cuda.synchronize()
skip = ('aten.view', 'aten.slice', 'aten.select', 'aten.detach', 'aten.empty', 'aten.alias', 'aten._unsafe_view', 'aten.unsqueeze', 'aten.as_strided',
        'aten.permute', 'aten.t.', 'aten.expand', 'aten.reshape', 'aten.is_pinned', 'aten.lift_fresh')
for f, w in log:
    if not any(f.startswith(k) for k in skip):
        print(f'{f:<40s} {w}')
print(len(log), 'ops dispatched in forward + backward')
