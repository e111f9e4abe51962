"""Same-process, same-box A/B of a training step under different args settings, alternating, medians.
    python tools/probes/train_ab.py bf16 bwd_prefill=0 bwd_prefill=1 bwd_prefill=24 [steps=8] [rounds=4]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
prec = sys.argv[1]
sets = [a for a in sys.argv[2:] if '=' in a and not a.startswith(('steps=', 'rounds='))]
kv = dict(a.split('=') for a in sys.argv[2:] if a.startswith(('steps=', 'rounds=')))
steps, rounds = int(kv.get('steps', 8)), int(kv.get('rounds', 4))
dev = torch.device('cuda:0')
B = 32
net = bench.build_net('kitti', prec, 5, dev).train()
sat, grd, extra = bench.make_inputs('kitti', B, (256, 1024), 512, dev, 0)
gt = [torch.rand(B, 1, device=dev) * 2 - 1 for _ in range(3)]
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
def step():
    opt.zero_grad(set_to_none=True)
    r = net(sat, grd, gt[0], gt[1], gt[2], mode='train')
    r[0].backward()
    opt.step()
def apply(setting):
    for part in setting.split(','):
        k, v = part.split('=')
        setattr(net.args, k, int(v))
for _ in range(6):
    step()
res = {s: [] for s in sets}
for r in range(rounds):
    for s in sets:
        apply(s)
        step(); step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        res[s].append((time.perf_counter() - t0) / steps * 1e3)
for s in sets:
    v = sorted(res[s])
    print(f'{prec} {s:<40s} median {v[len(v) // 2]:.3f} ms  all {[round(x, 3) for x in res[s]]}  -> {B / v[len(v) // 2] * 1e3:.1f} pairs/s')
