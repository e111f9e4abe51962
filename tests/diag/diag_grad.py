"""Full-shape B=1 training step: HIP gradients vs the fp64 CPU oracle's autograd, every parameter tensor."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ref_cpu as O
from highlyaccurate_amd.models_kitti import LM_S2GP
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.set_num_threads(16)
args = O.default_args()
sat, grd, gu, gv, gh = O.synth_images(seed + 100, 1)
on = O.build('kitti', args, seed, dtype=torch.float64)
torch.manual_seed(seed)
t0 = time.time()
res = on(sat.double(), grd.double(), gu.double(), gv.double(), gh.double(), mode='train')
res[0].backward()
print('oracle fwd+bwd %.1fs loss %.8f' % (time.time() - t0, float(res[0])))
d = torch.device('cuda:0')
net = LM_S2GP(args); net.load_state_dict(O.synth_model_state(seed)); net = net.to(d).train()
torch.manual_seed(seed)
r = net(sat.to(d), grd.to(d), gu.to(d), gv.to(d), gh.to(d), mode='train')
r[0].backward()
print('hip loss %.8f' % float(r[0]))
ref = dict(on.named_parameters())
for k, p in net.named_parameters():
    if p.grad is None: continue
    a, b = p.grad.double().cpu().numpy(), ref[k].grad.numpy()
    print(f'{k:34s} rel-l2 {np.linalg.norm(a-b)/np.linalg.norm(b):.2e}  max {np.abs(a-b).max()/np.abs(b).max():.2e}')
