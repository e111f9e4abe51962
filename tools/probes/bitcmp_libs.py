"""Are two builds of libhla bit-identical on a split-mode forward + training step?  (e.g. -DHLA_SPLIT4_ASM=0 against the default)
    python tools/probes/bitcmp_libs.py libhla_a.so libhla_b.so      # each library runs in its own process; outputs compared with ==
Inputs include exact zeros, denormal-range values and large magnitudes in the images and weights (the split's corner cases)."""
import os, subprocess, sys, tempfile
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WORKER = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from types import SimpleNamespace
from highlyaccurate_amd.models_kitti import LM_S2GP
torch.manual_seed(7); np.random.seed(7)
d = torch.device('cuda:0')
args = SimpleNamespace(level=3, N_iters=2, using_weight=0, loss_method=0, proj='geo', Optimizer='LM', rotation_range=10.0, shift_range_lat=20.0, shift_range_lon=20.0, damping=0.1, train_damping=0, dropout=0, use_hessian=0, use_gt_depth=0, visualize=0, coe_shift_lat=100.0, coe_shift_lon=100.0, coe_heading=100.0, coe_L1=100.0, coe_L2=100.0, coe_L3=100.0, coe_L4=100.0, estimate_depth=0, precision='fp16x3', bwd_two_streams=0)
net = LM_S2GP(args).to(d).train()
B = 3
sat, grd = torch.rand(B, 3, 128, 128, device=d), torch.rand(B, 3, 64, 256, device=d)
sat[0, :, :40] = 0; grd[1] *= 1e-30; sat[2] *= 3e3; grd[0, :, 10:20, 30:90] = 1e-41      # zeros, tiny, large, denormal inputs
gt = [torch.rand(B, 1, device=d) * 2 - 1 for _ in range(3)]
r = net(sat, grd, gt[0], gt[1], gt[2], mode='train')
r[0].backward()
out = {'loss': r[0].detach().cpu().numpy(), 'trace': net.last_trace.cpu().numpy()}
for n, p in net.named_parameters():
    if p.grad is not None: out['g_' + n] = p.grad.detach().cpu().numpy()
np.savez(sys.argv[1], **out)
'''
res = []
for lib in sys.argv[1:3]:
    f = tempfile.mktemp(suffix='.npz')
    env = dict(os.environ, HLA_LIB=os.path.join(root, 'highlyaccurate_amd', lib), HLA_ALLOW_STALE='1')
    subprocess.run([sys.executable, '-c', WORKER % root, f], env=env, check=True)
    res.append(np.load(f))
a, b = res
bad = [k for k in a.files if not np.array_equal(a[k], b[k], equal_nan=True)]
# (the LM backward scatters with fp32 atomics: gradients behind it may differ in the last bits from run to run; the forward may not)
print('keys', len(a.files), 'differing', bad[:8], 'forward identical:', 'trace' not in bad and 'loss' not in bad)
for k in bad[:8]:
    print(k, float(np.abs(a[k] - b[k]).max()), float(np.abs(a[k]).max()))
