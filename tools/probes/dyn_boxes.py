"""Which share of its backward tiles does the satellite branch visit in one bench-shaped training step?
(hla_vgg_backward_live_tiles: the data-dependent trimming of vgg_backward.hip)"""
import sys, torch
sys.path.insert(0, '/root/repo')
from highlyaccurate_amd import synthetic
from highlyaccurate_amd.models_kitti import LM_S2GP

d = torch.device('cuda:0')
net = LM_S2GP(synthetic.reference_args(precision='bf16'))
net.load_state_dict(synthetic.model_state(1))
net = net.to(d).train()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
sat, grd, gu, gv, gh = [t.to(d) for t in synthetic.images(3, B)]
net.bwd_stats = {}
r = net(sat, grd, gu, gv, gh, mode='train')
r[0].backward()
torch.cuda.synchronize()
st = net.bwd_stats
print(f"B = {B}: satellite-branch backward visits {st['live_tiles']} of {st['total_tiles']} tiles per sample "
      f"({st['live_tiles'] / max(st['total_tiles'], 1):.3f}), summed over its 12 dgrad and 11 wgrad launches")
