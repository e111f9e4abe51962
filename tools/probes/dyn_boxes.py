"""Print the boxes of the backward's data-dependent trimming (vgg_backward.hip, bwd_boxes_kernel) for one bench-shaped
training step: the last 1 KiB of hla_vgg_backward's workspace holds {raw[3][4] | 12 ConvDyn | 11 wgrad boxes}."""
import ctypes, sys, torch
sys.path.insert(0, '/root/repo')
import highlyaccurate_amd.VGG as V
from highlyaccurate_amd import synthetic
from highlyaccurate_amd.models_kitti import LM_S2GP

d = torch.device('cuda:0')
seen = []
orig_empty = torch.empty
net = LM_S2GP(synthetic.reference_args(precision='bf16'))
net.load_state_dict(synthetic.model_state(1))
net = net.to(d).train()
B = 32
sat, grd, gu, gv, gh = [t.to(d) for t in synthetic.images(3, B)]

real = V._lib.load().hla_vgg_backward
captured = []
class Spy:
    def __call__(self, *a):
        rc = real(*a)
        torch.cuda.synchronize()
        ws, nbytes = a[10], a[11]
        p = getattr(ws, 'value', ws)
        buf = torch.empty(256, dtype=torch.int32, device=d)
        # dyn is the last 256-B aligned region of the plan: 816 bytes -> 1024
        ctypes.CDLL('libamdhip64.so').hipMemcpy(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(p + nbytes - 1024), 1024, 3)
        captured.append((buf.cpu().tolist(), a[13], a[14]))
        return rc
lib = V._lib.load()
lib.hla_vgg_backward = Spy()
r = net(sat, grd, gu, gv, gh, mode='train')
r[0].backward()
torch.cuda.synchronize()
names_c = ['D10', 'D9U', 'D9S', 'D8', 'D7U', 'D7S', 'D6', 'D5', 'D4', 'D3', 'D2', 'D1']
names_w = ['W10', 'W9', 'W8', 'W7', 'W6', 'W5', 'W4', 'W3', 'W2', 'W1', 'W0']
for v, H, W in captured:
    print(f'--- branch H={H} W={W}')
    print('raw', [v[i * 4:(i + 1) * 4] for i in range(3)])
    for i, n in enumerate(names_c):
        o = 16 + 12 * i
        print(f'{n:4s} out {v[o:o+4]} src {v[o+4:o+8]} add {v[o+8:o+12]}')
    for i, n in enumerate(names_w):
        o = 16 + 12 * 12 + 4 * i
        print(f'{n:4s} g {v[o:o+4]}')
