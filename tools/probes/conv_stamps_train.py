"""Cycle stamps (see conv_stamps.py) of chosen conv3x3_kernel launches of one TRAINING step: ordinals count launch_conv calls from the
start of the step (forward: 9 per branch, satellite first; backward: 12 data-gradient launches per branch in the order of
vgg_backward.hip, satellite first).  HLA_LIB=...libhla_stamps.so HLA_ALLOW_STALE=1 python tools/probes/conv_stamps_train.py bf16 29 41"""
import ctypes as C, json, sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from highlyaccurate_amd import _lib
prec = sys.argv[1]
ords = [int(v) for v in sys.argv[2:]]
dev = torch.device('cuda:0')
B = 32
net = bench.build_net('kitti', prec, 5, dev).train()
net.args.bwd_two_streams = 0
sat, grd, extra = bench.make_inputs('kitti', B, (256, 1024), 512, dev, 0)
gt = [torch.rand(B, 1, device=dev) * 2 - 1 for _ in range(3)]
lib = _lib.load()
lib.hla_debug_conv_stamps.argtypes = [C.c_void_p, C.c_int]
def step():
    net.zero_grad(set_to_none=True)
    r = net(sat, grd, gt[0], gt[1], gt[2], mode='train')
    r[0].backward()
for _ in range(3): step()
torch.cuda.synchronize()
buf = torch.zeros(64 << 20, dtype=torch.uint8, device=dev)
for k in ords:
    buf.zero_()
    lib.hla_debug_conv_stamps(C.c_void_p(buf.data_ptr()), k)
    step()
    torch.cuda.synchronize()
    lib.hla_debug_conv_stamps(None, -1)
    gx, gy = C.c_uint(0), C.c_uint(0)
    lib.hla_debug_conv_stamps_grid(C.byref(gx), C.byref(gy))
    nw = gx.value * gy.value * 4
    s = buf[: nw * 8 * 8].cpu().numpy().view(np.uint64).reshape(nw, 8).astype(np.int64)
    s = s[s[:, 5] > 0]
    seg = {'prologue': s[:, 1] - s[:, 0], 'stage0_issue': s[:, 2] - s[:, 1], 'stage0_barrier_wait': s[:, 3] - s[:, 2],
           'later_stages': s[:, 4] - s[:, 3], 'epilogue': s[:, 5] - s[:, 4], 'life': s[:, 5] - s[:, 0]}
    hw = s[:, 7]
    cu = ((hw >> 32) & 0xf) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 0xf) * 16 + ((hw >> 4) & 3)
    res = []
    for c in np.unique(cu):
        m = cu == c
        res.append(float(seg['life'][m].sum()) / float(s[m, 5].max() - s[m, 0].min()))
    print(json.dumps({'ordinal': k, 'grid': [gx.value, gy.value], 'waves_stamped': int(len(s)), 'cycles_mean': {n: int(v.mean()) for n, v in seg.items()},
                      'cycles_median': {n: int(np.median(v)) for n, v in seg.items()}, 'resident_waves_per_simd': round(float(np.mean(res)), 2)}))
