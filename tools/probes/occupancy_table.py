"""Per launch of one inference step (BOTH branches): workgroups, resident capacity, generations, tail loss, microseconds, TFLOP/s
(VERDICT r04 #4).  Workgroup counts are computed from the layer geometry exactly as csrc/vgg.hip launches them (row trimming of
the cropped ground branch included); durations come from the library's per-launch events.
    python tools/probes/occupancy_table.py [bf16|fp16x3] > profiles/r05_per_layer.json"""
import json, sys, torch
sys.path.insert(0, '/root/repo')
from types import SimpleNamespace
from highlyaccurate_amd.models_kitti import LM_S2GP
from highlyaccurate_amd import _lib
from highlyaccurate_amd._s2gp import dead_ground_rows
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
d = torch.device('cuda:0')
args = SimpleNamespace(level=3, N_iters=5, using_weight=0, loss_method=0, proj='geo', Optimizer='LM', rotation_range=10.0, shift_range_lat=20.0, shift_range_lon=20.0, damping=0.1, train_damping=0, dropout=0, use_hessian=0, use_gt_depth=0, visualize=0, coe_shift_lat=100.0, coe_shift_lon=100.0, coe_heading=100.0, coe_L1=100.0, coe_L2=100.0, coe_L3=100.0, coe_L4=100.0, estimate_depth=0, precision=prec)
net = LM_S2GP(args).to(d).eval()
B = 32
sat, grd = torch.rand(B, 3, 512, 512, device=d), torch.rand(B, 3, 256, 1024, device=d)
with torch.no_grad():
    for _ in range(5): net(sat, grd, mode='test')
    torch.cuda.synchronize()
    _lib.prof_enable(True); _lib.prof_fetch()
    N = 6
    for _ in range(N): net(sat, grd, mode='test')
    recs = _lib.prof_fetch()
_lib.prof_enable(False)
n = len(recs) // N
conv = [k for k in range(n) if recs[k][0].startswith('conv')]
assert len(conv) == 20, len(conv)


def geometry(H, W, f):
    """(layer, Cout, H_l, W_l, row_begin, per-CU capacity) of the ten launches of one branch; f = first_row8 (0: whole image)"""
    sixteen = prec in ('bf16', 'fp16')
    r = lambda v: v if f else 0
    return [('conv0+conv2+pool', 64, H, W, r(8 * f - 32), 3 if sixteen else 2),
            ('conv5', 128, H // 2, W // 2, r(4 * f - 15), 2), ('conv7+pool', 128, H // 2, W // 2, r(4 * f - 14), 2),
            ('conv10', 256, H // 4, W // 4, r(2 * f - 6), 2), ('conv12', 256, H // 4, W // 4, r(2 * f - 5), 2),
            ('conv14+pool', 256, H // 4, W // 4, r(2 * f - 4), 2), ('dec1.1', 128, H // 4, W // 4, r(2 * f - 2), 2),
            ('dec1.3', 128, H // 4, W // 4, r(2 * f - 1), 2), ('dec2.1', 64, H // 2, W // 2, r(4 * f - 1), 3),
            ('dec2.3', 64, H // 2, W // 2, r(4 * f), 3)]


skip = dead_ground_rows(256)
f8 = (256 // 8) // 2 - skip // 8
rows = []
for branch, geo in (('sat', geometry(512, 512, 0)), ('grd', geometry(256 - skip, 1024, f8))):
    for i, (name, cout, Hl, Wl, rb, cap) in enumerate(geo):
        k = conv[i + (10 if branch == 'grd' else 0)]
        us = sum(recs[k + j * n][1] for j in range(N)) / N * 1e3
        fl = recs[k][2]
        wgs = ((Wl + 31) // 32) * ((Hl - rb + 7) // 8) * B * (cout // 128 if cout >= 128 else 1)
        gens = wgs / (cap * 256)
        import math
        tail = 1.0 - gens / math.ceil(gens)
        rows.append({'branch': branch, 'layer': name, 'rows': Hl - rb, 'workgroups': wgs, 'resident_capacity': cap * 256,
                     'generations': round(gens, 2), 'tail_loss_bound': round(tail, 3), 'us': round(us, 1), 'tflops': round(fl / us / 1e6, 1)})
tot = sum(r['us'] for r in rows)
for r in rows:
    r['share_of_conv_time'] = round(r['us'] / tot, 3)
print(json.dumps({'precision': prec, 'batch': B, 'note': 'tail_loss_bound = idle share of the last resident generation if workgroups '
                  'retired in lock-step (an upper bound: they do not); launches above 0.10 listed in over_10pct',
                  'over_10pct': [f"{r['branch']} {r['layer']} ({r['tail_loss_bound']}, {r['share_of_conv_time']} of the conv time)" for r in rows if r['tail_loss_bound'] > 0.10],
                  'launches': rows}, indent=1))
