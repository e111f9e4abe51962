"""Per-wave cycle accounting of the conv3x3_kernel launches of one inference forward: where a workgroup's life goes
on the short-K layers.  Needs the tooling build:
    python -m highlyaccurate_amd.build --out=libhla_stamps.so -DHLA_CONV_STAMPS=1
    HLA_LIB=$PWD/highlyaccurate_amd/libhla_stamps.so HLA_ALLOW_STALE=1 python tools/probes/conv_stamps.py [bf16|fp16x3] > profiles/r06_conv_cycle_table.json
Stamps (s_memtime = shader-clock cycles, lane 0 of every wave): 0 entry, 1 after the prologue's barrier (first halo tile in LDS),
2 first stage's MFMAs issued and the next tile awaited / written, 3 after that stage's barrier, 4 after the main loop's last barrier,
5 after the epilogue.  HW_ID / XCC_ID give the CU: per CU the union of wave lifetimes against their sum is the mean number of
resident waves; ideal = the wave's MFMA issue cycles (32 per v_mfma_f32_32x32x16)."""
import ctypes as C, json, sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace
from highlyaccurate_amd.models_kitti import LM_S2GP
from highlyaccurate_amd import _lib
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
d = torch.device('cuda:0')
args = SimpleNamespace(level=3, N_iters=5, using_weight=0, loss_method=0, proj='geo', Optimizer='LM', rotation_range=10.0, shift_range_lat=20.0, shift_range_lon=20.0, damping=0.1, train_damping=0, dropout=0, use_hessian=0, use_gt_depth=0, visualize=0, coe_shift_lat=100.0, coe_shift_lon=100.0, coe_heading=100.0, coe_L1=100.0, coe_L2=100.0, coe_L3=100.0, coe_L4=100.0, estimate_depth=0, precision=prec)
net = LM_S2GP(args).to(d).eval()
lib = _lib.load()
assert hasattr(lib, 'hla_debug_conv_stamps'), 'not a -DHLA_CONV_STAMPS=1 build (HLA_LIB=...libhla_stamps.so)'
lib.hla_debug_conv_stamps.argtypes = [C.c_void_p, C.c_int]
B = 32
sat, grd = torch.rand(B, 3, 512, 512, device=d), torch.rand(B, 3, 256, 1024, device=d)
NS = 8
buf = torch.zeros(64 << 20, dtype=torch.uint8, device=d)
names = ['conv5', 'conv7+pool', 'conv10', 'conv12', 'conv14+pool', 'dec1.1', 'dec1.3', 'dec2.1', 'dec2.3']
cin = [64, 128, 128, 256, 256, 384, 128, 192, 64]
cout = [128, 128, 256, 256, 256, 128, 128, 64, 64]
KC = 32 if prec in ('bf16', 'fp16') else 16
rows = []
with torch.no_grad():
    for _ in range(4): net(sat, grd, mode='test')
    torch.cuda.synchronize()
    for k in range(18):
        buf.zero_()
        lib.hla_debug_conv_stamps(C.c_void_p(buf.data_ptr()), k)
        net(sat, grd, mode='test')
        torch.cuda.synchronize()
        lib.hla_debug_conv_stamps(None, -1)
        gx, gy = C.c_uint(0), C.c_uint(0)
        lib.hla_debug_conv_stamps_grid(C.byref(gx), C.byref(gy))
        nw = gx.value * gy.value * 4
        s = buf[: nw * NS * 8].cpu().numpy().view(np.uint64).reshape(nw, NS).astype(np.int64)
        ok = s[:, 5] > 0
        s = s[ok]
        l = k % 9
        nt = 2 if cout[l] >= 128 else 1
        nstage = cin[l] // KC
        mf_stage = 9 * 8 * nt * (1.5 if prec == 'fp16x3' else 1.0) * 32         # MFMA issue cycles of one stage per wave
        seg = {'prologue': s[:, 1] - s[:, 0], 'stage0_issue': s[:, 2] - s[:, 1], 'stage0_barrier_wait': s[:, 3] - s[:, 2],
               'later_stages': s[:, 4] - s[:, 3], 'epilogue': s[:, 5] - s[:, 4], 'life': s[:, 5] - s[:, 0]}
        hw = s[:, 7]
        cu = ((hw >> 32) & 0xf) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 0xf) * 16 + ((hw >> 4) & 3)   # (xcc, se, sh, cu, simd)
        # per SIMD: mean resident waves = sum of lifetimes / span; MFMA duty = waves' ideal cycles / span
        res, duty = [], []
        for c in np.unique(cu):
            m = cu == c
            span = float(s[m, 5].max() - s[m, 0].min())
            res.append(float(seg['life'][m].sum()) / span)
            duty.append(float(m.sum()) * nstage * mf_stage / span)
        row = {'branch': 'sat' if k < 9 else 'grd', 'layer': names[l], 'waves': int(len(s)), 'stages': nstage, 'wave_tile_channels': 32 * nt,
               'mfma_issue_cycles_per_wave': int(nstage * mf_stage),
               'cycles_mean': {n: int(v.mean()) for n, v in seg.items()}, 'cycles_median': {n: int(np.median(v)) for n, v in seg.items()},
               'simds_seen': int(len(res)), 'resident_waves_per_simd': round(float(np.mean(res)), 2), 'mfma_duty_per_simd': round(float(np.mean(duty)), 3)}
        row['non_mfma_share_of_life'] = round(1.0 - row['mfma_issue_cycles_per_wave'] / max(1, row['cycles_mean']['life']), 3)
        rows.append(row)
        print(json.dumps(row), file=sys.stderr)
    # the fused conv0 + conv2 kernel: 0 entry, 1 image patch + conv0 weights in (barrier), 2 conv0 computed into the halo buffers
    # (im2col by VALU + MFMA), 3 after its barrier, 4 conv2's MFMAs done, 5 after the pooling epilogue  (16-bit types: one round)
    for k, br in ((-10, 'sat'), (-11, 'grd')):
        buf.zero_()
        lib.hla_debug_conv_stamps(C.c_void_p(buf.data_ptr()), k)
        net(sat, grd, mode='test')
        torch.cuda.synchronize()
        lib.hla_debug_conv_stamps(None, -1)
        gx, gy = C.c_uint(0), C.c_uint(0)
        lib.hla_debug_conv_stamps_grid(C.byref(gx), C.byref(gy))
        nw = gx.value * 4
        s = buf[: nw * NS * 8].cpu().numpy().view(np.uint64).reshape(nw, NS).astype(np.int64)
        s = s[s[:, 5] > 0]
        seg = {'A_patch_load': s[:, 1] - s[:, 0], 'B_conv0_im2col': s[:, 2] - s[:, 1], 'B_barrier_wait': s[:, 3] - s[:, 2],
               'C_conv2_mfma': s[:, 4] - s[:, 3], 'epilogue': s[:, 5] - s[:, 4], 'life': s[:, 5] - s[:, 0]}
        mf = (2 * 72 + 12) * 32 * (1.5 * 2 if prec == 'fp16x3' else 1.0)
        hw = s[:, 7]
        cu = ((hw >> 32) & 0xf) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 0xf) * 16 + ((hw >> 4) & 3)
        res, duty = [], []
        for c in np.unique(cu):
            m = cu == c
            span = float(s[m, 5].max() - s[m, 0].min())
            res.append(float(seg['life'][m].sum()) / span)
            duty.append(float(m.sum()) * mf / span)
        row = {'branch': br, 'layer': 'conv0+conv2+pool (conv02_kernel)', 'waves': int(len(s)), 'mfma_issue_cycles_per_wave': int(mf),
               'cycles_mean': {n: int(v.mean()) for n, v in seg.items()}, 'cycles_median': {n: int(np.median(v)) for n, v in seg.items()},
               'resident_waves_per_simd': round(float(np.mean(res)), 2), 'mfma_duty_per_simd': round(float(np.mean(duty)), 3)}
        rows.append(row)
        print(json.dumps(row), file=sys.stderr)
print(json.dumps({'precision': prec, 'batch': B, 'clock': 's_memtime (shader clock)', 'launches': rows}, indent=1))
