"""Where do a kernel's scratch (spill) accesses sit relative to its MFMA loop?  python tools/isa_report.py file.s [name-substring]
(file.s from: hipcc --offload-arch=gfx950 <build flags> -DHLA_TU_DTYPE=1 -S --cuda-device-only -o file.s csrc/vgg.hip)"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
want = sys.argv[2] if len(sys.argv) > 2 else ''
starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
ends = [i for i, l in enumerate(lines) if '.amdhsa_kernel' in l]
for (i0, name) in starts:
    if want not in name:
        continue
    i1 = min(e for e in ends if e > i0)
    body = lines[i0:i1]
    mf = [i for i, l in enumerate(body) if 'v_mfma' in l]
    if not mf:
        continue
    sc = [i for i, l in enumerate(body) if 'scratch_' in l]
    # the main loop = between the first backward branch target region: approximate by mfma span
    inside = [i for i in sc if mf[0] < i < mf[-1]]
    loop_lbl = [i for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)]
    print(f'{name[:80]}: {len(body)} lines, {len(mf)} mfma [{mf[0]}..{mf[-1]}], scratch {len(sc)} ({len(inside)} inside the mfma span), '
          f'ds_read_b128 {sum("ds_read_b128" in l for l in body)}, global_load {sum("global_load" in l for l in body)}, '
          f'v_mov {sum("v_mov_b32" in l for l in body[mf[0]:mf[-1]])} in span')
    if inside and len(sys.argv) > 3:
        for i in inside[:40]:
            print('   ', i, body[i].strip())
