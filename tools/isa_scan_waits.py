"""Find kernels whose global loads are followed by a full `s_waitcnt vmcnt(0)` before the next load is issued -- the signature of
branch-guarded loads that hipcc compiles into dependent round trips (round 5: the weight-gradient loaders, the data gradients'
masked flush).  Compiles the given csrc translation units to assembly and prints, per kernel, (#load->full-wait patterns, #loads).

    python tools/isa_scan_waits.py                    # vgg.hip / vgg_backward.hip for bf16 and fp16x3, the LM files
A count is a place to LOOK, not a verdict: a prologue load followed by a wait is fine; a loop of them on a hot path is not."""
import glob, os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, 'highlyaccurate_amd', 'csrc')
jobs = [('vgg.hip', 1), ('vgg.hip', 3), ('vgg_backward.hip', 1), ('vgg_backward.hip', 3), ('lm_solve.hip', -1), ('lm_backward.hip', -1), ('lm_g2s.hip', -1)]
tmp = tempfile.mkdtemp()
res = []
for f, d in jobs:
    out = os.path.join(tmp, f'{f}.{d}.s')
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=fast', '-munsafe-fp-atomics', f'-I{csrc}',
                    f'-DHLA_TU_DTYPE={d}', '-S', '--cuda-device-only', '-o', out, os.path.join(csrc, f)], stderr=subprocess.DEVNULL, check=True)
    name, body = None, []
    def flush(name, body):
        if not name:
            return
        ins = [l.strip() for l in body if l.strip() and not l.strip().startswith(('.', ';')) and not l.strip().endswith(':')]
        n = loads = 0
        for i, x in enumerate(ins):
            if re.match(r'(global_load|buffer_load)', x) and ' lds' not in x:
                loads += 1
                for y in ins[i + 1:i + 26]:
                    if re.match(r'(global_load|buffer_load)', y):
                        break
                    if y.startswith('s_waitcnt') and 'vmcnt(0)' in y:
                        n += 1
                        break
        if loads >= 4 and n >= 4:
            res.append((n, loads, name, f'{f} dtype {d}'))
    for l in open(out):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            flush(name, body); name, body = m.group(1), []
        else:
            body.append(l)
    flush(name, body)
for n, loads, name, where in sorted(res, reverse=True)[:int(sys.argv[1]) if len(sys.argv) > 1 else 30]:
    try:
        demangled = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        demangled = name
    print(f'{n:3d} of {loads:3d} loads  {demangled[:110]}  [{where}]')
