"""Same-box A/B of the LM backward's launches across builds of libhla (HLA_LIB): per level the mean lm_bwd_accum time of one step.
    gpurun -- 'python tools/probes/lmbwd_ab.py libhla.so libhla_v1.so libhla.so libhla_v1.so'"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for lib in sys.argv[1:] or ['libhla.so']:
    env = dict(os.environ, HLA_ALLOW_STALE='1', HLA_LIB=os.path.join(root, 'highlyaccurate_amd', lib))
    out = subprocess.run([sys.executable, os.path.join(root, 'tools/probes/train_launches.py'), os.environ.get('P', 'bf16'), '0'], env=env, capture_output=True, text=True)
    t = [float(l.split()[1]) for l in out.stdout.splitlines() if l.startswith('lm_bwd_accum ') and 'us' in l]
    if len(t) < 15:
        print(lib, 'FAILED', out.stderr[-400:]); continue
    # launch order: steps 14..0 = levels 2,1,0,2,1,0,...
    lv = [sum(t[i::3][:5]) / 5 for i in range(3)]
    tot = [l for l in out.stdout.splitlines() if l.startswith('lm_bwd_accum ') and ' x' in l]
    print(f'{lib:18s} C64 {lv[0]:6.1f}  C128 {lv[1]:6.1f}  C256 {lv[2]:6.1f} us   {tot[0] if tot else ""}', flush=True)
