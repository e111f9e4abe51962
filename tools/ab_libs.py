"""Same-box A/B: run bench.py's headline leg against several builds of libhla kept next to the product library (HLA_LIB=...,
e.g. `python -m highlyaccurate_amd.build --out=libhla_base.so` in a checkout of the previous commit) and print the kernel numbers.
    gpurun -- 'python tools/ab_libs.py libhla_base.so libhla.so libhla_base.so libhla.so'"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:] or ['libhla.so']
for lib in libs:
    env = dict(os.environ, HLA_BENCH_NOCHECK='1', HLA_ALLOW_STALE='1', HLA_LIB=os.path.join(root, 'highlyaccurate_amd', lib))
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '5', '--warmup', '2', '--no-cpu-baseline', '--train-steps', os.environ.get('VARIANTS_TRAIN', '0'), '--no-extra-legs', '--precision', os.environ.get('VARIANTS_PRECISION', 'bf16'),
                          '--train-precision', os.environ.get('VARIANTS_TRAIN_PRECISION', os.environ.get('VARIANTS_PRECISION', 'bf16'))],
                         env=env, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception:
        print(lib, 'FAILED', out.stderr[-500:]); continue
    k = d['kernels']
    conv = {n.replace('conv3x3_kernel', 'c'): (v['avg_us'], v['tflops']) for n, v in k.items() if n.startswith("conv")}
    if d.get('train'):
        tk = d['train']['kernels']
        print(f"{lib:16s} train {d['train']['ms_per_step']:7.2f} ms  " + '  '.join(f"{n}:{v['avg_us']:.0f}us" for n, v in tk.items() if v['share'] > 0.03), flush=True)
    lm = '  '.join(f"{n}:{v['avg_us']:.1f}us" for n, v in k.items() if n.startswith('lm_'))
    print(f"{lib:16s} {d['value']:8.1f} pairs/s {d['ms_per_step']:7.3f} ms  " + '  '.join(f'{n}:{u:.0f}us/{t:.0f}TF' for n, (u, t) in conv.items()) + '  ' + lm, flush=True)
