"""Regenerate DESIGN.md section 5's "Round-3 (measured ...)" paragraph from the committed profiles/r03_* files, so that the text
cannot drift from them:  python tools/design_numbers.py "<comma-separated pairs/s of the round's full runs>" """
import csv, json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda n: os.path.join(root, 'profiles', n)
d = json.loads(open(P('r03_bench.json')).read().strip().splitlines()[-1])
rows = list(csv.DictReader(open(P('r03_rocprofv3_kernel_stats.csv'))))
vals = [round(float(r['AverageNs']) / 1e3, 2) for r in rows[:7]]
u = json.loads(open(P('r03_bench_under_rocprof.json')).read().strip().splitlines()[-1])
c, lm, k, t = d['cpu_baseline'], d['lm_roofline'], d['kernels'], d['train']
runs = sys.argv[1] if len(sys.argv) > 1 else str(round(d['value']))
new = (f"Round-3 *(measured; `profiles/r03_bench.json` = the last complete `tools/make_profiles.sh r03` run, on the final kernels; the full "
       f"default-line runs of the round on the final or the few previous kernel states, different boxes: {runs} — the pool's boxes differ by 5 %; the slowest ran the timed leg right after six minutes of GPU tests)*: **{d['value']:.0f} pairs/s** ({d['ms_per_step']} ms per "
       f"32-pair step; {d['conv_tflops_executed']:.0f} TF over the 214.34 GFLOP per pair that are executed), training step {t['value']:.0f} pairs/s ({t['blocks_ms_per_step'][0]} / {t['blocks_ms_per_step'][1]} ms in its two timed blocks; {t['with_train_ground_crop']['value']:.0f} with "
       f"`train_ground_crop`), fp16x3 {d['by_precision']['fp16x3']['value']:.0f} / {t['by_precision']['fp16x3']['value']:.0f} (inference / training), exact fp32 {d['by_precision']['fp32']['value']:.0f} / {t['by_precision']['fp32']['value']:.0f}, fp16 (configs[4]'s dtype on this workload) {d['by_precision']['fp16']['value']:.0f}, Ford {d['secondary']['configs[3] Ford']['value']:.0f}, hi-res fp16 {d['secondary']['configs[4] hires fp16']['value']:.0f}, CPU port {c['inference_b1']:.2f} (B = 1) / "
       f"{c['inference_b8']:.2f} (B = 8) / {c['training_b1']:.2f} (training) pairs/s; `conv3x3<NT2>` {d['roofline']['avg_launch_us']} µs per launch by `bench.py`'s HIP events ({d['roofline']['achieved']:.0f} TF = {d['roofline']['frac']:.3f} of the peak; 493 MB "
       f"of counter traffic per launch against 411 MB algorithmic), {vals[0]} µs average over 1100 launches in "
       f"`profiles/r03_rocprofv3_kernel_stats.csv` (that run's own events: {u['roofline']['avg_launch_us']} µs); pooled {vals[1]} µs ({k['conv3x3_kernel<MT4,NT2,pool>']['tflops']:.0f} TF by the events), `conv02` {vals[2]} µs ({k['conv02_kernel']['tflops']:.0f} TF), `conv3x3<NT1>` "
       f"{vals[3]} µs ({k['conv3x3_kernel<MT4,NT1>']['tflops']:.0f} TF) in the same CSV (r02: 395 / 613 / 238); the LM loop: `lm_accum<64 / 128 / 256>` {vals[4]} / {vals[5]} / {vals[6]} µs = {(vals[4] + vals[5] + vals[6]) * 5 / 1e3:.3f} ms per forward "
       f"({lm['avg_launch_us']} µs average by the instrumented events); `lm_roofline`: 94.6 MB / {lm['avg_launch_us']} µs = {lm['achieved'] / 1e3:.2f} TB/s = {lm['frac']:.3f} of the HBM peak. ")
p = os.path.join(root, 'DESIGN.md')
s = open(p).read()
i = s.index("Round-3 *(measured; `profiles/r03_bench.json` = the last complete")
j = s.index("Per layer *(`tools/probes/infer_launches.py`, satellite branch, B = 32)*")
open(p, 'w').write(s[:i] + new + s[j:])
print(new)
