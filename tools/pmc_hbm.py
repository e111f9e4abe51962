"""Combine two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes) into
per-kernel HBM bytes per launch.  usage: pmc_hbm.py <dir_fetch> <dir_write> <out.json>"""
import collections, csv, glob, json, sys


def means(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


fetch, nf = means(sys.argv[1], 'FETCH_SIZE')
write, _ = means(sys.argv[2], 'WRITE_SIZE')
out = {'_note': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), mean per dispatch over the bench.py leg named by the file (B=32). '
                'Counter unit is KiB. Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE reports half of a wide coalesced read stream on '
                'gfx950, so hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; WRITE_SIZE is uncalibrated.'}
try:      # stamp with the content hash of the kernel sources: bench.py ignores a file measured on other kernels
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from highlyaccurate_amd import build as _b
    out['_source_hash'] = _b.lib_hash()
except Exception as e:
    out['_source_hash'] = None
for k in sorted(fetch, key=lambda k: -fetch[k] * nf[k]):
    if any(s in k for s in ('conv', 'lm_', 'wgrad', 'reduce_partials', 'l2bwd')):
        out[k] = {'dispatches': nf[k], 'FETCH_SIZE_KiB': fetch[k], 'WRITE_SIZE_KiB': write.get(k, 0.0),
                  'hbm_bytes_corrected': (2 * fetch[k] + write.get(k, 0.0)) * 1024}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps({k: v['hbm_bytes_corrected'] for k, v in out.items() if isinstance(v, dict)}, indent=1))
