"""Per-LAYER HBM fetch bytes of one inference step from a rocprofv3 --pmc FETCH_SIZE pass (dispatch order), optionally for two
runs side by side (e.g. HLA_VGG_CHUNK=0 against 8): does running the high-resolution chain in cache-sized chunks take its
reads off the memory interface?   usage: pmc_per_layer.py <dir_a> [<dir_b>] > out.json"""
import csv, glob, json, sys


def last_step(d):
    rows = []
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == 'FETCH_SIZE']
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    conv = [r for r in rows if 'conv' in r['Kernel_Name']]
    c02 = [i for i, r in enumerate(conv) if 'conv02' in r['Kernel_Name']]
    # a step = from one "first conv02 of the satellite branch" to the next; with chunking there are several conv02 launches per
    # branch, so cut at the launch count instead: the run is `steps` identical steps
    return conv, c02


out = {}
for tag, d in zip(('a', 'b'), sys.argv[1:3]):
    conv, c02 = last_step(d)
    per = {}
    for r in conv:
        k = r['Kernel_Name'].split('(')[0][:60] + ' grid ' + r.get('Grid_Size', '?')
        e = per.setdefault(k, [0, 0.0])
        e[0] += 1; e[1] += float(r['Counter_Value'])
    tot = sum(v[1] for v in per.values())
    out[tag] = {'dir': d, 'conv_dispatches': len(conv), 'conv_fetch_GB_corrected_total': round(2 * tot * 1024 / 1e9, 3),
                'by_kernel_and_grid': {k: {'dispatches': v[0], 'fetch_MB_corrected_per_dispatch': round(2 * v[1] * 1024 / v[0] / 1e6, 1)}
                                       for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])}}
out['_note'] = ('FETCH_SIZE in KiB, doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read stream); Infinity-Cache hits '
                'appear to be counted by this counter, so equal totals do not prove equal HBM traffic -- but the step TIMES are what decided')
print(json.dumps(out, indent=1))
