"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, mean of each counter per dispatch."""
import csv, sys, collections, glob
files = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '?')[:70]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in sorted(agg.items(), key=lambda kv: -sum(kv[1].get('SQ_WAVE_CYCLES', [0]))):
    n = max(len(v) for v in cs.values())
    print(f'{k}  (dispatches {n})')
    for c, v in sorted(cs.items()):
        print(f'    {c:32s} mean {sum(v)/len(v):.4g}')
