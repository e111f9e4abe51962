"""Per-launch durations of the split-K reductions (reduce_partials4_kernel) of ONE training step, from a rocprofv3 kernel trace:

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/tr -- python $REPO/bench.py --steps 1 --warmup 0 \
        --no-cpu-baseline --no-kernel-timing --no-extra-legs --train-steps 2;  cd $REPO && python tools/probes/reduce_trace.py

What it showed (round 3): on the stream that owns the chip a reduction takes 13-96 us (75 MB of partials each); on the side stream,
next to the other branch's MFMA kernels, the SAME launches show 31-453 us -- they wait for workgroup slots.  The 7.8 % this
kernel has in the training step's rocprof --stats CSV is mostly that waiting, not work."""
import csv, glob, sys
f=glob.glob('gpurun_out/tr/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
# last 44 reduce_partials4 launches (two branches x 11?) -> print with preceding wgrad
idx=[i for i,n in enumerate(names) if n.startswith('reduce_partials4')]
last=idx[-20:]
for i in last:
    r=rows[i]; d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    # find the preceding wgrad
    j=i-1
    while j>=0 and 'wgrad' not in names[j]: j-=1
    w=rows[j]; wd=(int(w['End_Timestamp'])-int(w['Start_Timestamp']))/1e3
    print(f"reduce {d:7.1f} us grid {r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size')} | wgrad {wd:7.1f} us grid {w.get('Grid_Size_X',w.get('Grid_Size'))}x{w.get('Grid_Size_Y','')}x{w.get('Grid_Size_Z','')} stream {r.get('Stream_Id', r.get('Queue_Id'))}")
