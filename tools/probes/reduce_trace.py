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
