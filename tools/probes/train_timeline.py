"""Timeline of ONE training step from a rocprofv3 kernel trace: which kernel ran when, on which stream (queue).

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/tl -- python $REPO/bench.py --steps 1 --warmup 0 \
        --no-cpu-baseline --no-kernel-timing --no-extra-legs --precision fp16x3 --train-precision fp16x3 --train-steps 2
    cd $REPO && python tools/probes/train_timeline.py gpurun_out/tl > gpurun_out/timeline.txt

Prints the LAST complete step (from the last conv02 pair's first launch back to the previous one): per launch the start offset,
the duration, the queue, and at the end per queue the busy time and the critical-path view (intervals where only one queue runs)."""
import csv, glob, sys
d = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/tl'
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
name = lambda r: r['Kernel_Name']
# a training step = two conv02 launches with a0 store (training forward) ... : cut at conv02 launches
c02 = [i for i, r in enumerate(rows) if 'conv02_kernel' in name(r)]
# steps start at every second conv02 (sat then grd); take the last full step
starts = c02[0::2]
if len(starts) < 2:
    print('need >= 2 steps'); sys.exit(1)
a, b = starts[-2], starts[-1]
seg = rows[a:b]
t0 = int(seg[0]['Start_Timestamp'])
qkey = 'Queue_Id' if 'Queue_Id' in seg[0] else 'Stream_Id'
import re
def short(n):
    n = re.sub(r'\(.*', '', n)
    n = n.replace('void ', '')
    return n[:70]
busy = {}
for r in seg:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    q = r.get(qkey, '?')
    busy[q] = busy.get(q, 0) + (e - s)
    print(f"{s/1e3:10.1f} us  +{(e-s)/1e3:8.1f} us  q{q:>3s}  {short(name(r))}")
end = max(int(r['End_Timestamp']) for r in seg) - t0
print(f'step span {end/1e6:.3f} ms; busy per queue:', {q: round(v / 1e6, 3) for q, v in busy.items()})
# the step PERIOD (this step's first launch to the next step's first launch) and what it does not contain: the idle gap between the
# last kernel of the step and the first one of the next -- the host's lateness, which no kernel trace row shows
nxt = int(rows[b]['Start_Timestamp']) - t0
print(f'step period {nxt/1e6:.3f} ms; idle between this step\'s last kernel and the next step\'s first: {(nxt - end)/1e3:.1f} us')
# gaps > 20 us inside the step (queue-agnostic: time during which NO kernel of either queue runs)
iv = sorted((int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0) for r in seg)
cur_end, gaps = iv[0][1], []
for s0, e0 in iv[1:]:
    if s0 - cur_end > 20000:
        gaps.append((cur_end, s0 - cur_end))
    cur_end = max(cur_end, e0)
print('idle gaps > 20 us inside the step (at us, length us):', [(round(a0 / 1e3, 1), round(g / 1e3, 1)) for a0, g in gaps],
      'sum', round(sum(g for _, g in gaps) / 1e3, 1))
