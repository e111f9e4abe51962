"""How does the CPU oracle scale with torch threads on this host?  (Informs bench.py's cpu_baseline.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import ref_cpu as O
net = O.build('kitti', O.default_args(), seed=1)
sat, grd, *_ = O.synth_images(101, 1)
for n in [int(x) for x in (sys.argv[1:] or ['8', '16', '32', '64'])]:
    torch.set_num_threads(n)
    with torch.no_grad():
        t0 = time.time(); net(sat, grd, mode='test'); t1 = time.time(); net(sat, grd, mode='test'); t2 = time.time()
    print(f'threads {n}: warm-up {t1 - t0:.2f}s, second {t2 - t1:.2f}s', flush=True)
