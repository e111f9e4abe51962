#!/usr/bin/env python3
"""bench.py -- image-pairs/sec through the N-iter LM pose loop (BASELINE.json metric).

One "step" = one pass of the hot path over one synthetic batch already resident in HBM:
LM_S2GP.forward(sat, grd, mode='test') = two VGG16-U-Nets (bf16 MFMA) + 15 fused
projection/Jacobian/normal-equation/solve steps.  Workload at N=1: BASELINE configs[1]
(KITTI shapes, batch 32 per GPU, VGG-16 two-branch, 5 LM iters, 3-DoF, bf16).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Multi-GPU: one process per GPU; the batch shards over ranks with no data-path collective
(every sample's solve is independent); a barrier + synchronize brackets the timed
region and the reported time is the MAX over ranks.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

# dense peaks, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
# fp16x3: every algorithmic FLOP costs three fp16 MFMA FLOPs, so the ceiling on ALGORITHMIC FLOPs is a third of the fp16 peak
PEAK_TFLOPS = {'bf16': 2500.0, 'fp16': 2500.0, 'fp32': 157.3, 'fp16x3': 2500.0 / 3}
PEAK_HBM_GBS = 8000.0
# conv FLOPs per pair, forward, live outputs at level 3: 272.4 GFLOP
GFLOP_PER_PAIR_LIVE = 272.4
from highlyaccurate_amd._s2gp import dead_ground_rows  # noqa: E402


def cpu_baseline(max_seconds=30.0):
    """Time the CPU oracle (oracle/ref_cpu.py, a port of the reference's PyTorch path) on this host:
    B=1 KITTI-shape forward(mode='test'), no_grad, fp32, all cores.  Bounded sample."""
    from oracle import ref_cpu as O
    # measured on the MI355X host (256 logical CPUs): B=1 forward takes 1.40 / 1.21 / 1.29 / 2.58 / 204 s with
    # 8 / 16 / 32 / 64 / 256 torch threads (tests/diag/cpu_threads.py) -- oversubscription kills it, 16 is the best
    cores = min(os.cpu_count() or 1, int(os.environ.get('HLA_CPU_THREADS', '16')))
    torch.set_num_threads(cores)
    net = O.build('kitti', O.default_args(), seed=1)
    sat, grd, *_ = O.synth_images(101, 1)
    n, t_tot = 0, 0.0
    with torch.no_grad():
        t0 = time.time()
        net(sat, grd, mode='test')                      # warm-up (also bounds the sample)
        warm = time.time() - t0
        # about 10-15 s of CPU work (1.1-1.4 s per pair on this host), never more than max_seconds
        reps = max(1, min(int(12.0 / max(warm, 1e-3)) + 1, int(max_seconds / max(warm, 1e-3)) - 1, 12))
        for _ in range(reps):
            t0 = time.time()
            net(sat, grd, mode='test')
            t_tot += time.time() - t0
            n += 1
    return {'value': round(n / t_tot, 4), 'unit': 'pairs/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n} x (B=1 KITTI-shape forward, mode=test, no_grad, fp32) after 1 warm-up, '
                      f'torch {torch.__version__} CPU, {torch.get_num_threads()} threads'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=32, help='pairs per GPU')
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp16', 'fp32', 'fp16x3'])
    ap.add_argument('--n-iters', type=int, default=None, help='LM iterations (default 5; 10 for --model ford = BASELINE configs[3])')
    ap.add_argument('--grd-hw', type=int, nargs=2, default=[256, 1024], help='ground image size (BASELINE configs[4]: 512 2048)')
    ap.add_argument('--sat-a', type=int, default=512, help='satellite image side (BASELINE configs[4]: 1024)')
    ap.add_argument('--model', default='kitti', choices=['kitti', 'ford', 'g2sp'],
                    help='kitti = LM_S2GP (the headline, BASELINE configs[1]); ford = LM_S2GP_Ford; g2sp = LM_G2SP')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--train-steps', type=int, default=6, help='extra: time this many training steps (0 = skip)')
    a = ap.parse_args()
    if a.n_iters is None:
        a.n_iters = 10 if a.model == 'ford' else 5

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs an MI355X'
    # HLA_BENCH_REHEARSE=1: run the N>1 control flow on a box with fewer GPUs than ranks (ranks share devices, the
    # collectives go over gloo instead of RCCL).  For checking the multi-rank logic only; the numbers mean nothing.
    rehearse = bool(os.environ.get('HLA_BENCH_REHEARSE'))
    if rehearse:
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if rehearse:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from types import SimpleNamespace
    from highlyaccurate_amd import _lib
    from highlyaccurate_amd.models_kitti import LM_G2SP, LM_S2GP
    from highlyaccurate_amd.models_ford import LM_S2GP_Ford
    _lib.load()
    args = SimpleNamespace(level=3, N_iters=a.n_iters, using_weight=0, loss_method=0, proj='geo', Optimizer='LM',
                           rotation_range=10.0, shift_range_lat=20.0, shift_range_lon=20.0, damping=0.1,
                           train_damping=0, dropout=0, use_hessian=0, use_gt_depth=0, visualize=0,
                           coe_shift_lat=100.0, coe_shift_lon=100.0, coe_heading=100.0, coe_L1=100.0, coe_L2=100.0,
                           coe_L3=100.0, coe_L4=100.0, estimate_depth=0, precision=a.precision)
    torch.manual_seed(1234)                  # identical replicas on every rank (data-parallel training needs that) ...
    net = {'kitti': LM_S2GP, 'ford': LM_S2GP_Ford, 'g2sp': LM_G2SP}[a.model](args)
    # random-init weights of the reference architecture: Kaiming-normal(fan_out), zero bias (torchvision's
    # non-pretrained VGG init; there is no network for the pretrained checkpoint)
    for m in net.modules():
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            if m.bias is not None:
                torch.nn.init.zeros_(m.bias)
    net = net.to(dev).eval()
    torch.manual_seed(1234 + rank)           # ... and a different synthetic shard per rank (also decorrelates the re-init draws)
    B = a.batch
    sat = torch.rand(B, 3, a.sat_a, a.sat_a, device=dev)
    grd = torch.rand(B, 3, a.grd_hw[0], a.grd_hw[1], device=dev)

    if a.model == 'ford':       # BASELINE configs[3] / SURVEY 8(d): fixed camera-to-body rotation, 112.64 m tile
        extra = (112.64, torch.tensor([[[0., 0., 1.], [1., 0., 0.], [0., 1., 0.]]], device=dev).repeat(B, 1, 1),
                 torch.tensor([[1.7, 0.3, -1.2]], device=dev).repeat(B, 1))
    elif a.model == 'g2sp':     # left_camera_k of the 256x1024 frame (models_kitti.py:657-660 values)
        extra = (torch.tensor([[[582.9802, 0., 496.2420], [0., 482.7076, 125.0034], [0., 0., 1.]]], device=dev).repeat(B, 1, 1),)
    else:
        extra = ()

    def step():
        with torch.no_grad():
            return net(sat, grd, *extra, mode='test')

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    want_kt = (rank == 0) and not a.no_kernel_timing
    # ---- the timed region: exactly K steps, no instrumentation
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # ---- the same K steps once more on rank 0 with a HIP-event pair around every kernel launch (hla_prof_*, events on the
    # launch stream): per-kernel durations for the roofline.  Kept out of `value`: the event pairs cost ~9 % wall time
    # (measured: 8.73 vs 7.97 ms/step), which would also have made rank 0 the slowest rank of every multi-GPU run.
    recs, dt_ev = [], None
    n_ev = min(a.steps, 50)               # bound the instrumented pass (130 event pairs per step)
    if want_kt:
        _lib.prof_enable(True)
        t1 = time.perf_counter()
        for _ in range(n_ev):
            step()
        torch.cuda.synchronize()
        dt_ev = time.perf_counter() - t1
        _lib.prof_enable(False)
        recs = _lib.prof_fetch()
    if dist:
        dist.barrier()
    if dist:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if not os.environ.get('HLA_BENCH_NOCHECK'):     # timing-ablation builds (tools/variants.py) produce garbage
        assert all(torch.isfinite(o).all() for o in out)

    # ---- extra: the training step (forward(train) + HIP backward + gradient all-reduce + Adam), same shapes
    train = None
    if a.train_steps > 0:
      try:
          from highlyaccurate_amd.parallel import GradSync
          net.train()
          if dist:
              net.grad_sync = GradSync()
          opt = torch.optim.Adam(net.parameters(), lr=1e-4)
          gt = [torch.rand(B, 1, device=dev) * 2 - 1 for _ in range(3)]
          if a.model == 'ford':      # Ford_dataset.py:211 collates python floats: [B] float64
              gt = [g[:, 0].double() for g in gt]

          def tstep():
              opt.zero_grad(set_to_none=True)
              r = net(sat, grd, *extra, gt[0], gt[1], gt[2], mode='train')
              r[0].backward()
              opt.step()
              return r[0]
          for _ in range(2):          # warm-up: Adam state, caching-allocator segments for the backward workspaces
              tstep()
          torch.cuda.synchronize()
          if dist:
              dist.barrier()
          torch.cuda.synchronize()
          t1 = time.perf_counter()
          ar0 = net.grad_sync.bytes_reduced if dist else 0
          for _ in range(a.train_steps):
              lossv = tstep()
          ar_bytes = ((net.grad_sync.bytes_reduced - ar0) // a.train_steps) if dist else 0
          torch.cuda.synchronize()
          if dist:
              dist.barrier()
          torch.cuda.synchronize()
          tdt = time.perf_counter() - t1
          trecs = []
          if not a.no_kernel_timing:  # per-kernel table from two extra steps (not part of the timing).  EVERY rank runs them --
              if want_kt:             # a training step contains the gradient all-reduce -- but only rank 0 is instrumented
                  _lib.prof_enable(True)
              for _ in range(2):
                  tstep()
              torch.cuda.synchronize()
              if want_kt:
                  _lib.prof_enable(False)
                  trecs = _lib.prof_fetch()
          if dist:
              dist.barrier()
          if dist:
              tt = torch.tensor([tdt], device=dev, dtype=torch.float64)
              dist.all_reduce(tt, op=dist.ReduceOp.MAX)
              tdt = float(tt.item())
          train = {'value': round(B * world * a.train_steps / tdt, 3), 'unit': 'pairs/s', 'steps': a.train_steps,
                   'ms_per_step': round(tdt / a.train_steps * 1e3, 3), 'loss_finite': bool(torch.isfinite(lossv)),
                   'what': "forward(mode='train') + HIP backward (LM loop + both VGGs) + gradient all-reduce + Adam",
                   'allreduce_bytes_per_step': ar_bytes}
          if a.model != 'g2sp':
              # secondary number: the same step with args.train_ground_crop=1 (an extension: the ground branch trains on the
              # image rows that can reach the loss; loss and gradients equal to rounding, the RETURNED confidence maps are
              # only computed from the crop on -- DESIGN.md 3.5).  Not the default, so it is not `train.value`.
              net.args.train_ground_crop = 1
              for _ in range(2):
                  tstep()
              torch.cuda.synchronize()
              if dist:
                  dist.barrier()
              torch.cuda.synchronize()
              t1 = time.perf_counter()
              for _ in range(a.train_steps):
                  tstep()
              torch.cuda.synchronize()
              if dist:
                  dist.barrier()
              torch.cuda.synchronize()
              cdt = time.perf_counter() - t1
              net.args.train_ground_crop = 0
              if dist:
                  tt = torch.tensor([cdt], device=dev, dtype=torch.float64)
                  dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                  cdt = float(tt.item())
              train['with_train_ground_crop'] = {'value': round(B * world * a.train_steps / cdt, 3), 'unit': 'pairs/s',
                                                 'ms_per_step': round(cdt / a.train_steps * 1e3, 3)}
          if trecs:
              tagg = {}
              for name, ms, fl, by in trecs:
                  e = tagg.setdefault(name, [0, 0.0, 0.0])
                  e[0] += 1; e[1] += ms; e[2] += fl
              tot = sum(v[1] for v in tagg.values())
              train['kernels'] = {k: {'launches': v[0], 'avg_us': round(v[1] / v[0] * 1e3, 1), 'share': round(v[1] / tot, 3),
                                      'tflops': round(v[2] / (v[1] * 1e-3) / 1e12, 1) if v[2] else None}
                                  for k, v in sorted(tagg.items(), key=lambda kv: -kv[1][1])[:8]}
      except Exception as e:      # the headline line must still be printed
        train = {'error': repr(e)[:300]}

    if rank == 0:
        pairs = B * world * a.steps
        res = {
            'metric': 'image-pairs/sec through N-iter LM pose loop, KITTI shapes',
            'value': round(pairs / dt, 3), 'unit': 'pairs/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(dt / a.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': a.precision, 'data': 'synthetic',
            'config': {'workload': {'kitti': ("BASELINE configs[1]: LM_S2GP" if (a.sat_a, tuple(a.grd_hw)) == (512, (256, 1024)) else "BASELINE configs[4] sizes: LM_S2GP"), 'ford': "BASELINE configs[3] shapes: LM_S2GP_Ford",
                                    'g2sp': "SURVEY 8(f).2: LM_G2SP"}[a.model] + f".forward(mode='test'), sat {a.sat_a}x{a.sat_a}, grd {a.grd_hw[0]}x{a.grd_hw[1]}, "
                                   f"VGG-16 two-branch, level 3, {a.n_iters} LM iters x 3 levels, 3-DoF, random-init weights",
                       'pairs_per_gpu': B, 'global_batch': B * world, 'parallelism': f'batch-sharded x{world}, no collective',
                       'dead_work_skipped': 'dec3/conf3 (VGG.py:153-155,163: computed and dropped by the reference at level 3); '
                                            f'ground-image rows 0..{dead_ground_rows(grd.shape[-2]) - 1} (cannot reach the bottom-half '
                                            'rows the LM loop reads) and, layer by layer, the feature rows those rows do not '
                                            'depend on; computed rows are bit-identical, DESIGN.md 3.5)'},
        }
        # whole-forward conv rate on the FLOPs that were actually executed (the reference's as-written count is 316.3
        # GFLOP/pair, 272.4 without dec3/conf3)
        if recs:
            conv_fl = sum(fl for name, ms, fl, by in recs if name.startswith('conv'))      # over the K instrumented steps
            res['conv_gflop_per_pair_executed'] = round(conv_fl / (B * n_ev) / 1e9, 2)
            res['conv_tflops_live'] = round(conv_fl / n_ev * a.steps / dt / 1e12, 2)   # this GPU, against the un-instrumented time
        if recs:
            agg = {}
            for name, ms, fl, by in recs:
                e = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
                e[0] += 1; e[1] += ms; e[2] += fl; e[3] += by
            tot_ms = sum(v[1] for v in agg.values())
            kern = {k: {'launches': v[0], 'avg_us': round(v[1] / v[0] * 1e3, 2), 'share': round(v[1] / tot_ms, 4),
                        'tflops': round(v[2] / (v[1] * 1e-3) / 1e12, 2) if v[2] else None,
                        'gbs': round(v[3] / (v[1] * 1e-3) / 1e9, 1) if v[3] else None}
                    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
            dom = max((k for k in agg if agg[k][2] > 0), key=lambda k: agg[k][1])
            n, ms, fl, by = agg[dom]
            ach = fl / (ms * 1e-3) / 1e12
            traffic, tsrc = None, None
            try:      # HBM bytes per launch from the committed rocprofv3 PMC passes (separate --pmc runs, gfx950 correction)
                pm = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_hbm_traffic.json')))
                sym = {'conv3x3_kernel<MT4,NT2>': 'Li4ELi2ELi2ELi2ELb0', 'conv3x3_kernel<MT4,NT2,pool>': 'Li4ELi2ELi2ELi2ELb1',
                       'conv3x3_kernel<MT4,NT1>': 'Li4ELi1ELi2ELi2ELb0', 'conv3x3_kernel<MT4,NT1,pool>': 'Li4ELi1ELi2ELi2ELb1'}.get(dom)
                if a.precision == 'bf16' and B == 32 and sym:
                    for kname, v in pm.items():
                        if sym in kname:
                            traffic, tsrc = v['hbm_bytes_corrected'], 'profiles/r01_pmc_hbm_traffic.json'
            except Exception:
                pass
            res['events_pass_ms_per_step'] = round(dt_ev / n_ev * 1e3, 3)
            res['roofline'] = {'kernel': dom, 'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK_TFLOPS[a.precision],
                               'unit': 'TFLOP/s', 'frac': round(ach / PEAK_TFLOPS[a.precision], 4), 'traffic': traffic,
                               'traffic_unit': 'bytes/launch', 'traffic_source': tsrc,
                               'launches': n, 'avg_launch_us': round(ms / n * 1e3, 2),
                               'flops_per_launch': round(fl / n / 1e9, 3), 'flops_unit': 'GFLOP'}
            lm = [k for k in agg if k.startswith('lm_accum')]
            if lm:
                lms, lby = sum(agg[k][1] for k in lm), sum(agg[k][3] for k in lm)
                res['lm_roofline'] = {'kernel': 'lm_accum<*>', 'bound': 'hbm', 'achieved': round(lby / (lms * 1e-3) / 1e9, 1),
                                      'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(lby / (lms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
            res['kernels'] = kern
        if train:
            res['train'] = train
        if not a.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline()
        print(json.dumps(res), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
