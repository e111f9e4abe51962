"""bench.py's launcher contract (no GPU needed): `--gpus N` is what decides the number of ranks."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_decides_the_world_size():
    import bench
    # no launcher: N = 1 runs in-process, N > 1 must spawn (None = "be the launcher")
    assert bench.resolve_world(1, {}) == (1, 0, 0)
    assert bench.resolve_world(2, {}) is None and bench.resolve_world(8, {}) is None
    # under a launcher the flag and WORLD_SIZE must agree ...
    assert bench.resolve_world(4, {'WORLD_SIZE': '4', 'RANK': '3', 'LOCAL_RANK': '3'}) == (4, 3, 3)
    assert bench.resolve_world(1, {'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'}) == (1, 0, 0)
    # ... a mismatch is refused instead of reporting a 1-GPU number under "--gpus 8"
    with pytest.raises(SystemExit):
        bench.resolve_world(8, {'WORLD_SIZE': '1'})
    with pytest.raises(SystemExit):
        bench.resolve_world(2, {'WORLD_SIZE': '4'})


def test_self_launch_command_is_the_drivers_line():
    import bench
    cmd = bench.launcher_command(8, ['--gpus', '8', '--steps', '5', '--warmup', '2'], 29511)
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1']
    assert '--nproc-per-node=8' in cmd and cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[cmd.index('--master-port') + 1] == '29511'
    assert cmd[-7] == os.path.join(ROOT, 'bench.py') and cmd[-6:] == ['--gpus', '8', '--steps', '5', '--warmup', '2']


def test_world_size_mismatch_exits_before_touching_the_gpu():
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8'], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in r.stderr


def test_telemetry_sampler_reads_hwmon_files_and_summarises(tmp_path, monkeypatch):
    """roofline.clock_mhz_mean / power_w_mean come from bench.Telemetry: a side thread reading the GPU's hwmon
    freq1_input (Hz) and power1_input (uW).  Pinned on fake sysfs files: units, the mean, and that an absent / ambiguous device
    yields an explicit 'samples: 0' instead of another GPU's numbers."""
    import time
    import bench
    f, p = tmp_path / 'freq1_input', tmp_path / 'power1_input'
    f.write_text('1650000000\n'); p.write_text('1250000000\n')
    monkeypatch.setattr(bench.Telemetry, '_find', classmethod(lambda cls, i: (str(f), str(p))))
    t = bench.Telemetry(0, hz=200.0)
    with t:
        time.sleep(0.05)
    s = t.summary()
    assert s['samples'] >= 3 and s['clock_mhz_mean'] == 1650.0 and s['power_w_mean'] == 1250.0 and s['power_w_max'] == 1250.0
    monkeypatch.setattr(bench.Telemetry, '_find', classmethod(lambda cls, i: None))
    t = bench.Telemetry(0)
    with t:
        pass
    assert t.summary()['samples'] == 0 and 'why' in t.summary()


def test_bench_line_carries_the_round5_fields():
    """The fields VERDICT r04 asked for are written by bench.py's source: the four telemetry fields of `roofline`, and
    `train.roofline` / `train.gflop_per_pair_executed` (the training step against the MFMA peak on executed FLOPs)."""
    import os
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py')).read()
    for key in ("res['roofline']['clock_mhz_mean']", "res['roofline']['power_w_mean']", "res['roofline']['mfma_sustained_clock_mhz_mean']",
                "res['roofline']['mfma_sustained_power_w_mean']", "train['roofline']", "train['gflop_per_pair_executed']", "train['telemetry']"):
        assert key in src, key


def test_bench_line_carries_the_round6_fields():
    """`matched_accuracy` at the top level (the fp16x3 numbers next to `value`), `train.step_breakdown` with the
    named phases and `inter_step_idle`, the secondary legs in fp16x3 with their pose deviation, and the headline's clock / power
    sampled over a separate >= 1 s block -- pinned on bench.py's source and on step_breakdown's arithmetic (fake events)."""
    import os
    import bench
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py')).read()
    for key in ("res['matched_accuracy']", "'step_breakdown': breakdown", "'inter_step_idle'", "secondary[key]['accuracy']", "'fp16x3'", "n_tele",
                "'final_pose_dev_shift_m': acc.get('final_pose_dev_shift_m')", "block_steps=n_tele"):
        assert key in src, key
    for k in ('fwd_sat', 'fwd_grd', 'lm_fwd', 'glue', 'lm_bwd', 'vgg_bwd', 'optimizer', 'inter_step_idle', 'host_enqueue_ms'):
        assert f"'{k}'" in src, k


def test_grad_sync_overlap_report_arithmetic_on_fake_events(monkeypatch):
    """parallel.GradSync.overlap_report (the evidence hook of the N > 1 line): per step, a bucket's issue time
    relative to the first wait and the time from the first wait to the last collective returning; bench.step_breakdown puts the
    medians into the line (pinned on its source)."""
    import os
    import torch
    from highlyaccurate_amd.parallel import GradSync
    clock = {'t': 0.0}

    class Ev:
        def __init__(self, enable_timing=True):
            self.t = None

        def record(self):
            self.t = clock['t']

        def elapsed_time(self, other):
            return other.t - self.t

    monkeypatch.setattr(torch.cuda, 'Event', Ev)
    gs = GradSync()
    gs.timing = True

    class Buf:
        is_cuda = True

        def __init__(self, n):
            self._n = n

        def numel(self):
            return self._n

    for step in range(2):
        clock['t'] += 10.0; gs._mark_start(Buf(10000))          # satellite bucket: issued 5 ms before the backward's end
        clock['t'] += 4.0; gs._mark_start(Buf(5000))            # ground bucket: 1 ms before
        clock['t'] += 1.0; gs._mark('pre', Buf(1)); clock['t'] += 0.25; gs._mark('post', Buf(1))
        gs._mark('pre', Buf(1)); clock['t'] += 0.5; gs._mark('post', Buf(1))        # second finish(): its wait extends the exposure
    rep = gs.overlap_report()
    assert len(rep) == 2 and gs._timeline == []
    for r in rep:
        assert r['bucket_issue_ms_before_backward_end'] == [5.0, 1.0] and r["bucket_bytes"] == [40000, 20000] and abs(r['exposed_ms'] - 0.75) < 1e-9, r
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py')).read()
    for key in ("out['allreduce_exposed_ms']", "out['allreduce_buckets']", "'issued_ms_before_backward_end'"):
        assert key in src, key


def test_step_breakdown_arithmetic_on_fake_events(monkeypatch):
    """bench.step_breakdown: phase = time between consecutive marks, inter_step_idle = optimizer mark -> next step's begin mark."""
    import torch
    import bench
    from highlyaccurate_amd import _s2gp
    clock = {'t': 0.0}

    class Ev:
        def __init__(self, enable_timing=True):
            self.t = None

        def record(self):
            self.t = clock['t']

        def elapsed_time(self, other):
            return other.t - self.t

    monkeypatch.setattr(torch.cuda, 'Event', Ev)
    monkeypatch.setattr(torch.cuda, 'synchronize', lambda *a, **k: None)
    durs = dict(fwd_sat=4.0, fwd_grd=3.0, lm_fwd=1.0, loss=0.25, lm_bwd=2.5, vgg_bwd=9.0)

    def tstep():
        for name, dt in durs.items():
            clock['t'] += dt
            _s2gp._phase(name)
        clock['t'] += 0.5          # the optimizer, marked by step_breakdown itself right after tstep returns
    def tstep_with_idle():
        clock['t'] += 0.125        # (work in front of the first mark of a step lands in its first phase)
        tstep()
    out = bench.step_breakdown(tstep_with_idle, 3)
    assert out['steps'] == 3 and out['fwd_sat'] == 4.125 and out['fwd_grd'] == 3.0 and out['glue'] == 0.25 and out['vgg_bwd'] == 9.0
    assert out['optimizer'] == 0.5 and out['inter_step_idle'] == 0.0 and abs(out['sum'] - 20.375) < 1e-9
    assert _s2gp.PHASE_HOOK is None
