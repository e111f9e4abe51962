"""N>1 path on CPU: two gloo processes shard a batch, average gradients through highlyaccurate_amd.parallel
and must reproduce the single-process full-batch gradient (what the RCCL path does on the GPUs)."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy():
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 3, padding=1))
    unused = torch.nn.Parameter(torch.zeros(3))        # like `damping`: never receives a gradient
    net.register_parameter('unused', unused)
    return net


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from highlyaccurate_amd import parallel as P
    r, w, _ = P.init_distributed('gloo')
    assert (r, w) == (rank, world)
    net = _toy()
    torch.manual_seed(1)
    x = torch.rand(4, 3, 8, 8)
    xs = P.shard_batch(x, rank, world)
    assert xs.shape[0] == 2
    loss = net(xs).abs().mean()                      # mean over the LOCAL batch, as loss_func does
    loss.backward()
    # the two-bucket path the model's backward uses
    gs = P.GradSync()
    named = dict(net.named_parameters())
    h1 = gs.start({k: named[k].grad for k in ('0.weight', '0.bias')})
    h2 = gs.start({k: named[k].grad for k in ('2.weight', '2.bias')})
    gs.finish(h1); gs.finish(h2)
    assert named['unused'].grad is None
    # the flat-buffer path: gradients that are views of one buffer are reduced IN PLACE by one collective, the rest is staged
    local = {k: named[k].grad.clone() for k in ('0.weight', '0.bias', '2.weight', '2.bias')}      # already averaged: use as data
    flat = torch.empty(sum(local[k].numel() for k in ('0.weight', '0.bias', '2.weight')))
    views, o = {}, 0
    for k in ('0.weight', '0.bias', '2.weight'):
        n = local[k].numel()
        views[k] = flat[o:o + n].view_as(local[k])
        views[k].copy_(local[k] * (rank + 1))          # rank-dependent values: the average is 1.5x
        o += n
    views['2.bias'] = local['2.bias'] * (rank + 1)     # NOT inside the flat buffer
    ptrs = {k: v.data_ptr() for k, v in views.items()}
    gs2 = P.GradSync()
    gs2.finish(gs2.start(views, flat))
    assert gs2.collectives == 2 and gs2.bytes_reduced == (flat.numel() + local['2.bias'].numel()) * 4
    for k in views:
        assert views[k].data_ptr() == ptrs[k]
        assert torch.allclose(views[k], local[k] * 1.5, atol=1e-7), k
    t = P.max_over_ranks(float(rank + 1), torch.device('cpu'))
    assert t == float(world)
    if rank == 0:
        torch.save({k: v.grad for k, v in named.items() if v.grad is not None}, out)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_gradient_average_matches_full_batch(tmp_path):
    out = str(tmp_path / 'g.pt')
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    net = _toy()
    torch.manual_seed(1)
    x = torch.rand(4, 3, 8, 8)
    net(x).abs().mean().backward()
    for k, p in net.named_parameters():
        if p.grad is None:
            assert k not in got
            continue
        assert torch.allclose(got[k], p.grad, atol=1e-6), k


def test_shard_batch_rejects_ragged():
    sys.path.insert(0, ROOT)
    from highlyaccurate_amd import parallel as P
    import pytest
    with pytest.raises(ValueError):
        P.shard_batch(torch.zeros(5, 2), 0, 2)
    assert P.shard_batch(torch.arange(8).view(4, 2), 1, 2).tolist() == [[4, 5], [6, 7]]
