"""One rank of the data-parallel model test (tests/test_gpu_parity.py::test_two_rank_real_model_gradients_match_full_batch).
Launched as `python dp_worker.py <out_dir>` with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment.  All ranks
share the one GPU of the box, so the process group is gloo (the collective code path above it is the one RCCL runs)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def build_case(kw):
    """The same model, global batch and targets on every rank and in the single-process run."""
    from oracle import ref_cpu as O
    args = O.default_args(N_iters=2, precision='fp32', **kw)
    sd = O.synth_model_state(4, bias_scale=0.02)
    sat, grd, gu, gv, gh = O.synth_images(44, 4, grd_hw=(64, 256), sat_a=128)
    return args, sd, (sat, grd, gu, gv, gh)


def run(net, batch, dev):
    sat, grd, gu, gv, gh = [t.to(dev) for t in batch]
    net.zero_grad(set_to_none=True)
    torch.manual_seed(0)
    out = net(sat, grd, gu, gv, gh, mode='train')
    out[0].backward()
    return {k: (p.grad.detach().cpu() if p.grad is not None else None) for k, p in net.named_parameters()}, float(out[0])


def main():
    import json
    from highlyaccurate_amd import parallel as P
    from highlyaccurate_amd.models_kitti import LM_S2GP
    out_dir, kw = sys.argv[1], json.loads(sys.argv[2])
    rank, world, _ = P.init_distributed('gloo')
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    args, sd, batch = build_case(kw)
    net = LM_S2GP(args)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    net.grad_sync = P.GradSync()
    shard = [P.shard_batch(t, rank, world) for t in batch]
    grads, loss = run(net, shard, dev)
    torch.save({'grads': grads, 'loss': loss, 'bytes': net.grad_sync.bytes_reduced, 'collectives': net.grad_sync.collectives},
               os.path.join(out_dir, f'rank{rank}.pt'))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
