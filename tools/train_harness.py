#!/usr/bin/env python3
"""The reference's training / evaluation call pattern (train_kitti.py:34-172, 319-423) on synthetic batches.

The reference's drivers never travel, and its datasets are not available, so this harness reproduces what they DO
with the model -- same argparse flag names and defaults (train_kitti.py:428-481), per-epoch Adam re-creation with
lr*(1-epoch/100) (329-333), zero_grad -> forward(mode='train') -> loss.backward() -> step (351-367), the progress
line built from the 14-tuple indices (373-406), and the test loop that calls .backward() on the test outputs
(49-64) -- against highlyaccurate_amd.models_kitti.LM_S2GP.  Multi-GPU: torchrun, batch sharded, GradSync.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--epochs', type=int, default=1)
    p.add_argument('--iters_per_epoch', type=int, default=4)
    p.add_argument('--lr', type=float, default=1e-4)
    p.add_argument('--rotation_range', type=float, default=10.)
    p.add_argument('--shift_range_lat', type=float, default=20.)
    p.add_argument('--shift_range_lon', type=float, default=20.)
    for k in ('coe_shift_lat', 'coe_shift_lon', 'coe_heading', 'coe_L1', 'coe_L2', 'coe_L3', 'coe_L4'):
        p.add_argument('--' + k, type=float, default=100.)
    p.add_argument('--batch_size', type=int, default=3)
    p.add_argument('--loss_method', type=int, default=0)
    p.add_argument('--level', type=int, default=3)
    p.add_argument('--N_iters', type=int, default=5)
    p.add_argument('--using_weight', type=int, default=0)
    p.add_argument('--damping', type=float, default=0.1)
    p.add_argument('--train_damping', type=int, default=0)
    p.add_argument('--direction', type=str, default='S2GP')
    p.add_argument('--Optimizer', type=str, default='LM')
    p.add_argument('--level_first', type=int, default=0)
    p.add_argument('--proj', type=str, default='geo')
    p.add_argument('--use_gt_depth', type=int, default=0)
    p.add_argument('--dropout', type=int, default=0)
    p.add_argument('--use_hessian', type=int, default=0)
    p.add_argument('--visualize', type=int, default=0)
    p.add_argument('--beta1', type=float, default=0.9)
    p.add_argument('--beta2', type=float, default=0.999)
    p.add_argument('--precision', type=str, default='bf16')
    p.add_argument('--grd_h', type=int, default=256)
    p.add_argument('--grd_w', type=int, default=1024)
    p.add_argument('--sat_a', type=int, default=512)
    return p.parse_args(argv)


def synthetic_batch(args, device, gen):
    B = args.batch_size
    sat = torch.rand(B, 3, args.sat_a, args.sat_a, generator=gen).to(device)
    grd = torch.rand(B, 3, args.grd_h, args.grd_w, generator=gen).to(device)
    gt = [(torch.rand(B, 1, generator=gen) * 2 - 1).to(device) for _ in range(3)]
    return sat, grd, gt


def main(argv=None):
    from highlyaccurate_amd.models_kitti import LM_S2GP, loss_func  # noqa: F401  (train_kitti.py:24)
    from highlyaccurate_amd import parallel as P
    args = parse_args(argv)
    rank, world, local = P.init_distributed()
    device = torch.device('cuda', local)
    net = eval('LM_' + args.direction)(args)                 # train_kitti.py:539
    net.to(device)                                           # train_kitti.py:542
    if world > 1:
        net.grad_sync = P.GradSync()
    gen = torch.Generator().manual_seed(2022 + rank)
    log = []
    for epoch in range(args.epochs):
        net.train()
        base_lr = args.lr * ((1.0 - float(epoch) / 100.0) ** 1.0)            # train_kitti.py:329
        optimizer = torch.optim.Adam(net.parameters(), lr=base_lr)           # re-created every epoch (333)
        optimizer.zero_grad()
        for Loop in range(args.iters_per_epoch):
            sat_map, grd_left_imgs, (gt_shift_u, gt_shift_v, gt_heading) = synthetic_batch(args, device, gen)
            t0 = time.time()
            optimizer.zero_grad()
            loss, loss_decrease, shift_lat_decrease, shift_lon_decrease, thetas_decrease, loss_last, \
                shift_lat_last, shift_lon_last, theta_last, L1, L2, L3, L4, grd_conf_list = \
                net(sat_map, grd_left_imgs, gt_shift_u, gt_shift_v, gt_heading, mode='train', file_name=None,
                    loop=Loop, level_first=args.level_first)
            loss.backward()
            optimizer.step()
            optimizer.zero_grad()
            level = args.level - 1
            line = ('Epoch: %d Loop: %d Delta: Level-%d: loss: %.4f lat: %.4f lon: %.4f rot: %.4f  Last: loss: %.4f lat: %.4f '
                    'lon: %.4f rot: %.4f  Time: %.3f' % (epoch, Loop, level, loss_decrease[level].item(),
                                                         shift_lat_decrease[level].item(), shift_lon_decrease[level].item(),
                                                         thetas_decrease[level].item(), loss_last[level].item(),
                                                         shift_lat_last[level].item(), shift_lon_last[level].item(),
                                                         theta_last[level].item(), time.time() - t0))
            log.append(float(loss.item()))
            if rank == 0:
                print(line, flush=True)
        # test1-style loop (train_kitti.py:34-100): no no_grad(), backward on the outputs "to release the graph"
        net.eval()
        sat_map, grd_left_imgs, (gt_shift_u, gt_shift_v, gt_heading) = synthetic_batch(args, device, gen)
        shifts_lat, shifts_lon, theta = net(sat_map, grd_left_imgs, mode='test')
        loss = torch.mean(shifts_lat - gt_shift_u)
        loss.backward()
        from highlyaccurate_amd.metrics import localisation_metrics
        shifts = torch.stack([shifts_lat, shifts_lon], dim=-1).data.cpu().numpy()          # train_kitti.py:55-70
        gt_shift = torch.cat([gt_shift_v, gt_shift_u], dim=-1).data.cpu().numpy()
        result, stats, lines = localisation_metrics(shifts, theta.unsqueeze(-1).data.cpu().numpy(), gt_shift,
                                                    gt_heading.data.cpu().numpy(), args.shift_range_lat, args.shift_range_lon,
                                                    args.rotation_range)                      # train_kitti.py:77-160
        if rank == 0:
            print('\n'.join(['====================================', '       EPOCH: ' + str(epoch), 'Validation results:',
                             'Init distance average:  %s' % stats['init_distance_mean'],
                             'Pred distance average:  %s' % stats['pred_distance_mean']] + lines), flush=True)
    return log


if __name__ == '__main__':
    main()
