#!/usr/bin/env python3
"""The reference's training / evaluation call pattern (train_kitti.py:34-172, 319-423) on synthetic batches.

The reference's drivers never travel, and its datasets are not available, so this harness reproduces what they DO
with the model -- same argparse flag names and defaults (train_kitti.py:428-481), per-epoch Adam re-creation with
lr*(1-epoch/100) (329-333), zero_grad -> forward(mode='train') -> loss.backward() -> step (351-367), the progress
line built from the 14-tuple indices (373-406), the test loop that calls .backward() on the test outputs (49-64) and
what it leaves on disk -- `model_<epoch % 100>.pth` after every epoch (409-414), `Test1_results.mat` / `.txt` (82,
97-161), `Model_best.pth` when the selection score improves (163-170), `--resume N` loading `model_<N-1>.pth` (552-554)
and `--test 1` loading `model_1.pth` (545-548) -- against highlyaccurate_amd.models_kitti.LM_S2GP.
Multi-GPU: torchrun, batch sharded, GradSync.
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
    p.add_argument('--save_path', type=str, default='', help="checkpoint / result directory (the reference's getSavePath(args)); '' = write nothing")
    p.add_argument('--resume', type=int, default=0, help='resume from model_<resume-1>.pth and continue with epoch <resume> (train_kitti.py:552-554)')
    p.add_argument('--test', type=int, default=0, help='1: load model_1.pth, run the test loop, write the result files, no training (545-548)')
    p.add_argument('--test_batches', type=int, default=2)
    p.add_argument('--reference_compat', type=int, default=1,
                   help="1 (default: this harness mirrors the reference driver, so it picks Model_best.pth the way "
                        "train_kitti.py:160-168 does): the reference's N x N model-selection score; 0: the per-sample score "
                        "(metrics.localisation_metrics; both are printed either way)")
    return p.parse_args(argv)


def test1(net, args, save_path, best_rank_result, epoch, device, gen, rank=0):
    """train_kitti.py:34-170 on synthetic batches: eval mode, NO no_grad (the outputs carry grad and .backward() is called
    "to release the graph"), predictions gathered on the host, result files, Model_best.pth."""
    import numpy as np
    from highlyaccurate_amd.metrics import localisation_metrics, write_test_results
    net.eval()
    pred_shifts, pred_headings, gt_shifts, gt_headings = [], [], [], []
    t0 = time.time()
    for _ in range(args.test_batches):
        sat_map, grd_left_imgs, (gt_shift_u, gt_shift_v, gt_heading) = synthetic_batch(args, device, gen)
        shifts_lat, shifts_lon, theta = net(sat_map, grd_left_imgs, mode='test')
        shifts = torch.stack([shifts_lat, shifts_lon], dim=-1)
        headings = theta.unsqueeze(dim=-1)
        gt_shift = torch.cat([gt_shift_v, gt_shift_u], dim=-1)
        if args.shift_range_lat == 0 and args.shift_range_lon == 0:
            loss = torch.mean(headings - gt_heading)
        else:
            loss = torch.mean(shifts_lat - gt_shift_u)
        loss.backward()
        pred_shifts.append(shifts.data.cpu().numpy())
        pred_headings.append(headings.data.cpu().numpy())
        gt_shifts.append(gt_shift.data.cpu().numpy())
        gt_headings.append(gt_heading.data.cpu().numpy())
    duration = (time.time() - t0) / args.test_batches
    ps, ph = np.concatenate(pred_shifts, 0), np.concatenate(pred_headings, 0)
    gs, gh = np.concatenate(gt_shifts, 0), np.concatenate(gt_headings, 0)
    rc = bool(args.reference_compat)
    if save_path and rank == 0:
        result, stats = write_test_results(save_path, 'Test1', epoch, duration, ps, ph, gs, gh, args.shift_range_lat,
                                           args.shift_range_lon, args.rotation_range, reference_compat=rc)
    else:
        result, stats, _ = localisation_metrics(ps, ph, gs, gh, args.shift_range_lat, args.shift_range_lon,
                                                args.rotation_range, reference_compat=rc)
    if rank == 0:
        print('====================================\n       EPOCH: %d\nValidation results:\nInit distance average:  %s\n'
              'Pred distance average:  %s\nresult: %s' % (epoch, stats['init_distance_mean'], stats['pred_distance_mean'], result),
              flush=True)
    net.train()
    if result > best_rank_result and save_path and rank == 0:                       # train_kitti.py:165-168
        os.makedirs(save_path, exist_ok=True)
        torch.save(net.state_dict(), os.path.join(save_path, 'Model_best.pth'))
    return result


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
    # The model draws its out-of-range re-initialisation values from torch's GLOBAL CPU generator every LM step
    # (models_kitti.py:1028-1029): after the replicas have been built identically, give every rank its own stream
    # (seed base + rank), or all ranks would re-initialise their shards with the same numbers.
    if world > 1:
        torch.manual_seed(2022 + rank)
        import numpy as _np
        _np.random.seed(2022 + rank)          # args.dropout's keep masks come from numpy's global generator (968-974)
    log = []
    save_path = args.save_path
    if args.test:                                                                  # train_kitti.py:545-548
        net.load_state_dict(torch.load(os.path.join(save_path, 'model_1.pth'), map_location=device))
        test1(net, args, save_path, 0., 0, device, gen, rank)
        return log
    if args.resume:                                                                # train_kitti.py:552-554
        net.load_state_dict(torch.load(os.path.join(save_path, 'model_' + str(args.resume - 1) + '.pth'), map_location=device))
        if rank == 0:
            print('resume from ' + 'model_' + str(args.resume - 1) + '.pth')
    best = 0.0
    for epoch in range(args.resume, args.epochs):
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
        compNum = epoch % 100                                                      # train_kitti.py:409-414
        if save_path and rank == 0:
            os.makedirs(save_path, exist_ok=True)
            torch.save(net.state_dict(), os.path.join(save_path, 'model_' + str(compNum) + '.pth'))
        current = test1(net, args, save_path, best, epoch, device, gen, rank)      # train_kitti.py:417-419
        if current > best:
            best = current
    return log


if __name__ == '__main__':
    main()
