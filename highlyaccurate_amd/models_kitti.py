"""``LM_S2GP`` -- KITTI satellite->ground localisation model with the reference's surface
(``models_kitti.py:598-1316``): ``LM_S2GP(args)``; ``forward(sat_map, grd_img_left, gt_shiftu, gt_shiftv,
gt_heading, mode, file_name, gt_depth, loop, level_first)``; 49-tensor state dict with identical keys.
The hot path (two VGG U-Nets, projection + Jacobian, N_iters x levels LM steps) runs in libhla (HIP, gfx950).
"""
from __future__ import annotations

import torch

from ._g2sp import LM_G2SP  # noqa: F401  (models_kitti.py:22-499)
from ._s2gp import S2GPBase, loss_func, loss_from_trace  # noqa: F401  (loss_func re-exported like the reference module)


class LM_S2GP(S2GPBase):
    ford = False

    def forward(self, sat_map, grd_img_left, gt_shiftu=None, gt_shiftv=None, gt_heading=None, mode='train',
                file_name=None, gt_depth=None, loop=0, level_first=0, init_pose=None):
        """sat_map [B,3,A,A], grd_img_left [B,3,H,W] fp32 in [0,1] on the GPU.
        mode='test'  -> (shift_lat[B], shift_lon[B], theta[B])   (models_kitti.py:1316)
        mode='train' -> the reference's 14-tuple                 (models_kitti.py:1312-1314)
        ``init_pose`` [B,3] (shift_u, shift_v, heading) is an extension; the reference always starts at 0."""
        if gt_depth is not None and getattr(self.args, 'use_gt_depth', 0):
            raise NotImplementedError('projection with a ground-truth depth map (models_kitti.py:741-747) is out of scope')
        want_conf = bool(self.using_weight) or mode == 'train'
        trace, grd_confs = self.localise(sat_map, grd_img_left, want_conf, None, level_first, init_pose,
                                          return_confs=(mode == 'train'))
        shift_lons, shift_lats, thetas = trace[..., 0], trace[..., 1], trace[..., 2]   # models_kitti.py:1281-1283
        if mode == 'train':
            a = self.args
            coe_heading = 0 if a.rotation_range == 0 else a.coe_heading
            # loss_func(shift_lats, shift_lons, thetas, gt_shiftv[:, 0], gt_shiftu[:, 0], gt_heading[:, 0], ...) of
            # models_kitti.py:1304-1310, on the trace's columns (lat = 1, lon = 0, theta = 2)
            out = loss_from_trace(self.loss_method, trace, (1, 0, 2), gt_shiftv[:, 0], gt_shiftu[:, 0], gt_heading[:, 0],
                                  a.coe_shift_lat, a.coe_shift_lon, coe_heading)
            return (*out, [c.unsqueeze(1) for c in grd_confs])
        res = (shift_lats[:, -1, -1], shift_lons[:, -1, -1], thetas[:, -1, -1])
        if torch.is_grad_enabled():
            # train_kitti.py:63-64 calls .backward() on the test outputs "to release the graph"
            res = tuple(r.clone().requires_grad_(True) for r in res)
        return res
