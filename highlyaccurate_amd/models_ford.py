"""``LM_S2GP_Ford`` -- Ford-AV model with the reference's surface (``models_ford.py:21-1036``) and
``loss_func`` (``models_ford.py:1041-1093``), executed by libhla."""
from __future__ import annotations

import torch

from ._s2gp import S2GPBase, loss_func, loss_from_trace  # noqa: F401


class LM_S2GP_Ford(S2GPBase):
    ford = True

    def forward(self, sat_map, grd_img_left, satmap_sidelength_meters, R_FL, T_FL,
                gt_shift_u=None, gt_shift_v=None, gt_theta=None, mode='train',
                file_name=None, level_first=0, loop=0, init_pose=None):
        """mode='test' -> (shift_u[B], shift_v[B], theta[B]) (models_ford.py:864-865);
        mode='train' -> 14-tuple (models_ford.py:858-862).  gt_* are [B] (float64 from the dataloader)."""
        want_conf = bool(self.using_weight) or mode == 'train'
        extra = dict(R_FL=R_FL, T_FL=T_FL, side_m=float(satmap_sidelength_meters))
        trace, grd_confs = self.localise(sat_map, grd_img_left, want_conf, extra, level_first, init_pose,
                                          return_confs=(mode == 'train'))
        us, vs, thetas = trace[..., 0], trace[..., 1], trace[..., 2]
        if mode == 'train':
            a = self.args
            coe_heading = 0 if a.rotation_range == 0 else a.coe_heading
            dev = trace.device
            # loss_func(us, vs, thetas, gt_shift_u, gt_shift_v, gt_theta, ...) of models_ford.py:837-839 on the trace's columns
            out = loss_from_trace(self.loss_method, trace, (0, 1, 2), gt_shift_u.to(dev), gt_shift_v.to(dev), gt_theta.to(dev),
                                  a.coe_shift_lat, a.coe_shift_lon, coe_heading)
            return (*out, [c.unsqueeze(1) for c in grd_confs])
        res = (us[:, -1, -1], vs[:, -1, -1], thetas[:, -1, -1])
        if torch.is_grad_enabled():
            res = tuple(r.clone().requires_grad_(True) for r in res)
        return res
