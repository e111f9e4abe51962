"""Portable synthetic weights and inputs of the reference architecture (SURVEY 8(d) "Synthetic inputs").

There is no network for the pretrained VGG-16 checkpoint or the datasets, so benchmarks and the committed golden vectors
run on torchvision's NON-pretrained VGG init (Kaiming-normal, fan_out, ReLU; zero bias) and iid U[0,1) images, both drawn
from ``numpy.random.RandomState(seed)`` (the legacy generator is bit-stable across numpy versions), so a fixture only has
to hold OUTPUTS.  ``tests/test_oracle_golden.py`` pins these generators to the ones the goldens were recorded with.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
import torch

# (state-dict name, Cout, Cin, has_bias) in state-dict order -- VGG.py:23-81
VGG_LAYOUT = [
    ('conv0', 64, 3, True), ('conv2', 64, 64, True), ('conv5', 128, 64, True), ('conv7', 128, 128, True),
    ('conv10', 256, 128, True), ('conv12', 256, 256, True), ('conv14', 256, 256, True),
    ('conv_dec1.1', 128, 384, False), ('conv_dec1.3', 128, 128, False),
    ('conv_dec2.1', 64, 192, False), ('conv_dec2.3', 64, 64, False),
    ('conv_dec3.1', 32, 128, False), ('conv_dec3.3', 16, 32, False),
    ('conf0.1', 1, 256, False), ('conf1.1', 1, 128, False), ('conf2.1', 1, 64, False), ('conf3.1', 1, 16, False),
]


def reference_args(**kw) -> SimpleNamespace:
    """The argparse defaults of train_kitti.py:428-481 that the models read."""
    d = dict(level=3, N_iters=5, using_weight=0, loss_method=0, proj='geo', Optimizer='LM',
             rotation_range=10.0, shift_range_lat=20.0, shift_range_lon=20.0,
             damping=0.1, train_damping=0, dropout=0, use_hessian=0, use_gt_depth=0, visualize=0,
             coe_shift_lat=100.0, coe_shift_lon=100.0, coe_heading=100.0,
             coe_L1=100.0, coe_L2=100.0, coe_L3=100.0, coe_L4=100.0,
             beta1=0.9, beta2=0.999, estimate_depth=0, level_first=0)
    d.update(kw)
    return SimpleNamespace(**d)


def vgg_state(rs: np.random.RandomState, bias_scale: float = 0.0) -> dict:
    sd = {}
    for name, co, ci, has_b in VGG_LAYOUT:
        std = math.sqrt(2.0 / (co * 9))
        sd[name + '.weight'] = torch.from_numpy((rs.standard_normal((co, ci, 3, 3)) * std).astype(np.float32))
        if has_b:
            sd[name + '.bias'] = torch.from_numpy((rs.standard_normal((co,)) * bias_scale).astype(np.float32))
    return sd


def model_state(seed: int, bias_scale: float = 0.0, rotation_range: float = 10.0) -> dict:
    """The 49-tensor state dict of LM_S2GP / LM_S2GP_Ford / LM_G2SP (SURVEY B-1)."""
    rs = np.random.RandomState(seed)
    sd = {'damping': torch.zeros(1, 3) if rotation_range > 0 else torch.zeros(())}
    for net in ('SatFeatureNet', 'GrdFeatureNet'):
        for k, v in vgg_state(rs, bias_scale).items():
            sd[net + '.' + k] = v
    return sd


def images(seed: int, B: int, grd_hw=(256, 1024), sat_a=512):
    """iid U[0,1) 'white noise' images and U(-1,1) ground-truth poses: (sat, grd, gt_u, gt_v, gt_heading)."""
    rs = np.random.RandomState(seed)
    sat = torch.from_numpy(rs.random_sample((B, 3, sat_a, sat_a)).astype(np.float32))
    grd = torch.from_numpy(rs.random_sample((B, 3, grd_hw[0], grd_hw[1])).astype(np.float32))
    gt = torch.from_numpy(rs.uniform(-1, 1, size=(3, B, 1)).astype(np.float32))
    return sat, grd, gt[0], gt[1], gt[2]


def fixture_sample_idx(numel: int, salt: int, n: int = 64):
    """The deterministic sample positions of the committed feature / gradient fixtures (tests/make_idx.py, oracle/make_golden.py)."""
    return np.random.RandomState(1000 + salt).randint(0, numel, size=n)
