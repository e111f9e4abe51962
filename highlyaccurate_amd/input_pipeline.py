"""On-device version of the reference's dataset-side satellite-tile preparation.

``KITTI_dataset.py:128-157`` / ``Ford_dataset.py:185-209`` rotate, shift, randomly perturb and crop every satellite image
with four Pillow calls on CPU worker processes (2 workers, ``KITTI_dataset.py:32``) -- about two orders of magnitude slower
than the localisation path built here.  ``sat_tile_kitti`` / ``sat_tile_ford`` take the raw uint8 images on the GPU and
return the ``[B,3,512,512]`` float tensors the model consumes, bit-identical to the Pillow chain (``hla_sat_tile``)."""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib, utils

CAMERA_GPS_SHIFT_LEFT = (1.08, 0.26)          # utils.CameraGPS_shift_left (reference utils.py:14)


def _rotate_matrix(w, h, angle_deg):
    """Image.rotate's output->source affine matrix (cos/sin rounded to 15 digits, centre (w/2, h/2))."""
    ang = -math.radians(angle_deg % 360.0)
    c, s = round(math.cos(ang), 15), round(math.sin(ang), 15)
    cx, cy = w / 2.0, h / 2.0
    return [c, s, c * -cx + s * -cy + cx, -s, c, -s * -cx + c * -cy + cy]


def _fix(v):
    return float(math.floor(v * 65536.0 + 0.5))


def _nearest(m):
    return [0.0, _fix(m[0]), _fix(m[1]), _fix(m[2] + m[0] * 0.5 + m[1] * 0.5), _fix(m[3]), _fix(m[4]),
            _fix(m[5] + m[3] * 0.5 + m[4] * 0.5), 0.0]


def _bilinear(tx, ty):
    return [1.0, 1.0, 0.0, float(tx), 0.0, 1.0, float(ty), 0.0]


@_lib.on_device(lambda sat_u8, *a, **k: sat_u8)
def _run(sat_u8: torch.Tensor, stages: np.ndarray, crop: int) -> torch.Tensor:
    lib = _lib.load()
    _lib.require_gpu(sat_u8, 'sat_u8')
    if sat_u8.dtype != torch.uint8 or sat_u8.dim() != 4 or sat_u8.shape[3] != 3 or sat_u8.shape[1] != sat_u8.shape[2]:
        raise ValueError(f'expected uint8 [B,S,S,3] satellite images, got {sat_u8.dtype} {tuple(sat_u8.shape)}')
    B, S = sat_u8.shape[0], sat_u8.shape[1]
    st = torch.from_numpy(np.ascontiguousarray(stages, dtype=np.float64).reshape(B, 4, 8)).to(sat_u8.device)
    out = torch.empty(B, 3, crop, crop, device=sat_u8.device, dtype=torch.float32)
    rc = lib.hla_sat_tile(_lib.ptr(sat_u8.contiguous()), _lib.ptr(st), _lib.ptr(out), B, S, crop, _lib.stream_ptr())
    _lib.check(rc, 'hla_sat_tile')
    return out


def sat_tile_kitti(sat_u8, heading, gt_shift_x, gt_shift_y, theta, shift_range_lat=20.0, shift_range_lon=20.0,
                   rotation_range=10.0, crop=None):
    """sat_u8 [B,S,S,3] uint8 on the GPU; heading (rad, from the oxts file), gt_shift_x / gt_shift_y / theta in [-1,1]
    (the dataset's np.random.uniform draws): sequences of length B.  Returns [B,3,crop,crop] fp32 (KITTI_dataset.py:128-157).
    The labels the dataset returns are (-gt_shift_x, -gt_shift_y, theta)."""
    B, S = sat_u8.shape[0], sat_u8.shape[1]
    crop = crop or utils.get_process_satmap_sidelength()
    mpp = utils.get_meter_per_pixel(scale=1)
    lat_px, lon_px = shift_range_lat / mpp, shift_range_lon / mpp
    st = np.zeros((B, 4, 8))
    for b in range(B):
        st[b, 0] = _nearest(_rotate_matrix(S, S, -float(heading[b]) / np.pi * 180))
        st[b, 1] = _bilinear(CAMERA_GPS_SHIFT_LEFT[0] / mpp, CAMERA_GPS_SHIFT_LEFT[1] / mpp)
        st[b, 2] = _bilinear(float(gt_shift_x[b]) * lon_px, -float(gt_shift_y[b]) * lat_px)
        st[b, 3] = _nearest(_rotate_matrix(S, S, float(theta[b]) * rotation_range))
    return _run(sat_u8, st, crop)


def sat_tile_ford(sat_u8, b_delta_u, b_delta_v, yaw_deg, gt_shift_u, gt_shift_v, theta, shift_range_pixels_lat,
                  shift_range_pixels_lon, rotation_range=10.0, crop=512):
    """Ford_dataset.py:185-207: AFFINE shift to the body location, rotate by yaw, random AFFINE shift, random rotation, crop."""
    B, S = sat_u8.shape[0], sat_u8.shape[1]
    st = np.zeros((B, 4, 8))
    for b in range(B):
        st[b, 0] = _bilinear(float(b_delta_u[b]), float(b_delta_v[b]))
        st[b, 1] = _nearest(_rotate_matrix(S, S, float(yaw_deg[b])))
        st[b, 2] = _bilinear(float(gt_shift_u[b]) * shift_range_pixels_lat, float(gt_shift_v[b]) * shift_range_pixels_lon)
        st[b, 3] = _nearest(_rotate_matrix(S, S, float(theta[b]) * rotation_range))
    return _run(sat_u8, st, crop)


def _resample_tables(insize: int, outsize: int, device):
    """Pillow's antialiased triangle filter as integer taps scaled by 2^22: (bounds [n,2], taps [n,ksize], ksize)."""
    scale = insize / outsize
    filterscale = max(scale, 1.0)
    support = filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((outsize, 2), np.int32)
    taps = np.zeros((outsize, ksize), np.int32)
    for xx in range(outsize):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        cnt = min(int(center + support + 0.5), insize) - xmin
        k = []
        for x in range(cnt):
            v = abs((x + xmin - center + 0.5) / filterscale)
            k.append(1.0 - v if v < 1.0 else 0.0)
        ww = sum(k)
        if ww != 0.0:
            k = [w / ww for w in k]
        bounds[xx] = (xmin, cnt)
        taps[xx, :cnt] = [int(w * (1 << 22) + 0.5) for w in k]
    return torch.from_numpy(bounds).to(device), torch.from_numpy(taps).to(device), ksize


_table_cache = {}


@_lib.on_device(lambda grd_u8, *a, **k: grd_u8)
def grd_resize(grd_u8: torch.Tensor, out_h: int = 256, out_w: int = 1024) -> torch.Tensor:
    """grd_u8 [B,H,W,3] uint8 on the GPU (e.g. the 375x1242 KITTI frames) -> [B,3,out_h,out_w] fp32 in [0,1]:
    ``transforms.Resize([out_h, out_w])`` + ``ToTensor`` (KITTI_dataset.py:300-311), bit-identical to Pillow."""
    lib = _lib.load()
    _lib.require_gpu(grd_u8, 'grd_u8')
    if grd_u8.dtype != torch.uint8 or grd_u8.dim() != 4 or grd_u8.shape[3] != 3:
        raise ValueError(f'expected uint8 [B,H,W,3] images, got {grd_u8.dtype} {tuple(grd_u8.shape)}')
    B, H, W = grd_u8.shape[:3]
    key = (H, W, out_h, out_w, str(grd_u8.device))
    if key not in _table_cache:
        _table_cache[key] = _resample_tables(W, out_w, grd_u8.device) + _resample_tables(H, out_h, grd_u8.device)
    hb, ht, hk, vb, vt, vk = _table_cache[key]
    mid = torch.empty(B, H, out_w, 3, device=grd_u8.device, dtype=torch.uint8)
    out = torch.empty(B, 3, out_h, out_w, device=grd_u8.device, dtype=torch.float32)
    rc = lib.hla_resize_bilinear(_lib.ptr(grd_u8.contiguous()), _lib.ptr(hb), _lib.ptr(ht), hk, _lib.ptr(vb), _lib.ptr(vt), vk,
                                 _lib.ptr(mid), _lib.ptr(out), B, H, W, out_h, out_w, _lib.stream_ptr())
    _lib.check(rc, 'hla_resize_bilinear')
    return out
