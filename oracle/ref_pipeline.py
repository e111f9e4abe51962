"""CPU restatement of the dataset-side image geometry of the reference -- TEST INFRASTRUCTURE ONLY.

The reference prepares a satellite tile per sample with a chain of Pillow calls
(KITTI ``dataLoader/KITTI_dataset.py:128-157``: rotate -> AFFINE shift -> AFFINE random shift -> rotate -> centre crop;
Ford ``dataLoader/Ford_dataset.py:185-209``: AFFINE shift -> rotate -> AFFINE random shift -> rotate -> centre crop), each
stage producing a uint8 image.  Pillow (pinned here: 12.2.0, the version in this image) is the third-party library whose
arithmetic this module restates, from its documented behaviour and verified bit-for-bit against the library itself in
``tests/test_pipeline.py``:

  * ``Image.rotate(angle)`` (default NEAREST, expand=False): the affine matrix of ``Image.rotate`` (cos/sin rounded to 15
    digits, centre = (w/2, h/2)) applied by the 16.16 fixed-point nearest path: coefficients FIX(v) = floor(v*65536+0.5),
    constant = FIX(m2 + m0/2 + m1/2) (half-pixel offset folded in before the conversion), source index = value >> 16,
    outside -> 0.
  * ``Image.transform(size, AFFINE, m, resample=BILINEAR)``: source coordinate m.(x+0.5, y+0.5) in double; outside
    [0,W)x[0,H) -> 0; otherwise minus 0.5, floor, clamped neighbours, the row below only if it exists (else repeated),
    linear interpolation in double, TRUNCATED to uint8.
  * ``TF.center_crop``: top = left = int(round((S - crop) / 2)).
  * ``ToTensor``: float32(v) / 255.

Because every stage output pixel depends on at most 4 pixels of the previous stage, the final crop can be evaluated
lazily (16 source pixels per output pixel at most) with results identical to materialising every intermediate image.
"""
from __future__ import annotations

import math

import numpy as np


def rotate_matrix(w: int, h: int, angle_deg: float):
    """The 6 affine coefficients Image.rotate builds (output pixel -> source pixel)."""
    ang = -math.radians(angle_deg % 360.0)
    m = [round(math.cos(ang), 15), round(math.sin(ang), 15), 0.0, round(-math.sin(ang), 15), round(math.cos(ang), 15), 0.0]
    cx, cy = w / 2.0, h / 2.0
    m[2] = m[0] * -cx + m[1] * -cy + cx
    m[5] = m[3] * -cx + m[4] * -cy + cy
    return m


def fix16(v: float) -> int:
    return int(math.floor(v * 65536.0 + 0.5))


def nearest_fixed_coeffs(m):
    a = [fix16(v) for v in m]
    a[2] = fix16(m[2] + m[0] * 0.5 + m[1] * 0.5)          # the half-pixel offset is folded in BEFORE the conversion
    a[5] = fix16(m[5] + m[3] * 0.5 + m[4] * 0.5)
    return a


class Stage:
    """One resampling stage: kind 'N' (nearest, fixed point) or 'B' (bilinear, double) with its 6 coefficients."""

    def __init__(self, kind: str, m):
        self.kind = kind
        self.m = [float(v) for v in m]
        self.fix = nearest_fixed_coeffs(m) if kind == 'N' else None


def _eval(src: np.ndarray, stages, k: int, x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """uint8 values [...,3] of stage k's OUTPUT image at integer pixel coordinates (x, y); k = 0 is the source image.
    Coordinates must be inside the image."""
    if k == 0:
        return src[y, x]
    H, W = src.shape[:2]
    st = stages[k - 1]
    if st.kind == 'N':
        a = st.fix
        xi = (a[2] + a[1] * y.astype(np.int64) + a[0] * x.astype(np.int64)) >> 16
        yi = (a[5] + a[4] * y.astype(np.int64) + a[3] * x.astype(np.int64)) >> 16
        ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
        v = _eval(src, stages, k - 1, np.clip(xi, 0, W - 1), np.clip(yi, 0, H - 1))
        return np.where(ok[..., None], v, 0).astype(np.uint8)
    m = st.m
    xx, yy = x + 0.5, y + 0.5
    xin = m[0] * xx + m[1] * yy + m[2]
    yin = m[3] * xx + m[4] * yy + m[5]
    ok = (xin >= 0.0) & (xin < W) & (yin >= 0.0) & (yin < H)
    xin, yin = xin - 0.5, yin - 0.5
    x0, y0 = np.floor(xin).astype(np.int64), np.floor(yin).astype(np.int64)
    dx, dy = (xin - x0)[..., None], (yin - y0)[..., None]
    cx0, cx1 = np.clip(x0, 0, W - 1), np.clip(x0 + 1, 0, W - 1)
    cy0, cy1 = np.clip(y0, 0, H - 1), np.clip(y0 + 1, 0, H - 1)
    p00 = _eval(src, stages, k - 1, cx0, cy0).astype(np.float64)
    p01 = _eval(src, stages, k - 1, cx1, cy0).astype(np.float64)
    v1 = p00 + (p01 - p00) * dx
    p10 = _eval(src, stages, k - 1, cx0, cy1).astype(np.float64)
    p11 = _eval(src, stages, k - 1, cx1, cy1).astype(np.float64)
    v2 = p10 + (p11 - p10) * dx
    below = ((y0 + 1) >= 0) & ((y0 + 1) < H)
    v2 = np.where(below[..., None], v2, v1)
    v = v1 + (v2 - v1) * dy
    return np.where(ok[..., None], v.astype(np.uint8), 0).astype(np.uint8)      # C cast: truncation


def sat_tile(src_u8: np.ndarray, stages, crop: int) -> np.ndarray:
    """Centre crop (crop x crop) of the last stage's output, uint8 [crop,crop,3]."""
    S = src_u8.shape[0]
    assert src_u8.shape[1] == S and src_u8.dtype == np.uint8
    top = int(round((S - crop) / 2.0))
    ys, xs = np.mgrid[top:top + crop, top:top + crop]
    return _eval(src_u8, stages, len(stages), xs, ys)


def to_tensor(u8: np.ndarray) -> np.ndarray:
    """ToTensor: HWC uint8 -> CHW float32 / 255."""
    return (u8.astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1)


# ---- parameter chains of the two datasets --------------------------------------------------------------------------
KITTI_CAMERA_GPS_SHIFT_LEFT = (1.08, 0.26)        # utils.CameraGPS_shift_left


def kitti_stages(S: int, heading_rad: float, gt_shift_x: float, gt_shift_y: float, theta: float, meter_per_pixel: float,
                 shift_range_pixels_lat: float, shift_range_pixels_lon: float, rotation_range: float):
    """KITTI_dataset.py:128-150."""
    return [Stage('N', rotate_matrix(S, S, -heading_rad / np.pi * 180)),
            Stage('B', (1, 0, KITTI_CAMERA_GPS_SHIFT_LEFT[0] / meter_per_pixel, 0, 1, KITTI_CAMERA_GPS_SHIFT_LEFT[1] / meter_per_pixel)),
            Stage('B', (1, 0, gt_shift_x * shift_range_pixels_lon, 0, 1, -gt_shift_y * shift_range_pixels_lat)),
            Stage('N', rotate_matrix(S, S, theta * rotation_range))]


def ford_stages(S: int, b_delta_u: float, b_delta_v: float, yaw_deg: float, gt_shift_u: float, gt_shift_v: float,
                theta: float, shift_range_pixels_lat: float, shift_range_pixels_lon: float, rotation_range: float):
    """Ford_dataset.py:185-206."""
    return [Stage('B', (1, 0, b_delta_u, 0, 1, b_delta_v)),
            Stage('N', rotate_matrix(S, S, yaw_deg)),
            Stage('B', (1, 0, gt_shift_u * shift_range_pixels_lat, 0, 1, gt_shift_v * shift_range_pixels_lon)),
            Stage('N', rotate_matrix(S, S, theta * rotation_range))]


# ---- ground image: torchvision Resize([256,1024]) == Image.resize(BILINEAR) (KITTI_dataset.py:300-311) ---------------
RESAMPLE_PRECISION_BITS = 32 - 8 - 2


def resample_coeffs(insize: int, outsize: int):
    """Pillow's antialiased bilinear (triangle) filter: per output index the first input index, the tap count and the
    taps as integers scaled by 2^22 (rounded half away from zero)."""
    scale = insize / outsize
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    bounds, taps = [], []
    for xx in range(outsize):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), insize) - xmin
        k = []
        for x in range(xmax):
            v = abs((x + xmin - center + 0.5) / filterscale)
            k.append(1.0 - v if v < 1.0 else 0.0)
        ww = sum(k)
        k = [w / ww for w in k] if ww != 0.0 else k
        taps.append([int(w * (1 << RESAMPLE_PRECISION_BITS) + 0.5) for w in k])
        bounds.append((xmin, xmax))
    return bounds, taps


def _resample_axis1(a: np.ndarray, outw: int) -> np.ndarray:
    H, W, C = a.shape
    bounds, taps = resample_coeffs(W, outw)
    out = np.zeros((H, outw, C), np.uint8)
    for xx in range(outw):
        xmin, n = bounds[xx]
        ss = np.full((H, C), 1 << (RESAMPLE_PRECISION_BITS - 1), np.int64)
        for x in range(n):
            ss += a[:, xmin + x].astype(np.int64) * taps[xx][x]
        out[:, xx] = np.clip(ss >> RESAMPLE_PRECISION_BITS, 0, 255)
    return out


def resize_bilinear(a: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """Image.resize((out_w, out_h), BILINEAR): horizontal pass, uint8 intermediate, vertical pass."""
    t = _resample_axis1(a, out_w)
    return _resample_axis1(t.transpose(1, 0, 2), out_h).transpose(1, 0, 2)
