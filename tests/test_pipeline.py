"""Dataset-side satellite-tile geometry.  CPU: the numpy restatement (oracle/ref_pipeline.py) against
Pillow itself running the same call chain as the reference's datasets; GPU: the HIP kernel against both."""
import numpy as np
import pytest
import torch

from oracle import ref_pipeline as RP

PIL = pytest.importorskip('PIL')
from PIL import Image  # noqa: E402


def _pil_chain(a, stages, crop):
    """The reference's Pillow calls (KITTI_dataset.py:128-152 / Ford_dataset.py:185-207) for a list of
    ('rotate', deg) / ('shift', tx, ty) operations, then TF.center_crop."""
    im = Image.fromarray(a)
    for op in stages:
        if op[0] == 'rotate':
            im = im.rotate(op[1])
        else:
            im = im.transform(im.size, Image.AFFINE, (1, 0, op[1], 0, 1, op[2]), resample=Image.BILINEAR)
    S = a.shape[0]
    top = int(round((S - crop) / 2.0))
    return np.array(im.crop((top, top, top + crop, top + crop)))


def _cases(rs, n):
    mpp = 0.2 * 512 / 1280 * 0 + 0.07843137                   # ~ utils.get_meter_per_pixel(scale=1) magnitude
    for _ in range(n):
        heading = rs.uniform(-np.pi, np.pi)
        sx, sy, th = rs.uniform(-1, 1, 3)
        yield heading, sx, sy, th, mpp


@pytest.mark.parametrize('S,crop', [(160, 64), (320, 128)])
def test_oracle_pipeline_matches_pillow_bit_for_bit(S, crop):
    rs = np.random.RandomState(S)
    a = rs.randint(0, 256, (S, S, 3), dtype=np.uint8)
    lat_px = lon_px = 20.0 / 0.2 * S / 1280                    # shift range in pixels, scaled to the test image
    for heading, sx, sy, th, mpp in _cases(rs, 4):
        st = RP.kitti_stages(S, heading, sx, sy, th, mpp * 1280 / S, lat_px, lon_px, 10.0)
        ref = _pil_chain(a, [('rotate', -heading / np.pi * 180),
                             ('shift', RP.KITTI_CAMERA_GPS_SHIFT_LEFT[0] / (mpp * 1280 / S), RP.KITTI_CAMERA_GPS_SHIFT_LEFT[1] / (mpp * 1280 / S)),
                             ('shift', sx * lon_px, -sy * lat_px), ('rotate', th * 10.0)], crop)
        got = RP.sat_tile(a, st, crop)
        assert np.array_equal(got, ref), (heading, sx, sy, th, int((got != ref).sum()))
        du, dv, yaw = rs.uniform(-30, 30), rs.uniform(-30, 30), rs.uniform(-180, 180)
        st = RP.ford_stages(S, du, dv, yaw, sx, sy, th, lat_px, lon_px, 10.0)
        ref = _pil_chain(a, [('shift', du, dv), ('rotate', yaw), ('shift', sx * lat_px, sy * lon_px), ('rotate', th * 10.0)], crop)
        got = RP.sat_tile(a, st, crop)
        assert np.array_equal(got, ref), ('ford', int((got != ref).sum()))


def test_oracle_pipeline_full_size_matches_pillow():
    """One sample at the real sizes: 1280x1280 satellite image -> 512x512 tile."""
    rs = np.random.RandomState(11)
    S, crop = 1280, 512
    a = rs.randint(0, 256, (S, S, 3), dtype=np.uint8)
    mpp = 0.07833140346
    heading, sx, sy, th = 1.2345, 0.61, -0.37, -0.83
    st = RP.kitti_stages(S, heading, sx, sy, th, mpp, 20.0 / mpp, 20.0 / mpp, 10.0)
    ref = _pil_chain(a, [('rotate', -heading / np.pi * 180), ('shift', 1.08 / mpp, 0.26 / mpp),
                         ('shift', sx * 20.0 / mpp, -sy * 20.0 / mpp), ('rotate', th * 10.0)], crop)
    assert np.array_equal(RP.sat_tile(a, st, crop), ref)


@pytest.mark.gpu
@pytest.mark.parametrize('S,crop,B', [(320, 128, 3), (1280, 512, 2)])
def test_hip_sat_tile_matches_pillow_and_oracle_bit_for_bit(S, crop, B):
    from highlyaccurate_amd import utils
    from highlyaccurate_amd.input_pipeline import sat_tile_ford, sat_tile_kitti
    assert torch.cuda.is_available()
    d = torch.device('cuda:0')
    rs = np.random.RandomState(S + B)
    a = rs.randint(0, 256, (B, S, S, 3), dtype=np.uint8)
    heading = rs.uniform(-np.pi, np.pi, B)
    sx, sy, th = rs.uniform(-1, 1, B), rs.uniform(-1, 1, B), rs.uniform(-1, 1, B)
    got = sat_tile_kitti(torch.from_numpy(a).to(d), heading, sx, sy, th, 20.0, 20.0, 10.0, crop=crop).cpu().numpy()
    mpp = utils.get_meter_per_pixel(scale=1)
    for b in range(B):
        ref = _pil_chain(a[b], [('rotate', -heading[b] / np.pi * 180), ('shift', 1.08 / mpp, 0.26 / mpp),
                                ('shift', sx[b] * 20.0 / mpp, -sy[b] * 20.0 / mpp), ('rotate', th[b] * 10.0)], crop)
        assert np.array_equal(got[b], RP.to_tensor(ref)), ('kitti', b, int((got[b] != RP.to_tensor(ref)).sum()))
        st = RP.kitti_stages(S, heading[b], sx[b], sy[b], th[b], mpp, 20.0 / mpp, 20.0 / mpp, 10.0)
        assert np.array_equal(got[b], RP.to_tensor(RP.sat_tile(a[b], st, crop)))
    du, dv, yaw = rs.uniform(-40, 40, B), rs.uniform(-40, 40, B), rs.uniform(-180, 180, B)
    got = sat_tile_ford(torch.from_numpy(a).to(d), du, dv, yaw, sx, sy, th, 90.9, 90.9, 10.0, crop=crop).cpu().numpy()
    for b in range(B):
        ref = _pil_chain(a[b], [('shift', du[b], dv[b]), ('rotate', yaw[b]), ('shift', sx[b] * 90.9, sy[b] * 90.9),
                                ('rotate', th[b] * 10.0)], crop)
        assert np.array_equal(got[b], RP.to_tensor(ref)), ('ford', b, int((got[b] != RP.to_tensor(ref)).sum()))


@pytest.mark.parametrize('H,W,oh,ow', [(75, 124, 52, 100), (375, 1242, 256, 1024), (86, 165, 26, 100), (64, 64, 100, 90)])
def test_oracle_resize_matches_pillow_bit_for_bit(H, W, oh, ow):
    a = np.random.RandomState(H).randint(0, 256, (H, W, 3), dtype=np.uint8)
    ref = np.array(Image.fromarray(a).resize((ow, oh), Image.BILINEAR))
    assert np.array_equal(RP.resize_bilinear(a, oh, ow), ref)


@pytest.mark.gpu
@pytest.mark.parametrize('H,W,oh,ow,B', [(375, 1242, 256, 1024, 2), (860, 1656, 256, 1024, 1), (75, 124, 52, 100, 3)])
def test_hip_grd_resize_matches_pillow_bit_for_bit(H, W, oh, ow, B):
    from highlyaccurate_amd.input_pipeline import grd_resize
    d = torch.device('cuda:0')
    a = np.random.RandomState(W).randint(0, 256, (B, H, W, 3), dtype=np.uint8)
    got = grd_resize(torch.from_numpy(a).to(d), oh, ow).cpu().numpy()
    for b in range(B):
        ref = np.array(Image.fromarray(a[b]).resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(got[b], RP.to_tensor(ref)), (b, int((got[b] != RP.to_tensor(ref)).sum()))
