"""Host-side logic that must reproduce the reference's side effects exactly (no GPU needed)."""
import pytest
import torch


@pytest.mark.parametrize('B,steps', [(1, 15), (2, 30), (3, 1), (17, 15), (32, 15), (32, 30), (64, 4)])
def test_reinit_draws_are_the_references_rng_stream(B, steps):
    """models_kitti.py:1028-1029 / models_ford.py:461-462: every LM step draws rand_u, rand_v = Uniform(-1, 1).sample([B, 1])
    from torch's GLOBAL CPU generator.  `_s2gp.draw_reinit` takes them in one call: same values in the same order, and the
    generator ends in the same state (what a later consumer of that RNG -- a shuffling DataLoader -- sees)."""
    from highlyaccurate_amd._s2gp import draw_reinit
    torch.manual_seed(1234 + B)
    ref = []
    for _ in range(steps):
        ru = torch.distributions.uniform.Uniform(-1, 1).sample([B, 1])
        rv = torch.distributions.uniform.Uniform(-1, 1).sample([B, 1])
        ref.append(torch.stack([ru[:, 0], rv[:, 0]], 0))
    ref = torch.stack(ref, 0)
    after_ref = torch.rand(5)
    torch.manual_seed(1234 + B)
    got = draw_reinit(steps, B, 'cpu')
    after = torch.rand(5)
    assert got.shape == (steps, 2, B) and got.dtype == torch.float32
    assert torch.equal(got, ref) and torch.equal(after, after_ref)


def test_image_window_is_passed_without_a_copy():
    """mode='test' hands the ground extractor rows skip.. of the image: a view whose channel planes are a full image apart."""
    from highlyaccurate_amd.VGG import _image_window
    img = torch.rand(2, 3, 64, 48)
    x, plane = _image_window(img)
    assert x.data_ptr() == img.data_ptr() and plane == 64 * 48
    win = img[:, :, 24:, :]
    x, plane = _image_window(win)
    assert x.data_ptr() == win.data_ptr() and plane == 64 * 48 and tuple(x.shape) == (2, 3, 40, 48)
    # anything the kernels cannot address (a column window, a permuted image, another dtype) is made dense
    for odd in (img[:, :, :, 8:], img.permute(0, 1, 3, 2), img.double()):
        x, plane = _image_window(odd)
        assert x.is_contiguous() and x.dtype == torch.float32 and plane == x.shape[2] * x.shape[3]
        assert torch.equal(x, odd.float())


@pytest.mark.parametrize('family', ['kitti', 'ford'])
def test_product_ground_plane_tables_are_the_references_bit_for_bit(kat, family):
    """`_s2gp.ground_plane_table` / `S2GPBase.xyz_tables` -- the tables the HIP LM kernels read -- against the tables the REAL
    reference built (grd_img2cam, models_kitti.py:655-682 / models_ford.py:110-155; recorded by oracle/make_golden.py):
    bit-identical fp32, levels 0 and 2 (the level-2 fixture is stored on a stride-8 lattice)."""
    import numpy as np
    from oracle import ref_cpu as O
    from highlyaccurate_amd.models_kitti import LM_S2GP
    from highlyaccurate_amd.models_ford import LM_S2GP_Ford
    net = (LM_S2GP if family == 'kitti' else LM_S2GP_Ford)(O.default_args())
    tables = net.xyz_tables(256, 1024, 'cpu')
    for level, st in ((0, 1), (2, 8)):
        ref = kat[f'{family}_xyz_l{level}']
        got = tables[level].numpy()[None, ::st, ::st]
        assert got.dtype == np.float32 and got.shape == ref.shape
        np.testing.assert_array_equal(got, ref)
