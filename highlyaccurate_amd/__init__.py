"""highlyaccurate_amd -- MI355X (gfx950) implementation of the HighlyAccurate per-pair localisation hot path
behind the reference's own module surface.

    from highlyaccurate_amd.models_kitti import LM_S2GP, loss_func
    from highlyaccurate_amd.models_ford import LM_S2GP_Ford
    from highlyaccurate_amd.VGG import VGGUnet
    from highlyaccurate_amd.jacobian import grid_sample
"""
__all__ = ['models_kitti', 'models_ford', 'VGG', 'jacobian', 'utils']
