"""``grid_sample(image, optical, jac=None)`` with the reference's signature (jacobian.py:138-205),
executed by libhla's ``hla_grid_sample``.  Inputs/outputs are NCHW-shaped like the reference's;
internally the image is read channels-last."""
from __future__ import annotations

import torch

from . import _lib


@_lib.on_device(lambda image, *a, **k: image)
def grid_sample(image: torch.Tensor, optical: torch.Tensor, jac: torch.Tensor | None = None):
    """image [N,C,IH,IW]; optical [N,H,W,2] pixel coordinates (x, y); jac [M,N,H,W,2] or None.
    Returns (out [N,C,H,W], jac_out [M,N,C,H,W] or None).  Out-of-bounds samples (and samples exactly on
    the last row/column) are 0, as in the reference."""
    _lib.require_gpu(image, 'grid_sample image')
    lib = _lib.load()
    N, Cc, IH, IW = image.shape
    _, H, W, _ = optical.shape
    img = image.float().permute(0, 2, 3, 1).contiguous()       # NHWC (no copy if already channels-last)
    opt = optical.float().contiguous()
    out = torch.empty(N, H, W, Cc, device=image.device, dtype=torch.float32)
    M = 0
    jin = jout = None
    if jac is not None:
        M = jac.shape[0]
        jin = jac.float().contiguous()
        jout = torch.empty(M, N, H, W, Cc, device=image.device, dtype=torch.float32)
    if not bool(((opt[..., 0] >= 0) & (opt[..., 0] <= IW - 1) & (opt[..., 1] >= 0) & (opt[..., 1] <= IH - 1)).any()):
        raise AssertionError('grid_sample: no sample falls inside the image')   # jacobian.py:172
    rc = lib.hla_grid_sample(_lib.ptr(img), _lib.ptr(opt), _lib.ptr(jin), _lib.ptr(out), _lib.ptr(jout),
                             N, Cc, IH, IW, H, W, M, _lib.stream_ptr())
    _lib.check(rc, 'hla_grid_sample')
    return out.permute(0, 3, 1, 2), (jout.permute(0, 1, 4, 2, 3) if jout is not None else None)
