"""Where does a fuzz case's deviation come from?  For one seed of tests/diag/fuzz_e2e.py: the extractor's maps (both
branches, every level, L2-normalised) of the HIP path in each fp32-class mode against the fp64 oracle's, then the LM trace
step by step.

    python tests/diag/fuzz_case_probe.py <seed> [precision ...]
"""
import os, sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_e2e as fz
from highlyaccurate_amd.VGG import vgg_forward_nhwc


def main():
    seed = int(sys.argv[1])
    precisions = sys.argv[2:] or ['fp32', 'fp16x3']
    for prec in precisions:
        os.environ['HLA_FUZZ_PRECISION'] = prec
        c = fz.case_setup(seed)
        net, onet = c['net'], c['onet']
        print(f"--- seed {seed} precision {prec}: fam {c['fam']} B{c['B']} grd {c['gh']}x{c['gw']} sat {c['sa']} {c['kw']}")
        with torch.no_grad():
            for name, img in (('SatFeatureNet', c['sat']), ('GrdFeatureNet', c['grd'])):
                of, oc = getattr(onet, name)(img.double())[:2]
                hf, hc, inv = vgg_forward_nhwc(getattr(net, name), img.to(fz.d), want_conf=True, defer_norm=False)
                for l, (a, b) in enumerate(zip(hf, of)):
                    b = b.permute(0, 2, 3, 1)                       # NCHW -> NHWC
                    a = a.double().cpu()[..., :b.shape[-1]]
                    err = (a - b).abs().amax(dim=(1, 2, 3)) / b.abs().amax(dim=(1, 2, 3))
                    cerr = (hc[l].double().cpu() - oc[l][:, 0]).abs().amax(dim=(1, 2)) if hc[l] is not None else None
                    print(f'  {name} level {l}: feature max err / max per sample {[f"{e:.1e}" for e in err.tolist()]}'
                          + (f', confidence {[f"{e:.1e}" for e in cerr.tolist()]}' if cerr is not None else ''))
            lfkw = {} if c['g2s'] else {'level_first': c['lf']}
            torch.manual_seed(seed)
            onet(c['sat'].double(), c['grd'].double(), *c['extra_o'], mode='test', **lfkw)
            torch.manual_seed(seed)
            net(c['sat'].to(fz.d), c['grd'].to(fz.d), *c['extra_g'], mode='test', **lfkw)
            tr = net.last_trace.double().cpu().numpy()
            otr = getattr(onet, 'trace', None)
            if otr is not None:
                lats, lons, thetas = otr                            # each [B, N_iters, levels]
                otr = torch.stack([lons, lats, thetas], -1).double().numpy().reshape(tr.shape)
                e = np.abs(tr - otr)
                print('  trace |HIP - fp64| per (iteration, level), worst over samples and components:')
                print('  ', np.array2string(e.max(axis=(0, -1)), precision=2))
                print('   worst sample', int(e.reshape(e.shape[0], -1).max(1).argmax()), 'per-sample worst', e.reshape(e.shape[0], -1).max(1))
            else:
                print('  (the oracle keeps no trace) HIP trace sample 0:', tr[0].reshape(-1, 3).round(6).tolist())


if __name__ == '__main__':
    main()
