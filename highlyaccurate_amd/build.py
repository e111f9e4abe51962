"""Build libhla.so (gfx950) in-tree with hipcc.  `python -m highlyaccurate_amd.build [--force] [--verbose] [--out=NAME.so] [-DX=1]`
(A/B tooling: `-DX=<flag>` passes <flag> to hipcc verbatim, e.g. `-DX=-mllvm -DX=-amdgpu-sched-strategy=max-ilp --out=libhla_ilp.so`.)"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libhla.so')
# (source, HLA_TU_DTYPE): the conv-heavy files are compiled once per dtype (kernels) plus once as the dispatcher (-1)
SOURCES = [('capi.hip', -1), ('prof.hip', -1), ('lm_solve.hip', -1), ('lm_backward.hip', -1), ('lm_g2s.hip', -1),
           ('grid_sample.hip', -1), ('sat_tile.hip', -1), ('pose_loss.hip', -1), ('fill.hip', -1)] + \
          [(f, d) for f in ('vgg.hip', 'vgg_backward.hip') for d in (0, 1, 2, 3, -1)]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=fast', '-munsafe-fp-atomics', '-Wno-unused-result']


HASH_MARK = b'HLA_SOURCE_HASH='


class SourcesMissing(RuntimeError):
    pass


def source_hash() -> str:
    """sha256 over everything the library is built from (csrc/*, include/hla.h, the compiler flags).  The build bakes it
    into the binary (``hla_source_hash()``), so a stale libhla.so is detected by CONTENT: file times do not survive the
    copy to the GPU box, and *.so is git-ignored but shipped prebuilt."""
    import hashlib
    h = hashlib.sha256(' '.join(FLAGS).encode())
    try:
        deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.h')))
        for d in deps + [os.path.join(HERE, '..', 'include', 'hla.h')]:
            h.update(os.path.basename(d).encode() + b'\0')
            h.update(open(d, 'rb').read())
    except OSError as e:        # a prebuilt library shipped without csrc/ or the repo-root include/: there is nothing to compare with
        raise SourcesMissing(f'cannot hash the library sources ({e.filename}: {e.strerror}); the tree needs '
                             f'highlyaccurate_amd/csrc/*.hip,*.h and include/hla.h next to the package') from None
    return h.hexdigest()


def lib_hash(path: str = None):
    """The source hash baked into a built library, read from the file (no dlopen), or None."""
    path = path or LIB
    try:
        blob = open(path, 'rb').read()
    except OSError:
        return None
    k = blob.find(HASH_MARK)
    if k < 0:
        return None
    return blob[k + len(HASH_MARK):k + len(HASH_MARK) + 64].decode('ascii', 'replace')


def have_compiler() -> bool:
    return os.path.exists(os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'))


def _stale(path: str = None) -> bool:
    return lib_hash(path or LIB) != source_hash()


def _resource_table(text: str) -> None:
    """Condense -Rpass-analysis=kernel-resource-usage remarks into one line per kernel."""
    import re
    cur = None
    rows = {}
    for line in text.splitlines():
        m = re.search(r'remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|'
                      r'VGPRs Spill|LDS Size \[bytes/block\]|SGPRs): (\S+)', line)
        if not m:
            if 'warning' in line or 'error' in line:
                print(line)
            continue
        k, v = m.group(1), m.group(2)
        if k == 'Function Name':
            cur = v
            rows[cur] = {}
        elif cur:
            rows[cur][k.split(' ')[0] + ('Spill' if 'Spill' in k else '')] = v
    for name, r in rows.items():
        try:
            name = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()[:90]
        except Exception:
            pass
        print(f"  {name:<90} vgpr {r.get('VGPRs','?'):>3} agpr {r.get('AGPRs','?'):>3} sgpr {r.get('SGPRs','?'):>3} "
              f"spill {r.get('VGPRsSpill','?'):>3} scratch {r.get('ScratchSize','?'):>4} occ {r.get('Occupancy','?')} "
              f"lds {r.get('LDS','?')}")


def build(force: bool = False, verbose: bool = False, out: str = None, defines=()) -> str:
    """Build highlyaccurate_amd/libhla.so.  ``out`` = another file name (an A/B build kept next to the product library and
    selected at run time with HLA_LIB=<path>, e.g. the previous commit's kernels for a same-box comparison: tools/ab_libs.py);
    ``defines`` = extra -D flags for such a build."""
    global LIB
    if out:
        LIB = out if os.path.isabs(out) else os.path.join(HERE, out)
        force = True
    if not force and not _stale():
        return LIB
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    # One builder at a time: under torchrun every rank reaches a stale library at once, and all of them would write the same
    # build/*.o (only the final link is atomic) -- a rank could link another rank's half-written object.  Whoever gets the lock
    # second finds the library fresh and returns.
    import fcntl
    with open(os.path.join(HERE, 'build', '.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not _stale():
            return LIB
        return _build_locked(verbose, tuple(defines))


def _prune_objects() -> int:
    """Delete build/*.o of libraries that no longer exist next to the package (objects are named <source>.<library>.t<tu>.o; A/B
    builds made with --out= leave 17 of them each, 4-5 MB a piece).  Objects of an existing library are kept: they are what makes a
    rebuild of one translation unit cheap."""
    bdir = os.path.join(HERE, 'build')
    have = {f for f in os.listdir(HERE) if f.endswith('.so')} | {os.path.basename(LIB)}
    n = 0
    for f in os.listdir(bdir):
        if not f.endswith('.o'):
            continue
        parts = f.split('.')           # <source>, <library name pieces...>, 't<k>', 'o'
        lib = '.'.join(parts[1:-2])
        if lib not in have:
            try:
                os.remove(os.path.join(bdir, f))
                n += 1
            except OSError:
                pass
    return n


def _build_locked(verbose: bool, defines: tuple) -> str:
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    srchash = source_hash()
    _prune_objects()
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    for s, tu in SOURCES:
        o = os.path.join(HERE, 'build', s.replace('.hip', f'.{os.path.basename(LIB)}.t{tu + 1}.o'))
        cmd = [hipcc, *FLAGS, *[(d[2:] if d.startswith('X=') else f'-D{d}') for d in defines], f'-DHLA_TU_DTYPE={tu}', '-c', os.path.join(CSRC, s), '-o', o]
        if s == 'capi.hip':
            cmd.insert(-4, f'-DHLA_SOURCE_HASH_HEX="{srchash}"')
        if verbose:
            cmd.insert(1, '-Rpass-analysis=kernel-resource-usage')
            print(' '.join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(o)
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            print(out)
        elif verbose:
            _resource_table(out)
        if p.returncode:
            raise RuntimeError(f'hipcc failed on {s}')
    # link to a temporary name and rename: a concurrent loader never sees a half-written library
    tmp = LIB + f'.tmp{os.getpid()}'
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', tmp]
    subprocess.run(cmd, check=True)
    os.replace(tmp, LIB)
    return LIB


if __name__ == '__main__':
    _o = [a.split('=', 1)[1] for a in sys.argv if a.startswith('--out=')]
    _d = [a[2:] for a in sys.argv if a.startswith('-D')]
    print(build(force='--force' in sys.argv, verbose='--verbose' in sys.argv, out=_o[0] if _o else None, defines=_d))
