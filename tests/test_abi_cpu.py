"""CPU-only checks of the C-ABI boundary: libhla.so builds (hipcc cross-compiles gfx950 without a GPU), loads, and
exports every function include/hla.h declares; the ctypes layer binds each of them; no compute call is made."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'hla.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(hla_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_the_documented_entry_points():
    names = _declared()
    for must in ('hla_vgg_forward', 'hla_vgg_backward', 'hla_s2g_lm_solve', 'hla_s2g_lm_solve_bwd', 'hla_g2s_lm_solve',
                 'hla_g2s_lm_solve_bwd', 'hla_grid_sample', 'hla_sat_tile', 'hla_resize_bilinear', 'hla_last_error',
                 'hla_abi_version'):
        assert must in names, must


def test_library_builds_loads_and_exports_every_declared_symbol():
    from highlyaccurate_amd import build as B
    path = B.build()                                  # no-op when the in-tree library is up to date
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, missing
    lib.hla_abi_version.restype = ctypes.c_int
    from highlyaccurate_amd import _lib
    assert lib.hla_abi_version() == _lib.ABI_VERSION


def test_ctypes_layer_binds_every_declared_symbol_and_rejects_cpu_tensors():
    import torch
    from highlyaccurate_amd import _lib
    lib = _lib.load()
    for n in _declared():
        fn = getattr(lib, n)
        if n not in ('hla_last_error', 'hla_abi_version', 'hla_source_hash', 'hla_prof_kernel_name'):
            assert fn.argtypes is not None, f'{n}: argtypes not declared in _lib.py'
    from highlyaccurate_amd.VGG import VGGUnet
    with pytest.raises(_lib.HlaError):                # the product path has no CPU fallback
        VGGUnet(3)(torch.zeros(1, 3, 32, 64))


def test_header_is_plain_c_and_links_from_a_c_program(tmp_path):
    """The boundary is a C ABI: include/hla.h must compile as C99 (no C++ or torch types) and a C program must link
    against libhla.so and call a non-compute entry point."""
    import shutil
    import subprocess
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc')
    hdr = os.path.join(ROOT, 'include', 'hla.h')
    subprocess.run([gcc, '-std=c99', '-pedantic', '-Wall', '-Werror', '-fsyntax-only', '-x', 'c', hdr], check=True)
    from highlyaccurate_amd import build as B
    lib_dir = os.path.dirname(B.build())
    src = tmp_path / 'user.c'
    src.write_text('#include "hla.h"\n'
                   'int main(void) { hla_s2g_config c; hla_s2g_level l[3]; hla_vgg_params p; (void)c; (void)l; (void)p;\n'
                   '  return (hla_abi_version() > 0 && hla_last_error() != 0) ? 0 : 1; }\n')
    exe = tmp_path / 'user'
    subprocess.run([gcc, '-std=c99', '-Wall', '-I', os.path.join(ROOT, 'include'), str(src), '-L', lib_dir, '-l:libhla.so',
                    f'-Wl,-rpath,{lib_dir}', '-o', str(exe)], check=True)
    assert subprocess.run([str(exe)]).returncode == 0


def _fresh_loader(monkeypatch):
    from highlyaccurate_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)           # force load() to run its checks again (restored afterwards)
    return _lib


def test_stale_binary_is_refused_or_rebuilt(monkeypatch, tmp_path):
    """*.so is git-ignored but shipped prebuilt, so a binary built from OTHER sources must never be called: load() compares
    the content hash baked into the library with the hash of the sources next to it."""
    import shutil
    from highlyaccurate_amd import build as B
    good = B.build()
    assert B.lib_hash(good) == B.source_hash() and not B._stale()
    # a library whose baked-in hash differs from the sources (here: flip one hex digit inside a copy of the file)
    stale = tmp_path / 'libhla.so'
    blob = bytearray(open(good, 'rb').read())
    k = blob.find(B.HASH_MARK) + len(B.HASH_MARK)
    blob[k] = ord('0') if blob[k] != ord('0') else ord('1')
    stale.write_bytes(bytes(blob))
    assert B.lib_hash(str(stale)) != B.source_hash() and B._stale(str(stale))
    _lib = _fresh_loader(monkeypatch)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(stale))
    monkeypatch.setattr(B, 'have_compiler', lambda: False)
    with pytest.raises(_lib.HlaError, match='stale'):
        _lib.load()
    # with a compiler present the loader rebuilds instead of calling the stale binary
    called = []
    monkeypatch.setattr(B, 'have_compiler', lambda: True)
    monkeypatch.setattr(B, 'build', lambda force=False, **kw: called.append(force) or shutil.copy(good, str(stale)))
    _lib.load()
    assert called == [False]       # build() itself re-checks staleness under its file lock: concurrent ranks build once
    # a missing library without a compiler: loud, no fallback
    _lib = _fresh_loader(monkeypatch)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nothing.so'))
    monkeypatch.setattr(B, 'have_compiler', lambda: False)
    with pytest.raises(_lib.HlaError, match='missing'):
        _lib.load()


def test_prebuilt_library_without_sources_loads_and_missing_everything_is_loud(monkeypatch, tmp_path):
    """A deployment that ships libhla.so without csrc/ or the repo-root include/ has nothing to hash: the library is taken as it
    is (ABI version and struct sizes are still checked); with neither sources nor library the error says so."""
    from highlyaccurate_amd import build as B
    good = B.build()

    def no_sources():
        raise B.SourcesMissing('cannot hash the library sources (include/hla.h: No such file or directory)')
    _lib = _fresh_loader(monkeypatch)
    monkeypatch.setattr(B, 'source_hash', no_sources)
    monkeypatch.setattr(B, 'have_compiler', lambda: False)
    assert _lib.load().hla_abi_version() == _lib.ABI_VERSION
    _lib = _fresh_loader(monkeypatch)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nothing.so'))
    with pytest.raises(_lib.HlaError, match='missing and so are its sources'):
        _lib.load()
    # and source_hash() itself reports a missing file as SourcesMissing, not a bare FileNotFoundError
    monkeypatch.undo()
    monkeypatch.setattr(B, 'CSRC', str(tmp_path / 'no_csrc'))
    with pytest.raises(B.SourcesMissing):
        B.source_hash()
    assert good


def test_abi_version_and_struct_sizes_are_enforced_at_load(monkeypatch):
    import ctypes as C
    _lib = _fresh_loader(monkeypatch)
    monkeypatch.setattr(_lib, 'ABI_VERSION', _lib.ABI_VERSION + 1)
    with pytest.raises(_lib.HlaError, match='ABI version'):
        _lib.load()
    _lib = _fresh_loader(monkeypatch)
    monkeypatch.undo()
    _lib = _fresh_loader(monkeypatch)

    class Wider(C.Structure):                          # a binding written against an older/newer struct layout
        _fields_ = list(_lib.S2GConfig._fields_) + [('extra', C.c_double)]
    monkeypatch.setattr(_lib, 'S2GConfig', Wider)
    with pytest.raises(_lib.HlaError, match='sizeof'):
        _lib.load()
