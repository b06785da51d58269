"""CPU-only: the C-ABI library loads without a GPU and exports exactly what include/tdk_hip.h
declares; the ctypes table mirrors the header."""

import ctypes
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / 'include' / 'tdk_hip.h'


def _declared_functions():
    text = re.sub(r'/\*.*?\*/', '', HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r'\b(tdk_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_the_expected_surface():
    names = _declared_functions()
    assert len(names) == 67
    assert not any('select_path' in n for n in names)  # no process-global test hooks in the product ABI: paths are per-call flags
    for must in ('tdk_rcd', 'tdk_ppg', 'tdk_bilinear5x5', 'tdk_wiener', 'tdk_bilateral', 'tdk_laplacian', 'tdk_tonemap',
                 'tdk_color_op', 'tdk_decode12_f32', 'tdk_postprocess', 'tdk_image_metrics_finish'):
        assert must in names


def test_library_exports_every_declared_symbol(td):
    lib = ctypes.CDLL(str(ROOT / 'torch-darktable_amd' / 'torch_darktable' / 'libtdk_hip.so'))
    for name in _declared_functions():
        assert hasattr(lib, name), f'{name} declared in tdk_hip.h but not exported'
    lib.tdk_abi_version.restype = ctypes.c_int
    assert lib.tdk_abi_version() == 4


def test_ctypes_table_matches_header(td):
    from torch_darktable import _native

    assert sorted(_native.SIGNATURES) == _declared_functions()
    # argument counts agree with the header
    text = re.sub(r'/\*.*?\*/', '', HEADER.read_text(), flags=re.S)
    for name, (_, argtypes) in _native.SIGNATURES.items():
        m = re.search(r'\b' + name + r'\s*\(([^;]*?)\)\s*;', text, flags=re.S)
        assert m, name
        args = m.group(1).strip()
        n = 0 if args in ('', 'void') else args.count(',') + 1
        assert n == len(argtypes), f'{name}: header has {n} parameters, ctypes table {len(argtypes)}'


def test_workspace_queries_run_on_the_host(td):
    """Size queries are pure host functions (no GPU needed) and scale as documented."""
    from torch_darktable._native import lib

    assert lib.tdk_rcd_workspace_bytes(4096, 3072) == 0
    assert lib.tdk_ppg_workspace_bytes(4096, 3072, 0.0) == 0
    assert lib.tdk_ppg_workspace_bytes(4096, 3072, 1.0) >= 4096 * 3072 * 4
    sz = (ctypes.c_int * 3)()
    assert lib.tdk_bilateral_grid_size(4096, 3072, 2.0, 0.2, sz) == 0 and tuple(sz) == (2049, 1537, 6)
    assert lib.tdk_bilateral_workspace_bytes(4096, 3072, 2.0, 0.2) >= 2 * 2049 * 1537 * 6 * 4
    assert lib.tdk_wiener_workspace_bytes(4096, 3072, 1, 32, 4) > 4096 * 3072 * 4
    assert lib.tdk_wiener_workspace_bytes(4096, 3072, 1, 24, 4) == 0
    assert lib.tdk_laplacian_workspace_bytes(4096, 3072, 6) > 0 and lib.tdk_laplacian_workspace_bytes(4096, 3072, 4) == 0


def test_invalid_arguments_report_through_last_error(td):
    from torch_darktable._native import lib

    rc = lib.tdk_rcd(None, None, None, 64, 64, 0x94949494, 0, None)
    assert rc == 1 and b'null pointer' in lib.tdk_last_error()
    rc = lib.tdk_wiener(1, 1, 1, 64, 64, 2, 32, 4, 1, 0, None)
    assert rc == 1 and b'channels' in lib.tdk_last_error()
