"""The C-ABI shared library loads (no GPU needed) and exports every symbol include/tcvom_hip.h declares;
the ctypes prototype table covers exactly that set."""
import ctypes
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(REPO, 'include', 'tcvom_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(tcvom_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    import tcvom_amd._lib as L
    lib = ctypes.CDLL(L.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), 'libtcvom_hip.so does not export %s' % n


def test_ctypes_table_matches_header():
    import tcvom_amd._lib as L
    assert sorted(L.EXPORTS) == declared_symbols()


def test_abi_version_and_error_channel():
    import tcvom_amd._lib as L
    assert L.call('tcvom_abi_version') == 1
    # argument validation happens on the host before any launch: a NULL descriptor must fail loudly
    try:
        L.call('tcvom_conv_igemm', None, None, None, None, None, None, None, None, None)
    except L.TcvomError as e:
        assert 'conv_igemm' in str(e)
    else:
        raise AssertionError('expected TcvomError')


def test_struct_layout_matches_header():
    import tcvom_amd._lib as L
    # 13 + 1 + 3*MAX_TAPS + 7 int32 fields, then 5 int64 (8-byte aligned)
    n_i32 = 14 + 3 * L.MAX_TAPS + 7
    assert ctypes.sizeof(L.ConvDesc) == ((n_i32 * 4 + 7) // 8) * 8 + 5 * 8
