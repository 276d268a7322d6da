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


def test_host_side_under_address_sanitizer():
    """SURVEY.md section 5 (sanitizers): the host half of the C ABI -- descriptor planning, tile selection, argument validation, the
    error channel -- built with -fsanitize=address (`make -C tcvom_amd/csrc asan`) and driven by the planning / ABI tests in a
    subprocess under the clang ASAN runtime; a heap overflow or use-after-free in that code aborts the run.  (Device code is
    compiled as usual: device-side ASAN needs xnack+ modes.)"""
    import glob
    import subprocess
    import sys
    rts = glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so')
    if not rts:
        import pytest
        pytest.skip('clang ASAN runtime not found')
    lib = os.path.join(REPO, 'tcvom_amd', 'lib', 'libtcvom_hip_asan.so')
    srcs = glob.glob(os.path.join(REPO, 'tcvom_amd', 'csrc', '*.hip')) + [os.path.join(REPO, 'tcvom_amd', 'csrc', 'common.h'),
                                                                         os.path.join(REPO, 'include', 'tcvom_hip.h')]
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(['make', '-C', os.path.join(REPO, 'tcvom_amd', 'csrc'), 'asan', '-j8'], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=rts[0], ASAN_OPTIONS='detect_leaks=0:abort_on_error=1', TCVOM_LIB=lib, TCVOM_DTYPE='bf16')
    cmd = [sys.executable, '-m', 'pytest', '-q', '-x', os.path.join(REPO, 'tests', 'test_conv_plan.py'),
           os.path.join(REPO, 'tests', 'test_abi.py'), '-k', 'not address_sanitizer']
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and ' passed' in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]
    assert 'AddressSanitizer' not in out.stderr
