"""CPU: the C-ABI library loads and exports every symbol include/prn.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "prn.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(prn_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_expected_families():
    syms = declared_symbols()
    for fam in ("prn_conv2d_fwd", "prn_conv2d_wgrad", "prn_dcn_sample", "prn_dcn_sample_bwd", "prn_bn_stats", "prn_bn_bwd",
                "prn_gn_relu_fwd", "prn_resize_bilinear_fwd", "prn_maxpool3s2_fwd", "prn_version", "prn_last_error"):
        assert fam in syms


def test_library_loads_and_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    lib = ctypes.CDLL(os.path.join(ROOT, "planerecnet_amd", "libprn_hip.so"))
    for s in declared_symbols():
        assert hasattr(lib, s), s
    lib.prn_version.restype = ctypes.c_int
    assert lib.prn_version() >= 100


def test_python_binding_covers_every_symbol():
    from planerecnet_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_argument_validation_without_gpu():
    """Descriptor validation happens on the host before any launch: must fail cleanly (rc != 0 + message)."""
    from planerecnet_amd import _lib
    d = _lib.ConvDesc(1, 4, 8, 8, 4, 5, 5, 1, 2, 8, 8, 0, 1, 0)       # 5x5 kernels are not part of the path
    rc = _lib.lib.prn_conv2d_fwd(ctypes.byref(d), None, None, None, None, None, None, None)
    assert rc != 0 and b"unsupported" in _lib.lib.prn_last_error()
