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


def test_split_gemm_plan_is_host_logic():
    """prn_gemm_pipe / the mode, kind and threshold switches of the 16-bit-pipe GEMM (csrc/prn_gemm_split.hip) decide on the host: checked here
    without a GPU.  Mode 0 keeps every launch on the fp32 MFMA kernel; mode 1 takes launches of at least min_tiles 128x128 tiles and 4 GFLOP whose
    last row tile is more than half full, K-splitting short-of-tiles launches; workspace sizes do not depend on the piece format."""
    from planerecnet_amd import _lib
    lib = _lib.lib
    old_mode, old_tiles, old_kind = lib.prn_split_gemm_mode(1), lib.prn_split_gemm_min_tiles(300), lib.prn_split_gemm_kind(16)
    try:
        assert lib.prn_gemm_pipe(1024, 256, 8, 1200, 1) == 1          # stage-3 expand: 640 tiles
        assert lib.prn_gemm_pipe(256, 1024, 8, 1200, 1) == 4          # stage-3 reduce: 160 tiles -> 4 K splits of 8 slices
        assert lib.prn_gemm_pipe(64, 256, 8, 19200, 1) == 0           # half-empty row tile
        assert lib.prn_gemm_pipe(256, 256, 1, 1200, 1) == 0           # batch-1 launch below the FLOP floor
        assert lib.prn_gemm_pipe(256, 256, 1, 9600, 36) == 1          # Winograd products, 36 batched GEMMs
        assert lib.prn_gemm_pipe(0, 256, 8, 1200, 1) == 0
        lib.prn_split_gemm_min_tiles(2500)
        assert lib.prn_gemm_pipe(1024, 256, 8, 1200, 1) == 0 and lib.prn_gemm_pipe(256, 256, 1, 9600, 36) == 1
        lib.prn_split_gemm_mode(0)
        assert lib.prn_gemm_pipe(256, 256, 1, 9600, 36) == 0
        lib.prn_split_gemm_mode(2)
        assert lib.prn_gemm_pipe(64, 256, 8, 19200, 1) >= 1
        nb = lib.prn_split_images_bytes(1000, 70, 3)
        assert nb == 3 * 8 * 3 * 1536 * 16 + 3 * 8 * 128 * 4           # bf16-sized images of 8 row tiles x 3 slices + the fp16 format's row exponents
        lib.prn_split_gemm_kind(0)
        assert lib.prn_split_images_bytes(1000, 70, 3) == nb and lib.prn_split_gemm_kind(-1) == 0
        assert lib.prn_split_images_bytes(0, 70, 3) == -1
    finally:
        lib.prn_split_gemm_mode(old_mode); lib.prn_split_gemm_min_tiles(old_tiles); lib.prn_split_gemm_kind(old_kind)
