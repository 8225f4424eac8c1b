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
    d = _lib.ConvDesc(1, 4, 8, 8, 4, 5, 5, 1, 2, 8, 8, 0, 1, 0)       # 5x5 kernels are not part of the path (options: all zero)
    rc = _lib.lib.prn_conv2d_fwd(ctypes.byref(d), None, None, None, None, None, None, None)
    assert rc != 0 and b"unsupported" in _lib.lib.prn_last_error()


def _opts(mode=1, kind=16, products=3, min_tiles=300, min_gflop=4.0, wgs=0, target=0, wgrad=0):
    from planerecnet_amd import _lib
    o = _lib.GemmOpts(mode, kind, products, min_tiles, min_gflop, wgs, target, wgrad)
    return o, ctypes.byref(o)


def test_split_gemm_plan_is_host_logic():
    """prn_gemm_pipe decides on the host from the CALL's prn_gemm_opts (mode, piece format, thresholds): checked here without a GPU.
    Mode 0 (and a NULL / all-zero opts) keeps every launch on the fp32 MFMA kernel; mode 1 takes launches of at least min_tiles 128x128 tiles
    and min_gflop GFLOP whose last row tile is more than half full, K-splitting short-of-tiles launches; image sizes cover both piece formats."""
    from planerecnet_amd import _lib
    lib = _lib.lib
    _, o = _opts()
    assert lib.prn_gemm_pipe(1024, 256, 8, 1200, 1, o) == 1          # stage-3 expand: 640 tiles
    assert lib.prn_gemm_pipe(256, 1024, 8, 1200, 1, o) == 3          # stage-3 reduce: 160 tiles -> 3 K splits (the cap; >= 8 slices each)
    assert lib.prn_gemm_pipe(512, 2048, 8, 300, 1, o) == 4           # stage-4 reduce: 96 tiles reach the 300-tile floor only with 4 splits: the cap yields
    assert lib.prn_gemm_pipe(64, 256, 8, 19200, 1, o) == 0           # half-empty row tile
    assert lib.prn_gemm_pipe(256, 256, 1, 1200, 1, o) == 0           # batch-1 launch below the FLOP floor
    assert lib.prn_gemm_pipe(256, 256, 1, 9600, 36, o) == 1          # Winograd products, 36 batched GEMMs
    assert lib.prn_gemm_pipe(0, 256, 8, 1200, 1, o) == 0
    _, o2500 = _opts(min_tiles=2500)
    assert lib.prn_gemm_pipe(1024, 256, 8, 1200, 1, o2500) == 0 and lib.prn_gemm_pipe(256, 256, 1, 9600, 36, o2500) == 1
    _, off = _opts(mode=0)
    assert lib.prn_gemm_pipe(256, 256, 1, 9600, 36, off) == 0
    assert lib.prn_gemm_pipe(256, 256, 1, 9600, 36, None) == 0       # no options: fp32 MFMA
    _, always = _opts(mode=2)
    assert lib.prn_gemm_pipe(64, 256, 8, 19200, 1, always) >= 1
    nb = lib.prn_split_images_bytes(1000, 70, 3)
    assert nb == 3 * 8 * 3 * 1536 * 16 + 3 * 8 * 128 * 4           # bf16-sized images of 8 row tiles x 3 slices + the fp16 format's row exponents
    assert lib.prn_split_images_bytes(0, 70, 3) == -1
    # the one call that made an answer depend on a process-wide switch is gone: the same query under two option values, interleaved
    assert [lib.prn_gemm_pipe(1024, 256, 8, 1200, 1, q) for q in (o, off, o2500, o)] == [1, 0, 0, 1]
    d = _lib.GemmOpts()
    lib.prn_gemm_opts_default(ctypes.byref(d))
    assert d.key() == (1, 16, 3, 300, 4.0, 0, 0, 1)


def test_workspace_sizes_follow_the_descriptors_options():
    """prn_conv2d_fwd_ws_bytes / _kernel_kind / the weight-gradient plan read the options INSIDE the descriptor; two threads asking with
    different options at the same time get their own answers (no process-wide mode)."""
    import threading
    from planerecnet_amd import _lib
    lib = _lib.lib

    def desc(o):
        return _lib.ConvDesc(8, 256, 30, 40, 1024, 1, 1, 1, 0, 30, 40, 0, 1, 0, 0, 0, 0, 0, o)
    on, _ = _opts()
    off, _ = _opts(mode=0)
    wgs, _ = _opts(mode=0, wgs=128)
    d_on, d_off, d_wgs = desc(on), desc(off), desc(wgs)
    assert lib.prn_conv2d_kernel_kind(ctypes.byref(d_on)) == 2 and lib.prn_conv2d_kernel_kind(ctypes.byref(d_off)) == 0
    assert lib.prn_conv2d_fwd_ws_bytes(ctypes.byref(d_on)) >= lib.prn_split_images_bytes(1024, 256, 1)
    assert lib.prn_conv2d_fwd_ws_bytes(ctypes.byref(d_off)) == 0
    full, part = lib.prn_conv2d_wgrad_ws_bytes(ctypes.byref(d_off)), lib.prn_conv2d_wgrad_ws_bytes(ctypes.byref(d_wgs))
    assert full > part > 0                                          # fewer workgroups per launch -> fewer pixel splits -> smaller workspace
    out = {}

    def worker(name, d, want):
        ok = True
        for _ in range(2000):
            ok = ok and lib.prn_conv2d_kernel_kind(ctypes.byref(d)) == want
        out[name] = ok
    ts = [threading.Thread(target=worker, args=("on", d_on, 2)), threading.Thread(target=worker, args=("off", d_off, 0))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert out == {"on": True, "off": True}


def test_library_allocates_nothing_and_keeps_no_registry():
    """caller-owned buffers, no global mutable state.  Source-level check of csrc/: no device allocation / free / blocking
    synchronisation, no process-wide containers or locks, no writes to the environment."""
    csrc = os.path.join(ROOT, "planerecnet_amd", "csrc")
    banned = ("hipMalloc", "hipFree", "hipHostMalloc", "hipStreamSynchronize", "hipDeviceSynchronize", "hipEventSynchronize", "std::map", "std::unordered_map",
              "std::mutex", "setenv", "putenv")
    hits = []
    for f in sorted(os.listdir(csrc)):
        txt = open(os.path.join(csrc, f)).read()
        txt = re.sub(r"//[^\n]*", "", txt)
        for b in banned:
            if b in txt:
                hits.append((f, b))
    assert hits == [], hits
    syms = declared_symbols()
    for gone in ("prn_split_gemm_mode", "prn_split_gemm_kind", "prn_split_gemm_min_tiles", "prn_split_images_register"):
        assert gone not in syms


def test_wgrad16_plan_is_host_logic():
    """Which weight gradients take the fp16-piece kernel (csrc/prn_wgrad16.hip) is decided on the host from the descriptor's options: dense
    stride-1 1x1 layers whose dW tiles are mostly full and that carry >= 1 GFLOP; the workspace then holds the pixel-split partial sums."""
    from planerecnet_amd import _lib
    lib = _lib.lib

    def desc(o, M=1024, C=256, k=1, stride=1, H=30, W=40):
        Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
        return _lib.ConvDesc(8, C, H, W, M, k, k, stride, k // 2, Ho, Wo, 0, 1, 0, 0, 0, 0, 0, o)
    on, _ = _opts(mode=0, wgrad=1)
    off, _ = _opts(mode=0, wgrad=0)
    d_on, d_off = desc(on), desc(off)
    b_on, b_off = lib.prn_conv2d_wgrad_ws_bytes(ctypes.byref(d_on)), lib.prn_conv2d_wgrad_ws_bytes(ctypes.byref(d_off))
    assert b_on > 0 and b_on % (1024 * 256 * 4) == 0 and b_off > 0
    d3 = desc(on, k=3)                                              # 3x3: the fp32 implicit GEMM whatever the option says
    assert lib.prn_conv2d_wgrad_ws_bytes(ctypes.byref(d3)) == lib.prn_conv2d_wgrad_ws_bytes(ctypes.byref(desc(off, k=3)))
    small = desc(on, M=64, C=64)                                    # half-empty tiles: fp32
    assert lib.prn_conv2d_wgrad_ws_bytes(ctypes.byref(small)) == lib.prn_conv2d_wgrad_ws_bytes(ctypes.byref(desc(off, M=64, C=64)))
    _, o_on = _opts(mode=0, wgrad=1)
    _, o_off = _opts(mode=0, wgrad=0)
    assert lib.prn_gemm_batched_nt_splits(256, 256, 9600, 36, o_on) >= 1 and lib.prn_gemm_batched_nt_splits(256, 256, 9600, 36, o_off) >= 1


def test_wgrad16_plan_does_not_depend_on_pointer_alignment():
    """The workspace of a weight gradient is sized from the descriptor (prn_conv2d_wgrad_ws_bytes, prn_gemm_batched_nt_splits).  An operand
    the planned 16-bit-pipe kernel cannot take (not 16-byte aligned: a C caller's offset sub-tensor) must be REFUSED on the host -- it used to
    fall back to the fp32 kernel, whose own split count then wrote past a workspace sized for the other plan (advisor, round 4)."""
    from planerecnet_amd import _lib
    lib = _lib.lib
    on, o_on = _opts(mode=0, wgrad=1)
    d = _lib.ConvDesc(8, 256, 30, 40, 1024, 1, 1, 1, 0, 30, 40, 0, 1, 0, 0, 0, 0, 0, on)
    assert lib.prn_conv2d_wgrad_kernel_kind(ctypes.byref(d), 1) == 2
    fake, odd = ctypes.c_void_p(0x10000), ctypes.c_void_p(0x10004)
    for x, dy in ((odd, fake), (fake, odd)):
        rc = lib.prn_conv2d_wgrad(ctypes.byref(d), x, dy, fake, fake, None)
        assert rc != 0 and b"16-byte aligned" in lib.prn_last_error()
    arr = (ctypes.c_void_p * 2)(0x10000, 0x10004)
    ok = (ctypes.c_void_p * 2)(0x10000, 0x20000)
    rc = lib.prn_conv2d_wgrad_grouped(ctypes.byref(d), 2, arr, ok, fake, fake, None)
    assert rc != 0 and b"16-byte aligned" in lib.prn_last_error()
    assert lib.prn_gemm_batched_nt_kind(256, 256, 9600, 36, o_on) == 2
    rc = lib.prn_gemm_batched_nt(256, 256, 9600, 36, odd, fake, fake, o_on, None)
    assert rc != 0 and b"16-byte aligned" in lib.prn_last_error()


def test_one_default_plan_in_one_place():
    """The shipping defaults of the per-call options exist once: prn_gemm_opts_default() in the library; the Python layer's policy (what
    ops.py puts into every descriptor when no environment variable overrides it) must be that value, field by field."""
    import subprocess
    import sys
    code = ("import ctypes, os\n"
            "for k in list(os.environ):\n"
            "    if k.startswith('PRN_SPLIT') or k.startswith('PRN_WGRAD'): del os.environ[k]\n"
            "from planerecnet_amd import _lib, ops\n"
            "d = _lib.GemmOpts(); _lib.lib.prn_gemm_opts_default(ctypes.byref(d))\n"
            "ops.split_gemm_policy('train'); a = ops.opts_key(); ops.split_gemm_policy('eval'); b = ops.opts_key()\n"
            "assert a == d.key() and b == d.key(), (a, b, d.key())\n"
            "assert _lib.lib.prn_conv2d_wgrad_kernel_kind(ctypes.byref(_lib.ConvDesc(8, 256, 30, 40, 1024, 1, 1, 1, 0, 30, 40, 0, 1, 0, 0, 0, 0, 0, d)), 1) == 2\n"
            "assert _lib.lib.prn_gemm_batched_nt_kind(256, 256, 9600, 36, ctypes.byref(d)) == 2 and _lib.lib.prn_gemm_batched_nt_kind(256, 256, 9600, 36, None) == 0\n"
            "print('ok')\n")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-800:]


def test_producer_to_batchnorm_hand_overs_are_host_logic_until_the_launch():
    """prn_conv2d_fwd_partials (where phase 1 of a K-split forward leaves its partial sums) is a function of the descriptor alone; the consumers
    (prn_bn_*_partials / _winograd, prn_winograd_output_bn_*) validate shapes, counts and the one-launch map size on the host before any launch."""
    from planerecnet_amd import _lib
    lib = _lib.lib
    o, _ = _opts()
    off = ctypes.c_int64(-1)

    def desc(C, M, H, W, opts, B=8):
        return _lib.ConvDesc(B, C, H, W, M, 1, 1, 1, 0, H, W, 0, 1, 0, 0, 0, 0, 0, opts)

    d = desc(1024, 256, 30, 40, o)                                 # stage-3 reducing layer: three K splits on the split kernel, behind its weight images
    assert lib.prn_conv2d_fwd_partials(ctypes.byref(d), ctypes.byref(off)) == 3
    assert off.value == (lib.prn_split_images_bytes(256, 1024, 1) + 255) // 256 * 256 and off.value % 16 == 0
    d = desc(256, 1024, 30, 40, o)                                 # the expanding layer fills the GPU without a split: nothing to hand over
    assert lib.prn_conv2d_fwd_partials(ctypes.byref(d), ctypes.byref(off)) == 0
    zero = _lib.GemmOpts(0, 0, 0, 0, 0.0, 0, 0, 0)
    d = desc(1024, 256, 30, 40, zero)                              # fp32 kernels only: their own K split, partial sums at the head of the workspace
    n = lib.prn_conv2d_fwd_partials(ctypes.byref(d), ctypes.byref(off))
    assert n > 1 and off.value == 0 and lib.prn_conv2d_fwd_ws_bytes(ctypes.byref(d)) == n * 8 * 256 * 1200 * 4
    assert lib.prn_conv2d_fwd_partials(ctypes.byref(d), None) != 0 and b"null offset" in lib.prn_last_error()
    # one-launch map sizes: 8 x 30x40 and 8 x 15x20 yes, 8 x 60x80 no; W % 4 != 0 no
    assert lib.prn_bn_kernel_kind(8, 1200) == 1 and lib.prn_bn_kernel_kind(8, 300) == 1 and lib.prn_bn_kernel_kind(8, 4800) == 0
    p = ctypes.c_void_p(4096)                                      # (never dereferenced: validation precedes the launch)
    assert lib.prn_bn_train_fwd_partials(p, 3, 8 * 256 * 4800, p, p, p, p, None, p, None, None, 8, 256, 4800, 1e-5, 0.1, 1, None) != 0
    assert b"one-pass" in lib.prn_last_error()
    assert lib.prn_bn_train_fwd_partials(p, 0, 0, p, p, p, p, None, p, None, None, 8, 256, 1200, 1e-5, 0.1, 1, None) != 0      # no partial sums at all
    assert lib.prn_bn_bwd_partials(p, 3, 100, p, None, p, p, p, p, None, None, None, 8, 256, 1200, 1, 0, None) != 0           # partial sums overlap
    assert lib.prn_bn_train_fwd_winograd(p, 1, 0, None, p, p, p, None, p, None, None, p, 8, 256, 30, 42, 1e-5, 0.1, 1, None) != 0   # W % 4 != 0
    assert lib.prn_winograd_output_bn_fwd(p, p, p, p, p, p, None, None, 8, 256, 60, 80, 1e-5, 0.1, 1, None) != 0                    # 2400 tiles: not a one-launch map
    assert lib.prn_vnl_trim_ws_bytes(600) == 600 * 16 * 12 and lib.prn_vnl_trim_ws_bytes(0) == -1


def _block_desc(B=8, C=1024, H=30, W=40, P=256, stride=1, dcn=0, ds=0, flags=15, max_offset=0.0):
    from planerecnet_amd import _lib
    o, _ = _opts(wgrad=1)
    f4 = ctypes.c_float * 4
    return _lib.BottleneckDesc(B, C, H, W, P, stride, dcn, ds, flags, f4(1e-5, 1e-5, 1e-5, 1e-5), f4(0.1, 0.1, 0.1, 0.1), max_offset, 0, o)


def _block_plan(d):
    from planerecnet_amd import _lib, blocks
    lib = _lib.lib
    buf = ctypes.create_string_buffer(lib.prn_bottleneck_plan_bytes())
    rc = lib.prn_bottleneck_plan(ctypes.byref(d), buf)
    info = (ctypes.c_int64 * blocks.INFO_COUNT)()
    if rc == 0:
        assert lib.prn_bottleneck_plan_info(buf, info, blocks.INFO_COUNT) == 0
    return rc, buf, list(info)


def test_bottleneck_plan_is_host_logic():
    """prn_bottleneck_plan (include/prn.h) is a function of the descriptor alone: which path conv2 takes, which producer -> BatchNorm hand-overs apply,
    buffer sizes and operand offsets -- checked here without a GPU for the block shapes of the benchmark (PlaneRecNet_101, B = 8, 480x640)."""
    from planerecnet_amd import blocks as bk
    # stage 3, plain block: Winograd conv2, V kept, every hand-over (30x40 maps: one-launch BatchNorm kernels, conv1 / conv3-dgrad K-split)
    rc, _, i = _block_plan(_block_desc())
    assert rc == 0 and i[bk.I_CONV2] == bk.CONV2_WINOGRAD and i[bk.I_KEEPS_V] == 1 and i[bk.I_HANDOVERS] == 0b111111 and (i[bk.I_HO], i[bk.I_WO]) == (30, 40)
    assert i[bk.I_BN_FLOATS] == 2 * 256 + 2 * 256 + 2 * 1024
    # ... the same without the hand-over flag: nothing handed over, same sizes of what the backward pass keeps
    rc, _, j = _block_plan(_block_desc(flags=15 & ~2))
    assert rc == 0 and j[bk.I_HANDOVERS] == 0 and j[bk.I_SAVE] == i[bk.I_SAVE]
    # stage 1 (120x160): two-launch BatchNorm kernels -> no hand-overs; Winograd with V kept (88 MB <= 128 MB)
    rc, _, i = _block_plan(_block_desc(C=256, H=120, W=160, P=64))
    assert rc == 0 and i[bk.I_CONV2] == bk.CONV2_WINOGRAD and i[bk.I_HANDOVERS] == 0 and i[bk.I_KEEPS_V] == 1
    # first block of stage 3: deformable conv2 at stride 2 with a downsample branch (60x80 -> 30x40): only conv3's input gradient hands its sums over
    rc, _, i = _block_plan(_block_desc(C=512, H=60, W=80, P=256, stride=2, dcn=1, ds=1, max_offset=20.0))
    assert rc == 0 and i[bk.I_CONV2] == bk.CONV2_DCN and (i[bk.I_HO], i[bk.I_WO]) == (30, 40) and i[bk.I_HANDOVERS] == 0b001000
    assert i[bk.I_BN_FLOATS] == 4 * 256 + 4 * 1024 and i[bk.I_OM] > 0 and i[bk.I_TABLE] > i[bk.I_OM] and i[bk.I_DD] > 0
    # every offset lies inside its buffer and is 256-byte aligned
    for k in (bk.I_A1, bk.I_V, bk.I_A2, bk.I_OM, bk.I_TABLE):
        assert 0 <= i[k] < i[bk.I_SAVE] and i[k] % 256 == 0
    for k in (bk.I_D1, bk.I_D2, bk.I_D3, bk.I_DD, bk.I_DOM):
        assert 0 <= i[k] < i[bk.I_GSAVE] and i[k] % 256 == 0
    # too few channels for F(4x4,3x3): the direct kernel
    rc, _, i = _block_plan(_block_desc(B=2, C=64, H=12, W=16, P=16))
    assert rc == 0 and i[bk.I_CONV2] == bk.CONV2_DIRECT


def test_bottleneck_entry_points_validate_before_any_launch():
    """Bad descriptors / plans / buffers are refused on the host (rc != 0 + message), nothing is launched."""
    from planerecnet_amd import _lib
    lib = _lib.lib
    err = lambda: lib.prn_last_error().decode()      # noqa: E731
    rc, _, _ = _block_plan(_block_desc(C=100))                         # no downsample branch, but the block would change the channel count
    assert rc != 0 and "downsample" in err()
    rc, _, _ = _block_plan(_block_desc(stride=3, ds=1))
    assert rc != 0 and "stride" in err()
    rc, _, _ = _block_plan(_block_desc(dcn=1))                         # deformable variant without its offset clamp
    assert rc != 0 and "max_offset" in err()
    assert lib.prn_bottleneck_plan(None, ctypes.create_string_buffer(64)) != 0
    assert lib.prn_bottleneck_plan(ctypes.byref(_block_desc()), None) != 0
    p = _lib.BottleneckParams()
    junk = ctypes.create_string_buffer(lib.prn_bottleneck_plan_bytes())      # not a plan prn_bottleneck_plan filled
    assert lib.prn_bottleneck_train_fwd(junk, ctypes.byref(p), 256, 256, 256, 256, None) != 0 and "plan" in err()
    assert lib.prn_bottleneck_train_bwd(junk, ctypes.byref(p), 256, 256, 256, 256, 0, 256, 256, 256, 256, None) != 0 and "plan" in err()
    rc, plan, _ = _block_plan(_block_desc())
    assert rc == 0
    assert lib.prn_bottleneck_train_fwd(plan, ctypes.byref(p), None, 256, 256, 256, None) != 0 and "null" in err()
    assert lib.prn_bottleneck_train_fwd(plan, ctypes.byref(p), 256, 256, 256, 256, None) != 0 and "null" in err()      # the parameter table is empty
    p.w1 = p.w3 = p.u2 = 4096
    for k in range(3):
        p.gamma[k] = p.beta[k] = p.running_mean[k] = p.running_var[k] = 4096
    assert lib.prn_bottleneck_train_fwd(plan, ctypes.byref(p), 4096, 4096, 4096 + 16, 4096, None) != 0 and "aligned" in err()
    assert lib.prn_bottleneck_train_bwd(plan, ctypes.byref(p), 4096, 4096, 4096, 4096, 0, 4096, 4096, 4096, 4096, None) != 0 and "layout" in err()
    p.w1_t = p.w3_t = p.ut2 = 4096
    assert lib.prn_bottleneck_train_bwd(plan, ctypes.byref(p), 4096, 4096, 4096, 4096, 1, 4096, 4096, 4096, 4096, None) != 0 and "dx_accumulate" in err()
    info = (ctypes.c_int64 * 4)()
    assert lib.prn_bottleneck_plan_info(plan, info, 4) != 0                  # too short an output array
