"""Live per-kernel-family timing for bench.py's roofline object.

When enabled, ops.py brackets every HIP launch of the heavy kernel families with a pair of HIP events recorded on the
stream the kernel is launched on (torch's current stream -- the same handle that is passed to the C ABI), and logs the
launch's work: FLOPs for the MFMA contraction kernels (2*M*K*N of the GEMM the launch evaluates; the zero taps of a
strided dgrad are not credited), bytes for the HBM-bound families (4 B x elements the op must read + write), plus -- where
it differs, i.e. on the Winograd path -- the FLOPs of the reference convolution the launch stands for (`ref`).
Disabled (the default) it costs one attribute test per launch.
"""
import torch

_enabled = False
_records = []          # (family, bound, work, start_event, end_event, reference-operator work)


def enable():
    global _enabled
    _records.clear()
    _enabled = True


def disable():
    global _enabled
    _enabled = False


def active():
    return _enabled


class span:
    """with span(family, bound, work): <launch>
    work: what the launch EXECUTES (FLOPs / bytes) -- the basis of the family's `achieved` rate, i.e. a utilisation.
    ref:  FLOPs of the reference operator the launch stands for, when that differs (a Winograd product executes a quarter of
          the direct convolution's multiply-adds; its transform kernels execute none): summed into `ref_work`."""
    __slots__ = ("family", "bound", "work", "ref", "nbytes", "tag", "s", "e")

    def __init__(self, family, bound, work, ref=None, nbytes=0.0, tag=None):
        """nbytes (MFMA families): algorithmic HBM bytes of the launch, 4 B x (operand + result elements)."""
        self.family, self.bound, self.work, self.ref, self.nbytes, self.tag = family, bound, work, (work if ref is None else ref), nbytes, tag

    def __enter__(self):
        if _enabled:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()

    def __exit__(self, *exc):
        if _enabled:
            self.e.record()
            _records.append((self.family, self.bound, self.work, self.s, self.e, self.ref, self.nbytes, self.tag))


def summary():
    """-> list of {kernel, bound, launches, time_ms, work, achieved, unit} sorted by time (call after a device sync)."""
    fams = {}
    for fam, bound, work, s, e, ref, nbytes, _tag in _records:
        f = fams.setdefault(fam, {"kernel": fam, "bound": bound, "launches": 0, "time_ms": 0.0, "work": 0.0, "ref_work": 0.0, "bytes": 0.0})
        f["bytes"] += float(nbytes)
        f["launches"] += 1
        f["time_ms"] += s.elapsed_time(e)
        f["work"] += float(work)
        f["ref_work"] += float(ref)
    out = []
    for f in fams.values():
        if f["bound"] == "mfma":
            f["achieved"] = f["work"] / (f["time_ms"] * 1e-3) / 1e12 if f["time_ms"] > 0 else 0.0
            f["unit"] = "TFLOP/s"
        else:
            f["achieved"] = f["work"] / (f["time_ms"] * 1e-3) / 1e9 if f["time_ms"] > 0 else 0.0
            f["unit"] = "GB/s"
        out.append(f)
    return sorted(out, key=lambda f: -f["time_ms"])


def by_shape():
    """-> list of (family, tag, launches, time_ms, work) aggregated over the launches that carry a shape tag (call after a sync)."""
    agg = {}
    for fam, bound, work, s, e, ref, nbytes, tag in _records:
        a = agg.setdefault((fam, tag), [0, 0.0, 0.0])
        a[0] += 1
        a[1] += s.elapsed_time(e)
        a[2] += float(work)
    return sorted(((k[0], k[1], v[0], v[1], v[2]) for k, v in agg.items()), key=lambda t: -t[3])
