"""Named stack timers over HIP events (reference API: utils/timer.py -- env/start/stop/reset/print_stats/
total_time/disable_all/enable_all/disable/enable).  `torch.cuda.Event` IS a hipEvent on ROCm; events are
created lazily so importing this module never touches the device (the reference creates them at import)."""
from collections import defaultdict

import torch

_totals = defaultdict(float)
_open = set()
_hidden = set()
_stack = []
_current = None
_off = False
_events = None


def _ev():
    global _events
    if _events is None:
        _events = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    return _events


def disable_all():
    global _off
    _off = True


def enable_all():
    global _off
    _off = False


def disable(name):
    _hidden.add(name)


def enable(name):
    _hidden.remove(name)


def reset():
    global _current
    _totals.clear()
    _open.clear()
    _stack.clear()
    _current = None


def _begin(name):
    _open.add(name)
    _ev()[0].record()


def _end(name):
    if name not in _open:
        print("Warning: timer for %s stopped before starting!" % name)
        return
    s, e = _ev()
    e.record()
    e.synchronize()
    _totals[name] += s.elapsed_time(e)


def start(name, use_stack=True):
    """With use_stack, starting a timer pauses the running one (exclusive times), as in the reference."""
    global _current
    if _off:
        return
    if not use_stack:
        return _begin(name)
    if _current is not None:
        _end(_current)
        _stack.append(_current)
    _begin(name)
    _current = name


def stop(name=None, use_stack=True):
    global _current
    if _off:
        return
    if not use_stack:
        return _end(name)
    if _current is None:
        print("Warning: timer stopped with no timer running!")
        return
    _end(_current)
    _current = _stack.pop() if _stack else None
    if _current is not None:
        _begin(_current)


def total_time():
    return sum(t for n, t in _totals.items() if n not in _hidden)


def print_stats():
    names = [n for n in _totals if n not in _hidden]
    width = max([len(n) for n in names] + [4])
    width += width % 2
    row = " {:>%d} | {:>10.4f} " % width
    head = (" {:^%d} | {:^10} " % width).format("Name", "Time (ms)")
    bar = "-" * head.find("|") + "+" + "-" * (len(head) - head.find("|") - 1)
    print("\n" + head + "\n" + bar)
    for n in names:
        print(row.format(n, _totals[n]))
    print(bar + "\n" + row.format("Total", total_time()) + "\n")


class env:
    """`with timer.env("backbone"): ...`"""

    def __init__(self, name, use_stack=True):
        self.name, self.use_stack = name, use_stack

    def __enter__(self):
        start(self.name, use_stack=self.use_stack)

    def __exit__(self, *exc):
        stop(self.name, use_stack=self.use_stack)
