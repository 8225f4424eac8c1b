"""Checkpoint naming and running averages (reference: utils/utils.py:12-51,102-165)."""
import math
import os
from collections import deque
from pathlib import Path


class MovingAverage:
    """Windowed mean that drops non-finite samples (reference utils/utils.py:21-23)."""

    def __init__(self, max_window_size=1000):
        self.max_window_size = max_window_size
        self.reset()

    def add(self, elem):
        if not math.isfinite(elem):
            print("Warning: Moving average ignored a value of %f" % elem)
            return
        self.window.append(elem)
        self.sum += elem
        if len(self.window) > self.max_window_size:
            self.sum -= self.window.popleft()

    append = add

    def reset(self):
        self.window, self.sum = deque(), 0

    def get_avg(self):
        return self.sum / max(len(self.window), 1)

    def __len__(self):
        return len(self.window)

    def __str__(self):
        return str(self.get_avg())

    __repr__ = __str__


class SavePath:
    """`<model>_<epoch>_<iteration>[_interrupt].pth` <-> (model, epoch, iteration)."""

    def __init__(self, model_name, epoch, iteration):
        self.model_name, self.epoch, self.iteration = model_name, epoch, iteration

    def get_path(self, root=""):
        return os.path.join(root, "%s_%s_%s.pth" % (self.model_name, self.epoch, self.iteration))

    @staticmethod
    def from_str(path):
        stem = os.path.basename(path)
        stem = stem[:-4] if stem.endswith(".pth") else stem
        parts = stem.split("_")
        if stem.endswith("interrupt"):
            parts = parts[:-1]
        return SavePath("_".join(parts[:-2]), int(parts[-2]), int(parts[-1]))

    @staticmethod
    def remove_interrupt(folder):
        for p in Path(folder).glob("*_interrupt.pth"):
            p.unlink()

    @staticmethod
    def get_interrupt(folder):
        return next((str(p) for p in Path(folder).glob("*_interrupt.pth")), None)

    @staticmethod
    def get_latest(folder, model_name):
        best, best_iter = None, -1
        for p in Path(folder).glob(model_name + "_*"):
            try:
                sp = SavePath.from_str(str(p))
            except Exception:
                continue
            if sp.model_name == model_name and sp.iteration > best_iter:
                best, best_iter = str(p), sp.iteration
        return best
