"""GPU: the two kept entry points run end to end (train.py on synthetic batches incl. checkpoint naming / resume;
simple_inference.py on an image file incl. the input:output syntax)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, cwd):
    r = subprocess.run([sys.executable] + cmd, cwd=cwd, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-3000:]
    return r.stdout


def test_train_synthetic_then_resume(tmp_path):
    common = ["--config", "PlaneRecNet_50_config", "--dataset", "synthetic", "--batch_size", "8", "--save_folder", str(tmp_path) + "/",
              "--num_workers", "0", "--no_tensorboard", "--synthetic_size", "16", "--save_interval", "2", "--reproductablity"]
    out = run([os.path.join(ROOT, "train.py")] + common + ["--max_iter", "2"], str(tmp_path))
    assert "Begin training!" in out and "total:" in out
    ck = sorted(p for p in os.listdir(tmp_path) if p.endswith(".pth"))
    assert "PlaneRecNet_50_0_2.pth" in ck, ck
    out = run([os.path.join(ROOT, "train.py")] + common + ["--max_iter", "3", "--resume", "latest"], str(tmp_path))
    assert "Resuming training" in out
    assert any(p.endswith("_3.pth") for p in os.listdir(tmp_path))


def test_simple_inference_image(tmp_path):
    from PIL import Image
    rng = np.random.RandomState(0)
    img = (rng.rand(240, 320, 3) * 255).astype(np.uint8)
    src = os.path.join(tmp_path, "frame.png")
    Image.fromarray(img).save(src)
    dst = os.path.join(tmp_path, "out.png")
    run([os.path.join(ROOT, "simple_inference.py"), "--config", "PlaneRecNet_50_config", "--image", src + ":" + dst, "--score_threshold", "0.05"],
        str(tmp_path))
    assert os.path.exists(dst) and os.path.exists(os.path.join(tmp_path, "out_dep.png"))
    seg = np.asarray(Image.open(dst))
    assert seg.shape == (480, 640, 3)               # resized to max_size=640 keeping the aspect ratio, padded to /32
