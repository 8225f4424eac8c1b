"""Seeded synthetic frames with the batch contract of the reference's dataset classes (data/datasets.py:54-57,100-125,250-273).

The annotated readers (PlaneAnnoDataset / ScanNetDataset / NYUDataset) need cv2 and pycocotools and are outside this build;
train.py and eval.py run on these samples instead: `dataset[i]` / `dataset.pull_item(i)` -> (image [3,H,W] float, instances
dict, depth [1,H,W] metres), `detection_collate` -> lists."""
import numpy as np
import torch


class SyntheticPlaneDataset(torch.utils.data.Dataset):
    """Seeded samples with the reference's contract: (image [3,H,W] float, instances dict, depth [1,H,W] metres)."""

    def __init__(self, length, hw=(480, 640)):
        self.length, self.hw = length, hw

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        return self.pull_item(idx)

    def pull_item(self, idx):
        H, W = self.hw
        rng = np.random.RandomState(idx)
        g = torch.Generator().manual_seed(idx)
        n = int(rng.randint(3, 9))
        masks, boxes = np.zeros((n, H, W), np.uint8), np.zeros((n, 4), np.float64)
        for i in range(n):
            bw, bh = int(rng.randint(max(W // 16, 8), W // 2)), int(rng.randint(max(H // 16, 8), H // 2))
            x0, y0 = int(rng.randint(0, W - bw)), int(rng.randint(0, H - bh))
            masks[i, y0:y0 + bh, x0:x0 + bw] = 1
            boxes[i] = (x0, y0, x0 + bw, y0 + bh)
        nrm = rng.randn(n, 3)
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        inst = {"masks": torch.from_numpy(masks), "boxes": torch.from_numpy(boxes), "classes": torch.zeros(n, dtype=torch.int64),
                "plane_paras": torch.from_numpy(np.concatenate([nrm, rng.rand(n, 1) * 3.0, np.zeros((n, 2))], 1)),
                "k_matrix": torch.tensor([[577.0, 0, W / 2], [0, 577.0, H / 2], [0, 0, 1]], dtype=torch.float64)}
        return torch.randn(3, H, W, generator=g), inst, 0.5 + 4.0 * torch.rand(1, H, W, generator=g)


def detection_collate(batch):
    """lists of images / instance dicts / depths (reference data/datasets.py:250-273)"""
    return [s[0] for s in batch], [s[1] for s in batch], [s[2] for s in batch]
