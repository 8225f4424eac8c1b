"""Host -> device staging of training batches (replaces the reference's per-sample `.to(device)` in prepare_data,
train.py:190-197): frames and GT depth maps of the NEXT batch are copied into one of a few rotating page-locked buffers and
uploaded on a side HIP stream while the current step computes; the consumer waits on the returned event.

A pageable upload would block the host until the compute stream has drained (78 vs 68 ms per iteration in round 1); a
`pin_memory=True` DataLoader re-allocates every batch page-locked and its pinning thread competes for the GIL.
"""
import torch


class FrameStager:
    def __init__(self, device, slots=3):
        self.device = torch.device(device)
        self.slots = [{} for _ in range(slots)]
        self.count = 0

    def upload(self, images, depths, main_stream, side_stream):
        """images / depths: lists of per-sample CPU tensors ([3,H,W], [1,H,W]).  -> (x [B,3,H,W], d [B,1,H,W] on the device,
        event recorded on `side_stream` after both copies).  The returned tensors may be used on `main_stream` after
        `main_stream.wait_event(event)`."""
        shape_x = (len(images),) + tuple(images[0].shape)
        shape_d = (len(depths),) + tuple(depths[0].shape)
        slot = self.slots[self.count % len(self.slots)]
        self.count += 1
        if slot.get("x") is None or slot["x"].shape != shape_x or slot["d"].shape != shape_d or slot["x"].dtype != images[0].dtype:
            slot["x"] = torch.empty(shape_x, dtype=images[0].dtype).pin_memory()
            slot["d"] = torch.empty(shape_d, dtype=depths[0].dtype).pin_memory()
        elif slot.get("ev") is not None:
            slot["ev"].synchronize()                       # the buffer's previous upload (len(slots) batches ago) must have left it
        for i, (im, dp) in enumerate(zip(images, depths)):
            slot["x"][i].copy_(im)
            slot["d"][i].copy_(dp)
        side_stream.wait_stream(main_stream)
        with torch.cuda.stream(side_stream):
            x = slot["x"].to(self.device, non_blocking=True)
            d = slot["d"].to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        slot["ev"] = ev
        x.record_stream(main_stream)
        d.record_stream(main_stream)
        return x, d, ev
