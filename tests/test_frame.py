"""Input staging of the inference entry point: the oracle's restatement of cv2.resize(INTER_LINEAR) (CPU properties), and the
fused HIP launch (resize + pad + FastBaseTransform on the uploaded uint8 frame) against it, bit for bit (GPU)."""
import numpy as np
import pytest
import torch


def test_resize_restatement_identity_and_exact_quarter():
    from oracle.frame_ref import resize_linear_u8
    from planerecnet_amd.funcs import quarter_mask_u8
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (48, 64, 3)).astype(np.uint8)
    assert np.array_equal(resize_linear_u8(img, (64, 48)), img)                       # same size: a copy
    q = resize_linear_u8(img, (16, 12))                                                # exact 1/4: rounded mean of the 2x2 centre pixels
    ref = np.stack([quarter_mask_u8(torch.from_numpy(img[:, :, c][None]))[0].numpy() for c in range(3)], -1)
    assert np.array_equal(q, ref)
    ramp = np.tile(np.arange(0, 250, 10, dtype=np.uint8)[None, :, None], (4, 1, 3))   # a linear ramp stays monotone, ends clamp
    up = resize_linear_u8(ramp, (50, 4))[0, :, 0].astype(int)
    assert up[0] == 0 and up[-1] == 240 and (np.diff(up) >= 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("hw,max_size", [((968, 1296), 640), ((480, 640), 640), ((375, 500), 640), ((720, 960), 960), ((333, 517), 640)])
def test_frame_to_input_matches_restatement(hw, max_size):
    from oracle.frame_ref import frame_to_input as ref_fn
    from planerecnet_amd.config import MEANS, STD, set_cfg
    from planerecnet_amd.funcs import FastBaseTransform, calc_size_preserve_ar, frame_to_input, pad_even_divided
    set_cfg("PlaneRecNet_50_config")
    rng = np.random.RandomState(1)
    img = rng.randint(0, 256, hw + (3,)).astype(np.uint8)
    size = calc_size_preserve_ar(hw[1], hw[0], max_size)
    x_ref, frame_ref = ref_fn(img, size, MEANS, STD)
    x, frame = frame_to_input(torch.from_numpy(img).cuda(), size)
    assert tuple(x.shape) == x_ref.shape and x.shape[2] % 32 == 0 and x.shape[3] % 32 == 0
    assert np.array_equal(frame.cpu().numpy(), frame_ref.astype(np.float32))          # the resized uint8 frame: bit exact
    assert np.abs(x.cpu().numpy() - x_ref).max() <= 2e-6                               # (x - mean) / std: one fp32 division
    if size == (hw[1], hw[0]):                                                          # no resize: the reference's own chain on the device
        padded = pad_even_divided(img)
        y = FastBaseTransform().cuda()(torch.from_numpy(padded).cuda().float().unsqueeze(0))
        assert torch.equal(x, y)


@pytest.mark.gpu
def test_frame_to_input_rejects_host_or_float_frames():
    from planerecnet_amd.funcs import frame_to_input
    with pytest.raises(RuntimeError):
        frame_to_input(torch.zeros(8, 8, 3, dtype=torch.uint8), (8, 8))
    with pytest.raises(RuntimeError):
        frame_to_input(torch.zeros(8, 8, 3, device="cuda"), (8, 8))
