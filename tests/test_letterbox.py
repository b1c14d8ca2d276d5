"""letterbox + HWC-BGR -> CHW-RGB on the device (mixed_datasets.py:722-752, :576) against the CPU restatement in
oracle/letterbox_ref.py.  Geometry, border and channel order are integer work: bit-exact.  The INTER_LINEAR resize restates
OpenCV's 8-bit fixed-point algorithm on both sides (cv2 is not available: parity unpinned, see the oracle header)."""
import numpy as np
import pytest
import torch


def test_letterbox_geometry_matches_the_reference_arithmetic():
    """The host arithmetic against hand-checked cases of mixed_datasets.py:724-750."""
    from desenet_amd.core.utils.augmentations import letterbox_geometry
    ratio, unpad, (dw, dh), (t, b, l, r) = letterbox_geometry((480, 640), 640, auto=False)
    assert ratio == (1.0, 1.0) and unpad == (640, 480) and (dw, dh) == (0.0, 80.0) and (t, b, l, r) == (80, 80, 0, 0)
    ratio, unpad, (dw, dh), (t, b, l, r) = letterbox_geometry((720, 1280), 640, auto=True, stride=32)
    assert ratio == (0.5, 0.5) and unpad == (640, 360) and (dw, dh) == (0.0, 12.0) and (t, b, l, r) == (12, 12, 0, 0)
    ratio, unpad, (dw, dh), (t, b, l, r) = letterbox_geometry((375, 500), (640, 640), auto=False, scaleup=False)
    assert ratio == (1.0, 1.0) and unpad == (500, 375) and (dw, dh) == (70.0, 132.5) and (t, b, l, r) == (132, 133, 70, 70)
    ratio, unpad, (dw, dh), _ = letterbox_geometry((300, 400), (640, 512), auto=False, scaleFill=True)
    assert unpad == (512, 640) and ratio == (512 / 400, 640 / 300) and (dw, dh) == (0.0, 0.0)


def test_oracle_resize_is_identity_and_exact_on_integer_ratios():
    from oracle.letterbox_ref import letterbox, resize_linear_u8, to_network_input
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (6, 8, 3)).astype(np.uint8)
    assert np.array_equal(resize_linear_u8(img, 8, 6), img)
    flat = np.full((5, 7, 3), 200, np.uint8)
    assert np.array_equal(resize_linear_u8(flat, 13, 9), np.full((9, 13, 3), 200, np.uint8))        # weights sum to 2048
    out, ratio, pad = letterbox(img, (16, 16), auto=False)
    assert out.shape == (16, 16, 3) and ratio == (2.0, 2.0) and pad == (0.0, 2.0)
    assert (out[:2] == 114).all() and (out[-2:] == 114).all()
    chw = to_network_input(out)
    assert chw.shape == (3, 16, 16) and np.array_equal(chw[0], out[:, :, 2])


@pytest.mark.gpu
@pytest.mark.parametrize("shape,new_shape,kw", [((480, 640), 640, dict(auto=False)), ((720, 1280), 640, dict(auto=True)),
                                                ((375, 500), (640, 640), dict(auto=False, scaleup=False)),
                                                ((97, 131), (320, 416), dict(auto=False)),
                                                ((300, 400), (640, 512), dict(auto=False, scaleFill=True)),
                                                ((640, 640), 640, dict(auto=False))])
def test_letterbox_device_vs_oracle(shape, new_shape, kw):
    from desenet_amd.core.utils.augmentations import letterbox
    from oracle import letterbox_ref as R
    rng = np.random.RandomState(shape[0])
    img = rng.randint(0, 256, shape + (3,)).astype(np.uint8)
    want, ratio, pad = R.letterbox(img, new_shape, **kw)
    got, ratio2, pad2 = letterbox(torch.from_numpy(img).cuda(), new_shape, **kw)
    assert ratio2 == ratio and pad2 == pad
    assert np.array_equal(got.cpu().numpy(), want)
    chw, _, _ = letterbox(torch.from_numpy(img).cuda(), new_shape, to_chw_rgb=True, **kw)
    assert np.array_equal(chw.cpu().numpy(), R.to_network_input(want))
