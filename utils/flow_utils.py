"""Optical-flow visualisation (Middlebury colour wheel: Baker et al., "A Database and Evaluation Methodology for
Optical Flow", ICCV 2007) for example_opt_flow.py."""
import numpy as np

_SEGMENTS = ((15, (255, None, 0)), (6, (None, 255, 0)), (4, (0, 255, None)), (11, (0, None, 255)),
             (13, (None, 0, 255)), (6, (255, 0, None)))       # RY YG GC CB BM MR; None = the ramping channel


def make_colorwheel() -> np.ndarray:
    rows = []
    for seg, (n, pattern) in enumerate(_SEGMENTS):
        ramp = np.floor(255 * np.arange(n) / n)
        falling = seg % 2 == 1
        block = np.zeros((n, 3))
        for ch, v in enumerate(pattern):
            block[:, ch] = (255 - ramp if falling else ramp) if v is None else v
        rows.append(block)
    return np.concatenate(rows, axis=0)


def flow_uv_to_colors(u, v, convert_to_bgr=False):
    wheel = make_colorwheel()
    n = wheel.shape[0]
    rad = np.sqrt(u ** 2 + v ** 2)
    pos = (np.arctan2(-v, -u) / np.pi + 1) / 2 * (n - 1)
    k0 = np.floor(pos).astype(np.int32)
    k1 = np.where(k0 + 1 == n, 0, k0 + 1)
    frac = pos - k0
    img = np.zeros(u.shape + (3,), np.uint8)
    for ch in range(3):
        col = (1 - frac) * wheel[k0, ch] / 255.0 + frac * wheel[k1, ch] / 255.0
        inside = rad <= 1
        col = np.where(inside, 1 - rad * (1 - col), col * 0.75)
        img[..., 2 - ch if convert_to_bgr else ch] = np.floor(255 * col)
    return img


def flow_to_image(flow_uv, clip_flow=None, convert_to_bgr=False):
    assert flow_uv.ndim == 3 and flow_uv.shape[2] == 2, "input flow must have shape [H,W,2]"
    if clip_flow is not None:
        flow_uv = np.clip(flow_uv, 0, clip_flow)
    u, v = flow_uv[..., 0], flow_uv[..., 1]
    scale = np.max(np.sqrt(u ** 2 + v ** 2)) + 1e-5
    return flow_uv_to_colors(u / scale, v / scale, convert_to_bgr)
