"""Synthetic frame pairs (SURVEY.md §8d "F2 smooth synthetic" / "F3 stress"), u8 HWC RGB, tightly packed."""
import numpy as np


def smooth_pair(w, h, seed=1000):
    """Frame 0 = sum of low-frequency sinusoids + filled discs; frame 1 = frame 0's content translated by
    (dx, dy) in [-8, 8] px plus +-2 LSB noise."""
    rng = np.random.default_rng(seed)
    dx, dy = rng.uniform(-8, 8, 2)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)

    def render(ox, oy, r):
        img = np.zeros((h, w, 3), np.float32)
        for c in range(3):
            acc = np.zeros((h, w), np.float32)
            for _ in range(8):
                fx, fy = r.uniform(0.002, 0.03, 2)
                ph = r.uniform(0, 2 * np.pi)
                acc += np.sin((xx - ox) * fx * 2 * np.pi + (yy - oy) * fy * 2 * np.pi + ph)
            img[..., c] = acc / 8.0
        img = 0.5 + 0.35 * img
        for _ in range(16):
            cx, cy = r.uniform(0, w), r.uniform(0, h)
            rad = r.uniform(0.02, 0.08) * min(w, h)
            col = r.uniform(0, 1, 3)
            mask = (xx - ox - cx) ** 2 + (yy - oy - cy) ** 2 < rad * rad
            img[mask] = col
        return img

    state = rng.bit_generator.state
    f0 = render(0.0, 0.0, np.random.default_rng(seed + 1))
    f1 = render(dx, dy, np.random.default_rng(seed + 1))
    rng.bit_generator.state = state
    n0 = rng.integers(-2, 3, f0.shape)
    n1 = rng.integers(-2, 3, f1.shape)
    a = np.clip(np.rint(f0 * 255) + n0, 0, 255).astype(np.uint8)
    b = np.clip(np.rint(f1 * 255) + n1, 0, 255).astype(np.uint8)
    return a, b


def noise_pair(w, h, seed=7):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (h, w, 3), dtype=np.uint8), rng.integers(0, 256, (h, w, 3), dtype=np.uint8)


def smooth_pair_native(w, h, seed=1000):
    """F2 at native resolution for the big frame sizes (3840x2160, 7680x4320): the same kind of content as smooth_pair() - 8 low-frequency
    sinusoids per channel + 16 filled discs, frame 1 = frame 0 translated by (dx, dy) in [-8, 8] px, +-2 LSB noise on both - generated from
    separable factors (sin(ax + by + p) = sin(ax + p) cos(by) + cos(ax + p) sin(by): O(w + h) transcendentals per sinusoid instead of O(w h)),
    a second or two at 4K instead of 17.  Not bit-identical to smooth_pair() (another rounding order): fixtures keep using that one."""
    rng = np.random.default_rng(seed)
    dx, dy = rng.uniform(-8, 8, 2)
    x = np.arange(w, dtype=np.float32); y = np.arange(h, dtype=np.float32)

    def render(ox, oy, r):
        img = np.empty((h, w, 3), np.float32)
        for c in range(3):
            acc = np.zeros((h, w), np.float32)
            for _ in range(8):
                fx, fy = r.uniform(0.002, 0.03, 2)
                ph = r.uniform(0, 2 * np.pi)
                ax = ((x - ox) * (fx * 2 * np.pi) + ph).astype(np.float32); by = ((y - oy) * (fy * 2 * np.pi)).astype(np.float32)
                acc += np.outer(np.cos(by), np.sin(ax)) + np.outer(np.sin(by), np.cos(ax))
            img[..., c] = 0.5 + (0.35 / 8.0) * acc
        for _ in range(16):
            cx, cy = r.uniform(0, w), r.uniform(0, h)
            rad = r.uniform(0.02, 0.08) * min(w, h)
            col = r.uniform(0, 1, 3).astype(np.float32)
            y0, y1 = max(0, int(cy + oy - rad) - 1), min(h, int(cy + oy + rad) + 2)
            x0, x1 = max(0, int(cx + ox - rad) - 1), min(w, int(cx + ox + rad) + 2)
            if y0 >= y1 or x0 >= x1:
                continue
            m = (x[None, x0:x1] - ox - cx) ** 2 + (y[y0:y1, None] - oy - cy) ** 2 < rad * rad
            img[y0:y1, x0:x1][m] = col
        return img

    f0 = render(0.0, 0.0, np.random.default_rng(seed + 1))
    f1 = render(dx, dy, np.random.default_rng(seed + 1))
    n0 = rng.integers(-2, 3, f0.shape, dtype=np.int8)
    n1 = rng.integers(-2, 3, f1.shape, dtype=np.int8)
    a = np.clip(np.rint(f0 * 255) + n0, 0, 255).astype(np.uint8)
    b = np.clip(np.rint(f1 * 255) + n1, 0, 255).astype(np.uint8)
    return a, b


def tiled_real_pair(tiles):
    """F1: the reference's real 640x360 frame pair images/0.png, images/1.png (committed byte for byte under tests/golden/ref/)
    tiled tiles x tiles: 1 -> 640x360, 3 -> 1920x1080, 6 -> 3840x2160, 12 -> 7680x4320."""
    import os
    from PIL import Image
    ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref", "images")
    fr = [np.asarray(Image.open(os.path.join(ref, n)).convert("RGB")) for n in ("0.png", "1.png")]
    return [np.ascontiguousarray(np.tile(f, (tiles, tiles, 1))) for f in fr]
