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
