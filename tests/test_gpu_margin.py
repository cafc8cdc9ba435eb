"""How much room is there under the 1-LSB bar?  flownet.bin is seeded synthetic everywhere (the trained blobs are missing upstream), and the
shipped split-f16 arithmetic already leaves 0.13 % of the channels of a 4K noise frame off by one (profiles/r4/parity_report.txt, C4-F3).  Trained weights with
a larger flow gain would move sampling coordinates further for the same relative error.  These tests turn that knob: rife-v4.6 models whose flow heads are
2 x and 4 x as strong (flows and mask logits of tens of pixels on noise), other seeds, the worst frame class (F3: per-pixel noise), up to 1920 x 1080 - against
the CPU oracle, same bar: max 1 LSB.  The share of channels that differ is printed, not asserted: it is the margin indicator to watch when the
arithmetic changes (fp8 `lo`, Winograd: DESIGN.md (f)-0)."""
import importlib

import numpy as np
import pytest

from oracle import pyoracle
from tools import gen_frames, gen_models

pytestmark = pytest.mark.gpu
amd = importlib.import_module("rife-ncnn-vulkan_amd")


@pytest.mark.parametrize("gain,seed,w,h", [(2.0, 0x51FE, 640, 360), (4.0, 0x51FE, 640, 360), (4.0, 0xBEEF, 1000, 520), (2.0, 0xA11CE, 1920, 1080), (4.0, 0xA11CE, 1920, 1080)])
def test_stronger_flow_heads_stay_within_1_lsb_on_noise(gain, seed, w, h):
    d = gen_models.ensure(None, "rife-v4.6", seed=seed, flow_gain=gain)
    g = amd.RIFE(0, rife_v4=True); g.load(d)
    o = pyoracle.OracleRIFE(rife_v4=True); o.set_gpu_crop(1); o.load(d)
    a, b = gen_frames.noise_pair(w, h, 7 + int(gain))
    for t in (0.5, 0.3):
        got, want = g.process(a, b, t), o.process(a, b, t)
        diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
        flow = np.abs(o.v4_extract(a, b, t, "flow3")[:4]).max() if t == 0.5 and w <= 1000 else float("nan")
        print("margin: gain %g seed %x %dx%d t %.1f: max %d LSB, %.4f %% of the channels differ, |flow3 delta| up to %.1f px" % (
            gain, seed, w, h, t, int(diff.max()), 100.0 * float((diff != 0).mean()), flow))
        assert diff.max() <= 1, (gain, seed, w, h, t, int(diff.max()), int((diff >= 2).sum()))
