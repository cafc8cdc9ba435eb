#!/usr/bin/env python3
"""Copy the reference's own data files that pin this repo's oracle and HIP path into tests/golden/ref/ (build container only).

The reference ships no tests or golden vectors, and its trained flownet / fusionnet weights are absent from the
snapshot.  What it does hold are (a) the nine REAL `models/*/contextnet.bin` weight files and (b) the real 640x360 frame pair
`images/0.png`, `images/1.png` (SURVEY §8d "F1").  /root/reference does not exist on the GPU box, so they are committed here
as fixtures, byte for byte, with a manifest of their md5 sums and origins.  These are data files, not sources.

    python tools/make_ref_fixtures.py [/root/reference]
"""
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DST = os.path.join(ROOT, "tests", "golden", "ref")
FAMILIES = ["rife", "rife-HD", "rife-UHD", "rife-anime", "rife-v2", "rife-v2.3", "rife-v2.4", "rife-v3.0", "rife-v3.1"]


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    files = [("models/%s/contextnet.bin" % f, "models/%s/contextnet.bin" % f) for f in FAMILIES]
    files += [("images/0.png", "images/0.png"), ("images/1.png", "images/1.png")]
    manifest = {}
    for src, dst in files:
        s, d = os.path.join(ref, src), os.path.join(DST, dst)
        os.makedirs(os.path.dirname(d), exist_ok=True)
        shutil.copyfile(s, d)
        manifest[dst] = {"origin": "nihui/rife-ncnn-vulkan " + src, "bytes": os.path.getsize(d), "md5": md5(d)}
    with open(os.path.join(DST, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote %d files, %.1f MB" % (len(files), sum(v["bytes"] for v in manifest.values()) / 1e6))


if __name__ == "__main__":
    main()
