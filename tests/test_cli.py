"""The reference's command line (src/main.cpp) re-hosted on the HIP engine: schedule + validation on CPU, end to end on GPU."""
import importlib
import os

import numpy as np
import pytest

from conftest import ROOT

import cli_harness as cli      # tests/cli_harness.py: the Python restatement of the command-line contract (the product CLI is rife-hip, csrc/main.cpp)


def test_schedule_default_doubles_frame_count():
    """numframe = 2*count: outputs alternate t=0 (a copy of frame k) and t=0.5; the tail clamps to (count-2, 1.0)
    (src/main.cpp:705-731)."""
    s = cli.build_schedule(4, 0)
    assert s == [(0, 0.0), (0, 0.5), (1, 0.0), (1, 0.5), (2, 0.0), (2, 0.5), (2, 1.0), (2, 1.0)]


def test_schedule_arbitrary_count():
    s = cli.build_schedule(3, 7)          # scale = 3/7
    assert len(s) == 7 and s[0] == (0, 0.0)
    for i, (sx, fx) in enumerate(s):
        assert 0 <= sx <= 1 and 0.0 <= fx <= 1.0
        if i * 3 / 7 < 2:
            assert abs(sx + fx - i * 3 / 7) < 1e-6
    assert s[-1] == (1, 1.0) or abs(s[-1][0] + s[-1][1] - 6 * 3 / 7) < 1e-6


def test_family_from_dir_name():
    assert cli.model_family("rife-v2.3") == (True, False)
    assert cli.model_family("models/rife-v3.1") == (True, False)
    assert cli.model_family("/x/rife-v4.6") == (False, True)
    assert cli.model_family("rife-anime") == (False, False)
    assert cli.model_family("other") is None


def test_validation_errors(tmp_path, capsys):
    assert cli.main([]) == -1
    assert cli.main(["-0", "a.png", "-1", "b.png", "-o", "o.png", "-s", "1.5", "-m", "rife-v4.6"]) == -1      # timestep range
    assert cli.main(["-0", "a.png", "-1", "b.png", "-o", "o.bmp", "-m", "rife-v4.6"]) == -1                   # extension
    assert cli.main(["-0", "a.png", "-1", "b.png", "-o", "o.png", "-s", "0.3", "-m", "rife-v2.3"]) == -1      # only v4 takes -s
    assert cli.main(["-0", "a.png", "-1", "b.png", "-o", "o.png", "-m", "nonsense"]) == -1                    # unknown family
    assert cli.main(["-0", "a.png", "-1", "b.png", "-o", "o.png", "-j", "1:2,2:2", "-m", "rife-v4.6"]) == -1  # -j / -g mismatch


@pytest.mark.gpu
def test_cli_directory_mode_end_to_end(modeldirs, tmp_path):
    from PIL import Image
    from tools import gen_frames
    amd = importlib.import_module("rife-ncnn-vulkan_amd")
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir(); outd.mkdir()
    a, b = gen_frames.smooth_pair(96, 64, 5)
    c, _ = gen_frames.smooth_pair(96, 64, 6)
    frames = [a, b, c]
    for i, f in enumerate(frames):
        Image.fromarray(f).save(ind / ("%03d.png" % i))
    rc = cli.main(["-i", str(ind), "-o", str(outd), "-m", modeldirs["rife-v4.6"], "-n", "5", "-j", "1:2:2", "-g", "0"])
    assert rc == 0
    outs = sorted(os.listdir(outd))
    assert outs == ["%08d.png" % i for i in range(1, 6)]
    g = amd.RIFE(0, rife_v4=True); g.load(modeldirs["rife-v4.6"])
    for i, (sx, fx) in enumerate(cli.build_schedule(3, 5)):
        got = np.asarray(Image.open(outd / ("%08d.png" % (i + 1))).convert("RGB"))
        assert np.array_equal(got, g.process(frames[sx], frames[sx + 1], fx)), i


@pytest.mark.gpu
def test_cli_file_mode(modeldirs, tmp_path):
    from PIL import Image
    from tools import gen_frames
    amd = importlib.import_module("rife-ncnn-vulkan_amd")
    a, b = gen_frames.smooth_pair(64, 48, 9)
    Image.fromarray(a).save(tmp_path / "a.png"); Image.fromarray(b).save(tmp_path / "b.png")
    assert cli.main(["-0", str(tmp_path / "a.png"), "-1", str(tmp_path / "b.png"), "-o", str(tmp_path / "o.png"), "-m", modeldirs["rife-v2.3"]]) == 0
    g = amd.RIFE(0, rife_v2=True); g.load(modeldirs["rife-v2.3"])
    assert np.array_equal(np.asarray(Image.open(tmp_path / "o.png").convert("RGB")), g.process(a, b, 0.5))


# ---- the C++ command line (rife-ncnn-vulkan_amd/rife-hip, csrc/main.cpp) ----
RIFE_HIP = os.environ.get("RIFE_HIP_BIN") or os.path.join(ROOT, "rife-ncnn-vulkan_amd", "rife-hip")      # RIFE_HIP_BIN: the sanitizer builds (tools/sanitize_run.sh)


def run_cpp(args):
    import subprocess
    p = subprocess.run([RIFE_HIP] + args, capture_output=True, text=True)
    return p.returncode, p.stderr


@pytest.mark.skipif(not os.path.exists(RIFE_HIP), reason="rife-hip is not built")
def test_cpp_cli_validation_matches_the_reference_messages():
    rc, err = run_cpp([])
    assert rc == 255 and "Usage:" in err
    rc, err = run_cpp(["-0", "a.png", "-1", "b.png", "-o", "o.png", "-s", "1.5", "-m", "rife-v4.6"])
    assert rc == 255 and "invalid timestep argument, must be 0~1" in err
    rc, err = run_cpp(["-0", "a.png", "-1", "b.png", "-o", "o.bmp", "-m", "rife-v4.6"])
    assert rc == 255 and "invalid outputpath extension type" in err
    rc, err = run_cpp(["-0", "a.png", "-1", "b.png", "-o", "o.png", "-s", "0.3", "-m", "rife-v2.3"])
    assert rc == 255 and "only rife-v4 model support custom numframe and timestep" in err
    rc, err = run_cpp(["-0", "a.png", "-1", "b.png", "-o", "o.png", "-m", "nonsense"])
    assert rc == 255 and "unknown model dir type" in err
    rc, err = run_cpp(["-0", "a.png", "-1", "b.png", "-o", "o.png", "-j", "1:2,2:2", "-m", "rife-v4.6"])
    assert rc == 255 and "invalid jobs_proc thread count argument" in err


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(RIFE_HIP), reason="rife-hip is not built")
def test_cpp_cli_directory_and_file_modes(modeldirs, tmp_path):
    """PNG in (written by PIL, adaptive filters) -> rife-hip -> PNG out (read back by PIL) equals the engine's own output."""
    from PIL import Image
    from tools import gen_frames
    amd = importlib.import_module("rife-ncnn-vulkan_amd")
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir(); outd.mkdir()
    frames = [gen_frames.smooth_pair(100, 60, 5)[0], gen_frames.smooth_pair(100, 60, 5)[1], gen_frames.smooth_pair(100, 60, 6)[0]]
    for i, f in enumerate(frames):
        Image.fromarray(f).save(ind / ("%03d.png" % i))
    model = modeldirs["rife-v4.6"]
    rc, err = run_cpp(["-i", str(ind), "-o", str(outd), "-m", model, "-n", "5", "-j", "1:2:2", "-g", "0", "-v"])
    assert rc == 0, err
    assert sorted(os.listdir(outd)) == ["%08d.png" % i for i in range(1, 6)]
    g = amd.RIFE(0, rife_v4=True); g.load(model)
    for i, (sx, fx) in enumerate(cli.build_schedule(3, 5)):
        got = np.asarray(Image.open(outd / ("%08d.png" % (i + 1))).convert("RGB"))
        assert np.array_equal(got, g.process(frames[sx], frames[sx + 1], fx)), i
    # file mode with a ppm output and an RGBA + a palette input
    Image.fromarray(frames[0]).convert("RGBA").save(tmp_path / "a.png")
    Image.fromarray(frames[1]).save(tmp_path / "b.png")
    rc, err = run_cpp(["-0", str(tmp_path / "a.png"), "-1", str(tmp_path / "b.png"), "-o", str(tmp_path / "o.ppm"), "-m", model, "-s", "0.25"])
    assert rc == 0, err
    assert np.array_equal(np.asarray(Image.open(tmp_path / "o.ppm").convert("RGB")), g.process(frames[0], frames[1], 0.25))


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(RIFE_HIP), reason="rife-hip is not built")
def test_cpp_cli_two_replicas_on_one_device_equal_one(modeldirs, tmp_path):
    """The per-device replica path (csrc/main.cpp: one RIFE object and its own proc threads per -g entry, one shared task queue;
    reference src/main.cpp:766-781, 849-866): `-g 0,0 -j 1:2,2:2` = two engines on device 0 fed from the same queue must write
    exactly the files `-g 0` writes.  Same for the Python CLI."""
    from PIL import Image
    from tools import gen_frames
    ind = tmp_path / "in"; ind.mkdir()
    for i in range(5):
        Image.fromarray(gen_frames.smooth_pair(160, 96, 20 + i)[0]).save(ind / ("%03d.png" % i))
    model = modeldirs["rife-v4.6"]
    outs = {}
    import re
    os.environ["RIFE_HIP_CLI_TIMING"] = "1"                    # the summary lines on stderr carry the number of tasks every replica took
    try:
        for name, extra in (("one", ["-g", "0", "-j", "1:2:2"]), ("two", ["-g", "0,0", "-j", "1:2,2:2"])):
            outd = tmp_path / name; outd.mkdir()
            rc, err = run_cpp(["-i", str(ind), "-o", str(outd), "-m", model, "-n", "29"] + extra)
            assert rc == 0, err
            outs[name] = outd
            took = [int(x) for x in re.findall(r"timing: replica \d+ \(gpu 0\) took (\d+) task", err)]
            assert len(took) == (2 if name == "two" else 1) and sum(took) == 29 and min(took) >= 1, err      # both replicas worked (src/main.cpp:849-866)
    finally:
        del os.environ["RIFE_HIP_CLI_TIMING"]
    names = sorted(os.listdir(outs["one"]))
    assert names == ["%08d.png" % i for i in range(1, 30)] and sorted(os.listdir(outs["two"])) == names
    for n in names:
        a = np.asarray(Image.open(outs["one"] / n).convert("RGB")); b = np.asarray(Image.open(outs["two"] / n).convert("RGB"))
        assert np.array_equal(a, b), n
    outp = tmp_path / "py"; outp.mkdir()
    assert cli.main(["-i", str(ind), "-o", str(outp), "-m", model, "-n", "29", "-g", "0,0", "-j", "1:2,2:2"]) == 0
    for n in names:
        assert np.array_equal(np.asarray(Image.open(outp / n).convert("RGB")), np.asarray(Image.open(outs["one"] / n).convert("RGB"))), n


@pytest.mark.skipif(not os.path.exists(RIFE_HIP), reason="rife-hip is not built")
def test_cpp_cli_png_and_ppm_codecs_round_trip(tmp_path):
    """rife-hip's own PNG reader (all five filter types, RGB / RGBA / gray / palette) and writer against PIL, no GPU involved."""
    import subprocess
    from PIL import Image
    from tools import gen_frames
    a, _ = gen_frames.smooth_pair(101, 67, 3)
    n, _ = gen_frames.noise_pair(64, 33, 4)
    cases = {"rgb": Image.fromarray(a), "noise": Image.fromarray(n), "rgba": Image.fromarray(a).convert("RGBA"), "gray": Image.fromarray(a).convert("L"),
             "pal": Image.fromarray(a).convert("P", palette=Image.ADAPTIVE, colors=64)}
    for name, im in cases.items():
        src = tmp_path / (name + ".png")
        im.save(src, optimize=(name == "rgb"))
        want = np.asarray(im.convert("RGB"))
        for ext in ("png", "ppm", "webp"):
            dst = tmp_path / (name + "_out." + ext)
            p = subprocess.run([RIFE_HIP, "--transcode", str(src), str(dst)], capture_output=True, text=True)
            assert p.returncode == 0, (name, ext, p.stderr)
            assert np.array_equal(np.asarray(Image.open(dst).convert("RGB")), want), (name, ext)
    # a frame big enough for several deflate bands (the PNG writer compresses ~1 MB bands on helper threads and concatenates them)
    big = np.kron(gen_frames.smooth_pair(320, 180, 5)[0], np.ones((4, 4, 1), np.uint8)) + gen_frames.noise_pair(1280, 720, 6)[0] % 3
    Image.fromarray(big).save(tmp_path / "big.ppm")
    p = subprocess.run([RIFE_HIP, "--transcode", str(tmp_path / "big.ppm"), str(tmp_path / "big.png")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    im = Image.open(tmp_path / "big.png")
    im.verify()                                                            # chunk CRCs
    assert np.array_equal(np.asarray(Image.open(tmp_path / "big.png").convert("RGB")), big)       # zlib stream incl. the combined Adler-32
    p = subprocess.run([RIFE_HIP, "--transcode", str(tmp_path / "big.png"), str(tmp_path / "big2.ppm")], capture_output=True, text=True)
    assert p.returncode == 0 and np.array_equal(np.asarray(Image.open(tmp_path / "big2.ppm")), big)
    Image.fromarray(a).save(tmp_path / "in.webp", lossless=True)          # WebP decode (lossless source -> exact pixels)
    p = subprocess.run([RIFE_HIP, "--transcode", str(tmp_path / "in.webp"), str(tmp_path / "from_webp.png")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert np.array_equal(np.asarray(Image.open(tmp_path / "from_webp.png").convert("RGB")), a)
    p = subprocess.run([RIFE_HIP, "--transcode", str(tmp_path / "missing.png"), str(tmp_path / "x.png")], capture_output=True, text=True)
    assert p.returncode == 1


@pytest.mark.skipif(not os.path.exists(RIFE_HIP), reason="rife-hip is not built")
def test_cpp_cli_jpeg_codec_against_libjpeg(tmp_path):
    """csrc/jpeg_codec.h (baseline decode with libjpeg-style fancy chroma upsampling; quality-100 encode) against PIL's libjpeg."""
    import subprocess
    from PIL import Image
    from tools import gen_frames
    for (w, h) in ((203, 117), (64, 48), (17, 9)):
        a, _ = gen_frames.smooth_pair(w, h, 3)
        variants = {"q95_420": dict(quality=95), "q100_444": dict(quality=100, subsampling=0), "q90_422": dict(quality=90, subsampling=1),
                    "q75_opt": dict(quality=75, optimize=True), "gray": dict(quality=95)}
        for name, kw in variants.items():
            src = tmp_path / ("%s_%d.jpg" % (name, w))
            (Image.fromarray(a).convert("L") if name == "gray" else Image.fromarray(a)).save(src, **kw)
            dst = tmp_path / ("%s_%d.png" % (name, w))
            p = subprocess.run([RIFE_HIP, "--transcode", str(src), str(dst)], capture_output=True, text=True)
            assert p.returncode == 0, (name, p.stderr)
            d = np.abs(np.asarray(Image.open(dst).convert("RGB")).astype(int) - np.asarray(Image.open(src).convert("RGB")).astype(int))
            assert d.max() <= 3 and d.mean() < 0.1, (name, w, int(d.max()), float(d.mean()))
        # encoder: quality 100, 4:4:4 -> PIL must read back (almost) the original
        Image.fromarray(a).save(tmp_path / "src.png")
        p = subprocess.run([RIFE_HIP, "--transcode", str(tmp_path / "src.png"), str(tmp_path / "enc.jpg")], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        d = np.abs(np.asarray(Image.open(tmp_path / "enc.jpg").convert("RGB")).astype(int) - a.astype(int))
        assert d.max() <= 4 and d.mean() < 1.0, (w, int(d.max()), float(d.mean()))
    # progressive DCT (spectral selection + successive approximation, what stb_image in the reference also reads)
    for (w, h) in ((203, 117), (64, 48), (17, 9), (640, 360)):
        a = gen_frames.smooth_pair(w, h, 5)[0] if w < 600 else (np.kron(gen_frames.smooth_pair(160, 90, 6)[0], np.ones((4, 4, 1), np.uint8)) + gen_frames.noise_pair(w, h, 7)[0] % 32)
        variants = {"p95_420": dict(quality=95), "p100_444": dict(quality=100, subsampling=0), "p85_422": dict(quality=85, subsampling=1),
                    "p60_opt": dict(quality=60, optimize=True), "pgray": dict(quality=90),
                    "p90_rst": dict(quality=90, restart_marker_blocks=3), "p90_rst_444": dict(quality=90, restart_marker_rows=1, subsampling=0)}
        for name, kw in variants.items():
            src = tmp_path / ("%s_%d.jpg" % (name, w))
            (Image.fromarray(a).convert("L") if name == "pgray" else Image.fromarray(a)).save(src, progressive=True, **kw)
            dst = tmp_path / ("%s_%d.png" % (name, w))
            p = subprocess.run([RIFE_HIP, "--transcode", str(src), str(dst)], capture_output=True, text=True)
            assert p.returncode == 0, (name, w, p.stderr)
            d = np.abs(np.asarray(Image.open(dst).convert("RGB")).astype(int) - np.asarray(Image.open(src).convert("RGB")).astype(int))
            assert d.max() <= 3 and d.mean() < 0.1, (name, w, int(d.max()), float(d.mean()))
    # arithmetic-coded / lossless files are still refused with a message
    bad = bytearray(open(tmp_path / "p95_420_64.jpg", "rb").read())
    i = bad.find(b"\xff\xc2")
    bad[i + 1] = 0xC9
    open(tmp_path / "arith.jpg", "wb").write(bytes(bad))
    p = subprocess.run([RIFE_HIP, "--transcode", str(tmp_path / "arith.jpg"), str(tmp_path / "arith.png")], capture_output=True, text=True)
    assert p.returncode == 1 and "not supported" in p.stderr


@pytest.mark.skipif(not os.path.exists(RIFE_HIP), reason="rife-hip is not built")
def test_cpp_cli_reads_every_input_format_of_the_reference(tmp_path):
    """The reference decodes inputs with stb_image (jpeg, png, bmp, pnm enabled: src/main.cpp:14-20): low-depth / 16-bit / Adam7 PNGs,
    24- / 32-bit / palette BMPs and P5 greymaps must read like stb reads them (3 channels, 16 -> 8 bits by the high byte, no alpha)."""
    import struct
    import subprocess
    import zlib
    from PIL import Image
    from tools import gen_frames
    a = gen_frames.smooth_pair(61, 37, 3)[0]
    h, w, _ = a.shape

    def transcode(src):
        dst = tmp_path / "o.ppm"
        p = subprocess.run([RIFE_HIP, "--transcode", str(src), str(dst)], capture_output=True, text=True)
        assert p.returncode == 0, (src, p.stderr)
        return np.asarray(Image.open(dst))

    def via_pil(name, im, **kw):
        src = tmp_path / name
        im.save(src, **kw)
        assert np.array_equal(transcode(src), np.asarray(Image.open(src).convert("RGB"))), name
    via_pil("b1.png", Image.fromarray(a).convert("1"))
    via_pil("p4.png", Image.fromarray(a).convert("P", palette=Image.ADAPTIVE, colors=16), bits=4)
    via_pil("p2.png", Image.fromarray(a).convert("P", palette=Image.ADAPTIVE, colors=4), bits=2)
    via_pil("t24.bmp", Image.fromarray(a))
    via_pil("t32.bmp", Image.fromarray(a).convert("RGBA"))
    via_pil("t8.bmp", Image.fromarray(a).convert("P", palette=Image.ADAPTIVE, colors=200))
    via_pil("t.pgm", Image.fromarray(a).convert("L"))
    g16 = a[:, :, 0].astype(np.uint16) * 257 + 13
    Image.fromarray(g16).save(tmp_path / "g16.png")
    assert np.array_equal(transcode(tmp_path / "g16.png")[:, :, 0], (g16 >> 8).astype(np.uint8))

    # hand-made files for what PIL cannot write: Adam7 interlace, 16-bit colour, 4-bit grey
    def png(depth, ctype, interlace, raw):
        def chunk(t, b):
            return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xffffffff)
        return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace)) + chunk(b"IDAT", zlib.compress(raw)) +
                chunk(b"IEND", b""))

    def rows(img, depth):
        out = b""
        for row in img:
            if depth == 8:
                b = row.astype(np.uint8).tobytes()
            elif depth == 16:
                b = row.astype(">u2").tobytes()
            else:
                flat = row.reshape(-1)
                per = 8 // depth
                flat = np.concatenate([flat, np.zeros((-len(flat)) % per, int)])
                b = bytes(int(sum(int(v) << ((per - 1 - i) * depth) for i, v in enumerate(flat[k:k + per]))) for k in range(0, len(flat), per))
            out += b"\x00" + b
        return out

    def adam7(img, depth):
        x0, y0, dx, dy = [0, 4, 0, 2, 0, 1, 0], [0, 0, 4, 0, 2, 0, 1], [8, 8, 4, 4, 2, 2, 1], [8, 8, 8, 4, 4, 2, 2]
        out = b""
        for p in range(7):
            sub = img[y0[p]::dy[p], x0[p]::dx[p]]
            if sub.shape[0] and sub.shape[1]:
                out += rows(sub, depth)
        return out
    a16 = a.astype(int) * 257
    g4 = (a[:, :, :1] >> 4).astype(int)
    cases = {"i8.png": (png(8, 2, 1, adam7(a.astype(int), 8)), a), "i16.png": (png(16, 2, 1, adam7(a16, 16)), a),
             "rgba16.png": (png(16, 6, 0, rows(np.concatenate([a16, np.full((h, w, 1), 65535)], 2), 16)), a),
             "g4.png": (png(4, 0, 0, rows(g4, 4)), np.repeat(g4 * 17, 3, axis=2)), "g4i.png": (png(4, 0, 1, adam7(g4, 4)), np.repeat(g4 * 17, 3, axis=2))}
    for name, (data, want) in cases.items():
        (tmp_path / name).write_bytes(data)
        assert np.array_equal(transcode(tmp_path / name), want.astype(np.uint8)), name
    assert np.array_equal(np.asarray(Image.open(tmp_path / "i8.png").convert("RGB")), a)          # the generator itself is a valid PNG writer


@pytest.mark.skipif(not os.path.exists(RIFE_HIP), reason="rife-hip is not built")
def test_cpp_cli_decoders_survive_corrupt_files(tmp_path):
    """Truncated and bit-flipped png / jpg (baseline + progressive) / bmp / pnm inputs: decode either works or fails with exit code 1 -
    no crash, no hang, no giant allocation from a lying header (found with an ASan build of main.cpp, kept as a regression net)."""
    import random
    import struct
    import subprocess
    from PIL import Image
    from tools import gen_frames
    a = gen_frames.smooth_pair(97, 61, 3)[0]
    files = {"a.png": {}, "a.jpg": dict(quality=90), "p.jpg": dict(quality=90, progressive=True), "a.bmp": {}, "a.ppm": {}}
    rng = random.Random(7)
    for name, kw in files.items():
        Image.fromarray(a).save(tmp_path / name, **kw)
        data = (tmp_path / name).read_bytes()
        cases = [data[:k] for k in (0, 1, 2, 8, 20, 40, len(data) // 3, len(data) // 2, len(data) - 5, len(data) - 1)]
        for _ in range(25):
            b = bytearray(data)
            for _ in range(rng.choice((1, 1, 2, 4, 8))):
                i = rng.randrange(min(len(b), 700)) if rng.random() < 0.6 else rng.randrange(len(b))      # mostly headers: that is where the structure is
                b[i] = rng.randrange(256)
            cases.append(bytes(b))
        if name == "a.png":       # the case that used to hang: a width of 167 million pixels
            cases.append(data[:16] + struct.pack(">I", 167772257) + data[20:])
        for k, c in enumerate(cases):
            (tmp_path / "t.bin").write_bytes(c)
            p = subprocess.run([RIFE_HIP, "--transcode", str(tmp_path / "t.bin"), str(tmp_path / "o.ppm")], capture_output=True, text=True, timeout=30)
            assert p.returncode in (0, 1), (name, k, p.returncode, p.stderr[-200:])


def _jpeg_segments(data):
    """[(marker, start, end)] of the header segments of a JPEG up to and including the first SOS header (start = the 0xFF byte)."""
    out, i = [], 2
    while i + 4 <= len(data) and data[i] == 0xFF:
        m = data[i + 1]
        n = (data[i + 2] << 8) | data[i + 3]
        out.append((m, i, i + 2 + n))
        if m == 0xDA:
            break
        i += 2 + n
    return out


@pytest.mark.skipif(not os.path.exists(RIFE_HIP), reason="rife-hip is not built")
def test_cpp_cli_decoders_refuse_crafted_headers(tmp_path):
    """(1) a second SOF after the first scan of a multi-scan JPEG re-sized w / h / sampling factors under the coefficient
    store (ASan: heap overflow in finish()): now "duplicate SOF"; (2) an IHDR chunk shorter than 13 bytes was read past its end;
    (3) a grey JPEG whose header carries 2x2 sampling factors is a non-interleaved scan and must decode like libjpeg decodes it."""
    import struct
    import subprocess
    from PIL import Image
    from tools import gen_frames
    a = gen_frames.smooth_pair(96, 64, 5)[0]

    def run(data):
        (tmp_path / "t.bin").write_bytes(data)
        return subprocess.run([RIFE_HIP, "--transcode", str(tmp_path / "t.bin"), str(tmp_path / "o.ppm")], capture_output=True, text=True, timeout=30)

    Image.fromarray(a).save(tmp_path / "p.jpg", quality=90, progressive=True)
    data = (tmp_path / "p.jpg").read_bytes()
    segs = _jpeg_segments(data)
    sof = next(s for s in segs if s[0] == 0xC2)
    sos = next(s for s in segs if s[0] == 0xDA)
    # entropy-coded data of the first scan ends at the next marker that is not a stuffed 0xFF00 / RSTn
    j = sos[2]
    while not (data[j] == 0xFF and data[j + 1] != 0 and not 0xD0 <= data[j + 1] <= 0xD7):
        j += 1
    big = bytearray(data[sof[1]:sof[2]])
    big[5:9] = struct.pack(">HH", 512, 512)                   # enlarge h, w
    swapped = bytearray(data[sof[1]:sof[2]])
    swapped[11], swapped[14] = swapped[14], swapped[11]       # swap the sampling factors of components 0 and 1
    for dup in (bytes(big), bytes(swapped), data[sof[1]:sof[2]]):
        p = run(data[:j] + dup + data[j:])
        assert p.returncode == 1 and "duplicate SOF" in (p.stderr + p.stdout), (p.returncode, p.stderr[-200:])
    assert run(data).returncode == 0                          # the unmodified file still decodes

    Image.fromarray(a).save(tmp_path / "a.png")
    png = (tmp_path / "a.png").read_bytes()
    short = png[:8] + struct.pack(">I", 5) + b"IHDR" + png[16:21] + b"\0\0\0\0"      # a 5-byte IHDR right before the end of the file
    assert run(short).returncode == 1
    late = png[:8] + png[33:45] + png[8:33] + png[45:]       # IHDR not the first chunk
    assert run(late).returncode in (0, 1)

    g = np.asarray(Image.fromarray(a).convert("L"))
    Image.fromarray(g).save(tmp_path / "g.jpg", quality=95)
    gd = bytearray((tmp_path / "g.jpg").read_bytes())
    gs = next(s for s in _jpeg_segments(bytes(gd)) if s[0] == 0xC0)
    assert gd[gs[1] + 9] == 1                                  # one component
    gd[gs[1] + 11] = 0x22                                      # ... now with 2x2 sampling factors in the header
    p = run(bytes(gd))
    assert p.returncode == 0, p.stderr[-200:]
    with open(tmp_path / "o.ppm", "rb") as f:
        assert f.readline() == b"P6\n"
        wh = f.readline().split()
        f.readline()
        got = np.frombuffer(f.read(), np.uint8).reshape(int(wh[1]), int(wh[0]), 3)
    want = np.asarray(Image.open(tmp_path / "g.jpg").convert("RGB")).astype(int)
    assert np.abs(got.astype(int) - want).max() <= 3          # same tolerance as the other JPEG-vs-libjpeg checks
