#!/bin/bash
# ASan + UBSan and TSan job for the HOST side (csrc/main.cpp, rife.cpp, jpeg_codec.h, the PNG band writer); no GPU needed: the engine behind the C-ABI is
# tests/sanitize/stub_engine.cpp.  Writes a log to stdout; exit code != 0 if any sanitizer reported.
#   tools/sanitize_run.sh > profiles/r5/sanitize.txt
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
make -s -C rife-ncnn-vulkan_amd/csrc sanitize || exit 2
ASAN=$ROOT/rife-ncnn-vulkan_amd/rife-hip-asan
TSAN=$ROOT/rife-ncnn-vulkan_amd/rife-hip-tsan
export ASAN_OPTIONS=exitcode=99:detect_leaks=1:abort_on_error=0
export UBSAN_OPTIONS=halt_on_error=1:exitcode=98:print_stacktrace=1
export TSAN_OPTIONS=halt_on_error=1:exitcode=66:second_deadlock_stack=1
FAIL=0
echo "== $(g++ --version | head -1); $(date -u +%Y-%m-%dT%H:%MZ); $(nproc) cores"
echo "== 1. decoder corpus of tests/test_cli.py (truncated / bit-flipped / crafted png, jpg baseline + progressive, bmp, pnm; PNG colour types; codec round trips) through the ASan + UBSan binary"
echo "   (a sanitizer report exits 99 / 98, which the tests' 'returncode in (0, 1)' assertions reject)"
RIFE_HIP_BIN=$ASAN python -m pytest tests/test_cli.py -q -m "not gpu" -k "cpp_cli" 2>&1 | tail -4
[ ${PIPESTATUS[0]} -eq 0 ] || FAIL=1
T=$(mktemp -d)
python - $T <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from PIL import Image
from tools import gen_frames
t = sys.argv[1]
for sub in ("png", "jpg", "ppm"):
    os.makedirs(os.path.join(t, "in_" + sub))
for i in range(7):
    a = gen_frames.smooth_pair(333, 241, 40 + i)[i & 1]           # ragged size: PNG filter rows / JPEG edge MCUs
    Image.fromarray(a).save(os.path.join(t, "in_png", "%03d.png" % i))
    Image.fromarray(a).save(os.path.join(t, "in_jpg", "%03d.jpg" % i), quality=92, progressive=bool(i & 1))
    Image.fromarray(a).save(os.path.join(t, "in_ppm", "%03d.ppm" % i))
big = gen_frames.smooth_pair(1920, 1080, 9)                       # large enough for the band-parallel PNG writer (1 MB bands)
os.makedirs(os.path.join(t, "in_big"))
for i in range(3):
    Image.fromarray(big[i & 1]).save(os.path.join(t, "in_big", "%03d.png" % i))
PY
run() {   # name binary args...
    local name=$1; shift
    "$@" > $T/log.txt 2>&1
    local rc=$?
    local nout=$(ls $T/out 2>/dev/null | wc -l)
    echo "   $name: rc $rc, $nout files"
    if [ $rc -ne 0 ]; then FAIL=1; tail -30 $T/log.txt; fi
}
echo "== 2. directory mode under TSan: 3-stage pipeline, two replicas (-g 0,1 and -g 0,0), several load / proc / save threads; outputs must equal the 1-replica run"
for fmt in png jpg ppm; do
    rm -rf $T/out $T/ref; mkdir -p $T/out
    run "tsan $fmt -g 0 -j 1:2:2" $TSAN -i $T/in_$fmt -o $T/out -m rife-v4.6 -n 19 -f %08d.$fmt -g 0 -j 1:2:2
    mv $T/out $T/ref; mkdir -p $T/out
    run "tsan $fmt -g 0,1 -j 3:2,3:4" $TSAN -i $T/in_$fmt -o $T/out -m rife-v4.6 -n 19 -f %08d.$fmt -g 0,1 -j 3:2,3:4
    diff -rq $T/ref $T/out > /dev/null && echo "      == the 1-replica outputs" || { echo "      DIFFERS from the 1-replica outputs"; FAIL=1; }
    rm -rf $T/out; mkdir -p $T/out
    run "tsan $fmt -g 0,0 -j 2:1,2:3" $TSAN -i $T/in_$fmt -o $T/out -m rife-v4.6 -n 19 -f %08d.$fmt -g 0,0 -j 2:1,2:3
    diff -rq $T/ref $T/out > /dev/null && echo "      == the 1-replica outputs" || { echo "      DIFFERS from the 1-replica outputs"; FAIL=1; }
done
echo "== 3. the same under ASan + UBSan (heap / bounds / UB in the pipeline, the frame cache and the codecs), and the band-parallel PNG writer at 1920x1080 under both"
for fmt in png jpg ppm; do
    rm -rf $T/out; mkdir -p $T/out
    run "asan $fmt -g 0,1 -j 3:2,3:4" $ASAN -i $T/in_$fmt -o $T/out -m rife-v4.6 -n 19 -f %08d.$fmt -g 0,1 -j 3:2,3:4
done
rm -rf $T/out; mkdir -p $T/out
run "asan 1080p png (band writer)" $ASAN -i $T/in_big -o $T/out -m rife-v4.6 -n 5 -g 0 -j 1:2:2
rm -rf $T/out; mkdir -p $T/out
run "tsan 1080p png (band writer)" $TSAN -i $T/in_big -o $T/out -m rife-v4.6 -n 5 -g 0 -j 2:2:2
python - $T <<'PY'
import sys, os, numpy as np
from PIL import Image
t = sys.argv[1]
a = np.asarray(Image.open(os.path.join(t, "in_big", "000.png")).convert("RGB")); b = np.asarray(Image.open(os.path.join(t, "in_big", "001.png")).convert("RGB"))
o = np.asarray(Image.open(os.path.join(t, "out", sorted(os.listdir(os.path.join(t, "out")))[1])).convert("RGB"))
want = (np.float32(0.4) * a.astype(np.float32) + np.float32(0.6) * b.astype(np.float32) + np.float32(0.5)).astype(np.uint8)      # -n 5 over 3 frames: output 2 = frames 0, 1 at timestep 0.6
print("   band-written 1080p PNG decodes (PIL) to the stub's blend of its inputs:", bool(np.abs(o.astype(int) - want.astype(int)).max() <= 1))
PY
echo "== 4. single-pair mode, error paths (missing file, size mismatch, bad extension) under ASan + UBSan"
rm -rf $T/out; mkdir -p $T/out
run "asan pair" $ASAN -0 $T/in_png/000.png -1 $T/in_png/001.png -o $T/out/o.png -m rife-v4.6 -s 0.3
$ASAN -0 $T/in_png/000.png -1 $T/nothing.png -o $T/out/o2.png -m rife-v4.6 > $T/log.txt 2>&1; rc=$?; echo "   asan missing input: rc $rc (sanitizer exit codes are 99 / 98)"; [ $rc -ge 98 ] && [ $rc -le 99 ] && FAIL=1
$ASAN -0 $T/in_png/000.png -1 $T/in_big/000.png -o $T/out/o3.png -m rife-v4.6 > $T/log.txt 2>&1; rc=$?; echo "   asan size mismatch: rc $rc"; [ $rc -ge 98 ] && [ $rc -le 99 ] && FAIL=1
rm -rf $T
echo "== result: $([ $FAIL -eq 0 ] && echo CLEAN || echo FAILED)"
exit $FAIL
