"""The sanitizer job of the host side (tools/sanitize_run.sh): csrc/main.cpp + csrc/rife.cpp + the codecs built with ASan + UBSan and with TSan against
tests/sanitize/stub_engine.cpp (a host-only stand-in for librife_hip.so behind the same C-ABI; no GPU needed), run over the corrupt / crafted decoder corpus of
tests/test_cli.py, the 3-stage pipeline with two replicas in png / jpg / ppm, the band-parallel PNG writer at 1080p and the error paths.  Any sanitizer report
fails the job.  A recorded run is in profiles/r5/sanitize.txt."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_host_side_is_clean_under_asan_ubsan_and_tsan(tmp_path):
    probe = tmp_path / "p.cpp"
    probe.write_text("int main() { return 0; }\n")
    for flag in ("-fsanitize=address,undefined", "-fsanitize=thread"):
        if subprocess.run(["g++", flag, str(probe), "-o", str(tmp_path / "p")], capture_output=True).returncode != 0:
            pytest.skip("this g++ has no %s runtime" % flag)
    p = subprocess.run(["bash", os.path.join(ROOT, "tools", "sanitize_run.sh")], capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0 and "== result: CLEAN" in p.stdout, p.stdout[-3000:] + p.stderr[-1500:]
    assert "DIFFERS" not in p.stdout and p.stdout.count("== the 1-replica outputs") == 6
