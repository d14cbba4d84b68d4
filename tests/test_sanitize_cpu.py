"""Sanitizer builds of the product's host-only C: `make -C ceracoder_amd/csrc sanitize` compiles
h264_host.c, ratecontrol.c and tsmux.c with -fsanitize=address,undefined and with -fsanitize=thread around
san_driver.c.  Inputs are real macroblock records and levels (from the oracle encoder on the synthetic clip); the access
unit the sanitized writer produces must equal the one the shipped library writes, on 1, 3 and 8 entropy-coding threads."""
import os
import struct
import subprocess

import numpy as np
import pytest

from ceracoder_amd import enc as E
from ceracoder_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ceracoder_amd", "csrc")


@pytest.fixture(scope="module")
def san():
    r = subprocess.run(["make", "-C", CSRC, "sanitize"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return os.path.join(CSRC, "san", "san_asan"), os.path.join(CSRC, "san", "san_tsan")


@pytest.fixture(scope="module")
def cases(tmp_path_factory, oracle):
    """An IDR and two P pictures of a 320x192 clip: case files for the driver + the access units the shipped library writes."""
    d = tmp_path_factory.mktemp("san")
    w, h, qp, fps = 320, 192, 26, 30
    oe = oracle.Encoder(w, h, fps=fps, gop=30, threads=4)
    out = []
    for i, (y, uv) in enumerate(synth.s2_frames(w, h, 3)):
        _, idr = oe.encode(y, uv, qp)
        mbi, lev = oe.mbinfo, oe.levels
        path = str(d / ("case%d.bin" % i))
        with open(path, "wb") as f:
            f.write(struct.pack("<10i", oe.mbw, oe.mbh, int(idr), i, 0, qp, 0, w, h, fps))
            f.write(mbi.tobytes())
            f.write(np.ascontiguousarray(lev, np.int16).tobytes())
        au = (E.host_write_headers(w, h, fps) if idr else b"") + E.host_write_slice(oe.mbw, oe.mbh, idr, i, 0, qp, mbi, lev)
        out.append((path, au))
    return out


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_host_code_is_clean_under_asan_and_ubsan(san, cases, tmp_path, threads):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    for path, au in cases:
        outp = str(tmp_path / "out.bin")
        r = subprocess.run([san[0], "code", path, outp, str(threads)], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-3000:]
        assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
        data = open(outp, "rb").read()
        assert data[:len(au)] == au                       # the sanitized writer and the shipped one agree
        assert (len(data) - len(au)) % 188 == 0 and data[len(au)] == 0x47


def test_setter_thread_against_row_parallel_coding_is_clean_under_tsan(san, cases):
    """The element's threading: setpoints from the control thread (atomic), the streaming thread latching them into the rate
    control and coding pictures on 8 row-parallel threads (mutex / condition-variable hand-off)."""
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0:second_deadlock_stack=1")
    for path, _ in cases[:2]:
        r = subprocess.run([san[1], "race", path, "150"], env=env, capture_output=True, text=True, timeout=600)
        assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
        assert r.returncode == 0, r.stderr[-3000:]
        assert '"stable":true' in r.stdout
