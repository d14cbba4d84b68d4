"""The oracle pinned by its independent decoder: decode(encode(x)) == encoder reconstruction,
bit for bit, over seeded sequences (SURVEY.md 8c KAT 5), plus golden access-unit digests."""
import hashlib
import json
import os

import numpy as np
import pytest

from ceracoder_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "oracle_stream_digests.json")


def run(oracle, w, h, n, qps, gop=4, kind="s2", threads=4):
    enc, dec = oracle.Encoder(w, h, gop=gop, threads=threads), oracle.Decoder()
    gen = synth.s2_frames(w, h, n) if kind == "s2" else synth.s3_frames(w, h, n)
    digest = hashlib.sha256()
    sizes = []
    for i, (y, uv) in enumerate(gen):
        au, idr = enc.encode(y, uv, qps[i % len(qps)])
        assert idr == (i % gop == 0)
        dy, duv = dec.decode(au)
        assert np.array_equal(dy, enc.recon_y), (i, "luma")
        assert np.array_equal(duv, enc.recon_uv), (i, "chroma")
        digest.update(au)
        sizes.append(len(au))
    assert dec.size == (w, h)
    return digest.hexdigest(), sizes


@pytest.mark.parametrize("qp", [0, 10, 26, 40, 51])
@pytest.mark.parametrize("w,h", [(16, 16), (64, 48), (50, 34), (176, 144)])
def test_encoder_recon_equals_decoder_output(oracle, w, h, qp):
    run(oracle, w, h, 6, [qp])


@pytest.mark.parametrize("qp", [0, 14, 30, 44])
def test_high_profile_8x8_transform_roundtrip(oracle, qp):
    """transform_8x8_mode on: High-profile SPS/PPS, transform_size_8x8_flag, 8x8 CAVLC interleave, 8.5.13 inverse,
    deblocking without edges 1/3 -- the independent decoder must still reproduce the encoder bit for bit."""
    oracle.set_transform8x8(True)
    try:
        run(oracle, 176, 144, 6, [qp], gop=5)
        run(oracle, 50, 34, 4, [qp, 51 - qp // 2], gop=3, kind="s3")
    finally:
        oracle.set_transform8x8(False)


def test_varying_qp_and_noise(oracle):
    run(oracle, 96, 80, 8, [51, 0, 30, 12, 44, 3], gop=3, kind="s3")


@pytest.mark.parametrize("w,h", [(1280, 720), (1920, 1080)])
def test_full_size_pictures(oracle, w, h):
    run(oracle, w, h, 2, [30, 33], gop=60, threads=8)


def test_flat_and_extreme_inputs(oracle):
    """All-black, all-white and a hard vertical edge: DC-only paths, P_Skip runs, bS 0 everywhere."""
    w, h = 64, 64
    enc, dec = oracle.Encoder(w, h, gop=3), oracle.Decoder()
    pics = [np.zeros((h, w), np.uint8), np.zeros((h, w), np.uint8), np.full((h, w), 255, np.uint8), np.full((h, w), 255, np.uint8)]
    edge = np.zeros((h, w), np.uint8); edge[:, w // 2:] = 255
    pics += [edge, edge]
    for i, y in enumerate(pics):
        uv = np.full((h // 2, w), 128, np.uint8)
        au, _ = enc.encode(y, uv, 24)
        dy, duv = dec.decode(au)
        assert np.array_equal(dy, enc.recon_y) and np.array_equal(duv, enc.recon_uv)
        if i in (1, 3, 5):  # a repeated picture costs (almost) nothing: skip runs and zero residuals
            assert len(au) < 100, len(au)
            assert np.abs(dy.astype(int) - y).max() <= 3


def test_golden_stream_digests(oracle):
    """Committed digests of oracle access units: the oracle may not drift silently between rounds."""
    cases = {"s2_176x144_qp28": (176, 144, 5, [28], "s2"), "s2_64x48_mixqp": (64, 48, 7, [30, 20, 40], "s2"), "s3_48x32_qp6": (48, 32, 4, [6], "s3")}
    got = {k: run(oracle, w, h, n, q, kind=kind)[0] for k, (w, h, n, q, kind) in cases.items()}
    if not os.path.exists(GOLD):
        pytest.skip("golden digests not generated yet: run tests/golden/make_oracle_digests.py")
    want = json.load(open(GOLD))
    assert got == want


def test_decoder_rejects_garbage(oracle):
    dec = oracle.Decoder()
    with pytest.raises(RuntimeError):
        dec.decode(bytes([0, 0, 0, 1, 0x65, 0x88, 0x84, 0x00]))


def test_scene_cut_recovery_rule(oracle):
    """A hard cut at picture 5 of a GOP of 30: the cut picture and the next stay P, picture 7 becomes IDR (the rule of
    orc_enc_frame / mi355enc collect()); without the option the GOP runs its length.  The stream still decodes to the
    encoder's reconstruction."""
    from tests.util import cut_clip
    w, h = 176, 144
    clip = cut_clip(w, h, 12, 5)
    for sc, want in ((True, [0, 7]), (False, [0])):
        oe = oracle.Encoder(w, h, gop=30, threads=4, scenecut=sc)
        dec = oracle.Decoder()
        idrs = []
        for i, (y, uv) in enumerate(clip):
            au, key = oe.encode(y, uv, 30)
            if key:
                idrs.append(i)
            dy, duv = dec.decode(au)
            assert np.array_equal(dy, oe.recon_y) and np.array_equal(duv, oe.recon_uv)
        assert idrs == want, (sc, idrs)


def test_scene_cut_recovery_lag_and_skip_runs(oracle):
    """orc_enc_set_sc_lag(3) (what the device does with three pictures in flight): the IDR picture lands one picture later.  And a
    picture that follows P_Skip-run pictures is searched against an older source: its cost neither triggers a cut nor enters
    the mean -- a cadence of coded and skipped pictures (rate control below the ladder) must not force IDR pictures."""
    from tests.util import cut_clip
    from ceracoder_amd import synth
    w, h = 176, 144
    clip = cut_clip(w, h, 12, 5)
    oe = oracle.Encoder(w, h, gop=30, threads=4, sc_lag=3)
    idrs = [i for i, (y, uv) in enumerate(clip) if oe.encode(y, uv, 30)[1]]
    assert idrs == [0, 8], idrs
    fr = list(synth.s2_frames(w, h, 24))
    for skip_runs, want in ((True, [0]), (False, None)):
        oe = oracle.Encoder(w, h, gop=60, threads=4)
        idrs = []
        for i, (y, uv) in enumerate(fr):
            drop = oracle.DROP_SKIP if (skip_runs and i > 3 and i % 4 != 0) else 0   # every fourth picture coded, the others P_Skip runs
            if oe.encode(y, uv, 30, drop=drop)[1]:
                idrs.append(i)
        if want is not None:
            assert idrs == want, idrs
