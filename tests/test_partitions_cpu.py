"""Inter partitions (P_L0_L0_16x8, P_L0_L0_8x16, P_8x8 with four P_L0_8x8), oracle side and host writer -- the groundwork for the
device stage: the oracle encoder's decision (ORC_F_PART, off by default), its syntax, and the independent decoder, which derives every partition's
predictor from 6.4.11.7 / 8.4.1.3 on its own.  No GPU."""
import numpy as np
import pytest

from ceracoder_amd import synth


def _encode(oracle, w, h, n, qp, feat, gop=30, clip=None):
    oracle.set_features(feat)
    try:
        e, d = oracle.Encoder(w, h, gop=gop), oracle.Decoder()
        out, shapes = [], np.zeros(4, int)
        for i, (y, uv) in enumerate(clip or synth.s2_frames(w, h, n)):
            au, key = e.encode(y, uv, qp)
            dy, duv = d.decode(au)
            assert np.array_equal(dy, e.recon_y) and np.array_equal(duv, e.recon_uv), (w, h, qp, i)
            out.append(au)
            if not key:
                m = e.mbinfo
                inter = m["mb_type"] == 1
                for s in range(4):
                    shapes[s] += int(((m["i16_mode"] == s) & inter).sum())
        return out, shapes, e
    finally:
        oracle.set_features(oracle.F_ALL)


@pytest.mark.parametrize("w,h,n", [(64, 48, 8), (176, 144, 6), (322, 182, 5), (640, 368, 4)])
@pytest.mark.parametrize("qp", [12, 26, 34, 44])
def test_partitioned_streams_decode_to_the_encoder_reconstruction(oracle, w, h, n, qp):
    """Every picture of a stream coded with partitions on goes through the independent decoder and must equal the encoder's reconstruction:
    mb_type / sub_mb_type, the vector differences against the directional and median predictors of every partition (neighbours inside the
    macroblock included), motion compensation per partition and the boundary strengths on the inner 8x8 edges all have to agree."""
    out, shapes, _ = _encode(oracle, w, h, n, qp, oracle.F_ALL | oracle.F_PART)
    if qp <= 34 and w >= 176:
        assert shapes[1] > 0 and shapes[2] > 0, shapes          # both two-partition shapes occur
    if qp <= 12:
        assert shapes[3] > 0, shapes                            # ... and P_8x8 (its nine header bits seldom pay above QP 20 with this search)


def test_partitions_are_off_by_default_and_cost_nothing_then(oracle):
    """ORC_F_PART is not part of ORC_F_ALL: the default stream has 16x16 partitions only, and it is the same stream as before the
    partition code existed (the golden digests of tests/test_published_kat.py pin it); with the flag -- and the search the device can
    afford: partitions choose among the vectors the macroblock's refinement visits -- the stream is rate-distortion neutral on the moving
    clip (within 1 % of the bytes at the same PSNR; the dearer search with whole-sample offsets of its own per partition measured 1-3 % fewer)."""
    w, h, n, qp = 320, 192, 6, 28
    base, shapes0, e0 = _encode(oracle, w, h, n, qp, oracle.F_ALL)
    part, shapes1, e1 = _encode(oracle, w, h, n, qp, oracle.F_ALL | oracle.F_PART)
    assert shapes0[1:].sum() == 0 and shapes1[1:].sum() > 0
    b0, b1 = sum(len(a) for a in base), sum(len(a) for a in part)
    assert abs(b1 - b0) < 0.01 * b0, (b0, b1)
    y = list(synth.s2_frames(w, h, n))[-1][0]
    assert synth.psnr(y[:h], e1.recon_y[:h, :w]) > synth.psnr(y[:h], e0.recon_y[:h, :w]) - 0.15


def test_partition_vectors_survive_the_drop_ladder_and_intra_macroblocks(oracle):
    """Partitions next to intra macroblocks (their refIdx is -1 in the predictors) and on rate control's ladder (prediction only: the vectors are still
    coded): a clip with a hard cut in the middle of a GOP and pictures coded at QP 51 with drop levels."""
    w, h = 320, 192
    a, b = list(synth.s2_frames(w, h, 4)), list(synth.s3_frames(w, h, 3))
    oracle.set_features(oracle.F_ALL | oracle.F_PART)
    try:
        e, d = oracle.Encoder(w, h, gop=30, scenecut=False), oracle.Decoder()
        intra_in_p = 0
        for i, ((y, uv), (qp, drop)) in enumerate(zip(a + b, [(30, 0), (30, 0), (51, 2), (30, 0), (30, 0), (51, 6), (28, 0)])):
            au, key = e.encode(y, uv, qp, drop=drop)
            dy, duv = d.decode(au)
            assert np.array_equal(dy, e.recon_y) and np.array_equal(duv, e.recon_uv), i
            if not key:
                intra_in_p += int((e.mbinfo["mb_type"] != 1).sum())
        assert intra_in_p > 0
    finally:
        oracle.set_features(oracle.F_ALL)


@pytest.mark.parametrize("w,h,qp", [(176, 144, 24), (322, 182, 30), (640, 368, 20), (1280, 720, 28)])
def test_host_writer_codes_partitioned_macroblocks_like_the_oracle(oracle, w, h, qp):
    """The product's slice writer on the oracle's records of partitioned pictures: mb_type / sub_mb_type, the partitions' vector differences against
    predictors it derives itself (quadrant vectors of the rows above through fill_ctx_row when row ranges are coded on several threads), the luma-DC
    slot of the packed stream carrying the vectors of partitions 1 .. 3 -- dense levels on one thread and the packed stream on several."""
    from ceracoder_amd import enc as E
    oracle.set_features(oracle.F_ALL | oracle.F_PART)
    try:
        oe = oracle.Encoder(w, h, gop=30, threads=8, intra_slices=1)
        seen = 0
        for i, (y, uv) in enumerate(synth.s2_frames(w, h, 5)):
            au, idr = oe.encode(y, uv, qp)
            hdr = oracle.write_headers(w, h, 60) if idr else b""
            assert hdr + E.host_write_slice(oe.mbw, oe.mbh, idr, i, 0, qp, oe.mbinfo, oe.levels) == au, i
            for thr in (1, 3, 8):
                assert hdr + E.host_write_slice_packed(oe.mbw, oe.mbh, idr, i, 0, qp, oe.mbinfo, oe.levels, threads=thr) == au, (i, thr)
            if not idr:
                seen += int(((oe.mbinfo["mb_type"] == 1) & (oe.mbinfo["i16_mode"] != 0)).sum())
        assert seen > 0
    finally:
        oracle.set_features(oracle.F_ALL)


@pytest.mark.parametrize("w,h,qp", [(64, 48, 30), (176, 144, 24), (322, 182, 34), (640, 368, 28), (1280, 720, 26)])
def test_intra8x8_streams_decode_and_the_host_writer_codes_them(oracle, w, h, qp):
    """Intra_8x8 (High profile; orc_set_i8x8): the oracle encoder's I pictures -- filtered-reference prediction, decision, 8x8 transform with the intra
    rounding -- through the independent decoder (its own 8.3.2 implementation, mode prediction across Intra_4x4 / Intra_8x8 / other neighbours), and the
    product's slice writer on the same records (transform_size_8x8_flag, prev_intra8x8_pred_mode), on one thread and on several, sliced and not."""
    from ceracoder_amd import enc as E
    oracle.set_transform8x8(True)
    oracle.set_i8x8(True)
    try:
        for slices in (1, 0):
            oe, dec = oracle.Encoder(w, h, gop=2, threads=8, intra_slices=slices), oracle.Decoder()
            rows = oracle.slice_rows_for(oe.mbh, slices)
            n8 = 0
            try:
                E.host_set_slice_rows(rows)
                for i, (y, uv) in enumerate(synth.s2_frames(w, h, 4)):
                    au, idr = oe.encode(y, uv, qp)
                    dy, duv = dec.decode(au)
                    assert np.array_equal(dy, oe.recon_y) and np.array_equal(duv, oe.recon_uv), i
                    hdr = E.host_write_headers(w, h, 60, transform8x8=True) if idr else b""
                    assert hdr + E.host_write_slice(oe.mbw, oe.mbh, idr, i % 2, i // 2, qp, oe.mbinfo, oe.levels, transform8x8=True) == au, i
                    for thr in (2, 5):
                        assert hdr + E.host_write_slice_packed(oe.mbw, oe.mbh, idr, i % 2, i // 2, qp, oe.mbinfo, oe.levels, threads=thr, transform8x8=True) == au, (i, thr)
                    if idr:
                        m = oe.mbinfo
                        n8 += int(((m["mb_type"] == 2) & (((m["nzmask"] >> 27) & 1) == 1)).sum())
            finally:
                E.host_set_slice_rows(0)
            assert n8 > 0
    finally:
        oracle.set_transform8x8(False)
        oracle.set_i8x8(False)
