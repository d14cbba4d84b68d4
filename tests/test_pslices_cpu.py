"""P pictures as slices, with and without slice-local deblocking (r04: cfg.slices / cfg.slice_deblock; what x264enc's threads do to a picture behind
/root/reference/pipeline/generic/x264_superfast_camlink:5).  Oracle side and host writer, no GPU: the oracle encoder's decisions (vector prediction, P_Skip
inference, intra availability, nC, QP_Y,PRED stop at a slice's first row; 8.7's filterTopMbEdgeFlag with disable_deblocking_filter_idc 2), the independent
decoder -- which takes availability from its own slice bookkeeping and the filter switch from the slice headers it parses -- and the product's slice writer
on one thread and on several."""
import numpy as np
import pytest

from ceracoder_amd import enc as E
from ceracoder_amd import synth


def _nal_types(au):
    out, i = [], 0
    while True:
        i = au.find(b"\x00\x00\x01", i)
        if i < 0:
            return out
        out.append(au[i + 3] & 31)
        i += 3


@pytest.mark.parametrize("w,h,slices,local,aq", [(64, 48, 2, 0, False), (176, 144, 2, 1, False), (320, 192, 3, 1, True), (322, 182, 4, 0, False),
                                                 (640, 368, 4, 1, False), (640, 368, 5, 1, True), (1280, 720, 3, 1, False), (1920, 1080, 5, 1, False)])
def test_p_slices_decode_and_the_host_writer_codes_them(oracle, w, h, slices, local, aq):
    """Every picture of a stream with sliced P pictures: independent decoder == encoder reconstruction, product writer (dense, packed on 1..n threads) ==
    oracle bytes, one NAL unit per slice with the picture's type, the slice heights a multiple of four rows with slice-local deblocking."""
    oe = oracle.Encoder(w, h, gop=4, threads=8, intra_slices=slices, p_slices=slices, slice_deblock_local=bool(local), aq=aq)
    rows = oracle.slice_rows_for(oe.mbh, slices, bool(local))
    want = (oe.mbh + rows - 1) // rows if rows else 1
    if local and rows:
        assert rows % 4 == 0
    dec = oracle.Decoder()
    try:
        E.host_set_slice_rows(rows)
        E.host_set_p_slices(rows, 2 if local else 0)
        for i, (y, uv) in enumerate(synth.s2_frames(w, h, 5 if w < 1280 else 3)):
            qp = 30 if i != 2 else 22
            au, idr = oe.encode(y, uv, qp)
            dy, duv = dec.decode(au)
            assert np.array_equal(dy, oe.recon_y) and np.array_equal(duv, oe.recon_uv), i
            hdr = oracle.write_headers(w, h, 60) if idr else b""
            assert hdr + E.host_write_slice(oe.mbw, oe.mbh, idr, i % 4, i // 4, qp, oe.mbinfo, oe.levels) == au, i
            for thr in (2, 3, 8):
                assert hdr + E.host_write_slice_packed(oe.mbw, oe.mbh, idr, i % 4, i // 4, qp, oe.mbinfo, oe.levels, threads=thr) == au, (i, thr)
            assert _nal_types(au).count(5 if idr else 1) == want, (i, _nal_types(au))
    finally:
        E.host_set_slice_rows(0)
        E.host_set_p_slices(0, 0)


def test_slice_local_deblocking_leaves_the_seam_rows_alone_and_nothing_else(oracle):
    """The stage function: with disable_deblocking_filter_idc 2 the top edge of a slice's first macroblock row is not filtered (the three lines above it and
    the three below keep their values unless an inner edge of their own macroblock reaches them), every other sample is what idc 0 gives as long as no
    filter's support reaches across the seam -- checked on an intra picture's records, where every edge has boundary strength 3 or 4."""
    w, h, qp = 320, 192, 34
    y, uv = next(iter(synth.s2_frames(w, h, 1)))
    oe = oracle.Encoder(w, h, gop=2, threads=4, intra_slices=1)
    oe.encode(y, uv, qp)
    pre_y, pre_uv, mbi = oe.prefilter_y.copy(), oe.prefilter_uv.copy(), oe.mbinfo.copy()
    full_y, full_uv = oracle.deblock_frame(pre_y, pre_uv, mbi)
    rows = 4
    try:
        oracle.set_slice_rows(rows)
        oracle.set_slice_deblock(2)
        loc_y, loc_uv = oracle.deblock_frame(pre_y, pre_uv, mbi)
    finally:
        oracle.set_slice_rows(0)
        oracle.set_slice_deblock(0)
    assert not np.array_equal(full_y, loc_y)
    for seam in range(rows * 16, h, rows * 16):
        # the seam itself differs from the filtered picture ...
        assert not np.array_equal(full_y[seam - 1:seam + 1], loc_y[seam - 1:seam + 1])
    far = np.ones(h, bool)
    for seam in range(rows * 16, h, rows * 16):
        far[seam - 8:seam + 8] = False  # a filter changes up to 3 samples and reads 4 on either side; vertical edges then carry a change sideways, never vertically
    assert np.array_equal(full_y[:h][far], loc_y[:h][far])


@pytest.mark.parametrize("feat_part", [False, True])
def test_p_slices_with_partitions_and_the_drop_ladder(oracle, feat_part):
    """Partitioned macroblocks take their directional / median predictors from neighbours of the same slice only, on the ladder below QP 51 as well; an
    all-skip picture stays ONE slice whatever the configuration."""
    w, h, slices = 320, 192, 3
    oracle.set_features(oracle.F_ALL | (oracle.F_PART if feat_part else 0))
    try:
        oe, dec = oracle.Encoder(w, h, gop=30, threads=4, p_slices=slices, slice_deblock_local=True), oracle.Decoder()
        rows = oracle.slice_rows_for(oe.mbh, slices, True)
        E.host_set_p_slices(rows, 2)
        for i, (y, uv) in enumerate(synth.s2_frames(w, h, 7)):
            qp, drop = (51, 4) if i == 3 else (51, E.DROP_SKIP) if i == 5 else (26, 0)
            au, idr = oe.encode(y, uv, qp, drop=drop)
            dy, duv = dec.decode(au)
            assert np.array_equal(dy, oe.recon_y) and np.array_equal(duv, oe.recon_uv), i
            if i == 5:
                assert _nal_types(au) == [1] and len(au) < 40
                E.host_set_p_slices(0, 2)
            if not idr:
                oracle.set_part_levels(oe.levels if feat_part else None)
                assert E.host_write_slice(oe.mbw, oe.mbh, False, i, 0, qp, oe.mbinfo, oe.levels) == au, i
                assert E.host_write_slice_packed(oe.mbw, oe.mbh, False, i, 0, qp, oe.mbinfo, oe.levels, threads=3) == au, i
            if i == 5:
                E.host_set_p_slices(rows, 2)
    finally:
        oracle.set_features(oracle.F_ALL)
        oracle.set_part_levels(None)
        E.host_set_p_slices(0, 0)
