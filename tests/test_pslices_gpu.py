"""P pictures as slices with slice-local deblocking on the device (r04; cfg.slices / cfg.slice_deblock): the kernels whose neighbourhood stops at a
slice's first row -- vector selection and the fused P stage (field_pred), the intra macroblock rows of P pictures, the band deblocker (a slice is a
whole number of bands; a slice's first row has no top edge and takes no strips, its last row keeps its bottom lines) and the per-diagonal form --
against the oracle's stage functions, then whole streams through every schedule."""
import numpy as np
import pytest

from tests.util import first_diff, frames, mbinfo_equal

pytestmark = pytest.mark.gpu

IMV_FIELDS = ("mvx", "mvy", "sad", "bits")


@pytest.mark.parametrize("w,h,rows", [(176, 144, 4), (320, 192, 4), (640, 368, 8), (1280, 720, 12), (1920, 1088, 16), (1920, 1088, 20)])
def test_vector_selection_stops_at_slice_tops(E, oracle, w, h, rows):
    """me_select_kernel with ctx->slice_rows: the 8.4.1.3 median and the P_Skip inference take nothing from the row above a slice's first row."""
    e = E.Encoder(w, h, fixed_qp=30)
    fr = frames(w, h, 2)
    cur, ref = fr[1][0], fr[0][0]
    o_surf, orc = oracle.me_frame(cur, ref, 16, 30, threads=8)
    d_surf, dev = e.stage_me(cur, ref, 30)
    assert mbinfo_equal(dev, orc, IMV_FIELDS)
    oracle.set_slice_rows(rows)
    try:
        e.stage_set_slice_rows(rows)
        mbw, mbh = w // 16, h // 16
        for it in range(3):
            orc2 = oracle.me_select(o_surf, orc, mbw, mbh, 16, 30, threads=8)
            dev2 = e.stage_me_select(d_surf, dev, 30)
            assert mbinfo_equal(dev2, orc2, IMV_FIELDS), (it, rows)
            orc, dev = orc2, dev2
    finally:
        oracle.set_slice_rows(0)
    e.close()


@pytest.mark.parametrize("w,h,rows", [(64, 48, 0), (176, 144, 0), (330, 182, 0), (320, 192, 4), (1280, 720, 12), (1920, 1088, 0), (1920, 1088, 20), (3840, 2160, 0)])
def test_later_selection_passes_copy_what_did_not_change(E, oracle, w, h, rows):
    """me_select_sparse_kernel (r04: the encoder's second and third selection pass -- a wave checks eight macroblocks' predictors and only walks the surfaces of those
    that changed): the same field as the dense pass and as the oracle, three passes in a row, with and without slices, down to pictures narrower than a wave's eight."""
    W, H = (w + 15) // 16 * 16, (h + 15) // 16 * 16
    e = E.Encoder(W, H, fixed_qp=30)
    fr = frames(w, h, 2)
    cur, ref = fr[1][0], fr[0][0]
    o_surf, orc = oracle.me_frame(cur, ref, 16, 30, threads=8)
    d_surf, dev = e.stage_me(cur, ref, 30)
    assert mbinfo_equal(dev, orc, IMV_FIELDS)
    oracle.set_slice_rows(rows)
    try:
        e.stage_set_slice_rows(rows)
        prev = dev
        cur_f = e.stage_me_select(d_surf, dev, 30)  # first pass: dense
        copied = 0
        for it in range(3):
            dense = e.stage_me_select(d_surf, cur_f, 30)
            sparse = e.stage_me_select_next(d_surf, cur_f, prev, 30)
            want = oracle.me_select(o_surf, cur_f, W // 16, H // 16, 16, 30, threads=8)
            assert mbinfo_equal(sparse, dense, IMV_FIELDS) and mbinfo_equal(sparse, want, IMV_FIELDS), (it, rows)
            copied += int((sparse == cur_f).sum())
            prev, cur_f = cur_f, sparse
        assert copied > 0
    finally:
        oracle.set_slice_rows(0)
    e.close()


@pytest.mark.parametrize("w,h,rows", [(64, 64, 4), (176, 144, 4), (320, 192, 8), (1280, 720, 12), (1920, 1088, 16), (1920, 1088, 20), (1920, 1088, 4)])
@pytest.mark.parametrize("qp", [20, 34, 51])
@pytest.mark.parametrize("mode", [0, 1])
def test_slice_local_deblocking_matches_oracle(E, oracle, w, h, rows, qp, mode):
    """stage_deblock with disable_deblocking_filter_idc 2 on the oracle's own pre-filter pictures (one I: every edge strong; one P) and records: the band
    kernel (slices are whole bands: 4-row slices make every band a slice) and the per-diagonal form."""
    if rows >= (h + 15) // 16:
        pytest.skip("one slice")
    oe = oracle.Encoder(w, h, gop=60, threads=8)
    e = E.Encoder((w + 15) // 16 * 16, (h + 15) // 16 * 16, fixed_qp=qp, deblock_mode=mode)
    e.stage_set_slice_rows(rows)
    e.stage_set_slice_deblock(2)
    try:
        for _, _, y, uv in frames(w, h, 2):
            oe.encode(y, uv, qp)
            oracle.set_slice_rows(rows)
            oracle.set_slice_deblock(2)
            want_y, want_uv = oracle.deblock_frame(oe.prefilter_y, oe.prefilter_uv, oe.mbinfo)
            oracle.set_slice_rows(0)
            oracle.set_slice_deblock(0)
            d_y, d_uv = e.stage_deblock(oe.prefilter_y, oe.prefilter_uv, oe.mbinfo)
            assert np.array_equal(d_y, want_y), first_diff(d_y, want_y)
            assert np.array_equal(d_uv, want_uv), first_diff(d_uv, want_uv)
            if qp >= 34:
                assert not np.array_equal(want_y, oe.recon_y)  # the seams are really left alone (at QP 20 the filter does nothing there anyway)
    finally:
        oracle.set_slice_rows(0)
        oracle.set_slice_deblock(0)
        e.close()


def _run_stream(E, oracle, w, h, n, qps, slices, local, depth=0, exclusive=False, single_stream=False, aq=False, partitions=False, intra_in_p=1, t8=False,
                islices=0, clip=None, gop=4, mode=0, thr=0, check_dec=True):
    oracle.set_features(oracle.F_ALL | (oracle.F_PART if partitions else 0) | (oracle.F_I4P if intra_in_p == 2 else 0))
    oracle.set_transform8x8(t8)
    try:
        e = E.Encoder(w, h, gop=gop, fixed_qp=30, pipeline_depth=depth, exclusive=exclusive, single_stream=single_stream, partitions=partitions, aq=aq,
                      intra_slices=islices, slices=slices, slice_deblock=local, intra_in_p=intra_in_p, transform8x8=t8, scenecut=False, deblock_mode=mode, cavlc_threads=thr)
        oe = oracle.Encoder(w, h, gop=gop, threads=16, aq=aq, intra_slices=islices, p_slices=slices, slice_deblock_local=local, scenecut=False)
        dec = oracle.Decoder()
        assert e.p_slice_rows == oracle.slice_rows_for(oe.mbh, slices, local)
        clip = clip or [(y, uv) for _, _, y, uv in frames(w, h, n)]
        got = []
        for i, (y, uv) in enumerate(clip):
            e.set_fixed_qp(qps[i % len(qps)])
            e.submit(y, uv, pts=i)
            if e.pending > depth:
                got.append(e.collect()[0])
        while e.pending:
            got.append(e.collect()[0])
        intra = 0
        for i, (y, uv) in enumerate(clip):
            ref_au, key = oe.encode(y, uv, qps[i % len(qps)])
            assert got[i] == ref_au, ("bitstream", i, len(got[i]), len(ref_au))
            if check_dec:
                dy, duv = dec.decode(ref_au)
                assert np.array_equal(dy, oe.recon_y) and np.array_equal(duv, oe.recon_uv), i
            if not key:
                intra += int((oe.mbinfo["mb_type"] != 1).sum())
        assert np.array_equal(e.fetch(E.FETCH_RECON_Y), oe.recon_y), first_diff(e.fetch(E.FETCH_RECON_Y), oe.recon_y)
        assert np.array_equal(e.fetch(E.FETCH_RECON_UV), oe.recon_uv)
        st = e.stats()
        assert st.recoveries == 0
        e.close()
        return intra
    finally:
        oracle.set_features(oracle.F_ALL)
        oracle.set_transform8x8(False)


@pytest.mark.parametrize("w,h,n,slices", [(176, 144, 7, 2), (322, 182, 6, 3), (640, 368, 6, 4), (1280, 720, 5, 3), (1920, 1080, 5, 4), (1920, 1080, 4, 5), (1920, 1080, 3, 17)])
@pytest.mark.parametrize("local", [False, True])
def test_sliced_p_pictures_equal_oracle(E, oracle, w, h, n, slices, local):
    """Whole path at pipeline depth 0: access units, reconstruction, the independent decoder -- P pictures cut into slices, the filter across the seams
    (idc 0) or stopping at them (idc 2)."""
    _run_stream(E, oracle, w, h, n, [30, 28, 33, 24, 40, 26, 51], slices, local)


@pytest.mark.parametrize("exclusive,single_stream,depth", [(False, False, 2), (True, False, 2), (False, True, 2), (True, False, 1)])
@pytest.mark.parametrize("partitions,aq,ip,t8", [(False, False, 1, False), (True, False, 1, False), (False, True, 1, False), (False, False, 2, False), (False, True, 1, True)])
def test_sliced_streams_through_schedules_and_options(E, oracle, exclusive, single_stream, depth, partitions, aq, ip, t8):
    """The slice-local stream whatever the schedule (kernels waiting for each other on the device, stream order, one stream per encoder; one to three
    pictures in flight) crossed with partitions, adaptive quantisation (the QP_Y chain starts again with every slice), Intra_4x4 in P pictures and the
    High-profile stream; 720p with a clip that has a cut, so that P pictures carry intra macroblocks on both sides of the seams."""
    from tests.util import cut_clip
    w, h, n = 1280, 720, 7
    intra = _run_stream(E, oracle, w, h, n, [26, 30, 24, 34, 28, 22, 38], 3, True, depth=depth, exclusive=exclusive, single_stream=single_stream, aq=aq, partitions=partitions,
                        intra_in_p=ip, t8=t8, clip=cut_clip(w, h, n, 4), gop=30)
    assert intra > 0


@pytest.mark.parametrize("w,h,n,slices", [(1920, 1080, 40, 4), (1920, 1080, 24, 5), (3840, 2160, 8, 8), (1280, 720, 40, 3), (640, 368, 60, 2)])
def test_three_pictures_in_flight_with_slices_equal_oracle(E, oracle, w, h, n, slices):
    """The free-running schedule (three pictures in flight, the fused P stage beside the previous picture's deblocking launch, the launch waiting on the
    device for its picture's rows) with every slice a wavefront of its own: the oracle's stream."""
    _run_stream(E, oracle, w, h, n, [30, 31, 29, 32], slices, True, depth=2, exclusive=True, gop=60, check_dec=False)


def test_per_diagonal_deblocker_and_the_writer_threads_with_slices(E, oracle):
    """deblock_mode 1 (one launch per wavefront step: the ladder's last level) and the row-parallel writer on explicit thread counts with sliced P pictures."""
    for thr in (1, 3, 8):
        _run_stream(E, oracle, 640, 368, 5, [28, 32, 26], 4, True, mode=1, thr=thr)
        _run_stream(E, oracle, 640, 368, 5, [28, 32, 26], 4, True, mode=0, thr=thr, depth=1)


@pytest.mark.parametrize("w,h,n,depth", [(1280, 720, 5, 0), (1920, 1080, 5, 2), (3840, 2160, 3, 2), (640, 368, 5, 1), (16, 16, 4, 2), (48, 32, 4, 0), (176, 144, 5, 2), (352, 288, 5, 2), (720, 576, 5, 1),
                                       (1920, 1200, 4, 2), (2560, 1440, 4, 2), (854, 480, 5, 2), (4096, 2304, 3, 2)])
def test_library_defaults_are_sliced_and_equal_oracle(E, oracle, w, h, n, depth):
    """mi355enc_default_cfg (what the element and bench.py get; this mirror's own defaults keep the one-slice P pictures of rounds 1-3): P pictures cut like I
    pictures -- about 17 macroblock rows per slice, rounded up to whole deblocking bands -- with slice-local deblocking.  Same stream as the oracle told the same."""
    e = E.Encoder(w, h, gop=3, fixed_qp=30, pipeline_depth=depth, exclusive=True, slices=None, slice_deblock=None, scenecut=False)
    mbh = (h + 15) // 16
    ns = oracle.auto_slices(mbh)
    oe = oracle.Encoder(w, h, gop=3, threads=16, intra_slices=0, p_slices=ns, slice_deblock_local=True, scenecut=False)
    assert e.p_slice_rows == e.slice_rows == oracle.slice_rows_for(mbh, ns, True)
    clip = [(y, uv) for _, _, y, uv in frames(w, h, n)]
    got = []
    for i, (y, uv) in enumerate(clip):
        e.submit(y, uv, pts=i)
        if e.pending > depth:
            got.append(e.collect()[0])
    while e.pending:
        got.append(e.collect()[0])
    for i, (y, uv) in enumerate(clip):
        assert got[i] == oe.encode(y, uv, 30)[0], i
    assert np.array_equal(e.fetch(E.FETCH_RECON_Y), oe.recon_y)
    e.close()


@pytest.mark.parametrize("reserve", ["40", "24,1"])
def test_reserved_compute_units_switch_keeps_the_stream(E, oracle, monkeypatch, reserve):
    """MI355ENC_RESERVE_CUS (development switch, read when an encoder is opened): the front and the intra stream are created with a compute-unit mask.  Placement
    only -- the stream is the oracle's."""
    monkeypatch.setenv("MI355ENC_RESERVE_CUS", reserve)
    _run_stream(E, oracle, 1280, 720, 6, [30, 28, 33], 3, True, depth=2, exclusive=True, gop=30, check_dec=False)


def test_bands_walked_whole_keep_the_stream(E, oracle, monkeypatch):
    """MI355ENC_NO_SPLIT (A/B switch, read when an encoder is opened): every deblocking band as one workgroup per plane, the round-3 form.  Same stream."""
    monkeypatch.setenv("MI355ENC_NO_SPLIT", "1")
    _run_stream(E, oracle, 1920, 1080, 5, [30, 28, 33], 4, True, depth=2, exclusive=True, gop=30, check_dec=False)


def test_bands_are_cut_where_the_filter_does_nothing(E):
    """The default: a P picture's deblocking bands are walked as two workgroups, cut at a column with bS = 0 in every row of the band (k_deblock.hip, "the cut").  After a few
    P pictures of the S2 clip most bands of the last launch carry a cut inside the row, tagged with that picture's epoch, and no cut lies left of the band above's
    in the same slice.  (That the result is the oracle's is what every other test here checks; this one checks that the feature is at work.)"""
    import ctypes as C
    from ceracoder_amd import synth
    w, h = 1920, 1080
    e = E.Encoder(w, h, gop=60, fixed_qp=31, pipeline_depth=0, slices=None, slice_deblock=None, scenecut=False)
    for y, uv in synth.s2_frames(w, h, 4):
        e.submit(y, uv, pts=0)
        e.collect(copy=False)
    nb = (e.mbh + 3) // 4
    buf = np.zeros(6 * nb, np.uint32)
    assert e.L.mi355enc_fetch(e.h, 102, buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
    rows = e.p_slice_rows
    e.close()
    cuts = buf[2 * nb:].reshape(nb, 2, 2)
    assert len(set(cuts[:, :, 1].reshape(-1).tolist())) == 1, "every band of the last launch left its cut"
    assert (buf[:2 * nb] % 2 == 0).all(), "both parts of every band have counted themselves"
    for plane in range(2):
        c = cuts[:, plane, 0].astype(int)
        assert ((c >= e.mbw // 4) & (c < e.mbw)).sum() >= nb // 2, c
        for b in range(1, nb):
            if rows and (b * 4) % rows != 0 and c[b - 1] < e.mbw and c[b] < e.mbw:
                assert c[b] >= c[b - 1], (b, c)
