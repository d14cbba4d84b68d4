"""GPU parity: every HIP kernel and the whole encoder against the CPU oracle, bit-exact.
All calls go through the C ABI (include/mi355enc.h) via ceracoder_amd.enc."""
import numpy as np
import pytest

from tests.util import first_diff, frames, mbinfo_equal

pytestmark = pytest.mark.gpu

SIZES = [(64, 48), (176, 144), (320, 180), (1280, 720)]


def _surf_equal(dev, orc, rng=16):
    """device (n, 35, 36) against oracle (n, 1089) inside the search range"""
    o = orc.reshape(-1, 33, 33)
    lo, hi = 16 - rng, 16 + rng + 1
    return np.array_equal(dev[:, lo:hi, lo:hi], o[:, lo:hi, lo:hi])


IMV_FIELDS = ("mvx", "mvy", "sad", "bits")


@pytest.mark.parametrize("w,h", SIZES + [(16, 16), (1920, 1088)])
@pytest.mark.parametrize("qp", [20, 34, 48])
def test_me_kernel_matches_oracle(E, oracle, w, h, qp):
    """Whole-sample search: every SAD of the +-16 window (vectors may leave the picture: clamped reference), the first
    selection, and the Jacobi iterations of the selection (me_select_kernel) one by one."""
    f = frames(w, h, 2)
    cur, ref = f[1][0], f[0][0]
    e = E.Encoder(cur.shape[1], cur.shape[0], fixed_qp=qp)
    d_surf, d_imv = e.stage_me(cur, ref, qp)
    o_surf, o_imv = oracle.me_frame(cur, ref, 16, qp, threads=8)
    assert _surf_equal(d_surf, o_surf)
    for fld in IMV_FIELDS:
        assert np.array_equal(d_imv[fld], o_imv[fld]), (fld, first_diff(d_imv[fld], o_imv[fld]))
    mbw, mbh = cur.shape[1] // 16, cur.shape[0] // 16
    for it in range(3):
        o_imv = oracle.me_select(o_surf, o_imv, mbw, mbh, 16, qp, threads=8)
        d_imv = e.stage_me_select(d_surf, d_imv, qp)
        for fld in IMV_FIELDS:
            assert np.array_equal(d_imv[fld], o_imv[fld]), (it, fld, first_diff(d_imv[fld], o_imv[fld]))
    e.close()


def test_me_kernel_ties_and_flat(E, oracle):
    """Flat and periodic content: many equal SADs -> the (cost, dy, dx) tie-break decides."""
    H, W = 96, 160
    flat = np.full((H, W), 77, np.uint8)
    x = np.arange(W)[None, :] + np.zeros((H, 1), int)
    per = ((x % 8) * 20 + 30).astype(np.uint8)
    e = E.Encoder(W, H, fixed_qp=26)
    for cur, ref in ((flat, flat), (per, per), (per, np.roll(per, 3, axis=1)), (flat, per)):
        (d_surf, dev), (o_surf, orc) = e.stage_me(cur, ref, 26), oracle.me_frame(cur, ref, 16, 26)
        assert _surf_equal(d_surf, o_surf) and mbinfo_equal(dev, orc, IMV_FIELDS)
        dev2, orc2 = e.stage_me_select(d_surf, dev, 26), oracle.me_select(o_surf, orc, W // 16, H // 16, 16, 26)
        assert mbinfo_equal(dev2, orc2, IMV_FIELDS)
    e.close()


def _settled_field(oracle, cy, ry, qp, iters=3):
    surf, imv = oracle.me_frame(cy, ry, 16, qp, threads=8)
    for _ in range(iters):
        imv = oracle.me_select(surf, imv, cy.shape[1] // 16, cy.shape[0] // 16, 16, qp, threads=8)
    return surf, imv


@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("qp", [18, 36])
def test_subpel_kernel_matches_oracle(E, oracle, w, h, qp):
    """Two-kernel form (High-profile path): half/quarter-sample refinement with absolute-vector costs against
    orc_subpel_frame, incl. picture borders and vectors that leave the picture."""
    f = frames(w, h, 2)
    cur, ref = f[1][0], f[0][0]
    mbi = oracle.imv_to_mbinfo(_settled_field(oracle, cur, ref, qp)[1], qp)
    orc = oracle.subpel_frame(cur, ref, mbi, qp, threads=8)
    e = E.Encoder(cur.shape[1], cur.shape[0], fixed_qp=qp)
    dev = e.stage_subpel(cur, ref, mbi, qp)
    for fld in ("mvx", "mvy", "cost"):
        assert np.array_equal(dev[fld], orc[fld]), (fld, first_diff(dev[fld], orc[fld]))
    assert (orc["mvx"] % 4 != 0).any() or (orc["mvy"] % 4 != 0).any()  # the refinement really moved something
    e.close()


@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("qp", [0, 18, 30, 51])
@pytest.mark.parametrize("sub", [False, True])
def test_inter_kernel_matches_oracle(E, oracle, w, h, qp, sub):
    f = frames(w, h, 2)
    (cy, cuv), (ry, ruv) = f[1][:2], f[0][:2]
    mbi = oracle.imv_to_mbinfo(_settled_field(oracle, cy, ry, qp)[1], qp)
    if sub:
        mbi = oracle.subpel_frame(cy, ry, mbi, qp, threads=8)
    o_y, o_uv, o_mbi, o_lev = oracle.inter_frame(cy, cuv, ry, ruv, mbi, qp)
    e = E.Encoder(cy.shape[1], cy.shape[0], fixed_qp=qp)
    d_y, d_uv, d_mbi, d_lev = e.stage_inter(cy, cuv, ry, ruv, mbi, qp)
    assert np.array_equal(d_lev, o_lev), first_diff(d_lev, o_lev)
    assert np.array_equal(d_y, o_y), first_diff(d_y, o_y)
    assert np.array_equal(d_uv, o_uv), first_diff(d_uv, o_uv)
    assert mbinfo_equal(d_mbi, o_mbi, ("mvx", "mvy", "mb_type", "qp", "nzmask"))
    e.close()


PMB_FIELDS = ("mvx", "mvy", "mb_type", "i16_mode", "chroma_mode", "qp", "nzmask", "cost")


@pytest.mark.parametrize("w,h", SIZES + [(16, 16), (50, 34), (1920, 1088)])
@pytest.mark.parametrize("qp,drop", [(0, 0), (18, 0), (30, 0), (40, 0), (51, 0), (51, 3), (51, 8), (44, 12)])
@pytest.mark.parametrize("sub,intra", [(True, True), (False, False), (True, False)])
def test_fused_p_kernel_matches_oracle(E, oracle, w, h, qp, drop, sub, intra):
    """pmb_kernel (+ intra_p_kernel): predictor estimates from the settled whole-sample field, skip probe, SAD / SATD
    refinement, intra-or-inter, residual with coefficient decimation, rate control's drop ladder -- against orc_pmb_frame
    (+ orc_intra_p_frame) from the same field, surfaces and intra decisions."""
    if (w, h) == (1920, 1088) and (qp, drop) not in ((30, 0), (51, 8)):
        pytest.skip("full size: two operating points")
    f = frames(w, h, 2)
    (cy, cuv), (ry, ruv) = f[1][:2], f[0][:2]
    surf, imv = _settled_field(oracle, cy, ry, qp)
    idec = None
    if intra:
        idec = oracle.intra_decide(oracle.intra_analyse(cy, cuv), cy.shape[1] // 16, cy.shape[0] // 16, qp, False)  # P pictures: Intra_16x16 only
    o_y, o_uv, o_mbi, o_lev, _ = oracle.pmb_frame(cy, cuv, ry, ruv, imv, surf, qp, drop=drop, refine=sub, idec=idec, threads=8)
    e = E.Encoder(cy.shape[1], cy.shape[0], fixed_qp=qp)
    d_y, d_uv, d_mbi, d_lev = e.stage_pmb(cy, cuv, ry, ruv, imv, oracle.surf_to_device(surf), qp, drop=drop, refine=sub, idec=idec)
    for fld in PMB_FIELDS:
        assert np.array_equal(d_mbi[fld], o_mbi[fld]), (fld, first_diff(d_mbi[fld], o_mbi[fld]))
    assert np.array_equal(d_lev, o_lev), first_diff(d_lev, o_lev)
    assert np.array_equal(d_y, o_y), first_diff(d_y, o_y)
    assert np.array_equal(d_uv, o_uv), first_diff(d_uv, o_uv)
    e.close()


def test_fused_p_kernel_exercises_every_branch(oracle):
    """The clip used above really contains what the kernel branches on (checked on the oracle, which the kernel equals):
    skipped macroblocks, refined quarter-sample vectors, decimated blocks, intra macroblocks in the P picture."""
    f = frames(1280, 720, 2)
    (cy, cuv), (ry, ruv) = f[1][:2], f[0][:2]
    surf, imv = _settled_field(oracle, cy, ry, 30)
    idec = oracle.intra_decide(oracle.intra_analyse(cy, cuv), 80, 45, 30, False)
    _, _, mbi, lev, (pre, _, _) = oracle.pmb_frame(cy, cuv, ry, ruv, imv, surf, 30, idec=idec, threads=8)
    inter = mbi["mb_type"] == 1
    assert (mbi["mb_type"] != 1).sum() > 5                                  # intra macroblocks in a P picture
    assert (inter & (mbi["nzmask"] == 0)).sum() > 20                        # nothing left after the probe / decimation
    assert ((mbi["mvx"] % 4 != 0) | (mbi["mvy"] % 4 != 0)).sum() > 20       # quarter-sample vectors
    assert (inter & (mbi["nzmask"] != 0)).sum() > 100


@pytest.mark.parametrize("w,h", SIZES + [(16, 16), (1920, 1088)])
def test_intra_analyse_kernel_matches_oracle(E, oracle, w, h):
    """Open-loop intra analysis (one flat launch): SAD of every I16 / chroma / I4x4 candidate, 152 u16 per macroblock,
    and the mode decisions taken from them (24 bytes per macroblock)."""
    cy, cuv = frames(w, h, 1)[0][:2]
    for qp, i4 in ((30, True), (12, True), (44, False)):
        e = E.Encoder(cy.shape[1], cy.shape[0], fixed_qp=qp, i4x4=i4)
        (dev, ddec), orc = e.stage_intra_analyse(cy, cuv, qp), oracle.intra_analyse(cy, cuv)
        assert np.array_equal(dev[:, :4], orc[:, :4]), ("i16", first_diff(dev[:, :4], orc[:, :4]))
        assert np.array_equal(dev[:, 4:8], orc[:, 4:8]), ("chroma", first_diff(dev[:, 4:8], orc[:, 4:8]))
        assert np.array_equal(dev[:, 8:], orc[:, 8:]), ("i4", first_diff(dev[:, 8:], orc[:, 8:]))
        odec = oracle.intra_decide(orc, cy.shape[1] // 16, cy.shape[0] // 16, qp, i4)
        for f in ("mode16", "cmode", "use_i4", "cost", "cost_luma", "modes4"):
            assert np.array_equal(ddec[f], odec[f]), (f, qp, i4, first_diff(ddec[f], odec[f]))
        e.close()


@pytest.mark.parametrize("w,h", [(176, 144), (320, 180), (1280, 720), (48, 272)])
@pytest.mark.parametrize("qp,drop", [(51, 1), (51, 2), (51, 4), (51, 12), (38, 7), (44, 3)])
@pytest.mark.parametrize("imode", [0, 1, 2])
def test_intra_kernel_drop_ladder_matches_oracle(E, oracle, w, h, qp, drop, imode):
    """Rate control's ladder for I pictures (below what QP 51 reaches): Intra_16x16 only, and a macroblock's luma / chroma
    levels are not sent when their magnitudes sum to no more than the level's threshold."""
    cy, cuv = frames(w, h, 1)[0][:2]
    o_y, o_uv, o_mbi, o_lev = oracle.intra_frame(cy, cuv, qp, drop)
    assert not (o_mbi["mb_type"] == 2).any()
    e = E.Encoder(cy.shape[1], cy.shape[0], fixed_qp=qp, intra_mode=imode)
    d_y, d_uv, d_mbi, d_lev = e.stage_intra(cy, cuv, qp, drop)
    assert mbinfo_equal(d_mbi, o_mbi, ("mb_type", "i16_mode", "chroma_mode", "cost", "qp", "nzmask")), first_diff(d_mbi["nzmask"], o_mbi["nzmask"])
    assert np.array_equal(d_lev, o_lev), first_diff(d_lev, o_lev)
    assert np.array_equal(d_y, o_y), first_diff(d_y, o_y)
    assert np.array_equal(d_uv, o_uv), first_diff(d_uv, o_uv)
    if drop == 12:
        assert not o_lev[:, :280].any() and not o_mbi["nzmask"].any()   # the last level: prediction only
    e.close()


@pytest.mark.parametrize("w,h", SIZES + [(48, 272), (80, 528)])
@pytest.mark.parametrize("qp", [0, 12, 28, 40, 51])
@pytest.mark.parametrize("i4", [True, False])
@pytest.mark.parametrize("imode", [0, 1, 2])
def test_intra_kernel_matches_oracle(E, oracle, w, h, qp, i4, imode):
    """imode 0: one workgroup per macroblock row, macroblocks overlapping at 4x4-block granularity (dataflow through LDS inside the row,
    tagged granules between rows); imode 1: one launch per anti-diagonal from a hipGraph; imode 2: the lock-step band kernel."""
    cy, cuv = frames(w, h, 1)[0][:2]
    oracle.set_i4x4(i4)
    try:
        o_y, o_uv, o_mbi, o_lev = oracle.intra_frame(cy, cuv, qp)
    finally:
        oracle.set_i4x4(True)
    if i4:
        assert (o_mbi["mb_type"] == 2).any() or qp >= 40
    e = E.Encoder(cy.shape[1], cy.shape[0], fixed_qp=qp, i4x4=i4, intra_mode=imode)
    d_y, d_uv, d_mbi, d_lev = e.stage_intra(cy, cuv, qp)
    assert mbinfo_equal(d_mbi, o_mbi, ("mb_type", "i16_mode", "chroma_mode", "cost")), \
        [(f, first_diff(d_mbi[f], o_mbi[f])) for f in ("mb_type", "i16_mode", "chroma_mode", "cost")]
    assert np.array_equal(d_lev, o_lev), first_diff(d_lev, o_lev)
    assert np.array_equal(d_y, o_y), first_diff(d_y, o_y)
    assert np.array_equal(d_uv, o_uv), first_diff(d_uv, o_uv)
    assert mbinfo_equal(d_mbi, o_mbi, ("mb_type", "qp", "nzmask", "mvx", "mvy"))
    e.close()


@pytest.mark.parametrize("w,h", SIZES + [(48, 272), (640, 368)])
@pytest.mark.parametrize("qp", [10, 24, 34, 46])
@pytest.mark.parametrize("rows", [0, 3])
def test_intra8x8_kernel_matches_oracle(E, oracle, w, h, qp, rows):
    """High profile: I pictures may hold Intra_8x8 macroblocks (filtered reference samples, nine modes, the 8x8 transform with the intra rounding; the rows
    kernel takes them whole, with the first eight samples of the macroblock above-right) -- decisions, records, levels and reconstruction equal the oracle's
    (orc_intra_decide8 / intra8x8_recon), with and without slices; not above picture QP 37; with i8x8 off (the default), or with an intra schedule that cannot have the macroblock above-right
    ready (intra_mode 1, 2), the picture is the one without them."""
    cy, cuv = frames(w, h, 1)[0][:2]
    oracle.set_transform8x8(True)
    oracle.set_slice_rows(rows)
    oracle.set_i8x8(True)
    try:
        o_y, o_uv, o_mbi, o_lev = oracle.intra_frame(cy, cuv, qp)
        oracle.set_i8x8(False)
        n_y, n_uv, n_mbi, n_lev = oracle.intra_frame(cy, cuv, qp)
    finally:
        oracle.set_transform8x8(False)
        oracle.set_i8x8(False)
        oracle.set_slice_rows(0)
    is8 = (o_mbi["mb_type"] == 2) & ((o_mbi["nzmask"] >> 27) & 1).astype(bool)
    assert is8.any() == (qp <= 37) or (qp == 10 and w == 64)   # (not above picture QP 37: ORC_I8_QP_MAX)
    for kw, (r_y, r_uv, r_mbi, r_lev) in (({"i8x8": True}, (o_y, o_uv, o_mbi, o_lev)), ({}, (n_y, n_uv, n_mbi, n_lev)), ({"i8x8": True, "intra_mode": 1}, (n_y, n_uv, n_mbi, n_lev)),
                                          ({"i8x8": True, "intra_mode": 2}, (n_y, n_uv, n_mbi, n_lev))):
        e = E.Encoder(cy.shape[1], cy.shape[0], fixed_qp=qp, transform8x8=True, **kw)
        e.stage_set_slice_rows(rows)
        d_y, d_uv, d_mbi, d_lev = e.stage_intra(cy, cuv, qp)
        assert mbinfo_equal(d_mbi, r_mbi, ("mb_type", "i16_mode", "chroma_mode", "cost")), \
            (kw, [(f, first_diff(d_mbi[f], r_mbi[f])) for f in ("mb_type", "i16_mode", "chroma_mode", "cost")])
        assert np.array_equal(d_lev, r_lev), (kw, first_diff(d_lev, r_lev))
        assert np.array_equal(d_y, r_y), (kw, first_diff(d_y, r_y))
        assert np.array_equal(d_uv, r_uv), (kw, first_diff(d_uv, r_uv))
        assert mbinfo_equal(d_mbi, r_mbi, ("mb_type", "qp", "nzmask", "mvx", "mvy")), (kw, first_diff(d_mbi["nzmask"], r_mbi["nzmask"]))
        e.close()


@pytest.mark.parametrize("w,h,rows", [(176, 144, 3), (320, 180, 4), (640, 368, 6), (48, 272, 5), (1280, 720, 12), (1920, 1080, 17)])
@pytest.mark.parametrize("qp", [12, 30, 44])
@pytest.mark.parametrize("imode", [0, 1, 2])
def test_intra_kernel_with_slices_matches_oracle(E, oracle, w, h, rows, qp, imode):
    """An I picture cut into slices of `rows` macroblock rows: the row above a slice's first row is not available to the analysis, to the
    Intra_4x4 mode decision or to the reconstruction (6.4.8) -- records, levels and reconstruction equal the oracle's (orc_set_slice_rows)."""
    cy, cuv = frames(w, h, 1)[0][:2]
    oracle.set_slice_rows(rows)
    try:
        o_y, o_uv, o_mbi, o_lev = oracle.intra_frame(cy, cuv, qp)
        oracle.set_slice_rows(0)
        p_y = oracle.intra_frame(cy, cuv, qp)[0]
    finally:
        oracle.set_slice_rows(0)
    assert not np.array_equal(p_y, o_y)  # the slices do change the picture
    e = E.Encoder(cy.shape[1], cy.shape[0], fixed_qp=qp, intra_mode=imode)
    e.stage_set_slice_rows(rows)
    d_y, d_uv, d_mbi, d_lev = e.stage_intra(cy, cuv, qp)
    assert mbinfo_equal(d_mbi, o_mbi, ("mb_type", "i16_mode", "chroma_mode", "cost")), \
        [(f, first_diff(d_mbi[f], o_mbi[f])) for f in ("mb_type", "i16_mode", "chroma_mode", "cost")]
    assert np.array_equal(d_lev, o_lev), first_diff(d_lev, o_lev)
    assert np.array_equal(d_y, o_y), first_diff(d_y, o_y)
    assert np.array_equal(d_uv, o_uv), first_diff(d_uv, o_uv)
    assert mbinfo_equal(d_mbi, o_mbi, ("mb_type", "qp", "nzmask", "mvx", "mvy"))
    e.close()


@pytest.mark.parametrize("w,h,n,slices,depth,aq", [(64, 48, 6, 3, 0, False), (176, 144, 7, 2, 0, True), (322, 182, 6, 4, 1, False), (640, 368, 7, 0, 1, True), (1280, 720, 5, 0, 2, False),
                                                 (1280, 720, 5, 5, 0, True), (1920, 1080, 6, 0, 2, False), (1920, 1080, 5, 8, 2, True), (1920, 1080, 4, 1, 0, False)])
@pytest.mark.parametrize("imode", [0, 1])
def test_sliced_intra_pictures_equal_oracle(E, oracle, w, h, n, slices, depth, aq, imode):
    """cfg.intra_slices (0: the default, about 17 macroblock rows per slice): every IDR picture is `slices` NAL units of type 5, the access units
    equal the oracle's, and the independent decoder -- which takes the slice structure from first_mb_in_slice alone -- reproduces the
    reconstruction; with adaptive quantisation the QP_Y chain starts again with every slice."""
    qps = [30, 27, 34, 24, 40, 30, 51]
    e = E.Encoder(w, h, gop=3, fixed_qp=30, pipeline_depth=depth, exclusive=True, aq=aq, intra_slices=slices, intra_mode=imode, cavlc_threads=3)
    oe = oracle.Encoder(w, h, gop=3, threads=8, aq=aq, intra_slices=slices)
    dec = oracle.Decoder()
    clip = [(y, uv) for _, _, y, uv in frames(w, h, n)]
    got = []
    for i, (y, uv) in enumerate(clip):
        e.set_fixed_qp(qps[i % len(qps)])
        e.submit(y, uv, pts=i)
        if e.pending > depth:
            got.append(e.collect()[:2])
    while e.pending:
        got.append(e.collect()[:2])
    mbh = (h + 15) // 16
    rows = oracle.slice_rows_for(mbh, slices)
    assert e.slice_rows == rows
    want = (mbh + rows - 1) // rows if rows else 1
    for i, (y, uv) in enumerate(clip):
        ref_au, ref_key = oe.encode(y, uv, qps[i % len(qps)])
        assert got[i][0] == ref_au, ("bitstream", i, len(got[i][0]), len(ref_au))
        if ref_key:
            assert ref_au.count(b"\x00\x00\x01\x65") == want, (i, want)
        dy, duv = dec.decode(ref_au)
        assert np.array_equal(dy, oe.recon_y) and np.array_equal(duv, oe.recon_uv), i
    assert np.array_equal(e.fetch(E.FETCH_RECON_Y), oe.recon_y) and np.array_equal(e.fetch(E.FETCH_RECON_UV), oe.recon_uv)
    e.close()


@pytest.mark.parametrize("w,h", SIZES + [(48, 272), (64, 256), (80, 528), (1920, 1080)])
@pytest.mark.parametrize("qp", [16, 30, 44, 51])
@pytest.mark.parametrize("mode", [0, 1])
def test_deblock_kernel_matches_oracle(E, oracle, w, h, qp, mode):
    """Feed the oracle's own pre-filter pictures (one I, one P) and records to the HIP filter
    (mode 0: prep kernel + persistent 16-row bands in x+y order with one barrier per step;
    mode 1: one launch per x+2y wavefront, the plain form kept as a cross-check)."""
    oe = oracle.Encoder(w, h, gop=60, threads=8)
    e = E.Encoder((w + 15) // 16 * 16, (h + 15) // 16 * 16, fixed_qp=qp, deblock_mode=mode)
    for _, _, y, uv in frames(w, h, 2):
        oe.encode(y, uv, qp)
        d_y, d_uv = e.stage_deblock(oe.prefilter_y, oe.prefilter_uv, oe.mbinfo)
        assert np.array_equal(d_y, oe.recon_y), first_diff(d_y, oe.recon_y)
        assert np.array_equal(d_uv, oe.recon_uv), first_diff(d_uv, oe.recon_uv)
    e.close()


@pytest.mark.parametrize("w,h,n", [(64, 48, 9), (176, 144, 7), (322, 182, 5), (1280, 720, 4), (1920, 1080, 3)])
@pytest.mark.parametrize("graphs,mode,sub,thr,imode", [(True, 0, True, 1, 0), (False, 0, False, 3, 1), (True, 1, True, 1, 1), (True, 0, True, 4, 0), (True, 0, True, 2, 2)])
def test_encoder_bitstream_equals_oracle(E, oracle, w, h, n, graphs, mode, sub, thr, imode):
    """Whole path: identical access units, identical reconstruction, and the independent
    decoder reproduces both."""
    qps = [30, 28, 33, 24, 40, 26, 30, 51, 10]
    e = E.Encoder(w, h, gop=4, fixed_qp=30, use_graphs=graphs, keep_prefilter=True, deblock_mode=mode, subpel=sub, cavlc_threads=thr, intra_mode=imode)
    oe = oracle.Encoder(w, h, gop=4, threads=8, subpel=sub)
    dec = oracle.Decoder()
    for i, (_, _, y, uv) in enumerate(frames(w, h, n)):
        qp = qps[i % len(qps)]
        e.set_fixed_qp(qp)
        au, key = e.encode(y, uv, pts=i)
        ref_au, ref_key = oe.encode(y, uv, qp)
        assert key == ref_key
        assert np.array_equal(e.fetch(E.FETCH_PREFILTER_Y), oe.prefilter_y), ("prefilter", i)
        assert np.array_equal(e.fetch(E.FETCH_RECON_Y), oe.recon_y), ("recon", i, first_diff(e.fetch(E.FETCH_RECON_Y), oe.recon_y))
        assert np.array_equal(e.fetch(E.FETCH_RECON_UV), oe.recon_uv), ("recon uv", i)
        assert au == ref_au, ("bitstream", i, len(au), len(ref_au))
        dy, duv = dec.decode(au)
        assert np.array_equal(dy, oe.recon_y) and np.array_equal(duv, oe.recon_uv)
    assert dec.size == (w, h)
    e.close()


@pytest.mark.parametrize("w,h,n", [(64, 48, 6), (176, 144, 6), (322, 182, 5), (1280, 720, 3)])
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("i8", [False, True])
def test_high_profile_8x8_transform_equals_oracle(E, oracle, w, h, n, mode, i8):
    """transform8x8=1: P macroblocks through the 8x8 transform kernel path, deblocking with 8x8 block edges; i8: Intra_8x8 macroblocks in the I pictures
    (their inner 4-sample edges are not deblocked either)."""
    oracle.set_transform8x8(True)
    oracle.set_i8x8(i8)
    try:
        e = E.Encoder(w, h, gop=4, fixed_qp=30, transform8x8=True, keep_prefilter=True, deblock_mode=mode, i8x8=i8)
        oe = oracle.Encoder(w, h, gop=4, threads=8)
        dec = oracle.Decoder()
        for i, (_, _, y, uv) in enumerate(frames(w, h, n)):
            qp = [30, 12, 40, 24, 2, 50][i % 6]
            e.set_fixed_qp(qp)
            au, _ = e.encode(y, uv, pts=i)
            ref_au, _ = oe.encode(y, uv, qp)
            assert np.array_equal(e.fetch(E.FETCH_LEVELS), oe.levels), ("levels", i, first_diff(e.fetch(E.FETCH_LEVELS), oe.levels))
            assert np.array_equal(e.fetch(E.FETCH_PREFILTER_Y), oe.prefilter_y), ("prefilter", i)
            assert np.array_equal(e.fetch(E.FETCH_RECON_Y), oe.recon_y), ("recon", i, first_diff(e.fetch(E.FETCH_RECON_Y), oe.recon_y))
            assert au == ref_au, ("bitstream", i)
            dy, duv = dec.decode(au)
            assert np.array_equal(dy, oe.recon_y) and np.array_equal(duv, oe.recon_uv)
        e.close()
    finally:
        oracle.set_i8x8(False)
        oracle.set_transform8x8(False)


@pytest.mark.parametrize("w,h,depth", [(176, 144, 0), (640, 368, 0), (640, 368, 1), (1920, 1080, 1), (1920, 1080, 2)])
def test_drop_ladder_and_all_skip_pictures_equal_oracle(E, oracle, w, h, depth):
    """What rate control does below QP 51, under test control (fixed QP + fixed drop level): P pictures on every part of the
    ladder, an IDR picture on its own ladder, runs of all-skip pictures (no device work: the reconstruction is the reference)
    and the return to normal coding -- access units, reconstruction and the independent decoder all agree with the oracle."""
    plan = [(40, 0), (51, 0), (51, 2), (51, 7), (51, 12), (51, 255), (51, 255), (51, 5), (51, 3), (46, 0), (51, 255), (51, 9), (30, 0), (51, 12)]
    n = len(plan) if w < 1000 else 9
    e = E.Encoder(w, h, gop=8, fixed_qp=40, pipeline_depth=depth, keep_prefilter=True)
    oe = oracle.Encoder(w, h, gop=8, threads=8)
    dec = oracle.Decoder()
    clip = [(y, uv) for _, _, y, uv in frames(w, h, n)]
    got = []
    for i, (y, uv) in enumerate(clip):
        qp, drop = plan[i]
        e.set_fixed_qp(qp); e.set_fixed_drop(drop)
        e.submit(y, uv, pts=i)
        if e.pending > depth:
            got.append(e.collect() + (e.last_drop,))
    while e.pending:
        got.append(e.collect() + (e.last_drop,))
    for i, (y, uv) in enumerate(clip):
        qp, drop = plan[i]
        ref_au, ref_key = oe.encode(y, uv, qp, drop=drop)
        au, key, pts, gqp, gdrop = got[i]
        assert (key, pts, gqp) == (ref_key, i, qp) and gdrop == (drop if not (key and drop == 255) else 0)
        assert au == ref_au, ("bitstream", i, len(au), len(ref_au))
        if drop == 255 and not key:
            assert len(au) < 40                                        # a slice header and one skip run
        dy, duv = dec.decode(au)
        assert np.array_equal(dy, oe.recon_y) and np.array_equal(duv, oe.recon_uv)
    assert np.array_equal(e.fetch(E.FETCH_RECON_Y), oe.recon_y) and np.array_equal(e.fetch(E.FETCH_RECON_UV), oe.recon_uv)
    e.close()


@pytest.mark.parametrize("w,h,n,depth", [(176, 144, 6, 0), (640, 368, 6, 0), (1280, 720, 5, 1), (1920, 1080, 5, 2)])
def test_intra4x4_macroblocks_in_p_pictures_equal_oracle(E, oracle, w, h, n, depth):
    """intra_in_p = 2: the intra macroblocks of P pictures may be Intra_4x4 (gated analysis with the nine-mode search, intra_p_row walking ten
    sub-steps, the tile stored write-through for the deblocker that follows it).  The clip uncovers texture behind moving rectangles every
    picture and cuts to another scene half way, so both kinds occur; access units, reconstruction and the independent decoder agree."""
    from tests.util import cut_clip
    clip = cut_clip(w, h, n, n // 2 + 1)
    oracle.set_features(63)
    try:
        e = E.Encoder(w, h, gop=30, fixed_qp=26, intra_in_p=2, pipeline_depth=depth, exclusive=True, scenecut=False)
        oe = oracle.Encoder(w, h, gop=30, threads=8, scenecut=False)
        dec = oracle.Decoder()
        got, n_i4 = [], 0
        for i, (y, uv) in enumerate(clip):
            e.submit(y, uv, pts=i)
            if e.pending > depth:
                got.append(e.collect()[0])
        while e.pending:
            got.append(e.collect()[0])
        for i, (y, uv) in enumerate(clip):
            ref_au, _ = oe.encode(y, uv, 26)
            assert got[i] == ref_au, ("bitstream", i, len(got[i]), len(ref_au))
            dy, duv = dec.decode(ref_au)
            assert np.array_equal(dy, oe.recon_y) and np.array_equal(duv, oe.recon_uv)
            if i:
                n_i4 += int((oe.mbinfo["mb_type"] == 2).sum())
        assert np.array_equal(e.fetch(E.FETCH_RECON_Y), oe.recon_y) and np.array_equal(e.fetch(E.FETCH_RECON_UV), oe.recon_uv)
        assert n_i4 > 0, "the clip produced no Intra_4x4 macroblock in a P picture"
        e.close()
    finally:
        oracle.set_features(31)


@pytest.mark.parametrize("w,h,n,depth", [(64, 48, 7, 0), (322, 182, 6, 0), (640, 368, 7, 1), (1920, 1080, 5, 2), (1920, 1080, 4, 0)])
def test_adaptive_quantisation_equals_oracle(E, oracle, w, h, n, depth):
    """aq_mode 1: a QP offset per macroblock from the variance of its source samples (aq_kernel == orc_aq_offsets), quantisation with the
    macroblock's own QP in the I and P stages, mb_qp_delta in the slice, and the QP_Y of macroblocks without mb_qp_delta taken from the
    macroblock before them for the deblocker (qp_chain_kernel == orc_qp_chain, 7.4.5).  Access units, reconstruction and the independent
    decoder -- which derives QP_Y from the bitstream alone -- all agree."""
    from tests.util import half_static_clip
    clip = half_static_clip(w, h, n, (h // 3) & ~15) if w >= 322 else [(y, uv) for _, _, y, uv in frames(w, h, n)]
    qps = [30, 26, 34, 22, 40, 30, 51]
    e = E.Encoder(w, h, gop=4, fixed_qp=30, aq=True, pipeline_depth=depth, exclusive=True)
    oe = oracle.Encoder(w, h, gop=4, threads=8, aq=True)
    dec = oracle.Decoder()
    got, seen = [], set()
    for i, (y, uv) in enumerate(clip):
        e.set_fixed_qp(qps[i % len(qps)])
        e.submit(y, uv, pts=i)
        if e.pending > depth:
            got.append(e.collect()[0])
    while e.pending:
        got.append(e.collect()[0])
    for i, (y, uv) in enumerate(clip):
        ref_au, _ = oe.encode(y, uv, qps[i % len(qps)])
        assert got[i] == ref_au, ("bitstream", i, len(got[i]), len(ref_au))
        dy, duv = dec.decode(ref_au)
        assert np.array_equal(dy, oe.recon_y) and np.array_equal(duv, oe.recon_uv), i
        seen |= set(int(q) - qps[i % len(qps)] for q in np.unique(oe.mbinfo["qp"]))
    assert np.array_equal(e.fetch(E.FETCH_RECON_Y), oe.recon_y) and np.array_equal(e.fetch(E.FETCH_RECON_UV), oe.recon_uv)
    assert len(seen) >= 2, seen  # more than one offset occurred
    e.close()


@pytest.mark.parametrize("w,h", [(16, 16), (32, 16), (16, 48), (18, 18), (4096, 32)])
def test_degenerate_geometries(E, oracle, w, h):
    """Single macroblock, single row/column, non-multiple-of-16, and the widest row the caps allow."""
    e = E.Encoder(w, h, gop=3, fixed_qp=27, exclusive=True)
    oe = oracle.Encoder(w, h, gop=3, threads=4)
    dec = oracle.Decoder()
    for i, (_, _, y, uv) in enumerate(frames(w, h, 5)):
        au, _ = e.encode(y, uv, pts=i)
        assert au == oe.encode(y, uv, 27)[0], i
        dy, duv = dec.decode(au)
        assert np.array_equal(dy, e.fetch(E.FETCH_RECON_Y)) and np.array_equal(duv, e.fetch(E.FETCH_RECON_UV))
    assert dec.size == (w, h)
    e.close()


@pytest.mark.parametrize("rng", [1, 5, 8])
def test_me_range_property(E, oracle, rng):
    """me-range < 16: candidates outside the range are masked, the rest of the kernels is unchanged."""
    f = frames(320, 192, 2)
    cur, ref = f[1][0], f[0][0]
    e = E.Encoder(320, 192, fixed_qp=30, me_range=rng)
    (d_surf, dev), (o_surf, orc) = e.stage_me(cur, ref, 30), oracle.me_frame(cur, ref, rng, 30, threads=4)
    assert _surf_equal(d_surf, o_surf, rng) and mbinfo_equal(dev, orc, IMV_FIELDS)
    dev2, orc2 = e.stage_me_select(d_surf, dev, 30), oracle.me_select(o_surf, orc, 20, 12, rng, 30)
    assert mbinfo_equal(dev2, orc2, IMV_FIELDS)
    assert np.abs(dev2["mvx"]).max() <= 4 * rng and np.abs(dev2["mvy"]).max() <= 4 * rng
    e.close()


def test_2160p_one_gop_head_equals_oracle(E, oracle):
    """BASELINE config 4 geometry at full size: IDR + P, bit-exact (32 400 macroblocks, 34 deblock bands)."""
    w, h = 3840, 2160
    e = E.Encoder(w, h, gop=60, fixed_qp=32, exclusive=True)
    oe = oracle.Encoder(w, h, gop=60, threads=16)
    for i, (_, _, y, uv) in enumerate(frames(w, h, 2)):
        au, _ = e.encode(y, uv, pts=i)
        assert au == oe.encode(y, uv, 32)[0], i
        assert np.array_equal(e.fetch(E.FETCH_RECON_Y), oe.recon_y)
    e.close()


def test_full_size_stream_properties_1080p(E, oracle):
    """Size-independent properties at the benchmark geometry over two GOPs with rate control on: every access unit
    decodes with the independent decoder to exactly the encoder's reconstruction (drift-free closed loop), IDR cadence
    and parameter sets are where the element promises them, and PSNR stays sane."""
    from ceracoder_amd import synth
    w, h, gop = 1920, 1080, 12
    e = E.Encoder(w, h, gop=gop, bitrate_bps=12_000_000, fps=60)
    dec = oracle.Decoder()
    for i, (y, uv) in enumerate(synth.s2_frames(w, h, 2 * gop + 1)):
        au, key = e.encode(y, uv, pts=i)
        assert key == (i % gop == 0)
        assert (au[:5] == b"\x00\x00\x00\x01\x67") == key
        dy, duv = dec.decode(au)
        assert np.array_equal(dy, e.fetch(E.FETCH_RECON_Y)) and np.array_equal(duv, e.fetch(E.FETCH_RECON_UV)), i
        assert synth.psnr(y, dy[:h, :w]) > 24.0
    e.close()


@pytest.mark.parametrize("depth", [1, 2])
def test_pipelined_submit_collect_equals_sync(E, oracle, depth):
    """pipeline_depth=1 (entropy coding overlapped with the next picture) and 2 (three pictures in flight) yield the same stream."""
    w, h, n = 320, 192, 11
    fr = frames(w, h, n)
    a = E.Encoder(w, h, gop=5, fixed_qp=29)
    b = E.Encoder(w, h, gop=5, fixed_qp=29, pipeline_depth=depth, exclusive=True)
    sync = [a.encode(y, uv)[0] for _, _, y, uv in fr]
    piped = []
    for i, (_, _, y, uv) in enumerate(fr):
        b.submit(y, uv, pts=i)
        if b.pending == depth + 1:
            piped.append(b.collect())
    while b.pending:
        piped.append(b.collect())
    assert [p[0] for p in piped] == sync
    assert [p[2] for p in piped] == list(range(n))
    a.close(); b.close()


@pytest.mark.parametrize("w,h,n", [(1920, 1080, 90), (3840, 2160, 14)])
def test_three_pictures_in_flight_equal_one_at_a_time(E, w, h, n):
    """Full-size pictures, fixed QP, scene-cut recovery off (its landing picture depends on the depth): with pipeline_depth 2 every P
    picture's fused stage runs beside the deblocking of the picture before it, workgroup by workgroup behind that launch's bands, and
    three pictures share the device (`exclusive`: the encoder has the GPU to itself, which is what allows kernels to wait for each other on it); with depth 0 every picture is alone on it.  Same access units, same reconstruction -- including
    across forced IDR pictures and straight after them."""
    from ceracoder_amd import synth
    clip = list(synth.s2_frames(w, h, 8))
    force = {9, 10, 23, 61}
    streams = []
    for depth in (0, 2):
        e = E.Encoder(w, h, gop=16, fixed_qp=30, pipeline_depth=depth, scenecut=False, exclusive=True)
        got = []
        for i in range(n):
            k = i % 14
            y, uv = clip[k if k < 8 else 14 - k]
            e.submit(y, uv, pts=i, force_idr=i in force)
            if e.pending > depth:
                got.append(e.collect())
        while e.pending:
            got.append(e.collect())
        streams.append(([g[0] for g in got], [g[1] for g in got], e.fetch(E.FETCH_RECON_Y), e.fetch(E.FETCH_RECON_UV)))
        e.close()
    a, b = streams
    assert a[1] == b[1] and sum(a[1]) >= 3
    for i, (x, y) in enumerate(zip(a[0], b[0])):
        assert x == y, ("access unit", i, len(x), len(y))
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])


@pytest.mark.parametrize("depth", [0, 1, 2])
def test_scene_cut_recovery_equals_oracle(E, oracle, depth):
    """cfg.scenecut: after a hard cut (picture 5) the picture two positions later is coded as IDR, identically to the oracle
    and independently of the pipeline depth up to 1 (the decision uses the cost sum that arrives with the hand-over of picture 5,
    and lands on the first picture that cannot have been submitted yet: with three pictures in flight that is one picture later)."""
    from tests.util import cut_clip
    w, h, n = 320, 192, 12
    clip = cut_clip(w, h, n, 5)
    e = E.Encoder(w, h, gop=30, fixed_qp=30, pipeline_depth=depth, exclusive=True)
    oe = oracle.Encoder(w, h, gop=30, threads=8, sc_lag=max(2, depth + 1))
    got = []
    for i, (y, uv) in enumerate(clip):
        e.submit(y, uv, pts=i)
        if e.pending > depth:
            got.append(e.collect())
    while e.pending:
        got.append(e.collect())
    keys = []
    for i, (y, uv) in enumerate(clip):
        ref_au, ref_key = oe.encode(y, uv, 30)
        assert got[i][0] == ref_au, ("bitstream", i, len(got[i][0]), len(ref_au))
        assert got[i][1] == ref_key
        if ref_key:
            keys.append(i)
    assert keys == [0, 5 + max(2, depth + 1)], keys
    assert np.array_equal(e.fetch(E.FETCH_RECON_Y), oe.recon_y)
    e.close()


@pytest.mark.parametrize("w,h,static_lines,depth", [(640, 528, 288, 0), (640, 528, 288, 1), (1280, 720, 512, 0), (1280, 720, 512, 2), (320, 1040, 800, 0)])
def test_idle_deblocking_bands_equal_oracle(E, oracle, w, h, static_lines, depth):
    """Still background over a moving scene: the upper 16-row deblocking bands of the P pictures have no edge with bS != 0,
    so their workgroups publish "done" and leave (deblock_prep_kernel's per-band flags); the bands below read the strips
    above them straight from the picture.  Streams and reconstructions must still equal the oracle's, picture by picture."""
    from tests.util import half_static_clip
    clip = half_static_clip(w, h, 7, static_lines)
    e = E.Encoder(w, h, gop=30, fixed_qp=38, pipeline_depth=depth, exclusive=True)
    oe = oracle.Encoder(w, h, gop=30, threads=8)
    got = []
    for i, (y, uv) in enumerate(clip):
        e.submit(y, uv, pts=i)
        if e.pending > depth:
            got.append(e.collect())
    while e.pending:
        got.append(e.collect())
    for i, (y, uv) in enumerate(clip):
        ref_au, _ = oe.encode(y, uv, 38)
        assert got[i][0] == ref_au, ("bitstream", i)
        if i == len(clip) - 1: # the top band really is idle by now: no coded block, no vector in its 16 macroblock rows
            mbi = oe.mbinfo.reshape(oe.mbh, oe.mbw)[:16]
            assert not mbi["nzmask"].any() and not mbi["mvx"].any() and not mbi["mvy"].any()
    assert np.array_equal(e.fetch(E.FETCH_RECON_Y), oe.recon_y), first_diff(e.fetch(E.FETCH_RECON_Y), oe.recon_y)
    assert np.array_equal(e.fetch(E.FETCH_RECON_UV), oe.recon_uv)
    e.close()


def test_noise_worst_case_roundtrip(E, oracle):
    """S3 (i.i.d. noise) at low QP: maximum-size levels, escape codes, every block coded."""
    w, h = 176, 144
    e = E.Encoder(w, h, gop=3, fixed_qp=4)
    oe = oracle.Encoder(w, h, gop=3, threads=8)
    dec = oracle.Decoder()
    for i, (_, _, y, uv) in enumerate(frames(w, h, 4, kind="s3")):
        qp = [4, 0, 8, 2][i]
        e.set_fixed_qp(qp)
        au, _ = e.encode(y, uv)
        assert au == oe.encode(y, uv, qp)[0]
        dy, duv = dec.decode(au)
        assert np.array_equal(dy, e.fetch(E.FETCH_RECON_Y)) and np.array_equal(duv, e.fetch(E.FETCH_RECON_UV))
    e.close()


@pytest.mark.parametrize("depth", [0, 1, 2])
def test_rate_control_emergency_drop_on_the_device(E, depth):
    """N3: a 4x cut of the setpoint between two key frames shows in the access-unit sizes within a few pictures
    (pipeline_depth 1 adds one picture of feedback delay)."""
    w, h, fps, gop = 640, 368, 30, 60
    e = E.Encoder(w, h, fps=fps, gop=gop, bitrate_bps=2_400_000, pipeline_depth=depth, exclusive=True)
    sizes, drop_at = [], 75
    fr = frames(w, h, 120)
    for i, (_, _, y, uv) in enumerate(fr):
        if i == drop_at:
            e.set_bitrate(600_000)
        e.submit(y, uv, pts=i)
        if e.pending > depth:
            sizes.append(len(e.collect()[0]))
    while e.pending:
        sizes.append(len(e.collect()[0]))
    per_frame = 600_000 / fps / 8
    assert np.mean(sizes[drop_at - 10:drop_at]) > 2.5 * per_frame          # it really was running at the old rate
    assert max(sizes[drop_at + 3 + depth:drop_at + 8]) < 1.6 * per_frame, sizes[drop_at:drop_at + 8]   # the cut lands within a few pictures
    assert max(sizes[drop_at + 3 + depth:drop_at + 25]) < 2.5 * per_frame and np.mean(sizes[drop_at + 3 + depth:drop_at + 25]) < 1.3 * per_frame
    assert sum(sizes[drop_at + 2:drop_at + 32]) * 8 * fps / 30 < 1.3 * 600_000
    e.close()


def test_gated_p_stage_gives_the_in_order_stream_under_rate_control(E):
    """1080p CBR with three pictures in flight: with `exclusive` every P picture's fused stage runs beside the previous picture's deblocking
    launch and waits for the bands it reads; without, it follows that launch in stream order.  Rate control sees the same sizes in the
    same order either way, so the two streams are the same bytes -- any macroblock that read a reference line too early shows up as a
    different size somewhere and a different stream from there on."""
    import hashlib
    from ceracoder_amd import synth
    w, h, n = 1920, 1080, 240
    clip = list(synth.s2_frames(w, h, 16))
    digests = []
    for exclusive in (False, True, True):
        e = E.Encoder(w, h, fps=60, gop=60, bitrate_bps=6_000_000, pipeline_depth=2, exclusive=exclusive)
        out = []
        for i in range(n):
            k = i % 30
            y, uv = clip[k if k < 16 else 30 - k]
            e.submit(y, uv, pts=i)
            if e.pending > 2:
                out.append(bytes(e.collect()[0]))
        while e.pending:
            out.append(bytes(e.collect()[0]))
        e.close()
        digests.append([hashlib.sha256(x).hexdigest()[:12] for x in out])
    for k in (1, 2):
        diff = next((i for i, (a, b) in enumerate(zip(digests[0], digests[k])) if a != b), None)
        assert diff is None, ("first differing access unit", diff, "run", k)


def test_stage_timers_on_the_free_running_schedule_change_nothing(E):
    """profile_overlap: sampled P pictures keep the overlapped schedule (event pairs on the streams the kernels are launched on; the gated P stage and the
    deblocking launch are timed with their device-side waits).  The stream is the same bytes as without any timer and as with in-order sampling, and
    every sampled picture delivered its timers."""
    import hashlib
    from ceracoder_amd import synth
    w, h, n = 1920, 1080, 150
    clip = list(synth.s2_frames(w, h, 16))
    digests, stats = [], []
    for events, overlap in ((0, False), (5, True), (5, False)):
        e = E.Encoder(w, h, fps=60, gop=60, bitrate_bps=6_000_000, pipeline_depth=2, exclusive=True, profile_events=events, profile_overlap=overlap)
        out = []
        for i in range(n):
            k = i % 30
            y, uv = clip[k if k < 16 else 30 - k]
            e.submit(y, uv, pts=i)
            if e.pending > 2:
                out.append(bytes(e.collect()[0]))
        while e.pending:
            out.append(bytes(e.collect()[0]))
        st = e.stats()
        stats.append((int(st.n_me), int(st.n_deblock), st.ms_me, st.ms_deblock, int(st.recoveries)))
        e.close()
        digests.append([hashlib.sha256(x).hexdigest()[:12] for x in out])
    assert digests[1] == digests[0] and digests[2] == digests[0]
    assert stats[0][:2] == (0, 0) and all(s[4] == 0 for s in stats)
    for n_me, n_db, ms_me, ms_db, _ in stats[1:]:
        assert n_me >= 25 and n_db >= n_me and 0.02 < ms_me / n_me < 0.5 and 0.05 < ms_db / n_db < 1.5, stats


@pytest.mark.parametrize("w,h", SIZES + [(16, 16), (50, 34), (1920, 1088)])
@pytest.mark.parametrize("qp,drop", [(8, 0), (20, 0), (30, 0), (40, 0), (51, 3)])
@pytest.mark.parametrize("intra", [True, False])
def test_fused_p_kernel_with_partitions_matches_oracle(E, oracle, w, h, qp, drop, intra):
    """cfg.partitions: every partition of 16x8 / 8x16 / 8x8 picks among the vectors the macroblock's refinement visits (quadrant SADs left behind by the
    same reductions that give the macroblock's), the shape with the lowest total decides, the prediction is taken with a vector per lane; shape in the
    record's i16_mode, the vectors of partitions 1 .. 3 in the luma-DC slot of the levels -- against orc_pmb_frame with ORC_F_PART."""
    if (w, h) == (1920, 1088) and qp != 20:
        pytest.skip("full size: one operating point")
    f = frames(w, h, 2)
    (cy, cuv), (ry, ruv) = f[1][:2], f[0][:2]
    surf, imv = _settled_field(oracle, cy, ry, qp)
    idec = oracle.intra_decide(oracle.intra_analyse(cy, cuv), cy.shape[1] // 16, cy.shape[0] // 16, qp, False) if intra else None
    oracle.set_features(oracle.F_ALL | oracle.F_PART)
    try:
        o_y, o_uv, o_mbi, o_lev, _ = oracle.pmb_frame(cy, cuv, ry, ruv, imv, surf, qp, drop=drop, refine=True, idec=idec, threads=8)
    finally:
        oracle.set_features(oracle.F_ALL)
    e = E.Encoder(cy.shape[1], cy.shape[0], fixed_qp=qp, partitions=True)
    d_y, d_uv, d_mbi, d_lev = e.stage_pmb(cy, cuv, ry, ruv, imv, oracle.surf_to_device(surf), qp, drop=drop, refine=True, idec=idec)
    for fld in PMB_FIELDS + ("i16_mode",):
        assert np.array_equal(d_mbi[fld], o_mbi[fld]), (fld, first_diff(d_mbi[fld], o_mbi[fld]))
    assert np.array_equal(d_lev, o_lev), first_diff(d_lev, o_lev)
    assert np.array_equal(d_y, o_y), first_diff(d_y, o_y)
    assert np.array_equal(d_uv, o_uv), first_diff(d_uv, o_uv)
    if qp <= 30 and cy.shape[1] >= 176:
        inter = o_mbi["mb_type"] == 1
        assert (inter & (o_mbi["i16_mode"] != 0)).any()
    e.close()


@pytest.mark.parametrize("w,h,n,depth,aq", [(64, 48, 7, 0, False), (176, 144, 7, 0, False), (322, 182, 6, 1, False), (640, 368, 6, 2, True), (1280, 720, 5, 2, False), (1920, 1080, 5, 2, False),
                                          (1920, 1080, 4, 0, True)])
def test_partitioned_streams_equal_oracle(E, oracle, w, h, n, depth, aq):
    """Whole path with cfg.partitions: the device's shapes and vectors, the deblocker's boundary strengths per 8x8 quadrant (vectors read from the levels'
    luma-DC slot, behind the row counts when the stages overlap), the hand-over of that slot and the host writer's partition syntax -- access units and
    reconstruction equal the oracle's, and the independent decoder reproduces them."""
    qps = [24, 20, 30, 26, 34, 22, 28]
    oracle.set_features(oracle.F_ALL | oracle.F_PART)
    try:
        e = E.Encoder(w, h, gop=30, fixed_qp=30, pipeline_depth=depth, exclusive=True, partitions=True, aq=aq, cavlc_threads=3)
        oe = oracle.Encoder(w, h, gop=30, threads=8, aq=aq)
        dec = oracle.Decoder()
        clip = [(y, uv) for _, _, y, uv in frames(w, h, n)]
        got = []
        for i, (y, uv) in enumerate(clip):
            e.set_fixed_qp(qps[i % len(qps)])
            e.submit(y, uv, pts=i)
            if e.pending > depth:
                got.append(e.collect()[0])
        while e.pending:
            got.append(e.collect()[0])
        split = 0
        for i, (y, uv) in enumerate(clip):
            ref_au, key = oe.encode(y, uv, qps[i % len(qps)])
            assert got[i] == ref_au, ("bitstream", i, len(got[i]), len(ref_au))
            dy, duv = dec.decode(ref_au)
            assert np.array_equal(dy, oe.recon_y) and np.array_equal(duv, oe.recon_uv), i
            if not key:
                split += int(((oe.mbinfo["mb_type"] == 1) & (oe.mbinfo["i16_mode"] != 0)).sum())
        assert np.array_equal(e.fetch(E.FETCH_RECON_Y), oe.recon_y) and np.array_equal(e.fetch(E.FETCH_RECON_UV), oe.recon_uv)
        assert split > 0 or w < 176
        e.close()
    finally:
        oracle.set_features(oracle.F_ALL)


@pytest.mark.parametrize("exclusive,single_stream,depth", [(False, False, 0), (False, False, 2), (True, False, 2), (False, True, 2), (True, False, 1)])
@pytest.mark.parametrize("partitions,aq,slices", [(False, False, 0), (True, False, 3), (False, True, 2), (True, True, 0)])
def test_schedules_and_options_give_the_oracle_stream(E, oracle, exclusive, single_stream, depth, partitions, aq, slices):
    """The same stream whatever the schedule: kernels waiting for each other on the device (exclusive), stream order (default), one stream per
    encoder (single_stream), one to three pictures in flight -- crossed with partitions, adaptive quantisation and sliced I pictures; 720p, so
    that the default slice count is above one."""
    w, h, n = 1280, 720, 5
    qps = [26, 30, 24, 34, 28]
    oracle.set_features(oracle.F_ALL | (oracle.F_PART if partitions else 0))
    try:
        e = E.Encoder(w, h, gop=3, fixed_qp=30, pipeline_depth=depth, exclusive=exclusive, single_stream=single_stream, partitions=partitions, aq=aq, intra_slices=slices)
        oe = oracle.Encoder(w, h, gop=3, threads=8, aq=aq, intra_slices=slices)
        clip = [(y, uv) for _, _, y, uv in frames(w, h, n)]
        got = []
        for i, (y, uv) in enumerate(clip):
            e.set_fixed_qp(qps[i])
            e.submit(y, uv, pts=i)
            if e.pending > depth:
                got.append(e.collect()[0])
        while e.pending:
            got.append(e.collect()[0])
        for i, (y, uv) in enumerate(clip):
            assert got[i] == oe.encode(y, uv, qps[i])[0], ("bitstream", i)
        assert np.array_equal(e.fetch(E.FETCH_RECON_Y), oe.recon_y)
        e.close()
    finally:
        oracle.set_features(oracle.F_ALL)


@pytest.mark.parametrize("w,h,n,depth,aq,i8", [(322, 182, 7, 0, False, False), (640, 368, 8, 1, True, True), (1280, 720, 6, 2, False, True), (1920, 1080, 6, 2, True, False),
                                                 (1920, 1080, 4, 2, False, True)])
def test_high_profile_stream_through_the_fused_stage_equals_oracle(E, oracle, w, h, n, depth, aq, i8):
    """transform8x8=1 since r03: the fused P stage (skip probe, refinement against the predictor estimates, intra macroblocks, the drop ladder) with the 8x8
    transform for the luma residual of the inter macroblocks (pmb_luma_t8), overlapped like the Baseline stream; a clip with a cut, so that P pictures
    carry intra macroblocks; with and without adaptive quantisation.  Access units, reconstruction and the independent decoder agree; 8x8-transform,
    skipped and intra macroblocks all occur."""
    oracle.set_transform8x8(True)
    oracle.set_i8x8(i8)
    try:
        from tests.util import half_static_clip
        clip = half_static_clip(w, h, n, (h // 3) & ~15)
        qps = [28, 24, 32, 51, 26, 30, 22, 36]
        e = E.Encoder(w, h, gop=30, fixed_qp=30, transform8x8=True, pipeline_depth=depth, exclusive=True, scenecut=False, aq=aq, i8x8=i8)
        oe = oracle.Encoder(w, h, gop=30, threads=8, scenecut=False, aq=aq)
        dec = oracle.Decoder()
        got = []
        for i, (y, uv) in enumerate(clip):
            e.set_fixed_qp(qps[i % 8])
            e.submit(y, uv, pts=i)
            if e.pending > depth:
                got.append(e.collect()[0])
        while e.pending:
            got.append(e.collect()[0])
        t8 = skip = intra = 0
        for i, (y, uv) in enumerate(clip):
            ref_au, key = oe.encode(y, uv, qps[i % 8])
            assert got[i] == ref_au, ("bitstream", i, len(got[i]), len(ref_au))
            dy, duv = dec.decode(ref_au)
            assert np.array_equal(dy, oe.recon_y) and np.array_equal(duv, oe.recon_uv), i
            if not key:
                m = oe.mbinfo
                t8 += int(((m["nzmask"] >> 27) & 1).sum())
                intra += int((m["mb_type"] != 1).sum())
                skip += int(((m["mb_type"] == 1) & (m["nzmask"] == 0)).sum())
        assert np.array_equal(e.fetch(E.FETCH_RECON_Y), oe.recon_y) and np.array_equal(e.fetch(E.FETCH_RECON_UV), oe.recon_uv)
        assert t8 > 0 and skip > 0, (t8, skip, intra)
        e.close()
    finally:
        oracle.set_i8x8(False)
        oracle.set_transform8x8(False)
