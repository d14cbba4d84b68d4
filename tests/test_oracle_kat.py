"""Known-answer tests that pin the oracle itself: Exp-Golomb, emulation
prevention, transform identities, and every constant table -- each VLC table is checked to
be prefix-free and compared between the two independent transcriptions (encoder: bit
strings as the standard prints them; decoder: numeric length/value arrays)."""
import ctypes as C

import numpy as np
import pytest


def _vlc(L, fn, table, a, b, c):
    ln, bits = C.c_int(0), C.c_int(0)
    ok = getattr(L, fn)(table, a, b, c, C.byref(ln), C.byref(bits))
    return (ln.value, bits.value) if ok else None


TABLES = [  # (table id, a-range, b-range, c-range)
    (0, 4, 17, 4), (1, 1, 5, 4), (2, 1, 15, 16), (3, 1, 3, 4), (4, 1, 7, 15)]


def test_exp_golomb_kat(oracle):
    L = oracle.lib()
    want = ["1", "010", "011", "00100", "00101", "00110", "00111", "0001000", "0001001"]
    for v, w in enumerate(want):
        code = C.c_uint32(0)
        n = L.orc_ue_bits(v, C.byref(code))
        assert format(code.value, "0%db" % n) == w
    assert L.orc_ue_bits(2 ** 16 - 2, None) == 31


def test_emulation_prevention_kat(oracle):
    L = oracle.lib()

    def esc(b):
        src = np.frombuffer(bytes(b), np.uint8)
        out = np.zeros(64, np.uint8)
        n = L.orc_nal_escape(src.ctypes.data, src.size, out.ctypes.data, out.size)
        return bytes(out[:n])
    for x in range(4):
        assert esc([0, 0, x]) == bytes([0, 0, 3, x])
    assert esc([0, 0, 4]) == bytes([0, 0, 4])
    assert esc([0, 0, 0, 0, 0, 1]) == bytes([0, 0, 3, 0, 0, 3, 0, 1])
    assert esc([1, 0, 0]) == bytes([1, 0, 0])


def test_transform_dc_identity_and_roundtrip(oracle):
    L = oracle.lib()
    i16 = C.c_int16 * 16
    for c in (-255, -3, 0, 1, 17, 255):
        out = i16()
        L.orc_fdct4(i16(*([c] * 16)), out)
        assert out[0] == 16 * c and not any(out[1:]), "flat residual c must give W00 = 16c only"
    rng = np.random.default_rng(1)
    for qp in (0, 10, 26, 40, 51):
        for _ in range(40):
            res = rng.integers(-255, 256, 16).astype(np.int16)
            co = i16()
            L.orc_fdct4(i16(*res), co)
            d = (C.c_int32 * 16)(*[L.orc_dequant4(L.orc_quant4(co[p], qp, p, 0), qp, p) for p in range(16)])
            pix = (C.c_uint8 * 16)(*([128] * 16))
            L.orc_idct4_add(d, pix, 4)
            rec = np.array(pix[:], np.int32) - 128
            err = np.abs(np.clip(res, -128, 127) - rec).max()
            qstep = 0.625 * 2 ** (qp / 6)
            assert err <= 2.5 * qstep + 2, (qp, err)


def test_vlc_tables_prefix_free_and_cross_checked(oracle):
    L = oracle.lib()
    for tid, na, nb, nc in TABLES:
        for a in range(na):
            for b in range(nb):
                codes = []
                for c in range(nc):
                    e, d = _vlc(L, "orc_enc_vlc", tid, a, b, c), _vlc(L, "orc_dec_vlc", tid, a, b, c)
                    assert e == d, ("transcriptions differ", tid, a, b, c, e, d)
                    if e and tid in (2, 3, 4):
                        codes.append(format(e[1], "0%db" % e[0]))
                for i, x in enumerate(codes):  # one context: total_zeros / run_before row
                    for j, y in enumerate(codes):
                        assert i == j or not y.startswith(x), (tid, b, x, y)
    # coeff_token: prefix-free across all (TotalCoeff, TrailingOnes) of one nC class
    for tid, na in ((0, 4), (1, 1)):
        for a in range(na):
            codes = []
            for b in range(17 if tid == 0 else 5):
                for c in range(4):
                    e = _vlc(L, "orc_enc_vlc", tid, a, b, c)
                    valid = c <= b
                    assert (e is not None) == valid, (tid, a, b, c)
                    if e:
                        codes.append(format(e[1], "0%db" % e[0]))
            assert len(set(codes)) == len(codes)
            for i, x in enumerate(codes):
                for j, y in enumerate(codes):
                    assert i == j or not y.startswith(x), (tid, a, x, y)
            if tid == 0 and a < 3:  # Kraft sum of a complete-enough code never exceeds 1
                assert sum(2.0 ** -len(c) for c in codes) <= 1.0


def test_cbp_mapping_tables_are_inverse_permutations(oracle):
    L = oracle.lib()
    for intra in (0, 1):
        fwd = [L.orc_enc_cbp_codenum(intra, cbp) for cbp in range(48)]
        assert sorted(fwd) == list(range(48))
        assert all(L.orc_dec_cbp(intra, fwd[cbp]) == cbp for cbp in range(48))
    assert L.orc_enc_cbp_codenum(1, 47) == 0 and L.orc_enc_cbp_codenum(0, 0) == 0  # Table 9-4 first row


def test_deblock_and_scaling_constants(oracle):
    L = oracle.lib()
    alpha = [L.orc_dec_const(0, i) for i in range(52)]
    beta = [L.orc_dec_const(1, i) for i in range(52)]
    assert alpha[:16] == [0] * 16 and alpha[16] == 4 and alpha[51] == 255 and alpha == sorted(alpha)
    assert beta[:16] == [0] * 16 and beta[16] == 2 and beta[51] == 18 and beta == sorted(beta)
    for col in (2, 3, 4):
        t = [L.orc_dec_const(col, i) for i in range(52)]
        assert t == sorted(t) and t[51] == {2: 13, 3: 17, 4: 25}[col]
    # Table 8-15 chroma QP
    qpc = [L.orc_dec_const(5, i) for i in range(52)]
    assert qpc[:30] == list(range(30)) and qpc[30:] == [29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39]
    # zig-zag is a permutation starting 0,1,4,8,5
    zz = [L.orc_dec_const(6, i) for i in range(16)]
    assert sorted(zz) == list(range(16)) and zz[:5] == [0, 1, 4, 8, 5]
    # normAdjust4x4 x encoder multiplier ~ 2^21 / {16, 25, 20}: the two tables agree with each other
    for m in range(6):
        for cls, g, pos in ((7, 16, 0), (8, 25, 5), (9, 20, 1)):
            v = L.orc_dec_const(cls, m)
            mf_level = 8 * L.orc_quant4(1 << 12, m, pos, 0)  # (2^12 * MF + f) >> 15 ~= MF / 8 (levels clamp at 2047)
            assert abs(mf_level * v * g - 2 ** 21) < 2 ** 21 * 0.006, (m, cls, mf_level, v)


def test_table_checksums_are_pinned(oracle):
    """Any edit to a constant table must be deliberate: the checksums are golden."""
    L = oracle.lib()
    got = [L.orc_table_checksum(i) for i in range(11)]
    want = pytest.importorskip("tests.golden.table_checksums").CHECKSUMS
    assert got == want


def test_me_lambda_monotone(oracle):
    L = oracle.lib()
    lam = [L.orc_me_lambda(q) for q in range(52)]
    assert lam == sorted(lam) and lam[0] == 1 and lam[51] == 91


def test_8x8_transform_tables_are_mutually_consistent(oracle):
    """High-profile 8x8 path: the forward/inverse pair, normAdjust8x8 and the quantiser multipliers were
    transcribed separately; they only fit together if all are right: MF * V * g(class) == 2^24 where g is the
    squared basis norm measured through the oracle's own fdct8(idct8(.)), and the zig-zag table (encoder:
    transcribed; decoder: generated by walking anti-diagonals) must agree."""
    L = oracle.lib()
    I64 = C.c_int * 64
    g = np.zeros((8, 8))
    for p in range(64):
        d = I64(); d[p] = 1 << 16
        r, c = I64(), I64()
        L.orc_idct8(d, r); L.orc_fdct8(r, c)
        g[p // 8, p % 8] = c[p] / float(1 << 16)
    assert np.allclose(sorted(set(np.round(g.ravel(), 2))), [25.0, 40.0, 45.16, 64.0, 72.25, 81.56], atol=0.011)
    for qp in range(6):
        for p in range(64):
            v = L.orc_dequant8(1, 36 + qp, p) // 16          # LevelScale8x8 / 16 at qP/6 == 6 (no shift)
            mf = 16 * L.orc_quant8(1 << 12, qp, p, 0)         # (2^12 * MF + f) >> 16 ~ MF / 16 (levels clamp at 2047)
            assert abs(mf * v * g[p // 8, p % 8] - 2 ** 24) < 2 ** 24 * 0.012, (qp, p, mf, v)
    assert [L.orc_zigzag8(k) for k in range(64)] == [L.orc_dec_zz8(k) for k in range(64)]
    assert sorted(L.orc_zigzag8(k) for k in range(64)) == list(range(64))
    rng = np.random.default_rng(5)
    for _ in range(20):  # near-identity at qp 0
        x = rng.integers(-255, 256, 64)
        c = I64(); L.orc_fdct8(I64(*x), c)
        dq = I64(*[L.orc_dequant8(L.orc_quant8(c[p], 0, p, 0), 0, p) for p in range(64)])
        r = I64(); L.orc_idct8(dq, r)
        assert np.abs((np.array(r[:]) + 32 >> 6) - x).max() <= 2
