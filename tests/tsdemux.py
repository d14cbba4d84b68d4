"""Independent MPEG-2 TS reader used by the tests (written from ISO/IEC 13818-1, shares no code
with ceracoder_amd/csrc/tsmux.c): splits packets, checks sync bytes and continuity counters,
validates PSI CRCs and returns the PES packets of the video PID with their PTS and PCR."""


def crc32_mpeg2(data):
    c = 0xFFFFFFFF
    for b in data:
        c ^= b << 24
        for _ in range(8):
            c = ((c << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if c & 0x80000000 else (c << 1) & 0xFFFFFFFF
    return c


def _ts33(b):
    return ((b[0] >> 1) & 7) << 30 | b[1] << 22 | (b[2] >> 1) << 15 | b[3] << 7 | b[4] >> 1


def demux(ts):
    """Returns dict(pat=[...], pmt=[...], pes=[{pts, pcr, rai, data, first_packet}], order=[('psi'|'pes', index)])."""
    assert len(ts) % 188 == 0, "not a whole number of packets"
    cc = {}
    pat, pmt, pes, order = [], [], [], []
    pmt_pid = video_pid = None
    cur = None
    for n in range(len(ts) // 188):
        p = ts[188 * n: 188 * n + 188]
        assert p[0] == 0x47, "sync byte"
        assert not p[1] & 0x80, "transport_error_indicator"
        pusi, pid = bool(p[1] & 0x40), ((p[1] & 0x1F) << 8) | p[2]
        afc, c = (p[3] >> 4) & 3, p[3] & 15
        assert not p[3] >> 6, "scrambled"
        assert afc in (1, 3), "every packet of this muxer carries payload"
        if pid in cc:
            assert c == (cc[pid] + 1) & 15, "continuity counter on PID %#x" % pid
        cc[pid] = c
        pos, pcr, rai = 4, None, False
        if afc & 2:
            al = p[4]
            assert al <= 183
            if al:
                fl = p[5]
                rai = bool(fl & 0x40)
                if fl & 0x10:
                    b = p[6:12]
                    pcr = (b[0] << 25 | b[1] << 17 | b[2] << 9 | b[3] << 1 | b[4] >> 7) * 300 + (((b[4] & 1) << 8) | b[5])
                    assert all(x == 0xFF for x in p[12:5 + al]), "stuffing bytes"
                else:
                    assert fl == 0 and all(x == 0xFF for x in p[6:5 + al])
            pos = 5 + al
        payload = p[pos:]
        if pid == 0 or pid == pmt_pid:
            assert pusi and payload[0] == 0, "pointer_field"
            sec = payload[1:]
            slen = ((sec[1] & 0x0F) << 8) | sec[2]
            body = sec[:3 + slen]
            assert sec[1] & 0x80, "section_syntax_indicator"
            assert crc32_mpeg2(body) == 0, "PSI CRC"
            assert all(x == 0xFF for x in sec[3 + slen:])
            if pid == 0:
                assert sec[0] == 0x00
                prog = (body[8] << 8) | body[9]
                pmt_pid = ((body[10] & 0x1F) << 8) | body[11]
                pat.append({"program": prog, "pmt_pid": pmt_pid, "tsid": (body[3] << 8) | body[4]})
                order.append(("pat", len(pat) - 1))
            else:
                assert sec[0] == 0x02
                pcr_pid = ((body[8] & 0x1F) << 8) | body[9]
                pil = ((body[10] & 0x0F) << 8) | body[11]
                es = body[12 + pil:-4]
                streams = []
                while es:
                    streams.append((es[0], ((es[1] & 0x1F) << 8) | es[2]))
                    es = es[5 + (((es[3] & 0x0F) << 8) | es[4]):]
                pmt.append({"pcr_pid": pcr_pid, "streams": streams})
                video_pid = streams[0][1]
                order.append(("pmt", len(pmt) - 1))
        else:
            assert video_pid is not None and pid == video_pid, "unexpected PID %#x" % pid
            if pusi:
                assert payload[:3] == b"\x00\x00\x01" and payload[3] == 0xE0
                assert payload[4] == 0 and payload[5] == 0, "video PES is unbounded"
                assert payload[6] & 0xC0 == 0x80 and payload[6] & 0x04, "data_alignment_indicator"
                assert payload[7] >> 6 == 2, "PTS only"
                hl = payload[8]
                assert payload[9] >> 4 == 2
                cur = {"pts": _ts33(payload[9:14]), "pcr": pcr, "rai": rai, "data": bytearray(payload[9 + hl:]), "first_packet": n}
                pes.append(cur)
                order.append(("pes", len(pes) - 1))
                assert pcr is not None, "PCR on every access unit start"
            else:
                assert cur is not None and pcr is None
                cur["data"] += payload
    return {"pat": pat, "pmt": pmt, "pes": pes, "order": order, "video_pid": video_pid, "pmt_pid": pmt_pid}
