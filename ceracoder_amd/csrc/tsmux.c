/* Minimal MPEG-2 TS muxer for one H.264 elementary stream (ISO/IEC 13818-1: 2.4.3.2 transport
 * packet, 2.4.3.4 adaptation field / PCR, 2.4.3.6 PES packet, 2.4.4.3 PAT, 2.4.4.8 PMT, 2.14.1 AVC
 * carriage).  Host-only; see include/mi355ts.h for what it stands in for in the reference's pipelines. */
#include "../../include/mi355ts.h"
#include <stdlib.h>
#include <string.h>

#define TS_CLOCK_BASE 90000       /* all time stamps start one second in, so PCR = PTS - delay never goes negative */
#define TS_PCR_LEAD 11250         /* PTS runs 125 ms ahead of the PCR carried in the same access unit */
#define TS_PSI_INTERVAL 9000      /* PAT/PMT at least every 100 ms of stream time */

struct mi355ts {
    unsigned cc_pat, cc_pmt, cc_vid;
    int64_t last_psi; /* 90 kHz time of the last PAT/PMT, or -1 */
    uint8_t pat[MI355TS_PACKET], pmt[MI355TS_PACKET]; /* pre-built, only the continuity counter changes */
};

uint32_t mi355ts_crc32(const uint8_t *p, size_t n) {
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) {
        c ^= (uint32_t)p[i] << 24;
        for (int k = 0; k < 8; k++) c = (c & 0x80000000u) ? (c << 1) ^ 0x04C11DB7u : (c << 1);
    }
    return c;
}

static void psi_packet(uint8_t *pkt, unsigned pid, const uint8_t *section, size_t n) { /* section without CRC */
    memset(pkt, 0xFF, MI355TS_PACKET);
    pkt[0] = 0x47; pkt[1] = (uint8_t)(0x40 | (pid >> 8)); pkt[2] = (uint8_t)pid; pkt[3] = 0x10; /* PUSI, payload only */
    pkt[4] = 0x00;                                                                              /* pointer_field */
    memcpy(pkt + 5, section, n);
    const uint32_t crc = mi355ts_crc32(section, n);
    pkt[5 + n] = (uint8_t)(crc >> 24); pkt[6 + n] = (uint8_t)(crc >> 16); pkt[7 + n] = (uint8_t)(crc >> 8); pkt[8 + n] = (uint8_t)crc;
}

mi355ts_t *mi355ts_open(void) {
    mi355ts_t *m = (mi355ts_t *)calloc(1, sizeof *m);
    if (!m) return NULL;
    m->last_psi = -1;
    /* program_association_section: transport_stream_id 1, program 1 -> PMT PID */
    const uint8_t pat[] = {0x00, 0xB0, 0x0D, 0x00, 0x01, 0xC1, 0x00, 0x00, 0x00, 0x01, (uint8_t)(0xE0 | (MI355TS_PID_PMT >> 8)), (uint8_t)MI355TS_PID_PMT};
    /* TS_program_map_section: PCR on the video PID, one stream of type 0x1B (AVC) */
    const uint8_t pmt[] = {0x02, 0xB0, 0x12, 0x00, 0x01, 0xC1, 0x00, 0x00, (uint8_t)(0xE0 | (MI355TS_PID_VIDEO >> 8)), (uint8_t)MI355TS_PID_VIDEO, 0xF0, 0x00,
                           0x1B, (uint8_t)(0xE0 | (MI355TS_PID_VIDEO >> 8)), (uint8_t)MI355TS_PID_VIDEO, 0xF0, 0x00};
    psi_packet(m->pat, MI355TS_PID_PAT, pat, sizeof pat);
    psi_packet(m->pmt, MI355TS_PID_PMT, pmt, sizeof pmt);
    return m;
}
void mi355ts_close(mi355ts_t *m) { free(m); }

size_t mi355ts_bound(size_t au_len) { return (size_t)MI355TS_PACKET * (2 + (au_len + 14 + 6 + 8 + 183) / 184 + 1); }

static void put_ts33(uint8_t *p, unsigned marker, uint64_t v) { /* 33-bit time stamp in the 5-byte PES layout */
    p[0] = (uint8_t)((marker << 4) | (((v >> 30) & 7) << 1) | 1);
    p[1] = (uint8_t)(v >> 22);
    p[2] = (uint8_t)((((v >> 15) & 0x7F) << 1) | 1);
    p[3] = (uint8_t)(v >> 7);
    p[4] = (uint8_t)(((v & 0x7F) << 1) | 1);
}

int mi355ts_mux(mi355ts_t *m, const uint8_t *au, size_t au_len, int64_t pts_ns, int keyframe, uint8_t *out, size_t cap, size_t *out_len) {
    if (!m || !au || !au_len || !out || !out_len || pts_ns < 0) return -1;
    if (cap < mi355ts_bound(au_len)) return -2;
    const uint64_t t90 = (uint64_t)(pts_ns / 100000 * 9 + (pts_ns % 100000) * 9 / 100000);
    const uint64_t pts = (t90 + TS_CLOCK_BASE) & 0x1FFFFFFFFull, pcr = (t90 + TS_CLOCK_BASE - TS_PCR_LEAD) & 0x1FFFFFFFFull;
    uint8_t *o = out;
    if (keyframe || m->last_psi < 0 || (int64_t)t90 - m->last_psi >= TS_PSI_INTERVAL) {
        memcpy(o, m->pat, MI355TS_PACKET); o[3] = (uint8_t)(0x10 | (m->cc_pat++ & 15)); o += MI355TS_PACKET;
        memcpy(o, m->pmt, MI355TS_PACKET); o[3] = (uint8_t)(0x10 | (m->cc_pmt++ & 15)); o += MI355TS_PACKET;
        m->last_psi = (int64_t)t90;
    }
    /* PES header (PTS only) + optional access unit delimiter, then the access unit */
    uint8_t head[20];
    size_t hn = 0;
    head[hn++] = 0; head[hn++] = 0; head[hn++] = 1; head[hn++] = 0xE0; /* packet_start_code_prefix, stream_id: video 0 */
    head[hn++] = 0; head[hn++] = 0;                                     /* PES_packet_length 0: unbounded (video in TS) */
    head[hn++] = 0x84;                                                  /* '10', data_alignment_indicator */
    head[hn++] = 0x80;                                                  /* PTS_DTS_flags '10' */
    head[hn++] = 5;
    put_ts33(head + hn, 2, pts); hn += 5;
    size_t skip = 0; /* locate the first NAL header to see whether an AUD is already there */
    while (skip + 3 < au_len && !(au[skip] == 0 && au[skip + 1] == 0 && au[skip + 2] == 1)) skip++;
    const int has_aud = skip + 3 < au_len && (au[skip + 3] & 0x1F) == 9;
    if (!has_aud) { const uint8_t aud[6] = {0, 0, 0, 1, 0x09, 0xF0}; memcpy(head + hn, aud, 6); hn += 6; }
    size_t left = hn + au_len, hpos = 0, apos = 0;
    int first = 1;
    while (left) {
        o[0] = 0x47;
        o[1] = (uint8_t)((first ? 0x40 : 0) | (MI355TS_PID_VIDEO >> 8));
        o[2] = (uint8_t)MI355TS_PID_VIDEO;
        size_t af = first ? 8 : 0; /* adaptation field bytes including its length byte */
        size_t room = 184 - af;
        if (left < room) { af = 184 - left; room = left; } /* stuffing grows (or creates) the adaptation field */
        o[3] = (uint8_t)((af ? 0x30 : 0x10) | (m->cc_vid++ & 15));
        uint8_t *p = o + 4;
        if (af) {
            p[0] = (uint8_t)(af - 1);
            if (af > 1) {
                p[1] = first ? (uint8_t)(0x10 | (keyframe ? 0x40 : 0)) : 0; /* PCR_flag, random_access_indicator */
                size_t used = 2;
                if (first) {
                    p[2] = (uint8_t)(pcr >> 25); p[3] = (uint8_t)(pcr >> 17); p[4] = (uint8_t)(pcr >> 9); p[5] = (uint8_t)(pcr >> 1);
                    p[6] = (uint8_t)(((pcr & 1) << 7) | 0x7E); p[7] = 0; /* reserved bits set, extension 0 */
                    used = 8;
                }
                memset(p + used, 0xFF, af - used);
            }
            p += af;
        }
        size_t n = room;
        if (hpos < hn) { size_t k = hn - hpos < n ? hn - hpos : n; memcpy(p, head + hpos, k); hpos += k; p += k; n -= k; }
        if (n) { memcpy(p, au + apos, n); apos += n; }
        left -= room;
        o += MI355TS_PACKET;
        first = 0;
    }
    *out_len = (size_t)(o - out);
    return 0;
}
