// k_handover.hip -- the edges of the device pipeline: input conversion / padding, level packing for the host entropy coder
// Hand-written HIP for gfx950 (CDNA4, wave64); part of libmi355enc (see kernels_common.hpp).
#include "kernels_common.hpp"

// =================================================================== hand-over to the host entropy coder
// The kernels above leave 408 int16 per macroblock in HBM; at streaming bit rates almost all of
// them are zero.  Instead of copying the dense array over PCIe (6.6 MB per 1080p picture) the
// device packs what the CAVLC writer will actually read, in the order it reads it, straight into
// the pinned host buffer: per macroblock, 32-byte blocks
//     [the luma-DC slot, if it holds the Intra4x4 modes (mb_type == 2) or the vectors of partitions 1 .. 3 (mb_type == 1 with a shape in i16_mode)] [Intra16x16 DC, if NZ_LDC] [luma blkIdx b for every set bit b of nzmask]
//     [chroma DC (Cb 4 + Cr 4), if NZ_CBDC | NZ_CRDC] [chroma AC block i for every set bit 16 + i]
// The host walks the stream with a running pointer and needs no per-macroblock offsets.
#define PACK_CAND 27
DEV bool pack_slot(unsigned ty) { return (ty & 255u) == 2u || ((ty & 255u) == 1u && ((ty >> 8) & 3u) != 0u); } // ty: mb_type | i16_mode << 8 (word 1 of the record)
DEV int pack_count(unsigned nz, unsigned ty) {
    return __popc(nz & 0x01FFFFFFu) + ((nz & (NZ_CBDC | NZ_CRDC)) ? 1 : 0) + (pack_slot(ty) ? 1 : 0);
}
// exclusive prefix sum of the block counts: one workgroup, thread t owns a run of PER consecutive macroblocks whose counts
// stay in registers (all PER loads are in flight together: the kernel sits on the latency path of every access unit).
// PER = 0: any picture size, counts re-read in the second pass.
template <int PER>
__global__ __launch_bounds__(1024) void levels_scan_kernel(const mb_info_t *__restrict__ mbi, int nmb, int mbw, unsigned *__restrict__ off,
                                                           unsigned *__restrict__ hdr, const unsigned *__restrict__ err) {
    __shared__ unsigned wsum[16], csum[16];
    const int tid = threadIdx.x, per = PER ? PER : (nmb + 1023) / 1024, base = tid * per;
    unsigned cnt[PER ? PER : 1];
    unsigned mine = 0, cost = 0; // cost: sum of the macroblocks' costs (scene-cut recovery on the host); < 2^20 each
    if (PER) {
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int mb = base + i;
            const uint4 r = ldg128(&mbi[mb < nmb ? mb : nmb - 1]);
            cnt[i] = mb < nmb ? (unsigned)pack_count(r.z, r.y & 0xFFFFu) : 0u;
            mine += cnt[i];
            cost += mb < nmb ? r.w : 0u;
        }
    } else {
        for (int i = 0; i < per; i++) {
            const int mb = base + i;
            if (mb < nmb) { const uint4 r = ldg128(&mbi[mb]); mine += (unsigned)pack_count(r.z, r.y & 0xFFFFu); cost += r.w; }
        }
    }
    cost = (unsigned)wave64_sum((int)cost); // < 2^20 * 8 * 64 per wave: fits
    unsigned incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned v = __shfl_up(incl, d); if ((tid & 63) >= d) incl += v; }
    if ((tid & 63) == 63) { wsum[tid >> 6] = incl; csum[tid >> 6] = cost; }
    __syncthreads();
    unsigned before = 0;
    for (int w = 0; w < (tid >> 6); w++) before += wsum[w];
    unsigned run = before + incl - mine;
    if (PER) {
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int mb = base + i;
            if (mb < nmb) {
                off[mb] = run;
                if (mb % mbw == 0) hdr[2 + mb / mbw] = run; // where each macroblock row starts: lets the host code rows in parallel
                run += cnt[i];
            }
        }
    } else {
        for (int i = 0; i < per; i++) {
            const int mb = base + i;
            if (mb < nmb) {
                off[mb] = run;
                if (mb % mbw == 0) hdr[2 + mb / mbw] = run;
                const uint4 r = ldg128(&mbi[mb]); run += (unsigned)pack_count(r.z, r.y & 0xFFFFu);
            }
        }
    }
    if (tid == 1023) {
        hdr[0] = run; hdr[1] = ldg32(err); // total blocks; sticky error word of the band deblocker
        unsigned long long c64 = 0;
        for (int w = 0; w < 16; w++) c64 += csum[w];
        const int rows = (nmb + mbw - 1) / mbw;
        hdr[2 + rows] = (unsigned)c64; hdr[3 + rows] = (unsigned)(c64 >> 32); // after the row offsets
    }
}
// one wave per macroblock: lane c < 27 is one candidate block of the stream order above
__global__ __launch_bounds__(256) void levels_pack_kernel(const mb_info_t *__restrict__ mbi, const int16_t *__restrict__ levels, int nmb,
                                                          const unsigned *__restrict__ off, mb_info_t *__restrict__ h_mbi, int16_t *__restrict__ h_packed) {
    const int mb = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c = threadIdx.x & 63;
    if (mb >= nmb) return;
    const uint4 r = ldg128(&mbi[mb]);
    const unsigned nz = r.z;
    bool present = false;
    int src = 0; // int16 offset inside the macroblock's 408 levels
    if (c == 0) { present = pack_slot(r.y & 0xFFFFu); src = L_LDC; }
    else if (c == 1) { present = (nz & NZ_LDC) != 0; src = L_LDC; }
    else if (c < 18) { present = (nz >> (c - 2)) & 1; src = L_LUMA + (c - 2) * 16; }
    else if (c == 18) { present = (nz & (NZ_CBDC | NZ_CRDC)) != 0; src = L_CDC; }
    else if (c < PACK_CAND) { present = (nz >> (16 + c - 19)) & 1; src = L_CAC + (c - 19) * 16; }
    const unsigned long long m = __ballot(present);
    if (present) {
        const int rank = __popcll(m & ((1ull << c) - 1));
        const int16_t *sp = levels + (size_t)mb * MB_LEVELS + src;
        int16_t *dp = h_packed + ((size_t)ldg32(&off[mb]) + rank) * 16;
        const uint4 a = ldg128(sp), b = ldg128(sp + 8);
        stg128(dp, a); stg128(dp + 8, b);
    }
    if (c == PACK_CAND) stg128(&h_mbi[mb], r);
}
void k_launch_pack(const mb_info_t *d_mbi, const int16_t *d_levels, int nmb, int mbw, unsigned *d_off, mb_info_t *h_mbi, int16_t *h_packed,
                   unsigned *h_hdr, const unsigned *d_err, hipStream_t s) {
    if (nmb <= 4 * 1024) hipLaunchKernelGGL(levels_scan_kernel<4>, dim3(1), dim3(1024), 0, s, d_mbi, nmb, mbw, d_off, h_hdr, d_err);
    else if (nmb <= 8 * 1024) hipLaunchKernelGGL(levels_scan_kernel<8>, dim3(1), dim3(1024), 0, s, d_mbi, nmb, mbw, d_off, h_hdr, d_err); // 1080p: 8160
    else if (nmb <= 32 * 1024) hipLaunchKernelGGL(levels_scan_kernel<32>, dim3(1), dim3(1024), 0, s, d_mbi, nmb, mbw, d_off, h_hdr, d_err); // 2160p: 32400
    else hipLaunchKernelGGL(levels_scan_kernel<0>, dim3(1), dim3(1024), 0, s, d_mbi, nmb, mbw, d_off, h_hdr, d_err);
    hipLaunchKernelGGL(levels_pack_kernel, dim3((nmb + 3) / 4), dim3(256), 0, s, d_mbi, d_levels, nmb, d_off, h_mbi, h_packed);
}

// =================================================================== staging helper
// Replicate the last visible column/row into the coded-size margin of a staged source surface.
__global__ void pad_kernel(uint8_t *y, uint8_t *uv, int stride, int vw, int vh, int W, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nY = W * H;
    if (i < nY) {
        int r = i / W, c = i - r * W;
        if (r >= vh || c >= vw) y[(size_t)r * stride + c] = y[(size_t)(r < vh ? r : vh - 1) * stride + (c < vw ? c : vw - 1)];
    } else if (i < nY + nY / 2) {
        int j = i - nY, r = j / W, c = j - r * W;
        if (r >= vh / 2 || c >= vw) {
            int sc = c < vw ? c : vw - 2 + (c & 1);
            uv[(size_t)r * stride + c] = uv[(size_t)(r < vh / 2 ? r : vh / 2 - 1) * stride + sc];
        }
    }
}

// =================================================================== input conversion to NV12
// Replaces the `videoconvert` hop of the reference's pipelines for the raw formats its sources deliver
// (/root/reference/pipeline/generic/x264_superfast_camlink:4: v4l2src ... ! videoconvert ! x264enc): planar I420
// (jpegdec, videotestsrc) and packed 4:2:2 YUY2 / UYVY (capture cards).  One thread converts an 8 x 2 luma patch and
// its 4 chroma pairs: every global access is an aligned 8- or 16-byte word, reads and writes are contiguous per
// row, so the kernel runs at copy speed (pure HBM traffic: 1.5 P in + 1.5 P out for I420, 2 P + 1.5 P for 4:2:2).
// 4:2:2 -> 4:2:0 takes the rounded mean of the two chroma rows.  The coded-size margin (width/height not a
// multiple of 16) is filled by clamping the source coordinate, so no separate padding pass is needed.
struct csc_args {
    const uint8_t *p0, *p1, *p2; // I420: Y, U, V planes; packed formats: p0 only
    int s0, s1, s2;              // their strides in bytes
    uint8_t *dy, *duv;           // NV12 destination, coded size W x H, stride W
    int vw, vh, W, H;            // visible and coded size
};
template <int FMT> // 1 I420, 2 YUY2 (Y0 U Y1 V), 3 UYVY (U Y0 V Y1)
__global__ __launch_bounds__(256) void csc_kernel(csc_args a) {
    const int tx = blockIdx.x * 256 + threadIdx.x, per_row = a.W >> 3, rows2 = a.H >> 1;
    if (tx >= per_row * rows2) return;
    const int ry = tx / per_row, cx = tx - ry * per_row; // output luma rows 2ry, 2ry+1; luma columns 8cx..8cx+7
    // visible width is even; a patch is either fully visible, or clamped per byte through the slow path
    const int x0 = cx * 8;
    const bool fast = x0 + 8 <= a.vw && (a.s0 & 7) == 0 && (((uintptr_t)a.p0) & 7) == 0;
    uint2 yrow[2];
    unsigned uvw[2]; // chroma of this patch: 4 (U,V) pairs = 8 bytes
    if (FMT == 1) {
        const int cy = (2 * ry < a.vh ? 2 * ry : a.vh - 2) >> 1;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int sy = 2 * ry + r < a.vh ? 2 * ry + r : a.vh - 1;
            const uint8_t *sp = a.p0 + (size_t)sy * a.s0;
            if (fast) yrow[r] = ldg64(sp + x0);
            else {
                unsigned w[2] = {0, 0};
                for (int i = 0; i < 8; i++) { const int sx = x0 + i < a.vw ? x0 + i : a.vw - 1; w[i >> 2] |= ldg8(sp + sx) << (8 * (i & 3)); }
                yrow[r] = make_uint2(w[0], w[1]);
            }
        }
        unsigned u = 0, v = 0;
        const uint8_t *up = a.p1 + (size_t)cy * a.s1, *vp = a.p2 + (size_t)cy * a.s2;
        const int cw = a.vw >> 1;
        if (cx * 4 + 4 <= cw && ((a.s1 | a.s2) & 3) == 0 && ((((uintptr_t)a.p1) | ((uintptr_t)a.p2)) & 3) == 0) { u = ldg32(up + cx * 4); v = ldg32(vp + cx * 4); }
        else
            for (int i = 0; i < 4; i++) { const int sx = cx * 4 + i < cw ? cx * 4 + i : cw - 1; u |= ldg8(up + sx) << (8 * i); v |= ldg8(vp + sx) << (8 * i); }
        uvw[0] = (u & 0xFF) | ((v & 0xFF) << 8) | ((u & 0xFF00) << 8) | ((v & 0xFF00) << 16);
        uvw[1] = ((u >> 16) & 0xFF) | (((v >> 16) & 0xFF) << 8) | (((u >> 24) & 0xFF) << 16) | ((v >> 24) << 24);
    } else {
        unsigned c[2][2]; // per source row: 4 (U,V) pairs
        const int base = 2 * ry < a.vh ? 2 * ry : a.vh - 2; // margin rows repeat the last chroma row (mean of the last two source rows)
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int sy = base + r;
            const uint8_t *sp = a.p0 + (size_t)sy * a.s0;
            unsigned w[4];
            if (fast && (a.s0 & 15) == 0 && (((uintptr_t)a.p0) & 15) == 0) { const uint4 q = ldg128(sp + 2 * x0); w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w; }
            else
                for (int i = 0; i < 4; i++) { const int sx = x0 + 2 * i < a.vw ? x0 + 2 * i : a.vw - 2; w[i] = ldg8(sp + 2 * sx) | (ldg8(sp + 2 * sx + 1) << 8) | (ldg8(sp + 2 * sx + 2) << 16) | (ldg8(sp + 2 * sx + 3) << 24); }
            unsigned yy[2] = {0, 0}, cc[2] = {0, 0};
#pragma unroll
            for (int i = 0; i < 4; i++) { // one macropixel: 2 luma + (U,V)
                const unsigned m = w[i];
                const unsigned y1 = FMT == 2 ? (m >> 16) & 0xFF : m >> 24;
                const unsigned y0 = x0 + 2 * i >= a.vw ? y1 : (FMT == 2 ? m & 0xFF : (m >> 8) & 0xFF); // margin: the last visible sample, not the last pair
                const unsigned u = FMT == 2 ? (m >> 8) & 0xFF : m & 0xFF, v = FMT == 2 ? m >> 24 : (m >> 16) & 0xFF;
                yy[i >> 1] |= (y0 | (y1 << 8)) << (16 * (i & 1));
                cc[i >> 1] |= (u | (v << 8)) << (16 * (i & 1));
            }
            yrow[r] = make_uint2(yy[0], yy[1]);
            c[r][0] = cc[0]; c[r][1] = cc[1];
        }
        if (2 * ry >= a.vh) yrow[0] = yrow[1]; // ... and the last luma row
        uvw[0] = avg4(c[0][0], c[1][0]); uvw[1] = avg4(c[0][1], c[1][1]);
    }
    v2u t;
    t.x = yrow[0].x; t.y = yrow[0].y; *(GAS v2u *)(a.dy + (size_t)(2 * ry) * a.W + x0) = t;
    t.x = yrow[1].x; t.y = yrow[1].y; *(GAS v2u *)(a.dy + (size_t)(2 * ry + 1) * a.W + x0) = t;
    t.x = uvw[0]; t.y = uvw[1]; *(GAS v2u *)(a.duv + (size_t)ry * a.W + x0) = t;
}
int k_launch_csc(int fmt, const uint8_t *p0, const uint8_t *p1, const uint8_t *p2, int s0, int s1, int s2, uint8_t *dy, uint8_t *duv,
                 int vw, int vh, int W, int H, hipStream_t s) {
    csc_args a;
    a.p0 = p0; a.p1 = p1; a.p2 = p2; a.s0 = s0; a.s1 = s1; a.s2 = s2; a.dy = dy; a.duv = duv; a.vw = vw; a.vh = vh; a.W = W; a.H = H;
    const int n = (W >> 3) * (H >> 1);
    if (fmt == 1) hipLaunchKernelGGL(csc_kernel<1>, dim3((n + 255) / 256), dim3(256), 0, s, a);
    else if (fmt == 2) hipLaunchKernelGGL(csc_kernel<2>, dim3((n + 255) / 256), dim3(256), 0, s, a);
    else if (fmt == 3) hipLaunchKernelGGL(csc_kernel<3>, dim3((n + 255) / 256), dim3(256), 0, s, a);
    else return -1;
    return 0;
}
void k_launch_pad(uint8_t *y, uint8_t *uv, int stride, int vis_w, int vis_h, int W, int H, hipStream_t s) {
    int n = W * H + W * H / 2;
    hipLaunchKernelGGL(pad_kernel, dim3((n + 255) / 256), dim3(256), 0, s, y, uv, stride, vis_w, vis_h, W, H);
}

// =================================================================== adaptive quantisation
// One wave per macroblock (four per workgroup): sum and sum of squares of the 256 source luma samples by two packed dot products per
// lane and a wave reduction, then the offset rule of the oracle (orc_aq_offset_of) in the same integer arithmetic.
__global__ __launch_bounds__(256) void aq_kernel(const frame_ctx_t cv, int8_t *__restrict__ off) {
    const frame_ctx_t *__restrict__ ctx = &cv;
    const int lane = threadIdx.x & 63, mbn = blockIdx.x * 4 + (int)(threadIdx.x >> 6), nmb = ctx->mbw * ctx->mbh;
    if (mbn >= nmb) return; // wave-uniform
    const int my = mbn / ctx->mbw, mx = mbn - my * ctx->mbw;
    int sy = my * 16 + (lane >> 2);
    sy = sy < ctx->vis_h ? sy : ctx->vis_h - 1;
    const unsigned w = ldg32(ctx->src_y + (size_t)sy * ctx->src_stride + mx * 16 + 4 * (lane & 3));
    unsigned s = __builtin_amdgcn_sad_u8(w, 0u, 0u), s2 = __builtin_amdgcn_udot4(w, w, 0u, false);
    for (int o = 32; o; o >>= 1) { s += (unsigned)__shfl_xor((int)s, o, 64); s2 += (unsigned)__shfl_xor((int)s2, o, 64); }
    if (lane == 0) {
        const unsigned v = s2 - ((s * s) >> 8);
        int L2 = 0;
        if (v > 1) { const int msb = 31 - __builtin_clz(v); L2 = 2 * msb + (int)((v >> (msb - 1)) & 1u); }
        int o = (3 * (L2 - 28) + 4) >> 3;
        o = o < -4 ? -4 : (o > 4 ? 4 : o);
        off[mbn] = (int8_t)o;
    }
}
void k_launch_aq(const frame_ctx_t *h_ctx, int8_t *d_off, hipStream_t s) {
    hipLaunchKernelGGL(aq_kernel, dim3((h_ctx->mbw * h_ctx->mbh + 3) / 4), dim3(256), 0, s, *h_ctx, d_off);
}
// 7.4.5: a macroblock without mb_qp_delta (not Intra_16x16 and no coded block) has the QP_Y of the macroblock before it in decoding
// order.  One workgroup: every thread walks a run of consecutive macroblocks, the runs' "last coded QP" are carried forward by a scan over
// the 1024 threads (a thread whose run has no coded macroblock passes its predecessor's on), then every thread rewrites the qp byte of its
// uncoded macroblocks.  The records are otherwise final; the deblocker that follows in stream order reads QP_Y from them.
__global__ __launch_bounds__(1024) void qp_chain_kernel(mb_info_t *__restrict__ mbi, int nmb, int slice_qp) {
    __shared__ int carry[1024];
    const int t = threadIdx.x, per = (nmb + 1023) / 1024, i0 = t * per, i1 = i0 + per < nmb ? i0 + per : nmb;
    int last = -1; // this run's last coded QP
    for (int i = i0; i < i1; i++) {
        const uint4 r = ldg128(&mbi[i]);
        if ((r.y & 255u) == 0u || (r.z & 0x07FFFFFFu) != 0u) last = (int)(r.y >> 24);
    }
    carry[t] = last;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { // inclusive "last defined" scan
        const int a = carry[t], b = t >= d ? carry[t - d] : -1;
        __syncthreads();
        carry[t] = a >= 0 ? a : b;
        __syncthreads();
    }
    int prev = t > 0 ? carry[t - 1] : -1;
    if (prev < 0) prev = slice_qp;
    for (int i = i0; i < i1; i++) {
        const uint4 r = ldg128(&mbi[i]);
        if ((r.y & 255u) == 0u || (r.z & 0x07FFFFFFu) != 0u) prev = (int)(r.y >> 24);
        else stg32((unsigned *)&mbi[i] + 1, (r.y & 0x00FFFFFFu) | ((unsigned)prev << 24));
    }
}
void k_launch_qp_chain(mb_info_t *d_mbi, int nmb, int slice_qp, int slice_mbs, hipStream_t s) { // slice_mbs > 0: the chain starts again with every slice
    if (slice_mbs <= 0) slice_mbs = nmb;
    for (int i = 0; i < nmb; i += slice_mbs) hipLaunchKernelGGL(qp_chain_kernel, dim3(1), dim3(1024), 0, s, d_mbi + i, nmb - i < slice_mbs ? nmb - i : slice_mbs, slice_qp);
}
