// k_motion.hip -- full-search motion estimation and sub-sample refinement
// Hand-written HIP for gfx950 (CDNA4, wave64); part of libmi355enc (see kernels_common.hpp).
#include "kernels_common.hpp"

// =================================================================== motion search
// One workgroup = ME_MBS horizontally adjacent macroblocks of one macroblock row, one wave per
// macroblock.  The 48 x (16*ME_MBS+32) luma search window (+-16 around the strip) is staged once in
// LDS.  Lane l < 63 of a wave owns the candidates
//   dy in [-16 + 5*(l/9), +5)   x   dx in [-16 + 4*(l%9), +4)
// (7 x 9 tiles cover 35 x 36 >= 33 x 33; 85 % of the computed SADs are real candidates) and
// accumulates them with v_qsad_pk_u16_u8 -- four 4-pixel SADs per instruction -- re-using each
// window row for the 5 dy it serves.  The (cost, dy, dx) minimum is reduced over the wave with
// cross-lane shuffles.
#ifndef ME_MBS
#define ME_MBS 4
#endif
#define ME_WQ (ME_MBS + 2)      /* uint4 per window row: 16*ME_MBS + 32 bytes */
#define ME_ROWS 50   /* 48 real rows + 2 that only masked candidates (dy = 17, 18) ever touch */
#define ME_STRIDE 53 /* words; 5*53 mod 32 = 9 -> consecutive dy-groups start 9 banks apart */
#define ME_K 5       /* dy per lane */

DEV unsigned long long qsad(unsigned lo, unsigned hi, unsigned cur, unsigned long long acc) {
    unsigned long long src = ((unsigned long long)hi << 32) | lo;
    return __builtin_amdgcn_qsad_pk_u16_u8(src, cur, acc);
}
DEV int mv_bits(int v) { // bits of se(4v): 1 for 0, else 7 + 2*floor(log2|v|)
    int a = iabs(v);
    return a == 0 ? 1 : 7 + 2 * (31 - __clz(a));
}

__global__ __launch_bounds__(64 * ME_MBS) void me_kernel(const frame_ctx_t cv, int row0) { // context by value: lives in the kernarg segment, no per-picture upload
    const frame_ctx_t *__restrict__ ctx = &cv;
    __shared__ unsigned win[ME_ROWS * ME_STRIDE];
    const int stride = ctx->stride, mbw = ctx->mbw, mbh = ctx->mbh;
    const int W = mbw * 16, H = mbh * 16;
    const int strips = (mbw + ME_MBS - 1) / ME_MBS;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int my = row0 + tile / strips, sx = tile % strips; // the launch covers macroblock rows row0 .. row0 + gridDim.x / strips - 1
    const int t = threadIdx.x;
    const uint8_t *__restrict__ ref = ctx->ref_y;

    // ---- stage the window: 48 rows x 10 uint4 (coalesced 16 B per lane)
    for (int i = t; i < 48 * ME_WQ; i += 64 * ME_MBS) {
        int row = i / ME_WQ, q = i - row * ME_WQ;
        int gy = my * 16 - 16 + row, gx = sx * (ME_MBS * 16) - 16 + 16 * q;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = ldg128(ref + (size_t)gy * stride + gx);
        unsigned *d = &win[row * ME_STRIDE + 4 * q];
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    if (t < ME_ROWS) win[t * ME_STRIDE + 4 * ME_WQ] = 0;
    if (t < 2 * (4 * ME_WQ + 1)) win[(48 + t / (4 * ME_WQ + 1)) * ME_STRIDE + t % (4 * ME_WQ + 1)] = 0;

    const int lane = t & 63, m = t >> 6;
    const int mx = sx * ME_MBS + m;
    const bool active = lane < 63 && mx < mbw;
    const int g = lane < 63 ? lane / 9 : 0, dxg = lane < 63 ? lane % 9 : 0;
    const int mxc = mx < mbw ? mx : mbw - 1;

    // ---- current macroblock: 16 rows x 4 words, identical in every lane of the wave
    unsigned c[16][4];
    {
        const uint8_t *__restrict__ src = ctx->src_y;
        const int ss = ctx->src_stride, vh = ctx->vis_h;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            int sy = my * 16 + r;
            sy = sy < vh ? sy : vh - 1;
            uint4 v = ldg128(src + (size_t)sy * ss + mxc * 16);
            c[r][0] = v.x; c[r][1] = v.y; c[r][2] = v.z; c[r][3] = v.w;
        }
    }
    __syncthreads();

    unsigned long long acc[ME_K];
#pragma unroll
    for (int d = 0; d < ME_K; d++) acc[d] = 0;
    const unsigned *wp = &win[(ME_K * g) * ME_STRIDE + 4 * m + dxg];
    // one window row ahead in registers; sched_barrier keeps the compiler from hoisting all 100 LDS
    // reads to the top (which costs > 200 VGPRs and halves the occupancy)
    unsigned w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3], w4 = wp[4];
#pragma unroll
    for (int j = 0; j < 16 + ME_K - 1; j++) {
        unsigned n0 = 0, n1 = 0, n2 = 0, n3 = 0, n4 = 0;
        if (j + 1 < 16 + ME_K - 1) {
            n0 = wp[(j + 1) * ME_STRIDE + 0]; n1 = wp[(j + 1) * ME_STRIDE + 1]; n2 = wp[(j + 1) * ME_STRIDE + 2];
            n3 = wp[(j + 1) * ME_STRIDE + 3]; n4 = wp[(j + 1) * ME_STRIDE + 4];
        }
#pragma unroll
        for (int d = 0; d < ME_K; d++) {
            const int r = j - d;
            if (r >= 0 && r < 16) {
                acc[d] = qsad(w0, w1, c[r][0], acc[d]);
                acc[d] = qsad(w1, w2, c[r][1], acc[d]);
                acc[d] = qsad(w2, w3, c[r][2], acc[d]);
                acc[d] = qsad(w3, w4, c[r][3], acc[d]);
            }
        }
        // pin this row's SADs here (pure intrinsics would otherwise sink below all the LDS reads)
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]));
        __builtin_amdgcn_sched_barrier(0);
        w0 = n0; w1 = n1; w2 = n2; w3 = n3; w4 = n4;
    }

    // ---- cost = SAD + lambda*(bits(dx)+bits(dy)); key = cost<<12 | (dy+16)<<6 | (dx+16)
    const int R = ctx->me_range, lambda = ctx->lambda;
    const int x0 = mxc * 16, y0 = my * 16;
    const int dx_lo = -R < -x0 ? -x0 : -R, dx_hi = R > W - 16 - x0 ? W - 16 - x0 : R;
    const int dy_lo = -R < -y0 ? -y0 : -R, dy_hi = R > H - 16 - y0 ? H - 16 - y0 : R;
    const unsigned INVALID = 0x40000000u;
    unsigned bo[4];
#pragma unroll
    for (int o = 0; o < 4; o++) {
        int dx = -16 + 4 * dxg + o;
        bo[o] = (dx >= dx_lo && dx <= dx_hi && active) ? (((unsigned)(lambda * mv_bits(dx)) << 12) | (unsigned)(dx + 16)) : INVALID;
    }
    unsigned best = 0xFFFFFFFFu;
#pragma unroll
    for (int d = 0; d < ME_K; d++) {
        int dy = -16 + ME_K * g + d;
        unsigned bd = (dy >= dy_lo && dy <= dy_hi) ? (((unsigned)(lambda * mv_bits(dy)) << 12) | ((unsigned)(dy + 16) << 6)) : INVALID;
        unsigned lo = (unsigned)acc[d], hi = (unsigned)(acc[d] >> 32);
        unsigned k0 = ((lo << 16) >> 4) + bd + bo[0];
        unsigned k1 = ((lo & 0xFFFF0000u) >> 4) + bd + bo[1];
        unsigned k2 = ((hi << 16) >> 4) + bd + bo[2];
        unsigned k3 = ((hi & 0xFFFF0000u) >> 4) + bd + bo[3];
        unsigned ka = k0 < k1 ? k0 : k1, kb = k2 < k3 ? k2 : k3;
        ka = ka < kb ? ka : kb;
        best = best < ka ? best : ka;
    }
    // ---- wave-wide minimum
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
        unsigned o = (unsigned)__shfl_xor((int)best, sft, 64);
        best = best < o ? best : o;
    }
    if (lane == 0 && mx < mbw) {
        mb_info_t *mb = &ctx->mbi[my * mbw + mx];
        const int bx_ = (int)(best & 63) - 16, by_ = (int)((best >> 6) & 63) - 16;
        stg32(&mb->mvx, ((unsigned)(uint16_t)(4 * bx_)) | ((unsigned)(uint16_t)(4 * by_) << 16)); // quarter-sample units
        stg32(&mb->cost, best >> 12);
    }
}

// =================================================================== sub-sample refinement
// One wave per macroblock.  Around the integer winner (ix, iy) the wave builds, in LDS, the
// integer samples G and the three half-sample planes of 8.4.2.2.1 (b: horizontal 6-tap,
// h: vertical 6-tap, j: centre, 6-tap over the unrounded horizontal intermediates) on an
// 18 x 18 (+1) grid; every quarter-sample candidate is then the rounded average of two plane
// entries (Table 8-12).  Two rounds (step 2, then step 1) of the 8 neighbours, visited in
// (dy, dx) raster order, strictly-lower cost wins -- the oracle's orc_subpel_frame.
#define SP_GS 24 /* G row stride (23 used) */
#define SP_PS 20 /* plane row stride (18/19 used) */
struct sp_lds {
    uint8_t G[23 * SP_GS];     // rows iy-3 .. iy+19, cols ix-3 .. ix+19 (+1 spare: a row is 6 dwords)
    int16_t H1[18 * SP_GS];    // unrounded vertical half samples at every G column: rows iy-1 .. iy+16 (taps G rows r .. r+5)
    uint8_t b[19 * SP_PS];     // rows iy-1 .. iy+17, cols ix-1 .. ix+16
    uint8_t h[18 * SP_GS];     // rows iy-1 .. iy+16, indexed by G column (ix-1 is column 2): dword stores stay aligned
    uint8_t j[18 * SP_PS];     // rows iy-1 .. iy+16, cols ix-1 .. ix+16
    uint8_t pad[16];           // lds4() may read one word past the last sample of a plane
};
typedef short sp_s2 __attribute__((ext_vector_type(2)));
DEV sp_s2 as_s2(unsigned v) { return __builtin_bit_cast(sp_s2, v); }
DEV unsigned as_u(sp_s2 v) { return __builtin_bit_cast(unsigned, v); }
// bytes 0,1 / 2,3 of a word as two 16-bit lanes (v_perm_b32; selector 0x0c = constant zero)
DEV sp_s2 bytes_lo(unsigned w) { return as_s2(__builtin_amdgcn_perm(0u, w, 0x0c010c00u)); }
DEV sp_s2 bytes_hi(unsigned w) { return as_s2(__builtin_amdgcn_perm(0u, w, 0x0c030c02u)); }
DEV sp_s2 clip255_s2(sp_s2 v) { return __builtin_elementwise_min(__builtin_elementwise_max(v, (sp_s2)(0)), (sp_s2)(255)); }
// the low bytes of the four 16-bit lanes of (lo, hi) as one word
// four samples (already shifted, not yet clipped; each fits 16 bits) -> clipped bytes of one word, through the packed 16-bit
// forms.  Not `clip255(a >> n) | clip255(b >> n) << 8 | ...` on 32-bit values: for that hipcc (ROCm 7.2) selects gfx950's
// v_ashr_pk_u8_i32 for the first pair and ORs the other two into bits 31:16 of its result, which the instruction does not
// clear on this hardware (tools/ubench_planes.hip shows samples 2 and 3 of every word wrong).
DEV sp_s2 pair_s2(int a, int b) { return as_s2(__builtin_amdgcn_perm((unsigned)b, (unsigned)a, 0x05040100u)); }
DEV unsigned pack_s2(sp_s2 lo, sp_s2 hi) { return __builtin_amdgcn_perm(as_u(hi), as_u(lo), 0x06040200u); }
DEV unsigned clip_pack4(int a, int b, int c, int d) { return pack_s2(clip255_s2(pair_s2(a, b)), clip255_s2(pair_s2(c, d))); }
DEV int mvq_bits(int q) { // bits of se(q)
    unsigned k = q > 0 ? (unsigned)(2 * q - 1) : (unsigned)(-2 * q);
    return 2 * (31 - __clz((int)(k + 1))) + 1;
}
// four horizontally adjacent bytes of an LDS plane starting at byte offset `o` (any alignment): two aligned words + one v_alignbyte
DEV unsigned lds4(const uint8_t *plane, int o) {
    const unsigned *w = (const unsigned *)(plane + (o & ~3));
    return __builtin_amdgcn_alignbyte(w[1], w[0], (unsigned)(o & 3));
}
// the four luma samples at plane positions (X..X+3, Y) (plane coordinates: 0 = ix-1 / iy-1) and fraction (fx, fy), one per byte
// (8.4.2.2.1, Table 8-12).  fx, fy are wave-uniform, so the case analysis costs no divergence.
DEV unsigned sp_sample4(const sp_lds *L, int X, int Y, int fx, int fy) {
#define SG(x, y) lds4(L->G, ((y) + 2) * SP_GS + (x) + 2)
#define SB(x, y) lds4(L->b, (y) * SP_PS + (x))
#define SH(x, y) lds4(L->h, (y) * SP_GS + (x) + 2)
#define SJ(x, y) lds4(L->j, (y) * SP_PS + (x))
    if (fy == 0) {
        if (fx == 0) return SG(X, Y);
        return fx == 2 ? SB(X, Y) : fx == 1 ? avg4(SG(X, Y), SB(X, Y)) : avg4(SG(X + 1, Y), SB(X, Y));
    }
    if (fx == 0) return fy == 2 ? SH(X, Y) : fy == 1 ? avg4(SG(X, Y), SH(X, Y)) : avg4(SG(X, Y + 1), SH(X, Y));
    if ((fx & 1) && (fy & 1)) return avg4(fy == 1 ? SB(X, Y) : SB(X, Y + 1), fx == 1 ? SH(X, Y) : SH(X + 1, Y));
    if (fx == 2 && fy == 2) return SJ(X, Y);
    if (fx == 2) return avg4(fy == 1 ? SB(X, Y) : SB(X, Y + 1), SJ(X, Y));
    return avg4(fx == 1 ? SH(X, Y) : SH(X + 1, Y), SJ(X, Y));
#undef SG
#undef SB
#undef SH
#undef SJ
}
// G (filled, visible to the wave) -> H1, h, b, j
DEV void sp_planes(sp_lds *L, const int lane) {
    // ---- the three half-sample planes (8.4.2.2.1), four samples per lane and pass:
    //  * vertical 6-tap as packed 16-bit arithmetic on whole words of G (a lane owns one word column and two output rows,
    //    sliding over 7 input rows): unrounded H1 at every G column, rounded h;
    //  * horizontal 6-tap as two v_dot4_i32_i8 per sample: samples are biased to signed bytes (x ^ 0x80), the taps
    //    (1,-5,20,20 | -5,1,0,0) are byte constants and the bias returns as 128 * 32 in the accumulator: rounded b;
    //  * centre samples j = horizontal 6-tap over H1 (the standard allows either order), three v_dot2_i32_i16 per sample.
    if (lane < 54) {
        const int d = lane % 6, seg = lane / 6;
        sp_s2 lo[7], hi[7];
#pragma unroll
        for (int r = 0; r < 7; r++) {
            const unsigned w = *(const unsigned *)&L->G[(2 * seg + r) * SP_GS + 4 * d];
            lo[r] = bytes_lo(w); hi[r] = bytes_hi(w);
        }
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const sp_s2 vl = (lo[t + 2] + lo[t + 3]) * (sp_s2)(20) - (lo[t + 1] + lo[t + 4]) * (sp_s2)(5) + (lo[t] + lo[t + 5]);
            const sp_s2 vh = (hi[t + 2] + hi[t + 3]) * (sp_s2)(20) - (hi[t + 1] + hi[t + 4]) * (sp_s2)(5) + (hi[t] + hi[t + 5]);
            const int R = 2 * seg + t;
            unsigned *o = (unsigned *)&L->H1[R * SP_GS + 4 * d];
            o[0] = as_u(vl); o[1] = as_u(vh);
            *(unsigned *)&L->h[R * SP_GS + 4 * d] = pack_s2(clip255_s2((vl + (sp_s2)(16)) >> (sp_s2)(5)), clip255_s2((vh + (sp_s2)(16)) >> (sp_s2)(5)));
        }
    }
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int i = lane + 64 * it;
        if (i < 19 * 5) {
            const int R = i / 5, g = i - R * 5;
            const unsigned *gw = (const unsigned *)&L->G[(R + 2) * SP_GS + 4 * g];
            const unsigned d0 = gw[0] ^ 0x80808080u, d1 = gw[1] ^ 0x80808080u, d2 = gw[2] ^ 0x80808080u; // g == 4 reads into the next row: unused samples only
            int o[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned w0 = k ? __builtin_amdgcn_alignbyte(d1, d0, (unsigned)k) : d0, w1 = k ? __builtin_amdgcn_alignbyte(d2, d1, (unsigned)k) : d1;
                const int v = __builtin_amdgcn_sdot4((int)w0, 0x1414FB01, __builtin_amdgcn_sdot4((int)w1, 0x000001FB, 4096, false), false);
                o[k] = (v + 16) >> 5;
            }
            *(unsigned *)&L->b[R * SP_PS + 4 * g] = clip_pack4(o[0], o[1], o[2], o[3]);
        }
    }
    WAVE_SYNC();
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int i = lane + 64 * it;
        if (i < 18 * 5) {
            const int R = i / 5, g = i - R * 5;
            const unsigned *hw = (const unsigned *)&L->H1[R * SP_GS + 4 * g]; // pairs (4g + 2m, 4g + 2m + 1)
            const unsigned p0 = hw[0], p1 = hw[1], p2 = hw[2], p3 = hw[3], p4 = hw[4];
            const unsigned q0 = __builtin_amdgcn_alignbyte(p1, p0, 2u), q1 = __builtin_amdgcn_alignbyte(p2, p1, 2u),
                           q2 = __builtin_amdgcn_alignbyte(p3, p2, 2u), q3 = __builtin_amdgcn_alignbyte(p4, p3, 2u);
            const sp_s2 ca = {1, -5}, cb = {20, 20}, cc = {-5, 1};
#define J3(a, b, c) __builtin_amdgcn_sdot2(as_s2(a), ca, __builtin_amdgcn_sdot2(as_s2(b), cb, __builtin_amdgcn_sdot2(as_s2(c), cc, 512, false), false), false)
            const int v0 = J3(p0, p1, p2), v1 = J3(q0, q1, q2), v2 = J3(p1, p2, p3), v3 = J3(q1, q2, q3);
#undef J3
            *(unsigned *)&L->j[R * SP_PS + 4 * g] = clip_pack4(v0 >> 10, v1 >> 10, v2 >> 10, v3 >> 10);
        }
    }
    WAVE_SYNC();
}
__global__ __launch_bounds__(256) void subpel_kernel(const frame_ctx_t cv, int mb0, int mb1) {
    const frame_ctx_t *__restrict__ ctx = &cv;
    __shared__ __attribute__((aligned(16))) sp_lds LD[4];
    const int mbw = ctx->mbw, mbh = ctx->mbh, stride = ctx->stride, W = mbw * 16, H = mbh * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int mbn = mb0 + blockIdx.x * 4 + wave; // the launch covers macroblocks mb0 .. mb1-1
    const bool ok = mbn < mb1;
    if (!ok) mbn = mb1 - 1;
    const int my = mbn / mbw, mx = mbn - my * mbw, x0 = mx * 16, y0 = my * 16;
    sp_lds *L = &LD[wave];
    const mb_info_t info = ld_mbinfo(&ctx->mbi[mbn]);
    const int ix = x0 + (info.mvx >> 2), iy = y0 + (info.mvy >> 2); // integer winner (vector is a multiple of 4 here)
    const uint8_t *__restrict__ ref = ctx->ref_y;
    // ---- G with the picture extended by coordinate clamping (8.4.2.2.1).  Window fully inside the picture (the usual
    // case, wave-uniform): 23 rows x 7 aligned words, shifted into place with v_alignbyte; otherwise byte by byte.
    if (ix - 3 >= 0 && iy - 3 >= 0 && ((ix - 3) & ~3) + 28 <= W && iy + 19 < H) {
        const int a = (ix - 3) & 3;
        const uint8_t *base = ref + (size_t)(iy - 3) * stride + ((ix - 3) & ~3);
        for (int i = lane; i < 23 * 6; i += 64) {
            const int r = i / 6, d = i - r * 6;
            const unsigned w0 = ldg32(base + (size_t)r * stride + 4 * d), w1 = ldg32(base + (size_t)r * stride + 4 * d + 4);
            *(unsigned *)&L->G[r * SP_GS + 4 * d] = __builtin_amdgcn_alignbyte(w1, w0, (unsigned)a);
        }
    } else
        for (int i = lane; i < 23 * 23; i += 64) {
            int r = i / 23, c = i - r * 23;
            int yy = clip3(0, H - 1, iy - 3 + r), xx = clip3(0, W - 1, ix - 3 + c);
            L->G[r * SP_GS + c] = (uint8_t)ldg8(ref + (size_t)yy * stride + xx);
        }
    // current macroblock: lane owns row lane>>2, columns 4*(lane&3) .. +3
    const int pr = lane >> 2, pc = (lane & 3) * 4;
    unsigned curw;
    {
        int sy = y0 + pr;
        sy = sy < ctx->vis_h ? sy : ctx->vis_h - 1;
        curw = ldg32(ctx->src_y + (size_t)sy * ctx->src_stride + x0 + pc);
    }
    WAVE_SYNC();
    sp_planes(L, lane);
    // ---- two refinement rounds (half, then quarter).  The 8 candidates of a round are scored together: per lane one
    // v_sad_u8 over its 4 pixels each, two 16-bit partial sums per register (64 lanes x 1020 < 65536), one wave reduction
    // for all of them; then the candidates are compared in scan order with a strict `<`, as the oracle does.
    const int lambda = ctx->lambda;
    int bqx = info.mvx, bqy = info.mvy;
    unsigned best = info.cost;
#pragma unroll 1
    for (int step = 2; step >= 1; step--) {
        const int cqx = bqx, cqy = bqy;
        unsigned acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int c8 = 0; c8 < 8; c8++) {
            const int k = c8 < 4 ? c8 : c8 + 1;
            const int qx = cqx + (k % 3 - 1) * step, qy = cqy + (k / 3 - 1) * step;
            const int ox = qx - info.mvx, oy = qy - info.mvy;               // -3 .. 3 relative to the integer winner
            const int X = 1 + (ox >> 2) + pc, Y = 1 + (oy >> 2) + pr;       // plane coordinates of this lane's first pixel
            const unsigned sad = __builtin_amdgcn_sad_u8(curw, sp_sample4(L, X, Y, ox & 3, oy & 3), 0u);
            acc[c8 >> 1] |= sad << (16 * (c8 & 1));
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            acc[q] = (unsigned)wave64_sum((int)acc[q]);
        }
#pragma unroll
        for (int c8 = 0; c8 < 8; c8++) {
            const int k = c8 < 4 ? c8 : c8 + 1;
            const int qx = cqx + (k % 3 - 1) * step, qy = cqy + (k / 3 - 1) * step;
            const unsigned sad = (acc[c8 >> 1] >> (16 * (c8 & 1))) & 0xFFFFu;
            const unsigned cost = sad + (unsigned)(lambda * (mvq_bits(qx) + mvq_bits(qy)));
            if (cost < best) { best = cost; bqx = qx; bqy = qy; }
        }
    }
    if (lane == 0 && ok) {
        mb_info_t *mb = &ctx->mbi[mbn];
        stg32(&mb->mvx, ((unsigned)(uint16_t)bqx) | ((unsigned)(uint16_t)bqy << 16));
        stg32(&mb->cost, best);
    }
}

// =================================================================== P macroblocks, fused: refinement + prediction + residual
// One wave = one macroblock (the shape of subpel_kernel, whose first half this repeats): after the refinement the
// half-sample planes are still in LDS, so the luma prediction of the winning vector is one sp_sample4() per lane -- the
// lane's four pixels of row lane>>2 -- instead of inter_kernel's 81 byte loads and per-pixel case analysis per 4x4 block.
// From there a 4x4 block lives on the four lanes that hold its rows (lane bits 3:2 = row in block, 1:0 = block column,
// 5:4 = block row): the row transforms are in-lane, the column transforms two-stage DPP butterflies inside the 16-lane row
// (as in the Intra4x4 path: coefficients stay in the lane order 0 2 1 3 and are addressed by frequency), so all 64 lanes
// carry luma; chroma runs on 32 lanes in the same layout (16-lane row = the four blocks Cb0 Cb1 Cr0 Cr1 of one block row),
// reads the reference and the source as words, and the Cb lanes store interleaved 8-byte row segments after fetching
// their Cr partners' samples with one DPP move.  Used for every P picture without the 8x8 transform; the separate
// subpel_kernel / inter_kernel remain for the High-profile path and the single-stage entry points.
__global__ __launch_bounds__(256) void pmb_kernel(const frame_ctx_t cv, int mb0, int mb1, int refine) {
    const frame_ctx_t *__restrict__ ctx = &cv;
    __shared__ __attribute__((aligned(16))) sp_lds LD[4];
    const int mbw = ctx->mbw, mbh = ctx->mbh, stride = ctx->stride, W = mbw * 16, H = mbh * 16, qp = ctx->qp;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int mbn = mb0 + blockIdx.x * 4 + wave; // the launch covers macroblocks mb0 .. mb1-1
    const bool ok = mbn < mb1;             // wave-uniform
    if (!ok) mbn = mb1 - 1;
    const int my = mbn / mbw, mx = mbn - my * mbw, x0 = mx * 16, y0 = my * 16;
    sp_lds *L = &LD[wave];
    const mb_info_t info = ld_mbinfo(&ctx->mbi[mbn]);
    const int ix = x0 + (info.mvx >> 2), iy = y0 + (info.mvy >> 2); // integer winner (vector is a multiple of 4 here)
    const uint8_t *__restrict__ ref = ctx->ref_y;
    if (ix - 3 >= 0 && iy - 3 >= 0 && ((ix - 3) & ~3) + 28 <= W && iy + 19 < H) {
        const int a = (ix - 3) & 3;
        const uint8_t *base = ref + (size_t)(iy - 3) * stride + ((ix - 3) & ~3);
        for (int i = lane; i < 23 * 6; i += 64) {
            const int r = i / 6, d = i - r * 6;
            const unsigned w0 = ldg32(base + (size_t)r * stride + 4 * d), w1 = ldg32(base + (size_t)r * stride + 4 * d + 4);
            *(unsigned *)&L->G[r * SP_GS + 4 * d] = __builtin_amdgcn_alignbyte(w1, w0, (unsigned)a);
        }
    } else
        for (int i = lane; i < 23 * 23; i += 64) {
            int r = i / 23, c = i - r * 23;
            int yy = clip3(0, H - 1, iy - 3 + r), xx = clip3(0, W - 1, ix - 3 + c);
            L->G[r * SP_GS + c] = (uint8_t)ldg8(ref + (size_t)yy * stride + xx);
        }
    const int pr = lane >> 2, pc = (lane & 3) * 4; // luma: lane owns row pr, columns pc .. pc+3
    unsigned curw;
    {
        int sy = y0 + pr;
        sy = sy < ctx->vis_h ? sy : ctx->vis_h - 1;
        curw = ldg32(ctx->src_y + (size_t)sy * ctx->src_stride + x0 + pc);
    }
    WAVE_SYNC();
    int bqx = info.mvx, bqy = info.mvy;
    unsigned best = info.cost;
    if (refine) { // ---- as subpel_kernel
        sp_planes(L, lane);
        const int lambda = ctx->lambda;
#pragma unroll 1
        for (int step = 2; step >= 1; step--) {
            const int cqx = bqx, cqy = bqy;
            unsigned acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int c8 = 0; c8 < 8; c8++) {
                const int k = c8 < 4 ? c8 : c8 + 1;
                const int qx = cqx + (k % 3 - 1) * step, qy = cqy + (k / 3 - 1) * step;
                const int ox = qx - info.mvx, oy = qy - info.mvy;
                const unsigned sad = __builtin_amdgcn_sad_u8(curw, sp_sample4(L, 1 + (ox >> 2) + pc, 1 + (oy >> 2) + pr, ox & 3, oy & 3), 0u);
                acc[c8 >> 1] |= sad << (16 * (c8 & 1));
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                acc[q] = (unsigned)wave64_sum((int)acc[q]);
            }
#pragma unroll
            for (int c8 = 0; c8 < 8; c8++) {
                const int k = c8 < 4 ? c8 : c8 + 1;
                const int qx = cqx + (k % 3 - 1) * step, qy = cqy + (k / 3 - 1) * step;
                const unsigned sad = (acc[c8 >> 1] >> (16 * (c8 & 1))) & 0xFFFFu;
                const unsigned cost = sad + (unsigned)(lambda * (mvq_bits(qx) + mvq_bits(qy)));
                if (cost < best) { best = cost; bqx = qx; bqy = qy; }
            }
        }
    }
    const dev_tables *T = &g_tab;
    const int py = (lane >> 2) & 3, fy = ((py & 1) << 1) | (py >> 1);
    const col_bf cb = make_col_bf(py);
    const int kz0 = (int)((0xFEA9DB83C7426510ull >> (16 * fy)) & 0xFFFF); // zig-zag positions of raster 4 fy + 0 .. 3, a nibble each
    int16_t *lv = ctx->levels + (size_t)mbn * MB_LEVELS;
    unsigned nz_luma = 0;
    { // ---- luma: prediction of the final vector, residual, transform, quantisation, reconstruction
        const int ox = bqx - info.mvx, oy = bqy - info.mvy;
        const unsigned pw = sp_sample4(L, 1 + (ox >> 2) + pc, 1 + (oy >> 2) + pr, ox & 3, oy & 3);
        const qparams q = make_q(T, qp, false);
        const int mfe = py < 2 ? q.mf[0] : q.mf[2], mfo = py < 2 ? q.mf[2] : q.mf[1], ve = py < 2 ? q.v[0] : q.v[2], vo = py < 2 ? q.v[2] : q.v[1];
        int x[4], lev[4];
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = byte_of(curw, i) - byte_of(pw, i);
        fwd_rows4(x);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int cf = fwd_col(x[i], cb);
            lev[i] = quant1(cf, (i & 1) ? mfo : mfe, q.f, q.qbits);
            x[i] = (lev[i] * ((i & 1) ? vo : ve)) << q.shift;
        }
        const int bx = lane & 3, by = lane >> 4, b = ((by >> 1) << 3) | ((bx >> 1) << 2) | ((by & 1) << 1) | (bx & 1);
        if (ok) {
#pragma unroll
            for (int i = 0; i < 4; i++) stg16(&lv[L_LUMA + b * 16 + ((kz0 >> (4 * i)) & 15)], lev[i]);
        }
        inv_rows4(x);
        int o[4];
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = clip255(byte_of(pw, i) + ((inv_col(x[i], cb) + 32) >> 6));
        if (ok) stg32(ctx->rec_y + (size_t)(y0 + pr) * stride + x0 + pc, pack4(o[0], o[1], o[2], o[3]));
        // non-zero blocks: OR over the block's four lanes, then into luma4x4BlkIdx order
        const unsigned long long bal = __ballot((lev[0] | lev[1] | lev[2] | lev[3]) != 0);
        const unsigned long long t = bal | (bal >> 4) | (bal >> 8) | (bal >> 12); // bit 16 by + bx
        const int rb = lane & 15, rbx = blkx(rb) >> 2, rby = blky(rb) >> 2;
        nz_luma = (unsigned)(__ballot(lane < 16 && ((t >> (16 * rby + rbx)) & 1)) & 0xFFFFull);
    }
    unsigned nz_c = 0, dc_c = 0;
    { // ---- chroma on lanes 0..31: bit 4 = block row, bits 3:2 = row in block, bit 1 = plane, bit 0 = block column
        const int cby = (lane >> 4) & 1, c = (lane >> 1) & 1, cbx = lane & 1;
        const int cx0 = x0 >> 1, cy0 = y0 >> 1, cw = W >> 1, ch = H >> 1;
        const int cy = cby * 4 + py, cxb = cbx * 4;
        // 8.4.1.4 / 8.4.2.2.2: the chroma vector is the luma vector read in 1/8 chroma-sample units
        const int xi = bqx >> 3, yi = bqy >> 3, xf = bqx & 7, yf = bqy & 7;
        const uint8_t *__restrict__ rf = ctx->ref_uv;
        int A[5], B[5];
        if (cx0 + xi >= 0 && cx0 + xi + 9 <= cw && cy0 + yi >= 0 && cy0 + yi + 9 <= ch) { // whole 9 x 9 neighbourhood inside (wave-uniform)
            const int o = 2 * (cx0 + cxb + xi) + c, a = o & 3;
            const uint8_t *r0 = rf + (size_t)(cy0 + cy + yi) * stride + (o & ~3), *r1 = r0 + stride;
            const unsigned a0 = ldg32(r0), a1 = ldg32(r0 + 4), a2 = ldg32(r0 + 8), b0 = ldg32(r1), b1 = ldg32(r1 + 4), b2 = ldg32(r1 + 8);
            const unsigned sa0 = __builtin_amdgcn_alignbyte(a1, a0, (unsigned)a), sa1 = __builtin_amdgcn_alignbyte(a2, a1, (unsigned)a);
            const unsigned sb0 = __builtin_amdgcn_alignbyte(b1, b0, (unsigned)a), sb1 = __builtin_amdgcn_alignbyte(b2, b1, (unsigned)a);
            A[0] = byte_of(sa0, 0); A[1] = byte_of(sa0, 2); A[2] = byte_of(sa1, 0); A[3] = byte_of(sa1, 2); A[4] = (int)((a2 >> (8 * a)) & 255);
            B[0] = byte_of(sb0, 0); B[1] = byte_of(sb0, 2); B[2] = byte_of(sb1, 0); B[3] = byte_of(sb1, 2); B[4] = (int)((b2 >> (8 * a)) & 255);
        } else {
            const int ya = clip3(0, ch - 1, cy0 + cy + yi), yb = clip3(0, ch - 1, cy0 + cy + yi + 1);
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const int xx = clip3(0, cw - 1, cx0 + cxb + i + xi);
                A[i] = (int)ldg8(rf + (size_t)ya * stride + 2 * xx + c);
                B[i] = (int)ldg8(rf + (size_t)yb * stride + 2 * xx + c);
            }
        }
        int sy = cy0 + cy;
        const int vh2 = ctx->vis_h >> 1;
        sy = sy < vh2 ? sy : vh2 - 1;
        const uint2 sw = ldg64(ctx->src_uv + (size_t)sy * ctx->src_stride + 2 * (cx0 + cxb));
        const unsigned slo = c ? (sw.x >> 8) : sw.x, shi = c ? (sw.y >> 8) : sw.y;
        const int sv[4] = {(int)(slo & 255), (int)((slo >> 16) & 255), (int)(shi & 255), (int)((shi >> 16) & 255)};
        const int w00 = (8 - xf) * (8 - yf), w10 = xf * (8 - yf), w01 = (8 - xf) * yf, w11 = xf * yf;
        int pd[4];
#pragma unroll
        for (int i = 0; i < 4; i++) pd[i] = (w00 * A[i] + w10 * A[i + 1] + w01 * B[i] + w11 * B[i + 1] + 32) >> 6;
        chroma_rows4(ctx, T, lv, cx0, cy0, lane, pd, sv, qp, false, ok, nullptr, nz_c, dc_c);
    }
    if (ok && lane == 0) {
        unsigned nzm = nz_luma | (nz_c << 16);
        if (dc_c & 1) nzm |= NZ_CBDC;
        if (dc_c & 2) nzm |= NZ_CRDC;
        mb_info_t *mb = &ctx->mbi[mbn];
        stg32(&mb->mvx, ((unsigned)(uint16_t)bqx) | ((unsigned)(uint16_t)bqy << 16));
        stg32(&mb->cost, best);
        stg32(&mb->mb_type, 1u | ((unsigned)qp << 24)); // mb_type 1, modes 0, qp
        stg32(&mb->nzmask, nzm);
    }
    if (ok && lane < 2) stg128(lv + L_LDC + 8 * lane, make_uint4(0, 0, 0, 0)); // luma DC levels: unused by P macroblocks, kept zero
}

// =================================================================== launchers
// The three P-picture kernels take a macroblock-row range [row0, row1): the host overlaps the upper part of picture n+1
// with the tail of picture n's deblocking (mi355enc.cpp, enqueue_picture).
void k_launch_me(const frame_ctx_t *h_ctx, int mbw, int row0, int row1, hipStream_t s) {
    int strips = (mbw + ME_MBS - 1) / ME_MBS;
    if (row1 > row0) hipLaunchKernelGGL(me_kernel, dim3(strips * (row1 - row0)), dim3(64 * ME_MBS), 0, s, *h_ctx, row0);
}
void k_launch_subpel(const frame_ctx_t *h_ctx, int mbw, int row0, int row1, hipStream_t s) {
    if (row1 > row0) hipLaunchKernelGGL(subpel_kernel, dim3((mbw * (row1 - row0) + 3) / 4), dim3(256), 0, s, *h_ctx, row0 * mbw, row1 * mbw);
}
void k_launch_pmb(const frame_ctx_t *h_ctx, int mbw, int row0, int row1, int refine, hipStream_t s) {
    int n = mbw * (row1 - row0);
    if (n > 0) hipLaunchKernelGGL(pmb_kernel, dim3((n + 3) / 4), dim3(256), 0, s, *h_ctx, row0 * mbw, row1 * mbw, refine);
}
