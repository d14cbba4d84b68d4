// k_motion.hip -- full-search motion estimation and sub-sample refinement
// Hand-written HIP for gfx950 (CDNA4, wave64); part of libmi355enc (see kernels_common.hpp).
#include "kernels_common.hpp"

// =================================================================== motion search
// One workgroup = ME_MBS horizontally adjacent macroblocks of one macroblock row, one wave per
// macroblock.  The 48 x (16*ME_MBS+32) luma search window (+-16 around the strip) is staged once in
// LDS, the reference extended beyond the picture by coordinate clamping (8.4.2.2: vectors may leave
// the picture).  Lane l < 63 of a wave owns the candidates
//   dy in [-16 + 5*(l/9), +5)   x   dx in [-16 + 4*(l%9), +4)
// (7 x 9 tiles cover 35 x 36 >= 33 x 33; 85 % of the computed SADs are real candidates) and
// accumulates them with v_qsad_pk_u16_u8 -- four 4-pixel SADs per instruction -- re-using each
// window row for the 5 dy it serves.  Every SAD goes to the macroblock's surface in HBM (SURF_U16
// uint16: the accumulators are four packed uint16 already, so a lane stores its tile as five 8-byte
// words), and a first vector selection is made with the bits charged against zero; me_select_kernel
// re-selects from the surfaces against the neighbours' choices (oracle: orc_me_frame / orc_me_select).
#ifndef ME_MBS
#define ME_MBS 4
#endif
#define ME_WQ (ME_MBS + 2)      /* uint4 per window row: 16*ME_MBS + 32 bytes */
#define ME_ROWS 50   /* 48 real rows + 2 that only out-of-range candidates (dy = 17, 18) ever touch */
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
DEV int med3(int a, int b, int c) {
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    return c < lo ? lo : (c > hi ? hi : c);
}
// 8.4.1.3 on a whole-sample vector field (every neighbour taken as inter, refIdx 0) and the 8.4.1.1 P_Skip inference on the
// same field; quarter-sample units.  Wave-uniform inputs -> wave-uniform result.  Oracle: field_pred.
struct fpred_t { int px, py, sx, sy; };
// top: the row above is available (row_has_top: not the picture's first row, not another slice's).
DEV fpred_t field_pred(const imv_t *__restrict__ f, int mbw, int mx, int my, bool top) {
    const bool avA = mx > 0, avB = top, avC = top && mx + 1 < mbw, avD = mx > 0 && top, hasC = avC || avD;
    const int self = my * mbw + mx;
    const unsigned wa = avA ? ldg32(f + self - 1) : 0u, wb = avB ? ldg32(f + self - mbw) : 0u;
    const unsigned wc = avC ? ldg32(f + self - mbw + 1) : (avD ? ldg32(f + self - mbw - 1) : 0u);
    const int ax = (int)(int16_t)(wa & 0xFFFF), ay = (int)(int16_t)(wa >> 16), bx = (int)(int16_t)(wb & 0xFFFF), by = (int)(int16_t)(wb >> 16);
    const int cx = (int)(int16_t)(wc & 0xFFFF), cy = (int)(int16_t)(wc >> 16);
    const int n = (avA ? 1 : 0) + (avB ? 1 : 0) + (hasC ? 1 : 0);
    fpred_t r;
    if (n == 1) { r.px = avA ? ax : avB ? bx : cx; r.py = avA ? ay : avB ? by : cy; }
    else { r.px = med3(ax, bx, cx); r.py = med3(ay, by, cy); }
    const bool zero = !avA || !avB || wa == 0u || wb == 0u;
    r.sx = zero ? 0 : r.px; r.sy = zero ? 0 : r.py;
    return r;
}
// The selection over one lane's 5 x 4 tile of SADs (acc[d]: four packed uint16, dx ascending) and the wave-wide minimum:
// key = cost << 12 | (dy+16) << 6 | (dx+16), cost = SAD + lambda * (bits(dx - px) + bits(dy - py) + SEL_BONUS) -- the candidate
// equal to the skip inference (sx, sy) is charged nothing; ties resolve to the first candidate in (dy, dx) raster order.
// px .. sy in whole samples.
// ZERO (the search's own first selection: all four predictors are zero): the skip candidate is handled once, by the one lane that holds (0, 0), behind the loop --
// charged nothing it can only lower that candidate's key, so the minimum is the one the replacement inside the loop gives.
template <bool ZERO = false>
DEV unsigned select_min(const unsigned long long *acc, int g, int dxg, bool active, int R, int lambda, int px, int py, int sx, int sy) {
    const unsigned INVALID = 0x40000000u;
    unsigned bo[4];
#pragma unroll
    for (int o = 0; o < 4; o++) {
        const int dx = -16 + 4 * dxg + o;
        bo[o] = (dx >= -R && dx <= R && active) ? (((unsigned)(lambda * mv_bits(dx - px)) << 12) | (unsigned)(dx + 16)) : INVALID;
    }
    const int so = sx + 16 - 4 * dxg; // which of this lane's four columns is the skip candidate's (0..3), if any
    unsigned best = 0xFFFFFFFFu;
#pragma unroll
    for (int d = 0; d < ME_K; d++) {
        const int dy = -16 + ME_K * g + d;
        const unsigned pos = (unsigned)(dy + 16) << 6;
        const unsigned bd = (dy >= -R && dy <= R) ? (((unsigned)(lambda * (mv_bits(dy - py) + SEL_BONUS)) << 12) | pos) : INVALID;
        const unsigned lo = (unsigned)acc[d], hi = (unsigned)(acc[d] >> 32);
        unsigned s[4] = {(lo << 16) >> 4, (lo & 0xFFFF0000u) >> 4, (hi << 16) >> 4, (hi & 0xFFFF0000u) >> 4}; // SAD << 12
        unsigned k[4];
#pragma unroll
        for (int o = 0; o < 4; o++) {
            const bool is_skip = !ZERO && dy == sy && so == o && active; // in range by construction: a median of in-range vectors
            k[o] = is_skip ? (s[o] | pos | (unsigned)(sx + 16)) : s[o] + bd + bo[o];
        }
        unsigned ka = k[0] < k[1] ? k[0] : k[1], kb = k[2] < k[3] ? k[2] : k[3];
        ka = ka < kb ? ka : kb;
        best = best < ka ? best : ka;
    }
    if (ZERO && active && ME_K * g <= 16 && 16 < ME_K * g + ME_K && dxg == 4) { // dy = -16 + ME_K g + d = 0, dx = -16 + 4 dxg + 0 = 0
        const unsigned long long a = acc[16 - ME_K * (16 / ME_K)];
        const unsigned k0 = (((unsigned)a << 16) >> 4) | (16u << 6) | 16u;
        best = best < k0 ? best : k0;
    }
    return wave64_umin(best);
}
DEV void store_imv(imv_t *dst, unsigned best, int lambda, int px, int py, int sx, int sy) {
    const int bx = (int)(best & 63) - 16, by = (int)((best >> 6) & 63) - 16;
    const unsigned bits = (bx == sx && by == sy) ? 0u : (unsigned)(mv_bits(bx - px) + mv_bits(by - py) + SEL_BONUS);
    const unsigned sad = (best >> 12) - (unsigned)lambda * bits;
    stg64(dst, make_uint2(((unsigned)(uint16_t)(4 * bx)) | ((unsigned)(uint16_t)(4 * by) << 16), sad | (bits << 16))); // quarter-sample units
}

// (r04 A/B: capping the resident waves per SIMD at 6 / 4 / 2 -- so that workgroups start staggered and the staging and surface stores of some run beside the SAD loop of
// others -- gave 34.9 / 36.3 / 40.6 us alone against 34.8, and -2 ... +3 % frames/s in the stream, inside the noise: the loop is at the issue ceiling either way.)
__global__ __launch_bounds__(64 * ME_MBS) void me_kernel(const frame_ctx_t cv, int row0) { // context by value: lives in the kernarg segment, no per-picture upload
    const frame_ctx_t *__restrict__ ctx = &cv;
    if (blockIdx.x == 0) tl_first(ctx, 0);
    __shared__ unsigned win[ME_ROWS * ME_STRIDE];
    const int stride = ctx->stride, mbw = ctx->mbw, mbh = ctx->mbh;
    const int W = mbw * 16, H = mbh * 16;
    const int strips = (mbw + ME_MBS - 1) / ME_MBS;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int my = row0 + tile / strips, sx = tile % strips; // the launch covers macroblock rows row0 .. row0 + gridDim.x / strips - 1
    const int t = threadIdx.x;
    const uint8_t *__restrict__ ref = ctx->me_ref_y; // source against source: the padded source luma of the last coded picture (oracle: orc_enc_frame2)

    // ---- stage the window: 48 rows x 10 uint4 (coalesced 16 B per lane); rows / 16-byte groups beyond the picture repeat its edge
    for (int i = t; i < 48 * ME_WQ; i += 64 * ME_MBS) {
        int row = i / ME_WQ, q = i - row * ME_WQ;
        const int gy = clip3(0, H - 1, my * 16 - 16 + row), gx = sx * (ME_MBS * 16) - 16 + 16 * q;
        uint4 v;
#ifdef ME_DBG_NOSTAGE
        v = make_uint4(i, i, i, i);
        if (gx != -12345) {}
#else
        if (gx >= 0 && gx < W) v = ldg128(ref + (size_t)gy * stride + gx);
#endif
        else { const unsigned e = ldg8(ref + (size_t)gy * stride + (gx < 0 ? 0 : W - 1)) * 0x01010101u; v = make_uint4(e, e, e, e); }
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
        // wave-uniform, so they live in SGPRs (v_qsad takes one scalar source): 64 VGPRs less, 8 waves per SIMD instead of 5 -- every workgroup of a
        // 1080p picture is resident at once (36.1 -> 35.2 us alone, 2160p 130 -> 123)
        // ... and arrive there through the scalar cache: 16 s_load_dwordx4 instead of 16 vector loads of one address and 64 v_readfirstlane (r04: 118 -> ... us at 2160p)
        const int mu = __builtin_amdgcn_readfirstlane(mxc);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            int sy = my * 16 + r;
            sy = sy < vh ? sy : vh - 1;
#ifdef ME_DBG_NOCUR
            const uint4 v = make_uint4(sy, mu, r, sy ^ mu);
#else
            const uint4 v = ldc128(src + (size_t)sy * ss + mu * 16);
#endif
            c[r][0] = __builtin_amdgcn_readfirstlane(v.x); c[r][1] = __builtin_amdgcn_readfirstlane(v.y);
            c[r][2] = __builtin_amdgcn_readfirstlane(v.z); c[r][3] = __builtin_amdgcn_readfirstlane(v.w);
        }
        // (behind the scalar loads: in front of them the compiler merges their address terms with this branch's through VGPRs and the loads turn into vector loads)
        if (lane < 16 && mx < mbw) { // the next picture searches against this: lane r copies row r
            int sy = my * 16 + lane;
            sy = sy < vh ? sy : vh - 1;
            stg128(ctx->psrc_out + (size_t)(my * 16 + lane) * stride + mu * 16, ldg128(src + (size_t)sy * ss + mu * 16));
        }
    }
    __syncthreads();

    unsigned long long acc[ME_K];
#pragma unroll
    for (int d = 0; d < ME_K; d++) acc[d] = 0;
    const unsigned *wp = &win[(ME_K * g) * ME_STRIDE + 4 * m + dxg];
    // one window row ahead in registers; sched_barrier keeps the compiler from hoisting all 100 LDS
    // reads to the top (which costs > 200 VGPRs and halves the occupancy)
#ifdef ME_DBG_REPEAT
    for (int rep = 0; rep < ME_DBG_REPEAT; rep++) {
#endif
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
#ifdef ME_DBG_REPEAT
    }
#endif
    // ---- the surface: this lane's tile, five 8-byte words
#ifdef ME_DBG_NOSURF
    if (active && acc[0] == 0x123456789ull) {
#else
    if (active) {
#endif
        uint16_t *sf = ctx->surf + (size_t)(my * mbw + mx) * SURF_U16 + 4 * dxg;
#pragma unroll
        for (int d = 0; d < ME_K; d++) stg64(sf + (ME_K * g + d) * SURF_COLS, make_uint2((unsigned)acc[d], (unsigned)(acc[d] >> 32)));
    }
    // ---- first selection: bits against the zero vector
#ifdef ME_DBG_NOSEL
    const unsigned best = ((((unsigned)acc[0] ^ (unsigned)acc[4]) & 0x0FFFu) << 12) | 0x410u; // (the zero vector: nothing downstream leaves its window)
#else
    const unsigned best = select_min<true>(acc, g, dxg, active, ctx->me_range, ctx->lambda, 0, 0, 0, 0);
#endif
    if (lane == 0 && mx < mbw) store_imv(&ctx->imv_a[my * mbw + mx], best, ctx->lambda, 0, 0, 0, 0);
    tl_last(ctx, 1);
}

// One Jacobi iteration of the selection: one wave per macroblock re-reads its surface (each lane the tile it wrote) and
// selects against the 8.4.1.3 median / 8.4.1.1 skip inference of the field `in`.  HBM-shaped: 2520 bytes per macroblock.
// An iteration changes a macroblock's result only if its predictors changed: the result is a function of the surface and of
// (px, py, sx, sy) alone, so where the field `prev` the previous iteration read gives the same four numbers (mode 2; mode 1: the
// search's own selection, which used zeros) the previous result is copied and the surface is not read (on the S2 clip 9 of 10
// macroblocks from the second iteration on).  mode 0: always recompute.
// (r04 A/B: asking for the surface before the predictors -- one memory round trip a wave instead of two, the loads wasted where the macroblock copies -- gave 0.945 x at 1080p
// and 0.99 x at 2160p, alternating processes: not kept.)
__global__ __launch_bounds__(256) void me_select_kernel(const frame_ctx_t cv, int mb0, int mb1, const imv_t *__restrict__ in, imv_t *__restrict__ out,
                                                        const imv_t *__restrict__ prev, int mode) {
    const frame_ctx_t *__restrict__ ctx = &cv;
    const int mbw = ctx->mbw;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // an SGPR: what is derived from it is scalar control flow
    const int mbn = mb0 + xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
    if (mbn >= mb1) return; // wave-uniform; no workgroup barrier below
    const int my = mbn / mbw, mx = mbn - my * mbw;
    const fpred_t fp = field_pred(in, mbw, mx, my, row_has_top(ctx, my));
    const int px = fp.px >> 2, py = fp.py >> 2, sx = fp.sx >> 2, sy = fp.sy >> 2;
    if (mode) {
        fpred_t fq = {0, 0, 0, 0};
        if (mode == 2) fq = field_pred(prev, mbw, mx, my, row_has_top(ctx, my));
        if (px == (fq.px >> 2) && py == (fq.py >> 2) && sx == (fq.sx >> 2) && sy == (fq.sy >> 2)) { // wave-uniform
            if (lane == 0) stg64(&out[mbn], ldg64(&in[mbn]));
            return;
        }
    }
    const bool active = lane < 63;
    const int g = active ? lane / 9 : 0, dxg = active ? lane % 9 : 0;
    unsigned long long acc[ME_K];
    const uint16_t *sf = ctx->surf + (size_t)mbn * SURF_U16 + 4 * dxg;
#pragma unroll
    for (int d = 0; d < ME_K; d++) { const uint2 v = ldg64(sf + (ME_K * g + d) * SURF_COLS); acc[d] = ((unsigned long long)v.y << 32) | v.x; }
    const unsigned best = select_min(acc, g, dxg, active, ctx->me_range, ctx->lambda, px, py, sx, sy);
    if (lane == 0) store_imv(&out[mbn], best, ctx->lambda, px, py, sx, sy);
}

// The same iteration for the later passes (mode 2), where nine macroblocks in ten only copy their previous result: one wave per macroblock then spends a launch of
// 8 160 waves (32 400 at 2160p) on reading predictors.  Here a wave takes SEL_SPW consecutive macroblocks: lanes 0 .. SEL_SPW-1 each check one (both fields' predictors,
// per-lane addresses), copy the unchanged ones, and the wave then walks the changed ones with all 64 lanes on the surface as above.  Same result bit for bit (the test
// is the same, the selection is the same function); an eighth of the waves, and almost none of them long.
#ifndef SEL_SPW /* (A/B on hardware, alternating processes: 4 per wave 0.99 x, 16 per wave 1.06 x in one noisy pair at 1080p and 1.00 x at 2160p: 8 stays) */
#define SEL_SPW 8
#endif
__global__ __launch_bounds__(256) void me_select_sparse_kernel(const frame_ctx_t cv, int mb0, int mb1, const imv_t *__restrict__ in, imv_t *__restrict__ out, const imv_t *__restrict__ prev) {
    const frame_ctx_t *__restrict__ ctx = &cv;
    const int mbw = ctx->mbw;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int base = mb0 + ((int)blockIdx.x * 4 + wave) * SEL_SPW;
    if (base >= mb1) return; // wave-uniform; no workgroup barrier below
    const int mine = base + (lane < SEL_SPW ? lane : 0);
    const bool have = lane < SEL_SPW && mine < mb1;
    const int mbn_l = have ? mine : base, my_l = mbn_l / mbw, mx_l = mbn_l - my_l * mbw;
    const bool top_l = row_has_top(ctx, my_l);
    const fpred_t fp = field_pred(in, mbw, mx_l, my_l, top_l), fq = field_pred(prev, mbw, mx_l, my_l, top_l);
    const int px = fp.px >> 2, py = fp.py >> 2, sx = fp.sx >> 2, sy = fp.sy >> 2;
    const bool same = px == (fq.px >> 2) && py == (fq.py >> 2) && sx == (fq.sx >> 2) && sy == (fq.sy >> 2);
    if (have && same) stg64(&out[mbn_l], ldg64(&in[mbn_l]));
    unsigned todo = (unsigned)(__ballot(have && !same) & ((1ull << SEL_SPW) - 1ull));
    const bool active = lane < 63;
    const int g = active ? lane / 9 : 0, dxg = active ? lane % 9 : 0;
    while (todo) { // wave-uniform
        const int j = __builtin_ctz(todo);
        todo &= todo - 1;
        const int mbn = base + j;
        const int jpx = __builtin_amdgcn_readlane(px, j), jpy = __builtin_amdgcn_readlane(py, j), jsx = __builtin_amdgcn_readlane(sx, j), jsy = __builtin_amdgcn_readlane(sy, j);
        unsigned long long acc[ME_K];
        const uint16_t *sf = ctx->surf + (size_t)mbn * SURF_U16 + 4 * dxg;
#pragma unroll
        for (int d = 0; d < ME_K; d++) { const uint2 v = ldg64(sf + (ME_K * g + d) * SURF_COLS); acc[d] = ((unsigned long long)v.y << 32) | v.x; }
        const unsigned best = select_min(acc, g, dxg, active, ctx->me_range, ctx->lambda, jpx, jpy, jsx, jsy);
        if (lane == 0) store_imv(&out[mbn], best, ctx->lambda, jpx, jpy, jsx, jsy);
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
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // an SGPR: what is derived from it is scalar control flow
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

// =================================================================== P macroblocks, fused stage
// One wave = one macroblock; everything a P macroblock needs after the vector field is settled (oracle: orc_pmb_frame):
//   1. predictor estimates p_est / ps_est from the whole-sample field (field_pred);
//   2. skip probe at ps_est -- worth running only when its SAD (already on the surface) is within lambda * SKIP_MARGIN_BITS
//      of the best whole-sample SAD: prediction straight from the reference picture, residual through the transform and
//      the quantiser with coefficient decimation; if nothing is left (luma, chroma AC and DC) the macroblock takes ps_est
//      without residual and the wave is done.  Rate control's ladder below QP 51 (ctx->drop_sad) passes the probe on the
//      SAD alone;
//   3. sub-sample refinement around the whole-sample winner (planes in LDS as in subpel_kernel), bits against p_est:
//      half-sample round by SAD, quarter-sample round by SATD (4x4 Hadamard on the four-lanes-per-block layout);
//   4. intra instead, when the macroblock was analysed (search cost >= INTRA_GATE) and the open-loop intra cost wins:
//      only type and modes go to the record, intra_p_kernel reconstructs;
//   5. residual of the final vector: a 4x4 block lives on the four lanes that hold its rows (lane bits 3:2 = row in
//      block, 1:0 = block column, 5:4 = block row), row transforms in-lane, column transforms as DPP butterflies;
//      decimation scores from the blocks' 16-bit significance masks (no run loops); chroma on 32 lanes (chroma_rows4).
DEV int dec_score(unsigned M) { return dec_score_mask(M); }
DEV int row_or4(int v) { // OR over the four lanes l, l+4, l+8, l+12 of a 16-lane row (the four rows of a 4x4 block)
    v |= __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, false); // row_ror:8
    v |= __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, false); // row_ror:4
    return v;
}
// four luma samples at whole-sample position (x .. x+3, y) of the reference, one per byte; `inside`: the macroblock's whole
// 16 x 16 block lies in the picture (wave-uniform), otherwise coordinates clamp (8.4.2.2)
DEV unsigned ref_word4(const uint8_t *__restrict__ ref, int stride, int W, int H, int x, int y, bool inside) {
    if (inside) {
        const uint8_t *b = ref + (size_t)y * stride + (x & ~3);
        return __builtin_amdgcn_alignbyte(ldg32(b + 4), ldg32(b), (unsigned)(x & 3)); // may touch 4 bytes past the row: SURF_PAD
    }
    const uint8_t *r = ref + (size_t)clip3(0, H - 1, y) * stride;
    return pack4((int)ldg8(r + clip3(0, W - 1, x)), (int)ldg8(r + clip3(0, W - 1, x + 1)), (int)ldg8(r + clip3(0, W - 1, x + 2)), (int)ldg8(r + clip3(0, W - 1, x + 3)));
}
// chroma prediction (8.4.2.2.2) of this lane's four samples for the quarter-sample luma vector (mvx, mvy); lanes 0..31:
// bit 4 = block row, bits 3:2 = row in block, bit 1 = plane, bit 0 = block column
DEV void chroma_pred4(const frame_ctx_t *__restrict__ ctx, int lane, int x0, int y0, int W, int H, int mvx, int mvy, int *pd) {
    const int stride = ctx->stride;
    const int py = (lane >> 2) & 3, cby = (lane >> 4) & 1, c = (lane >> 1) & 1, cbx = lane & 1;
    const int cx0 = x0 >> 1, cy0 = y0 >> 1, cw = W >> 1, ch = H >> 1;
    const int cy = cby * 4 + py, cxb = cbx * 4;
    const int xi = mvx >> 3, yi = mvy >> 3, xf = mvx & 7, yf = mvy & 7;
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
    const int w00 = (8 - xf) * (8 - yf), w10 = xf * (8 - yf), w01 = (8 - xf) * yf, w11 = xf * yf;
#pragma unroll
    for (int i = 0; i < 4; i++) pd[i] = (w00 * A[i] + w10 * A[i + 1] + w01 * B[i] + w11 * B[i + 1] + 32) >> 6;
}
// Luma of an inter macroblock on all 64 lanes: residual of the lane's source word against its prediction word, transform,
// quantiser, decimation.  lev[4]: this lane's levels (frequency row fy of its block) after decimation; x[4]: the dequantised
// coefficients, ready for inv_rows4 / inv_col.  Returns the blkIdx-order mask of blocks that keep levels (wave-uniform).
DEV unsigned pmb_luma_tq(const dev_tables *T, int lane, unsigned curw, unsigned pw, int qp, bool decimate, int *lev, int *x) {
    const int py = (lane >> 2) & 3, fy = ((py & 1) << 1) | (py >> 1);
    const col_bf cb = make_col_bf(py);
    const int kz0 = (int)((0xFEA9DB83C7426510ull >> (16 * fy)) & 0xFFFF); // zig-zag positions of raster 4 fy + 0 .. 3, a nibble each
    const qparams q = make_q(T, qp, false);
    const int mfe = py < 2 ? q.mf[0] : q.mf[2], mfo = py < 2 ? q.mf[2] : q.mf[1], ve = py < 2 ? q.v[0] : q.v[2], vo = py < 2 ? q.v[2] : q.v[1];
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = byte_of(curw, i) - byte_of(pw, i);
    fwd_rows4(x);
    unsigned m = 0, big = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int cf = fwd_col(x[i], cb);
        lev[i] = quant1(cf, (i & 1) ? mfo : mfe, q.f, q.qbits);
        m |= (lev[i] != 0 ? 1u : 0u) << ((kz0 >> (4 * i)) & 15);
        big |= (lev[i] > 1 || lev[i] < -1) ? 1u : 0u;
    }
    if (decimate) {
        const unsigned mm = (unsigned)row_or4((int)(m | (big << 16))); // the block's 16 significance bits + "has a level beyond +-1"
        int s = (mm >> 16) ? 9 : dec_score(((mm & 0xFFFFu) << 1) | 1u);
        int s8 = s + quad_xor<1>(s);            // the 8x8 block: block columns 2k, 2k+1 ...
        s8 += __shfl_xor(s8, 16, 64);           // ... and block rows 2k, 2k+1
        int tot = s8 + quad_xor<2>(s8);
        tot += __shfl_xor(tot, 32, 64);
        if (s8 < 4 || tot < 6) { lev[0] = lev[1] = lev[2] = lev[3] = 0; }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = (lev[i] * ((i & 1) ? vo : ve)) << q.shift;
    // non-zero blocks: OR over the block's four lanes, then into luma4x4BlkIdx order
    const unsigned long long bal = __ballot((lev[0] | lev[1] | lev[2] | lev[3]) != 0);
    const unsigned long long t = bal | (bal >> 4) | (bal >> 8) | (bal >> 12); // bit 16 by + bx
    const int rb = lane & 15, rbx = blkx(rb) >> 2, rby = blky(rb) >> 2;
    return (unsigned)(__ballot(lane < 16 && ((t >> (16 * rby + rbx)) & 1)) & 0xFFFFull);
}
// sum of |H d H^T| over the macroblock's sixteen 4x4 blocks for prediction word pw (unhalved; wave-uniform result)
DEV unsigned pmb_satd(int lane, unsigned curw, unsigned pw) {
    const int py = (lane >> 2) & 3, s1 = py < 2 ? 1 : -1, s2 = (py & 1) ? -1 : 1;
    const int d0 = byte_of(curw, 0) - byte_of(pw, 0), d1 = byte_of(curw, 1) - byte_of(pw, 1), d2 = byte_of(curw, 2) - byte_of(pw, 2), d3 = byte_of(curw, 3) - byte_of(pw, 3);
    const int a = d0 + d3, b = d1 + d2, c = d1 - d2, e = d0 - d3;
    int t[4] = {a + b, e + c, a - b, e - c};
    int acc = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int u = mad24(t[i], s1, row_xor8(t[i])); // rows py and py ^ 2
        const int w = mad24(u, s2, row_xor4(u));       // ... and py ^ 1
        acc += iabs(w);
    }
    return (unsigned)wave64_sum(acc);
}
// write a macroblock that carries no residual: the prediction is the reconstruction, every level is zero
template <bool SC1>
DEV void pmb_store_pred_only(const frame_ctx_t *__restrict__ ctx, int16_t *lv, int lane, int x0, int y0, unsigned pw, const int *pd) {
    const int stride = ctx->stride, pr = lane >> 2, pc = (lane & 3) * 4;
    stx32<SC1>(ctx->rec_y + (size_t)(y0 + pr) * stride + x0 + pc, pw);
    const int py = (lane >> 2) & 3, cby = (lane >> 4) & 1, c = (lane >> 1) & 1, cbx = lane & 1;
    const unsigned mine = pack4(pd[0], pd[1], pd[2], pd[3]), other = (unsigned)quad_xor<2>((int)mine);
    if (lane < 32 && c == 0) {
        uint2 out;
        out.x = __builtin_amdgcn_perm(other, mine, 0x05010400u); // U0 V0 U1 V1
        out.y = __builtin_amdgcn_perm(other, mine, 0x07030602u); // U2 V2 U3 V3
        stx64<SC1>(ctx->rec_uv + (size_t)((y0 >> 1) + cby * 4 + py) * stride + 2 * ((x0 >> 1) + cbx * 4), out);
    }
    if (lane < MB_LEVELS * 2 / 16) stg128(lv + 8 * lane, make_uint4(0, 0, 0, 0)); // 816 bytes = 51 x 16
}

// One P macroblock on one wave (no workgroup barrier anywhere: the four waves of a workgroup are independent).
// SC1: the reconstruction and the record are stored through to memory -- the picture's own deblocking launch reads them without a kernel boundary in between.
// High profile: the luma residual of an inter macroblock through the 8x8 transform (oracle: tq8_block).  The wave's 256 residual samples go to an LDS tile
// [8x8 block][row][column]; sixteen lanes -- four per block, two rows / two columns each -- run the separable passes through it (k_inter.hip has the same
// passes for the two-kernel form); levels are stored de-interleaved the way CAVLC sends them (4x4 "block" 4 i8 + j holds scan positions 4 k + j).  Returns the
// blkIdx mask of the sub-blocks with levels; the reconstruction is stored by all lanes.  The tile overlays the refinement's planes (dead by now).
template <bool SC1>
DEV unsigned pmb_luma_t8(const frame_ctx_t *__restrict__ ctx, int *tile, int lane, unsigned curw, unsigned pw, int qp, int mbn, int x0, int y0) {
    const int pr = lane >> 2, pc = (lane & 3) * 4;
    {
        int *t = tile + ((pr >> 3) * 2 + (pc >> 3)) * 64 + (pr & 7) * 8 + (pc & 7);
#pragma unroll
        for (int i = 0; i < 4; i++) t[i] = byte_of(curw, i) - byte_of(pw, i);
    }
    WAVE_SYNC();
    const bool act = lane < 16;
    const int i8 = (lane >> 2) & 3, j = lane & 3;
    int *tb = tile + i8 * 64;
    const int m6 = qp % 6, k6 = qp / 6;
    int rs[2][8], cw[2][8];
    unsigned submask = 0;
    if (act) { // rows 2j, 2j+1: first forward pass
#pragma unroll
        for (int r = 0; r < 2; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) rs[r][i] = tb[(2 * j + r) * 8 + i];
            fdct8_1d(rs[r]);
        }
    }
    WAVE_SYNC();
    if (act) {
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) tb[(2 * j + r) * 8 + i] = rs[r][i];
    }
    WAVE_SYNC();
    if (act) { // columns 2j, 2j+1: second forward pass, quantise, scale
#pragma unroll
        for (int c2 = 0; c2 < 2; c2++) {
#pragma unroll
            for (int r = 0; r < 8; r++) cw[c2][r] = tb[r * 8 + 2 * j + c2];
            fdct8_1d(cw[c2]);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int xx = 2 * j + c2, cl = pos_class8(r, xx);
                const int qbits = 16 + k6, f = (1 << qbits) / 6;
                const int a = iabs(cw[c2][r]);
                int l = (int)(((long long)a * g_tab.mf8[m6][cl] + f) >> qbits);
                l = l > 2047 ? 2047 : l;
                l = cw[c2][r] < 0 ? -l : l;
                const int kk = g_tab.izz8[r * 8 + xx];
                stg16(&ctx->levels[(size_t)mbn * MB_LEVELS + L_LUMA + (4 * i8 + (kk & 3)) * 16 + (kk >> 2)], l);
                if (l) submask |= 1u << (kk & 3);
                const int ls = 16 * g_tab.v8[m6][cl];
                cw[c2][r] = qp >= 36 ? (l * ls) << (k6 - 6) : (l * ls + (1 << (5 - k6))) >> (6 - k6);
            }
        }
    }
    WAVE_SYNC();
    if (act) {
#pragma unroll
        for (int c2 = 0; c2 < 2; c2++)
#pragma unroll
            for (int r = 0; r < 8; r++) tb[r * 8 + 2 * j + c2] = cw[c2][r];
    }
    submask |= (unsigned)quad_xor<1>((int)submask); // the 8x8 block's four lanes
    submask |= (unsigned)quad_xor<2>((int)submask);
    WAVE_SYNC();
    if (act) { // 8.5.13: rows first ...
#pragma unroll
        for (int r = 0; r < 2; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) rs[r][i] = tb[(2 * j + r) * 8 + i];
            idct8_1d(rs[r]);
        }
    }
    WAVE_SYNC();
    if (act) {
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) tb[(2 * j + r) * 8 + i] = rs[r][i];
    }
    WAVE_SYNC();
    if (act) { // ... then columns, rounding
#pragma unroll
        for (int c2 = 0; c2 < 2; c2++) {
#pragma unroll
            for (int r = 0; r < 8; r++) cw[c2][r] = tb[r * 8 + 2 * j + c2];
            idct8_1d(cw[c2]);
        }
    }
    WAVE_SYNC();
    if (act) {
#pragma unroll
        for (int c2 = 0; c2 < 2; c2++)
#pragma unroll
            for (int r = 0; r < 8; r++) tb[r * 8 + 2 * j + c2] = (cw[c2][r] + 32) >> 6;
    }
    WAVE_SYNC();
    {
        const int *t = tile + ((pr >> 3) * 2 + (pc >> 3)) * 64 + (pr & 7) * 8 + (pc & 7);
        int o[4];
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = clip255(byte_of(pw, i) + t[i]);
        stx32<SC1>(ctx->rec_y + (size_t)(y0 + pr) * ctx->stride + x0 + pc, pack4(o[0], o[1], o[2], o[3]));
    }
    return (unsigned)(__ballot(act && ((submask >> j) & 1u)) & 0xFFFFull); // lane 4 i8 + j reports sub-block j = blkIdx 4 i8 + j
}
// PART: partitions (oracle: ORC_F_PART).  A lane owns four samples of row lane >> 2, columns 4 (lane & 3) ..: its 8x8 quadrant is (lane >> 5, (lane >> 1) & 1), and a
// quadrant's lanes differ in lane bits 0, 2, 3, 4.  Every candidate vector the refinement visits leaves the SAD of each quadrant behind (the wave-wide sum is
// built from the quadrant sums, so they cost nothing); afterwards every partition of every shape picks the visited vector with the lowest SAD + lambda * bits,
// the shape with the lowest total (+ lambda * header bits) against the 16x16 cost decides, and the prediction is taken again with a vector per lane.
DEV unsigned part_gsum(unsigned v) { // sum over the lanes of this lane's quadrant (two packed 16-bit sums: a quadrant's SAD is < 2^14)
    v += (unsigned)quad_xor<1>((int)v);
    v += (unsigned)row_xor4((int)v);
    v += (unsigned)row_xor8((int)v);
    v += (unsigned)__shfl_xor((int)v, 16, 64);
    return v;
}
DEV unsigned part_total(unsigned g) { g += (unsigned)quad_xor<2>((int)g); return g + (unsigned)__shfl_xor((int)g, 32, 64); } // the four quadrants' sums -> the macroblock's
// development builds (-DPMB_PROF, tests/devtools/pmb_phases.py): cycles a wave spends between the marks below, left per macroblock in ctx->dbrec (word 16 mbn + k; 0: the wave never reached mark k)
#ifdef PMB_PROF
#define PMB_NPH 11
#define PMB_MARK(k) do { const unsigned long long t1_ = __builtin_readcyclecounter(); ph_[k] = t1_ - t0_; t0_ = t1_; } while (0)
#define PMB_MARK0() unsigned long long ph_[PMB_NPH] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long t0_ = __builtin_readcyclecounter()
#define PMB_FLUSH() do { if (lane == 0) for (int k_ = 0; k_ < PMB_NPH; k_++) stg32((unsigned *)ctx->dbrec + (size_t)mbn * 16 + k_, (unsigned)ph_[k_]); } while (0)
#else
#define PMB_MARK(k) ((void)0)
#define PMB_MARK0() ((void)0)
#define PMB_FLUSH() ((void)0)
#endif
// What a macroblock of the fused stage reads that does NOT depend on the reference picture: its vector and its neighbours' (hence the predictors), its source
// samples, the skip candidate's SAD from the surface and the intra decision.  A gated launch loads these BEFORE its workgroup waits for the reference's bands, so
// that the two dependent round trips (field -> surface) are over when the gate opens -- they were the first 11 000 of a wave's ~45 000 cycles behind the gate
// (profiles/r04_pmb_phases.txt), i.e. of the stretch by which a picture's fused stage trails the previous deblocking launch.
struct pmb_pre_t { uint2 selfw; fpred_t fp; unsigned curw; uint2 sw; unsigned ds; uint4 dw; };
DEV pmb_pre_t pmb_preload(const frame_ctx_t *__restrict__ ctx, const int mbn, const int lane) {
    const int mbw = ctx->mbw, my = mbn / mbw, mx = mbn - my * mbw, x0 = mx * 16, y0 = my * 16;
    pmb_pre_t p;
    const imv_t *__restrict__ field = k_final_imv_dev(ctx);
    p.selfw = ldg64(field + mbn);
    {
        int sy = y0 + (lane >> 2);
        sy = sy < ctx->vis_h ? sy : ctx->vis_h - 1;
        p.curw = ldg32(ctx->src_y + (size_t)sy * ctx->src_stride + x0 + (lane & 3) * 4);
    }
    {
        const int cby = (lane >> 4) & 1, cbx = lane & 1;
        int sy = (y0 >> 1) + cby * 4 + ((lane >> 2) & 3);
        const int vh2 = ctx->vis_h >> 1;
        sy = sy < vh2 ? sy : vh2 - 1;
        p.sw = ldg64(ctx->src_uv + (size_t)sy * ctx->src_stride + 2 * ((x0 >> 1) + cbx * 4));
    }
    p.dw = ctx->intra_p ? ldg128(ctx->idec + (size_t)mbn * IDEC_BYTES + 16) : make_uint4(0, 0, 0, 0); // mode16 | cmode << 8 | use_i4 << 16; cost; cost_luma; 0
    p.fp = field_pred(field, mbw, mx, my, row_has_top(ctx, my));
    p.ds = (unsigned)*(const GAS uint16_t *)(ctx->surf + (size_t)mbn * SURF_U16 + ((p.fp.sy >> 2) + 16) * SURF_COLS + (p.fp.sx >> 2) + 16);
    return p;
}
template <bool SC1, bool PART, bool T8>
DEV void pmb_mb(const frame_ctx_t *__restrict__ ctx, sp_lds *L, const int mbn, const int lane, const int refine, const pmb_pre_t &pre) {
    const int mbw = ctx->mbw, mbh = ctx->mbh, stride = ctx->stride, W = mbw * 16, H = mbh * 16, qp = mb_qp_dev(ctx, mbn), lambda = ctx->lambda; // (quantisation only: search, refinement and decisions keep the picture's lambda)
    const int my = mbn / mbw, mx = mbn - my * mbw, x0 = mx * 16, y0 = my * 16;
    const dev_tables *T = &g_tab;
    PMB_MARK0();
    const uint2 selfw = pre.selfw;
    const int imx = (int)(int16_t)(selfw.x & 0xFFFF), imy = (int)(int16_t)(selfw.x >> 16); // whole-sample winner, quarter-sample units
    const unsigned di = selfw.y & 0xFFFFu, ibits = selfw.y >> 16;
    const fpred_t fp = pre.fp;
    const int pr = lane >> 2, pc = (lane & 3) * 4; // luma: lane owns row pr, columns pc .. pc+3
    const int py = (lane >> 2) & 3;
    const unsigned curw = pre.curw;
    int sv[4]; // chroma source samples of this lane (lanes 0..31)
    {
        const int c = (lane >> 1) & 1;
        const uint2 sw = pre.sw;
        const unsigned slo = c ? (sw.x >> 8) : sw.x, shi = c ? (sw.y >> 8) : sw.y;
        sv[0] = (int)(slo & 255); sv[1] = (int)((slo >> 16) & 255); sv[2] = (int)(shi & 255); sv[3] = (int)((shi >> 16) & 255);
    }
    int16_t *lv = ctx->levels + (size_t)mbn * MB_LEVELS;
    mb_info_t *mb = &ctx->mbi[mbn];
    // ---- 2. skip probe
    {
        const unsigned ds = pre.ds;
        bool pass = ctx->drop_sad && ds < ctx->drop_sad;
        const bool worth = ds <= di + (unsigned)(lambda * SKIP_MARGIN_BITS);
        PMB_MARK(0); // the field, the predictors and the skip candidate's SAD have arrived
        if (pass || worth) {
            const int X = x0 + (fp.sx >> 2), Y = y0 + (fp.sy >> 2);
            const bool inside = X >= 0 && Y >= 0 && X + 16 <= W && Y + 16 <= H;
            const unsigned pw = ref_word4(ctx->ref_y, stride, W, H, X + pc, Y + pr, inside);
            int pd[4];
            chroma_pred4(ctx, lane, x0, y0, W, H, fp.sx, fp.sy, pd);
            if (!pass) {
                int lev[4], x[4];
                if (pmb_luma_tq(T, lane, curw, pw, qp, true, lev, x) == 0) {
                    unsigned nz_c = 0, dc_c = 0;
                    chroma_rows4(ctx, T, lv, x0 >> 1, y0 >> 1, lane, pd, sv, qp, false, false, nullptr, nz_c, dc_c, true, 0, SC1);
                    pass = (nz_c | dc_c) == 0;
                }
            }
            PMB_MARK(1); // skip probe: prediction loaded, transformed, decided
            if (pass) {
                pmb_store_pred_only<SC1>(ctx, lv, lane, x0, y0, pw, pd);
                if (lane == 0) {
                    mb_info_t m;
                    m.mvx = (int16_t)fp.sx; m.mvy = (int16_t)fp.sy; m.mb_type = 1; m.i16_mode = 0; m.chroma_mode = 0; m.qp = (uint8_t)qp; m.nzmask = 0; m.cost = di;
                    st_mbinfo_x<SC1>(mb, m);
                }
                PMB_FLUSH();
                return;
            }
        }
    }
    // ---- 3. refinement around the whole-sample winner
    const int ix = x0 + (imx >> 2), iy = y0 + (imy >> 2);
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
    WAVE_SYNC();
    PMB_MARK(2); // the refinement's window is in LDS
    int bqx = imx, bqy = imy;
    // the whole-sample winner's SAD against the REFERENCE (the surface holds its SAD against the previous source: the search runs
    // source against source)
    unsigned q0 = 0, qh[4] = {0, 0, 0, 0}, qq[4] = {0, 0, 0, 0}; // PART: this lane's quadrant's SAD of candidate 0 / the half-sample round's eight / the quarter-sample round's eight (16-bit pairs)
    int hbx = imx, hby = imy;                                    // ... and the centre of the quarter-sample round
    unsigned best;
    if (PART) {
        q0 = part_gsum(__builtin_amdgcn_sad_u8(curw, sp_sample4(L, 1 + pc, 1 + pr, 0, 0), 0u));
        best = part_total(q0) + (unsigned)(lambda * (mvq_bits(imx - fp.px) + mvq_bits(imy - fp.py)));
    } else
        best = (unsigned)wave64_sum((int)__builtin_amdgcn_sad_u8(curw, sp_sample4(L, 1 + pc, 1 + pr, 0, 0), 0u)) +
               (unsigned)(lambda * (mvq_bits(imx - fp.px) + mvq_bits(imy - fp.py)));
    PMB_MARK(3); // the whole-sample winner's SAD against the reference
    if (refine) {
        sp_planes(L, lane);
        PMB_MARK(4); // half-sample planes
        { // half-sample round: SAD, the 8 candidates scored together (two 16-bit partial sums per register)
            const int cqx = bqx, cqy = bqy;
            unsigned acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int c8 = 0; c8 < 8; c8++) {
                const int k = c8 < 4 ? c8 : c8 + 1;
                const int qx = cqx + (k % 3 - 1) * 2, qy = cqy + (k / 3 - 1) * 2;
                const int ox = qx - imx, oy = qy - imy;
                const unsigned sad = __builtin_amdgcn_sad_u8(curw, sp_sample4(L, 1 + (ox >> 2) + pc, 1 + (oy >> 2) + pr, ox & 3, oy & 3), 0u);
                acc[c8 >> 1] |= sad << (16 * (c8 & 1));
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (PART) { qh[q] = part_gsum(acc[q]); acc[q] = part_total(qh[q]); }
                else acc[q] = (unsigned)wave64_sum((int)acc[q]);
            }
#pragma unroll
            for (int c8 = 0; c8 < 8; c8++) {
                const int k = c8 < 4 ? c8 : c8 + 1;
                const int qx = cqx + (k % 3 - 1) * 2, qy = cqy + (k / 3 - 1) * 2;
                const unsigned sad = (acc[c8 >> 1] >> (16 * (c8 & 1))) & 0xFFFFu;
                const unsigned cost = sad + (unsigned)(lambda * (mvq_bits(qx - fp.px) + mvq_bits(qy - fp.py)));
                if (cost < best) { best = cost; bqx = qx; bqy = qy; }
            }
        }
        hbx = bqx; hby = bqy;
        PMB_MARK(5); // half-sample round
        { // quarter-sample round: SATD; the standing best is restated in the same measure first
            const int cqx = bqx, cqy = bqy;
            {
                const int ox = cqx - imx, oy = cqy - imy;
                best = (pmb_satd(lane, curw, sp_sample4(L, 1 + (ox >> 2) + pc, 1 + (oy >> 2) + pr, ox & 3, oy & 3)) >> 1) +
                       (unsigned)(lambda * (mvq_bits(cqx - fp.px) + mvq_bits(cqy - fp.py)));
            }
#pragma unroll 1
            for (int k = 0; k < 9; k++) {
                if (k == 4) continue;
                const int qx = cqx + (k % 3 - 1), qy = cqy + (k / 3 - 1);
                const int ox = qx - imx, oy = qy - imy;
                const unsigned cw = sp_sample4(L, 1 + (ox >> 2) + pc, 1 + (oy >> 2) + pr, ox & 3, oy & 3);
                if (PART) {
                    const int c8 = k < 4 ? k : k - 1;
                    const unsigned g = part_gsum(__builtin_amdgcn_sad_u8(curw, cw, 0u)) << (16 * (c8 & 1));
                    qq[0] |= (c8 >> 1) == 0 ? g : 0u; qq[1] |= (c8 >> 1) == 1 ? g : 0u; qq[2] |= (c8 >> 1) == 2 ? g : 0u; qq[3] |= (c8 >> 1) == 3 ? g : 0u;
                }
                const unsigned d = pmb_satd(lane, curw, cw) >> 1;
                const unsigned cost = d + (unsigned)(lambda * (mvq_bits(qx - fp.px) + mvq_bits(qy - fp.py)));
                if (cost < best) { best = cost; bqx = qx; bqy = qy; }
            }
        }
    }
    PMB_MARK(6); // quarter-sample round
    // ---- the final vector's prediction and its cost in the SAD domain
    const int ox = bqx - imx, oy = bqy - imy;
    unsigned pw = sp_sample4(L, 1 + (ox >> 2) + pc, 1 + (oy >> 2) + pr, ox & 3, oy & 3);
    unsigned dsad = (unsigned)wave64_sum((int)__builtin_amdgcn_sad_u8(curw, pw, 0u));
    unsigned jinter = dsad + (unsigned)(lambda * (mvq_bits(bqx - fp.px) + mvq_bits(bqy - fp.py)));
    // ---- 3b. partitions: every partition picks among the vectors visited above
    int shape = 0, lvx = bqx, lvy = bqy; // the shape taken; this lane's vector
    if (PART && refine) {
        // key = cost << 5 | candidate (ties: the candidate visited first); per shape, this lane's partition's best
        unsigned k8 = 0xFFFFFFFFu, k168 = 0xFFFFFFFFu, k816 = 0xFFFFFFFFu;
#pragma unroll 1
        for (int k = 0; k < 17; k++) {
            const int cx = k == 0 ? imx : k < 9 ? imx + (((k - 1 < 4 ? k - 1 : k) % 3) - 1) * 2 : hbx + (((k - 9 < 4 ? k - 9 : k - 8) % 3) - 1);
            const int cy = k == 0 ? imy : k < 9 ? imy + (((k - 1 < 4 ? k - 1 : k) / 3) - 1) * 2 : hby + (((k - 9 < 4 ? k - 9 : k - 8) / 3) - 1);
            const unsigned bits = (unsigned)(lambda * (mvq_bits(cx - fp.px) + mvq_bits(cy - fp.py)));
            const int c8 = k == 0 ? 0 : k < 9 ? k - 1 : k - 9;
            const unsigned pk = k == 0 ? q0 : k < 9 ? ((c8 >> 1) == 0 ? qh[0] : (c8 >> 1) == 1 ? qh[1] : (c8 >> 1) == 2 ? qh[2] : qh[3])
                                                    : ((c8 >> 1) == 0 ? qq[0] : (c8 >> 1) == 1 ? qq[1] : (c8 >> 1) == 2 ? qq[2] : qq[3]);
            const unsigned sq = k == 0 ? pk : (pk >> (16 * (c8 & 1))) & 0xFFFFu;                 // this quadrant
            const unsigned sh = sq + (unsigned)quad_xor<2>((int)sq);                              // + the quadrant beside it: the 16x8 partition
            const unsigned sv_ = sq + (unsigned)__shfl_xor((int)sq, 32, 64);                      // + the quadrant below / above it: the 8x16 partition
            const unsigned a = ((sq + bits) << 5) | (unsigned)k, b = ((sh + bits) << 5) | (unsigned)k, c = ((sv_ + bits) << 5) | (unsigned)k;
            k8 = a < k8 ? a : k8; k168 = b < k168 ? b : k168; k816 = c < k816 ? c : k816;
        }
        // totals: one partition per group of lanes -> sum over the groups
        const unsigned c8x8 = (k8 >> 5), c168 = (k168 >> 5), c816 = (k816 >> 5);
        unsigned t8 = c8x8 + (unsigned)quad_xor<2>((int)c8x8); t8 += (unsigned)__shfl_xor((int)t8, 32, 64);
        const unsigned t168 = c168 + (unsigned)__shfl_xor((int)c168, 32, 64);
        const unsigned t816 = c816 + (unsigned)quad_xor<2>((int)c816);
        unsigned jb = jinter + (unsigned)lambda; // header bits: 1 (16x16), 3 (16x8, 8x16), 9 (P_8x8 + four sub_mb_type)
        unsigned key = 0;
        const unsigned j1 = t168 + 3u * (unsigned)lambda, j2 = t816 + 3u * (unsigned)lambda, j3 = t8 + 9u * (unsigned)lambda;
        if (j1 < jb) { jb = j1; shape = 1; key = k168; }
        if (j2 < jb) { jb = j2; shape = 2; key = k816; }
        if (j3 < jb) { jb = j3; shape = 3; key = k8; }
        shape = __builtin_amdgcn_readfirstlane(shape);
        if (shape) {
            const int k = (int)(key & 31u);
            lvx = k == 0 ? imx : k < 9 ? imx + (((k - 1 < 4 ? k - 1 : k) % 3) - 1) * 2 : hbx + (((k - 9 < 4 ? k - 9 : k - 8) % 3) - 1);
            lvy = k == 0 ? imy : k < 9 ? imy + (((k - 1 < 4 ? k - 1 : k) / 3) - 1) * 2 : hby + (((k - 9 < 4 ? k - 9 : k - 8) / 3) - 1);
            const int px_ = lvx - imx, py_ = lvy - imy;
            pw = sp_sample4(L, 1 + (px_ >> 2) + pc, 1 + (py_ >> 2) + pr, px_ & 3, py_ & 3);
            dsad = (unsigned)wave64_sum((int)__builtin_amdgcn_sad_u8(curw, pw, 0u));
            jinter = jb - (unsigned)lambda; // (a partitioned macroblock pays for its longer header in the intra test too)
        }
    }
    // ---- 4. intra instead?
    if (ctx->intra_p && di + (unsigned)lambda * ibits >= INTRA_GATE(lambda)) {
        const uint4 dw = pre.dw;
        const unsigned jintra = dw.z + (dw.z >> 3) + (unsigned)(lambda * 12);
        if (jintra < jinter) {
            if (lane == 0) {
                const bool i4 = ((dw.x >> 16) & 255u) != 0;
                mb_info_t m;
                m.mvx = 0; m.mvy = 0; m.mb_type = i4 ? 2 : 0; m.i16_mode = i4 ? 0 : (uint8_t)(dw.x & 255u); m.chroma_mode = (uint8_t)((dw.x >> 8) & 255u);
                m.qp = (uint8_t)qp; m.nzmask = 0; m.cost = dw.y;
                st_mbinfo_x<SC1>(mb, m);
            }
            PMB_FLUSH();
            return;
        }
    }
    PMB_MARK(7); // final cost, intra test
    // ---- 5. residual
    int pd[4];
    unsigned slot = 0; // PART: this lane's dword of the luma-DC slot (lanes 0 .. 7): the vectors of partitions 1 .. 3
    if (PART && shape) {
        const int src = (((lane >> 4) & 1) << 5) | ((lane & 1) << 1); // a chroma lane's 4x4 block belongs to the luma quadrant (lane bit 4, lane bit 0)
        const int cvx = __shfl(lvx, src, 64), cvy = __shfl(lvy, src, 64);
        chroma_pred4(ctx, lane, x0, y0, W, H, cvx, cvy, pd);
        const unsigned mine = ((unsigned)(uint16_t)lvx) | ((unsigned)(uint16_t)lvy << 16);
        const unsigned v1 = (unsigned)__shfl((int)mine, shape == 1 ? 32 : 2, 64), v2 = (unsigned)__shfl((int)mine, 32, 64), v3 = (unsigned)__shfl((int)mine, 34, 64);
        slot = lane == 0 ? v1 : (shape == 3 && lane == 1) ? v2 : (shape == 3 && lane == 2) ? v3 : 0u;
        bqx = __builtin_amdgcn_readfirstlane(lvx); bqy = __builtin_amdgcn_readfirstlane(lvy); // partition 0's vector goes to the record
    } else chroma_pred4(ctx, lane, x0, y0, W, H, bqx, bqy, pd);
    if (ctx->drop_sad && dsad < ctx->drop_sad) { // rate control's ladder below QP 51: prediction only
        pmb_store_pred_only<SC1>(ctx, lv, lane, x0, y0, pw, pd);
        if (PART && shape && lane < 8) stx32<SC1>(lv + L_LDC + 2 * lane, slot);
        if (lane == 0) {
            mb_info_t m;
            m.mvx = (int16_t)bqx; m.mvy = (int16_t)bqy; m.mb_type = 1; m.i16_mode = (uint8_t)shape; m.chroma_mode = 0; m.qp = (uint8_t)qp; m.nzmask = 0; m.cost = di;
            st_mbinfo_x<SC1>(mb, m);
        }
        PMB_FLUSH();
        return;
    }
    PMB_MARK(8); // chroma prediction loaded
    unsigned nz_luma;
    if (T8) {
        nz_luma = pmb_luma_t8<SC1>(ctx, (int *)L, lane, curw, pw, qp, mbn, x0, y0);
        // the sub-blocks without levels keep their zeros: the macroblock's levels are cleared first
    } else {
        const int fy = ((py & 1) << 1) | (py >> 1);
        const col_bf cb = make_col_bf(py);
        const int kz0 = (int)((0xFEA9DB83C7426510ull >> (16 * fy)) & 0xFFFF);
        int lev[4], x[4];
        nz_luma = pmb_luma_tq(T, lane, curw, pw, qp, true, lev, x);
        const int bx = lane & 3, by = lane >> 4, b = ((by >> 1) << 3) | ((bx >> 1) << 2) | ((by & 1) << 1) | (bx & 1);
#pragma unroll
        for (int i = 0; i < 4; i++) stg16(&lv[L_LUMA + b * 16 + ((kz0 >> (4 * i)) & 15)], lev[i]);
        inv_rows4(x);
        int o[4];
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = clip255(byte_of(pw, i) + ((inv_col(x[i], cb) + 32) >> 6));
        stx32<SC1>(ctx->rec_y + (size_t)(y0 + pr) * stride + x0 + pc, pack4(o[0], o[1], o[2], o[3]));
    }
    PMB_MARK(9); // luma residual
    unsigned nz_c = 0, dc_c = 0;
    chroma_rows4(ctx, T, lv, x0 >> 1, y0 >> 1, lane, pd, sv, qp, false, true, nullptr, nz_c, dc_c, true, 0, SC1);
    PMB_MARK(10); // chroma residual
    if (lane == 0) {
        unsigned nzm = nz_luma | (nz_c << 16);
        if (dc_c & 1) nzm |= NZ_CBDC;
        if (dc_c & 2) nzm |= NZ_CRDC;
        if (T8 && nz_luma) nzm |= NZ_T8; // transform_size_8x8_flag: sent (and read by the deblocker) only with luma levels
        mb_info_t m;
        m.mvx = (int16_t)bqx; m.mvy = (int16_t)bqy; m.mb_type = 1; m.i16_mode = (uint8_t)shape; m.chroma_mode = 0; m.qp = (uint8_t)qp; m.nzmask = nzm; m.cost = di;
        st_mbinfo_x<SC1>(mb, m);
    }
    if (PART && shape) { if (lane < 8) stx32<SC1>(lv + L_LDC + 2 * lane, slot); } // the luma-DC slot: the vectors of partitions 1 .. 3 (the deblocker of this picture may read them behind the row counts)
    else if (lane < 2) stg128(lv + L_LDC + 8 * lane, make_uint4(0, 0, 0, 0)); // luma DC levels: unused by P macroblocks, kept zero
    PMB_FLUSH();
}

// The launch covers macroblocks mb0 .. mb1-1, four per workgroup and turn.
// GATED: the reference picture's band deblocker may still be running (this launch then sits on the intra stream, beside the
// reference's deblocking).  Macroblock row r reads reference lines up to 16 (r + 2) + 3 (whole-sample vectors of +-16, three more for
// the six-tap filter and the quarter sample; chroma likewise inside row r + 2), all of which the deblocker's row r + 2 stores, or
// the row below it in the same band or the next; upwards it reads from line 16 (r - 2) + 13 on.  So a workgroup waits until every band
// from the one that holds row r - 2 to the one that holds row r + 2 carries the reference's epoch for both planes, and then takes an
// acquire fence: the deblocker's workgroups sit on other XCDs.  Nothing below a band a wave has acquired is ever read, so no stale line can enter this XCD's L2 ahead of the
// fence.
// Waiting workgroups must never keep the kernel they wait for from being placed (it may not have started yet: after a picture whose
// stages ran in order this launch is early by a whole picture).  A band of the deblocker needs three waves of 112 VGPRs on each SIMD
// of one CU; four workgroups of this kernel on a CU (4 x 56 VGPRs per SIMD) leave no room for it, and a chip full of waiting
// workgroups leaves none anywhere (seen: every band timed out).  So a gated launch follows wait_started_kernel (k_deblock.hip) on its
// stream: by the time the first workgroup of this kernel is placed, every workgroup of the reference's deblocking launch is (nothing
// else that launch waits for can be outstanding: the reference's fused stage and intra_p_kernel are earlier on this stream).  The
// deblocker's bands finish within the last third of its run (they advance along x together), and that is when this launch, whole
// and resident, takes its macroblocks row by row behind them.
// ROWS (only with GATED): the picture's deblocking launch is already on the chip and waits for this kernel's rows -- samples and records
// are stored through to memory (sc1) and every macroblock is counted for its row.
template <bool GATED, bool ROWS, bool PART, bool T8>
__global__ __launch_bounds__(256) void pmb_kernel(const frame_ctx_t cv, int mb0, int mb1, int refine, const unsigned *__restrict__ gate_done, unsigned ref_epoch, unsigned *err, unsigned *row_done) {
    const frame_ctx_t *__restrict__ ctx = &cv;
    __shared__ __attribute__((aligned(16))) sp_lds LD[4];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // an SGPR: what is derived from it is scalar control flow
    const int mbn = mb0 + 4 * (int)blockIdx.x + wave;
    if (mb0 == 0 && blockIdx.x == 0) tl_first(ctx, 2);
    pmb_pre_t pre;
    if (mbn < mb1) pre = pmb_preload(ctx, mbn, lane); // (wave-uniform) nothing here reads the reference: issued in front of the gate
    if (GATED) { // wave 0 waits for the bands the workgroup's macroblocks read, the others for wave 0
        if (wave == 0) {
            // EVERY band the window touches, not only the lowest: a band's word says that ITS lines are in memory -- the band above it has
            // filtered further (its strips were needed), but may not have written its own lines back yet, and a band that had no strips
            // to wait for can be done long before the one above it.  (Waiting for the lowest band alone gave streams that differed from
            // the in-order path in one macroblock every few hundred pictures.)
            const int mbw = ctx->mbw, mbh = ctx->mbh;
            const int first = mb0 + 4 * (int)blockIdx.x, last = first + 3 < mb1 ? first + 3 : mb1 - 1;
            int r_lo = first / mbw - 2, r_hi = last / mbw + 2;
            r_lo = r_lo > 0 ? r_lo : 0;
            r_hi = r_hi < mbh ? r_hi : mbh - 1;
            const uint2 *flags = (const uint2 *)(gate_done + DB_DONE_STRIDE * (blockIdx.x & (DB_DONE_COPIES - 1)));
            const int b_hi = r_hi / MI355_BAND_ROWS;
#ifdef DBG_OLD_GATE /* what round 2 first shipped: the lowest band alone (tests/test_adversarial_gpu.py was checked to FAIL with this and the ADVBAND delay) */
            int spins = 0, b = b_hi;
#else
            int spins = 0, b = r_lo / MI355_BAND_ROWS;
#endif
            for (;;) {
                const uint2 f = ld64_sc1(flags + b);
                if (f.x == ref_epoch && f.y == ref_epoch) { if (++b > b_hi) break; continue; }
#ifndef PMB_POLL_SLEEP /* (A/B: 16 and 127 give 1.00 x at 1080p, 127 gives 1.01 x at 2160p: the back-off between polls is not where the time goes) */
#define PMB_POLL_SLEEP 64
#endif
                __builtin_amdgcn_s_sleep(PMB_POLL_SLEEP);
                if (++spins > DB_SPIN_MAX) { st_sc1(err, 3u); break; }
                if ((spins & 255) == 0 && ld_sc1(err)) break; // somebody else gave up: nobody waits again // bounded; the host reports the picture as failed
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        tl_last(ctx, 3);
    }
    if (mbn >= mb1) return; // wave-uniform
    pmb_mb<ROWS, PART, T8>(ctx, &LD[wave], mbn, lane, refine, pre);
    if (ROWS) { // this macroblock's samples and record are in memory: count it for its row (the picture's deblocking launch, already on the chip, waits for whole rows)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(row_done + (mbn / ctx->mbw) * MI355_PROG_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    tl_last(ctx, 4);
}

// I pictures: the padded source luma for the next picture's search (P pictures: me_kernel writes it on its way)
__global__ __launch_bounds__(256) void copy_luma_kernel(const frame_ctx_t cv) {
    const frame_ctx_t *__restrict__ ctx = &cv;
    const int W = ctx->mbw * 16, H = ctx->mbh * 16, per_row = W / 16;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= per_row * H) return;
    const int y = i / per_row, x = (i - y * per_row) * 16;
    const int sy = y < ctx->vis_h ? y : ctx->vis_h - 1;
    stg128(ctx->psrc_out + (size_t)y * ctx->stride + x, ldg128(ctx->src_y + (size_t)sy * ctx->src_stride + x));
}
void k_launch_copy_luma(const frame_ctx_t *h_ctx, hipStream_t s) {
    const int n = h_ctx->mbw * h_ctx->mbh * 16;
    hipLaunchKernelGGL(copy_luma_kernel, dim3((n + 255) / 256), dim3(256), 0, s, *h_ctx);
}

// whole-sample field -> records for the two-kernel (8x8-transform) path: subpel_kernel compares absolute-vector costs
__global__ __launch_bounds__(256) void imv_to_mbi_kernel(const frame_ctx_t cv, int mb0, int mb1) {
    const frame_ctx_t *__restrict__ ctx = &cv;
    const int i = mb0 + blockIdx.x * 256 + threadIdx.x;
    if (i >= mb1) return;
    const uint2 w = ldg64(k_final_imv_dev(ctx) + i);
    const int vx = (int)(int16_t)(w.x & 0xFFFF), vy = (int)(int16_t)(w.x >> 16);
    stg32(&ctx->mbi[i].mvx, w.x);
    stg32(&ctx->mbi[i].cost, (w.y & 0xFFFFu) + (unsigned)(ctx->lambda * (mvq_bits(vx) + mvq_bits(vy))));
}

// =================================================================== launchers
// The P-picture kernels take a macroblock-row range [row0, row1).
void k_launch_me(const frame_ctx_t *h_ctx, int mbw, int row0, int row1, hipStream_t s) {
    int strips = (mbw + ME_MBS - 1) / ME_MBS;
    if (row1 > row0) hipLaunchKernelGGL(me_kernel, dim3(strips * (row1 - row0)), dim3(64 * ME_MBS), 0, s, *h_ctx, row0);
}
void k_launch_me_select(const frame_ctx_t *h_ctx, int mbw, int row0, int row1, const imv_t *in, imv_t *out, const imv_t *prev, int mode, hipStream_t s) {
    int n = mbw * (row1 - row0);
    if (n <= 0) return;
#ifdef NO_SPARSE_SELECT /* A/B build (tools/build_variant.sh): the dense kernel for every pass */
    if (false) hipLaunchKernelGGL(me_select_sparse_kernel,
#else
    if (mode == 2) hipLaunchKernelGGL(me_select_sparse_kernel,
#endif
  dim3((n + 4 * SEL_SPW - 1) / (4 * SEL_SPW)), dim3(256), 0, s, *h_ctx, row0 * mbw, row1 * mbw, in, out, prev); // later passes: mostly copies
    else hipLaunchKernelGGL(me_select_kernel, dim3((n + 3) / 4), dim3(256), 0, s, *h_ctx, row0 * mbw, row1 * mbw, in, out, prev, mode);
}
// The ME_ITERS iterations walk a -> b -> c -> a ...: iteration k reads field k % 3, writes (k + 1) % 3 and compares with what iteration k - 1 read.
void k_launch_me_select_all(const frame_ctx_t *h_ctx, int mbw, int row0, int row1, hipStream_t s) {
    imv_t *const f[3] = {h_ctx->imv_a, h_ctx->imv_b, h_ctx->imv_c};
    for (int it = 0; it < ME_ITERS; it++) k_launch_me_select(h_ctx, mbw, row0, row1, f[it % 3], f[(it + 1) % 3], it ? f[(it - 1) % 3] : nullptr, it ? 2 : 1, s);
}
const imv_t *k_final_imv(const frame_ctx_t *h_ctx) { return ME_ITERS % 3 == 0 ? h_ctx->imv_a : ME_ITERS % 3 == 1 ? h_ctx->imv_b : h_ctx->imv_c; }
void k_launch_imv_to_mbi(const frame_ctx_t *h_ctx, int mbw, int row0, int row1, hipStream_t s) {
    int n = mbw * (row1 - row0);
    if (n > 0) hipLaunchKernelGGL(imv_to_mbi_kernel, dim3((n + 255) / 256), dim3(256), 0, s, *h_ctx, row0 * mbw, row1 * mbw);
}
void k_launch_subpel(const frame_ctx_t *h_ctx, int mbw, int row0, int row1, hipStream_t s) {
    if (row1 > row0) hipLaunchKernelGGL(subpel_kernel, dim3((mbw * (row1 - row0) + 3) / 4), dim3(256), 0, s, *h_ctx, row0 * mbw, row1 * mbw);
}
void k_launch_pmb(const frame_ctx_t *h_ctx, int mbw, int row0, int row1, int refine, const unsigned *gate_done, unsigned ref_epoch, unsigned *d_err, unsigned *d_row_done, hipStream_t s) {
    const int n = mbw * (row1 - row0), g = (n + 3) / 4;
    if (n <= 0) return;
#define PMB_LAUNCH(G, R, P, T) hipLaunchKernelGGL((pmb_kernel<G, R, P, T>), dim3(g), dim3(256), 0, s, *h_ctx, row0 * mbw, row1 * mbw, refine, gate_done, ref_epoch, d_err, d_row_done)
    if (h_ctx->t8) { // High profile: 8x8 transform for the inter macroblocks' luma (no partitions on this path)
        if (gate_done && d_row_done) PMB_LAUNCH(true, true, false, true);
        else if (gate_done) PMB_LAUNCH(true, false, false, true);
        else PMB_LAUNCH(false, false, false, true);
    } else if (h_ctx->partitions) {
        if (gate_done && d_row_done) PMB_LAUNCH(true, true, true, false);
        else if (gate_done) PMB_LAUNCH(true, false, true, false);
        else PMB_LAUNCH(false, false, true, false);
    } else if (gate_done && d_row_done) PMB_LAUNCH(true, true, false, false);
    else if (gate_done) PMB_LAUNCH(true, false, false, false);
    else PMB_LAUNCH(false, false, false, false);
#undef PMB_LAUNCH
}
