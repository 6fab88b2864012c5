// scl_fast4.h -- CA-SCL / SCL, L = 8, N = 1024: FOUR codewords per wavefront.
//
// k_scl_fast2 (two codewords per wavefront) still spends most of its instructions where a path has fewer values than
// lanes: the whole per-leaf decision (PHI, pruning, fork bookkeeping) carries data in one of a path's four lanes,
// level 0 in one, level 1 in two.  Here a path owns TWO lanes and every 8-lane path group is shared by four
// independent codewords:
//
//     lane = p*8 + c*2 + pos        p = path slot 0..7, c = codeword 0..3, pos = 0..1
//
// so every decision and level-0 instruction serves four codewords and level 1 fills its lanes.  Level t >= 2 holds
// 2^t/2 registers per lane (element e = pos + 2r, level t at A[2^t/2 .. 2^t)): levels 2..6 are 62 registers, twice
// k_scl_fast2's, which is what limits this kernel to two wavefronts per SIMD in f64 -- eight codewords per SIMD
// against six.  Pairs (e, e + 2^t/2) of every register level are lane-local; only level 0 needs its partner lane.
// Levels 7, 8, the top-left level and the channel vector live in the per-wave global scratch / the input exactly as in
// k_scl_fast2, and so do the lazy per-level owner pointers, the partial sums, the incremental CRC, the survivor
// matching and the tie rule.  The leaf schedule (frozen pattern) is common to the four codewords, so control flow
// stays wave-uniform.  Arithmetic is the reference's, operation for operation (SCL_1024.c:343-374, :404-448,
// :481-502, :547-680; CASCL_1024_L8.c:569-598, :725-755).
#pragma once
#include "scl_fast2.h"

namespace polar {

template <typename R>
struct Fast4Cfg {
    static constexpr int NLOG = 10, N = 1024, NW = 32, TOP = 9, HI = 8, L = 8, CW = 4;
    static constexpr int NA = 64;   // levels 2..6, level t at offset 2^t/2
    static constexpr int WAVES = 4;
#ifndef POLAR_F4_WAVES_F64
#define POLAR_F4_WAVES_F64 2
#endif
#ifndef POLAR_F4_WAVES_F32
#define POLAR_F4_WAVES_F32 3
#endif
    static constexpr int MIN_WAVES_PER_SIMD = sizeof(R) == 8 ? POLAR_F4_WAVES_F64 : POLAR_F4_WAVES_F32;
    static constexpr int NFA = HI - 3;  // pointer fields: LLR levels 4..8, then partial-sum levels 5..9
    // per-wave scratch (elements of R).  Levels 8 and 7 are stored LANE-INTERLEAVED: element e = pos + 2 rr + 64 k of the
    // row of lane l = slot*8 + c*2 + pos sits at ((k*32 + rr)*64 + l), so that one load or store instruction of the
    // wavefront -- fixed (k, rr), every lane its own or its owner slot's row -- touches ONE contiguous 512-byte block
    // (4 cache lines, written whole) instead of 32 scattered 16-byte pieces of 32 rows.
    static constexpr size_t sc_l8 = 0;                    // [4][32][64]
    static constexpr size_t sc_l7 = sc_l8 + 4 * 32 * 64;  // [2][32][64]
    static constexpr size_t sc_tl = sc_l7 + 2 * 32 * 64;  // [4 codewords][512]
    static constexpr size_t scratch_elems = sc_tl + CW * 512;  // per wave
    // block-shared LDS
    static constexpr size_t off_lut = 0;
    static constexpr size_t off_frz = off_lut + ((Lut<R>::bytes + 15) / 16) * 16;
    static constexpr size_t off_crc = off_frz + 4 * NW;
    static constexpr size_t off_kth = off_crc + 4 * N;      // kth[mask][k] = index of the k-th set bit of mask (u8)
    static constexpr size_t shared_bytes = off_kth + 256 * 8;
    // per-wave LDS
    static constexpr size_t off_bl = 0;                         // saved partial sums [4][8][NW]
    static constexpr size_t off_cw = off_bl + 4 * CW * 8 * NW;  // working partial sums [4][8][NW]
    static constexpr size_t off_cd = off_cw + 4 * CW * 8 * NW;  // candidates [4][16]
    static constexpr size_t off_ky = off_cd + sizeof(R) * 16 * CW;  // keys [4][16]
    static constexpr size_t off_sg = off_ky + 64 * CW;          // top-level staging [4][256]
    static constexpr size_t per_wave = off_sg + sizeof(R) * 256 * CW;
    static constexpr size_t total = shared_bytes + WAVES * per_wave;
};

template <typename R, typename IN, bool CRC_ON>
struct Fast4Dec {
    using C = Fast4Cfg<R>;
    static constexpr int N = C::N, NW = C::NW, TOP = C::TOP, HI = C::HI, L = 8, NFA = C::NFA;

    R A[C::NA];      // levels 2..6: level t at A[2^t/2 + r], element e = pos + 2r
    R a1;            // level 1 (element pos)
    R PM;            // valid at pos 0
    uint32_t ptr, crc, bl0, fl;
    int logact;
    int p, c, pos, lane, gl;   // gl = c*2 + pos: lane offset inside a path group
    uint32_t pos0_mask;        // ~0 in the lane that holds its path's metric (pos 0), else 0
    int own_addr;              // rank network: byte address of this ranking lane's key
    int rrow;                  // rank network: the codeword this lane ranks for (its row of 16 lanes)
    Lut<R> lut;
    R *cand, *stg;
    R *cand_rank;              // candidates of the codeword this lane ranks for
    uint32_t *blw, *curw, *keys;   // this lane's codeword slice of the per-wave arrays
    const unsigned char *keys_wave;
    const uint32_t *crct;
    const unsigned char *kth;
    R *scr;          // this wave's scratch
    const IN *src;   // this codeword's input row
    double sigma;

    __device__ __forceinline__ int pa(int t) const { return (ptr >> (3 * (t - 4))) & 7; }
    __device__ __forceinline__ void set_pa(int t, int v) { ptr = (ptr & ~(7u << (3 * (t - 4)))) | ((uint32_t)v << (3 * (t - 4))); }
    __device__ __forceinline__ int pb(int t) const { return (ptr >> (3 * (NFA + t - 5))) & 7; }
    __device__ __forceinline__ void set_pb(int t, int v) { ptr = (ptr & ~(7u << (3 * (NFA + t - 5)))) | ((uint32_t)v << (3 * (NFA + t - 5))); }
#ifndef POLAR_F4_WIDE_LUT1
#define POLAR_F4_WIDE_LUT1 0
#endif
    // wide steps: the compact two-round-trip form (the one-round-trip form, POLAR_F4_WIDE_LUT1=1, measured 13.0 -> 19.5 ms: spills)
    __device__ __forceinline__ R chk(R a, R b) const
    {
        if constexpr (sizeof(R) == 4 || (POLAR_F4_WIDE_LUT1 != 0)) return chk_lut1<R>(a, b, lut);
        else return chk_lut<R>(a, b, lut);
    }
    __device__ __forceinline__ R chks(R a, R b) const { return chk_lut1<R>(a, b, lut); }
    __device__ __forceinline__ R chv(int e) const
    {
        double v = (double)src[e];
        if (sigma > 0) v = llr_from_y(v, sigma);
        return (R)v;
    }
    static __device__ __forceinline__ unsigned fresh(unsigned off)
    {
        __asm__ volatile("" : "+v"(off));
        return off;
    }
    // level 8 / level 7 of slot `slot` (of this lane's codeword, this lane's pos): add ((k*32 + rr) << 6) for element pos + 2 rr + 64 k
    __device__ __forceinline__ R *l8(int slot) const { return scr + fresh(C::sc_l8 + slot * 8 + gl); }
    __device__ __forceinline__ R *l7(int slot) const { return scr + fresh(C::sc_l7 + slot * 8 + gl); }
    __device__ __forceinline__ R *tls() const { return scr + fresh(C::sc_tl + c * 512); }
    static __device__ __forceinline__ int at(int k, int rr) { return ((k * 32 + rr) << 6); }

    // ---- register levels: f on own data.  T in [2, 5]: level T from level T+1 ----
    template <int T>
    __device__ __forceinline__ void f_reg()
    {
        constexpr int RO = (1 << T) / 2;
#pragma unroll
        for (int r = 0; r < RO; ++r) A[RO + r] = chk(A[2 * RO + r], A[3 * RO + r]);
        if constexpr (T >= 4) set_pa(T, p);
    }
    // ---- register levels: g from the owner's level T+1 (bpermute), T in {4, 5} ----
    template <int T>
    __device__ __forceinline__ void g_reg()
    {
        constexpr int RO = (1 << T) / 2;
        const int sl = pa(T + 1) * 8 + gl;
        uint32_t w;
        if constexpr (T == 5) w = blw[pb(5) * NW + 1] >> pos;   // level 5: word 1, bit e = pos + 2r
        else w = bl0 >> (16 + pos);                              // level 4: bits 16 + e
#pragma unroll
        for (int r = 0; r < RO; ++r) {
            const R x = __shfl(A[2 * RO + r], sl), y = __shfl(A[3 * RO + r], sl);
            A[RO + r] = g_bit<R>(x, y, w, 2 * r);
        }
        set_pa(T, p);
    }
    __device__ __forceinline__ void g3()  // level 3 (eager) from the owner's level 4
    {
        const int sl = pa(4) * 8 + gl;
        const uint32_t w = bl0 >> (8 + pos);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const R x = __shfl(A[8 + r], sl), y = __shfl(A[12 + r], sl);
            A[4 + r] = g_bit<R>(x, y, w, 2 * r);
        }
    }

    // level-6 registers A[32 + 8q .. 32 + 8q + 8) <- t[0..8), q wave-uniform (keeps the pass loops rolled without a
    // dynamically indexed register)
    __device__ __forceinline__ void put_l6(int q, const R *t)
    {
        if (q == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) A[32 + i] = t[i];
        } else if (q == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) A[40 + i] = t[i];
        } else if (q == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) A[48 + i] = t[i];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) A[56 + i] = t[i];
        }
    }

    // ---- scratch levels ----
    // d >= 8 (octets 0, 32, 64, 96): level 8 from the top level (f, or g when gstep), f down to level 6.
    // Pass rr (0..31): level-8 elements e0 + 64k (k < 4), level-7 elements e0, e0 + 64, level-6 element e0 = pos + 2 rr.
    // The top-level operands are the same for all eight paths of a codeword, so the codeword's 16 lanes fetch them
    // once per chunk of 8 passes (16 consecutive values of e0: 16 segments of 16 elements when right, 8 when left),
    // park them in an LDS staging buffer and every path reads them from there; the next chunk's loads are in flight
    // during the compute.  Staging index = seg*16 + within, seg = k + 4h: right: h selects the channel offset
    // {0, 512, 256, 768} -> operands (ch[e], ch[e+512]) at h = 0, 1 and (ch[e+256], ch[e+768]) at h = 2, 3;
    // left: h in {0, 1} selects the top-left offset {0, 256}.
    __device__ __forceinline__ R top_src(bool right, int idx, int q) const
    {
        const int seg = idx >> 4, within = idx & 15;
        const int k = seg & 3, h = seg >> 2;
        const int e = 16 * q + within + 64 * k;
        if (right) {
            const int off = (h & 1) * 512 + (h >> 1) * 256;
            return chv(e + off);
        }
        return ld_sc(tls() + e + 256 * h);
    }
    __device__ __forceinline__ void from_top(bool right, bool gstep)
    {
        vm_drain();
        const uint32_t *bt = blw + pb(TOP) * NW + 16;  // beta_9: words 16..31
        const uint32_t *bh = blw + pb(HI) * NW + 8;    // beta_8: words 8..15
        R *o8 = l8(p), *o7 = l7(p);
        const int w16 = p * 2 + pos;                   // lane index inside the codeword
        const int nld = right ? 16 : 8;                // staged elements per lane and chunk
        R *pre = A + 2;    // levels 2..4 (A[2..15]) and A[16..17] are dead during this step (recomputed below)
#pragma unroll
        for (int m = 0; m < 16; ++m) pre[m] = (m < nld) ? top_src(right, w16 + 16 * m, 0) : R(0);
        for (int q = 0; q < 4; ++q) {
            lds_fence();
#pragma unroll
            for (int m = 0; m < 16; ++m)
                if (m < nld) stg[w16 + 16 * m] = pre[m];
            lds_fence();
            if (q < 3) {
#pragma unroll
                for (int m = 0; m < 16; ++m) pre[m] = (m < nld) ? top_src(right, w16 + 16 * m, q + 1) : R(0);
            }
            R t6[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int rr = 8 * q + i;
                const int e0 = pos + 2 * rr;       // 16 q + wi
                const int wi = 2 * i + pos;        // position inside a staged segment
                const int sh = e0 & 31;
                const int wq = e0 >> 5;            // word of a 64-bit slice: e0 + 64k sits in word 2k + wq
                R v8[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    R x, y;
                    if (right) {
                        const uint32_t w0 = bt[2 * k + wq], w1 = bt[8 + 2 * k + wq];
                        x = g_bit<R>(stg[(k + 0) * 16 + wi], stg[(k + 4) * 16 + wi], w0, sh);    // ch[e], ch[e+512]
                        y = g_bit<R>(stg[(k + 8) * 16 + wi], stg[(k + 12) * 16 + wi], w1, sh);   // ch[e+256], ch[e+768]
                    } else {
                        x = stg[(k + 0) * 16 + wi];   // tl[e]
                        y = stg[(k + 4) * 16 + wi];   // tl[e+256]
                    }
                    if (gstep) v8[k] = g_bit<R>(x, y, bh[2 * k + wq], sh);
                    else v8[k] = chk(x, y);
                    o8[at(k, rr)] = v8[k];
                }
                const R v70 = chk(v8[0], v8[2]), v71 = chk(v8[1], v8[3]);
                o7[at(0, rr)] = v70;
                o7[at(1, rr)] = v71;
                t6[i] = chk(v70, v71);
            }
            put_l6(q, t6);
        }
        set_pa(8, p);
        set_pa(7, p);
        set_pa(6, p);
    }
    // d == 7: g to level 7 from the owner's level 8, f to level 6.
    __device__ __forceinline__ void from_l8()
    {
        vm_drain();
        const R *s8 = l8(pa(8));
        const uint32_t *b7 = blw + pb(7) * NW + 4;  // beta_7: words 4..7
        R *o7 = l7(p);
#ifndef POLAR_F4_L8_BATCH
#define POLAR_F4_L8_BATCH 4
#endif
        constexpr int NB = POLAR_F4_L8_BATCH;   // passes per batch: 4 NB loads in flight per lane
        for (int q = 0; q < 4; ++q) {
            R t6[8];
#pragma unroll
            for (int i2 = 0; i2 < 8; i2 += NB) {
                R in[4 * NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int rr = 8 * q + i2 + j;
#pragma unroll
                    for (int k = 0; k < 4; ++k) in[4 * j + k] = ld_sc(s8 + at(k, rr));
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int rr = 8 * q + i2 + j;
                    const int e0 = pos + 2 * rr;
                    const int sh = e0 & 31, wq = e0 >> 5;
                    const R v70 = g_bit<R>(in[4 * j], in[4 * j + 2], b7[wq], sh);
                    const R v71 = g_bit<R>(in[4 * j + 1], in[4 * j + 3], b7[2 + wq], sh);
                    o7[at(0, rr)] = v70;
                    o7[at(1, rr)] = v71;
                    t6[i2 + j] = chk(v70, v71);
                }
            }
            put_l6(q, t6);
        }
        set_pa(7, p);
        set_pa(6, p);
    }
    __device__ __forceinline__ void from_l7()  // d == 6: level 6 = g of the owner's level 7
    {
        vm_drain();
        const R *s7 = l7(pa(7));
        const uint32_t *b6 = blw + pb(6) * NW + 2;  // beta_6: words 2, 3
#ifndef POLAR_F4_L7_BATCH
#define POLAR_F4_L7_BATCH 8
#endif
        constexpr int NB = POLAR_F4_L7_BATCH;   // passes per batch: 2 NB loads in flight per lane
        for (int q = 0; q < 4; ++q) {
            R t6[8];
#pragma unroll
            for (int i4 = 0; i4 < 8; i4 += NB) {
                R x[NB], y[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int rr = 8 * q + i4 + j;
                    x[j] = ld_sc(s7 + at(0, rr));
                    y[j] = ld_sc(s7 + at(1, rr));
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int e0 = pos + 2 * (8 * q + i4 + j);
                    t6[i4 + j] = g_bit<R>(x[j], y[j], b6[e0 >> 5], e0 & 31);
                }
            }
            put_l6(q, t6);
        }
        set_pa(6, p);
    }

    // head of octet o: g at level d = ctz(8o) (root f for o = 0), f chain down to level 3 (A[4..7])
    __device__ __forceinline__ void octet_head(int o)
    {
        const int d = (o == 0) ? 10 : 3 + __builtin_ctz((unsigned)o);
        if (d >= 8) from_top(/*right=*/o >= N / 16, /*gstep=*/d == 8);
        else if (d == 7) from_l8();
        else if (d == 6) from_l7();
        else if (d == 5) g_reg<5>();
        else if (d == 4) g_reg<4>();
        else g3();
        if (d > 5) f_reg<5>();
        if (d > 4) f_reg<4>();
        if (d > 3) f_reg<3>();
    }

    // ---- partial sums (per path, identical to k_scl_fast2) ----
    template <int K>
    __device__ __forceinline__ void set_bit_k(int o, uint32_t bit)
    {
        if constexpr ((K & 1) == 0) {
            bl0 = (bl0 & ~2u) | (bit << 1);
        } else if constexpr ((K & 3) == 1) {
            const uint32_t c1 = (((bl0 >> 1) & 1u) ^ bit) | (bit << 1);
            bl0 = (bl0 & ~0xCu) | (c1 << 2);
        } else if constexpr (K == 3) {
            const uint32_t c1 = (((bl0 >> 1) & 1u) ^ bit) | (bit << 1);
            const uint32_t c2 = (((bl0 >> 2) & 3u) ^ c1) | (c1 << 2);
            bl0 = (bl0 & ~0xF0u) | (c2 << 4);
        } else {
            set_bit_tail(8 * o + 7, bit);
        }
    }
    __device__ __forceinline__ void set_bit_tail(int j, uint32_t bit)
    {
        uint32_t cur = bit;
        const int z = __builtin_ctz(~(unsigned)j);
        const int zl = z < 5 ? z : 5;
        for (int t = 0; t < zl; ++t) {
            const int h = 1 << t;
            const uint32_t mask = (1u << h) - 1u;
            const uint32_t l = (bl0 >> h) & mask;
            cur = (l ^ (cur & mask)) | ((cur & mask) << h);
        }
        if (z < 5) {
            const int h = 1 << z;
            const uint32_t mask = (1u << h) - 1u;
            bl0 = (bl0 & ~(mask << h)) | ((cur & mask) << h);
            return;
        }
        lds_fence();
        if (pos == 0) curw[p * NW] = cur;
        lds_fence();
        int t = 5;
        while (t < 10 && ((j >> t) & 1)) {
            const int nw = 1 << (t - 5);
            const int sb = pb(t);
            for (int w = pos; w < nw; w += 2) {
                const uint32_t cc = curw[p * NW + w];
                const uint32_t l = blw[sb * NW + nw + w];
                curw[p * NW + w] = l ^ cc;
                curw[p * NW + w + nw] = cc;
            }
            lds_fence();
            ++t;
        }
        if (t < 10) {
            const int nw = 1 << (t - 5);
            for (int w = pos; w < nw; w += 2) blw[p * NW + nw + w] = curw[p * NW + w];
            set_pb(t, p);
            lds_fence();
        }
    }

    // ---- survivors (SCL_1024.c:612-633) for the four codewords; returns this lane's codeword's 16-bit mask ----
    // Row r of the wave (16 lanes) ranks codeword r: lane i of the row holds candidate i = slot + 8*branch and compares
    // its key with the fifteen others (row_ror:1 .. 15).
    template <int CTRL>
    __device__ __forceinline__ uint32_t rank_step(uint32_t acc, uint32_t own, uint32_t own1) const
    {
        // key_m - key_own - 1 is negative iff key_m <= key_own (keys < 2^31); v_alignbit shifts that sign bit in
        return __builtin_amdgcn_alignbit(acc, (uint32_t)dpp_i<CTRL>((int)own) - own1, 31);
    }
    __device__ __forceinline__ uint32_t survivors(R c0, R c1)
    {
        lds_fence();
        if (pos == 0) {
            keys[2 * p] = metric_key(c0);
            keys[2 * p + 1] = metric_key(c1);
        }
        lds_fence();
        const uint32_t own = *reinterpret_cast<const uint32_t *>(keys_wave + own_addr);
        const uint32_t own1 = own + 1u;
        uint32_t acc = 0;
        acc = rank_step<0x121>(acc, own, own1);
        acc = rank_step<0x122>(acc, own, own1);
        acc = rank_step<0x123>(acc, own, own1);
        acc = rank_step<0x124>(acc, own, own1);
        acc = rank_step<0x125>(acc, own, own1);
        acc = rank_step<0x126>(acc, own, own1);
        acc = rank_step<0x127>(acc, own, own1);
        acc = rank_step<0x128>(acc, own, own1);
        acc = rank_step<0x129>(acc, own, own1);
        acc = rank_step<0x12A>(acc, own, own1);
        acc = rank_step<0x12B>(acc, own, own1);
        acc = rank_step<0x12C>(acc, own, own1);
        acc = rank_step<0x12D>(acc, own, own1);
        acc = rank_step<0x12E>(acc, own, own1);
        acc = rank_step<0x12F>(acc, own, own1);
        const uint32_t cnt = (uint32_t)__popc(acc) + 1u;   // + itself
        uint64_t b = __ballot(cnt <= (uint32_t)L);
        uint32_t mine = (uint32_t)(b >> (16 * c)) & 0xFFFFu;
        if (sizeof(R) == 8) {
            const bool bad = __popc((uint32_t)b & 0xFFFFu) != L || __popc((uint32_t)(b >> 16) & 0xFFFFu) != L ||
                             __popc((uint32_t)(b >> 32) & 0xFFFFu) != L || __popc((uint32_t)(b >> 48)) != L;
            if (bad) {
                // a key tie across the boundary (or a true median tie) in some codeword: decide on the full metrics
                if (__popc(mine) != L) fl |= 0x4u;   // POLAR_FLAG_RERANK, this lane's codeword
                lds_fence();
                if (pos == 0) {
                    cand[p] = c0;
                    cand[8 + p] = c1;
                }
                lds_fence();
                const R me = cand_rank[lane & 15];
                int n = 0;
#pragma unroll
                for (int m = 0; m < 16; ++m) n += (cand_rank[m] <= me) ? 1 : 0;
                b = __ballot(n <= L);
                mine = (uint32_t)(b >> (16 * c)) & 0xFFFFu;
            }
        }
        return mine;
    }

    // true if every path of all four codewords keeps the branch its lambda favours (see k_scl_fast2's decide)
    __device__ __forceinline__ bool trivial_prune(R cb, R cw) const
    {
        uint32_t mx = metric_key(cb) & pos0_mask;
        mx = max(mx, (uint32_t)dpp_i<0x128>((int)mx));   // row_ror:8: the other path of this row of 16 lanes
        {
            auto a = __builtin_amdgcn_permlane16_swap(mx, mx, false, false);
            mx = max(a[0], a[1]);
        }
        {
            auto a = __builtin_amdgcn_permlane32_swap(mx, mx, false, false);
            mx = max(a[0], a[1]);
        }
        return __ballot(mx >= (metric_key(cw) | ~pos0_mask)) == 0ull;   // one compare per path instead of a min reduction
    }
    static __device__ __forceinline__ uint32_t sign_bit(double x) { return (uint32_t)__double2hiint(x) >> 31; }
    static __device__ __forceinline__ uint32_t sign_bit(float x) { return (uint32_t)__float_as_int(x) >> 31; }

    // ---- decision at leaf j = 8o + K; lambda valid at pos 0 ----
    template <int K>
    __device__ __forceinline__ void decide(int o, bool frozen, R lam)
    {
        const int j = 8 * o + K;
        uint32_t crcw = 0;
        if (CRC_ON && !frozen) crcw = crct[j];
        uint32_t bit = 0;
        const R tt = lut.tabv(lam);
        if (frozen) {
            PM += tt + negmax(lam);  // PHI(.,0)
        } else {
            if (logact < 3) {
                const R ph0 = tt + negmax(lam), ph1 = tt + posmax(lam);  // PHI(.,0), PHI(.,1)
                bit = (p >> logact) & 1;
                PM += bit ? ph1 : ph0;
                ++logact;
            } else {
                // the branch lambda favours costs T(|lambda|), the other one T(|lambda|) + |lambda| (SCL_1024.c:481-502)
                const R cb = PM + tt, cw = PM + (tt + absr(lam));
                const uint32_t lneg = sign_bit(lam);
                if (trivial_prune(cb, cw)) {
                    bit = (uint32_t)dpp_i<0xA0>((int)lneg);   // quad_perm [0,0,2,2]: pos 0 holds lambda
                    PM = cb;
                } else {
                    const R c0 = lneg ? cw : cb, c1 = lneg ? cb : cw;
                    const uint32_t mask = survivors(c0, c1);
                    const uint32_t m0 = mask & 0xFFu, m1 = mask >> 8;
                    const uint32_t m_both = m0 & m1, m_dead = ~(m0 | m1) & 0xFFu;
                    if (__popc(mask) < L) fl |= 0x1u;  // median tie in this lane's codeword
                    const bool s0 = (m0 >> p) & 1, s1 = (m1 >> p) & 1;
                    if (__ballot(m_dead != 0u) == 0ull) {
                        bit = (!s0 && s1) ? 1u : 0u;  // no codeword forks: every slot keeps exactly one branch
                        PM = bit ? c1 : c0;
                    } else {
                        // m-th both-survivor (ascending slot) forks into the m-th dead slot (:636-661), per codeword
                        const bool dead = !s0 && !s1;
                        const int myrank = __popc(m_dead & ((1u << p) - 1u));
                        const bool refilled = dead && (myrank < __popc(m_both));
                        const int sg = refilled ? (int)kth[m_both * 8 + myrank] : p;
                        const int sl = sg * 8 + gl;
                        const R c1s = __shfl(c1, sl);
                        ptr = __shfl(ptr, sl);
                        crc = __shfl(crc, sl);
                        bl0 = __shfl(bl0, sl);
                        if constexpr ((K & 4) == 0) {   // level 3, still to be read by g2
#pragma unroll
                            for (int r = 0; r < 4; ++r) A[4 + r] = __shfl(A[4 + r], sl);
                        }
                        if constexpr ((K & 2) == 0) { A[2] = __shfl(A[2], sl); A[3] = __shfl(A[3], sl); }   // level 2, read by g1
                        if constexpr ((K & 1) == 0) a1 = __shfl(a1, sl);                                     // level 1, read by g0
                        if (refilled) { bit = 1; PM = c1s; }
                        else if (s0) { bit = 0; PM = c0; }
                        else if (s1) { bit = 1; PM = c1; }
                        else { bit = 0; PM = c0; }  // tie rule: un-refilled dead slot continues as its 0-branch
                    }
                }
            }
            if (CRC_ON) crc ^= bit ? crcw : 0u;
        }
        set_bit_k<K>(o, bit);
    }

    // ---- the leading run of P all-frozen octets (leaves 0 .. 8P-1; 1 <= P <= 15), instead of octets 0 .. P-1 ----
    // As in k_scl_fast2: nothing has been decided, every partial sum is 0, so both children of every node are known as
    // soon as the node is (f(x, y) and y + x): the first 128-leaf subtree is evaluated as seven butterfly stages over one
    // 128-element LDS array per codeword (16 lanes, four butterflies per lane and stage), levels 8 and 7 above it once per
    // codeword into slot 0's scratch rows (every slot points there), the metric adds PHI(lambda_j, 0) in leaf order
    // (SCL_1024.c:601-604), and the registers are loaded with the nodes that contain leaf 8P at levels 4..6.
    __device__ __forceinline__ void frozen_prefix(int P)
    {
        const int w16 = p * 2 + pos, j0 = 8 * P;
        vm_drain();   // the top-left level written by the root step
        {
            const R *tl = tls();
            // slot 0's rows of this codeword in the lane-interleaved layout: element e belongs to lane c*2 + (e & 1)
            auto slot0 = [&](size_t base, int e) -> R * {
                return scr + fresh((unsigned)(base + at(e >> 6, (e & 63) >> 1) + c * 2 + (e & 1)));
            };
            R v8[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int e = w16 + 16 * k;
                v8[k] = chk(ld_sc(tl + e), ld_sc(tl + e + 256));
                *slot0(C::sc_l8, e) = v8[k];
            }
            lds_fence();
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const R v7 = chk(v8[k], v8[k + 8]);   // elements e and e + 128 of level 8
                *slot0(C::sc_l7, w16 + 16 * k) = v7;
                stg[w16 + 16 * k] = v7;
            }
            set_pa(8, 0);
            set_pa(7, 0);
        }
        lds_fence();
#pragma unroll
        for (int t = 6; t >= 0; --t) {
            const int h = 1 << t;
            int idx[4];
            R x[4], y[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int b = w16 + 16 * i;
                idx[i] = ((b >> t) << (t + 1)) | (b & (h - 1));
                x[i] = stg[idx[i]];
                y[i] = stg[idx[i] + h];
            }
            lds_fence();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                stg[idx[i]] = chk(x[i], y[i]);
                stg[idx[i] + h] = y[i] + x[i];
            }
            lds_fence();
            if (t >= 4) {   // node of level t that holds leaf j0: elements pos + 2r
                const R *node = stg + ((j0 >> t) << t) + pos;
#pragma unroll
                for (int r = 0; r < h / 2; ++r) A[h / 2 + r] = node[2 * r];
            }
        }
        // PHI(lambda_j, 0) of all 128 leaves, then the metric in leaf order
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const R lam = stg[w16 + 16 * k];
            stg[128 + w16 + 16 * k] = lut.tabv(lam) + negmax(lam);
        }
        lds_fence();
        R pm = PM;
#pragma unroll 8
        for (int j = 0; j < j0; ++j) pm += stg[128 + j];
        PM = pm;
        lds_fence();
    }

    // ---- octets whose first seven leaves are frozen: breadth-first (all partner bits are 0, every g is cL + cU) ----
    // Two level-1 nodes share each level-0 CHK: pos 0 evaluates f of one node, pos 1 f of the other.
    __device__ __forceinline__ void octet_frozen_prefix(int o, bool last_frozen)
    {
        // level 2: f half (leaves 0..3) and g half (leaves 4..7), elements pos and pos + 2 each
        const R xf0 = chks(A[4], A[6]), xf1 = chks(A[5], A[7]);
        const R xg0 = A[6] + A[4], xg1 = A[7] + A[5];
        // level 1: four nodes of two elements (element pos)
        const R ff = chks(xf0, xf1), gf = xf1 + xf0;   // leaves 0-1, 2-3
        const R fg = chks(xg0, xg1), gg = xg1 + xg0;   // leaves 4-5, 6-7
        // level 0
        const R pff = quadp<0xB1>(ff), pgf = quadp<0xB1>(gf), pfg = quadp<0xB1>(fg), pgg = quadp<0xB1>(gg);
        const R la = chks(pos ? pgf : ff, pos ? gf : pff);       // pos 0: lambda_0, pos 1: lambda_2
        const R lb = pos ? (gf + pgf) : (pff + ff);               // pos 0: lambda_1, pos 1: lambda_3
        const R lc = chks(pos ? pgg : fg, pos ? gg : pfg);       // pos 0: lambda_4, pos 1: lambda_6
        const R ld = pos ? (gg + pgg) : (pfg + fg);               // pos 0: lambda_5, pos 1: lambda_7
        const R pa_ = lut.tabv(la) + negmax(la), pb_ = lut.tabv(lb) + negmax(lb);   // PHI(lambda_k, 0)
        const R pc_ = lut.tabv(lc) + negmax(lc), pd_ = lut.tabv(ld) + negmax(ld);
        PM += pa_;
        PM += pb_;
        PM += quadp<0xB1>(pa_);
        PM += quadp<0xB1>(pb_);
        PM += pc_;
        PM += pd_;
        PM += quadp<0xB1>(pc_);
        bl0 &= ~0xFEu;
        if (last_frozen) {
            PM += quadp<0xB1>(pd_);
            set_bit_tail(8 * o + 7, 0u);
        } else {
            decide<7>(o, false, quadp<0xB1>(ld));
        }
    }

    // ---- the 8 leaves of octet o; A[4..7] hold the level-3 LLRs ----
    __device__ __forceinline__ R lam0() const { return chks(a1, quadp<0xB1>(a1)); }                      // f0, valid at pos 0
    __device__ __forceinline__ R lam1() const { return g_bit<R>(a1, quadp<0xB1>(a1), bl0, 1); }          // g0
    __device__ __forceinline__ void octet(int o, uint32_t fm)
    {
        // leaf 0: f2 f1 f0
        A[2] = chks(A[4], A[6]);
        A[3] = chks(A[5], A[7]);
        a1 = chks(A[2], A[3]);
        decide<0>(o, fm & 1, lam0());
        // leaf 1: g0
        decide<1>(o, (fm >> 1) & 1, lam1());
        // leaf 2: g1 f0
        a1 = g_bit<R>(A[2], A[3], bl0, 2 + pos);
        decide<2>(o, (fm >> 2) & 1, lam0());
        // leaf 3: g0
        decide<3>(o, (fm >> 3) & 1, lam1());
        // leaf 4: g2 f1 f0
        A[2] = g_bit<R>(A[4], A[6], bl0, 4 + pos);
        A[3] = g_bit<R>(A[5], A[7], bl0, 6 + pos);
        a1 = chks(A[2], A[3]);
        decide<4>(o, (fm >> 4) & 1, lam0());
        // leaf 5: g0
        decide<5>(o, (fm >> 5) & 1, lam1());
        // leaf 6: g1 f0
        a1 = g_bit<R>(A[2], A[3], bl0, 2 + pos);
        decide<6>(o, (fm >> 6) & 1, lam0());
        // leaf 7: g0
        decide<7>(o, (fm >> 7) & 1, lam1());
    }
};

template <typename R, typename IN, bool CRC_ON>
__global__ __launch_bounds__(256, (Fast4Cfg<R>::MIN_WAVES_PER_SIMD)) void k_scl_fast4(SclParams P)
{
#ifdef POLAR_STAMPS
    unsigned long long tsec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
    using D = Fast4Dec<R, IN, CRC_ON>;
    using C = Fast4Cfg<R>;
    constexpr int N = C::N, NW = C::NW, L = 8, CW = C::CW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6;
    unsigned char *base = smem + C::shared_bytes + (size_t)wave * C::per_wave;
    uint32_t *frz = reinterpret_cast<uint32_t *>(smem + C::off_frz);
    uint32_t *crct = reinterpret_cast<uint32_t *>(smem + C::off_crc);
    unsigned char *kth = smem + C::off_kth;

    Lut<R>::build(smem + C::off_lut, threadIdx.x, blockDim.x);
    for (int i = threadIdx.x; i < NW; i += blockDim.x) frz[i] = P.frozen[i];
    for (int i = threadIdx.x; i < N; i += blockDim.x) crct[i] = (CRC_ON && P.crc_tab) ? P.crc_tab[i] : 0u;
    for (int i = threadIdx.x; i < 256 * 8; i += blockDim.x) {
        const int m = i >> 3, k = i & 7;
        int idx = 0, seen = 0;
        for (int b = 0; b < 8; ++b)
            if ((m >> b) & 1) {
                if (seen == k) idx = b;
                ++seen;
            }
        kth[i] = (unsigned char)idx;
    }
    __syncthreads();

    D s;
    s.lane = threadIdx.x & 63;
    s.p = s.lane >> 3;
    s.c = (s.lane >> 1) & 3;
    s.pos = s.lane & 1;
    s.pos0_mask = s.pos == 0 ? 0xFFFFFFFFu : 0u;
    s.gl = s.lane & 7;
    s.lut.bind(smem + C::off_lut);
    s.crct = crct;
    s.kth = kth;
    s.blw = reinterpret_cast<uint32_t *>(base + C::off_bl) + s.c * 8 * NW;
    s.curw = reinterpret_cast<uint32_t *>(base + C::off_cw) + s.c * 8 * NW;
    s.cand = reinterpret_cast<R *>(base + C::off_cd) + s.c * 16;
    s.keys = reinterpret_cast<uint32_t *>(base + C::off_ky) + s.c * 16;
    s.stg = reinterpret_cast<R *>(base + C::off_sg) + s.c * 256;
    s.sigma = P.sigma;
    {   // rank network: lane (row, i) ranks candidate i = slot + 8*branch of codeword `row`, stored at keys[row][2*slot + branch]
        const int i = s.lane & 15, row = s.lane >> 4;
        s.rrow = row;
        s.keys_wave = base + C::off_ky;
        s.own_addr = 64 * row + 4 * (2 * (i & 7) + (i >> 3));
        s.cand_rank = reinterpret_cast<R *>(base + C::off_cd) + row * 16;
    }
    const int lane = s.lane, p = s.p, c = s.c, pos = s.pos;
    const int wave_global = blockIdx.x * C::WAVES + wave;
    const int waves_total = gridDim.x * C::WAVES;
    s.scr = reinterpret_cast<R *>(P.scratch) + (size_t)wave_global * C::scratch_elems;
    uint32_t *blw_wave = reinterpret_cast<uint32_t *>(base + C::off_bl);

    // leading all-frozen octets (at most 15: the run must end inside the first 128-leaf subtree)
    int lead = 0;
    while (lead < 15 && ((frz[lead >> 2] >> (8 * (lead & 3))) & 0xFFu) == 0xFFu) ++lead;
#ifdef POLAR_F4_NO_PREFIX
    lead = 0;
#endif
    lead = __builtin_amdgcn_readfirstlane(lead);

    for (int quad = wave_global; CW * quad < P.B; quad = next_job_wave(P.queue, quad, waves_total, (P.B + CW - 1) / CW)) {
        const int frame_raw = CW * quad + c;
        const bool live = frame_raw < P.B;
        const int frame = live ? frame_raw : P.B - 1;  // ragged tail: the idle quarters re-decode the last frame, no store
        s.src = reinterpret_cast<const IN *>(P.in) + (size_t)frame * N;
        {   // root f, single path per codeword: lanes of codeword c are w = p*2 + pos = 0..15
            R *t = s.tls();
            const int w = p * 2 + pos;
#pragma unroll 2
            for (int e = w; e < N / 2; e += 16) t[e] = s.chk(s.chv(e), s.chv(e + N / 2));
        }
        for (int w = lane; w < CW * 8 * NW; w += 64) blw_wave[w] = 0;
        lds_fence();

        s.PM = R(0);
        s.ptr = 0;
        for (int t = 4; t <= 8; ++t) s.set_pa(t, p);
        for (int t = 5; t <= 9; ++t) s.set_pb(t, p);
        s.crc = 0;
        s.bl0 = 0;
        s.a1 = R(0);
#pragma unroll
        for (int r = 0; r < C::NA; ++r) s.A[r] = R(0);
        s.fl = 0;
        s.logact = 0;
        uint32_t fword = 0;

        STAMP(2);
        int o_first = 0;
        if (lead > 0) {
            s.frozen_prefix(lead);
            STAMP(6);
            o_first = lead;
            fword = frz[o_first >> 2];
        }
        for (int o = o_first; o < N / 8; ++o) {
            if ((o & 3) == 0) fword = frz[o >> 2];
#ifdef POLAR_STAMPS
            {
                const int d = (o == 0) ? 10 : 3 + __builtin_ctz((unsigned)o);
                if (d >= 8) { s.from_top(o >= N / 16, d == 8); STAMP(0); }
                else if (d == 7) { s.from_l8(); STAMP(1); }
                else if (d == 6) { s.from_l7(); STAMP(7); }
                else if (d == 5) { s.template g_reg<5>(); STAMP(3); }
                else if (d == 4) { s.template g_reg<4>(); STAMP(3); }
                else { s.g3(); STAMP(3); }
                if (d > 5) s.template f_reg<5>();
                if (d > 4) s.template f_reg<4>();
                if (d > 3) s.template f_reg<3>();
                STAMP(4);
            }
#else
            s.octet_head(o);
#endif
            const uint32_t fm = (fword >> (8 * (o & 3))) & 0xFFu;
#ifndef POLAR_F4_NO_PREFIX
            if ((fm & 0x7Fu) == 0x7Fu) s.octet_frozen_prefix(o, fm == 0xFFu);
            else
#endif
                s.octet(o, fm);
            STAMP(5);
        }

        // ---- choose the path, per codeword (SCL_1024.c:667-674; CASCL_1024_L8.c:725-755) ----
        const bool pass = CRC_ON && (s.crc == 0);
        const uint64_t bp_ = __ballot(pass && pos == 0);
        // lanes of codeword c at pos 0 are lanes p*8 + 2c
        const uint64_t cwmask = 0x0101010101010101ull << (2 * c);
        const bool any = (bp_ & cwmask) != 0ull;
        int best = -1;
        R best_pm = R(0);
        for (int q = 0; q < L; ++q) {
            const R pq = __shfl(s.PM, q * 8 + 2 * c);
            const int okq = __shfl((int)(any ? pass : true), q * 8 + 2 * c);
            if (okq && (best < 0 || pq < best_pm)) {
                best = q;
                best_pm = pq;
            }
        }
        uint32_t fl = s.fl;
        if (any) fl |= 0x2u;
        // x_hat = root partial sums of the winner; u_hat = x_hat F^{(x)n}; this lane holds words w and w + 16, w = p*2 + pos
        lds_fence();
        {
            const int w = p * 2 + pos;
            uint32_t x0 = s.curw[best * NW + w], x1 = s.curw[best * NW + w + 16];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint32_t &x = h ? x1 : x0;
                x ^= (x >> 1) & 0x55555555u;
                x ^= (x >> 2) & 0x33333333u;
                x ^= (x >> 4) & 0x0F0F0F0Fu;
                x ^= (x >> 8) & 0x00FF00FFu;
                x ^= (x >> 16) & 0x0000FFFFu;
            }
#pragma unroll
            for (int hw = 1; hw < 16; hw <<= 1) {
                const int wo = w ^ hw;  // partner word; the lower one of the pair absorbs the upper one
                const int src_lane = (wo >> 1) * 8 + 2 * c + (wo & 1);
                const uint32_t o0 = __shfl(x0, src_lane), o1 = __shfl(x1, src_lane);
                if (!(w & hw)) { x0 ^= o0; x1 ^= o1; }
            }
            x0 ^= x1;   // hw = 16: words w and w + 16 are in the same lane
            if (live) {
                P.out_bits[(size_t)frame * NW + w] = x0;
                P.out_bits[(size_t)frame * NW + w + 16] = x1;
            }
        }
        if (live && p == 0 && pos == 0) {
            if (P.pm) P.pm[frame] = (double)best_pm;
            if (P.flags) P.flags[frame] = fl;
        }
        lds_fence();
        STAMP(6);
    }
#ifdef POLAR_STAMPS
    if (P.dbg && lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&P.dbg[i], tsec[i]);
#endif
}

}  // namespace polar
