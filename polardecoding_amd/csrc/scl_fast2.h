// scl_fast2.h -- CA-SCL / SCL, L = 8, N = 1024: TWO codewords per wavefront.
//
// k_scl_fast (scl_fast.h) is VALU-issue bound at 4 waves/SIMD, and most of its instructions -- the three
// narrowest LLR levels and the whole per-leaf decision (PHI, candidate ranking, fork bookkeeping) -- carry
// useful data in 8 of 64 lanes.  Here a path owns 4 lanes instead of 8 and the other half of every 8-lane
// path group belongs to a second, independent codeword:
//
//     lane = p*8 + c*4 + pos        p = path slot 0..7, c = codeword 0/1, pos = 0..3
//
// so every narrow-level and decision instruction serves two codewords.  Level t >= 2 holds 2^t/4 registers
// per lane (element e = pos + 4r, level t at A[2^t/4 ...]); levels 7, 8, the top-left level and the channel
// vector are in the per-wave global scratch / the input exactly as in k_scl_fast.  Everything that was
// wave-uniform scalar logic there and differs per codeword (survivor masks, fork matching) is per-lane here;
// the leaf schedule (frozen pattern) is the same for both codewords, so control flow stays uniform.
// Arithmetic is the same as k_scl_fast / the reference, operation for operation.
#pragma once
#include "scl_fast.h"

namespace polar {

template <typename R>
struct Fast2Cfg {
    static constexpr int NLOG = 10, N = 1024, NW = 32, TOP = 9, HI = 8, L = 8;
#ifndef POLAR_F2_L6S_F64
#define POLAR_F2_L6S_F64 0
#endif
#ifndef POLAR_F2_L6S_F32
#define POLAR_F2_L6S_F32 0
#endif
    // L6S: level 6 stays in the scratch as well (registers hold levels 2..5): 32 fewer VGPRs in f64
    static constexpr bool L6S = sizeof(R) == 8 ? (POLAR_F2_L6S_F64 != 0) : (POLAR_F2_L6S_F32 != 0);
    static constexpr int NA = L6S ? 16 : 32;   // levels 2..5 (or ..6), level t at offset 2^t/4
    static constexpr int WAVES = 4;
// selects of the by-product g steps by sign mask + v_bfi_b32 (1) or by v_cmp + v_cndmask (0): +-0 while the wavefronts took
// their jobs by a fixed stride (profiles/r03_ab_experiments.txt run 24), + 1.3 % since the work queue (run 34)
#ifndef POLAR_F2_GSEL_MASK
#define POLAR_F2_GSEL_MASK 1
#endif
#ifndef POLAR_F2_WAVES_F64
#define POLAR_F2_WAVES_F64 3
#endif
#ifndef POLAR_F2_WAVES_F32
#define POLAR_F2_WAVES_F32 4
#endif
    static constexpr int MIN_WAVES_PER_SIMD = sizeof(R) == 8 ? POLAR_F2_WAVES_F64 : POLAR_F2_WAVES_F32;
    static constexpr int NFA = HI - 3;  // pointer fields: LLR levels 4..8, then partial-sum levels 5..9
    // per-codeword scratch (elements of R)
    static constexpr size_t sc_l8 = 0;
    static constexpr size_t sc_l7 = sc_l8 + 8 * 256;
    static constexpr size_t sc_l6 = sc_l7 + 8 * 128;
    static constexpr size_t sc_tl = sc_l6 + 8 * 64;
    static constexpr size_t scratch_cw = sc_tl + 512;
    static constexpr size_t scratch_elems = 2 * scratch_cw;  // per wave
    // block-shared LDS
    static constexpr size_t off_lut = 0;
    static constexpr size_t off_frz = off_lut + ((Lut<R>::bytes + 15) / 16) * 16;
    static constexpr size_t off_crc = off_frz + 4 * NW;
    static constexpr size_t off_kth = off_crc + 4 * N;      // kth[mask][k] = index of the k-th set bit of mask (u8)
    static constexpr size_t off_stair = off_kth + 256 * 8;  // Stair<R>: thr[8], dlt[64] (chk_idx; experiments)
    static constexpr size_t shared_bytes = off_stair + ((Stair<R>::bytes + 15) / 16) * 16;
    // per-wave LDS
    static constexpr size_t off_bl = 0;                        // saved partial sums [2][8][NW]
    static constexpr size_t off_cw = off_bl + 4 * 2 * 8 * NW;  // working partial sums [2][8][NW]
    static constexpr size_t off_cd = off_cw + 4 * 2 * 8 * NW;  // candidates [2][16]
    static constexpr size_t off_ky = off_cd + sizeof(R) * 32;  // keys [2][16]
    static constexpr size_t off_sg = off_ky + 128;             // top-level staging [2][256]
    static constexpr size_t per_wave = off_sg + sizeof(R) * 512;
    static constexpr size_t total = shared_bytes + WAVES * per_wave;
};

template <int QP>
__device__ __forceinline__ double quadp(double x) { return quad_lanes<QP>(x); }
template <int QP>
__device__ __forceinline__ float quadp(float x) { return quad_lanes<QP>(x); }

template <typename R, typename IN, bool CRC_ON>
struct Fast2Dec {
    using C = Fast2Cfg<R>;
    static constexpr int N = C::N, NW = C::NW, TOP = C::TOP, HI = C::HI, L = 8, NFA = C::NFA;

    R A[C::NA];      // levels 2..6
    R a1;            // level 1 (pos 0, 1)
#ifdef POLAR_STAMPS_DECIDE
    unsigned long long dt_rank = 0, n_rank = 0, dt_info = 0, n_info = 0;   // diagnostic: time / count of ranked steps, of all phase-2 steps
#endif
    R bp_s, bp_d, bp_ts, bp_td;
    // Upper bound (+ 0.65) of the eight path metrics of the lane's codeword, the same in all its lanes, and whether it
    // is current (wave-uniform).  See decide_t: the cheap sufficient form of the trivial-prune test.
    R mb65;
    bool mb_ok;   // by-products of the last level-0 check node: x + y, x - y, T(|x + y|), T(|x - y|)
    R PM;            // valid at pos 0
    uint32_t ptr, crc, bl0, fl;
    int logact;
    int p, c, pos, lane, gl;   // gl = c*4 + pos: lane offset inside a path group
    int own_addr, oth_addr;    // rank network: byte addresses into keys[]
    uint32_t pos0_mask;        // ~0 in the lane that holds its path's metric (pos 0), else 0
    int cand_addr;
    Lut<R> lut;
#ifdef POLAR_F2_IDX
    Stair<R> st;
#endif
    R *cand, *stg;
    uint32_t *blw, *curw, *keys;   // this lane's codeword slice of the per-wave arrays
    const uint32_t *crct;
    const unsigned char *kth;
    // Scratch and input rows are read and written through buffer instructions: a wave-uniform resource descriptor
    // (4 SGPRs) plus a 32-bit per-lane byte offset.  The two codewords of a wavefront differ by a constant offset, so no
    // lane holds a 64-bit address: round 2's per-lane pointers were spilled and re-loaded from the stack in front of every
    // load of the scratch-level steps (a dependent memory round trip per load, in the same in-order vmcnt queue).
    __amdgpu_buffer_rsrc_t rs_scr;   // this wavefront's scratch (both codewords)
    __amdgpu_buffer_rsrc_t rs_in;    // the input rows of the wavefront's two codewords
    unsigned cscr;                   // element offset of the lane's codeword in the scratch: c * scratch_cw
    unsigned csrc;                   // element offset of the lane's codeword's input row: 0 or N
    static __device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes)
    {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
    }
    typedef unsigned u2_t __attribute__((ext_vector_type(2)));
    // scratch rows: sc1 loads (not served from a stale L1 line: other lanes of the wavefront wrote them), plain stores
    static __device__ __forceinline__ double ld_buf(__amdgpu_buffer_rsrc_t r, unsigned eoff, double, int aux)
    {
        return aux ? __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, eoff * 8u, 0, 16))
                   : __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, eoff * 8u, 0, 0));
    }
    static __device__ __forceinline__ float ld_buf(__amdgpu_buffer_rsrc_t r, unsigned eoff, float, int aux)
    {
        return aux ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, eoff * 4u, 0, 16))
                   : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, eoff * 4u, 0, 0));
    }
    static __device__ __forceinline__ void st_buf(__amdgpu_buffer_rsrc_t r, unsigned eoff, double v)
    {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_t, v), r, eoff * 8u, 0, 0);
    }
    static __device__ __forceinline__ void st_buf(__amdgpu_buffer_rsrc_t r, unsigned eoff, float v)
    {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, eoff * 4u, 0, 0);
    }
    // "pointer" into the wavefront's scratch: an element offset; q + k, q[k] = v and ld_sc(q) read like the pointer code
    struct SPtr {
        const Fast2Dec *d;
        unsigned off;
        struct Ref {
            const Fast2Dec *d;
            unsigned off;
            __device__ __forceinline__ void operator=(R v) const { st_buf(d->rs_scr, off, v); }
        };
        __device__ __forceinline__ SPtr operator+(int k) const { return SPtr{d, off + (unsigned)k}; }
        __device__ __forceinline__ Ref operator[](int k) const { return Ref{d, off + (unsigned)k}; }
    };
    static __device__ __forceinline__ R ld_sc(SPtr q) { return ld_buf(q.d->rs_scr, q.off, R(0), 1); }
    double sigma;

    __device__ __forceinline__ int pa(int t) const { return (ptr >> (3 * (t - 4))) & 7; }
    __device__ __forceinline__ void set_pa(int t, int v) { ptr = (ptr & ~(7u << (3 * (t - 4)))) | ((uint32_t)v << (3 * (t - 4))); }
    __device__ __forceinline__ int pb(int t) const { return (ptr >> (3 * (NFA + t - 5))) & 7; }
    __device__ __forceinline__ void set_pb(int t, int v) { ptr = (ptr & ~(7u << (3 * (NFA + t - 5)))) | ((uint32_t)v << (3 * (NFA + t - 5))); }
    // wide steps: the one-round-trip table form, as in the serial chains.  Round 2 had the compact two-round-trip form
    // here in f64; with the by-product octets (octet()) the one-round-trip form measures +2.2 % (same-box A/B, round 3),
    // without them +-0; in f32 it was already +2.4 %.
    __device__ __forceinline__ R chk(R a, R b) const
    {
#ifdef POLAR_F2_IDX
        return chk_idx<R>(a, b, st);
#endif
#ifdef POLAR_F2_WIDE_CHK2   // the compact two-round-trip form in the wide f64 steps (round 2's choice; see below)
        if constexpr (sizeof(R) == 8) return chk_lut<R>(a, b, lut);
#endif
        return chk_lut1<R>(a, b, lut);
    }
    // the narrow levels inside an octet are serial chains: the one-round-trip table form (four more issue slots,
    // one LDS latency less) measured +1.8 % there; POLAR_F2_CHK2 selects the compact form everywhere
#if defined(POLAR_F2_IDX) && POLAR_F2_IDX == 2
    __device__ __forceinline__ R chks(R a, R b) const { return chk_idx<R>(a, b, st); }
#elif defined(POLAR_F2_CHK2)
    __device__ __forceinline__ R chks(R a, R b) const { return chk_lut<R>(a, b, lut); }
#else
    __device__ __forceinline__ R chks(R a, R b) const { return chk_lut1<R>(a, b, lut); }
#endif
    __device__ __forceinline__ R chv(int e) const
    {
        double v = (double)ld_buf(rs_in, csrc + (unsigned)e, IN(0), 0);
        if (sigma > 0) v = llr_from_y(v, sigma);
        return (R)v;
    }
    // The row addresses are recomputed where they are used (a few instructions, 16 times per frame) instead of
    // being hoisted out of the frame loop into long-lived 64-bit registers: the empty asm stops the hoisting.
    static __device__ __forceinline__ unsigned fresh(unsigned off)
    {
        __asm__ volatile("" : "+v"(off));
        return off;
    }
    __device__ __forceinline__ SPtr l8(int slot) const { return SPtr{this, fresh(cscr + (unsigned)(C::sc_l8 + slot * 256))}; }
    __device__ __forceinline__ SPtr l7(int slot) const { return SPtr{this, fresh(cscr + (unsigned)(C::sc_l7 + slot * 128))}; }
    __device__ __forceinline__ SPtr l6s(int slot) const { return SPtr{this, fresh(cscr + (unsigned)(C::sc_l6 + slot * 64))}; }
    __device__ __forceinline__ SPtr tls() const { return SPtr{this, fresh(cscr + (unsigned)C::sc_tl)}; }

    // ---- register levels: f on own data ----
    template <int T>  // T in [2, 5]: level T from level T+1
    __device__ __forceinline__ void f_reg()
    {
        constexpr int RO = (1 << T) / 4;
        if constexpr (T == 5 && C::L6S) {
            vm_drain();
            const SPtr q = l6s(p) + pos;
#pragma unroll
            for (int r = 0; r < RO; ++r) A[RO + r] = chk(ld_sc(q + 4 * r), ld_sc(q + 4 * r + 32));
        } else {
#pragma unroll
            for (int r = 0; r < RO; ++r) A[RO + r] = chk(A[2 * RO + r], A[3 * RO + r]);
        }
        if constexpr (T >= 4) set_pa(T, p);
    }
    // ---- register levels: g from the owner's level T+1 (bpermute), T in {4, 5} ----
    template <int T>
    __device__ __forceinline__ void g_reg()
    {
        constexpr int RO = (1 << T) / 4;
        const int sl = pa(T + 1) * 8 + gl;
        uint32_t w;
        if constexpr (T == 5) w = blw[pb(5) * NW + 1] >> pos;   // level 5: word 1, bit e = pos + 4r
        else w = bl0 >> (16 + pos);                              // level 4: bits 16 + e
        if constexpr (T == 5 && C::L6S) {
            vm_drain();
            const SPtr q = l6s(pa(6)) + pos;
#pragma unroll
            for (int r = 0; r < RO; ++r) A[RO + r] = g_bit<R>(ld_sc(q + 4 * r), ld_sc(q + 4 * r + 32), w, 4 * r);
        } else {
#pragma unroll
            for (int r = 0; r < RO; ++r) {
                const R x = __shfl(A[2 * RO + r], sl), y = __shfl(A[3 * RO + r], sl);
                A[RO + r] = g_bit<R>(x, y, w, 4 * r);
            }
        }
        set_pa(T, p);
    }
    __device__ __forceinline__ void g3()  // level 3 (eager) from the owner's level 4
    {
        const int sl = pa(4) * 8 + gl;
        const uint32_t w = bl0 >> (8 + pos);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const R x = __shfl(A[4 + r], sl), y = __shfl(A[6 + r], sl);
            A[2 + r] = g_bit<R>(x, y, w, 4 * r);
        }
    }

    // ---- scratch levels ----
    // The level-6 value of pass rr (element pos + 4 rr).  Registers: the 16 level-6 registers act as a shift
    // register -- after the 16 passes of a step the value of pass rr sits in A[16 + rr] -- so the pass loops can
    // stay rolled without a dynamically indexed register and without a round trip through memory.
    __device__ __forceinline__ void push_l6(R v, SPtr o6, int e0)
    {
        if constexpr (C::L6S) {
            o6[e0] = v;
        } else {
#pragma unroll
            for (int r = 16; r < 31; ++r) A[r] = A[r + 1];
            A[31] = v;
        }
    }
    __device__ __forceinline__ void load_l6() { set_pa(6, p); }
    // d >= 8 (octets 0, 32, 64, 96): level 8 from the top level (f, or g when gstep), f down to level 6.
    // Pass rr: level-8 elements e0 + 64k (k < 4), level-7 elements e0, e0 + 64, level-6 element e0 = pos + 4 rr.
    // The top-level operands are the same for all eight paths of a codeword, so the codeword's 32 lanes fetch
    // them once per chunk of 4 passes (16 segments of 16 consecutive elements), park them in an LDS staging
    // buffer and every path reads them from there; the next chunk's loads are in flight during the compute.
    __device__ __forceinline__ R top_src(bool right, int idx, int q) const
    {
        // staging index idx = seg*16 + within; seg = k + 4*h: right: h selects ch offset {0,256,512,768};
        // left: h in {0,1} selects tl offset {0,256}
        const int seg = idx >> 4, within = idx & 15;
        const int e = 16 * q + within + 64 * (seg & 3) + 256 * (seg >> 2);
        return right ? chv(e) : ld_sc(tls() + e);
    }
    __device__ __forceinline__ void from_top(bool right, bool gstep)
    {
        vm_drain();
        const uint32_t *bt = blw + pb(TOP) * NW + 16;  // beta_9: words 16..31
        const uint32_t *bh = blw + pb(HI) * NW + 8;    // beta_8: words 8..15
        const SPtr o8 = l8(p), o7 = l7(p), o6 = l6s(p);
        const int w32 = (int)fresh((unsigned)(p * 4 + pos));   // lane index inside the codeword (recomputed here: the staged
                                                               // elements' offsets are not worth a register each across the frame loop)
        const int nld = right ? 8 : 4;                 // staged elements per lane and chunk
        R *pre = A + 4;    // levels 2..5 are dead during this step (recomputed below): reuse their registers
#pragma unroll
        for (int m = 0; m < 8; ++m) pre[m] = (m < nld) ? top_src(right, w32 + 32 * m, 0) : R(0);
        for (int q = 0; q < 4; ++q) {
            lds_fence();
#pragma unroll
            for (int m = 0; m < 8; ++m)
                if (m < nld) stg[w32 + 32 * m] = pre[m];
            lds_fence();
            if (q < 3) {
#pragma unroll
                for (int m = 0; m < 8; ++m) pre[m] = (m < nld) ? top_src(right, w32 + 32 * m, q + 1) : R(0);
            }
#ifndef POLAR_F2_TOP_UNROLL
#define POLAR_F2_TOP_UNROLL 1
#endif
#pragma unroll POLAR_F2_TOP_UNROLL
            for (int i = 0; i < 4; ++i) {
                const int rr = 4 * q + i;
                const int e0 = pos + 4 * rr;
                const int sh = 4 * (rr & 7);
                const int wq = rr >> 3;
                const int wi = 4 * i + pos;  // position inside a staged segment
                R *v8 = A + 12;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = e0 + 64 * k;
                    R x, y;
                    if (right) {
                        const uint32_t w0 = bt[2 * k + wq] >> pos, w1 = bt[8 + 2 * k + wq] >> pos;
                        x = g_bit<R>(stg[(k + 0) * 16 + wi], stg[(k + 8) * 16 + wi], w0, sh);    // ch[e], ch[e+512]
                        y = g_bit<R>(stg[(k + 4) * 16 + wi], stg[(k + 12) * 16 + wi], w1, sh);   // ch[e+256], ch[e+768]
                    } else {
                        x = stg[(k + 0) * 16 + wi];   // tl[e]
                        y = stg[(k + 4) * 16 + wi];   // tl[e+256]
                    }
                    if (gstep) v8[k] = g_bit<R>(x, y, bh[2 * k + wq] >> pos, sh);
                    else v8[k] = chk(x, y);
                    o8[e] = v8[k];
                }
                const R v70 = chk(v8[0], v8[2]), v71 = chk(v8[1], v8[3]);
                o7[e0] = v70;
                o7[e0 + 64] = v71;
                push_l6(chk(v70, v71), o6, e0);
            }
        }
        set_pa(8, p);
        set_pa(7, p);
        load_l6();
    }
    // d == 7: g to level 7 from the owner's level 8, f to level 6.  Four passes per chunk: their 16 loads go out
    // together into the (dead at this point) level-6 registers.
    __device__ __forceinline__ void from_l8()
    {
        vm_drain();
        const SPtr s8 = l8(pa(8));
        const uint32_t *b7 = blw + pb(7) * NW + 4;  // beta_7: words 4..7
        const SPtr o7 = l7(p), o6 = l6s(p);
#ifndef POLAR_F2_CP_F64
#define POLAR_F2_CP_F64 2
#endif
        constexpr int CP = sizeof(R) == 8 ? POLAR_F2_CP_F64 : 4;   // passes per chunk (4 loads each); f64: 8 in flight is the measured optimum
        for (int q = 0; q < 16 / CP; ++q) {
            R *in = A;       // levels 2..5: dead here, recomputed by the f chain below
#pragma unroll
            for (int i = 0; i < CP; ++i) {
                const int e0 = pos + 4 * (CP * q + i);
#pragma unroll
                for (int k = 0; k < 4; ++k) in[4 * i + k] = ld_sc(s8 + e0 + 64 * k);
            }
#pragma unroll
            for (int i = 0; i < CP; ++i) {
                const int rr = CP * q + i;
                const int e0 = pos + 4 * rr;
                const int sh = 4 * (rr & 7);
                const int wq = rr >> 3;
                const R v70 = g_bit<R>(in[4 * i], in[4 * i + 2], b7[wq] >> pos, sh);
                const R v71 = g_bit<R>(in[4 * i + 1], in[4 * i + 3], b7[2 + wq] >> pos, sh);
                o7[e0] = v70;
                o7[e0 + 64] = v71;
                push_l6(chk(v70, v71), o6, e0);
            }
        }
        set_pa(7, p);
        load_l6();
    }
    __device__ __forceinline__ void from_l7()  // d == 6
    {
        vm_drain();
        const SPtr s7 = l7(pa(7)) + pos;
        const uint32_t *b6 = blw + pb(6) * NW + 2;  // beta_6: words 2, 3
        const uint32_t w0 = b6[0] >> pos, w1 = b6[1] >> pos;
        const SPtr o6 = l6s(p) + pos;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const R x = ld_sc(s7 + 4 * r), y = ld_sc(s7 + 4 * r + 64);
            const R v = g_bit<R>(x, y, r < 8 ? w0 : w1, 4 * (r & 7));
            if constexpr (C::L6S) o6[4 * r] = v;
            else A[16 + r] = v;
        }
        set_pa(6, p);
    }

    // head of octet o: g at level d = ctz(8o) (root f for o = 0), f chain down to level 3 (A[2], A[3])
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

    // ---- partial sums (identical per path to k_scl_fast) ----
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
            for (int w = pos; w < nw; w += 4) {
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
            for (int w = pos; w < nw; w += 4) blw[p * NW + nw + w] = curw[p * NW + w];
            set_pb(t, p);
            lds_fence();
        }
    }

    // ---- survivors (SCL_1024.c:612-633) for both codewords; returns this lane's codeword's 16-bit mask ----
    // Rows 0,1 of the wave rank codeword 0, rows 2,3 codeword 1: row h of a pair compares its 16 keys with the
    // row rotated by 8h .. 8h+7; permlane16_swap adds the two partial counts.
    __device__ __forceinline__ uint32_t survivors(R c0, R c1)
    {
        lds_fence();
        if (pos == 0) {
            keys[2 * p] = metric_key(c0);
            keys[2 * p + 1] = metric_key(c1);
        }
        lds_fence();
        const unsigned char *kb = reinterpret_cast<const unsigned char *>(keys) - 64 * c;  // wave base of keys[2][16]
        const uint32_t own = *reinterpret_cast<const uint32_t *>(kb + own_addr);
        const uint32_t oth = *reinterpret_cast<const uint32_t *>(kb + oth_addr);
        // key_m - key_own - 1 is negative iff key_m <= key_own (keys < 2^31); one v_alignbit shifts that sign
        // bit into the accumulator: two full-rate instructions per comparison instead of cmp + addc
        const uint32_t own1 = own + 1u;
        uint32_t acc = (oth - own1) >> 31;
        acc = __builtin_amdgcn_alignbit(acc, (uint32_t)dpp_i<0x121>((int)oth) - own1, 31);
        acc = __builtin_amdgcn_alignbit(acc, (uint32_t)dpp_i<0x122>((int)oth) - own1, 31);
        acc = __builtin_amdgcn_alignbit(acc, (uint32_t)dpp_i<0x123>((int)oth) - own1, 31);
        acc = __builtin_amdgcn_alignbit(acc, (uint32_t)dpp_i<0x124>((int)oth) - own1, 31);
        acc = __builtin_amdgcn_alignbit(acc, (uint32_t)dpp_i<0x125>((int)oth) - own1, 31);
        acc = __builtin_amdgcn_alignbit(acc, (uint32_t)dpp_i<0x126>((int)oth) - own1, 31);
        acc = __builtin_amdgcn_alignbit(acc, (uint32_t)dpp_i<0x127>((int)oth) - own1, 31);
        uint32_t cnt = __popc(acc);
        {
            auto r = __builtin_amdgcn_permlane16_swap(cnt, cnt, false, false);
            cnt = r[0] + r[1];
        }
        uint64_t b = __ballot(cnt <= (uint32_t)L);
        uint32_t m_a = (uint32_t)b & 0xFFFFu, m_b = (uint32_t)(b >> 32) & 0xFFFFu;
        if (sizeof(R) == 8 && (__popc(m_a) != L || __popc(m_b) != L)) {
            // a key tie across the boundary (or a true median tie) in either codeword: decide on the full metrics
            if (__popc(c ? m_b : m_a) != L) fl |= 0x4u;   // POLAR_FLAG_RERANK, this lane's codeword
            lds_fence();
            if (pos == 0) {
                cand[p] = c0;
                cand[8 + p] = c1;
            }
            lds_fence();
            const R *cb = reinterpret_cast<const R *>(reinterpret_cast<const unsigned char *>(cand) - (int)sizeof(R) * 16 * c);
            const R *mycand = reinterpret_cast<const R *>(reinterpret_cast<const unsigned char *>(cb) + cand_addr);
            const R mine = mycand[lane & 15];
            int n = 0;
#pragma unroll
            for (int m = 0; m < 16; ++m) n += (mycand[m] <= mine) ? 1 : 0;
            b = __ballot(n <= L);
            m_a = (uint32_t)b & 0xFFFFu;
            m_b = (uint32_t)(b >> 32) & 0xFFFFu;
        }
        return c ? m_b : m_a;
    }

    // ---- "every path keeps the branch its lambda favours" (see decide): true if so for BOTH codewords ----
    // cb / cw = the path's metric with the favoured / the other branch, cb <= cw, valid at pos 0.
    __device__ __forceinline__ bool trivial_prune(R cb, R cw) const
    {
        uint32_t mx;
        return trivial_prune(cb, cw, mx);
    }
    // mx_out: the largest favoured key of the lane's codeword (valid in every lane)
    __device__ __forceinline__ bool trivial_prune(R cb, R cw, uint32_t &mx_out) const
    {
        // the largest favoured key of the lane's codeword (the other lanes of a path take no part: 0), then ONE compare per
        // path: "max favoured < min other" <=> every path's other key is above that maximum
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
        mx_out = mx;
        return __ballot(mx >= (metric_key(cw) | ~pos0_mask)) == 0ull;
    }
    // smallest value whose key is above `key`: an upper bound of every metric with a key <= `key`
    static __device__ __forceinline__ double above_key(uint32_t key, double) { return __hiloint2double((int)(key + 1u), 0); }
    static __device__ __forceinline__ float above_key(uint32_t key, float) { return __int_as_float((int)(key + 1u)); }
    static __device__ __forceinline__ uint32_t sign_bit(double x) { return (uint32_t)__double2hiint(x) >> 31; }
    static __device__ __forceinline__ uint32_t sign_bit(float x) { return (uint32_t)__float_as_int(x) >> 31; }

    // ---- decision at leaf j = 8o + K; lambda valid at pos 0 ----
    template <int K>
    __device__ __forceinline__ void decide(int o, bool frozen, R lam) { decide_t<K>(o, frozen, lam, lut.tabv(lam)); }
    // tt = T(|lambda|) (SCL_1024.c:352-359), looked up by the caller -- or, at the odd leaves, a by-product of the
    // check node that produced the even leaf's lambda (see octet())
    template <int K>
    __device__ __forceinline__ void decide_t(int o, bool frozen, R lam, R tt)
    {
        const int j = 8 * o + K;
        POLAR_MARK("d2_begin");
        uint32_t crcw = 0;
#ifdef POLAR_F2_CRC_BRANCH
        if (CRC_ON && !frozen) crcw = crct[j];
#else
        if (CRC_ON) crcw = crct[j];   // the table holds 0 for frozen leaves (make_crc_table): no branch around the read
#endif
        uint32_t bit = 0;
        if (frozen) {
            PM += tt + negmax(lam);  // PHI(.,0)
            mb_ok = false;
        } else {
            if (logact < 3) {
                mb_ok = false;
                const R ph0 = tt + negmax(lam), ph1 = tt + posmax(lam);  // PHI(.,0), PHI(.,1)
                bit = (p >> logact) & 1;
                PM += bit ? ph1 : ph0;
                ++logact;
            } else {
                POLAR_MARK("d2_phase2");
#ifdef POLAR_STAMPS_DECIDE
                const unsigned long long t_i0 = __builtin_amdgcn_s_memtime();
#endif
                // PHI of the branch lambda favours is T(|lambda|), of the other one T(|lambda|) + |lambda|
                // (SCL_1024.c:481-502; T + 0 is T, so these ARE c0 / c1 in the order the sign of lambda says).
                const uint32_t lneg = sign_bit(lam);   // lambda = +-0: cb == cw, never trivial, c0 == c1 below
#ifdef POLAR_F2_BOUND
                // The same test on bounds that need no reduction over the paths.  mb65 >= M + 0.65 with M >= every
                // metric of the codeword, so every favoured candidate PM + T(|lambda|) <= mb65 (T <= 0.65, rounding is
                // monotone); every other candidate PM + (T + |lambda|) >= PM + |lambda| (T >= 0).  If the key of mb65 is
                // below the key of PM + |lambda| on every path, max favoured < min other: the prune is trivial.  After
                // it every metric has grown by at most 0.65, so mb65 + 0.65 bounds the next leaf.  When the bound is
                // stale (after a frozen leaf or a ranked step) or too loose, the exact test below decides and renews it.
                if (mb_ok && __ballot(metric_key(mb65) >= (metric_key(PM + absr(lam)) | ~pos0_mask)) == 0ull) {
                    bit = (uint32_t)dpp_i<0x00>((int)lneg);
                    PM = PM + tt;
                    mb65 = mb65 + R(0.65);
                } else {
#endif
                const R cb = PM + tt, cw = PM + (tt + absr(lam));
                // Most information leaves (85 % at 1-3 dB) prune trivially: every path keeps its favoured branch.
                // That is certain when the largest of the eight favoured keys is below the smallest of the eight
                // others (the 8 favoured candidates are then the 8 smallest of the 16, all strictly below the median
                // of SCL_1024.c:619-633), and three max/min steps over the path lanes show it -- without the key
                // exchange through LDS, the rank network and the fork bookkeeping.  Both codewords must qualify.
                uint32_t mxk;
                if (trivial_prune(cb, cw, mxk)) {
                    bit = (uint32_t)dpp_i<0x00>((int)lneg);   // quad_perm [0,0,0,0]: pos 0 holds lambda
                    PM = cb;
#ifdef POLAR_F2_BOUND
                    mb65 = above_key(mxk, R(0)) + R(0.65);   // every new metric (= cb) has a key <= mxk
                    mb_ok = true;
#endif
                } else {
#ifdef POLAR_F2_BOUND
                mb_ok = false;
#endif
#ifdef POLAR_STAMPS_DECIDE
                const unsigned long long t_r0 = __builtin_amdgcn_s_memtime();
#endif
                const R c0 = lneg ? cw : cb, c1 = lneg ? cb : cw;
                const uint32_t mask = survivors(c0, c1);
                POLAR_MARK("d2_rank_end");
                const uint32_t m0 = mask & 0xFFu, m1 = mask >> 8;
                const uint32_t m_both = m0 & m1, m_dead = ~(m0 | m1) & 0xFFu;
                if (__popc(mask) < L) fl |= 0x1u;  // median tie in this lane's codeword
                const bool s0 = (m0 >> p) & 1, s1 = (m1 >> p) & 1;
                if (__ballot(m_dead != 0u) == 0ull) {
                    bit = (!s0 && s1) ? 1u : 0u;  // no codeword forks: every slot keeps exactly one branch
                    PM = bit ? c1 : c0;
                } else {
                    POLAR_MARK("d2_fork");
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
#ifdef POLAR_F2_NO_BYPROD
                    if constexpr ((K & 4) == 0) { A[2] = __shfl(A[2], sl); A[3] = __shfl(A[3], sl); }  // level 3, read by g2
                    if constexpr ((K & 2) == 0) A[1] = __shfl(A[1], sl);                                // level 2, read by g1
                    if constexpr ((K & 1) == 0) a1 = __shfl(a1, sl);                                    // level 1, read by g0
#else
                    // what the lower-node steps still to come in this octet read: the sum / difference pairs of the
                    // check nodes above them (octet())
                    if constexpr ((K & 4) == 0) { A[2] = __shfl(A[2], sl); A[3] = __shfl(A[3], sl); }  // s2, d2: g at level 2 (leaf 4)
                    if constexpr ((K & 2) == 0) { A[1] = __shfl(A[1], sl); a1 = __shfl(a1, sl); }       // s1, d1: g at level 1
                    if constexpr ((K & 1) == 0) { bp_d = __shfl(bp_d, sl); bp_td = __shfl(bp_td, sl); } // a forked copy continues with bit 1: -d0, T(|d0|)
#endif
                    if (refilled) { bit = 1; PM = c1s; }
                    else if (s0) { bit = 0; PM = c0; }
                    else if (s1) { bit = 1; PM = c1; }
                    else { bit = 0; PM = c0; }  // tie rule: un-refilled dead slot continues as its 0-branch
                }
#ifdef POLAR_STAMPS_DECIDE
                __asm__ volatile("" :: "v"(bit), "v"(PM));
                dt_rank += __builtin_amdgcn_s_memtime() - t_r0;
                ++n_rank;
#endif
                }
#ifdef POLAR_F2_BOUND
                }
#endif
#ifdef POLAR_STAMPS_DECIDE
                __asm__ volatile("" :: "v"(bit), "v"(PM));
                dt_info += __builtin_amdgcn_s_memtime() - t_i0;
                ++n_info;
#endif
            }
            POLAR_MARK("d2_fork_end");
            if (CRC_ON) crc ^= bit ? crcw : 0u;
        }
        POLAR_MARK("d2_setbit");
        set_bit_k<K>(o, bit);
        POLAR_MARK("d2_end");
    }

    // ---- the leading run of P all-frozen octets (leaves 0 .. 8P-1; 1 <= P <= 15), instead of octets 0 .. P-1 ----
    // Nothing has been decided yet, so every partial sum below the first level-7 node is 0 and BOTH children of
    // every node are known the moment the node is: f(x, y) and y + x.  The whole 128-leaf subtree is therefore
    // evaluated as seven butterfly stages over one 128-element array per codeword (32 lanes, two butterflies per
    // lane and stage) instead of octet by octet on eight replicated paths.  The values are those of the lazy
    // recursion, operation for operation; the path metric adds PHI(lambda_j, 0) for j = 0 .. 8P-1 in that order
    // (SCL_1024.c:601-604).  Afterwards the registers hold the nodes that contain leaf 8P at levels 4..6.
    // Levels 8 and 7 above it are computed once, by the codeword's 32 lanes together, into slot 0's scratch rows,
    // and every slot points there (the eight slots are replicas of one path until the first information leaf).
    __device__ __forceinline__ void frozen_prefix(int P)
    {
        const int w32 = p * 4 + pos, j0 = 8 * P;
        vm_drain();   // the top-left level written by the root step
        {
            const SPtr tl = tls();
            const SPtr o8 = l8(0), o7 = l7(0);
            R v8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = w32 + 32 * k;
                v8[k] = chk(ld_sc(tl + e), ld_sc(tl + e + 256));
                o8[e] = v8[k];
            }
            lds_fence();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const R v7 = chk(v8[k], v8[k + 4]);   // elements e and e + 128 of level 8
                o7[w32 + 32 * k] = v7;
                stg[w32 + 32 * k] = v7;
            }
            set_pa(8, 0);
            set_pa(7, 0);
        }
        lds_fence();
#pragma unroll
        for (int t = 6; t >= 0; --t) {
            const int h = 1 << t;
            int idx[2];
            R x[2], y[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int b = w32 + 32 * i;
                idx[i] = ((b >> t) << (t + 1)) | (b & (h - 1));
                x[i] = stg[idx[i]];
                y[i] = stg[idx[i] + h];
            }
            lds_fence();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                stg[idx[i]] = chk(x[i], y[i]);
                stg[idx[i] + h] = y[i] + x[i];
            }
            lds_fence();
            if (t >= 4) {   // node of level t that holds leaf j0: elements pos + 4r
                const R *node = stg + ((j0 >> t) << t) + pos;
                if (t == 6 && C::L6S) {
                    const SPtr o6 = l6s(p) + pos;
#pragma unroll
                    for (int r = 0; r < 16; ++r) o6[4 * r] = node[4 * r];
                } else {
#pragma unroll
                    for (int r = 0; r < h / 4; ++r) A[h / 4 + r] = node[4 * r];
                }
            }
        }
        // PHI(lambda_j, 0) of all 128 leaves, then the metric in leaf order
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const R lam = stg[w32 + 32 * k];
            stg[128 + w32 + 32 * k] = lut.tabv(lam) + negmax(lam);
        }
        lds_fence();
        R pm = PM;
#pragma unroll 8
        for (int j = 0; j < j0; ++j) pm += stg[128 + j];
        PM = pm;
        lds_fence();
    }

    // ---- octets whose first seven leaves are frozen: breadth-first (all partner bits are 0) ----
    __device__ __forceinline__ void octet_frozen_prefix(int o, bool last_frozen)
    {
        // level 2: f half (leaves 0..3) and g half (leaves 4..7), element pos each
        R xf = chks(A[2], A[3]), xg = A[3] + A[2];
        // level 1: lanes pos 0,1 = f results, lanes 2,3 = g results, for both halves
        R yf = quadp<0x4E>(xf), yg = quadp<0x4E>(xg);  // lane ^ 2
        {
            const R ff = chks(xf, yf), gf = xf + yf, fg = chks(xg, yg), gg = xg + yg;
            xf = (pos & 2) ? gf : ff;
            xg = (pos & 2) ? gg : fg;
        }
        // level 0: lambda_k in lane pos = k & 3 of (k < 4 ? xf : xg)
        yf = quadp<0xB1>(xf); yg = quadp<0xB1>(xg);    // lane ^ 1
        R lf, lg;
        {
            const R ff = chks(xf, yf), gf = xf + yf, fg = chks(xg, yg), gg = xg + yg;
            lf = (pos & 1) ? gf : ff;
            lg = (pos & 1) ? gg : fg;
        }
        const R pf = lut.tabv(lf) + negmax(lf), pg = lut.tabv(lg) + negmax(lg);  // PHI(lambda_k, 0)
        mb_ok = false;
        PM += pf;
        PM += quadp<0x55>(pf);  // lane 1 of the quad
        PM += quadp<0xAA>(pf);  // lane 2
        PM += quadp<0xFF>(pf);  // lane 3
        PM += pg;
        PM += quadp<0x55>(pg);
        PM += quadp<0xAA>(pg);
        bl0 &= ~0xFEu;
        if (last_frozen) {
            PM += quadp<0xFF>(pg);
            set_bit_tail(8 * o + 7, 0u);
        } else {
            decide<7>(o, false, quadp<0xFF>(lg));
        }
    }

    // Check node that keeps what it computed on the way: CHK(x, y) = sgn min + (T(|s|) - T(|d|)) with s = x + y,
    // d = x - y (SCL_1024.c:350-373).  The lower-node update of the same pair is cL +- cU = y +- x (:412-416), i.e.
    // s for partner bit 0 and y - x = -d for bit 1 -- same operands, one rounding, so the value is the one g_bit()
    // computes (a zero may come out as -0 where y - x gives +0; a zero lambda decides nothing: PHI adds no penalty
    // either way, it never prunes trivially, and c0 == c1 in the ranking).  T(|s|), T(|d|) are the staircase values the
    // odd leaf's PHI needs.  So inside an octet every g step and every second table look-up is a by-product.
    struct ChkBp { R v, s, d, ts, td; };
    __device__ __forceinline__ ChkBp chk_bp(R x, R y) const
    {
        ChkBp r;
        r.s = x + y;
        r.d = x - y;
        r.ts = lut.tabv(r.s);
        r.td = lut.tabv(r.d);
        r.v = xor_sign(minabs(x, y), x, y) + (r.ts - r.td);
        return r;
    }
    // s or -d by the partner bit at position sh of w
    static __device__ __forceinline__ R g_sel(R sum, R dif, uint32_t w, int sh)
    {
#if !POLAR_F2_GSEL_MASK
        const bool b = (w >> sh) & 1u;
        return b ? -dif : sum;
#else
        // bit -> all-ones mask (one v_bfe_i32), then v_bfi per word: no compare, no v_cndmask
        const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)w, sh, 1);
        return Lut<R>::sel_mask(m, -dif, sum);
#endif
    }
    template <int K>   // leaves K (even) and K + 1 from the level-1 pair in a1 (pos 0: x, pos 1: y)
    __device__ __forceinline__ void leaf_pair(int o, uint32_t fm, R x1)
    {
        const ChkBp q = chk_bp(x1, quadp<0xB1>(x1));
        bp_d = q.d;
        bp_td = q.td;
        decide<K>(o, (fm >> K) & 1, q.v);
        // leaf K + 1: g0 with the bit just decided (bit 1 of bl0, set_bit_k<even>); a slot refilled by a fork took
        // bp_d / bp_td of its source and continues with bit 1, every other slot still has its own q.s / q.ts
#if !POLAR_F2_GSEL_MASK
        const bool b1 = (bl0 >> 1) & 1u;
        decide_t<K + 1>(o, (fm >> (K + 1)) & 1, b1 ? -bp_d : q.s, b1 ? bp_td : q.ts);
#else
        const uint32_t m1 = (uint32_t)__builtin_amdgcn_sbfe((int)bl0, 1, 1);
        decide_t<K + 1>(o, (fm >> (K + 1)) & 1, Lut<R>::sel_mask(m1, -bp_d, q.s), Lut<R>::sel_mask(m1, bp_td, q.ts));
#endif
    }
    __device__ __forceinline__ void octet(int o, uint32_t fm)
    {
#ifndef POLAR_F2_NO_BYPROD
        // level 2: f of (A[2], A[3]); its sum / difference stay in A[2], A[3] for the g step at leaf 4
        {
            const R s2 = A[2] + A[3], d2 = A[2] - A[3];
            const R v2 = xor_sign(minabs(A[2], A[3]), A[2], A[3]) + (lut.tabv(s2) - lut.tabv(d2));
            A[2] = s2;
            A[3] = d2;
            // level 1: f; sum / difference in (A[1], a1) for the g steps at leaves 2 and 6
            const R y2 = quadp<0x4E>(v2);
            const R s1 = v2 + y2, d1 = v2 - y2;
            const R v1 = xor_sign(minabs(v2, y2), v2, y2) + (lut.tabv(s1) - lut.tabv(d1));
            A[1] = s1;
            a1 = d1;
            leaf_pair<0>(o, fm, v1);
        }
        leaf_pair<2>(o, fm, g_sel(A[1], a1, bl0, 2 + pos));
        {
            const R v2 = g_sel(A[2], A[3], bl0, 4 + pos);   // level 2, g
            const R y2 = quadp<0x4E>(v2);
            const R s1 = v2 + y2, d1 = v2 - y2;
            const R v1 = xor_sign(minabs(v2, y2), v2, y2) + (lut.tabv(s1) - lut.tabv(d1));
            A[1] = s1;
            a1 = d1;
            leaf_pair<4>(o, fm, v1);
        }
        leaf_pair<6>(o, fm, g_sel(A[1], a1, bl0, 2 + pos));
        return;
#endif
        // leaf 0: f2 f1 f0
        A[1] = chks(A[2], A[3]);
        a1 = chks(A[1], quadp<0x4E>(A[1]));
        decide<0>(o, fm & 1, chks(a1, quadp<0xB1>(a1)));
        // leaf 1: g0
        decide<1>(o, (fm >> 1) & 1, g_bit<R>(a1, quadp<0xB1>(a1), bl0, 1));
        // leaf 2: g1 f0
        a1 = g_bit<R>(A[1], quadp<0x4E>(A[1]), bl0, 2 + pos);
        decide<2>(o, (fm >> 2) & 1, chks(a1, quadp<0xB1>(a1)));
        // leaf 3: g0
        decide<3>(o, (fm >> 3) & 1, g_bit<R>(a1, quadp<0xB1>(a1), bl0, 1));
        // leaf 4: g2 f1 f0
        A[1] = g_bit<R>(A[2], A[3], bl0, 4 + pos);
        a1 = chks(A[1], quadp<0x4E>(A[1]));
        decide<4>(o, (fm >> 4) & 1, chks(a1, quadp<0xB1>(a1)));
        // leaf 5: g0
        decide<5>(o, (fm >> 5) & 1, g_bit<R>(a1, quadp<0xB1>(a1), bl0, 1));
        // leaf 6: g1 f0
        a1 = g_bit<R>(A[1], quadp<0x4E>(A[1]), bl0, 2 + pos);
        decide<6>(o, (fm >> 6) & 1, chks(a1, quadp<0xB1>(a1)));
        // leaf 7: g0
        decide<7>(o, (fm >> 7) & 1, g_bit<R>(a1, quadp<0xB1>(a1), bl0, 1));
    }
};

template <typename R, typename IN, bool CRC_ON>
__global__ __launch_bounds__(256, (Fast2Cfg<R>::MIN_WAVES_PER_SIMD)) void k_scl_fast2(SclParams P)
{
#ifdef POLAR_STAMPS
    unsigned long long tsec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#ifdef POLAR_STAMPS_XCC
    unsigned long long njobs = 0;
#endif
#endif
    using D = Fast2Dec<R, IN, CRC_ON>;
    using C = Fast2Cfg<R>;
    constexpr int N = C::N, NW = C::NW, L = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef POLAR_F2_WAVE_VGPR
    const int wave = threadIdx.x >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: the per-wave LDS base is scalar
#endif
    unsigned char *base = smem + C::shared_bytes + (size_t)wave * C::per_wave;
    uint32_t *frz = reinterpret_cast<uint32_t *>(smem + C::off_frz);
    uint32_t *crct = reinterpret_cast<uint32_t *>(smem + C::off_crc);
    unsigned char *kth = smem + C::off_kth;

    Lut<R>::build(smem + C::off_lut, threadIdx.x, blockDim.x);
#ifdef POLAR_F2_IDX
    Stair<R>::build(smem + C::off_stair, threadIdx.x, blockDim.x);
#endif
    for (int i = threadIdx.x; i < NW; i += blockDim.x) frz[i] = P.frozen[i];
    if (CRC_ON)
        for (int i = threadIdx.x; i < N; i += blockDim.x) crct[i] = P.crc_tab[i];
    for (int i = threadIdx.x; i < 256 * 8; i += blockDim.x) {
        int m = i >> 3, k = i & 7, idx = 0, seen = 0;
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
    s.c = (s.lane >> 2) & 1;
    s.pos = s.lane & 3;
    s.pos0_mask = s.pos == 0 ? 0xFFFFFFFFu : 0u;
    s.gl = s.lane & 7;
    s.lut.bind(smem + C::off_lut);
#ifdef POLAR_F2_IDX
    s.st.bind(smem + C::off_stair);
#endif
    s.crct = crct;
    s.kth = kth;
    s.blw = reinterpret_cast<uint32_t *>(base + C::off_bl) + s.c * 8 * NW;
    s.curw = reinterpret_cast<uint32_t *>(base + C::off_cw) + s.c * 8 * NW;
    s.cand = reinterpret_cast<R *>(base + C::off_cd) + s.c * 16;
    s.keys = reinterpret_cast<uint32_t *>(base + C::off_ky) + s.c * 16;
    s.stg = reinterpret_cast<R *>(base + C::off_sg) + s.c * 256;
    s.sigma = P.sigma;
    {   // rank network: lane (row, i): rows 0,1 -> codeword 0, rows 2,3 -> codeword 1; candidate i is
        // (slot i & 7, branch i >> 3) stored at keys[cw][2*slot + branch]
        const int i = s.lane & 15, row = s.lane >> 4, cwr = row >> 1, h = row & 1;
        const int jj = (i + 8 * h) & 15;
        s.own_addr = 64 * cwr + 4 * (2 * (i & 7) + (i >> 3));
        s.oth_addr = 64 * cwr + 4 * (2 * (jj & 7) + (jj >> 3));
        s.cand_addr = (int)sizeof(R) * 16 * cwr;
    }
    const int lane = s.lane, p = s.p, c = s.c, pos = s.pos;
    const int wave_global = blockIdx.x * C::WAVES + wave;
    const int waves_total = gridDim.x * C::WAVES;
    R *scr_wave = reinterpret_cast<R *>(P.scratch) + (size_t)wave_global * C::scratch_elems;
    s.rs_scr = D::make_rsrc(scr_wave, (unsigned)(C::scratch_elems * sizeof(R)));
    s.cscr = (unsigned)c * (unsigned)C::scratch_cw;

    // leading all-frozen octets (at most 15: the run must end inside the first 128-leaf subtree)
    int lead = 0;
    while (lead < 15 && ((frz[lead >> 2] >> (8 * (lead & 3))) & 0xFFu) == 0xFFu) ++lead;
#ifdef POLAR_F2_NO_PREFIX
    lead = 0;
#endif
    lead = __builtin_amdgcn_readfirstlane(lead);

    // jobs (pairs of frames): the first one by the wavefront's index, the others from the launch's work queue
    for (int pair = wave_global; 2 * pair < P.B;) {
        const int frame_raw = 2 * pair + c;
        const bool live = frame_raw < P.B;
        const int frame = live ? frame_raw : P.B - 1;  // odd tail: the idle half re-decodes the last frame, no store
        s.rs_in = D::make_rsrc(reinterpret_cast<const IN *>(P.in) + (size_t)(2 * pair) * N, (unsigned)(2 * N * sizeof(IN)));
        s.csrc = (unsigned)(frame - 2 * pair) * (unsigned)N;
        {   // root f, single path per codeword: lanes of codeword c are w = p*4 + pos = 0..31
            const auto t = s.tls();
            const int w = p * 4 + pos;
#pragma unroll 2
            for (int e = w; e < N / 2; e += 32) t[e] = s.chk(s.chv(e), s.chv(e + N / 2));
        }
        for (int w = lane; w < 2 * 8 * NW; w += 64) (s.blw - c * 8 * NW)[w] = 0;
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
        s.mb65 = R(0);
        s.mb_ok = false;
        uint32_t fword = 0;

        STAMP(2);
        int o_first = 0;
        if (lead > 0) {
            s.frozen_prefix(lead);
            STAMP(4);
            o_first = lead;
            fword = frz[o_first >> 2];
        }
        for (int o = o_first; o < N / 8; ++o) {
            if ((o & 3) == 0) fword = frz[o >> 2];
#ifdef POLAR_STAMPS_HEADS   // finer split of the octet heads: 0 from_top, 1 from_l8, 7 from_l7, 3 register g steps, 2 f chains
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
                STAMP(2);
            }
#else
            s.octet_head(o);
            if (o == 0 || (o & 7) == 0) STAMP(7); else STAMP(3);
#endif
            const uint32_t fm = (fword >> (8 * (o & 3))) & 0xFFu;
            if ((fm & 0x7Fu) == 0x7Fu) { s.octet_frozen_prefix(o, fm == 0xFFu); STAMP(4); }
            else { s.octet(o, fm); STAMP(5); }
        }

        // ---- choose the path, per codeword (SCL_1024.c:667-674; CASCL_1024_L8.c:725-755) ----
        const bool pass = CRC_ON && (s.crc == 0);
        const uint64_t bp_ = __ballot(pass && pos == 0);
        // lanes of codeword c at pos 0 are lanes p*8 + 4c: bits 4c, 8+4c, ...
        const uint64_t cwmask = 0x0101010101010101ull << (4 * c);
        const bool any = (bp_ & cwmask) != 0ull;
        int best = -1;
        R best_pm = R(0);
        for (int q = 0; q < L; ++q) {
            const R pq = __shfl(s.PM, q * 8 + 4 * c);
            const int okq = __shfl((int)(any ? pass : true), q * 8 + 4 * c);
            if (okq && (best < 0 || pq < best_pm)) {
                best = q;
                best_pm = pq;
            }
        }
        uint32_t fl = s.fl;
        if (any) fl |= 0x2u;
        // x_hat = root partial sums of the winner; u_hat = x_hat F^{(x)n}; word index w = p*4 + pos per codeword
        lds_fence();
        {
            const int w = p * 4 + pos;
            uint32_t x = s.curw[best * NW + w];
            x ^= (x >> 1) & 0x55555555u;
            x ^= (x >> 2) & 0x33333333u;
            x ^= (x >> 4) & 0x0F0F0F0Fu;
            x ^= (x >> 8) & 0x00FF00FFu;
            x ^= (x >> 16) & 0x0000FFFFu;
#pragma unroll
            for (int hw = 1; hw < NW; hw <<= 1) {
                const int wo = w ^ hw;  // partner word; the lower one of the pair absorbs the upper one
                const uint32_t ov = __shfl(x, (wo >> 2) * 8 + 4 * c + (wo & 3));
                if (!(w & hw)) x ^= ov;
            }
            if (live) P.out_bits[(size_t)frame * NW + w] = x;
        }
        if (live && p == 0 && pos == 0) {
            if (P.pm) P.pm[frame] = (double)best_pm;
            if (P.flags) P.flags[frame] = fl;
        }
        lds_fence();
        STAMP(6);
        pair = next_job_wave(P.queue, pair, waves_total, (P.B + 1) >> 1);
#ifdef POLAR_STAMPS_XCC
        ++njobs;
#endif
    }
#ifdef POLAR_STAMPS
#ifdef POLAR_STAMPS_XCC   // diagnostic: jobs per XCD (bucket = XCC_ID) or per shader engine (POLAR_STAMPS_XCC == 2), x 1000
    {
        unsigned xcc, hw;
        __asm__ volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        __asm__ volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned bucket = (POLAR_STAMPS_XCC == 2) ? ((hw >> 13) & 7u) : (xcc & 7u);
        if (POLAR_STAMPS_XCC == 3) {   // histogram of jobs per wavefront around the mean: <= mean-3, -2, -1, 0, +1, +2, +3, >= +4 (x 1000 x jobs)
            const int mean = (P.B / 2) / waves_total;
            bucket = (unsigned)min(max((int)njobs - mean + 3, 0), 7);
        }
        if (POLAR_STAMPS_XCC == 4) bucket = (hw >> 8) & 7u;    // CU_ID (low three bits)
        if (POLAR_STAMPS_XCC == 5) bucket = (hw >> 4) & 3u;    // SIMD_ID
        for (int i = 0; i < 8; ++i) tsec[i] = (i == (int)bucket) ? 1000ull * njobs : 0ull;
    }
#endif
#ifdef POLAR_STAMPS_DECIDE   // buckets 0 / 1: time and count (x 1000) of the ranked steps, 7 / 6: of all phase-2 information leaves
    tsec[0] = s.dt_rank; tsec[1] = s.n_rank * 1000; tsec[7] += s.dt_info; tsec[6] = s.n_info * 1000;
#endif
    if (P.dbg && lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&P.dbg[i], tsec[i]);
#endif
}

}  // namespace polar
