// polar_hip.hip -- C ABI (include/polar_hip.h) over the hand-written gfx950 kernels.
// Host side: code construction (frozen set, CRC table), kernel dispatch, buffers, stream.  The kernels and their launch
// code live in the k_*.hip translation units (polar_host.h declares what they export).
#include "polar_host.h"
#ifdef POLAR_TESTING
#include "../../include/polar_hip_testing.h"
#endif

#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>

#include "count_kernel.h"
#include "gen_kernel.h"

namespace {

const int kQ5G[1024] = {
#include "q5g_table.inc"
};

// Reliability order when the caller gives none: the 5G sequence restricted to < N for N <= 1024
// (what every reference program hard-codes, SC_1024.c:42-91 / SC_128.c:41-51); for N > 1024 the
// reference has no table and this build uses the polarization-weight (beta-expansion)
// order, beta = 2^(1/4) -- "parity unpinned" for those sizes.
std::vector<int> default_order(int N)
{
    std::vector<int> q;
    if (N <= 1024) {
        for (int i = 0; i < 1024; ++i)
            if (kQ5G[i] < N) q.push_back(kQ5G[i]);
    } else {
        int n = 0;
        while ((1 << n) < N) ++n;
        std::vector<std::pair<double, int>> w(N);
        const double beta = std::pow(2.0, 0.25);
        for (int i = 0; i < N; ++i) {
            double s = 0;
            for (int b = 0; b < n; ++b)
                if ((i >> b) & 1) s += std::pow(beta, b);
            w[i] = {s, i};
        }
        std::stable_sort(w.begin(), w.end());
        for (auto &x : w) q.push_back(x.second);
    }
    return q;
}

// D^i mod g(D) for i = position of leaf j in the reliability-ordered CRC codeword
// (CRcheck, CASCL_1024_L8.c:569-598: C[i] = u_hat[I[i]], long division by g, pass iff remainder 0;
// the remainder is linear in the bits, so it can be accumulated as bits are decided).
std::vector<uint32_t> make_crc_table(int N, int r, const std::vector<int> &taps, const std::vector<int> &I)
{
    std::vector<uint32_t> tab(N, 0u);
    uint64_t glow = 0;
    for (int t : taps)
        if (t < r) glow |= 1ull << t;
    const uint64_t top = 1ull << r;
    uint64_t rem = 1;  // D^0
    if (r == 0) return tab;
    for (size_t i = 0; i < I.size(); ++i) {
        tab[I[i]] = (uint32_t)rem;
        rem <<= 1;
        if (rem & top) rem = (rem ^ top) ^ glow;
    }
    return tab;
}

static bool sc_lanes_ok(const polar_ctx *c, size_t B)
{
    return c->cfg.algo == POLAR_ALGO_SC && !c->force_generic && c->cfg.N <= 2048 && B >= 64;
}

bool fast_ok(const polar_ctx *c, int in_is_f32)
{
    const polar_cfg &g = c->cfg;
    if (c->force_generic || c->force_spill) return false;
    if (g.algo != POLAR_ALGO_SCL && g.algo != POLAR_ALGO_CASCL) return false;
    if (g.L != 8 || (g.N != 1024 && g.N != 128)) return false;
    if (g.dtype == POLAR_F64 && in_is_f32) return false;
    return true;
}

int decode_device_impl(polar_ctx *c, const void *d_in, int in_is_f32, double sigma, size_t B, uint32_t *d_bits,
                       double *d_pm, uint32_t *d_flags, const uint32_t *d_frozen)
{
    if (!c || !d_in || !d_bits) return POLAR_EINVAL;
    if (B == 0) return POLAR_OK;
    if (B > 0x7fffffffull) return POLAR_EINVAL;
    const polar_cfg &g = c->cfg;
    const bool f32 = g.dtype == POLAR_F32;
    if (g.algo == POLAR_ALGO_BP) {
        polar::BpParams P{};
        P.in = d_in; P.sigma = sigma; P.out_bits = d_bits; P.frozen = d_frozen;
        P.N = g.N; P.n = c->n; P.B = (int)B; P.iters = g.bp_iters;
        if (d_pm) HIP_TRY(c, hipMemsetAsync(d_pm, 0, B * sizeof(double), c->stream));
        if (d_flags) HIP_TRY(c, hipMemsetAsync(d_flags, 0, B * sizeof(uint32_t), c->stream));
        return polar_tu::bp(c, P, f32, in_is_f32 != 0);
    }
    polar::SclParams P{};
    P.in = d_in; P.sigma = sigma; P.out_bits = d_bits; P.pm = d_pm; P.flags = d_flags;
    P.frozen = d_frozen;
    P.crc_tab = (g.algo == POLAR_ALGO_CASCL) ? c->d_crc_tab : nullptr;
    P.N = g.N; P.n = c->n; P.B = (int)B;
    P.sc_mode = (g.algo == POLAR_ALGO_SC) ? 1 : 0;
    P.dbg = nullptr;
    P.scratch = nullptr;
    P.queue = nullptr;
#ifdef POLAR_STAMPS   // diagnostic builds only (tools/): per-section cycle sums, see debug_stamps.inc
#include "debug_stamps.inc"
#endif
    const bool in32 = in_is_f32 != 0;
    if (sc_lanes_ok(c, B)) return polar_tu::sc_lanes(c, P, f32, in32);
    if (fast_ok(c, in_is_f32)) {
        const bool crc = g.algo == POLAR_ALGO_CASCL;
#ifdef POLAR_TESTING
        if (P.N == 1024 && c->use_fast4) return polar_tu::scl_fast4(c, P, f32, in32, crc);
#endif
        if (P.N == 1024 && c->use_fast2) return polar_tu::scl_fast2(c, P, f32, in32, crc);
        return polar_tu::scl_fast(c, P, f32, in32, crc);
    }
    // scl_big.h for SCL / CA-SCL with N >= 512, L >= 2 (shapes without a tuned kernel); else the generic kernel,
    // LDS-resident when the levels fit (160 KB per CU), global-scratch variant otherwise
    if (c->logL >= 1 && !c->force_generic && !P.sc_mode && P.n >= 9)
        return f32 ? polar_tu::scl_big_f32(c, P, in32) : polar_tu::scl_big_f64(c, P, in32);
    return polar_tu::scl_generic(c, P, f32, in32);
}

// the kernel instantiation decode_device_impl will launch for this ctx (mirrors its choices)
void refresh_kernel_name(polar_ctx *c)
{
    const polar_cfg &g = c->cfg;
    const char *ty = g.dtype == POLAR_F32 ? "float" : "double";
    char nm[128];
    if (g.algo == POLAR_ALGO_BP)
        snprintf(nm, sizeof nm, (g.N == 1024 && !c->force_generic) ? "k_bp_r4<%s>" : (g.N == 128 && !c->force_generic) ? "k_bp_w128<%s>" : "k_bp<%s>", ty);
    else
        snprintf(nm, sizeof nm, "k_scl_generic<%s,L=%d>", ty, g.L);
    if (g.algo != POLAR_ALGO_BP && g.algo != POLAR_ALGO_SC && !c->force_generic && c->n >= 9 && g.L >= 2)
        snprintf(nm, sizeof nm, "k_scl_big<%s,L=%d>", ty, g.L);
    if (g.algo == POLAR_ALGO_SC && !c->force_generic && g.N <= 2048)
        snprintf(nm, sizeof nm, "k_sc_lanes<%s> (batches of 64+; k_scl_generic below)", ty);
    if (fast_ok(c, g.dtype == POLAR_F32))
        snprintf(nm, sizeof nm, "k_scl_fast%s<%s,N=%d,L=8>", (g.N == 1024 && c->use_fast4) ? "4" : (g.N == 1024 && c->use_fast2) ? "2" : "", ty, g.N);
    c->kernel_name = nm;
}

std::vector<uint32_t> pack_mask(const unsigned char *m, int N, bool invert)
{
    std::vector<uint32_t> w(N / 32, 0u);
    for (int j = 0; j < N; ++j)
        if ((m[j] != 0) != invert) w[j >> 5] |= 1u << (j & 31);
    return w;
}

// packed decisions -> the reference's int u_hat[N] (0/1), eight bits per table look-up
void unpack_words(const uint32_t *w, int NW, int *out)
{
    struct Row { int v[8]; };
    static const std::vector<Row> tab = [] {
        std::vector<Row> t(256);
        for (int b = 0; b < 256; ++b)
            for (int k = 0; k < 8; ++k) t[(size_t)b].v[k] = (b >> k) & 1;
        return t;
    }();
    for (int i = 0; i < NW; ++i) {
        const uint32_t x = w[i];
        std::memcpy(out + 32 * i, &tab[x & 255u], sizeof(Row));
        std::memcpy(out + 32 * i + 8, &tab[(x >> 8) & 255u], sizeof(Row));
        std::memcpy(out + 32 * i + 16, &tab[(x >> 16) & 255u], sizeof(Row));
        std::memcpy(out + 32 * i + 24, &tab[x >> 24], sizeof(Row));
    }
}

// helper threads per direction of the host pipeline (staging of the caller's rows / unpacking of the decisions)
#ifndef POLAR_HOST_THREADS
#define POLAR_HOST_THREADS 6   // 4 -> 6: end_to_end 4.4 -> 4.6-4.7 M frames/s; 8 and 12 no more (run 35)
#endif
int host_batch(polar_ctx *c, const double *in, double sigma, const unsigned char *frozen_mask, size_t B,
               int *u_hat, double *pm_out, unsigned *flags)
{
    if (!c || !in || !u_hat) return POLAR_EINVAL;
    DeviceGuard guard(c->cfg.device);
    if (B == 0) return POLAR_OK;
    const int N = c->cfg.N, NW = c->NW;
    const uint32_t *d_frozen = c->d_frozen;
    if (frozen_mask) {
        if (c->cfg.algo == POLAR_ALGO_CASCL) return POLAR_EINVAL;
        std::vector<uint32_t> w = pack_mask(frozen_mask, N, false);
        if (!c->d_frozen_override) HIP_TRY(c, hipMalloc(&c->d_frozen_override, NW * sizeof(uint32_t)));
        HIP_TRY(c, hipMemcpyAsync(c->d_frozen_override, w.data(), NW * sizeof(uint32_t), hipMemcpyHostToDevice,
                                  c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));  // w goes out of scope
        d_frozen = c->d_frozen_override;
    }
    int rc;
    if ((rc = ensure(c, c->pm, B * sizeof(double)))) return rc;
    if ((rc = ensure(c, c->flags, B * sizeof(uint32_t)))) return rc;
    // Chunked pipeline: while chunk k is decoded, chunk k+1 crosses PCIe on a second stream and the decisions of
    // chunk k-1 are unpacked to the caller's int array by helper threads.  The input is pageable caller memory: a
    // hipMemcpyAsync from it is a single-threaded staging copy inside the runtime (about 18 GB/s) that blocks this thread.
    // Batches of several chunks are therefore staged HERE, by helper threads, into pinned buffers, and cross PCIe as true
    // asynchronous DMA.
    // chunk: 16384 frames, but at most 128 MiB of input (N = 1024: 16384 frames; N = 4096: 4096), so that the two pinned
    // staging buffers and the two device buffers stay at 256 MiB each whatever the block length
#ifndef POLAR_HOST_CHUNK
#define POLAR_HOST_CHUNK 16384
#endif
    const size_t CH = std::max<size_t>(256, std::min<size_t>(POLAR_HOST_CHUNK, ((size_t)128 << 20) / ((size_t)N * sizeof(double))));
    // Chunk boundaries.  Big batches ramp up and down (CH/8, CH/4, CH/2, CH ... CH, CH/2, CH/4, CH/8): nothing overlaps the
    // staging of the first chunk nor the copy-out and unpacking of the last one, so those two are small.
    std::vector<size_t> off{0};
    {
        std::vector<size_t> head, tail;
        size_t left = B;
        if (B >= 6 * CH && CH >= 2048) {
            for (size_t d = 8; d >= 2; d /= 2) {
                head.push_back(CH / d);
                tail.insert(tail.begin(), CH / d);
                left -= 2 * (CH / d);
            }
        }
        for (size_t h : head) off.push_back(off.back() + h);
        while (left > 0) {
            const size_t nfc = std::min(CH, left);
            off.push_back(off.back() + nfc);
            left -= nfc;
        }
        for (size_t t : tail) off.push_back(off.back() + t);
    }
    const size_t nch = off.size() - 1;
    const size_t chf = std::min(B, CH);
    for (int i = 0; i < 2; ++i) {
        if ((rc = ensure(c, c->in2[i], chf * N * sizeof(double)))) return rc;
        if ((rc = ensure(c, c->bits2[i], chf * NW * sizeof(uint32_t)))) return rc;
    }
    if (c->h_bits_cap < chf * NW * sizeof(uint32_t)) {
        c->h_bits_cap = 0;   // a failure below leaves "nothing allocated", not the old size over freed pointers
        for (int i = 0; i < 2; ++i) {
            if (c->h_bits[i]) HIP_TRY(c, hipHostFree(c->h_bits[i]));
            c->h_bits[i] = nullptr;
            HIP_TRY(c, hipHostMalloc((void **)&c->h_bits[i], chf * NW * sizeof(uint32_t), hipHostMallocDefault));
        }
        c->h_bits_cap = chf * NW * sizeof(uint32_t);
    }
    if (!c->copy_stream) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_in[i], hipEventDisableTiming));
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_free[i], hipEventDisableTiming));
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_out[i], hipEventDisableTiming));
        }
    }
    const bool staged = nch >= 3;
    if (staged && c->h_in_cap < chf * N * sizeof(double)) {
        c->h_in_cap = 0;
        for (int i = 0; i < 2; ++i) {
            if (c->h_in[i]) HIP_TRY(c, hipHostFree(c->h_in[i]));
            c->h_in[i] = nullptr;
            HIP_TRY(c, hipHostMalloc((void **)&c->h_in[i], chf * N * sizeof(double), hipHostMallocDefault));
        }
        c->h_in_cap = chf * N * sizeof(double);
    }
    auto stage_chunk = [&](size_t k) {    // caller's rows of chunk k -> pinned h_in[k & 1], four threads
        const size_t f0 = off[k], nf = off[k + 1] - off[k];
        const double *src = in + f0 * (size_t)N;
        double *dst = c->h_in[k & 1];
        const unsigned nthr = POLAR_HOST_THREADS;
        auto part = [=](unsigned t) {
            const size_t a = nf * t / nthr, b = nf * (t + 1) / nthr;
            std::memcpy(dst + a * (size_t)N, src + a * (size_t)N, (b - a) * (size_t)N * sizeof(double));
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nthr; ++t) pool.emplace_back(part, t);
        part(0);
        for (auto &th : pool) th.join();
    };
    std::thread worker;
    auto unpack_chunk = [&](size_t k) {   // decisions of chunk k: pinned words -> caller's int u_hat[][N]
        const size_t f0 = off[k], nf = off[k + 1] - off[k];
        const uint32_t *hb = c->h_bits[k & 1];
        const unsigned nthr = (unsigned)std::max<size_t>(1, std::min<size_t>(POLAR_HOST_THREADS, nf / 1024));
        auto part = [=](unsigned t) {
            const size_t a = nf * t / nthr, b = nf * (t + 1) / nthr;
            for (size_t f = a; f < b; ++f) unpack_words(hb + f * NW, NW, u_hat + (f0 + f) * (size_t)N);
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nthr; ++t) pool.emplace_back(part, t);
        part(0);
        for (auto &th : pool) th.join();
    };
    auto fail_join = [&](int code) {
        if (worker.joinable()) worker.join();
        (void)hipStreamSynchronize(c->copy_stream);
        (void)hipStreamSynchronize(c->stream);
        return code;
    };
    for (size_t k = 0; k < nch; ++k) {
        const int s = (int)(k & 1);
        const size_t f0 = off[k], nf = off[k + 1] - off[k];
        // the decode of chunk k-2 must be done with in2[s] before it is overwritten
        if (k >= 2 && hipStreamWaitEvent(c->copy_stream, c->ev_free[s], 0) != hipSuccess) return fail_join(POLAR_EDEVICE);
        const double *h_src = in + f0 * (size_t)N;
        if (staged) {
            // h_in[s] was the source of chunk k-2's DMA: that copy must have left it (ev_in[s] is recorded behind it)
            if (k >= 2 && hipEventSynchronize(c->ev_in[s]) != hipSuccess) return fail_join(POLAR_EDEVICE);
            stage_chunk(k);
            h_src = c->h_in[s];
        }
        if (hipMemcpyAsync(c->in2[s].p, h_src, nf * N * sizeof(double), hipMemcpyHostToDevice,
                           c->copy_stream) != hipSuccess) return fail_join(POLAR_EDEVICE);
        if (hipEventRecord(c->ev_in[s], c->copy_stream) != hipSuccess) return fail_join(POLAR_EDEVICE);
        if (hipStreamWaitEvent(c->stream, c->ev_in[s], 0) != hipSuccess) return fail_join(POLAR_EDEVICE);
        // h_bits[s] / bits2[s] were last used by chunk k-2, whose unpacking ran during the copy above
        if (worker.joinable()) worker.join();
        rc = decode_device_impl(c, c->in2[s].p, 0, sigma, nf, (uint32_t *)c->bits2[s].p, (double *)c->pm.p + f0,
                                (uint32_t *)c->flags.p + f0, d_frozen);
        if (rc) return fail_join(rc);
        if (hipEventRecord(c->ev_free[s], c->stream) != hipSuccess) return fail_join(POLAR_EDEVICE);
        if (hipMemcpyAsync(c->h_bits[s], c->bits2[s].p, nf * NW * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream) !=
            hipSuccess) return fail_join(POLAR_EDEVICE);
        if (hipEventRecord(c->ev_out[s], c->stream) != hipSuccess) return fail_join(POLAR_EDEVICE);
        // chunk k-1 was decoded while chunk k crossed PCIe: unpack it in the background during the next copy
        if (k >= 1) {
            if (hipEventSynchronize(c->ev_out[s ^ 1]) != hipSuccess) return fail_join(POLAR_EDEVICE);
            worker = std::thread(unpack_chunk, k - 1);
        }
    }
    if (worker.joinable()) worker.join();
    HIP_TRY(c, hipEventSynchronize(c->ev_out[(nch - 1) & 1]));
    unpack_chunk(nch - 1);
    if (pm_out) HIP_TRY(c, hipMemcpyAsync(pm_out, c->pm.p, B * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (flags) HIP_TRY(c, hipMemcpyAsync(flags, c->flags.p, B * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return POLAR_OK;
}

}  // namespace

extern "C" {

const char *polar_version(void) { return "polar_hip 0.1 (gfx950)"; }

const char *polar_strerror(int code)
{
    switch (code) {
    case POLAR_OK: return "ok";
    case POLAR_EINVAL: return "invalid argument or unsupported configuration";
    case POLAR_ENOMEM: return "out of memory";
    case POLAR_EDEVICE: return "HIP runtime error";
    case POLAR_ENOKERNEL: return "no kernel instantiation for this shape";
    }
    return "unknown error";
}

const char *polar_last_error(const polar_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int polar_create(const polar_cfg *cfg, polar_ctx **out)
{
    if (!cfg || !out) return POLAR_EINVAL;
    *out = nullptr;
    const int N = cfg->N;
    if (N < 32 || N > 4096 || (N & (N - 1))) return POLAR_EINVAL;
    if (cfg->K < 1 || cfg->crc_r < 0 || cfg->crc_r > 32 || cfg->K + cfg->crc_r > N) return POLAR_EINVAL;
    if (cfg->algo < POLAR_ALGO_SC || cfg->algo > POLAR_ALGO_CASCL) return POLAR_EINVAL;
    if (cfg->dtype != POLAR_F64 && cfg->dtype != POLAR_F32) return POLAR_EINVAL;
    int L = cfg->L;
    if (cfg->algo == POLAR_ALGO_SC || cfg->algo == POLAR_ALGO_BP) L = 1;
    if (L < 1 || L > 32 || (L & (L - 1))) return POLAR_EINVAL;
    if (cfg->algo == POLAR_ALGO_CASCL && (cfg->crc_r < 1 || !cfg->crc_taps || cfg->n_taps < 2)) return POLAR_EINVAL;
    if (cfg->algo == POLAR_ALGO_BP && cfg->bp_iters < 1) return POLAR_EINVAL;

    polar_ctx *c = new (std::nothrow) polar_ctx();
    if (!c) return POLAR_ENOMEM;
    c->cfg = *cfg;
    c->cfg.L = L;
    c->n = 0;
    while ((1 << c->n) < N) ++c->n;
    c->logL = 0;
    while ((1 << c->logL) < L) ++c->logL;
    c->NW = N / 32;
    const int r = (cfg->algo == POLAR_ALGO_CASCL) ? cfg->crc_r : 0;
    c->cfg.crc_r = r;
    c->A = cfg->K + r;
    if (r > 0) {
        c->taps.assign(cfg->crc_taps, cfg->crc_taps + cfg->n_taps);
        bool has0 = false, hasr = false;
        for (int t : c->taps) {
            if (t < 0 || t > r) { delete c; return POLAR_EINVAL; }
            has0 |= (t == 0);
            hasr |= (t == r);
        }
        if (!has0 || !hasr) { delete c; return POLAR_EINVAL; }
    }
    c->cfg.crc_taps = c->taps.empty() ? nullptr : c->taps.data();
    if (cfg->info_order) {
        c->info_order.assign(cfg->info_order, cfg->info_order + c->A);
    } else {
        std::vector<int> q = default_order(N);
        c->info_order.assign(q.end() - c->A, q.end());  // I[i] = Q[N-(K+r)+i], CASCL_1024_L8.c:214-217
    }
    c->cfg.info_order = c->info_order.data();
    c->frozen.assign(N, 1);
    for (int i = 0; i < c->A; ++i) {
        const int j = c->info_order[i];
        if (j < 0 || j >= N || c->frozen[j] == 0) { delete c; return POLAR_EINVAL; }
        c->frozen[j] = 0;
    }
    c->h_crc_tab = make_crc_table(N, r, c->taps, c->info_order);

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || cfg->device < 0 || cfg->device >= ndev) {
        delete c;
        return POLAR_EDEVICE;
    }
    DeviceGuard guard(cfg->device);   // the calling thread's current device is restored on return
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, cfg->device);
    if (e != hipSuccess) {
        delete c;
        return POLAR_EDEVICE;
    }
    c->num_cu = prop.multiProcessorCount;
    auto cleanup = [&](int rc) {
        polar_destroy(c);
        return rc;
    };
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return cleanup(POLAR_EDEVICE);
    c->own_stream = true;
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) return cleanup(POLAR_EDEVICE);
    std::vector<uint32_t> fw = pack_mask(c->frozen.data(), N, false);
    std::vector<uint32_t> iw = pack_mask(c->frozen.data(), N, true);
    c->cfg.crc_systematic = (cfg->crc_systematic && r > 0) ? 1 : 0;
    if (c->cfg.crc_systematic) {
        // error metric over the K payload positions I[r..K+r) only (CASCL_1024_sys.c:820-821)
        std::fill(iw.begin(), iw.end(), 0u);
        for (int i = r; i < c->A; ++i) iw[c->info_order[i] >> 5] |= 1u << (c->info_order[i] & 31);
        // generator rows D^(r+k) mod g (the literal Gc[K][r] of CASCL_1024_sys.c:48-561), bit j = coefficient of D^j
        std::vector<uint32_t> rows((size_t)cfg->K);
        uint64_t glow = 0;
        for (int t : c->taps)
            if (t < r) glow |= 1ull << t;
        const uint64_t top = 1ull << r;
        uint64_t rem = glow;   // D^r mod g
        for (int k = 0; k < cfg->K; ++k) {
            rows[(size_t)k] = (uint32_t)rem;
            rem <<= 1;
            if (rem & top) rem = (rem ^ top) ^ glow;
        }
        if (hipMalloc(&c->d_gc_rows, rows.size() * 4) != hipSuccess) return cleanup(POLAR_ENOMEM);
        if (hipMemcpy(c->d_gc_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
            return cleanup(POLAR_EDEVICE);
    }
    if (hipMalloc(&c->d_frozen, c->NW * 4) != hipSuccess || hipMalloc(&c->d_info, c->NW * 4) != hipSuccess)
        return cleanup(POLAR_ENOMEM);
    if (hipMemcpy(c->d_frozen, fw.data(), c->NW * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->d_info, iw.data(), c->NW * 4, hipMemcpyHostToDevice) != hipSuccess)
        return cleanup(POLAR_EDEVICE);
    if (r > 0) {
        if (hipMalloc(&c->d_crc_tab, N * 4) != hipSuccess) return cleanup(POLAR_ENOMEM);
        if (hipMemcpy(c->d_crc_tab, c->h_crc_tab.data(), N * 4, hipMemcpyHostToDevice) != hipSuccess)
            return cleanup(POLAR_EDEVICE);
    }
    refresh_kernel_name(c);
    *out = c;
    return POLAR_OK;
}

void polar_destroy(polar_ctx *c)
{
    if (!c) return;
    DeviceGuard guard(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    if (c->stream_b) (void)hipStreamSynchronize(c->stream_b);
    for (Buf *b : {&c->in, &c->bits, &c->pm, &c->flags, &c->scratch, &c->gen_llr, &c->gen_u, &c->gen_cnt, &c->in2[0],
                   &c->in2[1], &c->bits2[0], &c->bits2[1], &c->scratch_b})
    {
        if (b->p) (void)hipFree(b->p);
        if (b->queue) (void)hipFree(b->queue);
    }
    for (int i = 0; i < 2; ++i) {
        if (c->h_bits[i]) (void)hipHostFree(c->h_bits[i]);
        if (c->h_in[i]) (void)hipHostFree(c->h_in[i]);
        if (c->ev_in[i]) (void)hipEventDestroy(c->ev_in[i]);
        if (c->ev_free[i]) (void)hipEventDestroy(c->ev_free[i]);
        if (c->ev_out[i]) (void)hipEventDestroy(c->ev_out[i]);
    }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->stream_b) (void)hipStreamDestroy(c->stream_b);
    if (c->ev_b) (void)hipEventDestroy(c->ev_b);
    if (c->d_frozen) (void)hipFree(c->d_frozen);
    if (c->d_info) (void)hipFree(c->d_info);
    if (c->d_crc_tab) (void)hipFree(c->d_crc_tab);
    if (c->d_frozen_override) (void)hipFree(c->d_frozen_override);
    if (c->d_info_order) (void)hipFree(c->d_info_order);
    if (c->d_gc_rows) (void)hipFree(c->d_gc_rows);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int polar_set_stream(polar_ctx *c, void *s)
{
    if (!c) return POLAR_EINVAL;
    DeviceGuard guard(c->cfg.device);
    if (c->own_stream && c->stream) {
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamDestroy(c->stream);
    }
    c->stream = (hipStream_t)s;
    c->own_stream = false;
    return POLAR_OK;
}

void *polar_get_stream(polar_ctx *c) { return c ? (void *)c->stream : nullptr; }

int polar_synchronize(polar_ctx *c)
{
    if (!c) return POLAR_EINVAL;
    DeviceGuard guard(c->cfg.device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return POLAR_OK;
}

int polar_ctx_info(const polar_ctx *c, int *N, int *K, int *A, int *L, int *algo, int *dtype)
{
    if (!c) return POLAR_EINVAL;
    if (N) *N = c->cfg.N;
    if (K) *K = c->cfg.K;
    if (A) *A = c->A;
    if (L) *L = c->cfg.L;
    if (algo) *algo = c->cfg.algo;
    if (dtype) *dtype = c->cfg.dtype;
    return POLAR_OK;
}

int polar_info_order(const polar_ctx *c, int *out, int n)
{
    if (!c || !out || n < c->A) return POLAR_EINVAL;
    for (int i = 0; i < c->A; ++i) out[i] = c->info_order[i];
    return POLAR_OK;
}

const char *polar_kernel_name(const polar_ctx *c) { return c ? c->kernel_name.c_str() : ""; }

int polar_decode_device(polar_ctx *c, const void *d_in, int in_is_f32, double sigma, size_t B, uint32_t *d_bits,
                        double *d_pm, uint32_t *d_flags)
{
    if (!c) return POLAR_EINVAL;
    DeviceGuard guard(c->cfg.device);
    return decode_device_impl(c, d_in, in_is_f32, sigma, B, d_bits, d_pm, d_flags, c->d_frozen);
}

int polar_decode_batch(polar_ctx *c, const double *llr_in, const unsigned char *frozen_mask, size_t B, int *u_hat,
                       double *pm_out, unsigned *flags)
{
    return host_batch(c, llr_in, 0.0, frozen_mask, B, u_hat, pm_out, flags);
}

int polar_decode_batch_y(polar_ctx *c, const double *y, double sigma, size_t B, int *u_hat, double *pm_out,
                         unsigned *flags)
{
    if (!(sigma > 0)) return POLAR_EINVAL;
    return host_batch(c, y, sigma, nullptr, B, u_hat, pm_out, flags);
}

int polar_decode(polar_ctx *c, const double *y, double sigma, int *u_hat)
{
    if (!(sigma > 0)) return POLAR_EINVAL;
    return host_batch(c, y, sigma, nullptr, 1, u_hat, nullptr, nullptr);
}

int polar_decode_llr(const double *llr_in, const unsigned char *frozen_mask, int N, int L, int *u_hat)
{
    if (!llr_in || !frozen_mask || !u_hat) return POLAR_EINVAL;
    if (N < 32 || N > 4096 || (N & (N - 1)) || L < 1 || L > 32 || (L & (L - 1))) return POLAR_EINVAL;   // before frozen_mask[0..N) is read
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return POLAR_EDEVICE;
    static std::mutex mu;
    static std::vector<std::pair<std::vector<unsigned char>, polar_ctx *>> cache;
    std::lock_guard<std::mutex> lock(mu);
    std::vector<unsigned char> key(frozen_mask, frozen_mask + N);   // key = (mask, L, device): a ctx is bound to one GPU
    key.push_back((unsigned char)L);
    key.push_back((unsigned char)dev);
    polar_ctx *c = nullptr;
    for (auto &kv : cache)
        if (kv.first == key) c = kv.second;
    if (!c) {
        std::vector<int> info;
        for (int j = 0; j < N; ++j)
            if (!frozen_mask[j]) info.push_back(j);
        if (info.empty()) return POLAR_EINVAL;
        polar_cfg g{};
        g.N = N; g.K = (int)info.size(); g.L = L;
        g.algo = (L == 1) ? POLAR_ALGO_SC : POLAR_ALGO_SCL;
        g.info_order = info.data();
        g.dtype = POLAR_F64;
        g.device = dev;
        int rc = polar_create(&g, &c);
        if (rc) return rc;
        if (cache.size() >= 8) {
            polar_destroy(cache.front().second);
            cache.erase(cache.begin());
        }
        cache.emplace_back(key, c);
    }
    return host_batch(c, llr_in, 0.0, nullptr, 1, u_hat, nullptr, nullptr);
}

int polar_bp_readout_device(polar_ctx *c, const void *d_in, int in_is_f32, double sigma, size_t B,
                            const uint32_t *d_u_bits, const int *checkpoints, int ncp, unsigned long long *d_E,
                            uint32_t *d_uhat_bits)
{
    if (!c || !d_in || !d_u_bits || !checkpoints || !d_E) return POLAR_EINVAL;
    if (c->cfg.algo != POLAR_ALGO_BP || ncp < 1 || ncp > 8 || B > 0x7fffffffull) return POLAR_EINVAL;
    DeviceGuard guard(c->cfg.device);
    if (B == 0) return POLAR_OK;
    polar::BpReadoutParams P{};
    for (int i = 0; i < ncp; ++i) {
        if (checkpoints[i] < 1 || checkpoints[i] > c->cfg.bp_iters || (i && checkpoints[i] <= checkpoints[i - 1]))
            return POLAR_EINVAL;
        P.cp[i] = checkpoints[i];
    }
    P.ncp = ncp;
    P.in = d_in; P.sigma = sigma; P.out_bits = d_uhat_bits; P.frozen = c->d_frozen; P.info = c->d_info;
    P.u_bits = d_u_bits; P.E = d_E;
    P.N = c->cfg.N; P.n = c->n; P.B = (int)B; P.iters = c->cfg.bp_iters;
    return polar_tu::bp_readout(c, P, c->cfg.dtype == POLAR_F32, in_is_f32 != 0);
}

int polar_bp_readout_batch(polar_ctx *c, const double *in, double sigma, size_t B, const int *u, const int *checkpoints,
                           int ncp, unsigned long long *E, int *u_hat)
{
    if (!c || !in || !u || !checkpoints || !E) return POLAR_EINVAL;
    if (ncp < 1 || ncp > 8) return POLAR_EINVAL;
    DeviceGuard guard(c->cfg.device);
    if (B == 0) return POLAR_OK;
    const int N = c->cfg.N, NW = c->NW, n = c->n;
    int rc;
    if ((rc = ensure(c, c->in, B * N * sizeof(double)))) return rc;
    if ((rc = ensure(c, c->bits, B * NW * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(c, c->gen_u, B * NW * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(c, c->gen_cnt, sizeof(unsigned long long) * 8 * (size_t)(n + 1)))) return rc;
    std::vector<uint32_t> uw(B * (size_t)NW, 0u);
    for (size_t b = 0; b < B; ++b)
        for (int j = 0; j < N; ++j)
            if (u[b * N + j]) uw[b * NW + (j >> 5)] |= 1u << (j & 31);
    HIP_TRY(c, hipMemcpyAsync(c->in.p, in, B * N * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->gen_u.p, uw.data(), uw.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->gen_cnt.p, 0, sizeof(unsigned long long) * 8 * (size_t)(n + 1), c->stream));
    rc = polar_bp_readout_device(c, c->in.p, 0, sigma, B, (const uint32_t *)c->gen_u.p, checkpoints, ncp,
                                 (unsigned long long *)c->gen_cnt.p, (uint32_t *)c->bits.p);
    if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
    std::vector<unsigned long long> he((size_t)ncp * (n + 1));
    std::vector<uint32_t> hb(u_hat ? B * (size_t)NW : 0);
    HIP_TRY(c, hipMemcpyAsync(he.data(), c->gen_cnt.p, he.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    if (u_hat) HIP_TRY(c, hipMemcpyAsync(hb.data(), c->bits.p, hb.size() * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < he.size(); ++i) E[i] += he[i];
    if (u_hat)
        for (size_t b = 0; b < B; ++b) unpack_words(hb.data() + b * NW, NW, u_hat + b * (size_t)N);
    return POLAR_OK;
}

int polar_count_errors_device(polar_ctx *c, const uint32_t *d_uhat, const uint32_t *d_u, size_t B,
                              unsigned long long *d_counters, uint32_t *d_frame_err)
{
    if (!c || !d_uhat || !d_u || !d_counters) return POLAR_EINVAL;
    DeviceGuard guard(c->cfg.device);
    if (B == 0) return POLAR_OK;
    polar::CountParams P{d_uhat, d_u, c->d_info, d_counters, d_frame_err, c->NW, (int)B};
    const int waves_per_block = 4;
    int grid = (int)std::min<size_t>((B + waves_per_block - 1) / waves_per_block, (size_t)c->num_cu * 8);
    hipLaunchKernelGGL(polar::k_count_errors, dim3(grid), dim3(64 * waves_per_block), 0, c->stream, P);
    HIP_TRY(c, hipGetLastError());
    return POLAR_OK;
}

int polar_stop_rule_cut_device(polar_ctx *c, const uint32_t *d_frame_err, size_t B, unsigned need, size_t min_frames,
                               unsigned long long *d_out)
{
    if (!c || !d_frame_err || !d_out || (need < 1 && min_frames < 1) || B > 0x7fffffffull) return POLAR_EINVAL;
    DeviceGuard guard(c->cfg.device);
    if (B == 0) {
        HIP_TRY(c, hipMemsetAsync(d_out, 0, 3 * sizeof(unsigned long long), c->stream));
        return POLAR_OK;
    }
    hipLaunchKernelGGL(polar::k_stop_cut, dim3(1), dim3(1024), 0, c->stream, d_frame_err, (int)B, need,
                       (int)std::min<size_t>(min_frames, B), d_out);
    HIP_TRY(c, hipGetLastError());
    return POLAR_OK;
}

int polar_stop_rule_batch_y(polar_ctx *c, const double *y, double sigma, const uint32_t *u_bits, size_t B, unsigned need,
                            size_t min_frames, size_t *consumed, unsigned long long *block_errors,
                            unsigned long long *bit_errors)
{
    if (!c || !y || !u_bits || !consumed || !block_errors || !bit_errors || (need < 1 && min_frames < 1) || !(sigma > 0))
        return POLAR_EINVAL;
    *consumed = 0; *block_errors = 0; *bit_errors = 0;
    if (B == 0) return POLAR_OK;
    if (B > 0x7fffffffull) return POLAR_EINVAL;
    DeviceGuard guard(c->cfg.device);
    const int N = c->cfg.N, NW = c->NW;
    int rc;
    if ((rc = ensure(c, c->in, B * N * sizeof(double)))) return rc;
    if ((rc = ensure(c, c->bits, B * NW * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(c, c->gen_u, B * NW * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(c, c->flags, B * sizeof(uint32_t)))) return rc;           // per-frame error counts
    if ((rc = ensure(c, c->gen_cnt, 5 * sizeof(unsigned long long)))) return rc;   // [0..2) totals, [2..5) the cut
    unsigned long long *cnt = (unsigned long long *)c->gen_cnt.p;
    HIP_TRY(c, hipMemcpyAsync(c->in.p, y, B * N * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->gen_u.p, u_bits, B * NW * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(cnt, 0, 5 * sizeof(unsigned long long), c->stream));
    if ((rc = decode_device_impl(c, c->in.p, 0, sigma, B, (uint32_t *)c->bits.p, nullptr, nullptr, c->d_frozen))) return rc;
    if ((rc = polar_count_errors_device(c, (uint32_t *)c->bits.p, (uint32_t *)c->gen_u.p, B, cnt, (uint32_t *)c->flags.p)))
        return rc;
    if ((rc = polar_stop_rule_cut_device(c, (uint32_t *)c->flags.p, B, need, min_frames, cnt + 2))) return rc;
    unsigned long long h[3];
    HIP_TRY(c, hipMemcpyAsync(h, cnt + 2, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *consumed = (size_t)h[0];
    *block_errors = h[1];
    *bit_errors = h[2];
    return POLAR_OK;
}

int polar_generate_device(polar_ctx *c, unsigned long long seed, unsigned long long first_frame, double snr_db,
                          size_t B, void *d_out, int out_is_f32, int out_is_y, uint32_t *d_u_bits)
{
    if (!c || !d_out || B > 0x7fffffffull) return POLAR_EINVAL;
    DeviceGuard guard(c->cfg.device);
    if (B == 0) return POLAR_OK;
    const polar_cfg &g = c->cfg;
    if (g.N < 64) return POLAR_EINVAL;
    if (!c->d_info_order) {
        HIP_TRY(c, hipMalloc(&c->d_info_order, sizeof(int) * (size_t)c->A));
        HIP_TRY(c, hipMemcpy(c->d_info_order, c->info_order.data(), sizeof(int) * (size_t)c->A, hipMemcpyHostToDevice));
    }
    polar::GenParams P{};
    P.out = d_out; P.u_bits = d_u_bits; P.info_order = c->d_info_order;
    P.seed = seed; P.first_frame = first_frame;
    P.sigma = std::pow(10.0, snr_db / -20.0);  // SCL_1024.c:226
    P.crc_r = g.crc_r; P.crc_mask = 0; P.crc_top = 0;
    P.gc_rows = c->d_gc_rows;
    if (g.crc_r == 0) P.crc_mask = 1u;
    for (int t : c->taps) {
        if (t < 32) P.crc_mask |= 1u << t;
        else P.crc_top = 1u;
    }
    P.N = g.N; P.n = c->n; P.K = g.K; P.A = c->A; P.B = (int)B;
    P.out_is_f32 = out_is_f32; P.out_is_y = out_is_y;
    const int waves = 4;
    const size_t lds = (size_t)waves * (g.N + 2 * 1024);
    int grid = (int)std::min<size_t>((B + waves - 1) / waves, (size_t)c->num_cu * 8);
    hipLaunchKernelGGL(polar::k_generate, dim3(grid), dim3(64 * waves), lds, c->stream, P);
    HIP_TRY(c, hipGetLastError());
    return POLAR_OK;
}

// generate -> decode -> count for B frames; the two counters stay in c->gen_cnt (device) and are copied to h[2] if h != null
static int fer_batch_impl(polar_ctx *c, unsigned long long seed, unsigned long long first_frame, double snr_db, size_t B,
                          unsigned long long *h, uint32_t *d_frame_err = nullptr)
{
    const int N = c->cfg.N, NW = c->NW;
    const bool f32 = c->cfg.dtype == POLAR_F32;
    int rc;
    if ((rc = ensure(c, c->gen_llr, B * N * (f32 ? 4 : 8)))) return rc;
    if ((rc = ensure(c, c->gen_u, B * NW * 4))) return rc;
    if ((rc = ensure(c, c->bits, B * NW * 4))) return rc;
    if ((rc = ensure(c, c->gen_cnt, 16))) return rc;
    HIP_TRY(c, hipMemsetAsync(c->gen_cnt.p, 0, 16, c->stream));
    // Two halves on two streams (own decode scratch each): the generator of one half and the partly filled last
    // pass of its decode overlap the other half's decode.  Frame i of the batch is the same frame either way
    // (the generator is counter-based), and the two counters are atomics.
    const size_t half = (B >= 32768) ? (B / 2 + 63) / 64 * 64 : B;
    if (half < B && !c->stream_b) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->stream_b, hipStreamNonBlocking));
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_b, hipEventDisableTiming));
    }
    const size_t esz = f32 ? 4 : 8;
    auto run_part = [&](size_t f0, size_t nf) -> int {
        int r;
        if ((r = polar_generate_device(c, seed, first_frame + f0, snr_db, nf, (char *)c->gen_llr.p + f0 * N * esz, f32 ? 1 : 0, 0,
                                       (uint32_t *)c->gen_u.p + f0 * NW))) return r;
        if ((r = decode_device_impl(c, (char *)c->gen_llr.p + f0 * N * esz, f32 ? 1 : 0, 0.0, nf, (uint32_t *)c->bits.p + f0 * NW,
                                    nullptr, nullptr, c->d_frozen))) return r;
        return polar_count_errors_device(c, (uint32_t *)c->bits.p + f0 * NW, (uint32_t *)c->gen_u.p + f0 * NW, nf,
                                         (unsigned long long *)c->gen_cnt.p, d_frame_err ? d_frame_err + f0 : nullptr);
    };
    if (half < B) {
        // second half on stream_b, after the counters were cleared on the main stream
        HIP_TRY(c, hipEventRecord(c->ev_b, c->stream));
        HIP_TRY(c, hipStreamWaitEvent(c->stream_b, c->ev_b, 0));
        std::swap(c->stream, c->stream_b);
        std::swap(c->scratch, c->scratch_b);
        rc = run_part(half, B - half);
        hipError_t e = hipEventRecord(c->ev_b, c->stream);
        std::swap(c->stream, c->stream_b);
        std::swap(c->scratch, c->scratch_b);
        if (rc) return rc;
        HIP_TRY(c, e);
    }
    if ((rc = run_part(0, half))) return rc;
    if (half < B) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_b, 0));
    if (h) {
        HIP_TRY(c, hipMemcpyAsync(h, c->gen_cnt.p, 16, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return POLAR_OK;
}

int polar_fer_batch(polar_ctx *c, unsigned long long seed, unsigned long long first_frame, double snr_db, size_t B,
                    unsigned long long *block_errors, unsigned long long *bit_errors)
{
    if (!c || !block_errors || !bit_errors) return POLAR_EINVAL;
    DeviceGuard guard(c->cfg.device);
    if (B == 0) return POLAR_OK;
    unsigned long long h[2] = {0, 0};
    const int rc = fer_batch_impl(c, seed, first_frame, snr_db, B, h);
    if (rc) return rc;
    *block_errors += h[0];
    *bit_errors += h[1];
    return POLAR_OK;
}

// ---- frames sharded over the GPUs of one node, RCCL only for the final reduction -------------------------------
// RCCL is loaded on first use with dlopen (RTLD_LOCAL): the library has no link-time dependency on it, and a host process
// that carries its own copy (torch does) is not disturbed.
namespace {
struct RcclApi {
    void *handle = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};
// ncclDataType_t / ncclRedOp_t as rccl.h numbers them (stable across NCCL 2.x: ncclUint32 = 3, ncclUint64 = 5,
// ncclSum = 0).  They are NOT trusted: polar_group_create runs known values through the loaded library with exactly
// these numbers (group_self_test) and refuses the group if the answer is not the 64-bit integer sum / the 32-bit
// gather in rank order -- a different enum layout or ABI gives POLAR_EDEVICE, never wrong counters.
constexpr int kNcclUint32 = 3, kNcclUint64 = 5, kNcclSum = 0;

RcclApi &rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
        }
        if (!api.handle) return;
        api.CommInitAll = (int (*)(void **, int, const int *))dlsym(api.handle, "ncclCommInitAll");
        api.CommDestroy = (int (*)(void *))dlsym(api.handle, "ncclCommDestroy");
        api.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(api.handle, "ncclAllReduce");
        api.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(api.handle, "ncclAllGather");
        api.GroupStart = (int (*)())dlsym(api.handle, "ncclGroupStart");
        api.GroupEnd = (int (*)())dlsym(api.handle, "ncclGroupEnd");
        api.GetErrorString = (const char *(*)(int))dlsym(api.handle, "ncclGetErrorString");
        api.ok = api.CommInitAll && api.CommDestroy && api.AllReduce && api.AllGather && api.GroupStart && api.GroupEnd;
    });
    return api;
}
}  // namespace

struct polar_group {
    std::vector<polar_ctx *> ctx;
    std::vector<void *> comms;   // ncclComm_t per GPU
    std::vector<Buf> ferr;       // per GPU: its shard's per-frame error counts (exact stop rule)
    Buf gathered;                // GPU 0: the counts of all shards in frame order
    Buf cut_out;                 // GPU 0: k_stop_cut's three numbers
};

extern "C++" {
namespace {

// grouped collective over the ranks of the group, one call per rank between GroupStart / GroupEnd
template <typename F>
bool group_collective(polar_group *g, F &&per_rank)
{
    RcclApi &R = rccl();
    bool bad = R.GroupStart() != 0;
    for (int i = 0; i < (int)g->ctx.size() && !bad; ++i) {
        DeviceGuard guard(i);
        bad = per_rank(i) != 0;
    }
    return !((R.GroupEnd() != 0) || bad);
}

bool group_sync(polar_group *g)
{
    bool ok = true;
    for (int i = 0; i < (int)g->ctx.size(); ++i) {
        DeviceGuard guard(i);
        ok = (hipStreamSynchronize(g->ctx[(size_t)i]->stream) == hipSuccess) && ok;
    }
    return ok;
}

// Known answers through the loaded RCCL with the enum numbers this file uses: rank i contributes
// {2^40 + i + 1, 3} as uint64 -- the sum must be {n 2^40 + n(n+1)/2, 3n} (a 32-bit or floating type, or a different
// reduction, gives something else) -- and {0xC0DE0000 + i} as uint32, which must come back in rank order on every rank.
int group_self_test(polar_group *g)
{
    RcclApi &R = rccl();
    const int n = (int)g->ctx.size();
    for (int i = 0; i < n; ++i) {
        polar_ctx *c = g->ctx[(size_t)i];
        DeviceGuard guard(i);
        int rc = ensure(c, c->gen_cnt, 16);
        if (rc) return rc;
        if ((rc = ensure(c, g->ferr[(size_t)i], (size_t)(n + 1) * 4))) return rc;
        const unsigned long long v[2] = {(1ull << 40) + (unsigned long long)i + 1ull, 3ull};
        const uint32_t w = 0xC0DE0000u + (uint32_t)i;
        if (hipMemcpyAsync(c->gen_cnt.p, v, 16, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipMemcpyAsync(g->ferr[(size_t)i].p, &w, 4, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess)
            return POLAR_EDEVICE;
    }
    if (!group_collective(g, [&](int i) {
            void *buf = g->ctx[(size_t)i]->gen_cnt.p;
            return R.AllReduce(buf, buf, 2, kNcclUint64, kNcclSum, g->comms[(size_t)i], g->ctx[(size_t)i]->stream);
        }))
        return POLAR_EDEVICE;
    if (!group_collective(g, [&](int i) {
            uint32_t *b = (uint32_t *)g->ferr[(size_t)i].p;
            return R.AllGather(b, b + 1, 1, kNcclUint32, g->comms[(size_t)i], g->ctx[(size_t)i]->stream);
        }))
        return POLAR_EDEVICE;
    const unsigned long long want0 = (unsigned long long)n * (1ull << 40) + (unsigned long long)n * (n + 1) / 2;
    for (int i = 0; i < n; ++i) {
        polar_ctx *c = g->ctx[(size_t)i];
        DeviceGuard guard(i);
        unsigned long long h[2] = {0, 0};
        std::vector<uint32_t> got((size_t)n);
        if (hipMemcpyAsync(h, c->gen_cnt.p, 16, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipMemcpyAsync(got.data(), (uint32_t *)g->ferr[(size_t)i].p + 1, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess)
            return POLAR_EDEVICE;
        bool ok = h[0] == want0 && h[1] == 3ull * (unsigned long long)n;
        for (int q = 0; q < n; ++q) ok = ok && got[(size_t)q] == 0xC0DE0000u + (uint32_t)q;
        if (!ok) {
            c->last_error = "RCCL self-test: all-reduce(uint64, sum) / all-gather(uint32) did not return the known answer";
            return POLAR_EDEVICE;
        }
    }
    return POLAR_OK;
}

}  // namespace
}  // extern "C++"

void polar_group_destroy(polar_group *g)
{
    if (!g) return;
    RcclApi &R = rccl();
    for (void *cm : g->comms)
        if (cm && R.ok) (void)R.CommDestroy(cm);
    for (size_t i = 0; i < g->ferr.size(); ++i)
        if (g->ferr[i].p) {
            DeviceGuard guard((int)i);
            (void)hipFree(g->ferr[i].p);
        }
    {
        DeviceGuard guard(0);
        if (g->gathered.p) (void)hipFree(g->gathered.p);
        if (g->cut_out.p) (void)hipFree(g->cut_out.p);
    }
    for (polar_ctx *c : g->ctx) polar_destroy(c);
    delete g;
}

int polar_group_create(const polar_cfg *cfg, int ngpus, polar_group **out)
{
    if (!cfg || !out || ngpus < 1 || ngpus > 64) return POLAR_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ngpus > ndev) return POLAR_EDEVICE;
    RcclApi &R = rccl();
    if (!R.ok) return POLAR_EDEVICE;   // no RCCL on this machine
    polar_group *g = new (std::nothrow) polar_group();
    if (!g) return POLAR_ENOMEM;
    g->ctx.assign((size_t)ngpus, nullptr);
    g->comms.assign((size_t)ngpus, nullptr);
    g->ferr.assign((size_t)ngpus, Buf{});
    std::vector<int> devs((size_t)ngpus);
    for (int i = 0; i < ngpus; ++i) {
        devs[(size_t)i] = i;
        polar_cfg one = *cfg;
        one.device = i;
        const int rc = polar_create(&one, &g->ctx[(size_t)i]);
        if (rc) {
            polar_group_destroy(g);
            return rc;
        }
    }
    if (R.CommInitAll(g->comms.data(), ngpus, devs.data()) != 0) {
        polar_group_destroy(g);
        return POLAR_EDEVICE;
    }
    const int st = group_self_test(g);   // wrong enum numbers / ABI: refuse the group instead of returning wrong counters
    if (st) {
        polar_group_destroy(g);
        return st;
    }
    *out = g;
    return POLAR_OK;
}

int polar_group_size(const polar_group *g) { return g ? (int)g->ctx.size() : 0; }

int polar_group_fer_batch(polar_group *g, unsigned long long seed, unsigned long long first_frame, double snr_db,
                          size_t frames_per_gpu, unsigned long long *block_errors, unsigned long long *bit_errors,
                          double *seconds)
{
    if (!g || !block_errors || !bit_errors) return POLAR_EINVAL;
    if (frames_per_gpu == 0) return POLAR_OK;
    RcclApi &R = rccl();
    const int ngpus = (int)g->ctx.size();
    std::vector<int> rcs((size_t)ngpus, POLAR_OK);
    std::vector<double> secs((size_t)ngpus, 0.0);
    // one host thread per GPU: its shard of the frame range, no data-path collective
    {
        std::vector<std::thread> th;
        for (int i = 0; i < ngpus; ++i)
            th.emplace_back([&, i] {
                polar_ctx *c = g->ctx[(size_t)i];
                DeviceGuard guard(i);
                const auto t0 = std::chrono::steady_clock::now();
                rcs[(size_t)i] = fer_batch_impl(c, seed, first_frame + (unsigned long long)i * frames_per_gpu, snr_db,
                                                frames_per_gpu, nullptr);
                if (rcs[(size_t)i] == POLAR_OK && hipStreamSynchronize(c->stream) != hipSuccess) rcs[(size_t)i] = POLAR_EDEVICE;
                secs[(size_t)i] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            });
        for (auto &t : th) t.join();
    }
    int rc = POLAR_OK;
    for (int i = 0; i < ngpus; ++i)
        if (rcs[(size_t)i]) rc = rcs[(size_t)i];
    // the only exchange: sum of the two counters over the GPUs (16 bytes per rank over xGMI)
    if (rc == POLAR_OK &&
        !group_collective(g, [&](int i) {
            void *buf = g->ctx[(size_t)i]->gen_cnt.p;
            return R.AllReduce(buf, buf, 2, kNcclUint64, kNcclSum, g->comms[(size_t)i], g->ctx[(size_t)i]->stream);
        }))
        rc = POLAR_EDEVICE;
    if (rc == POLAR_OK) {
        unsigned long long h[2] = {0, 0};
        {
            DeviceGuard guard(0);
            if (hipMemcpyAsync(h, g->ctx[0]->gen_cnt.p, 16, hipMemcpyDeviceToHost, g->ctx[0]->stream) != hipSuccess ||
                hipStreamSynchronize(g->ctx[0]->stream) != hipSuccess)
                rc = POLAR_EDEVICE;
        }
        for (int i = 1; i < ngpus; ++i) {   // every rank holds the sum: drain the other streams before the next call reuses the buffers
            DeviceGuard gg(i);
            if (hipStreamSynchronize(g->ctx[(size_t)i]->stream) != hipSuccess) rc = POLAR_EDEVICE;
        }
        if (rc == POLAR_OK) {
            *block_errors += h[0];
            *bit_errors += h[1];
        }
    }
    if (seconds) *seconds = *std::max_element(secs.begin(), secs.end());
    return rc;
}

// The reference's sequential stop rule (`for (run = 0; errBlock < BLE; run++)`, SCL_1024.c:228) over a batch that was
// decoded in shards: every GPU leaves the per-frame error counts of its shard on the device (k_count_errors), ONE
// ncclAllGather of frames_per_gpu x uint32 per rank puts them in frame order on every GPU, and k_stop_cut on GPU 0 finds the
// frame that brings the block errors to `need` exactly as polar_stop_rule_cut_device does for one GPU.  Frame f of the
// range is the same frame for every ngpus, so the three numbers do not depend on how many GPUs shared the batch.
int polar_group_stop_rule_batch(polar_group *g, unsigned long long seed, unsigned long long first_frame, double snr_db,
                                size_t frames_per_gpu, unsigned need, size_t min_frames, size_t *frames_used,
                                unsigned long long *block_errors, unsigned long long *bit_errors)
{
    if (!g || !frames_used || !block_errors || !bit_errors) return POLAR_EINVAL;
    const int ngpus = (int)g->ctx.size();
    const size_t total = frames_per_gpu * (size_t)ngpus;
    if (frames_per_gpu == 0 || total > 0x7fffffffull || min_frames > total) return POLAR_EINVAL;
    RcclApi &R = rccl();
    int rc = POLAR_OK;
    for (int i = 0; i < ngpus && !rc; ++i) {
        DeviceGuard guard(i);
        rc = ensure(g->ctx[(size_t)i], g->ferr[(size_t)i], std::max(frames_per_gpu, (size_t)ngpus + 1) * 4);
    }
    {
        DeviceGuard guard(0);
        if (!rc) rc = ensure(g->ctx[0], g->gathered, total * 4);
        if (!rc) rc = ensure(g->ctx[0], g->cut_out, 3 * sizeof(unsigned long long));
    }
    if (rc) return rc;
    std::vector<int> rcs((size_t)ngpus, POLAR_OK);
    {
        std::vector<std::thread> th;
        for (int i = 0; i < ngpus; ++i)
            th.emplace_back([&, i] {
                polar_ctx *c = g->ctx[(size_t)i];
                DeviceGuard guard(i);
                rcs[(size_t)i] = fer_batch_impl(c, seed, first_frame + (unsigned long long)i * frames_per_gpu, snr_db,
                                                frames_per_gpu, nullptr, (uint32_t *)g->ferr[(size_t)i].p);
                if (rcs[(size_t)i] == POLAR_OK && hipStreamSynchronize(c->stream) != hipSuccess) rcs[(size_t)i] = POLAR_EDEVICE;
            });
        for (auto &t : th) t.join();
    }
    for (int i = 0; i < ngpus; ++i)
        if (rcs[(size_t)i]) return rcs[(size_t)i];
    // rank i's counts land at [i * frames_per_gpu, (i + 1) * frames_per_gpu) of every rank's receive buffer; only
    // GPU 0's copy is used (ranks > 0 receive into a buffer of the same size, as the collective requires)
    std::vector<Buf> recv((size_t)ngpus);
    recv[0] = g->gathered;
    for (int i = 1; i < ngpus && !rc; ++i) {
        DeviceGuard guard(i);
        if (hipMalloc(&recv[(size_t)i].p, total * 4) != hipSuccess) rc = POLAR_ENOMEM;
    }
    if (!rc && !group_collective(g, [&](int i) {
            return R.AllGather(g->ferr[(size_t)i].p, recv[(size_t)i].p, frames_per_gpu, kNcclUint32, g->comms[(size_t)i],
                               g->ctx[(size_t)i]->stream);
        }))
        rc = POLAR_EDEVICE;
    if (!rc && !group_sync(g)) rc = POLAR_EDEVICE;
    for (int i = 1; i < ngpus; ++i)
        if (recv[(size_t)i].p) {
            DeviceGuard guard(i);
            (void)hipFree(recv[(size_t)i].p);
        }
    if (rc) return rc;
    unsigned long long h[3] = {0, 0, 0};
    {
        DeviceGuard guard(0);
        polar_ctx *c0 = g->ctx[0];
        rc = polar_stop_rule_cut_device(c0, (const uint32_t *)g->gathered.p, total, need, min_frames,
                                        (unsigned long long *)g->cut_out.p);
        if (rc) return rc;
        if (hipMemcpyAsync(h, g->cut_out.p, sizeof h, hipMemcpyDeviceToHost, c0->stream) != hipSuccess ||
            hipStreamSynchronize(c0->stream) != hipSuccess)
            return POLAR_EDEVICE;
    }
    *frames_used = (size_t)h[0];
    *block_errors = h[1];
    *bit_errors = h[2];
    return POLAR_OK;
}

int polar_fer_multi_gpu(const polar_cfg *cfg, int ngpus, unsigned long long seed, unsigned long long first_frame, double snr_db,
                        size_t frames_per_gpu, unsigned long long *block_errors, unsigned long long *bit_errors,
                        double *seconds)
{
    if (!block_errors || !bit_errors) return POLAR_EINVAL;
    polar_group *g = nullptr;
    int rc = polar_group_create(cfg, ngpus, &g);
    if (rc) return rc;
    rc = polar_group_fer_batch(g, seed, first_frame, snr_db, frames_per_gpu, block_errors, bit_errors, seconds);
    polar_group_destroy(g);
    return rc;
}

#ifdef POLAR_TESTING
// ---- include/polar_hip_testing.h ----------------------------------------------------------------------------
int polar_testing_select_kernel(polar_ctx *c, int variant)
{
    if (!c || variant < POLAR_TEST_KERNEL_AUTO || variant > POLAR_TEST_KERNEL_FOUR_PER_WAVE) return POLAR_EINVAL;
    c->force_generic = (variant == POLAR_TEST_KERNEL_GENERIC || variant == POLAR_TEST_KERNEL_GENERIC_SPILL);
    c->force_spill = (variant == POLAR_TEST_KERNEL_GENERIC_SPILL || variant == POLAR_TEST_KERNEL_BIG);
    c->use_fast2 = (variant != POLAR_TEST_KERNEL_ONE_PER_WAVE);
    c->use_fast4 = (variant == POLAR_TEST_KERNEL_FOUR_PER_WAVE);
    refresh_kernel_name(c);
    return POLAR_OK;
}

int polar_testing_big_split(polar_ctx *c, int split)
{
    if (!c || (split != 0 && split != 35 && split != 46 && split != 57 && split != 351 && split != 371))
        return POLAR_EINVAL;
    c->big_split = split;
    return POLAR_OK;
}
#endif  // POLAR_TESTING

int polar_time_decode_device(polar_ctx *c, const void *d_in, int in_is_f32, double sigma, size_t B,
                             uint32_t *d_bits, int reps, float *ms)
{
    if (!c || !ms || reps < 1) return POLAR_EINVAL;
    DeviceGuard guard(c->cfg.device);
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    for (int i = 0; i < reps; ++i) {
        int rc = decode_device_impl(c, d_in, in_is_f32, sigma, B, d_bits, nullptr, nullptr, c->d_frozen);
        if (rc) return rc;
    }
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    HIP_TRY(c, hipEventSynchronize(c->ev1));
    float t = 0;
    HIP_TRY(c, hipEventElapsedTime(&t, c->ev0, c->ev1));
    *ms = t / (float)reps;
    return POLAR_OK;
}

}  // extern "C"
