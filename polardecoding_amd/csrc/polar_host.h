// polar_host.h -- internal: the context object behind the C ABI and the launcher each kernel translation unit exports.
// The library is built from several translation units (one per kernel family, compiled in parallel; see
// __graft_entry__.py): polar_hip.hip holds the C ABI and the host logic, k_*.hip the kernels with their launch code.
#pragma once
#include "../../include/polar_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "polar_params.h"

struct PolarBuf {
    void *p = nullptr;
    size_t cap = 0;
    unsigned *queue = nullptr;   // job counter of the persistent kernels that use this scratch buffer (work_queue() below)
};
typedef PolarBuf Buf;

struct polar_ctx {
    polar_cfg cfg{};
    int n = 0, A = 0, NW = 0, logL = 0;
    std::vector<int> info_order;          // I[]
    std::vector<unsigned char> frozen;    // [N]
    std::vector<int> taps;
    std::vector<uint32_t> h_crc_tab;      // [N]
    uint32_t *d_frozen = nullptr;         // [NW] bit = frozen
    uint32_t *d_info = nullptr;           // [NW] bit = unfrozen
    uint32_t *d_crc_tab = nullptr;        // [N] or null
    uint32_t *d_gc_rows = nullptr;        // [K] systematic CRC generator rows (D^(r+k) mod g), or null
    uint32_t *d_frozen_override = nullptr;
    int *d_info_order = nullptr;          // [A] for the device-side generator
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cu = 0;
    Buf in, bits, pm, flags;              // staging for the host-pointer entry points
    Buf in2[2], bits2[2];                 // chunked host pipeline: ping-pong device buffers
    uint32_t *h_bits[2] = {nullptr, nullptr};   // pinned host copies of the packed decisions
    size_t h_bits_cap = 0;
    double *h_in[2] = {nullptr, nullptr};       // pinned staging of the caller's (pageable) input chunks, big batches only
    size_t h_in_cap = 0;
    hipStream_t copy_stream = nullptr;    // host -> device copies overlap the decode of the previous chunk
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    Buf scratch;                          // k_scl_fast per-wave scratch
    Buf gen_llr, gen_u, gen_cnt;          // polar_fer_batch
    Buf scratch_b;                        // second decode scratch: polar_fer_batch runs its two halves on two streams
    hipStream_t stream_b = nullptr;
    hipEvent_t ev_b = nullptr;
    std::string last_error;
    std::string kernel_name;
    // kernel selection overrides, set only through include/polar_hip_testing.h (cross-checks of the tuned kernels)
    bool force_generic = false;
    bool use_fast2 = true;      // false: one codeword per wavefront (k_scl_fast) instead of two at N = 1024
    bool use_fast4 = false;     // four codewords per wavefront (k_scl_fast4) at N = 1024
    bool force_spill = false;   // no tuned L = 8 kernel; with force_generic: the global-scratch variant of k_scl_generic
    int big_split = 0;          // 35 | 46 | 57: LDS / scratch split of k_scl_big; 0 = the measured best
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

// Every entry point that allocates or launches runs on the ctx's device whatever the calling thread had current,
// and leaves the thread's current device as it found it.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard()
    {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

inline int fail(polar_ctx *c, hipError_t e, const char *what)
{
    if (c) c->last_error = std::string(what) + ": " + hipGetErrorString(e);
    return POLAR_EDEVICE;
}

#define HIP_TRY(c, expr)                                   \
    do {                                                   \
        hipError_t e_ = (expr);                            \
        if (e_ != hipSuccess) return fail(c, e_, #expr);   \
    } while (0)

inline int ensure(polar_ctx *c, Buf &b, size_t bytes)
{
    if (b.cap >= bytes) return POLAR_OK;
    if (b.p) HIP_TRY(c, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    HIP_TRY(c, hipMalloc(&b.p, bytes));
    b.cap = bytes;
    return POLAR_OK;
}

// Work queue of a persistent kernel.  The resident wavefronts take their first job by their index and every further one
// from a counter (atomic add), so a wavefront that gets fewer issue slots simply takes fewer jobs: a launch ends when the
// work does, not when the slowest statically assigned wavefront does.  One counter per scratch buffer
// (the buffer and its counter belong to one stream at a time); the kernel leaves it at zero (polar_params.h job_fetch).
inline int work_queue(polar_ctx *c, Buf &scratch, unsigned **counter)
{
    if (!scratch.queue) {
        HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&scratch.queue), 256));
        HIP_TRY(c, hipMemsetAsync(scratch.queue, 0, 256, c->stream));
    }
    *counter = scratch.queue;
    return POLAR_OK;
}

// Launchers exported by the kernel translation units.  r32 / in32: arithmetic type / input type is float (else double).
namespace polar_tu {
int scl_generic(polar_ctx *c, const polar::SclParams &P, bool r32, bool in32);             // k_generic.hip (uses c->logL, c->force_spill)
int scl_big_f64(polar_ctx *c, const polar::SclParams &P, bool in32);                       // k_big_f64.hip
int scl_big_f32(polar_ctx *c, const polar::SclParams &P, bool in32);                       // k_big_f32.hip
int sc_lanes(polar_ctx *c, const polar::SclParams &P, bool r32, bool in32);                // k_sc.hip
int scl_fast(polar_ctx *c, const polar::SclParams &P, bool r32, bool in32, bool crc);      // k_fast.hip: N = 128, N = 1024 one codeword per wave
int scl_fast2(polar_ctx *c, const polar::SclParams &P, bool r32, bool in32, bool crc);     // k_fast2.hip: N = 1024, two per wave (headline)
int bp(polar_ctx *c, const polar::BpParams &P, bool r32, bool in32);                       // k_bp.hip
int bp_readout(polar_ctx *c, const polar::BpReadoutParams &P, bool r32, bool in32);        // k_bp.hip
#ifdef POLAR_TESTING
int scl_fast4(polar_ctx *c, const polar::SclParams &P, bool r32, bool in32, bool crc);     // k_fast4.hip (libpolar_hip_testing.so only)
#endif
}  // namespace polar_tu
