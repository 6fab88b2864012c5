// calib_traffic.hip -- known-byte streaming kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
// in the access shapes the decoders use (MI355X_MICROARCH.md "HBM": FETCH_SIZE reports half
// the bytes of a wide coalesced read on gfx950, "other access widths are uncalibrated: calibrate on a known byte count
// in your own access pattern").
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/calib_traffic tools/calib_traffic.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT/fetch -- tools/calib_traffic
//   rocprofv3 --pmc WRITE_SIZE --output-format csv -d OUT/write -- tools/calib_traffic
//   python tools/calibrate_counters.py OUT > profiles/r03_counter_calibration.txt
//
// Every kernel streams once over a 1 GiB buffer (four times the 256 MiB Infinity Cache, 32 times the L2), so the
// bytes it must move at the L2 <-> fabric interface are known: the program prints them per kernel ("TRUE ..." lines).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static constexpr size_t BYTES = 1ull << 30;

// ---- reads -----------------------------------------------------------------------------------------------------
// 8 B per lane, 512 contiguous bytes per wave-instruction, plain global_load_dwordx2 (the decoders' input rows)
__global__ __launch_bounds__(256) void cal_read_b64_plain(const double *p, size_t n, double *sink)
{
    double acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 1.2345e300) *sink = acc;
}
// the same with the sc1 bit (relaxed agent-scope atomic load = the decoders' ld_bypass / ld_sc of scratch rows)
__global__ __launch_bounds__(256) void cal_read_b64_sc1(const double *p, size_t n, double *sink)
{
    double acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc += __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (acc == 1.2345e300) *sink = acc;
}
// 16 B per lane, global_load_dwordx4 (k_sc_lanes' channel rows; the guide's calibrated shape)
__global__ __launch_bounds__(256) void cal_read_b128(const double2 *p, size_t n, double *sink)
{
    double acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double2 v = p[i];
        acc += v.x + v.y;
    }
    if (acc == 1.2345e300) *sink = acc;
}
// 4 B per lane (f32 kernels)
__global__ __launch_bounds__(256) void cal_read_b32(const float *p, size_t n, double *sink)
{
    float acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 1.2345e30f) *sink = acc;
}
// the pair kernel's scratch rows: a wave-instruction reads sixteen 32-byte pieces (4 lanes x 8 B) that lie 2 KB apart,
// sc1; consecutive instructions of the wave read the next 32 bytes of each row, so every 128-byte line is consumed by
// four consecutive instructions of one wave.  A wave owns a 32 KB tile (16 rows x 2 KB).
__global__ __launch_bounds__(256) void cal_read_b64_sc1_pieces32(const double *p, size_t n, double *sink)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    double acc = 0;
    for (size_t t = wave; t < n / 4096; t += nwaves) {
        const double *tile = p + t * 4096 + (size_t)(lane >> 2) * 256 + (lane & 3);
#pragma unroll 8
        for (int r = 0; r < 64; ++r) acc += __hip_atomic_load(tile + 4 * r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (acc == 1.2345e300) *sink = acc;
}

// ---- writes ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cal_write_b64(double *p, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)i;
}
__global__ __launch_bounds__(256) void cal_write_b128(double2 *p, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_double2((double)i, 1.0);
}
__global__ __launch_bounds__(256) void cal_write_b32(float *p, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (float)i;
}
// the pair kernel's scratch stores: sixteen 32-byte pieces 2 KB apart per wave-instruction (see the read above)
__global__ __launch_bounds__(256) void cal_write_b64_pieces32(double *p, size_t n)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t t = wave; t < n / 4096; t += nwaves) {
        double *tile = p + t * 4096 + (size_t)(lane >> 2) * 256 + (lane & 3);
#pragma unroll 8
        for (int r = 0; r < 64; ++r) tile[4 * r] = (double)r;
    }
}

// ---- register spills (private segment) -------------------------------------------------------------------------
// Each lane keeps WORDS doubles in a dynamically indexed private array: the compiler places it in scratch memory,
// which is what spilled VGPRs use (scratch_store / scratch_load, or buffer_* with the scratch descriptor).  One round
// = every element written once, then read once, with every resident wave's whole array (64 x WORDS x 8 B) in between:
// with >= 2048 resident waves that is >= 256 MiB between the write and the read of a line, so both cross the fabric.
template <int WORDS>
__global__ __launch_bounds__(256) void cal_spill_roundtrip(int rounds, int rot, double *sink)
{
    double a[WORDS];
    double acc = 0;
    for (int it = 0; it < rounds; ++it) {
        for (int i = 0; i < WORDS; ++i) a[(i + rot) & (WORDS - 1)] = (double)(i + it) + acc;
        __builtin_amdgcn_s_waitcnt(0);
        for (int i = 0; i < WORDS; ++i) acc += a[(i * 7 + rot) & (WORDS - 1)];
    }
    if (acc == 1.2345e300) *sink = acc;
}

int main()
{
    void *buf = nullptr;
    double *sink = nullptr;
    CK(hipMalloc(&buf, BYTES));
    CK(hipMalloc(&sink, 8));
    CK(hipMemset(buf, 0, BYTES));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount * 8;
    printf("# device %s, %d CUs, buffer %zu bytes, grid %d x 256\n", prop.name, prop.multiProcessorCount, BYTES, grid);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timed = [&](const char *name, double rd, double wr, auto launch) {
        launch();   // warm-up (counted too: the script divides by the number of dispatches)
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("TRUE %-28s read_bytes %.0f write_bytes %.0f  ms %.3f  GB/s %.0f\n", name, rd, wr, ms, (rd + wr) / ms / 1e6);
    };
    timed("cal_read_b64_plain", BYTES, 0, [&] { hipLaunchKernelGGL(cal_read_b64_plain, dim3(grid), dim3(256), 0, 0, (const double *)buf, BYTES / 8, sink); });
    timed("cal_read_b64_sc1", BYTES, 0, [&] { hipLaunchKernelGGL(cal_read_b64_sc1, dim3(grid), dim3(256), 0, 0, (const double *)buf, BYTES / 8, sink); });
    timed("cal_read_b128", BYTES, 0, [&] { hipLaunchKernelGGL(cal_read_b128, dim3(grid), dim3(256), 0, 0, (const double2 *)buf, BYTES / 16, sink); });
    timed("cal_read_b32", BYTES, 0, [&] { hipLaunchKernelGGL(cal_read_b32, dim3(grid), dim3(256), 0, 0, (const float *)buf, BYTES / 4, sink); });
    timed("cal_read_b64_sc1_pieces32", BYTES, 0, [&] { hipLaunchKernelGGL(cal_read_b64_sc1_pieces32, dim3(grid), dim3(256), 0, 0, (const double *)buf, BYTES / 8, sink); });
    timed("cal_write_b64", 0, BYTES, [&] { hipLaunchKernelGGL(cal_write_b64, dim3(grid), dim3(256), 0, 0, (double *)buf, BYTES / 8); });
    timed("cal_write_b128", 0, BYTES, [&] { hipLaunchKernelGGL(cal_write_b128, dim3(grid), dim3(256), 0, 0, (double2 *)buf, BYTES / 16); });
    timed("cal_write_b32", 0, BYTES, [&] { hipLaunchKernelGGL(cal_write_b32, dim3(grid), dim3(256), 0, 0, (float *)buf, BYTES / 4); });
    timed("cal_write_b64_pieces32", 0, BYTES, [&] { hipLaunchKernelGGL(cal_write_b64_pieces32, dim3(grid), dim3(256), 0, 0, (double *)buf, BYTES / 8); });
    {
        // resident waves: the occupancy the runtime reports for this kernel x CUs; the grid is exactly that many
        // blocks, so every wave's private array stays live for the whole launch
        constexpr int WORDS = 256;
        int occ = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, cal_spill_roundtrip<WORDS>, 256, 0));
        const int g = occ * prop.multiProcessorCount;
        const int rounds = 4;
        const double per_round = (double)g * 256 * WORDS * 8;
        printf("# cal_spill_roundtrip: %d blocks per CU resident, %d blocks, private array %d B per lane, footprint %.0f MiB\n", occ, g,
               WORDS * 8, per_round / 1048576.0);
        timed("cal_spill_roundtrip", per_round * rounds, per_round * rounds,
              [&] { hipLaunchKernelGGL(cal_spill_roundtrip<WORDS>, dim3(g), dim3(256), 0, 0, rounds, 3, sink); });
    }
    CK(hipDeviceSynchronize());
    return 0;
}
