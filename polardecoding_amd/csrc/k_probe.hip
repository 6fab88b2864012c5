// k_probe.hip -- polar_testing_math: the kernels' scalar arithmetic on caller-chosen operands.  TEST LIBRARY ONLY
// (libpolar_hip_testing.so, include/polar_hip_testing.h).
#include "polar_host.h"
#include "../../include/polar_hip_testing.h"
#include "probe_kernel.h"

#ifndef POLAR_TESTING
#error "k_probe.hip belongs to libpolar_hip_testing.so (-DPOLAR_TESTING)"
#endif

extern "C" {

int polar_testing_math(int op, int is_f32, const void *a, const void *b, void *out, size_t n, int device)
{
    if (op < 0 || op > polar::PROBE_CHK_TAB || !a || !b || !out) return POLAR_EINVAL;
    if (n == 0) return POLAR_OK;
    DeviceGuard guard(device);
    const size_t es = is_f32 ? 4 : 8;
    void *da = nullptr, *db = nullptr, *d_out = nullptr;
    int rc = POLAR_OK;
    if (hipMalloc(&da, n * es) != hipSuccess || hipMalloc(&db, n * es) != hipSuccess || hipMalloc(&d_out, n * es) != hipSuccess)
        rc = POLAR_ENOMEM;
    if (!rc && (hipMemcpy(da, a, n * es, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(db, b, n * es, hipMemcpyHostToDevice) != hipSuccess))
        rc = POLAR_EDEVICE;
    if (!rc) {
        const int grid = (int)std::min<size_t>((n + 255) / 256, 1024);
        if (is_f32)
            hipLaunchKernelGGL(polar::k_probe_math<float>, dim3(grid), dim3(256), polar::Lut<float>::bytes + 16 + 64 * sizeof(float) + polar::Stair<float>::bytes, 0, op,
                               (const float *)da, (const float *)db, (float *)d_out, n);
        else
            hipLaunchKernelGGL(polar::k_probe_math<double>, dim3(grid), dim3(256), polar::Lut<double>::bytes + 16 + 64 * sizeof(double) + polar::Stair<double>::bytes, 0, op,
                               (const double *)da, (const double *)db, (double *)d_out, n);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
            hipMemcpy(out, d_out, n * es, hipMemcpyDeviceToHost) != hipSuccess)
            rc = POLAR_EDEVICE;
    }
    if (da) (void)hipFree(da);
    if (db) (void)hipFree(db);
    if (d_out) (void)hipFree(d_out);
    return rc;
}

}  // extern "C"
