// pmc_calib.hip -- known-byte-count micro-kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
// (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reads 1/2 of the bytes of a wide coalesced 16 B/lane stream; other access
// widths and WRITE_SIZE are uncalibrated -- "calibrate on a known byte count in your own access pattern").
//
//   hipcc --offload-arch=gfx950 -O3 tools/calib/pmc_calib.hip -o tools/calib/pmc_calib
//   rocprofv3 --pmc FETCH_SIZE -d out -o run -- ./tools/calib/pmc_calib        (and a second pass with WRITE_SIZE)
//
// Every kernel touches a 2 GiB buffer once (8x the 256 MiB Infinity Cache, so re-use cannot hide traffic), three dispatches each.
// Patterns:
//   calib_read16   one 16-byte load per lane, coalesced (the pattern the guide calibrated: factor 2)
//   calib_read8    one 8-byte load per lane, coalesced (k_x2 / k_xq reading a dictionary COLUMN: lane i reads row i)
//   calib_sector8  one 8-byte load per 64-byte sector: lane i reads the double at byte 504 * i  (k_xq reading a dictionary ROW
//                  of a column-major record with 63 rows: consecutive entries are 63 doubles apart)
//   calib_write16 / calib_write8 / calib_wsector8   the same three patterns as stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

__global__ void calib_read16(const double2 *__restrict__ in, size_t n, double *__restrict__ out) {
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const double2 v = in[i]; acc += v.x + v.y; }
    if (acc == 12345.678) out[0] = acc;
}
__global__ void calib_read8(const double *__restrict__ in, size_t n, double *__restrict__ out) {
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += in[i];
    if (acc == 12345.678) out[0] = acc;
}
// n_elem loads, element j at double index j * stride
__global__ void calib_sector8(const double *__restrict__ in, size_t n_elem, size_t stride, double *__restrict__ out) {
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_elem; i += (size_t)gridDim.x * blockDim.x) acc += in[i * stride];
    if (acc == 12345.678) out[0] = acc;
}
__global__ void calib_write16(double2 *__restrict__ o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) o[i] = make_double2(1.0, 2.0);
}
__global__ void calib_write8(double *__restrict__ o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) o[i] = 1.0;
}
__global__ void calib_wsector8(double *__restrict__ o, size_t n_elem, size_t stride) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_elem; i += (size_t)gridDim.x * blockDim.x) o[i * stride] = 1.0;
}

int main() {
    const size_t bytes = size_t(2) << 30, nd = bytes / 8;
    double *buf = nullptr, *out = nullptr;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(buf, 0, bytes));
    const size_t stride = 63, n_sec = nd / stride;   // 504-byte stride: every load in its own 64-byte sector
    const dim3 g(256 * 16), b(256);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto timed = [&](const char *name, double useful, auto launch) {
        for (int r = 0; r < 3; ++r) {
            CHECK(hipEventRecord(e0, 0));
            launch();
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::printf("{\"kernel\": \"%s\", \"useful_bytes\": %.0f, \"ms\": %.4f, \"useful_GBs\": %.1f}\n", name, useful, ms, useful / ms / 1e6);
        }
    };
    timed("calib_read16", (double)bytes, [&] { hipLaunchKernelGGL(calib_read16, g, b, 0, 0, reinterpret_cast<const double2 *>(buf), nd / 2, out); });
    timed("calib_read8", (double)bytes, [&] { hipLaunchKernelGGL(calib_read8, g, b, 0, 0, buf, nd, out); });
    timed("calib_sector8", 8.0 * n_sec, [&] { hipLaunchKernelGGL(calib_sector8, g, b, 0, 0, buf, n_sec, stride, out); });
    timed("calib_write16", (double)bytes, [&] { hipLaunchKernelGGL(calib_write16, g, b, 0, 0, reinterpret_cast<double2 *>(buf), nd / 2); });
    timed("calib_write8", (double)bytes, [&] { hipLaunchKernelGGL(calib_write8, g, b, 0, 0, buf, nd); });
    timed("calib_wsector8", 8.0 * n_sec, [&] { hipLaunchKernelGGL(calib_wsector8, g, b, 0, 0, buf, n_sec, stride); });
    CHECK(hipDeviceSynchronize());
    std::printf("{\"n_sectors\": %zu, \"buffer_bytes\": %zu}\n", n_sec, bytes);
    return 0;
}
