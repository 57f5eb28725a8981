// fetch_calib.hip — known-byte-count streaming kernels to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
// for the access widths the linearisation kernel uses (8 B and 4 B per lane, coalesced), as
// /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes ("calibrate on a known byte count in your own access pattern").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void read8(const double *p, size_t n, double *out) {
    double s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 12345.678) out[0] = s;
}
__global__ void read4(const int *p, size_t n, int *out) {
    int s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 123456789) out[0] = s;
}
__global__ void read16(const double2 *p, size_t n, double *out) {
    double s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = p[i]; s += v.x + v.y; }
    if (s == 12345.678) out[0] = s;
}
__global__ void write8(double *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0;
}
int main() {
    const size_t bytes = (size_t)1 << 30;            // 1 GiB: far beyond the 256 MiB Infinity Cache
    double *a, *o; hipMalloc(&a, bytes); hipMalloc(&o, 64); hipMemset(a, 0, bytes);
    for (int r = 0; r < 3; ++r) {
        read8<<<2048, 256>>>(a, bytes / 8, o);
        read4<<<2048, 256>>>((const int *)a, bytes / 4, (int *)o);
        read16<<<2048, 256>>>((const double2 *)a, bytes / 16, o);
        write8<<<2048, 256>>>(a, bytes / 8);
    }
    hipDeviceSynchronize();
    printf("each kernel moves %zu bytes\n", bytes);
    return 0;
}
