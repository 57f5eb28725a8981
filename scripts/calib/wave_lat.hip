// Single-wave memory latency vs. number of independent 512-byte (one 128-B line x 4) loads in flight, cold data.
// build: hipcc --offload-arch=gfx950 -O3 -o wave_lat wave_lat.hip ; run: ./wave_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int N, int STRIDE_DOUBLES>
__global__ void k(const double *buf, long long *out, double *sink, int nwaves_active) {
    const int lane = threadIdx.x & 63;
    const double *p = buf + (size_t)blockIdx.x * (1 << 16) + lane;
    __builtin_amdgcn_s_waitcnt(0);
    long long t0 = wall_clock64();
    double v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = p[(size_t)i * STRIDE_DOUBLES];
    double s = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) s += v[i];
    __builtin_amdgcn_s_waitcnt(0);
    long long t1 = wall_clock64();
    if (lane == 0) { out[blockIdx.x] = t1 - t0; }
    sink[blockIdx.x * 64 + lane] = s;
}
template <int N, int S> void run(const double *buf, long long *out, double *sink, int blocks, double *flush, size_t flush_n) {
    hipMemset(flush, 1, flush_n);               // push the test buffer out of L2 / MALL
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<N, S>), dim3(blocks), dim3(64), 0, 0, buf, out, sink, blocks);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks); hipMemcpy(h.data(), out, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; long long mx = 0; for (auto x : h) { avg += x; mx = x > mx ? x : mx; }
    printf("loads/wave %3d stride %5d B  blocks %5d : avg %.2f us  max %.2f us  (%.1f ns per 512-B load)\n", N, S * 8, blocks, avg / blocks / 100.0, mx / 100.0, avg / blocks * 10.0 / N);
}
int main() {
    double *buf, *sink, *flush; long long *out;
    const size_t n = (size_t)4096 * (1 << 16);   // 2 GiB
    hipMalloc(&buf, n * 8); hipMemset(buf, 0, n * 8); hipMalloc(&sink, 4096 * 64 * 8); hipMalloc(&out, 4096 * 8);
    const size_t flush_n = (size_t)1 << 30; hipMalloc(&flush, flush_n);
    for (int blocks : {1, 256, 2048}) {
        run<1, 64>(buf, out, sink, blocks, flush, flush_n);
        run<8, 64>(buf, out, sink, blocks, flush, flush_n);
        run<24, 64>(buf, out, sink, blocks, flush, flush_n);
        run<48, 64>(buf, out, sink, blocks, flush, flush_n);
        run<96, 64>(buf, out, sink, blocks, flush, flush_n);
        run<48, 512>(buf, out, sink, blocks, flush, flush_n);    // one load per 4 KiB
    }
    return 0;
}
