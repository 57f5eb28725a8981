// Producer -> consumer hand-off latency between two waves, same XCD vs different XCD, device-scope (sc1) vs L1-bypass
// (sc0) loads.  Producer writes 6 KB with write-through stores, drains, sets a flag; consumer polls the flag, then
// gathers the 6 KB.  Reported: producer start -> consumer done, and the part after the flag was seen.
// build: hipcc --offload-arch=gfx950 -O3 -o xcd_handoff xcd_handoff.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15; }   // HW_REG_XCC_ID
template <int MODE>   // 0: sc1 loads, 1: sc0 loads (workgroup scope: bypass L1, hit L2)
__global__ void k(double *buf, int *flag, long long *ts, int *xcc, int consumer_block, int epoch) {
    const int lane = threadIdx.x;
    if (blockIdx.x == 0) {                              // producer
        if (lane == 0) { xcc[0] = xcc_id(); ts[0] = wall_clock64(); }
        for (int i = 0; i < 12; ++i) __hip_atomic_store(buf + i * 64 + lane, (double)(epoch + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) { ts[1] = wall_clock64(); __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    } else if ((int)blockIdx.x == consumer_block) {     // consumer
        if (lane == 0) xcc[1] = xcc_id();
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch && spins < (1 << 22)) { __builtin_amdgcn_s_sleep(2); ++spins; }
        long long t2 = wall_clock64();
        double s = 0;
        for (int i = 0; i < 12; ++i) s += MODE == 0 ? __hip_atomic_load(buf + i * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                    : __hip_atomic_load(buf + i * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_s_waitcnt(0);
        long long t3 = wall_clock64();
        if (lane == 0) { ts[2] = t2; ts[3] = t3; }
        buf[4096 + lane] = s;                            // sum check: 12 * epoch + 66 per lane
    }
}
int main() {
    double *buf; int *flag, *xcc; long long *ts;
    hipMalloc(&buf, 8192 * 8); hipMalloc(&flag, 64); hipMalloc(&xcc, 64); hipMalloc(&ts, 64);
    hipMemset(buf, 0, 8192 * 8); hipMemset(flag, 0, 64);
    int epoch = 1;
    for (int mode = 0; mode < 2; ++mode)
        for (int cb : {8, 16, 1, 2, 3, 4, 9}) {
            for (int rep = 0; rep < 3; ++rep, ++epoch) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(32), dim3(64), 0, 0, buf, flag, ts, xcc, cb, epoch);
                else hipLaunchKernelGGL(k<1>, dim3(32), dim3(64), 0, 0, buf, flag, ts, xcc, cb, epoch);
                hipDeviceSynchronize();
                long long h[4]; int x[2]; double chk[64];
                hipMemcpy(h, ts, 32, hipMemcpyDeviceToHost); hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost); hipMemcpy(chk, buf + 4096, 512, hipMemcpyDeviceToHost);
                if (rep == 2) printf("loads %s consumer block %2d: producer xcc %d consumer xcc %d | store+drain %.2f us, flag seen +%.2f us, gather %.2f us | data %s\n",
                       mode ? "sc0" : "sc1", cb, x[0], x[1], (h[1] - h[0]) / 100.0, (h[2] - h[1]) / 100.0, (h[3] - h[2]) / 100.0,
                       chk[5] == 12.0 * epoch + 66 ? "ok" : "STALE");
            }
        }
    return 0;
}
