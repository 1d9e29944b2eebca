// How long after a kernel's last store has reached pinned host memory does hipStreamSynchronize return?  (round 6: is a spin on a published
// word a cheaper end of a level than the synchronisation?)   hipcc --offload-arch=gfx950 -O2 tools/ubench/sync_latency.hip -o tools/ubench/sync_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k_work(int *dst, int v, long long spin) {
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { __hip_atomic_store(dst, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}
int main() {
    int *flag; hipHostMalloc(&flag, 64, hipHostMallocDefault); *flag = 0;
    int *dflag; hipHostGetDevicePointer((void **)&dflag, flag, 0);
    hipStream_t st; hipStreamCreate(&st);
    std::vector<double> t_spin, t_sync, t_both;
    for (int it = 1; it <= 300; ++it) {
        auto a = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, dflag, it, 20000ll);   // ~10 us of device work
        if (it % 2) {
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != it) {}
            auto b = std::chrono::steady_clock::now();
            hipStreamSynchronize(st);
            auto c = std::chrono::steady_clock::now();
            t_spin.push_back(std::chrono::duration<double, std::micro>(b - a).count());
            t_both.push_back(std::chrono::duration<double, std::micro>(c - b).count());
        } else {
            hipStreamSynchronize(st);
            auto b = std::chrono::steady_clock::now();
            t_sync.push_back(std::chrono::duration<double, std::micro>(b - a).count());
        }
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("launch -> flag seen by a spin: %.1f us;  launch -> hipStreamSynchronize returns: %.1f us;  synchronise AFTER the flag was seen: %.1f us more\n", med(t_spin), med(t_sync), med(t_both));
    return 0;
}
