// What the scratch arena's first-call allocations cost on the box: hipMalloc / first touch / hipFree by block size.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/malloc_cost.hip -o tools/ubench/malloc_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(float* p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.f; }
int main() {
    hipFree(0);
    const size_t G = (size_t)1 << 30;
    struct Case { size_t bytes; int count; } cases[] = {{2 * G, 10}, {10 * G, 2}, {20 * G, 1}, {256 << 20, 16}, {2 * G, 10}};
    for (auto c : cases) {
        std::vector<void*> ps;
        double t0 = now();
        for (int i = 0; i < c.count; ++i) { void* p = nullptr; if (hipMalloc(&p, c.bytes) != hipSuccess) { printf("malloc failed\n"); return 1; } ps.push_back(p); }
        double t1 = now();
        for (void* p : ps) hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, (float*)p, c.bytes / 4);
        hipDeviceSynchronize();
        double t2 = now();
        for (void* p : ps) hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, (float*)p, c.bytes / 4);
        hipDeviceSynchronize();
        double t3 = now();
        for (void* p : ps) hipFree(p);
        double t4 = now();
        printf("%2d x %6.2f GiB: hipMalloc %8.2f ms  first touch %8.2f ms  second touch %8.2f ms  hipFree %8.2f ms\n", c.count, (double)c.bytes / G, t1 - t0, t2 - t1, t3 - t2, t4 - t3);
    }
    return 0;
}
