// tools/ubench_rmw.hip — what an in-place read-modify-write sweep of a 16 GB u64 array achieves on one MI355X, by shape:
//   A  grid-stride, 256 threads, one 16-byte load + add + store a thread and trip (the plain streaming form)
//   B  one workgroup of 1024 threads a 128 KB block: eight 16-byte loads a thread, then eight stores (k_sxb_consume's tail)
//   C  as B with 256 threads a 32 KB block
//   R  read only (B's loads, summed), W write only (B's stores)
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_rmw.hip -o tools/ubench_rmw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
__global__ __launch_bounds__(256) void kA(ulonglong2* p, size_t n2) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        ulonglong2 v = p[i]; v.x += 1; v.y += 2; p[i] = v;
    }
}
template <int NT, int MODE>
__global__ __launch_bounds__(NT) void kB(ulonglong2* p, u64* sink) {
    ulonglong2* b = p + (size_t)blockIdx.x * NT * 8;
    ulonglong2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { v[u].x = v[u].y = 0; if (MODE != 2) v[u] = b[threadIdx.x + u * NT]; }
    if (MODE == 1) { u64 s = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u].x + v[u].y;
        if (s == 0x123456789ull) sink[0] = s; return; }
#pragma unroll
    for (int u = 0; u < 8; ++u) { v[u].x += 1; v[u].y += 2; b[threadIdx.x + u * NT] = v[u]; }
}
int main() {
    const size_t bytes = (size_t)16 << 30, n2 = bytes / 16;
    ulonglong2* p; u64* sink;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) return 1;
    (void)hipMemset(p, 0, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto time = [&](const char* name, double moved, auto launch) {
        launch(); (void)hipDeviceSynchronize();
        float best = 1e9f;
        for (int r = 0; r < 3; ++r) { (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; }
        printf("%-58s %7.3f ms  %6.2f TB/s of moved bytes\n", name, best, moved / (best * 1e-3) / 1e12);
    };
    time("A  grid-stride RMW, 256 threads, 16 B a trip", 2.0 * bytes, [&] { hipLaunchKernelGGL(kA, dim3(256 * 16), dim3(256), 0, 0, p, n2); });
    time("B  RMW, 1024 threads a 128 KB block (8 loads, 8 stores)", 2.0 * bytes, [&] { hipLaunchKernelGGL((kB<1024, 0>), dim3(n2 / 8192), dim3(1024), 0, 0, p, sink); });
    time("C  RMW, 256 threads a 32 KB block", 2.0 * bytes, [&] { hipLaunchKernelGGL((kB<256, 0>), dim3(n2 / 2048), dim3(256), 0, 0, p, sink); });
    time("R  read only, 1024 threads a 128 KB block", 1.0 * bytes, [&] { hipLaunchKernelGGL((kB<1024, 1>), dim3(n2 / 8192), dim3(1024), 0, 0, p, sink); });
    time("W  write only, 1024 threads a 128 KB block", 1.0 * bytes, [&] { hipLaunchKernelGGL((kB<1024, 2>), dim3(n2 / 8192), dim3(1024), 0, 0, p, sink); });
    time("R  read only, 256 threads a 32 KB block", 1.0 * bytes, [&] { hipLaunchKernelGGL((kB<256, 1>), dim3(n2 / 2048), dim3(256), 0, 0, p, sink); });
    time("W  write only, 256 threads a 32 KB block", 1.0 * bytes, [&] { hipLaunchKernelGGL((kB<256, 2>), dim3(n2 / 2048), dim3(256), 0, 0, p, sink); });
    return 0;
}
