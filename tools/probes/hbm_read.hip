// Read-only HBM bandwidth on MI355X with plain 16-byte loads: what a streaming reader (the weight-gradient stage)
// can hope for.  hipcc --offload-arch=gfx950 -O3 tools/probes/hbm_read.hip -o /tmp/hbm_read && /tmp/hbm_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int UNROLL>
__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ x, size_t n4, float* out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; i < n4; i += stride) { const float4 v = x[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    const float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123.456f) out[0] = s;   // keep the loads
}

// each workgroup streams its own contiguous chunk (the access pattern of a split-K row range)
template <int UNROLL>
__global__ __launch_bounds__(256) void read_chunked(const float4* __restrict__ x, size_t n4, float* out) {
    const size_t per = n4 / gridDim.x;
    const float4* p = x + per * blockIdx.x;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = threadIdx.x; i + (UNROLL - 1) * 256 < per; i += UNROLL * 256) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = p[i + u * 256];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    const float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123.456f) out[0] = s;
}

int main() {
    const size_t bytes = (size_t)8 << 30, n4 = bytes / 16;
    float4* x; float* out;
    hipMalloc(&x, bytes); hipMalloc(&out, 4);
    hipMemset(x, 0, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto time = [&](auto launch, const char* name, int grid) {
        launch(grid); hipDeviceSynchronize();
        hipEventRecord(a);
        for (int r = 0; r < 5; ++r) launch(grid);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-28s grid %6d  %.0f GB/s\n", name, grid, bytes * 5 / (ms * 1e-3) / 1e9);
    };
    for (int grid : {256, 512, 1024, 2048, 4096, 16384}) {
        time([&](int g) { hipLaunchKernelGGL(read_kernel<4>, dim3(g), dim3(256), 0, 0, x, n4, out); }, "grid-stride, 4 loads/thread", grid);
        time([&](int g) { hipLaunchKernelGGL(read_kernel<8>, dim3(g), dim3(256), 0, 0, x, n4, out); }, "grid-stride, 8 loads/thread", grid);
        time([&](int g) { hipLaunchKernelGGL(read_chunked<8>, dim3(g), dim3(256), 0, 0, x, n4, out); }, "chunk per WG, 8 loads/thread", grid);
    }
    return 0;
}
