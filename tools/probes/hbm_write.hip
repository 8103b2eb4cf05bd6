// Write-only (and write-mostly) HBM bandwidth on MI355X: the ceiling of a plane-writing kernel -- the dgrad kernel
// writes 4.9 KB per row and reads ~0.3 KB -- next to hbm_read.hip's read ceiling for the weight-gradient stage.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/hbm_write.hip -o /tmp/hbm_write && /tmp/hbm_write
// Every wave-level store instruction writes one contiguous KiB (64 lanes x 16 B), as the kernels' plane stores do.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT, int UNROLL>
__global__ __launch_bounds__(256) void write_stride(f32x4* __restrict__ x, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const f32x4 v = {1.0f, 2.0f, 3.0f, (float)threadIdx.x};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (NT) __builtin_nontemporal_store(v, x + i + u * stride);
            else x[i + u * stride] = v;
        }
    }
}

// each workgroup fills its own contiguous chunk (a row tile's planes)
template <bool NT, int UNROLL>
__global__ __launch_bounds__(256) void write_chunked(f32x4* __restrict__ x, size_t n4) {
    const size_t per = n4 / gridDim.x;
    f32x4* p = x + per * blockIdx.x;
    const f32x4 v = {1.0f, 2.0f, 3.0f, (float)threadIdx.x};
    for (size_t i = threadIdx.x; i + (UNROLL - 1) * 256 < per; i += UNROLL * 256) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (NT) __builtin_nontemporal_store(v, p + i + u * 256);
            else p[i + u * 256] = v;
        }
    }
}

// 16 stores per load (the dgrad kernel's mix: ~4.9 KB written, ~0.3 KB read per row)
template <bool NT>
__global__ __launch_bounds__(256) void write_16_read_1(f32x4* __restrict__ x, const f32x4* __restrict__ y, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t i = j; i + 15 * stride < n4; i += 16 * stride, j += stride) {
        const f32x4 v = y[j];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (NT) __builtin_nontemporal_store(v, x + i + u * stride);
            else x[i + u * stride] = v;
        }
    }
}

int main() {
    const size_t bytes = (size_t)8 << 30, n4 = bytes / 16;
    f32x4 *x, *y;
    if (hipMalloc(&x, bytes) != hipSuccess || hipMalloc(&y, bytes / 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(x, 0, bytes); hipMemset(y, 0, bytes / 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto time = [&](auto launch, const char* name, int grid, double moved) {
        launch(grid); hipDeviceSynchronize();
        hipEventRecord(a);
        for (int r = 0; r < 5; ++r) launch(grid);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-44s grid %6d  %.0f GB/s\n", name, grid, moved * 5 / (ms * 1e-3) / 1e9);
    };
    for (int grid : {256, 512, 1024, 2048, 4096, 16384}) {
        time([&](int g) { hipLaunchKernelGGL((write_stride<false, 8>), dim3(g), dim3(256), 0, 0, x, n4); }, "grid-stride, plain stores", grid, (double)bytes);
        time([&](int g) { hipLaunchKernelGGL((write_stride<true, 8>), dim3(g), dim3(256), 0, 0, x, n4); }, "grid-stride, non-temporal stores", grid, (double)bytes);
        time([&](int g) { hipLaunchKernelGGL((write_chunked<false, 8>), dim3(g), dim3(256), 0, 0, x, n4); }, "chunk per WG, plain stores", grid, (double)bytes);
        time([&](int g) { hipLaunchKernelGGL((write_chunked<true, 8>), dim3(g), dim3(256), 0, 0, x, n4); }, "chunk per WG, non-temporal stores", grid, (double)bytes);
        time([&](int g) { hipLaunchKernelGGL((write_16_read_1<true>), dim3(g), dim3(256), 0, 0, x, y, n4); }, "16 nt stores per load (bytes = W + R)", grid, (double)bytes * 17.0 / 16.0);
    }
    return 0;
}
