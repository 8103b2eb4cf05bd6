// Which access pattern lets 256 streaming workgroups (the shape of wgrad_main_kernel: one per CU, 512 threads, two planes,
// 32 KB stages) pull the most from the HBM?  `chunk`: every workgroup walks its own contiguous range (what the split-K
// row ranges do today); `sweep`: stage s of workgroup w is stage s * G + w of the plane (all workgroups move through the
// plane together); each with plain and non-temporal loads.
// hipcc --offload-arch=gfx950 -O3 tools/probes/hbm_read_patterns.hip -o /tmp/hbm_read_patterns && /tmp/hbm_read_patterns
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v4f __attribute__((ext_vector_type(4)));

template <bool SWEEP, bool NT, int STAGE_KB>
__global__ __launch_bounds__(512) void reader(const v4f* __restrict__ p0, const v4f* __restrict__ p1, size_t stages, float* out) {
    constexpr int PER = STAGE_KB * 1024 / 16 / 512;      // 16-byte loads per thread per stage and plane
    const size_t per_wg = stages / gridDim.x;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t s = 0; s < per_wg; ++s) {
        const size_t st = SWEEP ? s * gridDim.x + blockIdx.x : blockIdx.x * per_wg + s;
        const v4f* a = p0 + st * (STAGE_KB * 64) + threadIdx.x;
        const v4f* b = p1 + st * (STAGE_KB * 64) + threadIdx.x;
        v4f va[PER], vb[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            va[u] = NT ? __builtin_nontemporal_load(a + u * 512) : a[u * 512];
            vb[u] = NT ? __builtin_nontemporal_load(b + u * 512) : b[u * 512];
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) acc += va[u] + vb[u];
    }
    const float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123.456f) out[0] = s;
}

int main() {
    const size_t plane = (size_t)4 << 30;
    v4f *p0, *p1; float* out;
    hipMalloc(&p0, plane); hipMalloc(&p1, plane); hipMalloc(&out, 4);
    hipMemset(p0, 0, plane); hipMemset(p1, 0, plane);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto time = [&](auto kern, const char* name, int grid, int stage_kb) {
        const size_t stages = plane / ((size_t)stage_kb * 1024);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, p0, p1, stages, out); hipDeviceSynchronize();
        float best = 1e9f;
        for (int r = 0; r < 4; ++r) {
            hipEventRecord(a);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, p0, p1, stages, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
        }
        printf("%-34s grid %5d  %.0f GB/s\n", name, grid, 2.0 * plane / (best * 1e-3) / 1e9);
    };
    for (int grid : {256, 512}) {
        time(reader<false, false, 32>, "chunk, plain, 32 KB stages", grid, 32);
        time(reader<true, false, 32>, "sweep, plain, 32 KB stages", grid, 32);
        time(reader<false, true, 32>, "chunk, non-temporal, 32 KB stages", grid, 32);
        time(reader<true, true, 32>, "sweep, non-temporal, 32 KB stages", grid, 32);
        time(reader<false, false, 64>, "chunk, plain, 64 KB stages", grid, 64);
        time(reader<true, false, 64>, "sweep, plain, 64 KB stages", grid, 64);
        time(reader<false, true, 64>, "chunk, non-temporal, 64 KB stages", grid, 64);
        time(reader<true, true, 64>, "sweep, non-temporal, 64 KB stages", grid, 64);
    }
    return 0;
}
