// A torch-free host of the C ABI (include/plnerf_hip.h): what a maintainer binding libplnerf_hip.so from C, cgo or ctypes
// would write for one network evaluation and its backward -- device memory from the HIP runtime, every pointer a plain
// device pointer, the stream NULL.  Test infrastructure (tests/test_gpu_parity.py::test_c_host_without_torch builds it with
// g++, feeds it a file of inputs and compares the file of outputs with the oracle); not part of the product.
//
//   c_abi_gpu <precision> <R> <S> <in.bin> <out.bin>
//
// The chain: plnerf_mlp_pack_weights -> plnerf_mlp_fwd (run_network, run_plnerf.py:78-92) -> plnerf_quad_fwd (raw2outputs,
// :553-624, piecewise-linear, midpoint colours, white background) -> plnerf_quad_bwd -> plnerf_mlp_bwd (autograd of the two,
// run_plnerf.py:1300) -> plnerf_adam_step on the first weight tensor (:1302-1303).
// in.bin  (fp32, in this order): the 24 parameter tensors in state_dict order, pts [R,S,3], viewdirs [R,3], z [R,S],
//         near [R], far [R], rays_d [R,3], g_rgb [R,3], g_depth [R], g_acc [R]
// out.bin (fp32): raw [R,S,4], rgb [R,3], disp [R], acc [R], depth [R], weights [R,S+1], g_raw [R,S,4], the 24 gradients,
//         the first weight tensor after one Adam step
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "plnerf_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 10; } } while (0)
#define PL_OK(x) do { int rc_ = (x); if (rc_ != PLNERF_OK) { std::fprintf(stderr, "%s: %s\n", #x, plnerf_error_string(rc_)); return 11; } } while (0)

namespace {
constexpr int W = 256, XYZ = 63, DIR = 27;

std::vector<size_t> param_sizes() {      // state_dict order (run_nerf_helpers.py:87-101)
    std::vector<size_t> n;
    for (int i = 0; i < 8; ++i) {
        const size_t fan_in = i == 0 ? XYZ : (i == 5 ? W + XYZ : W);
        n.push_back(W * fan_in);
        n.push_back(W);
    }
    n.push_back((size_t)(W / 2) * (W + DIR)); n.push_back(W / 2);      // views_linears.0
    n.push_back((size_t)W * W); n.push_back(W);                        // feature_linear
    n.push_back(W); n.push_back(1);                                    // alpha_linear
    n.push_back(3 * (W / 2)); n.push_back(3);                          // rgb_linear
    return n;
}

float* to_device(const std::vector<float>& h) {
    float* d = nullptr;
    if (hipMalloc((void**)&d, h.size() * sizeof(float) + 16) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}
}  // namespace

int main(int argc, char** argv) {
    if (argc != 6 && argc != 7) { std::fprintf(stderr, "usage: %s precision R S in.bin out.bin [fwd_kernel]\n", argv[0]); return 2; }
    const int prec = std::atoi(argv[1]), R = std::atoi(argv[2]), S = std::atoi(argv[3]);
    const int fwd_kernel = argc == 7 ? std::atoi(argv[6]) : PLNERF_FWD_KERNEL_AUTO;      // (the suite's forced-kernel passes hand theirs in)
    const int n_rows = R * S;
    if (plnerf_version() != PLNERF_VERSION) { std::fprintf(stderr, "library / header version mismatch\n"); return 3; }
    std::FILE* in = std::fopen(argv[4], "rb");
    if (!in) return 4;
    auto read = [&](size_t n) { std::vector<float> v(n); if (std::fread(v.data(), 4, n, in) != n) v.clear(); return v; };
    const std::vector<size_t> sizes = param_sizes();
    const float* params[PLNERF_N_PARAM_TENSORS];
    float* params_dev[PLNERF_N_PARAM_TENSORS];
    float* grads[PLNERF_N_PARAM_TENSORS];
    for (int i = 0; i < PLNERF_N_PARAM_TENSORS; ++i) {
        const std::vector<float> h = read(sizes[i]);
        if (h.empty()) return 5;
        params[i] = params_dev[i] = to_device(h);
        HIP_OK(hipMalloc((void**)&grads[i], sizes[i] * sizeof(float)));
        if (!params[i]) return 6;
    }
    const std::vector<float> h_pts = read((size_t)n_rows * 3), h_vd = read((size_t)R * 3), h_z = read(n_rows), h_near = read(R),
                             h_far = read(R), h_d = read((size_t)R * 3), h_grgb = read((size_t)R * 3), h_gdep = read(R),
                             h_gacc = read(R);
    std::fclose(in);
    if (h_gacc.empty()) return 5;
    float *pts = to_device(h_pts), *vd = to_device(h_vd), *z = to_device(h_z), *near = to_device(h_near), *far = to_device(h_far),
          *rays_d = to_device(h_d), *g_rgb = to_device(h_grgb), *g_depth = to_device(h_gdep), *g_acc = to_device(h_gacc);
    if (!pts || !vd || !z || !near || !far || !rays_d || !g_rgb || !g_depth || !g_acc) return 6;

    // weights into the kernels' fragment order; the status word behind them starts at zero (the library only ORs into it)
    const size_t packed_bytes = plnerf_mlp_packed_bytes(prec);
    if (packed_bytes == 0) { std::fprintf(stderr, "precision mode %d is not built\n", prec); return 7; }
    void* packed = nullptr;
    HIP_OK(hipMalloc(&packed, packed_bytes));
    HIP_OK(hipMemset(packed, 0, packed_bytes));
    PL_OK(plnerf_mlp_pack_weights(params, prec, XYZ, DIR, packed, nullptr));

    // forward with saved state
    void *saved = nullptr, *workspace = nullptr;
    HIP_OK(hipMalloc(&saved, plnerf_mlp_saved_bytes(n_rows, prec)));
    HIP_OK(hipMalloc(&workspace, plnerf_mlp_bwd_workspace_bytes(n_rows, prec)));
    float *raw, *rgb, *disp, *acc, *depth, *weights, *tau, *T, *g_raw;
    HIP_OK(hipMalloc((void**)&raw, (size_t)n_rows * 4 * 4));
    HIP_OK(hipMalloc((void**)&g_raw, (size_t)n_rows * 4 * 4));
    HIP_OK(hipMalloc((void**)&rgb, (size_t)R * 3 * 4));
    HIP_OK(hipMalloc((void**)&disp, (size_t)R * 4));
    HIP_OK(hipMalloc((void**)&acc, (size_t)R * 4));
    HIP_OK(hipMalloc((void**)&depth, (size_t)R * 4));
    HIP_OK(hipMalloc((void**)&weights, (size_t)R * (S + 1) * 4));
    HIP_OK(hipMalloc((void**)&tau, (size_t)R * (S + 2) * 4));
    HIP_OK(hipMalloc((void**)&T, (size_t)R * (S + 2) * 4));
    PL_OK(plnerf_mlp_fwd(packed, prec, pts, vd, nullptr, XYZ, DIR, n_rows, S, 1.0f, 0.0f, raw, saved, fwd_kernel,
                         nullptr));
    PL_OK(plnerf_quad_fwd(raw, z, near, far, rays_d, nullptr, R, S, PLNERF_MODE_LINEAR, PLNERF_COLOR_MIDPOINT, 1, 0, rgb, disp,
                          acc, depth, weights, tau, T, nullptr));
    // backward: d loss / d maps come from the caller
    PL_OK(plnerf_quad_bwd(raw, z, near, far, rays_d, nullptr, R, S, PLNERF_MODE_LINEAR, PLNERF_COLOR_MIDPOINT, 1, 0, g_rgb, g_depth,
                          g_acc, nullptr, nullptr, nullptr, g_raw, nullptr, nullptr));
    const int layout = plnerf_mlp_saved_layout(prec, 0, fwd_kernel);
    if (layout < 0) return 8;
    PL_OK(plnerf_mlp_bwd(packed, prec, g_raw, nullptr, 0, XYZ, DIR, n_rows, saved, layout, nullptr, 0.0f, workspace, grads, nullptr,
                         nullptr));
    // one Adam step on pts_linears.0.weight (step 1, lr 5e-4, no clipping, no guard words)
    float *m, *v;
    HIP_OK(hipMalloc((void**)&m, sizes[0] * 4));
    HIP_OK(hipMalloc((void**)&v, sizes[0] * 4));
    HIP_OK(hipMemset(m, 0, sizes[0] * 4));
    HIP_OK(hipMemset(v, 0, sizes[0] * 4));
    PL_OK(plnerf_adam_step(params_dev[0], grads[0], m, v, (int64_t)sizes[0], 5e-4f, 0.9f, 0.999f, 1e-8f, 1, 1.0f, 0.0f, nullptr,
                           nullptr, nullptr, nullptr));
    HIP_OK(hipDeviceSynchronize());

    std::FILE* out = std::fopen(argv[5], "wb");
    if (!out) return 4;
    auto write = [&](const float* d, size_t n) {
        std::vector<float> h(n);
        if (hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
        return std::fwrite(h.data(), 4, n, out) == n;
    };
    bool ok = write(raw, (size_t)n_rows * 4) && write(rgb, (size_t)R * 3) && write(disp, R) && write(acc, R) && write(depth, R) &&
              write(weights, (size_t)R * (S + 1)) && write(g_raw, (size_t)n_rows * 4);
    for (int i = 0; ok && i < PLNERF_N_PARAM_TENSORS; ++i) ok = write(grads[i], sizes[i]);
    ok = ok && write(params_dev[0], sizes[0]);
    std::fclose(out);
    return ok ? 0 : 9;
}
