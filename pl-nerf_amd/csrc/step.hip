// The caller side of the path as kernels: what the reference's training loop does around
// render() with a dozen host / element-wise torch operations per step.
//
//   plnerf_uniform         counter-based uniform draws (philox.h): the t_rand / u tensors, world-size invariant
//   plnerf_select_rays     run_plnerf.py:1259-1281: N_rand distinct random pixels of one view -> their rays
//                          (get_rays, run_nerf_helpers.py:162-171), unit view directions (run_plnerf.py:148-150),
//                          near / far columns and the target colours -- without building the H x W ray grid
//   plnerf_coarse_samples  run_plnerf.py:683-708: stratified depths (jitter drawn in the kernel or read from a tensor)
//                          and the sample positions, one launch
//   plnerf_image_loss      run_plnerf.py:1287-1296: img2mse(rgb, target) + img2mse(rgb0, target) and both gradients
//
// Built with -ffp-contract=off like the other per-ray kernels: products and sums are rounded separately, in the
// reference's order.
#include "common.h"
#include "philox.h"

using namespace plnerf;

namespace {

constexpr int WAVES = 4;

__global__ __launch_bounds__(256) void uniform_kernel(const RngArgs g, const int R, const int n, float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // one Philox block = 4 columns
    const int blocks_per_row = (n + 3) >> 2;
    if (idx >= (size_t)R * blocks_per_row) return;
    const int r = (int)(idx / blocks_per_row), b = (int)(idx - (size_t)r * blocks_per_row);
    uint32_t c[4] = {(uint32_t)(g.ray_id0 + r), (uint32_t)b, g.stream, g.step};
    philox4x32_10(c, g.seed_lo, g.seed_hi);
    float* row = out + (size_t)r * n + 4 * b;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (4 * b + k < n) row[k] = u01(c[k]);
}

// ---- pixel choice: a keyed bijection of [0, M) (4-round Feistel network on 2 hb bits, cycle-walked into the
// domain), evaluated at the global ray ids: distinct ids -> distinct pixels, i.e. a draw WITHOUT replacement like the
// reference's np.random.choice(..., replace=False), with no H x W permutation to build.
struct PixelPerm {
    uint32_t key[4];
    uint32_t M;
    int hb;
};

__device__ __forceinline__ uint32_t mix32(uint32_t h) {
    h *= 0x9E3779B1u; h ^= h >> 15; h *= 0x85EBCA77u; h ^= h >> 13; h *= 0xC2B2AE3Du; h ^= h >> 16;
    return h;
}

__device__ __forceinline__ uint32_t perm_index(const PixelPerm& p, uint32_t x) {
    const uint32_t mask = (1u << p.hb) - 1u;
    do {
        uint32_t L = x >> p.hb, Rr = x & mask;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t F = mix32(Rr + p.key[r]) & mask;
            const uint32_t nl = Rr;
            Rr = L ^ F;
            L = nl;
        }
        x = (L << p.hb) | Rr;
    } while (x >= p.M);
    return x;
}

struct SelectArgs {
    int H, W;
    float fx, fy, cx, cy;
    float c2w[12];          // rows of the 3x4 camera-to-world matrix
    const float* image;     // [H, W, 3] or null
    int r0, c0, nr, nc;     // pixel window (precrop, run_plnerf.py:1261-1270) -- the whole image otherwise
    PixelPerm perm;
    int ray_id0, R;
    float near, far;
    float* rays_o;
    float* rays_d;
    float* viewdirs;
    float* near_out;
    float* far_out;
    float* target;
    int* pixels;            // [R, 2] (row, col) or null
};

__global__ __launch_bounds__(256) void select_rays_kernel(const SelectArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.R) return;
    const uint32_t pick = perm_index(a.perm, (uint32_t)(a.ray_id0 + i));
    const int row = a.r0 + (int)(pick / (uint32_t)a.nc), col = a.c0 + (int)(pick % (uint32_t)a.nc);
    // dirs = ((i - cx) / fx, -(j - cy) / fy, -1);  rays_d[k] = sum_j dirs[j] * c2w[k][j]   (run_nerf_helpers.py:166-169)
    const float d0 = ((float)col - a.cx) / a.fx, d1 = -((float)row - a.cy) / a.fy, d2 = -1.0f;
    float d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) d[k] = (d0 * a.c2w[4 * k + 0] + d1 * a.c2w[4 * k + 1]) + d2 * a.c2w[4 * k + 2];
    const float nrm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        a.rays_o[3 * (size_t)i + k] = a.c2w[4 * k + 3];
        a.rays_d[3 * (size_t)i + k] = d[k];
        if (a.viewdirs) a.viewdirs[3 * (size_t)i + k] = d[k] / nrm;
    }
    a.near_out[i] = a.near;
    a.far_out[i] = a.far;
    if (a.target && a.image) {
        const float* px = a.image + ((size_t)row * a.W + col) * 3;
        a.target[3 * (size_t)i + 0] = px[0];
        a.target[3 * (size_t)i + 1] = px[1];
        a.target[3 * (size_t)i + 2] = px[2];
    }
    if (a.pixels) { a.pixels[2 * (size_t)i] = row; a.pixels[2 * (size_t)i + 1] = col; }
}

// ---- coarse depths + positions, one wavefront per ray ----
__device__ __forceinline__ float coarse_depth(const float nr, const float fr, const float t, const int lindisp) {
    const float omt = 1.0f - t;
    if (!lindisp) return nr * omt + fr * t;                 // near * (1 - t) + far * t
    return 1.0f / (1.0f / nr * omt + 1.0f / fr * t);        // 1 / (1/near * (1 - t) + 1/far * t)
}

struct CoarseArgs {
    const float* rays_o;
    const float* rays_d;
    const float* near;
    const float* far;
    const float* t_vals;    // [S] = torch.linspace(0, 1, S) as the reference's CPU path computes it
    const float* t_rand;    // [R, S] or null
    RngArgs rng;            // used when perturb and t_rand is null
    int R, S, lindisp, perturb;
    float* z_vals;
    float* pts;
};

__global__ __launch_bounds__(256) void coarse_samples_kernel(const CoarseArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int ray = blockIdx.x * WAVES + wave;
    const bool live = ray < a.R;
    if (!live) ray = a.R - 1;
    const int S = a.S;
    float* zs = smem + (size_t)wave * ((S + 3) & ~3);
    const float nr = a.near[ray], fr = a.far[ray];
    for (int s = lane; s < S; s += 64) {
        float z = coarse_depth(nr, fr, a.t_vals[s], a.lindisp);
        if (a.perturb) {
            // mids = .5 (z[1:] + z[:-1]); upper = [mids, z[-1]]; lower = [z[0], mids]; z = lower + (upper - lower) t_rand
            const float zl = s > 0 ? coarse_depth(nr, fr, a.t_vals[s - 1], a.lindisp) : z;
            const float zu = s + 1 < S ? coarse_depth(nr, fr, a.t_vals[s + 1], a.lindisp) : z;
            const float lower = s > 0 ? 0.5f * (z + zl) : z;
            const float upper = s + 1 < S ? 0.5f * (zu + z) : z;
            const float t = a.t_rand ? a.t_rand[(size_t)ray * S + s] : rng_uniform(a.rng, a.rng.ray_id0 + ray, s);
            z = lower + (upper - lower) * t;
        }
        zs[s] = z;
        if (live) a.z_vals[(size_t)ray * S + s] = z;
    }
    __syncthreads();
    if (!live) return;
    const float o[3] = {a.rays_o[3 * (size_t)ray], a.rays_o[3 * (size_t)ray + 1], a.rays_o[3 * (size_t)ray + 2]};
    const float d[3] = {a.rays_d[3 * (size_t)ray], a.rays_d[3 * (size_t)ray + 1], a.rays_d[3 * (size_t)ray + 2]};
    float* prow = a.pts + (size_t)ray * 3 * S;
    if ((3 * S) % 4 == 0 && ((uintptr_t)a.pts & 15) == 0) {
        for (int q = lane; q < 3 * S / 4; q += 64) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = 4 * q + k, si = e / 3, c = e - 3 * si;
                const float oc = c == 0 ? o[0] : (c == 1 ? o[1] : o[2]);
                const float dc = c == 0 ? d[0] : (c == 1 ? d[1] : d[2]);
                v[k] = oc + dc * zs[si];
            }
            reinterpret_cast<float4*>(prow)[q] = make_float4(v[0], v[1], v[2], v[3]);
        }
    } else {
        for (int e = lane; e < 3 * S; e += 64) {
            const int si = e / 3, c = e - 3 * si;
            prow[e] = (c == 0 ? o[0] : (c == 1 ? o[1] : o[2])) + (c == 0 ? d[0] : (c == 1 ? d[1] : d[2])) * zs[si];
        }
    }
}

// ---- loss = mean((rgb - t)^2) + mean((rgb0 - t)^2), and d loss / d rgb, d loss / d rgb0 (for an upstream gradient of
// one; the caller scales).  One workgroup: n = 3 R elements (12,288 at N_rand 4096), fp64 partial sums in a fixed
// order -> deterministic.
__global__ __launch_bounds__(1024) void image_loss_kernel(const float* __restrict__ rgb, const float* __restrict__ rgb0,
                                                           const float* __restrict__ target, const int n,
                                                           float* __restrict__ loss3 /* [4] */, float* __restrict__ g_rgb,
                                                           float* __restrict__ g_rgb0) {
    __shared__ double part[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float scale = 2.0f / (float)n;
    double s1 = 0.0, s0 = 0.0;
    for (int i = tid; i < n; i += 1024) {
        const float t = target[i];
        const float d1 = rgb[i] - t;
        s1 += (double)(d1 * d1);
        g_rgb[i] = d1 * scale;
        if (rgb0) {
            const float d0 = rgb0[i] - t;
            s0 += (double)(d0 * d0);
            g_rgb0[i] = d0 * scale;
        }
    }
    s1 = wave_sum(s1);
    s0 = wave_sum(s0);
    if (lane == 0) { part[0][wave] = s1; part[1][wave] = s0; }
    __syncthreads();
    if (tid == 0) {
        double a1 = 0.0, a0 = 0.0;
        for (int w = 0; w < 16; ++w) { a1 += part[0][w]; a0 += part[1][w]; }
        const float fine = (float)(a1 / (double)n), coarse = (float)(a0 / (double)n);
        loss3[0] = fine + coarse;
        loss3[1] = fine;
        loss3[2] = coarse;
        loss3[3] = -10.0f * log10f(fine);      // mse2psnr(img_loss) (run_nerf_helpers.py:18, run_plnerf.py:1290)
    }
}

}  // namespace

extern "C" int plnerf_uniform(uint64_t seed, uint32_t stream_id, uint32_t step, int ray_id0, int R, int n, float* out,
                              plnerf_stream_t stream) {
    if (R < 0 || n < 1) return PLNERF_EINVAL;
    if (R == 0) return PLNERF_OK;
    if (!out) return PLNERF_EINVAL;
    const RngArgs g{(uint32_t)seed, (uint32_t)(seed >> 32), stream_id, step, ray_id0, 1};
    const size_t blocks = (size_t)R * ((n + 3) / 4);
    hipLaunchKernelGGL(uniform_kernel, dim3((unsigned)((blocks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, R, n,
                       out);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}

extern "C" int plnerf_select_rays(int H, int W, float fx, float fy, float cx, float cy, const float* c2w_host,
                                  const float* image, int crop_r0, int crop_c0, int crop_rows, int crop_cols,
                                  uint64_t seed, uint32_t step, int ray_id0, int R, float near, float far,
                                  float* rays_o, float* rays_d, float* viewdirs, float* near_out, float* far_out,
                                  float* target, int* pixels, plnerf_stream_t stream) {
    if (H < 1 || W < 1 || R < 0 || ray_id0 < 0 || !c2w_host) return PLNERF_EINVAL;
    if (crop_rows < 1 || crop_cols < 1 || crop_r0 < 0 || crop_c0 < 0 || crop_r0 + crop_rows > H || crop_c0 + crop_cols > W)
        return PLNERF_EINVAL;
    const uint64_t M = (uint64_t)crop_rows * (uint64_t)crop_cols;
    if (M > (1ull << 30) || (uint64_t)ray_id0 + (uint64_t)R > M) return PLNERF_ERANGE;   // distinct pixels only
    if (R == 0) return PLNERF_OK;
    if (!rays_o || !rays_d || !near_out || !far_out) return PLNERF_EINVAL;
    SelectArgs a{};
    a.H = H; a.W = W; a.fx = fx; a.fy = fy; a.cx = cx; a.cy = cy;
    for (int i = 0; i < 12; ++i) a.c2w[i] = c2w_host[i];
    a.image = image; a.r0 = crop_r0; a.c0 = crop_c0; a.nr = crop_rows; a.nc = crop_cols;
    int bits = 1;
    while ((1ull << bits) < M) ++bits;
    a.perm.hb = (bits + 1) / 2;
    if (a.perm.hb < 1) a.perm.hb = 1;
    a.perm.M = (uint32_t)M;
    uint32_t c[4] = {0x5e1ec7u, 0u, 0xffffffffu, step};     // round keys: one Philox block per (seed, step)
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    for (int i = 0; i < 4; ++i) a.perm.key[i] = c[i];
    a.ray_id0 = ray_id0; a.R = R; a.near = near; a.far = far;
    a.rays_o = rays_o; a.rays_d = rays_d; a.viewdirs = viewdirs; a.near_out = near_out; a.far_out = far_out;
    a.target = target; a.pixels = pixels;
    hipLaunchKernelGGL(select_rays_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}

extern "C" int plnerf_coarse_samples(const float* rays_o, const float* rays_d, const float* near, const float* far,
                                     const float* t_vals, const float* t_rand, uint64_t seed, uint32_t step,
                                     int ray_id0, int R, int S, int lindisp, int perturb, float* z_vals, float* pts,
                                     plnerf_stream_t stream) {
    if (R < 0 || S < 1) return PLNERF_EINVAL;
    if (S > PLNERF_MAX_SAMPLES) return PLNERF_ERANGE;
    if (R == 0) return PLNERF_OK;
    if (!rays_o || !rays_d || !near || !far || !t_vals || !z_vals || !pts) return PLNERF_EINVAL;
    CoarseArgs a{rays_o, rays_d, near, far, t_vals, t_rand,
                 RngArgs{(uint32_t)seed, (uint32_t)(seed >> 32), 0u, step, ray_id0, t_rand ? 0 : 1},
                 R, S, lindisp, perturb, z_vals, pts};
    const size_t lds = (size_t)WAVES * ((S + 3) & ~3) * sizeof(float);
    hipLaunchKernelGGL(coarse_samples_kernel, dim3((R + WAVES - 1) / WAVES), dim3(WAVES * 64), lds, (hipStream_t)stream,
                       a);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}

extern "C" int plnerf_image_loss(const float* rgb, const float* rgb0, const float* target, int R, float* loss3,
                                 float* g_rgb, float* g_rgb0, plnerf_stream_t stream) {
    if (R < 1 || !rgb || !target || !loss3 || !g_rgb || (rgb0 && !g_rgb0)) return PLNERF_EINVAL;
    hipLaunchKernelGGL(image_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, rgb, rgb0, target, 3 * R, loss3,
                       g_rgb, g_rgb0);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}
