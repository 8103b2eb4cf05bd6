// The caller side of the path as kernels: what the reference's training loop does around
// render() with a dozen host / element-wise torch operations per step.
//
//   plnerf_uniform         counter-based uniform draws (philox.h): the t_rand / u tensors, world-size invariant
//   plnerf_normal          the same counters through Box-Muller: the density noise of raw2outputs (run_plnerf.py:568-570)
//   plnerf_select_rays     run_plnerf.py:1259-1281: N_rand distinct random pixels of one view -> their rays
//   plnerf_ndc_rays        run_nerf_helpers.py:184-201: the NDC warp of forward-facing rays
//                          (get_rays, run_nerf_helpers.py:162-171), unit view directions (run_plnerf.py:148-150),
//                          near / far columns and the target colours -- without building the H x W ray grid
//   plnerf_coarse_samples  run_plnerf.py:683-708: stratified depths (jitter drawn in the kernel or read from a tensor)
//                          and the sample positions, one launch
//   plnerf_image_loss      run_plnerf.py:1287-1296: img2mse(rgb, target) + img2mse(rgb0, target) and both gradients
//   plnerf_depth_loss      depth_supervised_exps/run_nerf_sample_based_depth.py:1126-1150: the same two image terms plus
//                          space_carving_weight * compute_space_carving_loss(pred_hyp, target_h)
//                          (model/run_nerf_helpers.py:52-86), and the three gradients, one launch
//   plnerf_embed_rows      depth_supervised_exps/run_nerf_sample_based_depth.py:52-68: run_network's input assembly for
//                          a network whose encoding is not the in-kernel one -- bounding-box affine, gamma(x) with an
//                          optional input scale (model/run_nerf_helpers.py:100-130: sin / cos(x * pi * 2^k)), the ray's
//                          direction encoding and the per-image camera code repeated over its samples -- one launch
//                          instead of ~100 element-wise / cat launches per network evaluation
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

// standard normal draws from the same counters (Box-Muller on the block's two uniform pairs): the density noise of
// raw2outputs (run_plnerf.py:568-570: randn * raw_noise_std) keyed on the GLOBAL ray id, so that a sharded batch sees the
// noise one rank would see -- the LLFF configurations train with raw_noise_std = 1
__global__ __launch_bounds__(256) void normal_kernel(const RngArgs g, const int R, const int n, float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int blocks_per_row = (n + 3) >> 2;
    if (idx >= (size_t)R * blocks_per_row) return;
    const int r = (int)(idx / blocks_per_row), b = (int)(idx - (size_t)r * blocks_per_row);
    uint32_t c[4] = {(uint32_t)(g.ray_id0 + r), (uint32_t)b, g.stream, g.step};
    philox4x32_10(c, g.seed_lo, g.seed_hi);
    float z[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float u1 = 1.0f - u01(c[2 * h]);          // (0, 1]: the logarithm is finite
        const float u2 = u01(c[2 * h + 1]);
        const float rad = sqrtf(-2.0f * logf(u1));
        float sn, cs;
        sincosf(6.28318530717958647692f * u2, &sn, &cs);
        z[2 * h] = rad * cs;
        z[2 * h + 1] = rad * sn;
    }
    float* row = out + (size_t)r * n + 4 * b;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (4 * b + k < n) row[k] = z[k];
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

// ---- normalised device coordinates of forward-facing rays (run_nerf_helpers.py:184-201), one thread per ray ----
// Every operation separately rounded, in the order torch evaluates the reference's expressions (the scalars sx, sy,
// near, 2 near, -2 near arrive already rounded to fp32, as torch rounds a Python scalar that meets an fp32 tensor).
struct NdcArgs {
    const float* rays_o;
    const float* rays_d;
    int n;
    float sx, sy, near, two_near, minus_two_near;
    float* o_out;
    float* d_out;
};

__global__ __launch_bounds__(256) void ndc_rays_kernel(const NdcArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const float ox = a.rays_o[3 * (size_t)i], oy = a.rays_o[3 * (size_t)i + 1], oz = a.rays_o[3 * (size_t)i + 2];
    const float dx = a.rays_d[3 * (size_t)i], dy = a.rays_d[3 * (size_t)i + 1], dz = a.rays_d[3 * (size_t)i + 2];
    const float t = -(a.near + oz) / dz;                          // shift the origin to the near plane (:186-187)
    const float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;
    const float ox_oz = px / pz, oy_oz = py / pz;
    a.o_out[3 * (size_t)i + 0] = (a.sx * px) / pz;                // o0, o1, o2 (:190-192): "s * x / z" is (s x) / z
    a.o_out[3 * (size_t)i + 1] = (a.sy * py) / pz;
    a.o_out[3 * (size_t)i + 2] = 1.0f + a.two_near / pz;
    a.d_out[3 * (size_t)i + 0] = a.sx * (dx / dz - ox_oz);        // d0, d1, d2 (:194-196)
    a.d_out[3 * (size_t)i + 1] = a.sy * (dy / dz - oy_oz);
    a.d_out[3 * (size_t)i + 2] = a.minus_two_near / pz;
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
// one; the caller scales).  n = 3 R elements (12,288 at N_rand 4096) over IL_BLOCKS workgroups: fp64 partial sums per
// workgroup, the last one to finish (ticket counter in the caller's zeroed workspace, which it resets) adds them in
// workgroup order -> deterministic.  (One workgroup walking the batch measured 14 us of serialised load latency.)
// `coarse_in`: the loss4 of an earlier launch on the coarse image alone (its [1] = mean((rgb0 - t)^2)): a caller that starts
// the coarse network's backward before the fine pass exists (rounds 3-4's two-stream schedules did; no caller in the package
// since round 5) computes the coarse term first; total = fine + that, as one launch over both images computes it.
constexpr int IL_BLOCKS = 16, IL_THREADS = 256;
__global__ __launch_bounds__(IL_THREADS) void image_loss_kernel(const float* __restrict__ rgb, const float* __restrict__ rgb0,
                                                                 const float* __restrict__ target, const int n,
                                                                 float* __restrict__ loss3 /* [4] */, float* __restrict__ g_rgb,
                                                                 float* __restrict__ g_rgb0, const float* __restrict__ coarse_in,
                                                                 double* __restrict__ partial, unsigned* __restrict__ ticket) {
    __shared__ double part[2][IL_THREADS / 64];
    __shared__ unsigned last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const float scale = 2.0f / (float)n;
    const int per = (n + IL_BLOCKS - 1) / IL_BLOCKS;
    const int lo = b * per, hi = min(n, lo + per);
    double s1 = 0.0, s0 = 0.0;
    for (int i = lo + tid; i < hi; i += IL_THREADS) {
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
        for (int w = 0; w < IL_THREADS / 64; ++w) { a1 += part[0][w]; a0 += part[1][w]; }
        partial[2 * b + 0] = a1;
        partial[2 * b + 1] = a0;
        __threadfence();
        last = atomicAdd(ticket, 1u) == IL_BLOCKS - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!last || tid != 0) return;
    __threadfence();
    double a1 = 0.0, a0 = 0.0;
    for (int w = 0; w < IL_BLOCKS; ++w) {
        a1 += __builtin_nontemporal_load(partial + 2 * w + 0);
        a0 += __builtin_nontemporal_load(partial + 2 * w + 1);
    }
    *ticket = 0u;      // left as the next launch expects it
    const float fine = (float)(a1 / (double)n);
    const float coarse = coarse_in ? coarse_in[1] : (float)(a0 / (double)n);
    loss3[0] = fine + coarse;
    loss3[1] = fine;
    loss3[2] = coarse;
    loss3[3] = -10.0f * log10f(fine);      // mse2psnr(img_loss) (run_nerf_helpers.py:18, run_plnerf.py:1290)
}

// ---- depth-supervised loss: DL_BLOCKS workgroups over contiguous slices, fp64 partial sums per workgroup in a fixed
// order, the last workgroup to finish (a ticket counter in the caller's workspace, which it resets) adds the partials in
// workgroup order: deterministic, and enough loads in flight (one workgroup walking 262,144 hypotheses measured 305 us
// of serialised memory latency) ----
struct DepthLossArgs {
    const float* rgb; const float* rgb0; const float* target;      // [R, 3]
    const float* hyp;          // pred_hyp [R, P]
    const float* target_h;     // [H, R, PT] with PT = 1 (one depth per ray and hypothesis) or P
    const float* mask;         // [R] or nullptr
    int R, P, H, PT;
    int is_joint;              // the hypothesis is chosen per IMAGE (per point column), not per ray (model/run_nerf_helpers.py:72-77)
    const int* joint_choice;   // is_joint: nullptr = choose here, from this call's rays; else [P] hypothesis per column, chosen by the
                               // caller from the column sums of EVERY shard of the batch (plnerf_depth_joint_sums + one all-reduce)
    float weight, threshold;
    float* loss5;              // {total, img, img0, space carving (unweighted), psnr of img}
    float* g_rgb; float* g_rgb0; float* g_hyp;
    double* partial;           // workspace: [DL_BLOCKS][3] partial sums ...
    unsigned* ticket;          // ... and the ticket counter behind them (zero between launches)
};
constexpr int DL_BLOCKS = 128, DL_THREADS = 256;

__global__ __launch_bounds__(DL_THREADS) void depth_loss_kernel(const DepthLossArgs a) {
    __shared__ double part[3][DL_THREADS / 64];
    __shared__ double jred[DL_THREADS / 64];
    __shared__ double jbest;
    __shared__ int jarg;
    __shared__ unsigned last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const int n = 3 * a.R;
    const float scale = 2.0f / (float)n;
    double s1 = 0.0, s0 = 0.0, sc = 0.0;
    {
        const int per = (n + DL_BLOCKS - 1) / DL_BLOCKS, lo = b * per, hi = min(n, lo + per);
        for (int i = lo + tid; i < hi; i += DL_THREADS) {
            const float t = a.target[i];
            const float d1 = a.rgb[i] - t;
            s1 += (double)(d1 * d1);
            a.g_rgb[i] = d1 * scale;
            if (a.rgb0) {
                const float d0 = a.rgb0[i] - t;
                s0 += (double)(d0 * d0);
                a.g_rgb0[i] = d0 * scale;
            }
        }
    }
    if (a.hyp && a.is_joint) {
        // is_joint (model/run_nerf_helpers.py:72-77): quantile_mean[h, p] = mean over the RAYS of distances[h, r, p]; the
        // hypothesis is chosen per point column, min over h; loss = mean over p.  A workgroup owns whole columns
        // (p = b, b + DL_BLOCKS, ...): it sums a column over all rays itself -- fixed thread-strided order, fixed tree --
        // so the choice needs no second launch and the result is deterministic; its partial is the chosen column sum.
        const float gs = a.weight / ((float)a.R * (float)a.P);
        for (int p = b; p < a.P; p += DL_BLOCKS) {
            const int h_lo = a.joint_choice ? a.joint_choice[p] : 0, h_hi = a.joint_choice ? h_lo + 1 : a.H;
            for (int h = h_lo; h < h_hi; ++h) {
                double sum = 0.0;
                for (int r = tid; r < a.R; r += DL_THREADS) {
                    const float m = a.mask ? a.mask[r] : 1.0f;
                    const float t = a.target_h[((size_t)h * a.R + r) * a.PT + (a.PT == 1 ? 0 : p)];
                    float d = fabsf(a.hyp[(size_t)r * a.P + p] - t) * m;
                    if (a.threshold > 0.0f && d < a.threshold) d = 0.0f;
                    sum += (double)d;
                }
                sum = wave_sum(sum);
                if (lane == 0) jred[wave] = sum;
                __syncthreads();
                if (tid == 0) {
                    double tot = 0.0;
                    for (int w = 0; w < DL_THREADS / 64; ++w) tot += jred[w];
                    // (compared as the reference compares them: fp32 means; first minimum on ties, like torch.min)
                    const float mean = (float)(tot / (double)a.R);
                    if (h == h_lo || mean < (float)(jbest / (double)a.R)) { jbest = tot; jarg = h; }
                }
                __syncthreads();
            }
            const int hs = jarg;
            if (tid == 0) sc += jbest;
            for (int r = tid; r < a.R; r += DL_THREADS) {
                const float m = a.mask ? a.mask[r] : 1.0f;
                const float t = a.target_h[((size_t)hs * a.R + r) * a.PT + (a.PT == 1 ? 0 : p)];
                const float diff = a.hyp[(size_t)r * a.P + p] - t;
                float gd = diff > 0.0f ? m : (diff < 0.0f ? -m : 0.0f);
                if (a.threshold > 0.0f && fabsf(diff) * m < a.threshold) gd = 0.0f;
                a.g_hyp[(size_t)r * a.P + p] = gd * gs;
            }
            __syncthreads();
        }
    } else if (a.hyp) {
        // distances[h, r, p] = mask[r] * |pred[r, p] - target[h, r, p]|  (torch.norm over a trailing axis of length 1),
        // zeroed below the threshold; best = min over h (first minimum on ties, like torch.min); loss = mean over r, p
        const int np = a.R * a.P;
        const float gs = a.weight / (float)np;
        const int per = (np + DL_BLOCKS - 1) / DL_BLOCKS, lo = b * per, hi = min(np, lo + per);
        for (int i = lo + tid; i < hi; i += DL_THREADS) {
            const int r = i / a.P, p = i - r * a.P;
            const float x = a.hyp[i];
            const float m = a.mask ? a.mask[r] : 1.0f;
            float best = 0.0f, gbest = 0.0f;
            for (int h = 0; h < a.H; ++h) {
                const float t = a.target_h[((size_t)h * a.R + r) * a.PT + (a.PT == 1 ? 0 : p)];
                const float diff = x - t;
                float d = fabsf(diff) * m;
                float gd = diff > 0.0f ? m : (diff < 0.0f ? -m : 0.0f);      // d |x| / dx with 0 at 0 (torch.norm's rule)
                if (a.threshold > 0.0f && d < a.threshold) { d = 0.0f; gd = 0.0f; }
                if (h == 0 || d < best) { best = d; gbest = gd; }
            }
            sc += (double)best;
            a.g_hyp[i] = gbest * gs;
        }
    }
    s1 = wave_sum(s1);
    s0 = wave_sum(s0);
    sc = wave_sum(sc);
    if (lane == 0) { part[0][wave] = s1; part[1][wave] = s0; part[2][wave] = sc; }
    __syncthreads();
    if (tid == 0) {
        double v[3] = {0.0, 0.0, 0.0};
        for (int w = 0; w < DL_THREADS / 64; ++w) { v[0] += part[0][w]; v[1] += part[1][w]; v[2] += part[2][w]; }
        a.partial[3 * b + 0] = v[0]; a.partial[3 * b + 1] = v[1]; a.partial[3 * b + 2] = v[2];
        __threadfence();
        last = atomicAdd(a.ticket, 1u) == (unsigned)(DL_BLOCKS - 1);
    }
    __syncthreads();
    if (last && tid == 0) {
        __threadfence();
        double a1 = 0.0, a0 = 0.0, ac = 0.0;
        for (int w = 0; w < DL_BLOCKS; ++w) {      // workgroup order, whatever order they finished in
            a1 += __builtin_nontemporal_load(a.partial + 3 * w + 0);
            a0 += __builtin_nontemporal_load(a.partial + 3 * w + 1);
            ac += __builtin_nontemporal_load(a.partial + 3 * w + 2);
        }
        *a.ticket = 0u;      // ready for the next launch
        const float fine = (float)(a1 / (double)n), coarse = (float)(a0 / (double)n);
        const float carve = a.hyp ? (float)(ac / ((double)a.R * (double)a.P)) : 0.0f;
        // loss = img_loss + weight * sc + img_loss0, in the reference's order (run_nerf_sample_based_depth.py:1131-1150)
        float total = fine;
        if (a.hyp) total = total + a.weight * carve;
        if (a.rgb0) total = total + coarse;
        a.loss5[0] = total;
        a.loss5[1] = fine;
        a.loss5[2] = coarse;
        a.loss5[3] = carve;
        a.loss5[4] = -10.0f * log10f(fine);
    }
}

// ---- run_network's input assembly: 64 rows per workgroup through LDS, so that the row-major [n_rows, C] matrix leaves
// as one contiguous, coalesced run per tile ----
struct EmbedArgs {
    const float* pts;          // [n_rows, 3]
    const float* viewdirs;     // [n_rows / spr, 3] or nullptr
    const float* cam;          // [n_cam] or nullptr
    int n_rows, spr, fx, fd, n_cam;      // fx, fd: frequency counts of the position / direction encodings
    float scale;               // the encoder's input scale: 1 (run_nerf_helpers.py:24-54) or pi (depth variant)
    float cx, cy, cz, bscale;  // bounding-box affine applied to the positions: (x - c) * bscale
    float* out;                // [n_rows, C], C = 3 + 6 fx (+ 3 + 6 fd + n_cam with view directions)
};
constexpr int EMB_ROWS = 64;

__global__ __launch_bounds__(256) void embed_rows_kernel(const EmbedArgs a) {
    extern __shared__ float tile[];      // [EMB_ROWS][C]
    const int cx_ = 3 + 6 * a.fx, cd_ = a.viewdirs ? 3 + 6 * a.fd : 0, C = cx_ + cd_ + (a.viewdirs ? a.n_cam : 0);
    const int tid = threadIdx.x, r = tid & (EMB_ROWS - 1), part = tid >> 6;      // 4 threads per row
    const int row0 = blockIdx.x * EMB_ROWS, row = row0 + r;
    const int rows_valid = min(EMB_ROWS, a.n_rows - row0);
    if (r < rows_valid) {
        float* o = tile + (size_t)r * C;
        // position: (x - center) * bb_scale, each rounded (run_nerf_sample_based_depth.py:56)
        const float c3[3] = {a.cx, a.cy, a.cz};
        float x[3], xs[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            x[d] = (a.pts[(size_t)row * 3 + d] - c3[d]) * a.bscale;
            xs[d] = x[d] * a.scale;      // (x * pi) * freq: the reference's association; the power of two is exact
        }
        if (part == 0) { o[0] = x[0]; o[1] = x[1]; o[2] = x[2]; }
        for (int k = part; k < a.fx; k += 4) {
            const float f = ldexpf(1.0f, k);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                float sv, cv;
                sincosf(xs[d] * f, &sv, &cv);
                o[3 + 6 * k + d] = sv;
                o[3 + 6 * k + 3 + d] = cv;
            }
        }
        if (a.viewdirs) {
            const float* v = a.viewdirs + (size_t)(row / a.spr) * 3;
            float* od = o + cx_;
            float vs[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) vs[d] = v[d] * a.scale;
            if (part == 1) { od[0] = v[0]; od[1] = v[1]; od[2] = v[2]; }
            for (int k = part; k < a.fd; k += 4) {
                const float f = ldexpf(1.0f, k);
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    float sv, cv;
                    sincosf(vs[d] * f, &sv, &cv);
                    od[3 + 6 * k + d] = sv;
                    od[3 + 6 * k + 3 + d] = cv;
                }
            }
            for (int c = part; c < a.n_cam; c += 4) od[cd_ + c] = a.cam[c];
        }
    }
    __syncthreads();
    const size_t base = (size_t)row0 * C;
    for (int i = tid; i < rows_valid * C; i += 256) a.out[base + i] = tile[i];
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

extern "C" int plnerf_normal(uint64_t seed, uint32_t stream_id, uint32_t step, int ray_id0, int R, int n, float* out,
                             plnerf_stream_t stream) {
    if (R < 0 || n < 1) return PLNERF_EINVAL;
    if (R == 0) return PLNERF_OK;
    if (!out) return PLNERF_EINVAL;
    const RngArgs g{(uint32_t)seed, (uint32_t)(seed >> 32), stream_id, step, ray_id0, 1};
    const size_t blocks = (size_t)R * ((n + 3) / 4);
    hipLaunchKernelGGL(normal_kernel, dim3((unsigned)((blocks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, R, n,
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

extern "C" int plnerf_ndc_rays(int H, int W, double focal, double near, const float* rays_o, const float* rays_d, int n,
                               float* o_out, float* d_out, plnerf_stream_t stream) {
    if (H < 1 || W < 1 || n < 0 || !(focal != 0.0)) return PLNERF_EINVAL;
    if (n == 0) return PLNERF_OK;
    if (!rays_o || !rays_d || !o_out || !d_out) return PLNERF_EINVAL;
    NdcArgs a{};
    a.rays_o = rays_o; a.rays_d = rays_d; a.n = n; a.o_out = o_out; a.d_out = d_out;
    a.sx = (float)(-1.0 / ((double)W / (2.0 * focal)));      // the reference's Python arithmetic: double, then one rounding
    a.sy = (float)(-1.0 / ((double)H / (2.0 * focal)));
    a.near = (float)near; a.two_near = (float)(2.0 * near); a.minus_two_near = (float)(-2.0 * near);
    hipLaunchKernelGGL(ndc_rays_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
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
                                 float* g_rgb, float* g_rgb0, const float* coarse_loss, void* workspace,
                                 plnerf_stream_t stream) {
    static_assert(IL_BLOCKS * 2 * sizeof(double) + sizeof(unsigned) <= PLNERF_IMAGE_LOSS_WORKSPACE_BYTES, "workspace");
    if (R < 1 || !rgb || !target || !loss3 || !g_rgb || (rgb0 && !g_rgb0) || (rgb0 && coarse_loss) || !workspace)
        return PLNERF_EINVAL;
    hipLaunchKernelGGL(image_loss_kernel, dim3(IL_BLOCKS), dim3(IL_THREADS), 0, (hipStream_t)stream, rgb, rgb0, target,
                       3 * R, loss3, g_rgb, g_rgb0, coarse_loss, (double*)workspace,
                       (unsigned*)((double*)workspace + IL_BLOCKS * 2));
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}

// ---- is_joint over a sharded batch: the column sums sum[h, p] = sum over this call's rays of distances[h, r, p] (fp64, fixed
// order), which the caller adds over the shards before choosing the hypothesis of each column ----
struct JointSumsArgs {
    const float* hyp; const float* target_h; const float* mask;
    int R, P, H, PT;
    float threshold;
    double* sums;      // [H, P]
};
__global__ __launch_bounds__(DL_THREADS) void depth_joint_sums_kernel(const JointSumsArgs a) {
    __shared__ double jred[DL_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = blockIdx.x % a.P, h = blockIdx.x / a.P;
    double sum = 0.0;
    for (int r = tid; r < a.R; r += DL_THREADS) {      // (the arithmetic of depth_loss_kernel's is_joint branch)
        const float m = a.mask ? a.mask[r] : 1.0f;
        const float t = a.target_h[((size_t)h * a.R + r) * a.PT + (a.PT == 1 ? 0 : p)];
        float d = fabsf(a.hyp[(size_t)r * a.P + p] - t) * m;
        if (a.threshold > 0.0f && d < a.threshold) d = 0.0f;
        sum += (double)d;
    }
    sum = wave_sum(sum);
    if (lane == 0) jred[wave] = sum;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int w = 0; w < DL_THREADS / 64; ++w) tot += jred[w];
        a.sums[(size_t)h * a.P + p] = tot;
    }
}

extern "C" int plnerf_depth_joint_sums(const float* pred_hyp, const float* target_h, const float* mask, int R, int n_points,
                                       int n_hyp, int target_points, float threshold, double* sums, plnerf_stream_t stream) {
    if (R < 1 || !pred_hyp || !target_h || !sums || n_points < 1 || n_hyp < 1 ||
        (target_points != 1 && target_points != n_points))
        return PLNERF_EINVAL;
    JointSumsArgs a{pred_hyp, target_h, mask, R, n_points, n_hyp, target_points, threshold, sums};
    hipLaunchKernelGGL(depth_joint_sums_kernel, dim3(n_points * n_hyp), dim3(DL_THREADS), 0, (hipStream_t)stream, a);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}

extern "C" int plnerf_depth_loss(const float* rgb, const float* rgb0, const float* target, const float* pred_hyp,
                                 const float* target_h, const float* mask, int R, int n_points, int n_hyp,
                                 int target_points, int is_joint, const int* joint_choice, float space_carving_weight,
                                 float threshold, float* loss5,
                                 float* g_rgb, float* g_rgb0, float* g_hyp, void* workspace, plnerf_stream_t stream) {
    static_assert(DL_BLOCKS * 3 * sizeof(double) + sizeof(unsigned) <= PLNERF_DEPTH_LOSS_WORKSPACE_BYTES, "workspace");
    if (R < 1 || !rgb || !target || !loss5 || !g_rgb || (rgb0 && !g_rgb0) || !workspace) return PLNERF_EINVAL;
    if (pred_hyp && (!target_h || !g_hyp || n_points < 1 || n_hyp < 1 || (target_points != 1 && target_points != n_points)))
        return PLNERF_EINVAL;
    DepthLossArgs a{rgb, rgb0, target, pred_hyp, target_h, mask, R, n_points, n_hyp, target_points, is_joint ? 1 : 0,
                    is_joint ? joint_choice : nullptr, space_carving_weight, threshold, loss5, g_rgb, g_rgb0, g_hyp, (double*)workspace,
                    (unsigned*)((double*)workspace + DL_BLOCKS * 3)};
    hipLaunchKernelGGL(depth_loss_kernel, dim3(DL_BLOCKS), dim3(DL_THREADS), 0, (hipStream_t)stream, a);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}

extern "C" int plnerf_embed_rows(const float* pts, const float* viewdirs, const float* cam, int n_rows,
                                 int samples_per_ray, int n_freqs_xyz, int n_freqs_dir, int n_cam, float input_scale,
                                 const float* bb_center_host, float bb_scale, float* embedded, plnerf_stream_t stream) {
    if (n_rows < 0 || n_freqs_xyz < 0 || n_freqs_xyz > 16 || n_freqs_dir < 0 || n_freqs_dir > 16 || n_cam < 0 || n_cam > 64)
        return PLNERF_EINVAL;
    if (n_rows == 0) return PLNERF_OK;
    if (!pts || !embedded || (viewdirs && samples_per_ray < 1) || (n_cam > 0 && (!cam || !viewdirs))) return PLNERF_EINVAL;
    EmbedArgs a{pts, viewdirs, cam, n_rows, samples_per_ray < 1 ? 1 : samples_per_ray, n_freqs_xyz, n_freqs_dir, n_cam,
                input_scale, bb_center_host ? bb_center_host[0] : 0.0f, bb_center_host ? bb_center_host[1] : 0.0f,
                bb_center_host ? bb_center_host[2] : 0.0f, bb_scale, embedded};
    const int C = 3 + 6 * n_freqs_xyz + (viewdirs ? 3 + 6 * n_freqs_dir + n_cam : 0);
    const size_t lds = (size_t)EMB_ROWS * C * sizeof(float);
    // (C reaches 262 at the widest admitted arguments: 67 KB, beyond the 64 KB a launch may ask for without this)
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)embed_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PLNERF_EINVAL;
    hipLaunchKernelGGL(embed_rows_kernel, dim3((n_rows + EMB_ROWS - 1) / EMB_ROWS), dim3(256), lds, (hipStream_t)stream, a);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}
