// Probe of ds_read_b64_tr_b16 semantics on gfx950: every lane supplies its own 8-byte LDS address;
// prints, per lane, the four 16-bit values it receives so the cross-lane transpose can be read off.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    // mode 0: lane l reads the linear 8-byte chunk l (elements 4l..4l+3)
    // mode 1: lane l reads row (l&15)>>2 of a 64-element-stride matrix, cols 4*((l&15)&3) + 16*((l>>4)&1), rows +8*(l>>5)
    int elem;
    if (mode == 0) elem = 4 * l;
    else {
        const int lam = l & 15, gam = l >> 4;
        const int row = (lam >> 2) + 8 * (gam >> 1), col = 16 * (gam & 1) + 4 * (lam & 3);
        elem = row * 64 + col;
    }
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + elem));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)r[j];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int j = 0; j < 4; ++j) {
                if (mode == 0) printf(" %4d", h[l * 4 + j]);
                else printf(" (r%2d,c%2d)", h[l * 4 + j] / 64, h[l * 4 + j] % 64);
            }
            printf("\n");
        }
    }
    return 0;
}
