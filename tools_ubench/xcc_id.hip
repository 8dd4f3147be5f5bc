// Which XCD does block b run on?  s_getreg_b32 HW_REG_XCC_ID (hwreg 20, bits 3:0) per block of a 256-block grid, against the
// b % 8 placement the PAIR variants count on for SPEED (never for correctness).   hipcc --offload-arch=gfx950 xcc_id.hip -o xcc_id.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11));
}
int main() {
    const int nb = 256;
    unsigned* d; unsigned h[nb];
    hipMalloc(&d, nb * sizeof(unsigned));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k, dim3(nb), dim3(512), 160 * 1024 - 64, 0, d);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        int same = 0, pairs_same = 0;
        for (int b = 0; b < nb; ++b) same += (h[b] == (unsigned)(b % 8));
        for (int b = 0; b < nb; ++b) if (((b >> 3) & 1) == 0) pairs_same += (h[b] == h[b + 8]);
        printf("rep %d: blocks on XCD b %% 8: %d / %d; pairs (b, b + 8) on one XCD: %d / %d; first 16:", rep, same, nb, pairs_same, nb / 2);
        for (int b = 0; b < 16; ++b) printf(" %u", h[b]);
        printf("\n");
    }
    return 0;
}
