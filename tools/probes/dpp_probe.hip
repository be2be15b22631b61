#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    int lane = threadIdx.x;
    int v = lane * 10;
    int r = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);   // wave_shr:1
    int r2 = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);  // wave_shl:1
    out[lane] = r; out[64 + lane] = r2;
}
int main() {
    int* d; hipMalloc(&d, 128 * 4);
    k<<<1, 64>>>(d);
    int h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 6; i++) printf("%d ", h[i]); printf("... %d %d | ", h[62], h[63]);
    for (int i = 0; i < 4; i++) printf("%d ", h[64 + i]); printf("... %d %d\n", h[64 + 62], h[64 + 63]);
    return 0;
}
