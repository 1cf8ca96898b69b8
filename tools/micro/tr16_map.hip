// Prints what ds_read_b64_tr_b16 delivers: LDS image [row][64 cols] of shorts with value = row * 100 + col;
// lane l supplies the address of (row = (l & 15) >> 2, col = 16 * (l >> 4) + 4 * (l & 3)) -- the per-16-lane-group rule of
// cdna_hip_programming.md T10.  Expected: lane i of group g receives column 16 g + i of rows 0..3.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/tr16_map.hip -o /tmp/tr16_map && /tmp/tr16_map
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[8 * 64];
    for (int i = threadIdx.x; i < 8 * 64; i += 64) lds[i] = (short)((i / 64) * 100 + (i % 64));
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + q * 64 + 16 * g + 4 * p));
    out[l * 4 + 0] = v.x; out[l * 4 + 1] = v.y; out[l * 4 + 2] = v.z; out[l * 4 + 3] = v.w;
}
int main() {
    short* d; short h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) { printf(" %4d", h[l * 4 + j]); if (h[l * 4 + j] != j * 100 + 16 * (l >> 4) + (l & 15)) ++bad; }
        printf("\n");
    }
    printf("mismatches vs expected (row j, col 16g+i): %d\n", bad);
    return 0;
}
