// probe of ds_read_b64_tr_b16 semantics on gfx950 (prints what each lane receives)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* o) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    // lane l supplies the address of "pixel row" p = 8*(l>>4) + ((l&15)>>2), 4-channel piece (l&3); row pitch 64 elements
    const int p = 8 * (l >> 4) + ((l & 15) >> 2);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + p * 64 + 4 * (l & 3)));
    for (int j = 0; j < 4; ++j) o[l * 4 + j] = v[j];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d   (row,ch)= (%d,%d) (%d,%d) (%d,%d) (%d,%d)\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3],
        h[4*l]/64, h[4*l]%64, h[4*l+1]/64, h[4*l+1]%64, h[4*l+2]/64, h[4*l+2]%64, h[4*l+3]/64, h[4*l+3]%64);
    return 0;
}
