#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((ext_vector_type(4))) __bf16 b4;
__global__ void k(uint16_t* out, int rowstride_elems) {
    __shared__ uint16_t lds[64*64];
    for (int i = threadIdx.x; i < 64*64; i += 64) lds[i] = (uint16_t)i;   // value = row*64+col for [64][64] image
    __syncthreads();
    int l = threadIdx.x;
    int g = l >> 4, i = l & 15;
    // 16-lane group g reads block rows 4g..4g+3 (k), cols 0..15: lane i reads row 4g + i/4, cols 4*(i%4)..+3
    uint16_t* p = &lds[(4*g + (i>>2)) * rowstride_elems + 4*(i&3)];
    s4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)p);
    for (int j = 0; j < 4; ++j) out[l*4+j] = (uint16_t)r[j];
}
int main() {
    uint16_t* d; hipMalloc(&d, 64*4*2);
    k<<<1,64>>>(d, 64);
    uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" (%d,%d)", h[l*4+j]/64, h[l*4+j]%64); printf("\n"); }
    return 0;
}
