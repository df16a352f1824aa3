// Does global_load_lds_dword (LDS-DMA) accept global addresses that are not 4-byte aligned?  (pais_tile.hpp stages image
// rows that start at arbitrary byte offsets.)  Every lane loads the dword at buf + shift + stride * lane into LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
__global__ void k(const unsigned char *buf, int shift, int stride, uint32_t *out)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[64];
    lds[threadIdx.x] = 0xdeadbeef;
    __syncthreads();
    const unsigned char *gsrc = buf + shift + (size_t)stride * threadIdx.x;
    unsigned keep;
    const unsigned dst = (unsigned)(uintptr_t)lds; // wave-uniform LDS byte address
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
    __syncthreads();
    out[threadIdx.x] = lds[threadIdx.x];
}
int main()
{
    const int N = 1 << 16;
    std::vector<unsigned char> h(N);
    for (int i = 0; i < N; ++i) h[i] = (unsigned char)(i * 7 + (i >> 8));
    unsigned char *d; uint32_t *o;
    hipMalloc(&d, N); hipMalloc(&o, 256);
    hipMemcpy(d, h.data(), N, hipMemcpyHostToDevice);
    int bad = 0;
    for (int shift = 0; shift < 8; ++shift)
        for (int stride : {4, 5, 76, 3277}) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, shift, stride, o);
            uint32_t r[64];
            hipMemcpy(r, o, 256, hipMemcpyDeviceToHost);
            int b = 0;
            for (int l = 0; l < 64; ++l) {
                uint32_t want;
                memcpy(&want, &h[shift + (size_t)stride * l], 4);
                b += r[l] != want;
            }
            printf("shift %d stride %4d: %s (%d lanes wrong)\n", shift, stride, b ? "WRONG" : "ok", b);
            bad += b;
        }
    printf(bad ? "LDS-DMA from unaligned global addresses: NOT usable\n" : "LDS-DMA from unaligned global addresses: ok\n");
    return 0;
}
