// which XCD does workgroup b of a launch run on?  (HW_REG_XCC_ID of every workgroup of grids of several sizes; k_pso_ring assumes b % 8)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(unsigned *out)
{
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = x;
}
int main()
{
    unsigned *d;
    hipMalloc(&d, 1 << 20);
    for (int grid : {8, 64, 3072, 20000}) {
        hipMemset(d, 0xff, 1 << 20);
        hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, d);
        std::vector<unsigned> h(grid);
        hipMemcpy(h.data(), d, grid * 4, hipMemcpyDeviceToHost);
        int mismatch = 0, cnt[16] = {0};
        for (int b = 0; b < grid; ++b) { cnt[h[b] & 15]++; if ((h[b] & 15) != (unsigned)(b % 8)) ++mismatch; }
        printf("grid %6d: raw[0..3] = %08x %08x %08x %08x  workgroups with XCC_ID != b %% 8: %d   per XCC:", grid, h[0], h[1], h[2], h[3], mismatch);
        for (int x = 0; x < 16; ++x) if (cnt[x]) printf(" %d:%d", x, cnt[x]);
        printf("\n");
    }
    return 0;
}
