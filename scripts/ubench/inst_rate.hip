// inst_rate.hip -- issue cost (cycles per wave64 instruction per SIMD) of the instructions the cost kernel uses.
// Build: hipcc --offload-arch=gfx950 -O3 inst_rate.hip -o inst_rate ; run on the MI355X.
// Each test: every wave runs ITER iterations of 8 independent copies of one instruction (asm volatile, so nothing
// is folded); 4 waves per SIMD on every SIMD.  cycles/instr = elapsed_cycles * 1 / (ITER * 8 * waves_per_simd).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define ITER 4096
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define KERNEL(name, decl, body)                                                   \
    __global__ __launch_bounds__(256) void name(double *out, long long *cyc)       \
    {                                                                              \
        decl;                                                                      \
        long long t0 = clock64();                                                  \
        for (int it = 0; it < ITER; ++it) { body; }                                \
        long long t1 = clock64();                                                  \
        if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                   \
        out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7; \
    }
#define DECL_D double r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; double a = 1.0000001, b = 0.5; float fa = threadIdx.x; int ia = threadIdx.x; (void)a; (void)b; (void)fa; (void)ia

#define FMA(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r##i) : "v"(a), "v"(b));
KERNEL(k_fma, DECL_D, REP8(FMA))
#define MUL(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r##i) : "v"(a));
KERNEL(k_mul, DECL_D, REP8(MUL))
#define ADD(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(r##i) : "v"(b));
KERNEL(k_add, DECL_D, REP8(ADD))
#define FRACT(i) asm volatile("v_fract_f64 %0, %0" : "+v"(r##i));
KERNEL(k_fract, DECL_D, REP8(FRACT))
#define RNDNE(i) asm volatile("v_rndne_f64 %0, %0" : "+v"(r##i));
KERNEL(k_rndne, DECL_D, REP8(RNDNE))
#define RCP(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(r##i));
KERNEL(k_rcp, DECL_D, REP8(RCP))
#define CVTF(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(r##i) : "v"(fa));
KERNEL(k_cvt_f64_f32, DECL_D, REP8(CVTF))
#define CVTI(i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r##i) : "v"(ia));
KERNEL(k_cvt_f64_i32, DECL_D, REP8(CVTI))
#define CVTU(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(r##i) : "v"(ia));
KERNEL(k_cvt_f64_u32, DECL_D, REP8(CVTU))
#define CVTDI(i) { int t_; asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(t_) : "v"(r##i)); asm volatile("" :: "v"(t_)); }
KERNEL(k_cvt_i32_f64, DECL_D, REP8(CVTDI))
#define DIVS(i) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(r##i) : "v"(a) : "vcc");
KERNEL(k_div_scale, DECL_D, REP8(DIVS))
#define DIVF(i) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(r##i) : "v"(a), "v"(b));
KERNEL(k_div_fixup, DECL_D, REP8(DIVF))
#define DIVM(i) asm volatile("v_div_fmas_f64 %0, %0, %1, %2" : "+v"(r##i) : "v"(a), "v"(b) : "vcc");
KERNEL(k_div_fmas, DECL_D, REP8(DIVM))
#define MED3(i) { int t_; asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(t_) : "v"(ia), "v"(2), "v"(100)); asm volatile("" :: "v"(t_)); }
KERNEL(k_med3_i32, DECL_D, REP8(MED3))
#define MAD64(i) { long long t_; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(t_) : "v"(ia), "v"(ia), "v"(t0) : "vcc"); asm volatile("" :: "v"(t_)); }
KERNEL(k_mad_u64_u32, DECL_D, REP8(MAD64))
#define ADDF(i) { float t_; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(t_) : "v"(fa), "v"(fa)); asm volatile("" :: "v"(t_)); }
KERNEL(k_sub_f32, DECL_D, REP8(ADDF))
#define MOV64(i) asm volatile("v_mov_b64 %0, %1" : "=v"(r##i) : "v"(a));
KERNEL(k_mov_b64, DECL_D, REP8(MOV64))
#define CND(i) { int t_; asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(t_) : "v"(ia), "v"(ia) : "vcc"); asm volatile("" :: "v"(t_)); }
KERNEL(k_cndmask, DECL_D, REP8(CND))
#define CMP(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(r##i), "v"(a) : "vcc");
KERNEL(k_cmp_f64, DECL_D, REP8(CMP))

// LDS: every lane reads the same address (broadcast), 8 independent reads per iteration
__global__ __launch_bounds__(256) void k_lds_bcast(double *out, long long *cyc)
{
    __shared__ double buf[512];
    for (int i = threadIdx.x; i < 512; i += 256) buf[i] = i;
    __syncthreads();
    double acc = 0;
    long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
        const int o = (it & 31) * 8;
        double v0, v1, v2, v3, v4, v5, v6, v7;
        asm volatile("ds_read_b64 %0, %1" : "=v"(v0) : "v"((o + 0) * 8));
        asm volatile("ds_read_b64 %0, %1" : "=v"(v1) : "v"((o + 1) * 8));
        asm volatile("ds_read_b64 %0, %1" : "=v"(v2) : "v"((o + 2) * 8));
        asm volatile("ds_read_b64 %0, %1" : "=v"(v3) : "v"((o + 3) * 8));
        asm volatile("ds_read_b64 %0, %1" : "=v"(v4) : "v"((o + 4) * 8));
        asm volatile("ds_read_b64 %0, %1" : "=v"(v5) : "v"((o + 5) * 8));
        asm volatile("ds_read_b64 %0, %1" : "=v"(v6) : "v"((o + 6) * 8));
        asm volatile("ds_read_b64 %0, %1" : "=v"(v7) : "v"((o + 7) * 8));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("" :: "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7));
    }
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
// LDS: per-lane distinct addresses (stride 8 B), 8 reads per iteration
__global__ __launch_bounds__(256) void k_lds_lane(double *out, long long *cyc)
{
    __shared__ double buf[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) buf[i] = i;
    __syncthreads();
    long long t0 = clock64();
    const int lo = (threadIdx.x & 63) * 8;
    for (int it = 0; it < ITER; ++it) {
        double v0, v1, v2, v3, v4, v5, v6, v7;
        asm volatile("ds_read_b64 %0, %1" : "=v"(v0) : "v"(lo));
        asm volatile("ds_read_b64 %0, %1 offset:512" : "=v"(v1) : "v"(lo));
        asm volatile("ds_read_b64 %0, %1 offset:1024" : "=v"(v2) : "v"(lo));
        asm volatile("ds_read_b64 %0, %1 offset:1536" : "=v"(v3) : "v"(lo));
        asm volatile("ds_read_b64 %0, %1 offset:2048" : "=v"(v4) : "v"(lo));
        asm volatile("ds_read_b64 %0, %1 offset:2560" : "=v"(v5) : "v"(lo));
        asm volatile("ds_read_b64 %0, %1 offset:3072" : "=v"(v6) : "v"(lo));
        asm volatile("ds_read_b64 %0, %1 offset:3584" : "=v"(v7) : "v"(lo));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("" :: "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7));
    }
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = 0;
}

// LDS: wave-uniform 16-byte reads (a camera's homography record in the tile kernels: five per camera and trip)
__global__ __launch_bounds__(256) void k_lds_b128_bcast(double *out, long long *cyc)
{
    __shared__ __attribute__((aligned(16))) double buf[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) buf[i] = i;
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
        const int o = (it & 31) * 128;
        typedef int v4i __attribute__((ext_vector_type(4)));
        v4i v0, v1, v2, v3, v4, v5, v6, v7;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v0) : "v"(o));
        asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(v1) : "v"(o));
        asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(v2) : "v"(o));
        asm volatile("ds_read_b128 %0, %1 offset:48" : "=v"(v3) : "v"(o));
        asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(v4) : "v"(o));
        asm volatile("ds_read_b128 %0, %1 offset:80" : "=v"(v5) : "v"(o));
        asm volatile("ds_read_b128 %0, %1 offset:96" : "=v"(v6) : "v"(o));
        asm volatile("ds_read_b128 %0, %1 offset:112" : "=v"(v7) : "v"(o));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("" :: "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7));
    }
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = 0;
}
// LDS: byte reads at scattered per-lane addresses (the bilinear taps from the tiles: a smooth map of the window pixels -- lanes
// of a row 1..2 bytes apart, rows a tile stride apart)
__global__ __launch_bounds__(256) void k_lds_u8_taps(double *out, long long *cyc)
{
    __shared__ unsigned char buf[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) buf[i] = (unsigned char)i;
    __syncthreads();
    long long t0 = clock64();
    const int l = threadIdx.x & 63;
    const int a0 = ((l % 51) * 5) / 4 + (l / 51) * 84; // ~1.25 bytes per window pixel along a row, 84-byte tile stride
    int acc = 0;
    for (int it = 0; it < ITER; ++it) {
        const int o = a0 + (it & 63) * 100;
        int v0, v1, v2, v3, v4, v5, v6, v7;
        asm volatile("ds_read_u8 %0, %1" : "=v"(v0) : "v"(o));
        asm volatile("ds_read_u8 %0, %1 offset:1" : "=v"(v1) : "v"(o));
        asm volatile("ds_read_u8 %0, %1 offset:84" : "=v"(v2) : "v"(o));
        asm volatile("ds_read_u8 %0, %1 offset:85" : "=v"(v3) : "v"(o));
        asm volatile("ds_read_u8 %0, %1 offset:4096" : "=v"(v4) : "v"(o));
        asm volatile("ds_read_u8 %0, %1 offset:4097" : "=v"(v5) : "v"(o));
        asm volatile("ds_read_u8 %0, %1 offset:4180" : "=v"(v6) : "v"(o));
        asm volatile("ds_read_u8 %0, %1 offset:4181" : "=v"(v7) : "v"(o));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("" :: "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7));
    }
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
#define RDL(i) { int t_; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(t_) : "v"(ia)); asm volatile("" :: "s"(t_)); }
KERNEL(k_readlane, DECL_D, REP8(RDL))
// s_barrier of a 16-wave workgroup (one per CU): cycles per barrier when every wave arrives at once
__global__ __launch_bounds__(1024) void k_barrier16(double *out, long long *cyc)
{
    long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
        asm volatile("s_barrier" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
    }
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
    if (threadIdx.x < 256) out[blockIdx.x * 256 + threadIdx.x] = 0;
}

typedef void (*kern_t)(double *, long long *);
int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    double *out;
    long long *cyc, h;
    const int wgPerCU = 4; // 4 workgroups x 4 waves = 16 waves per CU = 4 per SIMD
    hipMalloc(&out, sizeof(double) * 256 * cus * wgPerCU);
    hipMalloc(&cyc, sizeof(long long));
    struct { const char *n; kern_t k; } T[] = {
        {"v_fma_f64", k_fma}, {"v_mul_f64", k_mul}, {"v_add_f64", k_add}, {"v_fract_f64", k_fract}, {"v_rndne_f64", k_rndne},
        {"v_rcp_f64", k_rcp}, {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_cvt_f64_i32", k_cvt_f64_i32}, {"v_cvt_f64_u32", k_cvt_f64_u32},
        {"v_cvt_i32_f64", k_cvt_i32_f64}, {"v_div_scale_f64", k_div_scale}, {"v_div_fixup_f64", k_div_fixup}, {"v_div_fmas_f64", k_div_fmas},
        {"v_med3_i32", k_med3_i32}, {"v_mad_u64_u32", k_mad_u64_u32}, {"v_sub_f32", k_sub_f32}, {"v_mov_b64", k_mov_b64},
        {"v_cndmask_b32", k_cndmask}, {"v_cmp_lt_f64", k_cmp_f64}, {"ds_read_b64 broadcast", k_lds_bcast}, {"ds_read_b64 per-lane", k_lds_lane},
        {"ds_read_b128 broadcast", k_lds_b128_bcast}, {"ds_read_u8 tile taps", k_lds_u8_taps}, {"v_readlane_b32", k_readlane}};
    printf("%d CUs; cycles per wave64 instruction per SIMD with 4 waves/SIMD (clock64 ticks = shader cycles)\n", cus);
    for (auto &t : T) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(t.k, dim3(cus * wgPerCU), dim3(256), 0, 0, out, cyc);
            hipDeviceSynchronize();
        }
        hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-24s %6.2f\n", t.n, (double)h / ((double)ITER * 8 * 4));
    }
    // (LDS rows above: 16 waves of a CU share ONE LDS pipe -- the figure is cycles per instruction per SIMD with all four SIMDs
    //  issuing, i.e. a quarter of the pipe each: cycles per instruction of the CU's LDS pipe = the figure / 4)
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_barrier16, dim3(cus), dim3(1024), 0, 0, out, cyc);
        hipDeviceSynchronize();
    }
    hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-24s %6.2f   (cycles per s_barrier of a 16-wave workgroup, every wave arriving at once)\n", "s_barrier x16 waves", (double)h / ((double)ITER * 8));
    return 0;
}
