// launch_gap.hip -- time of a chain of dependent short kernels: stream launches vs a captured hipGraph.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_work(double *p, int iters)
{
    double v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0000001 + 1e-9;
    p[threadIdx.x] = v;
}
int main()
{
    double *d;
    hipMalloc(&d, 64 * sizeof(double));
    hipMemset(d, 0, 64 * sizeof(double));
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int CH = 33;
    for (int iters : {100, 4000}) {
        // stream
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a, st);
            for (int i = 0; i < CH; ++i) hipLaunchKernelGGL(k_work, dim3(64), dim3(64), 0, st, d, iters);
            hipEventRecord(b, st);
            hipStreamSynchronize(st);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep == 2) printf("iters %5d stream : %.1f us total, %.2f us per kernel\n", iters, ms * 1e3, ms * 1e3 / CH);
        }
        hipGraph_t g;
        hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < CH; ++i) hipLaunchKernelGGL(k_work, dim3(64), dim3(64), 0, st, d, iters);
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a, st);
            hipGraphLaunch(ge, st);
            hipEventRecord(b, st);
            hipStreamSynchronize(st);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep == 2) printf("iters %5d graph  : %.1f us total, %.2f us per kernel\n", iters, ms * 1e3, ms * 1e3 / CH);
        }
        hipGraphExecDestroy(ge);
        hipGraphDestroy(g);
    }
    return 0;
}
