// microbenchmark: dispatch boundary between dependent kernels, stream launches vs hipGraph
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_busy(float *p, int iters) {
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    if (v == 12345.f) p[threadIdx.x] = v;
}
int main() {
    float *d; CK(hipMalloc(&d, 4096));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int N = 4000;
    for (int iters : {0, 100, 200, 400, 800}) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipStreamSynchronize(s));
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_busy, dim3(256), dim3(256), 0, s, d, iters);
            CK(hipStreamSynchronize(s));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rep) printf("stream launches, iters %d: %.2f us per kernel\n", iters, us / N);
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 400; ++i) hipLaunchKernelGGL(k_busy, dim3(256), dim3(256), 0, s, d, iters);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipStreamSynchronize(s));
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N / 400; ++i) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rep) printf("graph of 400,    iters %d: %.2f us per kernel\n", iters, us / N);
        }
    }
    return 0;
}
