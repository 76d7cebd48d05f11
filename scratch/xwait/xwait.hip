// What does a cross-stream dependency cost the waiting stream?  Producer stream A: a short kernel per iteration, then a
// signal; consumer stream B: wait, then a ~20 us kernel.  The producer runs far ahead (its kernels are short), so every
// wait finds its condition already true -- the cost measured is the wait packet itself.
//   mode 0: no dependency (B alone)   1: hipEventRecord + hipStreamWaitEvent   2: hipStreamWriteValue32 + hipStreamWaitValue32
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_short(int *p) { if (threadIdx.x == 0) p[blockIdx.x] += 1; }
__global__ void k_long(float *p, int iters) {
    float a = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
    p[threadIdx.x] = a;
}
int main(int argc, char **argv) {
    const int N = 2000, iters = argc > 1 ? atoi(argv[1]) : 9000;
    hipStream_t A, B;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    int *pi; float *pf; unsigned *flag = nullptr;
    CK(hipMalloc(&pi, 4096)); CK(hipMalloc(&pf, 4096));
    CK(hipMemset(pi, 0, 4096)); CK(hipMemset(pf, 0, 4096));
    hipError_t fe = hipExtMallocWithFlags((void **)&flag, 8, hipMallocSignalMemory);
    if (fe != hipSuccess) { printf("signal memory: %s\n", hipGetErrorString(fe)); flag = nullptr; }
    else CK(hipMemset(flag, 0, 8));
    hipEvent_t ev[16], t0, t1;
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    for (int mode = 0; mode < 3; ++mode) {
        if (mode == 2 && !flag) continue;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            if (flag) CK(hipMemset(flag, 0, 8));
            CK(hipDeviceSynchronize());
            auto h0 = std::chrono::steady_clock::now();
            CK(hipEventRecord(t0, B));
            for (int i = 0; i < N; ++i) {
                if (mode) hipLaunchKernelGGL(k_short, dim3(1), dim3(64), 0, A, pi);
                if (mode == 1) { CK(hipEventRecord(ev[i & 15], A)); CK(hipStreamWaitEvent(B, ev[i & 15], 0)); }
                if (mode == 2) {
                    CK(hipStreamWriteValue32(A, flag, (unsigned)(i + 1), 0));
                    CK(hipStreamWaitValue32(B, flag, (unsigned)(i + 1), hipStreamWaitValueGte, 0xffffffffu));
                }
                hipLaunchKernelGGL(k_long, dim3(256), dim3(256), 0, B, pf, iters);
            }
            CK(hipEventRecord(t1, B));
            auto h1 = std::chrono::steady_clock::now();
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, t0, t1));
            if (rep) printf("mode %d: %.2f us per iteration on the consumer stream (host issue %.2f us)\n", mode, ms * 1e3 / N,
                            std::chrono::duration<double, std::micro>(h1 - h0).count() / N);
        }
    }
    return 0;
}
