// microbenchmark: does hipExtAnyOrderLaunch (AQL barrier bit 0) let a dependent launch's workgroups start while the
// previous launch of the same stream drains, with the dependency carried by a device counter instead?
// P = 246 small workgroups ("k_dw_adam"), C = 256 workgroups with 104 KB of LDS ("k_abc").
//   mode 0: P, C ordinary launches                        (two dispatch boundaries per iteration)
//   mode 1: C any-order behind P, waits on P's counter    (one boundary)
//   mode 2: both any-order, P waits on C's counter too    (none)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned long long u64;

__device__ __forceinline__ u64 now() { return __builtin_amdgcn_s_memrealtime(); }   // 100 MHz

__device__ bool wait_for(unsigned *cnt, unsigned target, unsigned *abort_word) {
    u64 t0 = now();
    for (;;) {
        unsigned v = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)(v - target) >= 0) return true;
        if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
        if (now() - t0 > 200000) { __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        __builtin_amdgcn_s_sleep(2);
    }
}

__global__ __launch_bounds__(256) void k_p(unsigned *cntP, unsigned *cntC, unsigned wait_target, int do_wait, int ticks,
                                           u64 *st, unsigned *abort_word) {
    extern __shared__ float lds[];
    u64 t0 = now();
    if (do_wait && threadIdx.x == 0) wait_for(cntC, wait_target, abort_word);
    __syncthreads();
    u64 t1 = now();
    while (now() - t1 < (u64)ticks) __builtin_amdgcn_s_sleep(1);
    if (threadIdx.x == 1) lds[0] = 1.f;
    __syncthreads();
    u64 t2 = now();
    if (threadIdx.x == 0) {
        st[blockIdx.x * 3 + 0] = t0; st[blockIdx.x * 3 + 1] = t1; st[blockIdx.x * 3 + 2] = t2;
        __hip_atomic_fetch_add(cntP, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(256) void k_c(unsigned *cntP, unsigned *cntC, unsigned wait_target, int do_wait, int ticks,
                                           u64 *st, unsigned *abort_word) {
    extern __shared__ float lds[];
    u64 t0 = now();
    if (do_wait && threadIdx.x == 0) wait_for(cntP, wait_target, abort_word);
    __syncthreads();
    u64 t1 = now();
    while (now() - t1 < (u64)ticks) __builtin_amdgcn_s_sleep(1);
    if (threadIdx.x == 1) lds[0] = 1.f;
    __syncthreads();
    u64 t2 = now();
    if (threadIdx.x == 0) {
        st[blockIdx.x * 3 + 0] = t0; st[blockIdx.x * 3 + 1] = t1; st[blockIdx.x * 3 + 2] = t2;
        __hip_atomic_fetch_add(cntC, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

static void span(const char *name, const std::vector<u64> &s, int nb, u64 base) {
    u64 lo[3] = {~0ull, ~0ull, ~0ull}, hi[3] = {0, 0, 0};
    for (int b = 0; b < nb; ++b) for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], s[b * 3 + k]); hi[k] = std::max(hi[k], s[b * 3 + k]); }
    printf("    %s: start %.2f..%.2f  waited %.2f..%.2f  end %.2f..%.2f us\n", name, (lo[0] - base) * 0.01, (hi[0] - base) * 0.01,
           (lo[1] - base) * 0.01, (hi[1] - base) * 0.01, (lo[2] - base) * 0.01, (hi[2] - base) * 0.01);
}

int main() {
    unsigned *cnt; CK(hipMalloc(&cnt, 4096)); CK(hipMemset(cnt, 0, 4096));
    unsigned *cntP = cnt, *cntC = cnt + 64, *abort_word = cnt + 128;
    u64 *stP, *stC; CK(hipMalloc(&stP, 256 * 3 * 8)); CK(hipMalloc(&stC, 256 * 3 * 8));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipFuncSetAttribute((const void *)k_c, hipFuncAttributeMaxDynamicSharedMemorySize, 104 * 1024));
    CK(hipFuncSetAttribute((const void *)k_p, hipFuncAttributeMaxDynamicSharedMemorySize, 104 * 1024));
    const int NP = 246, NC = 256, N = 2000;
    const int ticksP = 450, ticksC = 1700;     // 4.5 us, 17 us
    hipStream_t s2; int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi));
    hipStream_t s1; CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, lo));
    hipStream_t s3; CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
    printf("stream priorities: least %d greatest %d\n", lo, hi);
    for (int ldsP : {1024, 80 * 1024}) {
        // mode 3: ordinary launches (<<< >>>), one stream; mode 4: P on its own stream, C on another, both wait in-kernel
        for (int mode = 3; mode < 6; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemset(cnt, 0, 4096));
                CK(hipDeviceSynchronize());
                auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < N; ++i) {
                    unsigned tp = (unsigned)(NC * i), tc = (unsigned)(NP * (i + 1));
                    int w = (mode >= 4);
                    // mode 4: two ordinary streams; mode 5: P's stream of higher priority
                    hipStream_t sp_ = mode == 3 ? s : (mode == 4 ? s3 : s2), sc_ = mode == 3 ? s : (mode == 4 ? s : s1);
                    hipLaunchKernelGGL(k_p, dim3(NP), dim3(256), ldsP, sp_, cntP, cntC, tp, w, ticksP, stP, abort_word);
                    hipLaunchKernelGGL(k_c, dim3(NC), dim3(256), 104 * 1024, sc_, cntP, cntC, tc, w, ticksC, stC, abort_word);
                }
                CK(hipDeviceSynchronize());
                double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (rep) {
                    unsigned h[192]; CK(hipMemcpy(h, cnt, sizeof(h), hipMemcpyDeviceToHost));
                    printf("P lds %3d KB, mode %d: %.2f us per iteration, counters %u %u abort %u\n", ldsP / 1024, mode, us / N, h[0], h[64], h[128]);
                    std::vector<u64> hp(256 * 3), hc(256 * 3);
                    CK(hipMemcpy(hp.data(), stP, NP * 24, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(hc.data(), stC, NC * 24, hipMemcpyDeviceToHost));
                    u64 base = ~0ull; for (int b = 0; b < NP; ++b) base = std::min(base, hp[b * 3]);
                    span("P", hp, NP, base); span("C", hc, NC, base);
                }
            }
        }
        for (int mode = 0; mode < 1; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemset(cnt, 0, 4096));
                CK(hipStreamSynchronize(s));
                auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < N; ++i) {
                    unsigned *a0 = cntP, *a1 = cntC, *a6 = abort_word;
                    unsigned tp = (unsigned)(NC * i), tc = (unsigned)(NP * (i + 1));
                    int wp = (mode == 2), wc = (mode >= 1), tkP = ticksP, tkC = ticksC;
                    u64 *sp = stP, *sc = stC;
                    void *argsP[] = {&a0, &a1, &tp, &wp, &tkP, &sp, &a6};
                    void *argsC[] = {&a0, &a1, &tc, &wc, &tkC, &sc, &a6};
                    CK(hipExtLaunchKernel((const void *)k_p, dim3(NP), dim3(256), argsP, ldsP, s, nullptr, nullptr,
                                          mode == 2 && i > 0 ? hipExtAnyOrderLaunch : 0));
                    CK(hipExtLaunchKernel((const void *)k_c, dim3(NC), dim3(256), argsC, 104 * 1024, s, nullptr, nullptr,
                                          mode >= 1 ? hipExtAnyOrderLaunch : 0));
                }
                CK(hipStreamSynchronize(s));
                double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (rep) {
                    unsigned h[192]; CK(hipMemcpy(h, cnt, sizeof(h), hipMemcpyDeviceToHost));
                    printf("P lds %3d KB, mode %d: %.2f us per iteration (P %.1f + C %.1f us of spinning), counters %u %u abort %u\n",
                           ldsP / 1024, mode, us / N, ticksP * 0.01, ticksC * 0.01, h[0], h[64], h[128]);
                    std::vector<u64> hp(256 * 3), hc(256 * 3);
                    CK(hipMemcpy(hp.data(), stP, NP * 24, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(hc.data(), stC, NC * 24, hipMemcpyDeviceToHost));
                    u64 base = ~0ull; for (int b = 0; b < NP; ++b) base = std::min(base, hp[b * 3]);
                    span("P", hp, NP, base); span("C", hc, NC, base);
                }
            }
        }
    }
    return 0;
}
