// Roofline denominators measured on the box (SURVEY.md section 8d "Bounding roofline": stream-copy GB/s and an
// FP32 MFMA microbenchmark, vendor figures only as a cross-check).  Measurement helper of the C ABI; nothing on
// the training path calls it.
#include "sac_common.h"

namespace sac {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// float4 copy (read + write counted): a workgroup moves 16-KB tiles -- four 1-KB wave-instructions per wave in flight,
// each lane 16 B -- dealt round-robin over the grid
__global__ __launch_bounds__(256) void k_peak_copy(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n4) {
    const size_t tiles = n4 / 1024;                              // 1024 float4 = 16 KB
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const f32x4 *s = src + t * 1024 + threadIdx.x;
        f32x4 *o = dst + t * 1024 + threadIdx.x;
        const f32x4 a = s[0], b = s[256], c = s[512], e = s[768];
        o[0] = a; o[256] = b; o[512] = c; o[768] = e;
    }
}

// back-to-back v_mfma_f32_16x16x4_f32 on 8 independent accumulators per wave, operands in registers (non-trivial
// values: the clock the chip holds under load depends on the data), one wave per SIMD
__global__ __launch_bounds__(256) void k_peak_mfma(float *__restrict__ out, int iters, float seed) {
    f32x4 acc[8];
    const float a = seed + 0.001f * (float)(threadIdx.x & 63), b = 1.0f - 0.002f * (float)(threadIdx.x & 31);
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int t = 1; t < 8; ++t) s += acc[t];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

}  // namespace sac

using namespace sac;

extern "C" int sac_measure_peaks(int device, float out[4]) {
    SAC_REQUIRE(out != nullptr, "null out pointer");
    SAC_REQUIRE(sac_device_count() > 0, "no HIP device visible: libsac_hip has no CPU fallback");
    SAC_HIP(hipSetDevice(device));
    hipStream_t s;
    SAC_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    SAC_HIP(hipEventCreate(&e0));
    SAC_HIP(hipEventCreate(&e1));
    int rc = 0;
    float *src = nullptr, *dst = nullptr, *mo = nullptr;
    const size_t bytes = (size_t)1 << 30;                      // 1 GiB each way: four times the 256-MiB Infinity Cache
    do {
        if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&dst, bytes) != hipSuccess ||
            hipMalloc(&mo, sizeof(float) * 1024 * 256) != hipSuccess) {
            sac::set_error("sac_measure_peaks: allocation failed");
            rc = -1;
            break;
        }
        (void)hipMemsetAsync(src, 0x3c, bytes, s);
        (void)hipMemsetAsync(dst, 0, bytes, s);
        float best_copy = 0.f;
        const int grids[3] = {256 * 8, 256 * 16, 256 * 32};
        for (int rep = 0; rep < 7; ++rep) {
            (void)hipEventRecord(e0, s);
            hipLaunchKernelGGL(k_peak_copy, dim3(grids[rep % 3]), dim3(256), 0, s, (const f32x4 *)src, (f32x4 *)dst, bytes / 16);
            (void)hipEventRecord(e1, s);
            if (hipStreamSynchronize(s) != hipSuccess) { sac::set_error("sac_measure_peaks: copy kernel failed"); rc = -1; break; }
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const float gbs = (float)(2.0 * (double)bytes / (ms * 1e-3) / 1e9);
            if (rep > 0 && gbs > best_copy) best_copy = gbs;    // (the first pass also pages the buffers in)
        }
        if (rc) break;
        float best_mfma = 0.f, mfma_ms = 0.f;
        // (one and four waves per SIMD; the chip settles on its clock under load within the first pass)
        for (int rep = 0; rep < 5; ++rep) {
            const int wgs = (rep & 1) ? 256 : 1024, iters = (rep & 1) ? 60000 : 20000;
            (void)hipEventRecord(e0, s);
            hipLaunchKernelGGL(k_peak_mfma, dim3(wgs), dim3(256), 0, s, mo, iters, 0.5f + 0.1f * rep);
            (void)hipEventRecord(e1, s);
            if (hipStreamSynchronize(s) != hipSuccess) { sac::set_error("sac_measure_peaks: mfma kernel failed"); rc = -1; break; }
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)wgs * 4.0 * iters * 8.0 * 2048.0;     // 16*16*4 MACs = 2048 FLOP per wave-MFMA
            const float tf = (float)(flop / (ms * 1e-3) / 1e12);
            if (rep > 0 && tf > best_mfma) { best_mfma = tf; mfma_ms = ms; }
        }
        if (rc) break;
        out[0] = best_copy;          // GB/s, read + write
        out[1] = best_mfma;          // TFLOP/s, fp32 MFMA 16x16x4
        out[2] = (float)(2.0 * (double)bytes / 1e9);
        out[3] = mfma_ms;
    } while (0);
    (void)hipFree(src); (void)hipFree(dst); (void)hipFree(mo);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipStreamDestroy(s);
    return rc;
}
