// SAC (and TD3) gradient step on MI355X (gfx950): four row-block / tile-owner kernels per step.
//
// Replaces rlkit SACTrainer.train_from_torch (+ np_to_pytorch_batch, soft_update_from_to); call
// sites /root/reference/util/rlkit_utils.py:64-106, /root/reference/util/rlkit_custom.py:238.
// Normative step order: SURVEY.md Appendix A (lines 1-18), ordering O1, logging quirk Q1.
// TD3 (rlkit TD3Trainer, /root/reference/util/rlkit_utils.py:107-135) runs on the same kernels as compile-time
// variants (MODE = M_TD3_CRITIC / M_TD3_ACTOR, see launch_step_td3).
//
// Decomposition (DESIGN.md "Kernels"): batch rows are independent through forward and backward-dX,
// so a workgroup owns a 16-row block (one MFMA 16x16x4 M-tile) and chains whole layers through LDS;
// the only all-to-all seams are (a) mean(log_pi) -> alpha, (b) min over twin nets, (c) the weight
// gradient's contraction over the batch.  Each seam is a kernel boundary (cheaper than an in-launch
// grid barrier on 8 XCDs); the weight-gradient kernel is tile-owner parallel and applies Adam and
// the Polyak update in its epilogue, so gradients never round-trip through HBM.
//
//   A k_fwd_a       16*B/16 WGs   pi(s), pi(s') -> head partials; Q1,Q2(s,a) -> q partials (+ first-layer pre-activations z)
//   B k_fwd_b       16*B/16 WGs   tanh-Gaussian head; Q1,Q2(s,a_new) (first layer = z + W_a (a_new - a)), T1,T2(s',a')
//                                 -> q partials; the Q(s,a_new) blocks continue to the UNIT gradient dQ/da (partials)
//   C k_bwd         12*B/16 WGs   critic dL/dh (2 nets) | policy: min-select, head gradient (reparameterised), dL/dh
//   D k_dw_adam     ~250  WGs     dW = dY^T X over the batch (MFMA), Adam, Polyak, diagnostics
// (every 256-wide layer is split over 4 workgroups per 16-row block; partial sums meet at launch boundaries)
//
// Latency rules every kernel follows (a step is ~0.6 GFLOP: it is bound by dependent memory round trips and by
// the ~20 B/clk a CU's fill path sustains, not by FLOPs) -- DESIGN.md section 4 has the list with the measurements:
// launch arguments instead of device counters, kernel arguments prefetched into the scalar cache, requests in
// consumption order and paced between independent work, register-ring weight stream, LDS-only barriers,
// unconditional loads, global stores behind the last load request, compile-time variants inside GEMM loops.
#include "sac_common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <mutex>
#include <vector>

namespace sac {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int H = 256;            // hidden width (every shipped variant.json)
constexpr float LOG_SIG_MAX = 2.0f, LOG_SIG_MIN = -20.0f, TANH_EPS = 1e-6f;
constexpr float ADAM_B1 = 0.9f, ADAM_B2 = 0.999f, ADAM_EPS = 1e-8f;
// torch computes 1 - beta in double and applies it as an fp32 scalar: (float)(1.0 - 0.9), (float)(1.0 - 0.999)
constexpr float ADAM_1MB1 = (float)(1.0 - 0.9), ADAM_1MB2 = (float)(1.0 - 0.999);
constexpr int DIAG_TRACE_CAP = 4096;
// kernel variants: the SAC step, the TD3 critic pass (target policy, T1/T2, critic backward) and the TD3 actor pass
constexpr int M_SAC = 0, M_TD3_CRITIC = 1, M_TD3_ACTOR = 2;
constexpr int RD = 4;             // ring depth (k-chunks in flight) for the runtime-K first layers

// Step counters are HOST state passed as launch arguments (a device-side counter would put a
// dependent, wave-uniform load -- readfirstlane + vmcnt(0) -- in front of every kernel's first
// data access); only what the device itself produces lives here.
struct StepArg {
    long long step_now;            // rlkit _n_train_steps_total of this step (pre-increment)
    long long adam_t;              // optimizer step count of this step (1-based)
    int loop_pos;                  // index of this step inside the current sac_train_loop
    int pad;
    double bc1, bc2s;              // 1 - beta1^t and sqrt(1 - beta2^t), computed on the host in double like torch
    unsigned seq, pad2;            // fused step: number of this launch (1, 2, ...), the hand-off counters count in units of it
};

struct Ctl {                       // device-resident state of the entropy coefficient (Adam on log_alpha)
    float log_alpha, a_m, a_v, alpha, alpha_loss;
    int pad[3];
};

struct AlphaStep { float alpha, alpha_loss, log_alpha, m, v; };

struct Layer {                     // one nn.Linear in the padded device layout
    int N, K, Np, Kp;
    long long offW, offB;          // in P / M / V / G : W [Np][Kp] fragment-major (frag_off), b [Np]
    long long offWt;               // in PT / MT / VT : W^T [Kp + 16][Np] fragment-major (frag_off)
};

struct Net {
    float *P = nullptr, *M = nullptr, *V = nullptr, *PT = nullptr, *MT = nullptr, *VT = nullptr, *G = nullptr;
    long long nP = 0, nPT = 0;
    Layer L[3];
};

// everything the step kernels need, passed by value
struct Dev {
    int B, Bt;                     // batch rows padded to a multiple of 16 / rows that count (the rest has zero weight)
    int O, A, KP, KQ, NH, NB;      // KP = pad16(O), KQ = KP + 16 (Q input: [obs | pad | act | pad]), NH = pad16(2A), NB = B/16
    float discount, reward_scale, tau, target_entropy, alpha_lr;
    int period, auto_alpha;
    unsigned long long noise_seed;
    Ctl *ctl;
    // nets: 0 policy, 1 qf1, 2 qf2, 3 tqf1, 4 tqf2, 5 target policy (TD3)
    const float *P[6];
    int algo, sp;                  // 0 SAC, 1 TD3; column split of the step kernels (layout of qpart)
    float td3_sigma, td3_clip;     // target-policy smoothing noise: clamp(N(0,1) * sigma, +-clip)
    const float *PT[3];
    Layer LP[3], LQ[3];
    // policy activations (s rows) feature-major [256][B]; per-row head values row-major [B][16]
    float *PH1T, *PH2T, *mu, *ls, *lsok, *z, *anew, *epsv, *logpi, *a2, *logpi2, *part_logpi;
    // Q forward: passes 0..3 keep h1/h2 feature-major; q values for all 6 passes
    float *QH1T, *QH2T, *q;
    float *QU;                     // Q1,Q2(s,a) first-layer pre-activations, feature-major [2][256][B]
    // backward
    float *y, *dq16T, *dQH2T, *dQH1T, *dheadT, *dPH2T, *dPH1T;
    // partial sums of the column-split layers (added by the consuming launch, fixed order)
    float *headpart, *qpart, *dapart;
    // fused step (k_abc): per-producer 128-B lines of q partials [pass][row-block][part][32] and of log pi(a'|s')
    // [row-block][32], hand-off counters (one per 128-B line: head[side][rb], phase-B-done[rb], log-pi partials) and
    // the sticky abort word (a wait timed out)
    float *qpart2, *logpi2p;
    float *hv4;                    // fused step: {a_new, log_std, eps, clamp mask} of a (row, action) as ONE 16-byte value [B][16][4]
    // fused step, wide first layers: the four column parts of a row-block compute a quarter of the first layer each and
    // exchange the pre-activations: zx[chain][rb][part][wave] = one 16 x 16 tile (1 KB, MFMA C layout); chains 0,1 =
    // Q1,Q2(s,a), 2,3 = pi(s), pi(s'), 4,5 = T1,T2(s',a')
    float *zx;
    unsigned *cnt, *abort_flag;
    // diagnostics
    float *diag_first, *diag_last, *diag_trace;      // first / last: mapped pinned host memory; trace: device
    float *diag_dev;                                  // device scratch for the steps whose diagnostics nobody reads
    // caller-supplied noise (NULL => counter-based device stream)
    const float *eps1, *eps2;
};

// weight-gradient work table (kernel argument of K5): one entry per trained layer
struct DwLayer {
    const float *dYT, *XT;         // dY^T [Np][B]; X^T [>= 64*nk rows][B] (or offset into the slot)
    long long xt_off;
    float *P, *PT, *MT, *VT, *G;   // forward copy [Np][Kp], transposed copy + Adam moments [Kp+16][Np]
    float *bias, *mb, *vb, *gb;    // bias, its moments, its gradient (debug)
    float *TP, *Tbias;             // Polyak target (forward copy / bias) or null
    int N, K, ldp, ldt, nk, job0, xt_from_slot;
    float lr;
    int k_base, pad_;              // first column of this entry's strips (a first layer's partial last strip is an entry of its own)
};
constexpr int NDW = 12;           // 3 nets x 3 layers + the tail strips of the first layers (see build_table)
struct DwTable {                   // L: device array, read with scalar loads
    const DwLayer *L;
    int job0[NDW];
    int njobs;
    const unsigned *abort;         // fused step only: nonzero => the forward/backward launch gave up, apply NOTHING
};

// ------------------------------------------------------------------------------------------
// in-kernel stamps (diagnostic build only, -DSAC_STAMPS; the shipped library has none)
// ------------------------------------------------------------------------------------------
#ifdef SAC_STAMPS
__device__ unsigned long long g_stamps[5 * 512 * 16];
#define STAMP(kid, i)                                                                              \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (threadIdx.x == 0 && blockIdx.x < 512) {                                                \
            g_stamps[((kid) * 512 + blockIdx.x) * 16 + (i)] = wall_clock64();                      \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#define STAMP_CLK(kid, i)                                                                          \
    do {                                                                                           \
        if (threadIdx.x == 0 && blockIdx.x < 512) g_stamps[((kid) * 512 + blockIdx.x) * 16 + (i)] = clock64(); \
    } while (0)
#else
#define STAMP(kid, i) do { } while (0)
#define STAMP_CLK(kid, i) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int lds_off(int row, int k, int KL) {
    return row * KL + ((((k >> 2) ^ (row & 15)) << 2) | (k & 3));
}
// hipcc sinks prefetch loads down to their first use (and hoists bulk loads above small critical
// ones); vmcnt completes in issue order, so the ISSUE ORDER is part of the design.  SB pins it.
#define SB() __builtin_amdgcn_sched_barrier(0)
// IR-level code motion hoists pure arithmetic (expf, fminf ...) up to right behind the load that
// feeds it -- across sched_barrier -- which puts an s_waitcnt in the middle of the prologue's load
// burst.  USE_FROM_HERE(x) makes x opaque at this point, so nothing computed from it can move above.
#define USE_FROM_HERE(x) asm volatile("" : "+v"(x))
// Workgroup barrier for LDS hand-offs only.  __syncthreads() also carries a workgroup-scope fence
// for global memory, which hipcc lowers to s_waitcnt vmcnt(0): vmcnt counts loads AND stores in
// issue order, so every barrier would drain the weight stream and the prefetches in flight (a full
// L2 / Infinity-Cache round trip per barrier).  The step kernels exchange data between waves only
// through LDS (and through explicit agent-scope release/acquire where global memory is involved),
// so their barriers wait for this wave's LDS operations only.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Wave-uniform device scalars (alpha, Adam bias corrections): read through the scalar cache
// (s_load, waited for at first use).  A plain load of a uniform address becomes a vector load +
// v_readfirstlane with an immediate vmcnt(0), i.e. a serial round trip in front of the prologue's
// loads.  The scalar cache is invalidated at kernel start, and these words are only ever written
// by an EARLIER kernel of the stream.
template <typename T>
__device__ __forceinline__ T sload(const T *p) {
    return *(const __attribute__((address_space(4))) T *)(uintptr_t)p;
}
__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
// Accesses through an explicitly GLOBAL pointer (a pointer that reaches a kernel inside a table read with scalar loads,
// DwLayer, has no address space the compiler can infer: its accesses become flat_load / flat_store).  Measured, not
// assumed: k_dw_adam's OPERAND loads are left generic on purpose -- with global loads the same launch is 0.7 us longer
// by rocprofv3 on the same box (7.45 against 6.7 us, scratch/ab_libs.sh), although its in-kernel stamps are not.
__device__ __forceinline__ float ld1g(const float *p) { return *(const __attribute__((address_space(1))) float *)(uintptr_t)p; }
// (write-through: a plain global store would sit dirty in the L2 until the end-of-kernel write-back)
__device__ __forceinline__ void st1g(float *p, float v) {
    __hip_atomic_store((__attribute__((address_space(1))) float *)(uintptr_t)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Kernel arguments are ~0.7 KB (11 scalar-cache lines) and hipcc loads them lazily -- an s_load right
// before each first use, each followed by s_waitcnt lgkmcnt(0): a chain of serial misses (the scalar cache is
// invalidated at every dispatch) worth 2-3 us in front of the prologue's loads.  Touching one dword per 64-B
// line at entry turns that into ONE round trip; the lazy loads then hit the scalar cache.
template <int NBYTES>
__device__ __forceinline__ void kernarg_prefetch() {
    typedef const __attribute__((address_space(4))) int *kptr;
    kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int NL = (NBYTES + 63) / 64;
    static_assert(NL <= 16, "kernarg_prefetch: more than 16 lines");
    int x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = ka[i < NL ? i * 16 : 0];
    asm volatile("" ::"s"(x[0]), "s"(x[1]), "s"(x[2]), "s"(x[3]), "s"(x[4]), "s"(x[5]), "s"(x[6]), "s"(x[7]), "s"(x[8]),
                 "s"(x[9]), "s"(x[10]), "s"(x[11]), "s"(x[12]), "s"(x[13]), "s"(x[14]), "s"(x[15]));
}

// SURVEY Appendix A lines 4-6: alpha_loss = -mean(log_alpha * (log_pi + H)); one Adam step on log_alpha;
// alpha = exp(log_alpha) (post-step).  sum(log_pi) arrives as NB per-row-block partials (written by the
// previous launch); every workgroup that needs alpha recomputes this handful of scalar operations from the
// same inputs in the same order -- bit-identical everywhere, no reduction kernel, no in-launch hand-off.
// The new state is stored once, by the diagnostics block of the last launch of the step.
__device__ __forceinline__ AlphaStep alpha_step(const Ctl *ctl, const float *part_logpi, int NB, int B, float target_entropy,
                                                float lr, int auto_alpha, double bc1, double bc2s) {
    AlphaStep r;
    const float la = sload(&ctl->log_alpha), m0 = sload(&ctl->a_m), v0 = sload(&ctl->a_v);
    if (!auto_alpha) { r.alpha = 1.0f; r.alpha_loss = 0.0f; r.log_alpha = la; r.m = m0; r.v = v0; return r; }
    float sum = 0.f;
    for (int i = 0; i < NB; ++i) sum += sload(part_logpi + i);
    const float mean_lp = sum / (float)B + target_entropy;               // mean(log_pi + H)
    // torch: -(log_alpha * x).mean(); the mean's running sum starts at +0 => log_alpha == 0 logs -0.0
    r.alpha_loss = -((la * mean_lp) + 0.0f);
    const float gr = -mean_lp;                                           // d alpha_loss / d log_alpha
    r.m = m0 + ADAM_1MB1 * (gr - m0);
    r.v = v0 * ADAM_B2 + ADAM_1MB2 * gr * gr;
    const float step_size = (float)((double)lr / bc1);
    const float denom = sqrtf(r.v) / (float)bc2s + 1e-8f;
    r.log_alpha = la + (-step_size * r.m) / denom;
    r.alpha = expf(r.log_alpha);
    return r;
}
__device__ __forceinline__ void st4(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }

// Weight stream of one wave: NT column tiles of 16 outputs, k-chunks of 16 held in a D-deep
// register ring.  W is [n][ldw] in fragment-major order (frag_off); lane (c = lane&15, g = lane>>4) loads the 16 B
// W[n0_t + c][16 S + 4 g .. +3], i.e. lane group g owns contraction index k = 16S + 4g + i.
template <int NT, int D = RD>
struct WRing {
    f32x4 b[D][NT];
    const float *wp[NT];
    __device__ __forceinline__ void init(const float *W, int ldw, int n_base, int n_stride, int s_off = 0) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int t = 0; t < NT; ++t)
            wp[t] = W + ((size_t)((n_base + t * n_stride) >> 4) * (ldw >> 4) + s_off) * 256 + 4 * lane;   // frag_off of this lane's 16 B
    }
    // chunks [u0, u1) of the first min(D, KS): the prologues issue a ring in pieces between independent
    // work -- a wave whose loads outrun the CU's fill path (~20 B/clk) just stalls at issue
    __device__ __forceinline__ void fill_part(int KS, int u0, int u1, int lo = 0) {
#pragma unroll
        for (int u = 0; u < D; ++u)
            if (u >= u0 && u < u1 && u < KS && u >= lo) {
#pragma unroll
                for (int t = 0; t < NT; ++t) b[u][t] = ld4(wp[t] + 256 * u);
            }
    }
    __device__ __forceinline__ void fill(int KS) {        // chunks 0 .. min(D, KS)-1 into flight
#pragma unroll
        for (int u = 0; u < D; ++u)
            if (u < KS) {
#pragma unroll
                for (int t = 0; t < NT; ++t) b[u][t] = ld4(wp[t] + 256 * u);
            }
    }
};

// acc[t] += X[16 x 16*KS] * W_t^T.  X: LDS row-block, row stride KL (multiple of 64), 16-B chunks
// XOR-swizzled by row (conflict-free ds_read_b128 for the MFMA A operand).  The ring must have been
// fill()ed; chunk S+RD is requested as soon as chunk S has been copied out.
// `stage` != null: every weight fragment is also copied to LDS as it leaves the ring, TRANSPOSED: stage[k][row of W]
// with row stride WLD (this wave's tile t owns columns 16 t ..), for a later contraction over the rows of W
// (gemm_lds_rows) -- the same bytes a backward pass would otherwise fetch again from the transposed copy.
constexpr int WLD = 64 + 4;       // 64 staged weight rows; +4: scatter writes and 16-B reads are bank-conflict free
// ALT = false: a one-tile GEMM accumulates in ONE register like a tile of a four-tile GEMM does (bit-identical to it).
template <bool STAGED = false, bool ALT = true, int NT, int D>
__device__ __forceinline__ void gemm_ring(WRing<NT, D> &R, const float *X, int KL, int KS, f32x4 (&acc)[NT],
                                          int s_off = 0, float *stage = nullptr) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    const float *xrow = X + r * KL;
    f32x4 a_cur = ld4(xrow + 4 * ((4 * s_off + g) ^ r));
    f32x4 acc_odd = {0.f, 0.f, 0.f, 0.f};
    for (int S0 = 0; S0 < KS; S0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int S = S0 + u;
            if (S < KS) {
                // next chunk's A fragment (LDS) is requested in front of this chunk's MFMAs; the MFMAs read the ring slot
                // in place (an MFMA takes its operands at issue) and the slot's refill (global) is requested right
                // behind them -- no copy-out of the slot, whose v_movs would serialise with the MFMAs of a wave that
                // has the SIMD to itself
                const int Sn = (S + 1 < KS) ? S + 1 : S;
                const f32x4 a_nxt = ld4(xrow + 4 * ((4 * (Sn + s_off) + g) ^ r));
                if constexpr (STAGED) {     // (compile-time: a run-time test here made hipcc mis-order the MFMAs' waits)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int i = 0; i < 4; ++i) stage[(16 * S + 4 * g + i) * WLD + 16 * t + r] = R.b[u][t][i];
                }
                SB();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (NT == 1 && ALT) {
                        // one tile per wave = one dependency chain: consecutive MFMAs alternate between two accumulators
                        // (a dependent v_mfma_f32_16x16x4_f32 issues after 40 cycles, an independent one after 32)
                        if (i & 1) acc_odd = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[i], R.b[u][0][i], acc_odd, 0, 0, 0);
                        else acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[i], R.b[u][0][i], acc[0], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[i], R.b[u][t][i], acc[t], 0, 0, 0);
                    }
                }
                SB();
                if (S + D < KS) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) R.b[u][t] = ld4(R.wp[t] + 256 * (S + D));
                }
                SB();
                a_cur = a_nxt;
            }
        }
    }
    if constexpr (NT == 1 && ALT) acc[0] += acc_odd;
}

// acc[t] += X[16 x 16*KS] * W[16*KS rows][16 columns at col0 + 16 t] from the slice staged by gemm_ring
// (WL[column][row]): contraction over the ROWS of W, lane group g owns rows 16S + 4g + i like everywhere else,
// so a lane's four B values are one 16-B LDS read.  All KS*NT fragments are read before the first MFMA.
template <int NT, int KS>
__device__ __forceinline__ void gemm_lds_rows(const float *WL, int col0, const float *X, int KL, f32x4 (&acc)[NT]) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    const float *xrow = X + r * KL;
    f32x4 a[KS], bf[KS][NT];
#pragma unroll
    for (int S = 0; S < KS; ++S) {
        a[S] = ld4(xrow + 4 * ((4 * S + g) ^ r));
#pragma unroll
        for (int t = 0; t < NT; ++t) bf[S][t] = ld4(WL + (col0 + 16 * t + r) * WLD + 16 * S + 4 * g);
    }
#pragma unroll
    for (int S = 0; S < KS; ++S)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[S][i], bf[S][t][i], acc[t], 0, 0, 0);
}

// First layers (K = obs or obs+act, a few k-chunks): every chunk was requested by fill(), nothing is
// refilled, and the loop is straight-line code -- hipcc's s_waitcnt insertion counts loads exactly
// only when no loop back-edge separates a load from its use (with a runtime trip count it falls
// back to vmcnt(0), which would drain the NEXT layer's 64 KB prefetch before this GEMM starts).
template <int NT, int D>
__device__ __forceinline__ void gemm_straight(WRing<NT, D> &R, const float *X, int KL, int KS, f32x4 (&acc)[NT],
                                              int lo = 0) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    const float *xrow = X + r * KL;
#pragma unroll
    for (int u = 0; u < D; ++u) {
        if (u >= lo && u < KS) {
            const f32x4 a = ld4(xrow + 4 * ((4 * u + g) ^ r));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], R.b[u][t][i], acc[t], 0, 0, 0);
            }
        }
    }
}
// Same, with the NEXT layer's ring requested in four pieces behind the first four chunks' MFMAs: issued at entry
// those loads would only keep the wave stalled at issue (the CU's fill path is the bottleneck) in front of this GEMM.
template <int NT, int D, int NT1, int D1>
__device__ __forceinline__ void gemm_straight_pf(WRing<NT, D> &R, const float *X, int KL, int KS, f32x4 (&acc)[NT],
                                                 WRing<NT1, D1> &R1, int KS1, int pre = D) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    const float *xrow = X + r * KL;
    constexpr int Q1 = (D1 + 3) / 4;
    // chunks [0, pre) of this layer were requested before the call; chunk u + pre is requested in front of chunk
    // u's MFMAs (which run in the background while the wave waits at the next request), then the next layer's ring
#pragma unroll
    for (int u = 0; u < D; ++u) {
        if (u + pre < D) {
            SB();
            R.fill_part(KS, u + pre, u + pre + 1);
            SB();
        }
        if (u < KS) {
            const f32x4 a = ld4(xrow + 4 * ((4 * u + g) ^ r));
#ifdef SAC_STAMPS
            if (u < 4) { float probe = R.b[u][NT - 1][3]; asm volatile("" ::"v"(probe)); STAMP(0, 5 + u); }   // chunk u has arrived
#endif
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], R.b[u][t][i], acc[t], 0, 0, 0);
            }
        }
        if (u < 4) {
            SB();
            R1.fill_part(KS1, u * Q1, (u + 1) * Q1);
            SB();
        }
    }
}
constexpr int RD0 = 8;            // narrow first layers: up to 8 k-chunks (K <= 128) held at once

// split-K epilogue: the four waves each hold a partial [16 x 16*NTT]; sum them through LDS into
// `out` (row-major [16][ldo]) + bias.  red = 4*NTT*256 floats.
// Stores / loads of data that another workgroup of the SAME launch consumes (fused step, sac_fused.h): agent-scope
// write-through stores (global_store ... sc1) and L1-bypassing 4-byte loads.
__device__ __forceinline__ void st_sc1(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// (inline asm: the compiler's hazard recogniser does not see a VMEM store in it, and a store of more than 8 bytes
//  followed by a VALU write of its data registers needs 2 wait states on gfx94x/gfx950 -- hence the s_nop)
__device__ __forceinline__ void st4_sc1(float *p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

template <int NTT, bool SC1 = false>
__device__ __forceinline__ void splitk_reduce(const f32x4 (&acc)[NTT], const float *__restrict__ bias, float *red,
                                              float *out, int ldo) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int t = 0; t < NTT; ++t) st4(red + ((wave * NTT + t) * 64 + lane) * 4, acc[t]);
    lds_barrier();
    if constexpr (SC1) {
        // hand-off form (no bias): a thread sums four CONSECUTIVE columns of one row and publishes them with one 16-byte
        // write-through store -- per byte a dword sc1 store costs ~6x a dwordx4 one, and a consumer block is waiting for these
        for (int e = threadIdx.x; e < NTT * 64; e += 256) {
            const int t = e >> 6, row = (e >> 2) & 15, c0 = 4 * (e & 3);
            f32x4 s4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int l = (row >> 2) * 16 + c0 + j, i = row & 3;
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) s += red[((w * NTT + t) * 64 + l) * 4 + i];
                s4[j] = s;
            }
            st4_sc1(out + row * ldo + 16 * t + c0, s4);
        }
    } else {
        for (int e = threadIdx.x; e < NTT * 256; e += 256) {
            const int t = e >> 8, l = (e >> 2) & 63, i = e & 3;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) s += red[((w * NTT + t) * 64 + l) * 4 + i];
            const int row = 4 * (l >> 4) + i, col = 16 * t + (l & 15);
            out[row * ldo + col] = s + (bias ? bias[col] : 0.f);
        }
    }
    lds_barrier();
}

__device__ __forceinline__ float group16_sum(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}

// Philox4x32-10 -> one N(0,1) (Box-Muller); counter = (step, element), key = seed
__device__ __forceinline__ float philox_normal(unsigned long long seed, unsigned long long step, unsigned idx,
                                               unsigned stream) {
    unsigned c0 = idx, c1 = stream, c2 = (unsigned)step, c3 = (unsigned)(step >> 32);
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const float u1 = ((float)(c0 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

// Scalar expressions of the step that contain TWO candidate multiplies for one fused multiply-add.  hipcc (fp-contract
// =fast) picks one of them per call site, and which one depends on unrelated code around it -- the fused kernel and
// the four-launch kernels then stop being bit-identical (seen in round 2: turning three stores of the fused kernel into
// write-through stores changed the last bit of the policy's head gradient).  Spelled out here, used by both.
__device__ __forceinline__ float bellman_target(float reward_scale, float r, float term, float discount, float tq) {
    return fmaf(reward_scale, r, __fmul_rn(__fmul_rn(1.0f - term, discount), tq));
}
__device__ __forceinline__ float actor_da(float da1, float dq1, float da2, float dq2) { return fmaf(da1, dq1, __fmul_rn(da2, dq2)); }
// d/dz of (alpha log_pi - min Q) through a = tanh(z): da (1 - a^2) + (alpha/B) 2 a (1 - a^2) / (1 - a^2 + eps)
__device__ __forceinline__ float actor_dz(float da, float om, float alpha_invB, float act) {
    return fmaf(da, om, __fmul_rn(alpha_invB, __fdiv_rn(__fmul_rn(__fmul_rn(2.0f, act), om), __fadd_rn(om, TANH_EPS))));
}
__device__ __forceinline__ float actor_dls(float dz, float stdv, float eps, float alpha_invB, float ok) {
    return __fmul_rn(fmaf(__fmul_rn(dz, stdv), eps, -alpha_invB), ok);
}

// swizzled LDS row-block [16][KL] from row-major global rows (two sources concatenated), in two
// phases: issue() puts the loads in flight early, commit() writes LDS once they are needed.
// Thread (row = tid / 16, p = tid % 16) owns columns p, p + 16, ... of its row: no division, and every
// load is unconditional (clamped address + select) -- a conditional load is a branch whose merge waits.
template <int ROWS_MAXE>                      // Kfill <= 16 * ROWS_MAXE
struct RowRegs {
    float v[ROWS_MAXE];
    // columns [0, n0) from s0, [c1, c1 + n1) from s1, zero elsewhere
    __device__ __forceinline__ void issue(int Kfill, const float *__restrict__ s0, int n0, int ld0,
                                          const float *__restrict__ s1, int n1, int ld1, int c1) {
        const int nper = Kfill >> 4;          // RB * Kfill / 256
        const int row = threadIdx.x >> 4, p = threadIdx.x & 15;
        const float *r0 = s0 + row * ld0;
        const float *r1 = (n1 > 0 ? s1 + row * ld1 : r0) - (n1 > 0 ? c1 : 0);
#pragma unroll
        for (int i = 0; i < ROWS_MAXE; ++i) {
            v[i] = 0.f;
            if (i < nper) {
                const int k = p + 16 * i;
                const float *src = (k < n0) ? r0 + k : ((k >= c1 && k < c1 + n1) ? r1 + k : r0);
                v[i] = *src;
            }
        }
    }
    // columns [skip_lo, skip_hi) are left to another writer (the policy head's action)
    __device__ __forceinline__ void commit(float *X, int KL, int Kfill, int n0, int c1, int n1, int skip_lo = 0,
                                           int skip_hi = 0) const {
        const int nper = Kfill >> 4;
        const int row = threadIdx.x >> 4, p = threadIdx.x & 15;
#pragma unroll
        for (int i = 0; i < ROWS_MAXE; ++i)
            if (i < nper) {
                const int k = p + 16 * i;
                const bool valid = (k < n0) || (k >= c1 && k < c1 + n1);
                if (k < skip_lo || k >= skip_hi) X[lds_off(row, k, KL)] = valid ? v[i] : 0.f;
            }
    }
};

// epilogue of a hidden layer: bias + relu from the accumulators into the next LDS row-block; the values stay in
// `keep` for store_features().  The feature-major global copy [n][B] (for the weight-gradient kernel) is NOT stored
// here: vmcnt retires loads and stores in issue order, so a store issued in front of the next GEMM's weight
// requests would put its ~1.5 us write latency into every later s_waitcnt -- it is issued once the kernel has
// requested all its loads.
template <int NT>
__device__ __forceinline__ void hidden_epilogue(const f32x4 (&acc)[NT], int n_base, int n_stride,
                                                const float (&bv)[NT], float *Xn, int KL, f32x4 (&keep)[NT]) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n_base + t * n_stride + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            keep[t][i] = fmaxf(acc[t][i] + bv[t], 0.f);
            Xn[lds_off(4 * g + i, n, KL)] = keep[t][i];
        }
    }
}
template <int NT, bool SC1 = false>
__device__ __forceinline__ void store_features(const f32x4 (&v)[NT], int n_base, int n_stride, float *outT, int B, int row0) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float *p = outT + frag_off(n_base + t * n_stride + c, row0 + 4 * g, B);
        if constexpr (SC1) st4_sc1(p, v[t]);
        else st4(p, v[t]);
    }
}

// ------------------------------------------------------------------------------------------
// Step kernels.  Every 256-wide hidden layer is split over SPLIT = 4 workgroups per 16-row block
// (64 output columns = one MFMA tile per wave): a CU's fill path sustains only ~20 B/clk, so the
// 256 KB of a 256x256 layer cost one CU 5.7 us but four CUs 1.4 us each.  No CU ever waits for
// another one inside a launch: the small first layers are recomputed by all four workgroups
// (their input is the whole row), and what follows a split layer is always a contraction over its
// columns (head, q value, d/da), so each workgroup emits a PARTIAL sum over its 64 columns and the
// consumer -- always the next launch -- adds the four partials in a fixed order.
//
//   A k_fwd_a   16*B/16 WGs  pi(s), pi(s') -> head partials;   Q1,Q2(s,a) -> q partials
//   B k_fwd_b   16*B/16 WGs  tanh-Gaussian head (from the partials); Q1,Q2(s,a_new), T1,T2(s',a') -> q partials;
//                            the Q(s,a_new) blocks go on to the unit input gradient dQ_i/da (partials): the chain
//                            is linear in dq, so the min over the twins can be applied by the consumer
//   C k_bwd     12*B/16 WGs  q = sum of partials; critic dL/dh (kept for dW) | policy head gradient
//                            (reparameterised, analytic) from sum_i dq_i * dQ_i/da, dL/dh
//   D k_dw_adam              (below)
// ------------------------------------------------------------------------------------------
// SP = column split of the 256-wide layers (4, 2 or 1 workgroups per row-block), NTW = 4 / SP tiles per wave.
// Small batches want SP = 4 (more CUs, 64 KB of weights each); B >= 512 already fills the chip with
// row-blocks, so it takes SP = 2 / 1 (fewer, fatter workgroups; less recomputation of the first layers).
constexpr int ring_depth(int ntw) { return ntw == 1 ? 8 : 4; }   // k-chunks in flight: 8 KB (1 tile) .. 16 KB (4 tiles) per wave

// relu(acc + bias) of this wave's NTW tiles -> local slice buffer XS[16][64*NTW] and (optionally) the
// feature-major global rows of those features
template <int NTW, bool SC1 = false>
__device__ __forceinline__ void slice_epilogue(const f32x4 (&acc)[NTW], const float (&bv)[NTW], int wave, float *XS,
                                               float *outT, int n_first, int B, int row0) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = fmaxf(acc[t][i] + bv[t], 0.f);
            XS[lds_off(4 * g + i, 16 * (NTW * wave + t) + c, 64 * NTW)] = v[i];
        }
        if (outT) {
            float *p = outT + frag_off(n_first + 16 * t + c, row0 + 4 * g, B);
            if constexpr (SC1) st4_sc1(p, v);
            else st4(p, v);
        }
    }
}

// MODE M_TD3_CRITIC: the policy blocks run the TARGET policy on s' (sq == 1) and, only when `aux` is set (policy
// steps), the online policy on s (sq == 0); the Q blocks are the same.
template <int NTH, bool WIDE, int SP, int MODE = M_SAC>
__global__ __launch_bounds__(256) void k_fwd_a(Dev d, const float *__restrict__ S, SlotLayout SL, int aux) {
    kernarg_prefetch<sizeof(Dev) + 8 + sizeof(SlotLayout) + 4>();
    constexpr int NTW = 4 / SP, SW = 64 * NTW;               // tiles per wave, slice width
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int B = d.B, O = d.O, A = d.A, NB = d.NB;
    const int KLmax = (d.KQ + 63) & ~63;
    float *X0 = lds;                     // [16][KL0]
    float *X1 = X0 + RB * KLmax;         // [16][256]  full first hidden layer
    float *XS = X1 + RB * H;             // [16][SW]   this block's columns of the second one
    float *red = XS + RB * SW;           // split-K scratch 4*NTH*256
    // XCD-aware block -> work map (speed only): workgroups are dealt round-robin over the 8 XCDs, and each
    // XCD's L2 has to pull every weight matrix its workgroups touch from the Infinity Cache once per launch
    // (weights change every step).  Blocks with b % 8 in {0..3} take the policy, {4,5} Q1, {6,7} Q2, so an
    // XCD fetches ONE network instead of all three.
    const int xq = blockIdx.x >> 3, xr = blockIdx.x & 7;
    const bool is_pi = xr < 4;
    const int b = is_pi ? 4 * xq + xr : 2 * xq + (xr & 1);          // index inside the network's blocks
    const int part = b % SP, rb = (b / SP) % NB;
    const int sq = is_pi ? (b / SP) / NB : ((xr - 4) >> 1);          // sq: side (pi) / twin (Q)
    const int row0 = rb * RB;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15;
    const int row = threadIdx.x >> 4, p16 = threadIdx.x & 15;
    if constexpr (MODE != M_SAC) { if (is_pi && sq == 0 && !aux) return; }
    const float *P = is_pi ? ((MODE != M_SAC && sq == 1) ? d.P[5] : d.P[0]) : d.P[1 + sq];
    const Layer L0 = is_pi ? d.LP[0] : d.LQ[0], L1 = is_pi ? d.LP[1] : d.LQ[1], L2 = is_pi ? d.LP[2] : d.LQ[2];
    const int K0 = is_pi ? d.KP : d.KQ, KL0 = (K0 + 63) & ~63;
    const int n0 = SW * part + 16 * NTW * wave;              // this wave's first tile of the split layer
    const float *obs = S + ((is_pi && sq) ? SL.off_nobs : SL.off_obs) + (size_t)row0 * O;

    STAMP(0, 0);
    // ---- requests, in consumption order ----
    RowRegs<WIDE ? 32 : 8> rows;
    rows.issue(K0, obs, O, O, S + SL.off_act + (size_t)row0 * A, is_pi ? 0 : A, A, d.KP);
    WRing<4, WIDE ? RD : RD0> r0;
    r0.init(P + L0.offW, L0.Kp, 64 * wave, 16);
    // narrow first layers: two chunks now, the rest inside the first GEMM (a wave stalled at issue cannot commit the
    // rows or start its MFMAs; the requests pace themselves at the CU's fill rate either way)
    constexpr int PRE0 = 2;
    if constexpr (WIDE) r0.fill(K0 >> 4);
    else r0.fill_part(K0 >> 4, 0, PRE0);
    float bv0[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) bv0[t] = P[L0.offB + 64 * wave + 16 * t + c];
    SB();
    WRing<NTW, ring_depth(NTW)> r1;
    r1.init(P + L1.offW, H, n0, 16);
    if constexpr (WIDE) r1.fill(H >> 4);                     // (narrow first layers: requested inside the first GEMM)
    float bv1[NTW];
    WRing<NTH, NTW> rh;                                      // pi: head rows x this wave's 16*NTW columns
    float w3[4 * NTW];
#define FWD_A_LATE_REQUESTS()                                                                          \
    do {                                                                                               \
        _Pragma("unroll") for (int t = 0; t < NTW; ++t) bv1[t] = P[L1.offB + n0 + 16 * t + c];         \
        _Pragma("unroll") for (int u = 0; u < 4 * NTW; ++u) w3[u] = 0.f;                               \
        if (is_pi) {                                                                                   \
            rh.init(P + L2.offW, H, 0, 16, 4 * NTW * part + NTW * wave);                               \
            rh.fill(NTW);                                                                              \
        } else {                                                                                       \
            _Pragma("unroll") for (int u = 0; u < 4 * NTW; ++u) w3[u] = P[L2.offW + frag_off(0, SW * part + p16 + 16 * u, H)]; \
        }                                                                                              \
        SB();                                                                                          \
    } while (0)
    if constexpr (WIDE) FWD_A_LATE_REQUESTS();
    rows.commit(X0, KL0, K0, O, d.KP, is_pi ? 0 : A);
    lds_barrier();
    STAMP(0, 1);
    f32x4 keep1[4], zkeep[4];
    {   // first layer, all 256 features (recomputed by the SP blocks of this row-block)
        f32x4 acc[4] = {};
        if constexpr (WIDE) gemm_ring(r0, X0, KL0, K0 >> 4, acc);
        else {
            gemm_straight_pf(r0, X0, KL0, K0 >> 4, acc, r1, H >> 4, PRE0);
        }
        // Q blocks keep this quarter's PRE-activation z = W1 [s, a] + b for launch B: the first layer is linear in
        // the action, so Q_i(s, a_new) only needs z + W1[:, action chunk] (a_new - a) there instead of streaming
        // the whole first layer again
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) zkeep[t][i] = acc[t][i] + bv0[t];
        hidden_epilogue<4>(acc, 64 * wave, 16, bv0, X1, H, keep1);
    }
    lds_barrier();
    STAMP(0, 2);
    // (the slice's bias and the head weights are only needed behind the slice GEMM: requested behind the barrier, where
    //  a wave stalled at issue holds nobody up)
    if constexpr (!WIDE) FWD_A_LATE_REQUESTS();
    {   // this block's columns of the 256x256 layer
        f32x4 acc[NTW] = {};
        gemm_ring(r1, X1, H, H >> 4, acc);
        // every load of the kernel has been requested: now the feature-major copies of this block's quarter
        if (wave / NTW == part) {
            float *h1T = is_pi ? (sq == 0 ? d.PH1T : nullptr) : d.QH1T + (size_t)sq * H * B;
            if (h1T) store_features<4>(keep1, 64 * wave, 16, h1T, B, row0);
            if (MODE == M_SAC && !is_pi) store_features<4>(zkeep, 64 * wave, 16, d.QU + (size_t)sq * H * B, B, row0);
        }
        float *h2T = is_pi ? (sq == 0 ? d.PH2T : nullptr) : d.QH2T + (size_t)sq * H * B;
        slice_epilogue<NTW>(acc, bv1, wave, XS, h2T, n0, B, row0);
    }
    lds_barrier();
    STAMP(0, 3);
    if (is_pi) {
        // partial head pre-activations over these columns (each wave contracts its own 16*NTW)
        f32x4 acc[NTH] = {};
        gemm_ring(rh, XS, SW, NTW, acc, NTW * wave);
        splitk_reduce<NTH>(acc, nullptr, red, d.headpart + ((size_t)(sq * NB + rb) * SP + part) * (RB * 32), 32);
    } else {
        // partial q over these columns
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 4 * NTW; ++u) s += XS[lds_off(row, p16 + 16 * u, SW)] * w3[u];
        s = group16_sum(s);
        if (p16 == 0) d.qpart[((size_t)sq * SP + part) * B + row0 + row] = s;
    }
    STAMP(0, 4);
}

// MODE M_TD3_CRITIC: target-net blocks only (grid 2*SP*NB); the head is the TARGET policy's tanh(mean) on s' plus the
// clipped smoothing noise.  MODE M_TD3_ACTOR: Q1(s, policy(s)) blocks only (grid SP*NB) with the unit-gradient tail;
// the head is tanh(mean) of the online policy on s.
template <int NTH, bool WIDE, int SP, int MODE = M_SAC>
__global__ __launch_bounds__(256) void k_fwd_b(Dev d, const float *__restrict__ S, SlotLayout SL, StepArg sa) {
    kernarg_prefetch<sizeof(Dev) + 8 + sizeof(SlotLayout) + sizeof(StepArg)>();
    constexpr int NTW = 4 / SP, SW = 64 * NTW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int B = d.B, O = d.O, A = d.A, NB = d.NB;
    const int KLQ = (d.KQ + 63) & ~63;
    float *XQ = lds;                     // [16][KLQ]  cat(obs, action)
    float *X1 = XQ + RB * KLQ;           // [16][256]
    float *XS = X1 + RB * H;             // [16][SW]
    float *red = XS + RB * SW;           // 1024 floats (split-K scratch of the actor tail)
    constexpr bool STAGE = (SP == 4);    // 64-row weight slice (68 KB transposed) fits LDS: the actor tail reads it from there
    float *WL = red + 1024;              // [256][WLD]  (STAGE only)
    // XCD-aware map (see k_fwd_a): b % 8 in {0,1} -> Q1, {2,3} -> Q2, {4,5} -> T1, {6,7} -> T2
    const int xq = blockIdx.x >> 3, xr = blockIdx.x & 7;
    // SAC: Q1, Q2 on (s,a_new); T1, T2 on (s',a').  TD3 critic pass: b % 8 in {0..3} -> T1, {4..7} -> T2.  TD3 actor: Q1.
    const int p4 = (MODE == M_SAC) ? (xr >> 1) : (MODE == M_TD3_CRITIC ? 2 + (xr >> 2) : 0);
    const int b = (MODE == M_SAC) ? 2 * xq + (xr & 1) : (MODE == M_TD3_CRITIC ? 4 * xq + (xr & 3) : (int)blockIdx.x);
    // (TD3 critic pass: groups of four blocks per twin -- the grid is rounded up to whole groups, the blocks beyond the last
    //  row-block leave.  Round 2 launched 2 SP NB blocks whatever SP NB was: with SP NB % 4 == 2 the last two row-blocks of
    //  twin 2 were never computed and twin 1 ran two row-blocks past the batch -- found by scratch/fuzz_fused.py in round 3)
    if (MODE == M_TD3_CRITIC && b >= SP * NB) return;
    const int part = b % SP, rb = b / SP;
    const int side = p4 >> 1, pass = 2 + p4;
    const int row0 = rb * RB;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15;
    const int row = threadIdx.x >> 4, a = threadIdx.x & 15, grow = row0 + row;
    const float *PQ = d.P[1 + p4];                           // nets 1,2 (Q1,Q2) and 3,4 (T1,T2)
    const int n0 = SW * part + 16 * NTW * wave;
    const bool own_s = (p4 == 0) && (part == 0), own_n = (p4 == 2) && (part == 0);

    STAMP(1, 0);
    STAMP_CLK(1, 14);
#ifdef SAC_STAMPS
    { const int kb = d.NB; asm volatile("" ::"s"(kb)); }     // first kernel argument has arrived
    STAMP(1, 8);
#endif
    // ---- requests, in consumption order ----
    float hm[SP], hr[SP], hbm = 0.f, hbr = 0.f, epsin = 0.f;
    const float *epp = side ? d.eps2 : d.eps1;
    {
        const float *hp = d.headpart + (size_t)(side * NB + rb) * SP * (RB * 32) + row * 32;
        const int am = (a < A) ? a : 0;
        const float *PH = (MODE == M_TD3_CRITIC) ? d.P[5] : d.P[0];      // head bias: target policy / online policy
#pragma unroll
        for (int p = 0; p < SP; ++p) {
            hm[p] = hp[p * (RB * 32) + am];
            hr[p] = (MODE == M_SAC) ? hp[p * (RB * 32) + A + am] : 0.f;  // (TD3 heads have no log-std rows)
        }
        hbm = PH[d.LP[2].offB + am];
        if constexpr (MODE == M_SAC) hbr = PH[d.LP[2].offB + A + am];
        if (MODE != M_TD3_ACTOR && epp) epsin = epp[grow * A + am];
    }
    // Q1/Q2(s, a_new) blocks: launch A left the first-layer pre-activations z of Q_i(s, a) in QU and the layer is linear
    // in the action, so z + W1[:, action chunk] (a_new - a) needs only the action chunk of the weights and of the
    // input; target-net blocks do the whole layer.  (Wide first layers: the ring starts AT the action chunk.)
    const bool act_only = (MODE == M_SAC) && (p4 < 2);
    constexpr int D0 = WIDE ? RD : RD0, Q0 = D0 / 4;         // first-layer ring, issued in four pieces
    const int KS0 = d.KQ >> 4;
    const int lo0 = (act_only && !WIDE) ? KS0 - 1 : 0;       // narrow: first chunk of the first layer this block computes
    const int so0 = (act_only && WIDE) ? KS0 - 1 : 0;        // wide: the ring's first chunk
    const int ks0 = (act_only && WIDE) ? 1 : KS0;            // chunks the ring walks
    RowRegs<WIDE ? 32 : 8> rows;
    f32x4 acc0[4] = {}, keep1[4];                            // first-layer accumulators (start: 0, or launch A's z)
    if (!act_only) rows.issue(d.KQ, S + (side ? SL.off_nobs : SL.off_obs) + (size_t)row0 * O, O, O, nullptr, 0, 0, 0);
    WRing<4, D0> r0;
    r0.init(PQ + d.LQ[0].offW, d.LQ[0].Kp, 64 * wave, 16, so0);
    float bv0[4];
    float abat = 0.f;                                        // the batch action a (act_only blocks)
    if (act_only) {
        abat = S[SL.off_act + (size_t)grow * A + ((a < A) ? a : 0)];
        const float *qu = d.QU + (size_t)p4 * H * B + frag_off(64 * wave + c, row0 + 4 * (lane >> 4), B);
#pragma unroll
        for (int t = 0; t < 4; ++t) { acc0[t] = ld4(qu + (size_t)t * (B >> 4) * 256); bv0[t] = 0.f; }   // next feature tile
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) bv0[t] = PQ[d.LQ[0].offB + 64 * wave + 16 * t + c];
    }
    // the 256x256 slice's ring: target-net blocks request it behind the first layer's weights (around the last piece
    // of the head math), act_only blocks -- which stream no first layer -- in four pieces from the start
    constexpr int RD1 = ring_depth(NTW), QR = RD1 / 4;
    WRing<NTW, RD1> r1;
    r1.init(PQ + d.LQ[1].offW, H, n0, 16);
    r0.fill_part(ks0, 0, Q0, lo0);
    r1.fill_part(H >> 4, 0, act_only ? QR : 0);
    SB();
    STAMP(1, 9);
#pragma unroll
    for (int p = 0; p < SP; ++p) { USE_FROM_HERE(hm[p]); USE_FROM_HERE(hr[p]); }
    USE_FROM_HERE(hbm); USE_FROM_HERE(hbr); USE_FROM_HERE(epsin);
    STAMP(1, 10);
    if (!act_only) rows.commit(XQ, KLQ, d.KQ, O, 0, 0, d.KP, d.KP + 16);     // the head writes the action chunk
    STAMP(1, 1);
    SB();
    r0.fill_part(ks0, Q0, 2 * Q0, lo0);
    r1.fill_part(H >> 4, QR, act_only ? 2 * QR : 0);
    SB();
    // ---- tanh-Gaussian head on this block's rows (every block of the row-block computes the same), in three
    // pieces with the rest of the weight requests in between ----
    float lp = 0.f, mean = 0.f, raw = 0.f, lstd = 0.f, stdv = 1.f, eps = 0.f, zz = 0.f, act = 0.f;
    if (a < A) {
        mean = hm[0]; raw = hr[0];
#pragma unroll
        for (int p = 1; p < SP; ++p) { mean += hm[p]; raw += hr[p]; }     // fixed order
        mean += hbm; raw += hbr;
        if constexpr (MODE == M_SAC) {
            lstd = fminf(fmaxf(raw, LOG_SIG_MIN), LOG_SIG_MAX);
            stdv = expf(lstd);
        }
        if constexpr (MODE != M_TD3_ACTOR)
            eps = epp ? epsin
                      : philox_normal(d.noise_seed, (unsigned long long)sa.step_now, (unsigned)(grow * 16 + a), side ? 1u : 0u);
    }
    SB();
    r0.fill_part(ks0, 2 * Q0, 3 * Q0, lo0);
    r1.fill_part(H >> 4, 2 * QR, act_only ? 3 * QR : 0);
    SB();
    if (a < A) {
        if constexpr (MODE == M_SAC) {
            zz = __fadd_rn(mean, __fmul_rn(stdv, eps));                  // TanhNormal.rsample
            act = tanhf(zz);
        } else if constexpr (MODE == M_TD3_CRITIC) {
            // TD3 target smoothing: a' + clamp(N(0,1) * sigma, +-clip); the sum is NOT clipped to the action range
            zz = tanhf(mean);
            act = zz + fminf(fmaxf(eps * d.td3_sigma, -d.td3_clip), d.td3_clip);
        } else {
            zz = mean;                                                   // pre-tanh output of the online policy
            act = tanhf(mean);
        }
    }
    // the whole action chunk (0 beyond A); act_only blocks contract the difference to the batch action
    XQ[lds_off(row, d.KP + a, KLQ)] = (a < A) ? (act_only ? act - abat : act) : 0.f;
    SB();
    r0.fill_part(ks0, 3 * Q0, D0, lo0);
    r1.fill_part(H >> 4, act_only ? 3 * QR : 0, act_only ? RD1 : RD1 / 2);
    SB();
    if (MODE == M_SAC && a < A) {
        const float dd = __fsub_rn(zz, mean);                            // Normal.log_prob(z)
        const float var = __fmul_rn(stdv, stdv);
        const float nlp = -(dd * dd) / (2.0f * var) - logf(stdv) - 0.91893853320467274178f;
        lp = nlp - logf(1.0f - act * act + TANH_EPS);
    }
    SB();
    r1.fill_part(H >> 4, RD1 / 2, act_only ? 0 : RD1);
    float bv1[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) bv1[t] = PQ[d.LQ[1].offB + n0 + 16 * t + c];
    float w3[4 * NTW];
#pragma unroll
    for (int u = 0; u < 4 * NTW; ++u) w3[u] = PQ[d.LQ[2].offW + frag_off(0, SW * part + a + 16 * u, H)];
    SB();
    const float lsum = group16_sum(lp);
    if (own_s && a == 0) red[row] = (grow < d.Bt) ? lsum : 0.f;          // (pad rows carry no weight)
    lds_barrier();
    float lsum_blk = 0.f;
    if (own_s && threadIdx.x == 0) {          // this row-block's sum(log_pi), fixed order
        for (int i = 0; i < RB; ++i) lsum_blk += red[i];
    }
    // (the head's global results are stored behind the slice GEMM, once every load has been requested: vmcnt
    //  retires in issue order, a store in front of the weight requests would sit in all their waits)
    STAMP(1, 2);
    // ---- Q / target-Q net on cat(obs, action) ----
    {
        if constexpr (WIDE) gemm_ring(r0, XQ, KLQ, ks0, acc0, so0);
        else gemm_straight(r0, XQ, KLQ, KS0, acc0, lo0);
        hidden_epilogue<4>(acc0, 64 * wave, 16, bv0, X1, H, keep1);
    }
    // Q1/Q2(s,a_new) blocks go on to the actor's input gradient (below): request its weights now (the first
    // layer's ring registers are free), so they arrive under the 256x256 slice.
    WRing<STAGE ? 1 : 4, STAGE ? 1 : 4> rw;                  // W2^T: all 256 first-hidden features x this block's columns
    WRing<1, 4> ra;                                          // W1^T action rows: this wave's 64 first-hidden features
    if (p4 < 2) {
        SB();
        const float *PT = d.PT[1 + p4];
        if constexpr (!STAGE) {
            rw.init(PT + d.LQ[1].offWt, H, 64 * wave, 16, 4 * NTW * part);
            rw.fill(4 * NTW);
        }
        ra.init(PT + d.LQ[0].offWt, H, d.KP, 16, 4 * wave);      // rows KP.. of W1^T = the action columns
        ra.fill(4);
        SB();
    }
    lds_barrier();
    STAMP(1, 3);
    {
        f32x4 acc[NTW] = {};
        if (STAGE && p4 < 2) gemm_ring<true>(r1, X1, H, H >> 4, acc, 0, WL + 16 * NTW * wave);
        else gemm_ring(r1, X1, H, H >> 4, acc);
        if (p4 < 2 && (wave / NTW == part))       // (all loads of the forward part have been requested)
            store_features<4>(keep1, 64 * wave, 16, d.QH1T + (size_t)pass * H * B, B, row0);
        if (own_s) {
            if (a < A) {
                d.mu[grow * 16 + a] = mean;
                if constexpr (MODE == M_SAC) {
                    d.ls[grow * 16 + a] = lstd;
                    d.lsok[grow * 16 + a] = (raw >= LOG_SIG_MIN && raw <= LOG_SIG_MAX) ? 1.f : 0.f;
                    d.z[grow * 16 + a] = zz;
                    d.epsv[grow * 16 + a] = eps;
                }
            }
            d.anew[grow * 16 + a] = act;                      // (0 beyond A)
            if constexpr (MODE == M_SAC) {
                if (a == 0) d.logpi[grow] = lsum;
                if (threadIdx.x == 0) d.part_logpi[rb] = lsum_blk;
            }
        } else if (own_n) {
            d.a2[grow * 16 + a] = act;
            if (MODE == M_SAC && a == 0) d.logpi2[grow] = lsum;
        }
        float *h2T = (p4 < 2) ? d.QH2T + (size_t)pass * H * B : nullptr;
        slice_epilogue<NTW>(acc, bv1, wave, XS, h2T, n0, B, row0);
    }
    lds_barrier();
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 4 * NTW; ++u) s += XS[lds_off(row, a + 16 * u, SW)] * w3[u];
    s = group16_sum(s);
    if (a == 0) d.qpart[((size_t)pass * SP + part) * B + grow] = s;
    STAMP(1, 4);
    STAMP_CLK(1, 15);
    if (p4 >= 2) return;
    // ---- actor path: UNIT input gradient of Q_i(s, a_new) (dq = 1), partial over this block's columns ----
    // The chain dq -> dh2 -> dh1 -> da is linear in the per-row scalar dq = -sel/B, which needs min(Q1, Q2),
    // i.e. the sum of all partial q's: the policy-backward blocks of the next launch apply it to the sum of
    // these partials.  Everything else the chain needs is already in this block's LDS.
#pragma unroll
    for (int u = 0; u < 4 * NTW; ++u) {                      // dq/dh2 = w3 * relu'(h2), in place (own elements)
        const int off = lds_off(row, a + 16 * u, SW);
        XS[off] = (XS[off] > 0.f) ? w3[u] : 0.f;
    }
    lds_barrier();
    STAMP(1, 5);
    {   // partial dq/dh1 over ALL first-hidden features = dq/dh2[:, slice] . W2[slice, :], masked in place
        f32x4 acc[4] = {};
        if constexpr (STAGE) gemm_lds_rows<4, 4 * NTW>(WL, 64 * wave, XS, SW, acc);
        else gemm_ring(rw, XS, SW, 4 * NTW, acc);
        const int g = lane >> 4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int off = lds_off(4 * g + i, 64 * wave + 16 * t + c, H);
                X1[off] = (X1[off] > 0.f) ? acc[t][i] : 0.f;
            }
        }
    }
    lds_barrier();
    STAMP(1, 6);
    {   // partial dq/da = dq/dh1 . W1[:, O:O+A]  (contraction split over the waves)
        f32x4 acc[1] = {};
        gemm_ring(ra, X1, H, 4, acc, 4 * wave);
        splitk_reduce<1>(acc, nullptr, red, d.dapart + (((size_t)p4 * SP + part) * B + row0) * 16, 16);
    }
    STAMP(1, 7);
}

// ------------------------------------------------------------------------------------------
// C k_bwd: the two critic backward passes (dL/dh kept for dW) and the policy backward, side by side.
// Both depend only on launch B (q partials, sum(log_pi) partials, unit d/da partials).
// ------------------------------------------------------------------------------------------
// critic Q_i(s,a): y, dq = 2(q - y)/B, dL/dh2 (all features, recomputed by the SP blocks), dL/dh1 (this
// block's features)
template <int SP, int MODE = M_SAC>
__device__ __forceinline__ void critic_bwd_block(const Dev &d, const float *__restrict__ S, const SlotLayout &SL,
                                                 const StepArg &sa, int qi, int b) {
    constexpr int NTW = 4 / SP, SW = 64 * NTW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int B = d.B, NB = d.NB;
    float *X2 = lds;                 // dL/dh2 row-block [16][256]
    __shared__ float s_dq[RB];
    const int part = b % SP, rb = b / SP;
    const int row0 = rb * RB;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const float *P = d.P[1 + qi];
    const float *PT = d.PT[1 + qi];
    const float invB = 1.0f / (float)d.Bt;                   // means run over the true batch; pad rows get dq = 0
    const float *h2T = d.QH2T + (size_t)qi * H * B;
    const float *h1T = d.QH1T + (size_t)qi * H * B;
    const int n0 = SW * part + 16 * NTW * wave;
    const long long oB3 = d.LQ[2].offB;

    // ---- up-front requests, in consumption order (vmcnt retires in issue order) ----
    const float b3a = sload(d.P[3] + oB3), b3b = sload(d.P[4] + oB3), b3q = sload(P + oB3);
    float qa[SP], qb[SP], qq[SP], in_c = 0.f, in_r = 0.f, in_t = 0.f;      // loads only: no use before SB
#pragma unroll
    for (int p = 0; p < SP; ++p) { qa[p] = 0.f; qb[p] = 0.f; qq[p] = 0.f; }
    if (threadIdx.x < RB) {
        const int r = row0 + threadIdx.x;
#pragma unroll
        for (int p = 0; p < SP; ++p) {
            qa[p] = d.qpart[((size_t)4 * SP + p) * B + r];
            qb[p] = d.qpart[((size_t)5 * SP + p) * B + r];
            qq[p] = d.qpart[((size_t)qi * SP + p) * B + r];
        }
        if constexpr (MODE == M_SAC) in_c = d.logpi2[r];
        in_r = S[SL.off_rew + r]; in_t = S[SL.off_term + r];
    }
    const int k = threadIdx.x;
    const float wk = P[d.LQ[2].offW + frag_off(0, k, H)];
    f32x4 h2v[4];
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) h2v[qd] = ld4(h2T + frag_off(k, row0 + 4 * qd, B));
    SB();
    WRing<NTW, ring_depth(NTW)> r1;
    r1.init(PT + d.LQ[1].offWt, H, n0, 16);
    r1.fill(H >> 4);
    SB();
    f32x4 h1v[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) h1v[t] = ld4(h1T + frag_off(n0 + 16 * t + c, row0 + 4 * g, B));
    SB();
    // (scalar loads + a few scalar flops; placed behind the vector-load burst so its s_waitcnt does not delay it)
    float alpha = 0.f;                                                   // (TD3: no entropy term in the target)
    if constexpr (MODE == M_SAC)
        alpha = alpha_step(d.ctl, d.part_logpi, NB, d.Bt, d.target_entropy, d.alpha_lr, d.auto_alpha, sa.bc1, sa.bc2s).alpha;
#pragma unroll
    for (int p = 0; p < SP; ++p) { USE_FROM_HERE(qa[p]); USE_FROM_HERE(qb[p]); USE_FROM_HERE(qq[p]); }
    USE_FROM_HERE(in_c); USE_FROM_HERE(in_r); USE_FROM_HERE(in_t);
    float va = 0.f, vb = 0.f, vq = 0.f, yv = 0.f, dq = 0.f;
    if (threadIdx.x < RB) {
        va = qa[0]; vb = qb[0]; vq = qq[0];
#pragma unroll
        for (int p = 1; p < SP; ++p) { va += qa[p]; vb += qb[p]; vq += qq[p]; }      // fixed order
        va += b3a;                                                       // T1(s',a')
        vb += b3b;                                                       // T2(s',a')
        vq += b3q;                                                       // Q_i(s,a)
        const float tq = fminf(va, vb) - alpha * in_c;
        yv = bellman_target(d.reward_scale, in_r, in_t, d.discount, tq);
        dq = (row0 + (int)threadIdx.x < d.Bt) ? 2.0f * (vq - yv) * invB : 0.f;
        s_dq[threadIdx.x] = dq;
    }
    lds_barrier();
    // dL/dh2 = dq * w3 * relu'(h2)   (thread = feature k, 4-row groups); kept for dW by the owner block
    f32x4 gv2[4];
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            gv2[qd][i] = (h2v[qd][i] > 0.f) ? s_dq[4 * qd + i] * wk : 0.f;
            X2[lds_off(4 * qd + i, k, H)] = gv2[qd][i];
        }
    }
    lds_barrier();
    // dL/dh1[:, this block's features] = (dL/dh2 . W2)[:, slice] * relu'(h1)
    {
        f32x4 acc[NTW] = {};
        gemm_ring(r1, X2, H, H >> 4, acc);
        // global results only now that every load has been requested (vmcnt retires loads and stores in issue order)
        if (threadIdx.x < RB && part == 0) {
            const int r = row0 + threadIdx.x;
            d.q[(size_t)qi * B + r] = vq;
            d.dq16T[(size_t)qi * 16 * B + frag_off(0, r, B)] = dq;      // row 0 of the padded [16][B]
            if (qi == 0) { d.y[r] = yv; d.q[4 * (size_t)B + r] = va; d.q[5 * (size_t)B + r] = vb; }
        }
        if ((k / SW) == part) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) st4(d.dQH2T + (size_t)qi * H * B + frag_off(k, row0 + 4 * qd, B), gv2[qd]);
        }
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            f32x4 gv;
#pragma unroll
            for (int i = 0; i < 4; ++i) gv[i] = (h1v[t][i] > 0.f) ? acc[t][i] : 0.f;
            st4(d.dQH1T + (size_t)qi * H * B + frag_off(n0 + 16 * t + c, row0 + 4 * g, B), gv);
        }
    }
}

// policy backward (actor loss = mean(alpha*log_pi - min Q)), analytic head gradient.
//   da         = sum_i dq_i * (unit d Q_i/da from launch B),  dq_i = -sel_i / B  (torch.min backward)
//   dL/dz      = da*(1-a^2) + (alpha/B) * 2a(1-a^2)/(1-a^2+1e-6)
//   dL/dmu     = dL/dz                      (the Normal terms cancel exactly under rsample)
//   dL/dlogstd = dL/dz * std*eps - alpha/B  (masked by the clamp)
// MODE M_TD3_ACTOR: loss = -mean(Q1(s, tanh(mean))): da = -(1/B) dQ1/da, dL/dmean = da (1 - a^2); no log-std head.
template <int NTH, int SP, int MODE = M_SAC>
__device__ __forceinline__ void policy_bwd_block(const Dev &d, const StepArg &sa, int b) {
    constexpr int NTW = 4 / SP, SW = 64 * NTW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int B = d.B, A = d.A;
    float *XH = lds;                 // [16][64] head gradient row-block
    float *X2 = XH + RB * 64;        // [16][256] dL/dh2 (recomputed by the SP blocks)
    const int part = b % SP, rb = b / SP, row0 = rb * RB;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;
    const float *PT = d.PT[0];
    const float invB = 1.0f / (float)d.Bt;                   // means run over the true batch; pad rows get no gradient
    const int n0 = SW * part + 16 * NTW * wave;
    const long long oB3 = d.LQ[2].offB;

    // ---- up-front requests, in consumption order (vmcnt retires in issue order) ----
    const int row = threadIdx.x >> 4, a = threadIdx.x & 15;
    const int gi = (row0 + row) * 16 + a;
    const float b3a = sload(d.P[1] + oB3), b3b = sload(d.P[2] + oB3);
    float act = 0.f, dap[2 * SP], qa[SP], qb[SP], lsv = 0.f, epv = 0.f, okv = 0.f;   // loads only: no use before SB
#pragma unroll
    for (int p = 0; p < 2 * SP; ++p) dap[p] = 0.f;
#pragma unroll
    for (int p = 0; p < SP; ++p) {                           // Q1, Q2(s, a_new) partials of this row
        qa[p] = d.qpart[((size_t)2 * SP + p) * B + row0 + row];
        qb[p] = (MODE == M_SAC) ? d.qpart[((size_t)3 * SP + p) * B + row0 + row] : 0.f;
    }
    if (a < A) {
        act = d.anew[gi];
#pragma unroll
        for (int p = 0; p < 2 * SP; ++p) dap[p] = (MODE == M_SAC || p < SP) ? d.dapart[(size_t)p * B * 16 + gi] : 0.f;
        if constexpr (MODE == M_SAC) { lsv = d.ls[gi]; epv = d.epsv[gi]; okv = d.lsok[gi]; }
    }
    SB();
    WRing<4> rh;
    rh.init(PT + d.LP[2].offWt, d.LP[2].Np, 64 * wave, 16);
    rh.fill(NTH);
    f32x4 h2v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) h2v[t] = ld4(d.PH2T + frag_off(64 * wave + 16 * t + c, row0 + 4 * g, B));
    SB();
    // One tile per wave (SP = 4): the contraction over the 256 dL/dh2 features runs as two halves of eight k-chunks,
    // each with its own ring (everything requested up front, no refills) and its own accumulators, added at the end --
    // the order in which the fused step (sac_fused.h), where two waves share a tile, sums them.
    WRing<NTW, ring_depth(NTW)> r1, r1b;
    r1.init(PT + d.LP[1].offWt, H, n0, 16);
    if constexpr (NTW == 1) {
        r1.fill(8);
        r1b.init(PT + d.LP[1].offWt, H, n0, 16, 8);
        r1b.fill(8);
    } else {
        r1.fill(H >> 4);
    }
    SB();
    f32x4 h1v[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) h1v[t] = ld4(d.PH1T + frag_off(n0 + 16 * t + c, row0 + 4 * g, B));
    SB();
    for (int e = threadIdx.x; e < RB * 64; e += 256) XH[e] = 0.f;
    float alpha = 0.f;
    if constexpr (MODE == M_SAC)
        alpha = alpha_step(d.ctl, d.part_logpi, d.NB, d.Bt, d.target_entropy, d.alpha_lr, d.auto_alpha, sa.bc1, sa.bc2s).alpha;
    lds_barrier();
    USE_FROM_HERE(act); USE_FROM_HERE(lsv); USE_FROM_HERE(epv); USE_FROM_HERE(okv);
#pragma unroll
    for (int p = 0; p < 2 * SP; ++p) USE_FROM_HERE(dap[p]);
#pragma unroll
    for (int p = 0; p < SP; ++p) { USE_FROM_HERE(qa[p]); USE_FROM_HERE(qb[p]); }
    float qnew1 = 0.f, qnew2 = 0.f, dz = 0.f, dls = 0.f;
    {
        float va = qa[0], vb = qb[0];
#pragma unroll
        for (int p = 1; p < SP; ++p) { va += qa[p]; vb += qb[p]; }                   // fixed order
        va += b3a; vb += b3b;                                                        // Q1, Q2(s, a_new)
        // torch.min backward: the smaller one takes the gradient, a tie splits it
        const float sel1 = (MODE == M_SAC) ? ((va < vb) ? 1.0f : ((va == vb) ? 0.5f : 0.0f)) : 1.0f;   // TD3: Q1 only
        const float dq1 = -invB * sel1, dq2 = -invB * (1.0f - sel1);
        qnew1 = va; qnew2 = vb;
        if (a < A && row0 + row < d.Bt) {
            float da1 = dap[0], da2 = dap[SP];
#pragma unroll
            for (int p = 1; p < SP; ++p) { da1 += dap[p]; da2 += dap[SP + p]; }      // fixed order
            const float da = actor_da(da1, dq1, da2, dq2);
            const float om = 1.0f - act * act;
            if constexpr (MODE == M_SAC) {
                const float alpha_invB = __fmul_rn(alpha, invB);
                dz = actor_dz(da, om, alpha_invB, act);
                const float stdv = expf(lsv);
                dls = actor_dls(dz, stdv, epv, alpha_invB, okv);
                XH[lds_off(row, A + a, 64)] = dls;
            } else {
                dz = da * om;
            }
            XH[lds_off(row, a, 64)] = dz;
        }
    }
    lds_barrier();
    f32x4 gk2[4];
    {
        f32x4 acc[4] = {};
        gemm_ring(rh, XH, 64, NTH, acc);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n = 64 * wave + 16 * t + c;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                gk2[t][i] = (h2v[t][i] > 0.f) ? acc[t][i] : 0.f;
                X2[lds_off(4 * g + i, n, H)] = gk2[t][i];
            }
        }
    }
    lds_barrier();
    {
        f32x4 acc[NTW] = {};
        if constexpr (NTW == 1) {
            f32x4 acc_hi[NTW] = {};
            gemm_ring(r1, X2, H, 8, acc);
            gemm_ring(r1b, X2, H, 8, acc_hi, 8);
            acc[0] += acc_hi[0];
        } else {
            gemm_ring(r1, X2, H, H >> 4, acc);
        }
        // global results only now that every load has been requested (vmcnt retires loads and stores in issue order)
        if (part == 0) {
            if (a == 0) { d.q[2 * (size_t)B + row0 + row] = qnew1; d.q[3 * (size_t)B + row0 + row] = qnew2; }
            if (a < A) {
                d.dheadT[frag_off(a, row0 + row, B)] = dz;
                if constexpr (MODE == M_SAC) d.dheadT[frag_off(A + a, row0 + row, B)] = dls;
            }
        }
        if (wave / NTW == part) store_features<4>(gk2, 64 * wave, 16, d.dPH2T, B, row0);
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            f32x4 gv;
#pragma unroll
            for (int i = 0; i < 4; ++i) gv[i] = (h1v[t][i] > 0.f) ? acc[t][i] : 0.f;
            st4(d.dPH1T + frag_off(n0 + 16 * t + c, row0 + 4 * g, B), gv);
        }
    }
}

// Block -> work map.  Three equal groups of SP*NB blocks: critic Q1, critic Q2, policy.  While they fit one
// per CU on six XCDs (3*SP*NB <= 192) the map is XCD-aware (b % 8 in {0,1} -> Q1, {2,3} -> Q2, {4,5} -> policy,
// {6,7} idle: an XCD's L2 pulls one network's transposed weights); larger batches use every CU instead.
// MODE M_TD3_CRITIC: critic blocks only (grid: whole groups of eight, b % 8 in {0..3} -> Q1, {4..7} -> Q2); M_TD3_ACTOR: policy blocks only.
template <int NTH, int SP, int MODE = M_SAC>
__global__ __launch_bounds__(256) void k_bwd(Dev d, const float *__restrict__ S, SlotLayout SL, StepArg sa, int compact) {
    kernarg_prefetch<sizeof(Dev) + 8 + sizeof(SlotLayout) + sizeof(StepArg)>();
    if constexpr (MODE == M_TD3_CRITIC) {
        const int bq = 4 * (blockIdx.x >> 3) + (blockIdx.x & 3);
        if (bq >= SP * d.NB) return;         // (grid rounded up to whole groups of four blocks per twin: see k_fwd_b)
        critic_bwd_block<SP, MODE>(d, S, SL, sa, (blockIdx.x & 7) >> 2, bq);
    } else if constexpr (MODE == M_TD3_ACTOR) {
        policy_bwd_block<NTH, SP, MODE>(d, sa, blockIdx.x);
    } else {
        int cls, b;
        if (compact) { cls = (blockIdx.x & 7) >> 1; b = 2 * (blockIdx.x >> 3) + (blockIdx.x & 1); }
        else { cls = blockIdx.x % 3; b = blockIdx.x / 3; }
        if (cls > 2) return;
        if (cls < 2) critic_bwd_block<SP>(d, S, SL, sa, cls, b);
        else policy_bwd_block<NTH, SP>(d, sa, b);
    }
}

// ------------------------------------------------------------------------------------------
// K5: weight gradients + Adam + Polyak, tile-owner parallel.  One WG owns a 16 (out) x 64 (in)
// tile of one layer: dW = sum_b dY[b][n] X[b][k] with the batch split over the 4 waves (MFMA
// 16x16x4, both operands read feature-major so every lane load is 16 B), reduced through LDS,
// then the owner applies torch.optim.Adam's update, writes the forward ([n][k]) and transposed
// ([k][n]) copies of the new weights and (every target_update_period steps) the Polyak average of
// the target net.  Adam moments live in the transposed layout, so the owner's old weight / m / v
// are three 16-B loads per lane, issued before the MFMA loop so their latency overlaps it.
// The work table is a kernel argument (no descriptor load in front of the data loads).
// The extra last block computes the step's diagnostics.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void adam_update(float &p, float &m, float &v, float g, float step_size, float bc2s) {
    m = m + ADAM_1MB1 * (g - m);
    v = v * ADAM_B2 + ADAM_1MB2 * g * g;
    const float denom = sqrtf(v) / bc2s + ADAM_EPS;
    p = p + (-step_size * m) / denom;
}

constexpr int NSTAT = 6;     // q1, q2, q_target, log_pi, mu, log_std

// TD3 statistics (rlkit TD3Trainer: QF1/QF2 Loss, Policy Loss, Q1/Q2 Predictions, Q Targets, Bellman Errors 1/2,
// Policy Action), in the slots of the SAC vector: 16-19 Bellman Errors 1, 20-23 Bellman Errors 2, 24-27 Policy Action.
// sa.pad bit 0: the critic part (every step, from the critic launch), bit 1: the policy part (policy / statistics
// steps, from the launch that follows the actor pass).
__device__ __forceinline__ void td3_diagnostics(const Dev &d, const StepArg &sa, float *red) {
    const int B = d.Bt, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, loop_pos = sa.loop_pos;   // statistics over the true batch
    const size_t Bs = (size_t)d.B;                           // (row stride of the per-row arrays: the padded batch)
    constexpr int NS = 6;        // q1, q2, y, be1, be2, policy action
    double sm[NS], sq[NS], lsum = 0;
    float mx[NS], mn[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) { sm[q] = 0; sq[q] = 0; mx[q] = -INFINITY; mn[q] = INFINITY; }
    auto acc1 = [&](int q, float v) { sm[q] += v; sq[q] += (double)v * v; mx[q] = fmaxf(mx[q], v); mn[q] = fminf(mn[q], v); };
    if (sa.pad & 1)
        for (int i = threadIdx.x; i < B; i += 256) {
            const float yv = d.y[i], q1 = d.q[i], q2 = d.q[Bs + i];
            acc1(0, q1); acc1(1, q2); acc1(2, yv); acc1(3, (q1 - yv) * (q1 - yv)); acc1(4, (q2 - yv) * (q2 - yv));
        }
    if (sa.pad & 2) {
        const float b3 = sload(d.P[1] + d.LQ[2].offB);
        for (int i = threadIdx.x; i < B; i += 256) {                                             // Q1(s, policy(s))
            float qv = d.qpart[((size_t)2 * d.sp) * Bs + i];
            for (int p = 1; p < d.sp; ++p) qv += d.qpart[((size_t)2 * d.sp + p) * Bs + i];       // fixed order
            lsum += (double)(qv + b3);
            d.q[2 * Bs + i] = qv + b3;
        }
        for (int e = threadIdx.x; e < B * d.A; e += 256) { const int i = e / d.A; acc1(5, d.anew[i * 16 + (e - i * d.A)]); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            sm[q] += __shfl_xor(sm[q], o); sq[q] += __shfl_xor(sq[q], o);
            mx[q] = fmaxf(mx[q], __shfl_xor(mx[q], o)); mn[q] = fminf(mn[q], __shfl_xor(mn[q], o));
        }
        lsum += __shfl_xor(lsum, o);
    }
    double *sh = reinterpret_cast<double *>(red);      // [4 waves][32]
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            sh[wave * 32 + q] = sm[q]; sh[wave * 32 + 6 + q] = sq[q]; sh[wave * 32 + 12 + q] = mx[q]; sh[wave * 32 + 18 + q] = mn[q];
        }
        sh[wave * 32 + 24] = lsum;
    }
    lds_barrier();
    // TD3's vector is filled by two launches, its policy half on policy steps only: every launch updates the DEVICE copy
    // (which so always holds the most recent value of each entry), and a launch whose caller reads the diagnostics
    // (sa.pad2 bit 1: the last step of a loop, single steps) copies the whole vector to the mapped host buffer at its end
    float *const dlast = d.diag_dev;
    auto put = [&](int di, float v) {
        dlast[di] = v;
        if (loop_pos == 0) d.diag_first[di] = v;
        if (loop_pos < DIAG_TRACE_CAP) d.diag_trace[(size_t)loop_pos * SAC_DIAG_N + di] = v;
    };
    if (threadIdx.x < NS) {
        const int q = threadIdx.x;
        if ((q < 5) ? (sa.pad & 1) : (sa.pad & 2)) {
            double s = 0, s2 = 0, MX = -INFINITY, MN = INFINITY;
            for (int w = 0; w < 4; ++w) {
                s += sh[w * 32 + q]; s2 += sh[w * 32 + 6 + q];
                MX = fmax(MX, sh[w * 32 + 12 + q]); MN = fmin(MN, sh[w * 32 + 18 + q]);
            }
            const double cnt = (q < 5) ? (double)B : (double)B * d.A;
            const double mean = s / cnt;
            double var = s2 / cnt - mean * mean;
            if (var < 0) var = 0;
            const int base = (q < 3) ? SAC_D_Q1_MEAN + 4 * q : SAC_D_LOGPI_MEAN + 4 * (q - 3);
            put(base, (float)mean); put(base + 1, (float)sqrt(var)); put(base + 2, (float)MX); put(base + 3, (float)MN);
            if (q == 3) put(SAC_D_QF1_LOSS, (float)mean);           // MSE = mean Bellman error
            if (q == 4) put(SAC_D_QF2_LOSS, (float)mean);
        }
    } else if (threadIdx.x == 64 && (sa.pad & 2)) {
        double s = 0;
        for (int w = 0; w < 4; ++w) s += sh[w * 32 + 24];
        put(SAC_D_POLICY_LOSS, (float)(-s / B));
    }
    if (sa.pad2 & 2u) {
        __syncthreads();
        if (threadIdx.x < SAC_DIAG_N) d.diag_last[threadIdx.x] = ld_sc1(d.diag_dev + threadIdx.x);
    }
}

// The body of the weight-gradient / Adam launch for virtual block `vblock` (0: the step's diagnostics; 1 .. njobs: tile
// owners).  (A function of its own since the one-launch experiment -- this body as a phase D of k_abc behind a grid-wide
// counter hand-off: the hand-off took 6.5 us against ~3 us for the dispatch boundary plus start-up it replaced, DESIGN.md
// section 7 -- and kept that way.)  red: 4096 floats, redb: 128 floats.
__device__ __forceinline__ void dw_adam_body(const Dev &d, const DwTable &T, const float *__restrict__ S, const StepArg &sa,
                                             float *red, float *redb, float *trs, int vblock, unsigned aborted) {
    const int B = d.B;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    const Ctl *cp = d.ctl;
    // block 0: the step's diagnostics (dispatched first: it is the longest block of the launch); blocks 1 .. njobs: tiles.
    // Diagnostics go to the mapped pinned host buffer only on the steps whose caller reads them (sa.pad2 bit 1: the last
    // step of a loop, single steps) -- a store over the link on every step costs the launch ~1 us at its end -- and to a
    // device scratch otherwise (diag_first: the first step of a loop, always host).
    float *const dlast = (sa.pad2 & 2u) ? d.diag_last : d.diag_dev;
    const bool keep_grad = (sa.pad2 & 2u) != 0;      // the flat gradient copies (sac_debug_fetch "g_*") follow the same rule
    const int jb = vblock - 1;
    if (jb >= 0) {
        STAMP(4 - 2 * (sa.loop_pos & 1), 0);
        int li = 0;
#pragma unroll
        for (int q = 1; q < NDW; ++q) li = (jb >= T.job0[q]) ? q : li;
        // the layer descriptor through the scalar cache (uniform index): one s_load burst, no vector
        // load + readfirstlane round trip in front of the operand loads
        union { DwLayer J; unsigned long long w[sizeof(DwLayer) / 8]; } ud;
        {
            const __attribute__((address_space(4))) unsigned long long *src =
                (const __attribute__((address_space(4))) unsigned long long *)(uintptr_t)(T.L + li);
#pragma unroll
            for (int q = 0; q < (int)(sizeof(DwLayer) / 8); ++q) ud.w[q] = src[q];
        }
        const DwLayer &J = ud.J;
#ifdef SAC_STAMPS
        { unsigned long long probe = ud.w[0]; asm volatile("" :: "s"(probe)); }
        STAMP(4, 3);
#endif
        const int jj = jb - J.job0;
        const int n0 = 16 * (jj / J.nk), k0 = J.k_base + 64 * (jj % J.nk);
        // owner of tile t == wave: lane (c = r, g) holds rows n0+4g+i, col k0 + 16*wave + c
        const int k_own = k0 + 16 * wave + r;
        const bool own_valid = k_own < J.K;
        const size_t ot = frag_off(k_own, n0 + 4 * g, J.ldt);    // this lane's 16 B of the transposed copy / Adam moments
        // dY^T and X^T are fragment-major [feature][batch]: feature tile f, batch chunk q = 1 KB at ((f * B/16) + q) * 256
        const size_t tile_floats = (size_t)(B >> 4) * 256;
        const float *XT = (J.xt_from_slot ? S + J.xt_off : J.XT) + (size_t)(k0 >> 4) * tile_floats;
        const float *YT = J.dYT + (size_t)(n0 >> 4) * tile_floats;
        const int per = (B / 16) / 4;                 // 16-row chunks of the batch per wave
        const int rem = (B / 16) - 4 * per;
        const int s0 = wave * per + (wave < rem ? wave : rem);
        const int s1 = s0 + per + (wave < rem ? 1 : 0);
        const float *yp = YT + 4 * lane;              // this lane's 16 B of a fragment: (feature r, batch rows 4g..4g+3)
        const float *xp = XT + 4 * lane;
        f32x4 acc[4] = {};
        float bsum = 0.f;
        // first group of operand loads, then the owner's state, then the scalars: all in flight together.
        // Loads are unconditional (clamped chunk index, select on the loaded VALUE): a conditional load
        // becomes a branch whose merge point needs the data, i.e. a vmcnt(0) right behind the load.
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        auto chunk_of = [&](int sq) { const int m = (sq < s1 - 1) ? sq : s1 - 1; return m > 0 ? m : 0; };
        // k tiles of this strip that lie inside the layer (a first layer of 80 input columns has a second strip with ONE):
        // a padding tile re-reads the last valid one -- lines that are in flight anyway -- instead of streaming 16 KB of
        // zero (or foreign) rows; its products are never used (tile_ok below)
        int nv = (J.ldp - k0) >> 4;
        nv = nv > 4 ? 4 : (nv < 1 ? 1 : nv);
        size_t toff[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) toff[t] = (size_t)(t < nv ? t : nv - 1) * tile_floats;
        f32x4 a[4], b[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cq = chunk_of(s0 + u);
            a[u] = ld4(yp + 256 * cq);
#pragma unroll
            for (int t = 0; t < 4; ++t) b[u][t] = ld4(xp + toff[t] + 256 * cq);
        }
        f32x4 p4 = {0.f, 0.f, 0.f, 0.f}, m4 = p4, v4 = p4;
        if (own_valid) {
            p4 = ld4(J.PT + ot);
            m4 = ld4(J.MT + ot);
            v4 = ld4(J.VT + ot);
        }
        const double bc1 = sa.bc1, bc2sd = sa.bc2s;
        const bool polyak = (J.TP != nullptr) && (sa.step_now % d.period == 0);
        // The forward copy P [n][k] (and the Polyak target, same layout) holds this wave's 16 x 16 tile as ONE contiguous
        // 1-KB block, lane (c', g') = (n0 + c', k 4g'..4g'+3): written below with one 16-B store per lane after a
        // transpose through LDS (four 4-B stores + four 4-B target loads per lane before: the launch ends when its stores
        // have drained).
        const bool tile_ok = (k0 + 16 * wave) < J.ldp;                     // (wave-uniform: the last k tile of a 48-wide layer)
        const size_t pblk = frag_off(n0, k0 + 16 * wave, J.ldp) + 4 * lane;
        f32x4 tp4 = {0.f, 0.f, 0.f, 0.f};
        if (polyak && tile_ok) tp4 = ld4(J.TP + pblk);
        float pb = 0.f, mbv = 0.f, vbv = 0.f, tbv = 0.f;
        const bool bias_lane = (k0 == 0) && threadIdx.x < 16 && (n0 + (int)threadIdx.x) < J.N;
        if (bias_lane) {
            const int n = n0 + threadIdx.x;
            pb = ld1g(J.bias + n); mbv = ld1g(J.mb + n); vbv = ld1g(J.vb + n);
            if (polyak) tbv = ld1g(J.Tbias + n);
        }
#ifdef SAC_STAMPS
        { float probe = a[0][0] + b[3][3][3]; asm volatile("" :: "v"(probe)); }
        STAMP(4, 4);
#endif
        for (int sI = s0; sI < s1; sI += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 au = (sI + u < s1) ? a[u] : zero4;
                bsum += (au[0] + au[1]) + (au[2] + au[3]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(au[i], b[u][t][i], acc[t], 0, 0, 0);
                }
            }
            if (sI + 4 < s1) {                       // batches above 256 rows: next group
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int cq = chunk_of(sI + 4 + u);
                    a[u] = ld4(yp + 256 * cq);
#pragma unroll
                    for (int t = 0; t < 4; ++t) b[u][t] = ld4(xp + toff[t] + 256 * cq);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) st4(red + ((wave * 4 + t) * 64 + lane) * 4, acc[t]);
        bsum += __shfl_xor(bsum, 16);
        bsum += __shfl_xor(bsum, 32);
        if (g == 0) redb[wave * 16 + r] = bsum;
        lds_barrier();
        STAMP(4, 1);
        const float step_size = (float)((double)J.lr / bc1), bc2s = (float)bc2sd;
        if (aborted) return;
        if (tile_ok) {
            f32x4 gsum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) gsum += ld4(red + ((w * 4 + wave) * 64 + lane) * 4);
            if (own_valid) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int n = n0 + 4 * g + i;
                    if (n < J.N) {
                        float p = p4[i], m = m4[i], v = v4[i];
                        adam_update(p, m, v, gsum[i], step_size, bc2s);
                        p4[i] = p; m4[i] = m; v4[i] = v;
                        if (J.G && keep_grad) st1g(J.G + frag_off(n, k_own, J.ldp), gsum[i]);
                    }
                }
                // 16-B stores of the launch's ~4 MB of new state go out write-through (sc1): plain stores would sit dirty
                // in the L2s until the end-of-kernel write-back, in front of the next launch
                st4_sc1(J.PT + ot, p4);
                st4_sc1(J.MT + ot, m4);
                st4_sc1(J.VT + ot, v4);
            }
            // (lanes outside the layer hold zeros: p4 was never loaded or never updated, and padding is zero by invariant)
            float *tw = trs + wave * 256;
#pragma unroll
            for (int i = 0; i < 4; ++i) tw[(4 * g + i) * 16 + r] = p4[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const f32x4 pf = ld4(tw + r * 16 + 4 * g);                      // this lane as (c', g') of the forward copy
            st4_sc1(J.P + pblk, pf);
            if (polyak) st4_sc1(J.TP + pblk, tp4 * (1.0f - d.tau) + pf * d.tau);
        }
        if (bias_lane) {
            const int n = n0 + threadIdx.x;
            const float gb = (redb[threadIdx.x] + redb[16 + threadIdx.x]) + (redb[32 + threadIdx.x] + redb[48 + threadIdx.x]);
            adam_update(pb, mbv, vbv, gb, step_size, bc2s);
            st1g(J.bias + n, pb); st1g(J.mb + n, mbv); st1g(J.vb + n, vbv);
            if (J.gb && keep_grad) st1g(J.gb + n, gb);
            if (polyak) st1g(J.Tbias + n, tbv * (1.0f - d.tau) + pb * d.tau);
        }
        STAMP(4 - 2 * (sa.loop_pos & 1), 2);
    } else if (aborted) {
        // tell the host which launch was the first one not applied (diagnostic slots 30 / 31 are unused)
        if (threadIdx.x == 0 && d.diag_last[31] == 0.f) { d.diag_last[30] = __builtin_bit_cast(float, sa.seq); d.diag_last[31] = 1.f; }
    } else if (d.algo == 1) {
        td3_diagnostics(d, sa, red);
    } else {
        // ---- diagnostics block (SURVEY Appendix A line 17): one pass, wave-shuffle reductions ----
        STAMP(3, 0);
        // This ONE block must not outlast the ~245 tile-owner blocks of the launch (it did: 7.6 us against 4.9): every
        // load of a pass is requested up front -- thread = batch row, its 16-float rows of mu / log_std as four 16-B loads
        // each, unconditional with a clamped index -- and only then the entropy coefficient (scalar loads + expf) and the
        // accumulation run, so the block pays one memory round trip instead of one per loop iteration.
        const int loop_pos = sa.loop_pos;
        const int Bt = d.Bt;                 // statistics and losses run over the true batch (B is the padded row stride)
        const int i_first = (threadIdx.x < Bt) ? (int)threadIdx.x : 0;
        float yv0 = d.y[i_first], q10 = d.q[i_first], q20 = d.q[(size_t)B + i_first], lp0 = d.logpi[i_first];
        float qa0 = d.q[2 * (size_t)B + i_first], qb0 = d.q[3 * (size_t)B + i_first];
        f32x4 mu0[4], ls0[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { mu0[j] = ld4(d.mu + (size_t)i_first * 16 + 4 * j); ls0[j] = ld4(d.ls + (size_t)i_first * 16 + 4 * j); }
        SB();
        const AlphaStep as = alpha_step(cp, d.part_logpi, d.NB, d.Bt, d.target_entropy, d.alpha_lr, d.auto_alpha, sa.bc1, sa.bc2s);
        const float alpha = as.alpha, alpha_loss = as.alpha_loss;
#ifdef SAC_STAMPS
        { float probe = alpha; asm volatile("" ::"v"(probe)); STAMP(3, 2); }
        { float probe = ls0[3][3] + yv0; asm volatile("" ::"v"(probe)); STAMP(3, 3); }
#endif
        // Reduction: a row contributes 16 per-row quantities (below); they go to LDS as a [16][256] float panel, and 16
        // groups of 16 lanes reduce one quantity each in double -- 16 LDS reads and four 16-lane shuffle levels per lane.
        // (A full-wave butterfly over 28 per-thread partials, most of them doubles, cost 264 ds_bpermute per lane: 3.6 us.)
        //   0 q1  1 q2  2 y  3 log_pi (each: sum, sum of squares, max, min)   4 (q1-y)^2  5 (q2-y)^2  6 log_pi - q_new
        //   7 alpha log_pi - q_new (sums)   8 sum mu  9 sum mu^2  10 max mu  11 min mu  12-15 the same for log_std
        float *panel = red;                                   // [16][256]
        const int qd = threadIdx.x >> 4, pl = threadIdx.x & 15;   // this lane's quantity / its share of the rows
        double r_s = 0, r_s2 = 0;
        float r_mx = -INFINITY, r_mn = INFINITY;
        for (int i0 = 0; i0 < Bt; i0 += 256) {
            const int i = i0 + threadIdx.x;
            {
                const float e1 = q10 - yv0, e2 = q20 - yv0;
                const float qn = fminf(qa0, qb0);
                // the row's <= 16 values of mu / log_std: sums in float (seven to sixteen terms per row)
                float s_mu = 0.f, s_mu2 = 0.f, s_ls = 0.f, s_ls2 = 0.f, mx_mu = -INFINITY, mn_mu = INFINITY, mx_ls = -INFINITY, mn_ls = INFINITY;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (4 * j + u < d.A) {
                            const float m_ = mu0[j][u], l_ = ls0[j][u];
                            s_mu += m_; s_mu2 += m_ * m_; mx_mu = fmaxf(mx_mu, m_); mn_mu = fminf(mn_mu, m_);
                            s_ls += l_; s_ls2 += l_ * l_; mx_ls = fmaxf(mx_ls, l_); mn_ls = fminf(mn_ls, l_);
                        }
                const float vals[16] = {q10, q20, yv0, lp0, e1 * e1, e2 * e2, lp0 - qn, alpha * lp0 - qn,
                                        s_mu, s_mu2, mx_mu, mn_mu, s_ls, s_ls2, mx_ls, mn_ls};
#pragma unroll
                for (int q = 0; q < 16; ++q) panel[q * 256 + threadIdx.x] = vals[q];
            }
            lds_barrier();
            const int nrow = (Bt - i0 < 256) ? Bt - i0 : 256;
            for (int rr = pl; rr < nrow; rr += 16) {
                const float v = panel[qd * 256 + rr];
                r_s += (double)v;
                r_s2 += (double)v * (double)v;              // (used for quantities 0-3 only)
                r_mx = fmaxf(r_mx, v); r_mn = fminf(r_mn, v);
            }
            lds_barrier();                                    // panel free for the next pass
            if (i0 + 256 < Bt) {             // batches above 256 rows: the next pass's row
                const int n = (i + 256 < Bt) ? i + 256 : 0;
                yv0 = d.y[n]; q10 = d.q[n]; q20 = d.q[(size_t)B + n]; lp0 = d.logpi[n];
                qa0 = d.q[2 * (size_t)B + n]; qb0 = d.q[3 * (size_t)B + n];
#pragma unroll
                for (int j = 0; j < 4; ++j) { mu0[j] = ld4(d.mu + (size_t)n * 16 + 4 * j); ls0[j] = ld4(d.ls + (size_t)n * 16 + 4 * j); }
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            r_s += __shfl_xor(r_s, o); r_s2 += __shfl_xor(r_s2, o);
            r_mx = fmaxf(r_mx, __shfl_xor(r_mx, o)); r_mn = fminf(r_mn, __shfl_xor(r_mn, o));
        }
#ifdef SAC_STAMPS
        { float probe = (float)r_s; asm volatile("" ::"v"(probe)); STAMP(3, 4); }
#endif
        double *tot = reinterpret_cast<double *>(redb);      // [16 quantities] x {sum, sum of squares, max, min}: 512 B
        if (pl == 0) { tot[qd * 4 + 0] = r_s; tot[qd * 4 + 1] = r_s2; tot[qd * 4 + 2] = r_mx; tot[qd * 4 + 3] = r_mn; }
        lds_barrier();
        STAMP(3, 5);
        if (threadIdx.x < NSTAT) {
            const int q = threadIdx.x;               // q1, q2, q_target, log_pi | mu, log_std
            double s, s2, MX, MN;
            if (q < 4) { s = tot[q * 4]; s2 = tot[q * 4 + 1]; MX = tot[q * 4 + 2]; MN = tot[q * 4 + 3]; }
            else { const int b0 = 8 + 4 * (q - 4); s = tot[b0 * 4]; s2 = tot[(b0 + 1) * 4]; MX = tot[(b0 + 2) * 4 + 2]; MN = tot[(b0 + 3) * 4 + 3]; }
            const double cnt = (q < 4) ? (double)Bt : (double)Bt * d.A;
            const double mean = s / cnt;
            double var = s2 / cnt - mean * mean;
            if (var < 0) var = 0;
            const float o4[4] = {(float)mean, (float)sqrt(var), (float)MX, (float)MN};
            for (int u = 0; u < 4; ++u) {
                const int di = SAC_D_Q1_MEAN + 4 * q + u;
                dlast[di] = o4[u];
                if (loop_pos == 0) d.diag_first[di] = o4[u];
                if (loop_pos < DIAG_TRACE_CAP) d.diag_trace[(size_t)loop_pos * SAC_DIAG_N + di] = o4[u];
            }
        } else if (threadIdx.x >= 64 && threadIdx.x < 64 + 8) {
            const int q = threadIdx.x - 64;        // 0..3 losses, 4 alpha, 5 alpha loss, 6/7 unused
            float v = 0.f;
            int di = SAC_D_QF1_LOSS + q;
            if (q < 4) {
                v = (float)(tot[(4 + q) * 4] / Bt);
            } else if (q == 4) {
                v = alpha; di = SAC_D_ALPHA;
                Ctl *cw = d.ctl;                   // the step's only writer of the entropy-coefficient state
                cw->log_alpha = as.log_alpha; cw->a_m = as.m; cw->a_v = as.v; cw->alpha = as.alpha;
                cw->alpha_loss = as.alpha_loss;
            }
            else if (q == 5) { v = alpha_loss; di = SAC_D_ALPHA_LOSS; }
            else { di = 30 + (q - 6); }
            dlast[di] = v;
            if (loop_pos == 0) d.diag_first[di] = v;
            if (loop_pos < DIAG_TRACE_CAP) d.diag_trace[(size_t)loop_pos * SAC_DIAG_N + di] = v;
        }
        STAMP(3, 1);
    }
}

__global__ __launch_bounds__(256) void k_dw_adam(Dev d, DwTable T, const float *__restrict__ S, StepArg sa) {
    kernarg_prefetch<sizeof(Dev) + sizeof(DwTable) + 8 + sizeof(StepArg)>();
    __shared__ __attribute__((aligned(16))) float red[4 * 4 * 64 * 4];   // 16 KB (also diag scratch)
    __shared__ __attribute__((aligned(16))) float redb[4 * 16 * 2];
    __shared__ __attribute__((aligned(16))) float trs[4 * 256];           // a wave's 16 x 16 tile on its way to the forward copy
    // fused step: the forward/backward launch gave up (a hand-off wait timed out) => this launch, the step's only
    // writer of weights, Adam state, targets and the entropy coefficient, applies NOTHING
    const unsigned aborted = T.abort ? sload(T.abort) : 0u;
    dw_adam_body(d, T, S, sa, red, redb, trs, (int)blockIdx.x, aborted);
}

#include "sac_bwd8.h"
#include "sac_fused.h"
#include "sac_chain.h"
#include "sac_general.h"

}  // namespace sac

// ==========================================================================================
// host side
// ==========================================================================================
using namespace sac;

struct sac_general;

struct sac_trainer {
    sac_config_t cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    int B = 0, Bt = 0, O = 0, A = 0, KP = 0, KQ = 0, NH = 0, NB = 0, SP = 4;   // B: batch padded to row-blocks, Bt: true batch
    Net net[6];                                       // 5: TD3 target policy
    int HP[2] = {256, 256}, HQ[2] = {256, 256};       // logical hidden sizes (<= 256: zero-padded to the kernels' 256)
    int algo = 0;                                     // 0 SAC, 1 TD3
    int td3_period = 2;                               // policy_and_target_update_period
    long long adam_t_pi = 0;                          // TD3: optimizer steps of the policy (delayed)
    DwTable dw_q{}, dw_q_tp{}, dw_pi{}, dw_none{};    // TD3 work tables: critics (without / with Polyak), policy, diagnostics only
    void (*fwd_b2)(Dev, const float *, SlotLayout, StepArg) = nullptr;      // TD3 actor pass
    void (*bwd2)(Dev, const float *, SlotLayout, StepArg, int) = nullptr;
    Dev dev{};
    DwTable dw{};
    DwLayer *d_dwl = nullptr;
    char *arena = nullptr;                            // the one device allocation everything lives in
    float *ws = nullptr; int64_t ws_floats = 0;      // all activations / gradients
    float *ext_slot = nullptr; SlotLayout ext_layout{};
    float *d_eps = nullptr;                           // [2][B*A]
    float *d_diag = nullptr;                          // first[32] last[32] trace[CAP][32]
    Ctl *d_ctl = nullptr;
    void *h_stage = nullptr; size_t stage_bytes = 0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    static constexpr int NLOOP_EV = 12;
    hipEvent_t ev_ready[NLOOP_EV] = {}, ev_done[NLOOP_EV] = {};                      // chunks of sac_train_loop in flight
    float *h_diag = nullptr, *d_diag_host = nullptr;  // mapped pinned: first[32] | last[32] (host view, device view)
    float last_ms[4] = {0, 0, 0, 0};
    bool loop_primed = false;
    bool gate_exempt = false;                         // fused on CUs of its own (sac_trainer_set_xcd_mask): not serialised with other trainers
    bool timing_pending = false;                      // the last loop's event intervals have not been read yet (read lazily)
    hipEvent_t ev_tm[3] = {nullptr, nullptr, nullptr};  // around the draw and the gather of the loop's timed chunk
    std::vector<float> h_policy;                      // host mirror for acting
    bool mirror_valid = false;
    size_t lds_bw = 0;
    // fused step (k_abc, sac_fused.h): launches A + B + C as one launch with in-launch hand-offs
    bool fused = false;
    unsigned fused_seq = 0;                           // launches so far: the hand-off counters count in units of it
    unsigned fused_unchecked = 0;                     // fused launches since the host last looked at the abort marker
    bool publish_diag = true;                         // the next step's diagnostics go to the pinned host buffer (its caller reads them)
    unsigned test_stall_at = 0;                       // SAC_FUSED_TEST_STALL=<n>: the n-th fused launch loses a producer (tests)
    // Fused steps launched by sac_step_device that the host has not yet seen applied: should one of them give up, these
    // are re-run on the four-launch step from their slots (still intact: the host never runs more than two groups of
    // sixteen steps ahead of the device on this path -- thr_ev, one event per sixteen steps).
    static constexpr int NPEND = 64;
    struct Pending { sac_buffer *buf; int64_t token; };
    Pending pend[NPEND];
    int pend_n = 0;                                   // entries pend[(pend_head + i) % NPEND], i < pend_n, oldest first
    int pend_head = 0;
    long long dev_steps = 0;                          // fused device-batch steps launched so far (sixteen per throttle event)
    hipEvent_t thr_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int fallbacks = 0;                                // times the fused step gave up (at most once: the fall-back is for good)
    void (*abc)(Dev, const float *, SlotLayout, StepArg) = nullptr;
    size_t lds_abc = 0;
    int abc_grid = 0, abc_threads = 256;              // the fused launch's shape (k_abc: 16 NB x 256; k_chain8<.., BWD>: 4 NB x 512)
    bool chain_bwd = false;                           // batch 1024: the fused launch is k_chain8 with the backward blocks inside
    unsigned *d_sync = nullptr; size_t sync_bytes = 0;   // counters (one per 128-B line) + abort word
    void (*fwd_a)(Dev, const float *, SlotLayout, int) = nullptr;
    void (*fwd_b)(Dev, const float *, SlotLayout, StepArg) = nullptr;
    void (*bwd)(Dev, const float *, SlotLayout, StepArg, int) = nullptr;
    size_t lds_fa = 0, lds_fb = 0;
    // column split 1 (batch >= 1024): launches A + B as one launch without any hand-off (k_chain, sac_chain.h)
    bool chain = false, chain8 = false;               // chain8: the eight-wave variant of k_chain (512 threads per workgroup)
    bool bwd8 = false;                                // the backward launch at column split 1 on eight waves (k_bwd8, sac_bwd8.h)
    void (*chaink)(Dev, const float *, SlotLayout, StepArg) = nullptr;
    size_t lds_chain = 0;
    long long n_train_steps_total = 0, adam_t = 0;   // host-side step counters (rlkit _n_train_steps_total)
    // hidden_sizes beyond two layers of <= 256 units: the general step (sac_general.h); everything above that belongs to
    // the fused kernels stays unused
    sac_general *gen = nullptr;
};

#include "sac_general_host.h"

namespace {

void build_layers(Net &n, const int (*shape)[2], int nl) {
    long long off = 0, offt = 0;
    for (int l = 0; l < nl; ++l) {
        Layer &L = n.L[l];
        L.N = shape[l][0]; L.K = shape[l][1];
        L.Np = round_up(L.N, 16); L.Kp = round_up(L.K, 16);
        L.offW = off; off += (long long)L.Np * L.Kp;
        L.offB = off; off += L.Np;
        off = round_up64(off, 4);
        L.offWt = offt; offt += (long long)(L.Kp + 16) * L.Np;
    }
    n.nP = off; n.nPT = offt;
}

// One arena for every device buffer of a trainer (weights, Adam state, activations, control words):
// a single 2-MiB-aligned allocation keeps the whole working set (a few MB) on a handful of large
// pages, so a kernel's first touches do not each pay an address-translation miss.
struct Arena {
    std::vector<std::pair<void **, size_t>> req;
    size_t total = 0;
    void reserve(void **p, size_t bytes) {
        req.emplace_back(p, total);
        total += (bytes + 255) & ~(size_t)255;
    }
};
thread_local Arena *g_arena = nullptr;      // only during sac_trainer_create

int alloc_zero(float **p, long long n, hipStream_t) {
    g_arena->reserve(reinterpret_cast<void **>(p), sizeof(float) * (size_t)n);
    return 0;
}

int arena_commit(Arena &a, char **base, hipStream_t s) {
    const size_t bytes = (a.total + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    SAC_HIP(hipMalloc(reinterpret_cast<void **>(base), bytes));
    SAC_HIP(hipMemsetAsync(*base, 0, bytes, s));
    for (auto &r : a.req) *r.first = *base + r.second;
    return 0;
}

// flat nn.Linear layout <-> padded device layout (host vectors)
struct FlatMap { int nl; int N[4], K[4]; };   // logical layers in the flat vector

FlatMap flat_map(const sac_trainer *t, int netid) {
    FlatMap f{};
    if (netid == SAC_NET_POLICY || netid == 5) {
        f.nl = t->algo == 1 ? 3 : 4;                        // TD3: TanhMlpPolicy (one head); SAC: mean + log-std heads
        f.N[0] = t->HP[0]; f.K[0] = t->O; f.N[1] = t->HP[1]; f.K[1] = t->HP[0];
        f.N[2] = t->A; f.K[2] = t->HP[1]; f.N[3] = t->A; f.K[3] = t->HP[1];
    } else {
        f.nl = 3;
        f.N[0] = t->HQ[0]; f.K[0] = t->O + t->A; f.N[1] = t->HQ[1]; f.K[1] = t->HQ[0]; f.N[2] = 1; f.K[2] = t->HQ[1];
    }
    return f;
}

int64_t flat_count(const FlatMap &f) {
    int64_t n = 0;
    for (int l = 0; l < f.nl; ++l) n += (int64_t)f.N[l] * f.K[l] + f.N[l];
    return n;
}

// iterate (flat index) -> (device layer, n, k | bias)
template <typename F>
void for_each_param(const sac_trainer *t, int netid, F &&fn) {
    const FlatMap f = flat_map(t, netid);
    int64_t fi = 0;
    for (int l = 0; l < f.nl; ++l) {
        const int dl = (l < 2) ? l : 2;                       // policy heads share device layer 2
        const int nshift = (netid == SAC_NET_POLICY && l == 3) ? t->A : 0;
        const bool qin = (netid >= 1 && netid <= 4) && l == 0;   // Q first layer: action columns start at KP
        for (int n = 0; n < f.N[l]; ++n)
            for (int k = 0; k < f.K[l]; ++k) fn(fi++, dl, n + nshift, (qin && k >= t->O) ? t->KP + (k - t->O) : k, false);
        for (int n = 0; n < f.N[l]; ++n) fn(fi++, dl, n + nshift, 0, true);
    }
}

int ensure_stage_t(sac_trainer *t, size_t bytes) {
    if (t->stage_bytes >= bytes) return 0;
    if (t->h_stage) SAC_HIP(hipHostFree(t->h_stage));
    t->h_stage = nullptr; t->stage_bytes = 0;
    SAC_HIP(hipHostMalloc(&t->h_stage, bytes, hipHostMallocDefault));
    t->stage_bytes = bytes;
    return 0;
}

// TD3 step (rlkit TD3Trainer.train_from_torch): critic pass = 4 launches every step; on policy steps
// (n_train_steps_total % policy_and_target_update_period == 0) the critics' Adam launch also soft-updates their
// targets and the actor pass follows -- Q1(s, policy(s)) through the ALREADY UPDATED qf1, policy backward, policy
// Adam + soft update of the target policy.  want_stats: also produce Policy Loss / Policy Action on a non-policy
// step (rlkit recomputes them for the epoch statistics), without any update.
// Fused launches of different trainers (streams) of one process must not overlap on a device: two half-resident grids
// would wait for each other's CUs until the hand-off timeout.  While more than one fused trainer lives on a device,
// each fused launch waits for the previous one there and records an event behind itself (host section under a lock).
struct FusedGate { std::mutex mu; hipEvent_t ev = nullptr; hipStream_t last = nullptr; int live = 0; };
FusedGate g_gate[64];

// the fused forward/backward launch of a step (k_abc: SAC, or the TD3 critic pass); sa gets the launch number
int launch_fused_abc(sac_trainer *t, const float *S, const SlotLayout &SL, StepArg &sa, unsigned extra_flags) {
    hipStream_t s = t->stream;
    sa.seq = ++t->fused_seq;
    sa.pad2 |= extra_flags | ((t->test_stall_at && sa.seq == t->test_stall_at) ? 1u : 0u);
    t->fused_unchecked += 1;
    FusedGate &G = g_gate[t->device & 63];
    std::lock_guard<std::mutex> lk(G.mu);
    const bool gate = G.live > 1 && !t->gate_exempt;
    if (gate && G.last && G.last != s) SAC_HIP(hipStreamWaitEvent(s, G.ev, 0));
    hipLaunchKernelGGL(t->abc, dim3(t->abc_grid), dim3(t->abc_threads), t->lds_abc, s, t->dev, S, SL, sa);
    if (gate) { SAC_HIP(hipEventRecord(G.ev, s)); G.last = s; }
    return 0;
}

int launch_step_td3(sac_trainer *t, const float *S, const SlotLayout &SL, int j, bool want_stats) {
    const Dev &d = t->dev;
    hipStream_t s = t->stream;
    const int NB = t->NB, SPv = t->SP;
    const bool pstep = (t->n_train_steps_total % t->td3_period) == 0;
    const bool actor = pstep || want_stats;
    const double tq = (double)(t->adam_t + 1), tp = (double)(t->adam_t_pi + 1);
    StepArg sq{t->n_train_steps_total, t->adam_t + 1, j, 1, 1.0 - std::pow(0.9, tq), std::sqrt(1.0 - std::pow(0.999, tq))};
    StepArg sp{t->n_train_steps_total, t->adam_t_pi + 1, j, 2, 1.0 - std::pow(0.9, tp), std::sqrt(1.0 - std::pow(0.999, tp))};
    // (the diagnostics -- and the flat gradient copies of sac_debug_fetch -- go out only on the steps whose caller reads
    //  them, like the SAC step's; td3_diagnostics keeps "last" the most recent value of each entry across launches)
    sq.pad2 = sp.pad2 = t->publish_diag ? 2u : 0u;
    if (t->fused) {
        // the critic pass as ONE launch (k_abc<.., M_TD3_CRITIC>, sac_fused.h) + its weight-gradient launch
        if (launch_fused_abc(t, S, SL, sq, actor ? 4u : 0u)) return -1;
    } else {
        hipLaunchKernelGGL(t->fwd_a, dim3(4 * SPv * NB), dim3(256), t->lds_fa, s, d, S, SL, actor ? 1 : 0);
        const unsigned g2 = 8u * (unsigned)((SPv * NB + 3) / 4);      // two twins x groups of four blocks (k_fwd_b / k_bwd, TD3 critic map)
        hipLaunchKernelGGL(t->fwd_b, dim3(g2), dim3(256), t->lds_fb, s, d, S, SL, sq);
        hipLaunchKernelGGL(t->bwd, dim3(g2), dim3(256), t->lds_bw, s, d, S, SL, sq, 0);
    }
    const DwTable &Tq = pstep ? t->dw_q_tp : t->dw_q;
    hipLaunchKernelGGL(k_dw_adam, dim3(Tq.njobs + 1), dim3(256), 0, s, d, Tq, S, sq);
    if (actor) {
        hipLaunchKernelGGL(t->fwd_b2, dim3(SPv * NB), dim3(256), t->lds_fb, s, d, S, SL, sp);
        if (pstep) {
            hipLaunchKernelGGL(t->bwd2, dim3(SPv * NB), dim3(256), t->lds_bw, s, d, S, SL, sp, 0);
            hipLaunchKernelGGL(k_dw_adam, dim3(t->dw_pi.njobs + 1), dim3(256), 0, s, d, t->dw_pi, S, sp);
            t->adam_t_pi += 1;
        } else {
            hipLaunchKernelGGL(k_dw_adam, dim3(1), dim3(256), 0, s, d, t->dw_none, S, sp);      // statistics only
        }
    }
    SAC_HIP(hipGetLastError());
    t->n_train_steps_total += 1;
    t->adam_t += 1;
    return 0;
}

// the four launches of step j of the current chunk, on minibatch slot S; ev != null => HIP events
// between the launches (profiling pass only)

int launch_step(sac_trainer *t, const float *S, const SlotLayout &SL, int j, hipEvent_t *ev = nullptr, bool want_stats = false) {
    if (t->gen) return gen_launch_step(t, S, SL, j, want_stats);
    if (t->algo == 1) return launch_step_td3(t, S, SL, j, want_stats);
    const Dev &d = t->dev;
    hipStream_t s = t->stream;
    const int NB = t->NB;
    const double tt = (double)(t->adam_t + 1);
    StepArg sa{t->n_train_steps_total, t->adam_t + 1, j, 0, 1.0 - std::pow(0.9, tt), std::sqrt(1.0 - std::pow(0.999, tt))};
    sa.pad2 = t->publish_diag ? 2u : 0u;
    const int SPv = t->SP;
    if (ev) SAC_HIP(hipEventRecord(ev[0], s));
    if (t->fused) {
        // two launches: A + B + C as k_abc (in-launch hand-offs), then the weight-gradient / Adam launch
        if (launch_fused_abc(t, S, SL, sa, 0u)) return -1;
        if (ev) { SAC_HIP(hipEventRecord(ev[1], s)); SAC_HIP(hipEventRecord(ev[2], s)); SAC_HIP(hipEventRecord(ev[3], s)); }
    } else {
        if (t->chain) {
            hipLaunchKernelGGL(t->chaink, dim3(4 * NB), dim3(t->chain8 ? 512 : 256), t->lds_chain, s, d, S, SL, sa);
            if (ev) { SAC_HIP(hipEventRecord(ev[1], s)); SAC_HIP(hipEventRecord(ev[2], s)); }
        } else {
            hipLaunchKernelGGL(t->fwd_a, dim3(4 * SPv * NB), dim3(256), t->lds_fa, s, d, S, SL, 0);
            if (ev) SAC_HIP(hipEventRecord(ev[1], s));
            hipLaunchKernelGGL(t->fwd_b, dim3(4 * SPv * NB), dim3(256), t->lds_fb, s, d, S, SL, sa);
            if (ev) SAC_HIP(hipEventRecord(ev[2], s));
        }
        const int compact = (3 * SPv * NB <= 192) ? 1 : 0;     // see k_bwd
        hipLaunchKernelGGL(t->bwd, dim3(compact ? 4 * SPv * NB : 3 * SPv * NB), dim3(t->bwd8 ? 512 : 256), t->lds_bw, s, d, S, SL, sa, compact);
        if (ev) SAC_HIP(hipEventRecord(ev[3], s));
    }
    if (ev) SAC_HIP(hipEventRecord(ev[4], s));
    hipLaunchKernelGGL(k_dw_adam, dim3(t->dw.njobs + 1), dim3(256), 0, s, d, t->dw, S, sa);
    if (ev) { SAC_HIP(hipEventRecord(ev[5], s)); SAC_HIP(hipEventRecord(ev[6], s)); }
    SAC_HIP(hipGetLastError());
    t->n_train_steps_total += 1;
    t->adam_t += 1;
    return 0;
}

// Wait until everything launched on the trainer's stream so far has finished: a user-space poll (up to 2 ms) of an event
// recorded behind it, then the blocking wait.  hipStreamSynchronize costs ~16 us even on a stream that is already idle
// and its wake-up 10-20 us on one that is not -- a fifth of a single step, 2.5 % of a 20-step loop call.
static int wait_trainer_stream(sac_trainer *t, hipEvent_t recorded = nullptr) {
    hipEvent_t e = recorded;
    if (!e) { e = t->ev[3]; SAC_HIP(hipEventRecord(e, t->stream)); }
    const auto spin_until = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
    hipError_t st;
    while ((st = hipEventQuery(e)) == hipErrorNotReady && std::chrono::steady_clock::now() < spin_until) { }
    if (st == hipSuccess) return 0;
    SAC_HIP(hipStreamSynchronize(t->stream));
    return 0;
}

// every weight-gradient table of the trainer looks at (or stops looking at) the fused step's give-up word
static void set_abort_ptrs(sac_trainer *t, const unsigned *p) {
    t->dw.abort = p; t->dw_q.abort = p; t->dw_q_tp.abort = p; t->dw_pi.abort = p; t->dw_none.abort = p;
}

// wait for ONE event (not for the stream behind it): a short user-space poll, then the blocking wait
static int wait_event(hipEvent_t e) {
    const auto spin_until = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
    hipError_t st;
    while ((st = hipEventQuery(e)) == hipErrorNotReady && std::chrono::steady_clock::now() < spin_until) { }
    if (st == hipSuccess) return 0;
    if (st == hipErrorNotReady) (void)hipGetLastError();
    SAC_HIP(hipEventSynchronize(e));
    return 0;
}

// After the stream has drained: did a fused launch give up?  (Launch D of the first such step left the launch number in
// the pinned diagnostics, applied nothing, and so did every step behind it.)  Roll the host counters back to the
// applied steps, fall back to the four-launch step for good, and report.
// Returns 0 (nothing happened), 1 (the fused step gave up: *lost_out steps at the end of what was launched were not
// applied, the counters are back at the applied ones, the trainer is on the four-launch step now -- the CALLER re-runs
// the lost steps), <0 on a HIP error.
int check_fused_abort(sac_trainer *t, unsigned *lost_out = nullptr) {
    const unsigned launched = t->fused_unchecked;
    t->fused_unchecked = 0;
    if (lost_out) *lost_out = 0;
    if (!t->fused || t->h_diag[SAC_DIAG_N + 31] == 0.f) { t->pend_n = 0; return 0; }
    unsigned first_bad = 0;
    memcpy(&first_bad, &t->h_diag[SAC_DIAG_N + 30], sizeof(unsigned));
    unsigned lost = t->fused_seq - first_bad + 1;
    if (lost > launched) lost = launched;
    t->n_train_steps_total -= lost;
    t->adam_t -= lost;
    t->fused = false;
    set_abort_ptrs(t, nullptr);
    if (t->algo == 1) {          // TD3: the policy's optimizer steps among the lost steps (every td3_period-th step number)
        for (long long k = t->n_train_steps_total; k < t->n_train_steps_total + (long long)lost; ++k)
            if (k % t->td3_period == 0) t->adam_t_pi -= 1;
    }
    t->h_diag[SAC_DIAG_N + 30] = t->h_diag[SAC_DIAG_N + 31] = 0.f;
    SAC_HIP(hipMemsetAsync(t->d_sync, 0, t->sync_bytes, t->stream));
    SAC_HIP(hipStreamSynchronize(t->stream));
    if (!t->gate_exempt) {
        FusedGate &G = g_gate[t->device & 63];
        std::lock_guard<std::mutex> lk(G.mu);
        G.live -= 1;
    }
    t->gate_exempt = false;
    t->fallbacks += 1;
    if (lost_out) *lost_out = lost;
    fprintf(stderr, "[libsac_hip] warning: the fused SAC step gave up -- a hand-off between its workgroups timed out (is another "
                    "process or kernel using this GPU?).  This trainer continues on the four-launch step; the %u step(s) "
                    "that were not applied are re-run on it (SAC_FUSED=0 selects the four-launch step from the start).\n", lost);
    return 1;
}

// Re-run the last `n` device-batch steps that gave up, oldest first, on the step the trainer uses now (four launches).
// Their batches still sit in their slots: sac_step_device keeps the host within 47 steps of the device, a slot is
// reused 64 batches later and the read-ahead draws at most 16 batches beyond the last one handed out.
int replay_pending(sac_trainer *t, unsigned n) {
    SAC_REQUIRE((int)n <= t->pend_n, "internal: %u fused steps were lost but only %d are on record", n, t->pend_n);
    if (n == 0) { t->pend_n = 0; return 0; }
    sac_buffer *seen[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = t->pend_n - (int)n; i < t->pend_n; ++i) {
        const sac_trainer::Pending &P = t->pend[(t->pend_head + i) % sac_trainer::NPEND];
        const int slot = sac_ring_slot_of(P.buf, P.token);
        if (slot < 0) return -1;
        t->dev.eps1 = t->dev.eps2 = nullptr;
        t->publish_diag = (i == t->pend_n - 1);
        const int rc = launch_step(t, P.buf->d_ring + (size_t)slot * P.buf->ring_layout.slot_floats, P.buf->ring_layout, 0,
                                   nullptr, i == t->pend_n - 1);
        t->publish_diag = true;
        if (rc) return -1;
        for (auto &sb : seen) { if (sb == P.buf) break; if (!sb) { sb = P.buf; break; } }
    }
    // the slot-release events of those steps were recorded behind the launches that gave up: the buffers' streams now
    // wait for the re-runs themselves before they may overwrite a slot
    SAC_HIP(hipEventRecord(t->ev[2], t->stream));
    for (sac_buffer *sb : seen) if (sb) SAC_HIP(hipStreamWaitEvent(sb->stream, t->ev[2], 0));
    t->pend_n = 0;
    return 0;
}

// the trainer's stream has drained: if a fused step gave up, fall back and re-run what was lost (device-batch steps)
int recover_device_steps(sac_trainer *t) {
    unsigned lost = 0;
    const int fa = check_fused_abort(t, &lost);
    if (fa <= 0) return fa;
    if (replay_pending(t, lost)) return -1;
    return wait_trainer_stream(t);
}

// sample + gather all slots of a loop on the buffer's stream, make the trainer's stream wait
int stage_batches(sac_trainer *t, sac_buffer *b, int64_t n_steps) {
    if (readahead_rollback(b)) return -1;
    if (ensure_slots(b, t->Bt, n_steps)) return -1;
    SAC_HIP(hipEventRecord(b->ev[0], b->stream));
    if (launch_sample(b, t->Bt, n_steps)) return -1;
    SAC_HIP(hipEventRecord(b->ev[1], b->stream));
    if (launch_gather(b, b->d_idx, t->B, n_steps, b->d_slots, b->slot, 1)) return -1;
    SAC_HIP(hipEventRecord(b->ev[2], b->stream));
    SAC_HIP(hipStreamWaitEvent(t->stream, b->ev[2], 0));
    return 0;
}

}  // namespace

extern "C" {

static int trainer_create(sac_trainer_t **out, const sac_config_t *cfg, const td3_config_t *td3);

int sac_trainer_create(sac_trainer_t **out, const sac_config_t *cfg) {
    SAC_REQUIRE(out && cfg, "null argument to sac_trainer_create");
    return trainer_create(out, cfg, nullptr);
}

// TD3 (/root/reference/util/rlkit_utils.py:107-135, scripts/train.py:38-47): the same handle
// type and the same sac_* accessors; net id 5 is the target policy.
int td3_trainer_create(sac_trainer_t **out, const td3_config_t *c) {
    SAC_REQUIRE(out && c, "null argument to td3_trainer_create");
    SAC_REQUIRE(c->policy_and_target_update_period > 0, "policy_and_target_update_period must be positive");
    SAC_REQUIRE(c->target_policy_noise >= 0.f && c->target_policy_noise_clip >= 0.f, "negative target policy noise");
    sac_config_t s{};
    s.obs_dim = c->obs_dim; s.act_dim = c->act_dim; s.hidden = c->hidden; s.batch = c->batch;
    for (int i = 0; i < 2; ++i) { s.policy_hidden[i] = c->policy_hidden[i]; s.qf_hidden[i] = c->qf_hidden[i]; }
    s.discount = c->discount; s.reward_scale = c->reward_scale;
    s.policy_lr = c->policy_learning_rate; s.qf_lr = c->qf_learning_rate;
    s.soft_target_tau = c->tau; s.target_update_period = 1; s.use_automatic_entropy_tuning = 0;
    s.target_entropy = 0.f; s.noise_seed = c->noise_seed; s.device = c->device;
    return trainer_create(out, &s, c);
}

static int trainer_build(sac_trainer *t, const sac_config_t *cfg, const td3_config_t *td3);

static int trainer_create(sac_trainer_t **out, const sac_config_t *cfg, const td3_config_t *td3) {
    *out = nullptr;
    SAC_REQUIRE(sac_device_count() > 0, "no HIP device visible: libsac_hip has no CPU fallback");
    for (int i = 0; i < 2; ++i) {
        const int hp = cfg->policy_hidden[i] ? cfg->policy_hidden[i] : cfg->hidden, hq = cfg->qf_hidden[i] ? cfg->qf_hidden[i] : cfg->hidden;
        SAC_REQUIRE(hp >= 1 && hp <= H && hq >= 1 && hq <= H,
                    "hidden sizes policy %d / qf %d unsupported: two hidden layers of 1..256 units each (wider or deeper "
                    "networks do not fit the kernels' 256-wide layers)", hp, hq);
    }
    SAC_REQUIRE(cfg->obs_dim > 0 && cfg->act_dim > 0 && cfg->act_dim <= 16,
                "unsupported dims obs=%d act=%d (act_dim must be in 1..16)", cfg->obs_dim, cfg->act_dim);
    SAC_REQUIRE(cfg->obs_dim <= 496, "obs_dim %d unsupported: the row staging of the step kernels holds cat(obs, act) rows of at "
                "most 512 columns (obs_dim <= 496; the shipped tasks reach 379)", cfg->obs_dim);
    SAC_REQUIRE(cfg->batch > 0, "batch size %d must be positive", cfg->batch);
    SAC_REQUIRE(cfg->target_update_period > 0, "target_update_period must be positive");
    SAC_HIP(hipSetDevice(cfg->device));
    sac_trainer *t = new sac_trainer();
    if (trainer_build(t, cfg, td3)) {        // (the error message is already set)
        sac_trainer_destroy(t);
        return -1;
    }
    *out = t;
    return 0;
}

// what every kind of trainer has: stream, events, the pinned diagnostics
static int trainer_common_init(sac_trainer *t, const sac_config_t *cfg) {
    t->cfg = *cfg; t->device = cfg->device;
    t->Bt = cfg->batch; t->B = round_up(cfg->batch, RB); t->O = cfg->obs_dim; t->A = cfg->act_dim;
    SAC_HIP(hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking));
    sac::stream_register(t->stream);
    for (auto &e : t->thr_ev) SAC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : t->ev) SAC_HIP(hipEventCreate(&e));
    for (auto &e : t->ev_tm) SAC_HIP(hipEventCreate(&e));
    for (auto &e : t->ev_ready) SAC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : t->ev_done) SAC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    // first[32] | last[32] of the diagnostics live in MAPPED PINNED HOST memory: the one diagnostics workgroup of a step
    // writes its ~30 floats straight over the link, and a caller reads them after the stream has drained without a
    // device-to-host copy in its call (that copy was ~10 us of every sac_train_loop / sac_step)
    SAC_HIP(hipHostMalloc(reinterpret_cast<void **>(&t->h_diag), sizeof(float) * 2 * SAC_DIAG_N, hipHostMallocMapped));
    memset(t->h_diag, 0, sizeof(float) * 2 * SAC_DIAG_N);
    SAC_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&t->d_diag_host), t->h_diag, 0));
    return 0;
}

// every event is recorded once at creation: the runtime sets an event's signal up at its first record, which otherwise
// happens inside the first loop that is long enough to use it (a 20-step call behind a 5-step one: +40 us)
static int trainer_prime_events(sac_trainer *t) {
    hipStream_t s = t->stream;
    for (auto &e : t->ev) SAC_HIP(hipEventRecord(e, s));
    for (auto &e : t->ev_tm) SAC_HIP(hipEventRecord(e, s));
    for (auto &e : t->ev_ready) SAC_HIP(hipEventRecord(e, s));
    for (auto &e : t->ev_done) SAC_HIP(hipEventRecord(e, s));
    SAC_HIP(hipStreamSynchronize(s));
    return 0;
}

static int trainer_build(sac_trainer *t, const sac_config_t *cfg, const td3_config_t *td3) {
    if (trainer_common_init(t, cfg)) return -1;
    t->algo = td3 ? 1 : 0;
    for (int i = 0; i < 2; ++i) {
        t->HP[i] = cfg->policy_hidden[i] ? cfg->policy_hidden[i] : cfg->hidden;
        t->HQ[i] = cfg->qf_hidden[i] ? cfg->qf_hidden[i] : cfg->hidden;
    }
    if (td3) t->td3_period = td3->policy_and_target_update_period;
    t->KP = round_up(t->O, 16); t->KQ = t->KP + 16; t->NH = round_up((td3 ? 1 : 2) * t->A, 16);
    t->NB = t->B / 16;
    // column split: small batches spread every 256-wide layer over 4 workgroups per row-block; once the
    // row-blocks alone fill the 256 CUs (B >= 512) fewer, fatter workgroups win.  SP*NB stays even (XCD map).
    t->SP = (t->NB <= 16) ? 4 : (t->NB <= 32 ? 2 : ((t->NB & 1) ? 2 : 1));
    if (const char *e = getenv("SAC_FORCE_SP")) {       // tuning experiments only
        const int v = atoi(e);
        if ((v == 1 || v == 2 || v == 4) && ((v * t->NB) % 2 == 0)) t->SP = v;
    }
    hipStream_t s = t->stream;
    const int B = t->B;

    Arena arena;
    g_arena = &arena;
    const int shp[3][2] = {{H, t->O}, {H, H}, {(td3 ? 1 : 2) * t->A, H}};
    const int shq[3][2] = {{H, t->KQ}, {H, H}, {1, H}};     // device K of the Q first layer: padded [obs | act] layout
    const int nnets = td3 ? 6 : 5;
    for (int i = 0; i < nnets; ++i) {
        Net &n = t->net[i];
        build_layers(n, (i == 0 || i == 5) ? shp : shq, 3);
        if (alloc_zero(&n.P, n.nP, s)) return -1;
        if (i < 3) {
            if (alloc_zero(&n.M, n.nP, s) || alloc_zero(&n.V, n.nP, s) || alloc_zero(&n.PT, n.nPT, s) ||
                alloc_zero(&n.MT, n.nPT, s) || alloc_zero(&n.VT, n.nPT, s) || alloc_zero(&n.G, n.nP, s)) return -1;
        }
    }
    // workspace carve
    Dev &d = t->dev;
    std::vector<std::pair<float **, long long>> parts = {
        {&d.PH1T, (long long)H * B}, {&d.PH2T, (long long)H * B},
        {&d.mu, 16LL * B}, {&d.ls, 16LL * B}, {&d.lsok, 16LL * B}, {&d.z, 16LL * B}, {&d.anew, 16LL * B},
        {&d.epsv, 16LL * B}, {&d.logpi, B}, {&d.a2, 16LL * B}, {&d.logpi2, B}, {&d.part_logpi, round_up(t->NB, 64)},
        {&d.QH1T, 4LL * H * B}, {&d.QH2T, 4LL * H * B}, {&d.q, 6LL * B}, {&d.QU, 2LL * H * B},
        {&d.y, B}, {&d.dq16T, 2LL * 16 * B}, {&d.dQH2T, 2LL * H * B}, {&d.dQH1T, 2LL * H * B},
        {&d.headpart, 2LL * t->NB * 4 * RB * 32}, {&d.qpart, 6LL * 4 * B}, {&d.dapart, 2LL * 4 * B * 16},
        {&d.qpart2, 6LL * t->NB * 4 * 32}, {&d.logpi2p, 32LL * t->NB}, {&d.hv4, 64LL * B}, {&d.zx, 6LL * H * B},
        {&d.dheadT, (long long)t->NH * B}, {&d.dPH2T, (long long)H * B}, {&d.dPH1T, (long long)H * B}};
    long long tot = 0;
    for (auto &p : parts) tot += round_up64(p.second, 64);
    if (alloc_zero(&t->ws, tot, s)) return -1;
    t->ws_floats = tot;
    t->ext_layout = make_slot_layout(t->Bt, t->O, t->A);
    if (alloc_zero(&t->ext_slot, t->ext_layout.slot_floats, s)) return -1;
    if (alloc_zero(&t->d_eps, 2LL * B * t->A, s)) return -1;
    if (alloc_zero(&t->d_diag, (long long)SAC_DIAG_N * (2 + DIAG_TRACE_CAP), s)) return -1;
    arena.reserve(reinterpret_cast<void **>(&t->d_ctl), sizeof(Ctl));
    // head[2][NB], qa / tq / ac [NB], log-pi, abort; then zx[6][NB] (the exchange of the split first layers)
    t->sync_bytes = sizeof(unsigned) * (size_t)CNT_STRIDE * (5 * t->NB + 2 + 6 * t->NB + 1);   // (+1: k_abc's tagged sums of log pi)
    arena.reserve(reinterpret_cast<void **>(&t->d_sync), t->sync_bytes);
    arena.reserve(reinterpret_cast<void **>(&t->d_dwl), sizeof(DwLayer) * NDW * 3);
    g_arena = nullptr;
    if (arena_commit(arena, &t->arena, s)) return -1;
    tot = 0;
    for (auto &p : parts) { *p.first = t->ws + tot; tot += round_up64(p.second, 64); }

    d.B = B; d.Bt = t->Bt; d.O = t->O; d.A = t->A; d.KP = t->KP; d.KQ = t->KQ; d.NH = t->NH; d.NB = t->NB;
    d.discount = cfg->discount; d.reward_scale = cfg->reward_scale; d.tau = cfg->soft_target_tau;
    d.target_entropy = std::isnan(cfg->target_entropy) ? -(float)t->A : cfg->target_entropy;
    d.alpha_lr = cfg->policy_lr; d.period = cfg->target_update_period;
    d.auto_alpha = cfg->use_automatic_entropy_tuning; d.noise_seed = cfg->noise_seed;
    d.ctl = t->d_ctl;
    for (int i = 0; i < 6; ++i) d.P[i] = (i < nnets) ? t->net[i].P : nullptr;
    d.algo = t->algo; d.sp = t->SP;
    d.td3_sigma = td3 ? td3->target_policy_noise : 0.f;
    d.td3_clip = td3 ? td3->target_policy_noise_clip : 0.f;
    for (int i = 0; i < 3; ++i) d.PT[i] = t->net[i].PT;
    for (int l = 0; l < 3; ++l) { d.LP[l] = t->net[0].L[l]; d.LQ[l] = t->net[1].L[l]; }
    d.diag_first = t->d_diag_host; d.diag_last = t->d_diag_host + SAC_DIAG_N; d.diag_trace = t->d_diag + 2 * SAC_DIAG_N;
    d.diag_dev = t->d_diag;
    d.eps1 = d.eps2 = nullptr;
    d.cnt = t->d_sync;
    d.abort_flag = t->d_sync + (size_t)CNT_STRIDE * (5 * t->NB + 1);

    // weight-gradient work tables: the 256x256 layers first (longest jobs).  SAC: one table (3 nets).  TD3: the two
    // critics without / with the Polyak targets (the soft update follows the critics' step on policy steps only),
    // the policy (Polyak target = target policy), and an empty one (diagnostics block only).
    std::vector<DwLayer> hl;
    auto build_table = [&](DwTable &T, const std::vector<int> &nets, bool with_targets) {
        int job = 0, nl = 0;
        const size_t base = hl.size();
        for (int li = 0; li < NDW; ++li) T.job0[li] = 1 << 30;
        // part 0: the layer's full 64-column strips (all of them unless the layer has a tail); part 1: the TAIL -- the partial
        // last strip of a first layer with more than one strip (80 input columns: one valid tile of four; the padding tiles
        // re-read the valid ones).  Tails come last in the table: the launch has more jobs than CUs on those shapes, the
        // jobs at the end of the table are the ones that share a CU with an earlier job, and a tail job streams 32-64 KB
        // where a full one streams 80.
        auto add_layer = [&](int netid, int l, const float *dYT, const float *XT, int from_slot, float lr, int part) {
            Net &n = t->net[netid];
            const Layer &L = n.L[l];
            const int nk_all = (L.Kp + 63) / 64;
            const bool has_tail = (l == 0) && (L.Kp % 64 != 0) && nk_all > 1;
            if (part == 1 && !has_tail) return;
            T.job0[nl++] = job;
            hl.emplace_back();
            DwLayer &J = hl.back();
            J.dYT = dYT; J.XT = XT; J.xt_from_slot = from_slot; J.xt_off = t->ext_layout.off_saT;
            J.P = n.P + L.offW; J.G = n.G + L.offW;
            J.PT = n.PT + L.offWt; J.MT = n.MT + L.offWt; J.VT = n.VT + L.offWt;
            J.bias = n.P + L.offB; J.mb = n.M + L.offB; J.vb = n.V + L.offB; J.gb = n.G + L.offB;
            J.TP = nullptr; J.Tbias = nullptr;
            const int tgt = (netid == 0) ? (td3 ? 5 : -1) : netid + 2;
            if (with_targets && tgt >= 0) {
                J.TP = t->net[tgt].P + L.offW;
                J.Tbias = t->net[tgt].P + L.offB;
            }
            J.ldp = L.Kp; J.ldt = L.Np; J.N = L.N; J.K = L.K; J.lr = lr;
            J.nk = (part == 1) ? 1 : (has_tail ? nk_all - 1 : nk_all);
            J.k_base = (part == 1) ? 64 * (nk_all - 1) : 0; J.pad_ = 0;
            J.job0 = job;
            job += (L.Np / 16) * J.nk;
        };
        const float *dY1[3] = {d.dPH2T, d.dQH2T, d.dQH2T + (size_t)H * B}, *X1[3] = {d.PH1T, d.QH1T, d.QH1T + (size_t)H * B};
        const float *dY0[3] = {d.dPH1T, d.dQH1T, d.dQH1T + (size_t)H * B};
        const float *dY2[3] = {d.dheadT, d.dq16T, d.dq16T + (size_t)16 * B}, *X2[3] = {d.PH2T, d.QH2T, d.QH2T + (size_t)H * B};
        const float lrs[3] = {cfg->policy_lr, cfg->qf_lr, cfg->qf_lr};
        for (int n : nets) add_layer(n, 1, dY1[n], X1[n], 0, lrs[n], 0);
        for (int n : nets) add_layer(n, 0, dY0[n], nullptr, 1, lrs[n], 0);
        for (int n : nets) add_layer(n, 2, dY2[n], X2[n], 0, lrs[n], 0);
        for (int n : nets) add_layer(n, 0, dY0[n], nullptr, 1, lrs[n], 1);
        T.njobs = job;
        T.L = t->d_dwl + base;
        T.abort = nullptr;
    };
    if (!td3) build_table(t->dw, {0, 1, 2}, true);
    else {
        build_table(t->dw_q, {1, 2}, false);
        build_table(t->dw_q_tp, {1, 2}, true);
        build_table(t->dw_pi, {0}, true);
        for (int li = 0; li < NDW; ++li) t->dw_none.job0[li] = 1 << 30;
        t->dw_none.njobs = 0; t->dw_none.L = t->d_dwl; t->dw_none.abort = nullptr;
        t->dw = t->dw_q;
    }
    SAC_HIP(hipMemcpyAsync(t->d_dwl, hl.data(), sizeof(DwLayer) * hl.size(), hipMemcpyHostToDevice, s));
    SAC_HIP(hipStreamSynchronize(s));      // hl is a local

    const int KL0q = round_up(t->KQ, 64);
    const int nth = t->NH / 16;
    const int sw = 64 * (4 / t->SP);
    t->lds_fa = sizeof(float) * (size_t)(RB * KL0q + RB * H + RB * sw + 4 * nth * 256);
    t->lds_fb = sizeof(float) * (size_t)(RB * KL0q + RB * H + RB * sw + 1024 + (t->SP == 4 ? H * WLD : 0));
    t->lds_bw = sizeof(float) * (size_t)(RB * 64 + RB * H);
    SAC_REQUIRE(t->lds_fa <= 160 * 1024 - 512, "observation too wide for the LDS row-block budget (obs_dim=%d)", t->O);
    int wide_min = 16 * RD0 + 1;                       // (SAC_WIDE_MIN_KQ: tuning experiments only)
    if (const char *e = getenv("SAC_WIDE_MIN_KQ")) wide_min = atoi(e);
    const bool wide = t->KQ >= wide_min;
#define SAC_PICK(SPV)                                                                                      \
    do {                                                                                                   \
        t->fwd_a = (nth == 1) ? (wide ? &k_fwd_a<1, true, SPV> : &k_fwd_a<1, false, SPV>)                   \
                              : (wide ? &k_fwd_a<2, true, SPV> : &k_fwd_a<2, false, SPV>);                  \
        t->fwd_b = (nth == 1) ? (wide ? &k_fwd_b<1, true, SPV> : &k_fwd_b<1, false, SPV>)                   \
                              : (wide ? &k_fwd_b<2, true, SPV> : &k_fwd_b<2, false, SPV>);                  \
        t->bwd = (nth == 1) ? &k_bwd<1, SPV> : &k_bwd<2, SPV>;                                             \
    } while (0)
#define TD3_PICK(SPV)                                                                                      \
    do {                                                                                                   \
        t->fwd_a = wide ? &k_fwd_a<1, true, SPV, M_TD3_CRITIC> : &k_fwd_a<1, false, SPV, M_TD3_CRITIC>;     \
        t->fwd_b = wide ? &k_fwd_b<1, true, SPV, M_TD3_CRITIC> : &k_fwd_b<1, false, SPV, M_TD3_CRITIC>;     \
        t->fwd_b2 = wide ? &k_fwd_b<1, true, SPV, M_TD3_ACTOR> : &k_fwd_b<1, false, SPV, M_TD3_ACTOR>;      \
        t->bwd = &k_bwd<1, SPV, M_TD3_CRITIC>;                                                             \
        t->bwd2 = &k_bwd<1, SPV, M_TD3_ACTOR>;                                                             \
    } while (0)
    if (!td3) {
        if (t->SP == 4) SAC_PICK(4);
        else if (t->SP == 2) SAC_PICK(2);
        else SAC_PICK(1);
    } else {
        if (t->SP == 4) TD3_PICK(4);
        else if (t->SP == 2) TD3_PICK(2);
        else TD3_PICK(1);
    }
#undef SAC_PICK
#undef TD3_PICK
    // The fused step (sac_fused.h) needs: SAC, column split 4 (at most 16 row-blocks), and every one of its 16*NB
    // workgroups resident at once (one per CU: 100-160 KB of LDS each, which also bounds obs_dim to ~1000).  SAC_FUSED=0 selects the
    // four-launch step (co-tenant processes on one GPU; ablations).
    {
        int cus = 0;
        SAC_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, t->device));
        const char *e = getenv("SAC_FUSED");
        t->lds_abc = sizeof(float) * (size_t)(RB * KL0q + RB * H + RB * 64 + FUSED_RED + H * WLD);
        t->fused = t->SP == 4 && t->NB <= 16 && 16 * t->NB <= cus && t->lds_abc <= 160 * 1024 - 512 && !(e && atoi(e) == 0);
        if (const char *ts = getenv("SAC_FUSED_TEST_STALL")) t->test_stall_at = (unsigned)atoi(ts);
        if (td3) t->abc = wide ? &k_abc<1, true, M_TD3_CRITIC> : &k_abc<1, false, M_TD3_CRITIC>;
        else t->abc = (nth == 1) ? (wide ? &k_abc<1, true> : &k_abc<1, false>) : (wide ? &k_abc<2, true> : &k_abc<2, false>);
        t->abc_grid = 16 * t->NB; t->abc_threads = 256;
        if (t->fused) {
            SAC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(t->abc), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)t->lds_abc));
            set_abort_ptrs(t, d.abort_flag);
            FusedGate &G = g_gate[t->device & 63];
            std::lock_guard<std::mutex> lk(G.mu);
            if (!G.ev) SAC_HIP(hipEventCreateWithFlags(&G.ev, hipEventDisableTiming));
            G.live += 1;
        }
    }
    {   // column split 1: the forward launches as one (sac_chain.h); SAC_CHAIN=0: the four-launch step (A/B comparisons)
        const char *e = getenv("SAC_CHAIN");
        t->lds_chain = sizeof(float) * (size_t)(RB * KL0q + 2 * RB * H + 4 * nth * 256 + RB * 32);
        // Where it pays (measured, scripts/large_batch_matrix.sh; round 3's second half with the eight-wave kernel): first layers
        // of at most eight k-chunks -- Door 46/7 batch 1024 58.1 -> 51.8 us per step, TwoArmHandoff 86/14 64.9 -> 60.9, and
        // now batches of more than one round of workgroups too (Door batch 1536 87.7 -> 83.9, batch 2048 95.3 -> 91.7: the
        // four-wave kernel's 350 registers lost there); Wipe's 25-chunk first layers (recomputed by both P items) still lose
        // (83.4 against 86.6).
        int cus = 0;
        SAC_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, t->device));
        const char *e8 = getenv("SAC_CHAIN8");                // (A/B comparisons: 0 = the four-wave kernel)
        t->chain8 = !(e8 && atoi(e8) == 0);
        const bool pays = ((t->chain8 || 4 * t->NB <= cus) && t->KQ <= 128) || (e && atoi(e) == 1);
        t->chain = !td3 && !t->fused && t->SP == 1 && (t->NB % 2) == 0 && t->lds_chain <= 160 * 1024 - 512 && pays && !(e && atoi(e) == 0);
        {
            const bool wide4 = t->KQ > 64;       // first layers of more than four k-chunks
            t->chaink = (nth == 1) ? (wide4 ? &k_chain<1, true> : &k_chain<1, false>) : (wide4 ? &k_chain<2, true> : &k_chain<2, false>);
            if (t->chain8)
                t->chaink = (nth == 1) ? (wide4 ? &k_chain8<1, true> : &k_chain8<1, false>) : (wide4 ? &k_chain8<2, true> : &k_chain8<2, false>);
        }
        {   // column split 1, SAC: the backward launch on eight waves too (SAC_BWD8=0: the four-wave kernel, A/B comparisons)
            const char *eb = getenv("SAC_BWD8");
            t->bwd8 = !td3 && t->SP == 1 && !(eb && atoi(eb) == 0);
            if (t->bwd8) t->bwd = (nth == 1) ? &k_bwd8<1> : &k_bwd8<2>;
        }
        {   // one round of workgroups (batch 1024): the backward blocks inside the forward launch behind in-launch hand-offs
            // (k_chain8<.., BWD>, sac_chain.h) -- a fused step like k_abc's: same give-up protocol, same fall-back (to k_chain8 + k_bwd8)
            // Where it pays (A/B on one box, 2 x 2000 steps): exactly one workgroup per CU and narrow first layers -- Door 46/7
            // batch 1024 19 290 -> 19 860 steps/s (the launch 33.6 us for 22.3 + 10.6 + a boundary); TwoArmHandoff 86/14 batch 1024
            // -3.5 %, batch 992 -3 %, batch 800 -8 % (fewer workgroups than CUs: the separate backward launch was spreading its
            // 192 blocks over idle CUs).  SAC_CHAIN_BWD=1 / 0 forces it (any batch whose workgroups are all resident) / off.
            const char *ecb = getenv("SAC_CHAIN_BWD");
            const bool pays_b = 4 * t->NB == cus && t->KQ <= 64;
            t->chain_bwd = t->chain && t->chain8 && t->bwd8 && 4 * t->NB <= cus && (ecb ? atoi(ecb) == 1 : pays_b);
            if (t->chain_bwd) {
                const bool wide4 = t->KQ > 64;
                t->abc = (nth == 1) ? (wide4 ? &k_chain8<1, true, true> : &k_chain8<1, false, true>)
                                    : (wide4 ? &k_chain8<2, true, true> : &k_chain8<2, false, true>);
                t->lds_abc = t->lds_chain > sizeof(float) * (size_t)(RB * 64 + RB * H) ? t->lds_chain : sizeof(float) * (size_t)(RB * 64 + RB * H);
                t->abc_grid = 4 * t->NB; t->abc_threads = 512;
                t->fused = true;
                if (const char *ts = getenv("SAC_FUSED_TEST_STALL")) t->test_stall_at = (unsigned)atoi(ts);
                SAC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(t->abc), hipFuncAttributeMaxDynamicSharedMemorySize, (int)t->lds_abc));
                set_abort_ptrs(t, d.abort_flag);
                FusedGate &G = g_gate[t->device & 63];
                std::lock_guard<std::mutex> lk(G.mu);
                if (!G.ev) SAC_HIP(hipEventCreateWithFlags(&G.ev, hipEventDisableTiming));
                G.live += 1;
            }
        }
        if (t->chain && t->lds_chain > 64 * 1024)
            SAC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(t->chaink), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)t->lds_chain));
    }
    if (t->lds_fa > 64 * 1024)
        SAC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(t->fwd_a),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)t->lds_fa));
    if (t->lds_fb > 64 * 1024) {
        SAC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(t->fwd_b),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)t->lds_fb));
        if (t->fwd_b2)
            SAC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(t->fwd_b2),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)t->lds_fb));
    }
    SAC_REQUIRE(t->lds_fb <= 160 * 1024 - 512, "observation too wide for the LDS row-block budget (obs_dim=%d)", t->O);
    return trainer_prime_events(t);
}

// hidden_sizes of any depth (sac_hip.h): shapes the fused kernels carry -- two hidden layers of at most 256 units per
// family -- get them; everything else gets the general step (sac_general.h).  SAC_GENERAL=1 in the environment selects
// the general step for every shape (cross-checks).
int sac_trainer_create_mlp(sac_trainer_t **out, const sac_config_t *cfg, const int32_t *policy_hidden, int32_t n_policy_hidden,
                           const int32_t *qf_hidden, int32_t n_qf_hidden) {
    SAC_REQUIRE(out && cfg && policy_hidden && qf_hidden, "null argument to sac_trainer_create_mlp");
    *out = nullptr;
    SAC_REQUIRE(n_policy_hidden >= 1 && n_policy_hidden < gen::GMAXL && n_qf_hidden >= 1 && n_qf_hidden < gen::GMAXL,
                "%d / %d hidden layers unsupported: 1..%d per network", n_policy_hidden, n_qf_hidden, gen::GMAXL - 1);
    bool fits = n_policy_hidden == 2 && n_qf_hidden == 2;
    for (int i = 0; i < n_policy_hidden; ++i) {
        SAC_REQUIRE(policy_hidden[i] >= 1 && policy_hidden[i] <= 4096, "policy hidden size %d unsupported (1..4096)", policy_hidden[i]);
        fits = fits && policy_hidden[i] <= H;
    }
    for (int i = 0; i < n_qf_hidden; ++i) {
        SAC_REQUIRE(qf_hidden[i] >= 1 && qf_hidden[i] <= 4096, "qf hidden size %d unsupported (1..4096)", qf_hidden[i]);
        fits = fits && qf_hidden[i] <= H;
    }
    const char *force = getenv("SAC_GENERAL");
    if (fits && !(force && atoi(force) == 1)) {
        sac_config_t c = *cfg;
        for (int i = 0; i < 2; ++i) { c.policy_hidden[i] = policy_hidden[i]; c.qf_hidden[i] = qf_hidden[i]; }
        return sac_trainer_create(out, &c);
    }
    SAC_REQUIRE(sac_device_count() > 0, "no HIP device visible: libsac_hip has no CPU fallback");
    SAC_REQUIRE(cfg->obs_dim > 0 && cfg->act_dim > 0 && cfg->act_dim <= 16,
                "unsupported dims obs=%d act=%d (act_dim must be in 1..16)", cfg->obs_dim, cfg->act_dim);
    SAC_REQUIRE(cfg->obs_dim <= 496, "obs_dim %d unsupported (the minibatch slots hold cat(obs, act) rows of at most 512 columns)",
                cfg->obs_dim);
    SAC_REQUIRE(cfg->batch > 0, "batch size %d must be positive", cfg->batch);
    SAC_REQUIRE(cfg->target_update_period > 0, "target_update_period must be positive");
    SAC_HIP(hipSetDevice(cfg->device));
    sac_trainer *t = new sac_trainer();
    int hp[gen::GMAXL], hq[gen::GMAXL];
    for (int i = 0; i < n_policy_hidden; ++i) hp[i] = policy_hidden[i];
    for (int i = 0; i < n_qf_hidden; ++i) hq[i] = qf_hidden[i];
    if (trainer_common_init(t, cfg) || gen_build(t, hp, n_policy_hidden, hq, n_qf_hidden) || trainer_prime_events(t)) {
        sac_trainer_destroy(t);
        return -1;
    }
    *out = t;
    return 0;
}

// TD3 with hidden_sizes of any depth (sac_hip.h): the fused kernels' shapes through td3_trainer_create, everything else on
// the general step (gen_build_td3)
int td3_trainer_create_mlp(sac_trainer_t **out, const td3_config_t *c, const int32_t *policy_hidden, int32_t n_policy_hidden,
                           const int32_t *qf_hidden, int32_t n_qf_hidden) {
    SAC_REQUIRE(out && c && policy_hidden && qf_hidden, "null argument to td3_trainer_create_mlp");
    *out = nullptr;
    SAC_REQUIRE(n_policy_hidden >= 1 && n_policy_hidden < gen::GMAXL && n_qf_hidden >= 1 && n_qf_hidden < gen::GMAXL,
                "%d / %d hidden layers unsupported: 1..%d per network", n_policy_hidden, n_qf_hidden, gen::GMAXL - 1);
    bool fits = n_policy_hidden == 2 && n_qf_hidden == 2;
    for (int i = 0; i < n_policy_hidden; ++i) {
        SAC_REQUIRE(policy_hidden[i] >= 1 && policy_hidden[i] <= 4096, "policy hidden size %d unsupported (1..4096)", policy_hidden[i]);
        fits = fits && policy_hidden[i] <= H;
    }
    for (int i = 0; i < n_qf_hidden; ++i) {
        SAC_REQUIRE(qf_hidden[i] >= 1 && qf_hidden[i] <= 4096, "qf hidden size %d unsupported (1..4096)", qf_hidden[i]);
        fits = fits && qf_hidden[i] <= H;
    }
    const char *force = getenv("SAC_GENERAL");
    if (fits && !(force && atoi(force) == 1)) {
        td3_config_t cc = *c;
        for (int i = 0; i < 2; ++i) { cc.policy_hidden[i] = policy_hidden[i]; cc.qf_hidden[i] = qf_hidden[i]; }
        return td3_trainer_create(out, &cc);
    }
    SAC_REQUIRE(sac_device_count() > 0, "no HIP device visible: libsac_hip has no CPU fallback");
    SAC_REQUIRE(c->policy_and_target_update_period > 0, "policy_and_target_update_period must be positive");
    SAC_REQUIRE(c->target_policy_noise >= 0.f && c->target_policy_noise_clip >= 0.f, "negative target policy noise");
    SAC_REQUIRE(c->obs_dim > 0 && c->act_dim > 0 && c->act_dim <= 16, "unsupported dims obs=%d act=%d (act_dim must be in 1..16)",
                c->obs_dim, c->act_dim);
    SAC_REQUIRE(c->obs_dim <= 496, "obs_dim %d unsupported (the minibatch slots hold cat(obs, act) rows of at most 512 columns)", c->obs_dim);
    SAC_REQUIRE(c->batch > 0, "batch size %d must be positive", c->batch);
    SAC_HIP(hipSetDevice(c->device));
    sac_config_t s{};
    s.obs_dim = c->obs_dim; s.act_dim = c->act_dim; s.hidden = c->hidden; s.batch = c->batch;
    s.discount = c->discount; s.reward_scale = c->reward_scale;
    s.policy_lr = c->policy_learning_rate; s.qf_lr = c->qf_learning_rate;
    s.soft_target_tau = c->tau; s.target_update_period = 1; s.use_automatic_entropy_tuning = 0;
    s.target_entropy = 0.f; s.noise_seed = c->noise_seed; s.device = c->device;
    sac_trainer *t = new sac_trainer();
    t->algo = 1;
    t->td3_period = c->policy_and_target_update_period;
    int hp[gen::GMAXL], hq[gen::GMAXL];
    for (int i = 0; i < n_policy_hidden; ++i) hp[i] = policy_hidden[i];
    for (int i = 0; i < n_qf_hidden; ++i) hq[i] = qf_hidden[i];
    if (trainer_common_init(t, &s) || gen_build_td3(t, c, hp, n_policy_hidden, hq, n_qf_hidden) || trainer_prime_events(t)) {
        sac_trainer_destroy(t);
        return -1;
    }
    *out = t;
    return 0;
}

int sac_trainer_destroy(sac_trainer_t *t) {
    if (!t) return 0;
    (void)hipSetDevice(t->device);
    if (t->stream) (void)hipStreamSynchronize(t->stream);
    if (t->fused) {
        FusedGate &G = g_gate[t->device & 63];
        std::lock_guard<std::mutex> lk(G.mu);
        if (!t->gate_exempt) G.live -= 1;
        if (G.last == t->stream) G.last = nullptr;
    }
    (void)hipFree(t->arena);
    gen_destroy(t->gen);
    if (t->h_stage) (void)hipHostFree(t->h_stage);
    if (t->h_diag) (void)hipHostFree(t->h_diag);
    for (auto &e : t->ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : t->ev_tm) if (e) (void)hipEventDestroy(e);
    for (auto &e : t->ev_ready) if (e) (void)hipEventDestroy(e);
    for (auto &e : t->ev_done) if (e) (void)hipEventDestroy(e);
    for (auto &e : t->thr_ev) if (e) (void)hipEventDestroy(e);
    if (t->stream) { sac::stream_unregister(t->stream); (void)hipStreamDestroy(t->stream); }
    delete t;
    return 0;
}

int64_t sac_param_count(const sac_trainer_t *t, int net) {
    if (!t || net < 0 || net > (t->algo == 1 ? 5 : 4)) return -1;
    if (t->gen) return t->gen->net[net].nflat;
    return flat_count(flat_map(t, net));
}

static int upload_padded(sac_trainer *t, int net, const float *flat, float *devbuf, float *devT) {
    Net &n = t->net[net];
    const bool also_transposed = devT != nullptr;
    std::vector<float> P((size_t)n.nP, 0.f), PT;
    if (also_transposed) PT.assign((size_t)n.nPT, 0.f);
    for_each_param(t, net, [&](int64_t fi, int l, int nn, int k, bool is_b) {
        const Layer &L = n.L[l];
        if (is_b) P[L.offB + nn] = flat[fi];
        else {
            P[L.offW + frag_off(nn, k, L.Kp)] = flat[fi];
            if (also_transposed) PT[L.offWt + frag_off(k, nn, L.Np)] = flat[fi];
        }
    });
    SAC_HIP(hipMemcpyAsync(devbuf, P.data(), sizeof(float) * P.size(), hipMemcpyHostToDevice, t->stream));
    if (also_transposed)
        SAC_HIP(hipMemcpyAsync(devT, PT.data(), sizeof(float) * PT.size(), hipMemcpyHostToDevice, t->stream));
    SAC_HIP(hipStreamSynchronize(t->stream));
    return 0;
}

// weights come from the transposed buffer when devT is given (Adam moments), else from devbuf
static int download_padded(sac_trainer *t, int net, const float *devbuf, float *flat, const float *devT = nullptr) {
    Net &n = t->net[net];
    std::vector<float> P((size_t)n.nP), PT;
    SAC_HIP(hipMemcpyAsync(P.data(), devbuf, sizeof(float) * P.size(), hipMemcpyDeviceToHost, t->stream));
    if (devT) {
        PT.resize((size_t)n.nPT);
        SAC_HIP(hipMemcpyAsync(PT.data(), devT, sizeof(float) * PT.size(), hipMemcpyDeviceToHost, t->stream));
    }
    SAC_HIP(hipStreamSynchronize(t->stream));
    for_each_param(t, net, [&](int64_t fi, int l, int nn, int k, bool is_b) {
        const Layer &L = n.L[l];
        if (is_b) flat[fi] = P[L.offB + nn];
        else flat[fi] = devT ? PT[L.offWt + frag_off(k, nn, L.Np)] : P[L.offW + frag_off(nn, k, L.Kp)];
    });
    return 0;
}

int sac_set_params(sac_trainer_t *t, int net, const float *flat, int64_t n) {
    SAC_REQUIRE(t && flat && net >= 0 && net <= (t->algo == 1 ? 5 : 4), "bad arguments to sac_set_params");
    SAC_REQUIRE(n == sac_param_count(t, net), "net %d expects %lld parameters, got %lld", net,
                (long long)sac_param_count(t, net), (long long)n);
    SAC_HIP(hipSetDevice(t->device));
    t->mirror_valid = false;
    if (t->gen) return gen_upload(t, net, flat, t->gen->net[net].P);
    return upload_padded(t, net, flat, t->net[net].P, net < 3 ? t->net[net].PT : nullptr);
}

int sac_get_params(sac_trainer_t *t, int net, float *flat, int64_t n) {
    SAC_REQUIRE(t && flat && net >= 0 && net <= (t->algo == 1 ? 5 : 4), "bad arguments to sac_get_params");
    SAC_REQUIRE(n == sac_param_count(t, net), "net %d holds %lld parameters, buffer has %lld", net,
                (long long)sac_param_count(t, net), (long long)n);
    SAC_HIP(hipSetDevice(t->device));
    if (t->gen) return gen_download(t, net, t->gen->net[net].P, flat);
    return download_padded(t, net, t->net[net].P, flat);
}

int sac_set_opt_state(sac_trainer_t *t, int net, const float *m, const float *v, int64_t n) {
    SAC_REQUIRE(t && m && v && net >= 0 && net <= 2, "bad arguments to sac_set_opt_state (trained nets are 0..2)");
    SAC_REQUIRE(n == sac_param_count(t, net), "size mismatch in sac_set_opt_state");
    SAC_HIP(hipSetDevice(t->device));
    if (t->gen) return gen_upload(t, net, m, t->gen->net[net].M) || gen_upload(t, net, v, t->gen->net[net].V) ? -1 : 0;
    if (upload_padded(t, net, m, t->net[net].M, t->net[net].MT)) return -1;
    return upload_padded(t, net, v, t->net[net].V, t->net[net].VT);
}

int sac_get_opt_state(sac_trainer_t *t, int net, float *m, float *v, int64_t n) {
    SAC_REQUIRE(t && m && v && net >= 0 && net <= 2, "bad arguments to sac_get_opt_state (trained nets are 0..2)");
    SAC_REQUIRE(n == sac_param_count(t, net), "size mismatch in sac_get_opt_state");
    SAC_HIP(hipSetDevice(t->device));
    if (t->gen) return gen_download(t, net, t->gen->net[net].M, m) || gen_download(t, net, t->gen->net[net].V, v) ? -1 : 0;
    if (download_padded(t, net, t->net[net].M, m, t->net[net].MT)) return -1;
    return download_padded(t, net, t->net[net].V, v, t->net[net].VT);
}

int sac_set_scalars(sac_trainer_t *t, const double sc[6]) {
    SAC_REQUIRE(t && sc, "bad arguments to sac_set_scalars");
    SAC_HIP(hipSetDevice(t->device));
    Ctl c;
    memset(&c, 0, sizeof(c));
    if (t->algo == 1) t->adam_t_pi = (long long)sc[0];            // TD3: sc[0] = optimizer steps of the (delayed) policy
    else { c.log_alpha = (float)sc[0]; c.a_m = (float)sc[1]; c.a_v = (float)sc[2]; }
    t->adam_t = (long long)sc[3]; t->n_train_steps_total = (long long)sc[4];
    c.alpha = t->cfg.use_automatic_entropy_tuning ? expf(c.log_alpha) : 1.0f;
    SAC_HIP(hipMemcpyAsync(t->d_ctl, &c, sizeof(c), hipMemcpyHostToDevice, t->stream));
    SAC_HIP(hipStreamSynchronize(t->stream));
    return 0;
}

int sac_get_scalars(sac_trainer_t *t, double sc[6]) {
    SAC_REQUIRE(t && sc, "bad arguments to sac_get_scalars");
    SAC_HIP(hipSetDevice(t->device));
    Ctl c;
    SAC_HIP(hipMemcpyAsync(&c, t->d_ctl, sizeof(c), hipMemcpyDeviceToHost, t->stream));
    SAC_HIP(hipStreamSynchronize(t->stream));
    sc[0] = c.log_alpha; sc[1] = c.a_m; sc[2] = c.a_v; sc[3] = (double)t->adam_t;
    sc[4] = (double)t->n_train_steps_total; sc[5] = c.alpha;
    if (t->algo == 1) { sc[0] = (double)t->adam_t_pi; sc[1] = sc[2] = sc[5] = 0.0; }
    return 0;
}

int sac_step(sac_trainer_t *t, const float *obs, const float *act, const float *rew, const float *term,
             const float *next_obs, const float *eps1, const float *eps2, float *diag) {
    SAC_REQUIRE(t && obs && act && rew && term && next_obs, "null batch pointer in sac_step");
    SAC_REQUIRE(t->algo == 1 || (eps1 == nullptr) == (eps2 == nullptr), "eps1 and eps2 must both be given or both be NULL");
    if (t->algo == 1) eps1 = eps2;                       // TD3 draws one noise tensor (target smoothing): eps2
    SAC_HIP(hipSetDevice(t->device));
    const int B = t->B, Bt = t->Bt, O = t->O, A = t->A;     // B: slot rows (padded to row-blocks), Bt: rows given
    const SlotLayout &L = t->ext_layout;
    hipStream_t s = t->stream;
    // np_to_pytorch_batch: host fp32 -> pinned -> HBM slot (row-major part), + feature-major saT; rows beyond the
    // batch (row-block padding) are zero and carry no weight
    const size_t nfl = (size_t)B * (2 * O + A + 2) + (size_t)L.KQ64 * B + (eps1 ? 2 * (size_t)B * A : 0);
    if (ensure_stage_t(t, sizeof(float) * nfl)) return -1;
    float *st = (float *)t->h_stage;
    memset(st, 0, sizeof(float) * nfl);
    float *so = st, *sa = so + (size_t)B * O, *sr = sa + (size_t)B * A, *stt = sr + B, *sn = stt + B,
          *sT = sn + (size_t)B * O, *se = sT + (size_t)L.KQ64 * B;
    memcpy(so, obs, sizeof(float) * Bt * O); memcpy(sa, act, sizeof(float) * Bt * A);
    memcpy(sr, rew, sizeof(float) * Bt); memcpy(stt, term, sizeof(float) * Bt);
    memcpy(sn, next_obs, sizeof(float) * Bt * O);
    for (int b = 0; b < Bt; ++b) {
        for (int k = 0; k < O; ++k) sT[frag_off(k, b, B)] = obs[(size_t)b * O + k];
        for (int k = 0; k < A; ++k) sT[frag_off(L.KA + k, b, B)] = act[(size_t)b * A + k];
    }
    float *E = t->ext_slot;
    SAC_HIP(hipMemcpyAsync(E + L.off_obs, so, sizeof(float) * B * O, hipMemcpyHostToDevice, s));
    SAC_HIP(hipMemcpyAsync(E + L.off_act, sa, sizeof(float) * B * A, hipMemcpyHostToDevice, s));
    SAC_HIP(hipMemcpyAsync(E + L.off_rew, sr, sizeof(float) * B, hipMemcpyHostToDevice, s));
    SAC_HIP(hipMemcpyAsync(E + L.off_term, stt, sizeof(float) * B, hipMemcpyHostToDevice, s));
    SAC_HIP(hipMemcpyAsync(E + L.off_nobs, sn, sizeof(float) * B * O, hipMemcpyHostToDevice, s));
    SAC_HIP(hipMemcpyAsync(E + L.off_saT, sT, sizeof(float) * (size_t)L.KQ64 * B, hipMemcpyHostToDevice, s));
    if (eps1) {
        memcpy(se, eps1, sizeof(float) * Bt * A); memcpy(se + (size_t)B * A, eps2, sizeof(float) * Bt * A);
        SAC_HIP(hipMemcpyAsync(t->d_eps, se, sizeof(float) * 2 * B * A, hipMemcpyHostToDevice, s));
        t->dev.eps1 = t->d_eps; t->dev.eps2 = t->d_eps + (size_t)B * A;
    } else {
        t->dev.eps1 = t->dev.eps2 = nullptr;
    }
    if (launch_step(t, t->ext_slot, L, 0, nullptr, diag != nullptr)) return -1;
    if (wait_trainer_stream(t)) return -1;
    {   // a fused step that gave up: earlier device-batch steps still on record first, then this one again
        unsigned lost = 0;
        const int fa = check_fused_abort(t, &lost);
        if (fa < 0) return -1;
        if (fa == 1) {
            if (lost > 1 && replay_pending(t, lost - 1)) return -1;
            t->pend_n = 0;
            t->dev.eps1 = eps1 ? t->d_eps : nullptr; t->dev.eps2 = eps1 ? t->d_eps + (size_t)B * A : nullptr;
            if (launch_step(t, t->ext_slot, L, 0, nullptr, diag != nullptr)) return -1;
            if (wait_trainer_stream(t)) return -1;
        }
    }
    if (diag) memcpy(diag, t->h_diag + SAC_DIAG_N, sizeof(float) * SAC_DIAG_N);
    t->mirror_valid = false;
    return 0;
}

// trainer.train(batch) on a device-resident batch of sac_random_batch_device: no host copy, no synchronisation
// (diag != null: the step's diagnostics are copied out, which waits for the step).
int sac_step_device(sac_trainer_t *t, sac_buffer_t *b, int64_t token, float diag[SAC_DIAG_N]) {
    SAC_REQUIRE(t && b, "null argument to sac_step_device");
    SAC_REQUIRE(b->device == t->device && b->O == t->O && b->A == t->A, "buffer does not match trainer");
    const int slot = sac_ring_slot_of(b, token);
    if (slot < 0) return -1;
    SAC_REQUIRE(b->ring_layout.Bt == t->Bt, "device batch holds %d rows, the trainer was created for %d", b->ring_layout.Bt, t->Bt);
    SAC_HIP(hipSetDevice(t->device));
    hipStream_t s = t->stream;
    // one event per CHUNK of batches the buffer drew and gathered together (read-ahead, sac_random_batch_device): a step on
    // a later batch of a chunk this stream has already waited for needs no wait of its own (an in-stream wait costs the
    // stream ~3 us even when its event fired long ago)
    if (b->multi_stream || b->waited_stream != s || b->ring_chunk_token[slot] > b->waited_chunk_token) {
        SAC_HIP(hipStreamWaitEvent(s, b->ring_ready[b->ring_first[slot]], 0));
        b->waited_stream = s;
        b->waited_chunk_token = b->ring_chunk_token[slot];
    }
    t->dev.eps1 = t->dev.eps2 = nullptr;
    t->publish_diag = diag != nullptr;
    const bool was_fused = t->fused;
    const int rc_step = launch_step(t, b->d_ring + (size_t)slot * b->ring_layout.slot_floats, b->ring_layout, 0, nullptr, diag != nullptr);
    t->publish_diag = true;
    if (rc_step) return -1;
    if (was_fused) {
        // on record until the host has seen it applied (a fused step that gives up is re-run from its slot)
        if (t->pend_n == sac_trainer::NPEND) { t->pend_head = (t->pend_head + 1) % sac_trainer::NPEND; t->pend_n -= 1; }
        t->pend[(t->pend_head + t->pend_n++) % sac_trainer::NPEND] = sac_trainer::Pending{b, token};
        const long long k = t->dev_steps++;
        if ((k & 15) == 15) SAC_HIP(hipEventRecord(t->thr_ev[(k >> 4) & 3], s));
    }
    {   // (see sac_buffer::free4: one "done with the slots so far" event per sixteen steps)
        sac::forget_dead_step_stream(b);
        if (b->step_stream && b->step_stream != s && !b->multi_stream) {
            // a second trainer: one event per step from here on.  The first trainer's steps in flight were covered by the
            // per-sixteen-steps events only: the buffer's stream waits for all of them once, here
            SAC_HIP(hipEventRecord(b->free4[0], b->step_stream));
            SAC_HIP(hipStreamWaitEvent(b->stream, b->free4[0], 0));
            b->multi_stream = true;
        }
        b->step_stream = s;
        const int64_t k = b->step_seq++;
        b->slot_seq[slot] = k;
        if (b->multi_stream) SAC_HIP(hipEventRecord(b->ring_free[slot], s));
        else if ((k & 15) == 15) { SAC_HIP(hipEventRecord(b->free4[(k >> 4) & 3], s)); b->free4_seq[(k >> 4) & 3] = k; }   // (one per SIXTEEN steps: 4 events cover the ring of 64)
        b->ring_in_use[slot] = true;
    }
    t->mirror_valid = false;
    if (diag) {
        if (wait_trainer_stream(t)) return -1;
        if (recover_device_steps(t) < 0) return -1;
        memcpy(diag, t->h_diag + SAC_DIAG_N, sizeof(float) * SAC_DIAG_N);
    } else if (was_fused) {
        // Nobody waits for these steps.  The host looks at the give-up word (mapped pinned memory: a plain read) on every
        // call, and when it enters a new group of sixteen steps it makes sure the device is through the group before the
        // last one -- normally long true -- so that at most 47 steps are ever unverified and their slots still intact.
        const long long k = t->dev_steps - 1;
        bool gave_up = t->h_diag[SAC_DIAG_N + 31] != 0.f;
        if (!gave_up && (k & 15) == 0 && k >= 32) {
            if (wait_event(t->thr_ev[((k >> 4) - 2) & 3])) return -1;
            gave_up = t->h_diag[SAC_DIAG_N + 31] != 0.f;
            if (!gave_up && t->pend_n > 17) { t->pend_head = (t->pend_head + t->pend_n - 17) % sac_trainer::NPEND; t->pend_n = 17; }
        }
        if (gave_up) {
            if (wait_trainer_stream(t)) return -1;
            if (recover_device_steps(t) < 0) return -1;
        }
    }
    return 0;
}

// Slots of a loop live in a ring of LOOP_RING minibatch slots that is allocated ONCE (first use; never resized with
// n_steps: a 20-step call behind a 2000-step one must not free / allocate / clear inside the call).  The loop is cut
// into chunks -- 4, 12, 48, 192, then 256 steps each -- laid out back to back in the ring: while the trainer's stream
// runs the steps of chunk c, the buffer's stream draws the indices of chunk c+1 (one serial wave, ~1.1 us per batch) and
// gathers its slots; the short first chunks let step 0 start behind a 4-step draw + gather instead of a 256-step one.  The
// index stream is one in-order sequence on the buffer's stream, so it consumes NumPy's generator exactly like
// n_steps random_batch calls.
constexpr int64_t LOOP_CH = 256, LOOP_RING = 2 * LOOP_CH;
// Lengths of the first chunks (they sum to LOOP_CH, so every later chunk starts on a multiple of it).
static const int64_t *loop_plan() {
    static int64_t plan[8] = {4, 12, 48, 192, 0, 0, 0, 0};      // (1,3,12,48,192 and 2,6,24,96,128 measured the same within 1 %)
    static bool init = false;
    if (!init) {
        init = true;
        if (const char *e = getenv("SAC_CHUNK_PLAN")) {       // tuning experiments only: e.g. "4,12,48,192"
            int64_t v[8] = {0}, sum = 0;
            int n = 0;
            for (const char *p = e; *p && n < 7; ++n) {
                v[n] = strtoll(p, const_cast<char **>(&p), 10);
                sum += v[n];
                if (*p == ',') ++p;
            }
            if (sum == LOOP_CH) for (int i = 0; i < 8; ++i) plan[i] = v[i];
        }
    }
    return plan;
}
static inline int64_t loop_chunk_len(int64_t done) {
    if (done >= LOOP_CH) return LOOP_CH;
    const int64_t *plan = loop_plan();
    int64_t at = 0;
    for (int i = 0; i < 8 && plan[i] > 0; ++i) {
        if (done == at) return plan[i];
        at += plan[i];
    }
    return LOOP_CH - done;
}

// n_steps steps of a loop whose first step is number step0 of the caller's loop (step0 > 0: the remainder of a call
// whose fused step gave up).  *lost: steps at the end that a fused launch giving up left unapplied (0: none).
static int train_loop_run(sac_trainer_t *t, sac_buffer_t *b, int64_t n_steps, int64_t step0, unsigned *lost_out) {
    static const bool host_timing = getenv("SAC_HOST_TIMING") != nullptr;      // diagnostic: where the host spends the call
    const auto ht0 = std::chrono::steady_clock::now();
    auto ht = [&](const char *what) {
        if (host_timing) fprintf(stderr, "[sac_train_loop %lld] %s at %.1f us\n", (long long)n_steps, what,
                                 std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ht0).count());
    };
    hipStream_t s = t->stream;
    t->dev.eps1 = t->dev.eps2 = nullptr;
    b->in_loop = true;                   // (this call's own draws and set-up are not outside touches of the buffer)
    struct InLoop { sac_buffer *b; ~InLoop() { b->in_loop = false; } } in_loop_guard{b};
    if (ensure_slots(b, t->Bt, LOOP_RING)) return -1;
    if (ensure_idx(b, LOOP_RING * t->B)) return -1;
    ht("set-up done");
    struct Live { int64_t pos, m; int ev; };
    Live live[sac_trainer::NLOOP_EV + 4];
    int n_live = 0;
    // Chunk 0 is drawn and gathered on the TRAINER's stream, in front of its steps: handing it over from the buffer's
    // stream put a cross-stream event wait and a second stream's launch latency in front of the first step (16 us of a
    // 20-step call).  The buffer's stream carries the later chunks; if it still has work in flight (an ingest, a
    // random_batch nobody consumed) the trainer's stream waits for it first.
    if (hipStreamQuery(b->stream) != hipSuccess) {
        SAC_HIP(hipEventRecord(b->ev[3], b->stream));
        SAC_HIP(hipStreamWaitEvent(s, b->ev[3], 0));
    }
    if (!t->loop_primed) {
        // first loop of this trainer: one empty hand-shake between the two streams in each direction, so that the first
        // multi-chunk call does not pay the runtime's first-use cost of those paths (~30 us) in front of its second chunk
        t->loop_primed = true;
        SAC_HIP(hipEventRecord(t->ev_ready[1], b->stream));
        SAC_HIP(hipStreamWaitEvent(s, t->ev_ready[1], 0));
        SAC_HIP(hipEventRecord(t->ev_done[1], s));
        SAC_HIP(hipStreamWaitEvent(b->stream, t->ev_done[1], 0));
    }
    // a call continues in the slot ring where the previous one ended -- if all of it (and a speculative chunk behind it) fits
    // without wrapping; else it starts at slot 0 like every call used to (the previous calls' steps are long done either way)
    int64_t done = 0, pos = b->loop_pos;
    if (pos + n_steps + sac_buffer::LOOP_SPEC > LOOP_RING) pos = 0;
    int timed = 0;
    // A speculative first chunk left by the previous call (sac_buffer::spec_valid): still what this call would draw?
    const bool spec_hit = b->spec_valid && step0 == 0 && b->spec_batch == t->Bt && b->spec_size == b->size &&
                          n_steps >= sac_buffer::LOOP_SPEC && b->spec_pos == pos;
    if (b->spec_valid && !spec_hit && loop_spec_drop(b)) return -1;
    b->spec_valid = false;
    // the host mirror of the generator (sac_buffer::host_seen; np.random's own words when bound) follows in ONE piece
    // behind the last launch: ~1.3 us of host time per batch that no launch of this call has to wait for
    struct MirrorLater {
        sac_buffer *b; int batch;
        ~MirrorLater() { b->defer_mirror = false; if (b->deferred_batches) host_rng_advance(b, batch, b->deferred_batches); b->deferred_batches = 0; }
    } mirror_later{b, t->Bt};
    b->defer_mirror = true; b->deferred_batches = 0;
    // The next call's first chunk, speculatively (see sac_buffer::spec_valid): only for loop calls in a row, only in slots
    // this call does not use, drawn behind this call's LAST draw on the buffer's stream -- it runs under this call's steps.
    // The mirror is NOT advanced: these batches have not been handed out.
    const bool spec_ok = step0 == 0 && b->loop_streak >= 1 && pos + n_steps + sac_buffer::LOOP_SPEC <= LOOP_RING && b->size > 1 &&
                         !getenv("SAC_LOOP_SPEC_OFF");
    bool spec_launched = false;
    auto launch_spec = [&](int64_t sp) -> int {
        if (!b->spec_ev) SAC_HIP(hipEventCreateWithFlags(&b->spec_ev, hipEventDisableTiming));
        b->ra_internal = true;           // (a draw of the library's own: no roll-back, no mirror)
        const int rc = launch_sample(b, t->Bt, sac_buffer::LOOP_SPEC, sp * t->B, nullptr, b->stream);
        b->ra_internal = false;
        if (rc) return -1;
        if (launch_gather(b, b->d_idx + sp * t->B, t->B, sac_buffer::LOOP_SPEC, b->d_slots + (size_t)sp * b->slot.slot_floats, b->slot, 1, b->stream))
            return -1;
        SAC_HIP(hipEventRecord(b->spec_ev, b->stream));
        b->spec_batch = t->Bt; b->spec_size = b->size; b->spec_pos = sp;
        b->spec_valid = true;            // (from here on the device generator is ahead of the mirror: every exit path must know)
        spec_launched = true;
        return 0;
    };
    for (int c = 0; done < n_steps; ++c) {
        const int64_t first = done, want = loop_chunk_len(done);
        int64_t m = (n_steps - first < want) ? n_steps - first : want;
        // (a tail shorter than half this chunk joins it: every chunk boundary is a cross-stream wait)
        if (const int64_t rest = n_steps - first - m; rest > 0 && 2 * rest < m && m + rest <= LOOP_CH) m += rest;
        const bool use_spec = (c == 0) && spec_hit;
        if (use_spec) m = sac_buffer::LOOP_SPEC;       // (drawn and gathered by the previous call, at `pos`)
        if (pos + m > LOOP_RING) pos = 0;
        const int e = c % sac_trainer::NLOOP_EV;
        hipStream_t q = (c == 0) ? s : b->stream;
        // chunks that still own some of these slots: the trainer must be done with them
        for (int i = 0; i < n_live;) {
            if (live[i].pos < pos + m && pos < live[i].pos + live[i].m) {
                SAC_HIP(hipStreamWaitEvent(b->stream, t->ev_done[live[i].ev], 0));
                live[i] = live[--n_live];
            } else ++i;
        }
        // sac_last_loop_ms reports the draw and the gather of ONE chunk: chunk 1 (on the buffer's stream, which has slack)
        // when the call has one, else chunk 0 -- whose events then sit in front of the first step (~3 us each of host time)
        if (c == 0) timed = (n_steps > m) ? 1 : (use_spec ? -1 : 0);
        if (use_spec) {
            // the chunk is there (normally for a long time: it ran under the previous call's last steps); the host mirror
            // now counts its batches as handed out
            b->deferred_batches += m;
            hipError_t st = hipEventQuery(b->spec_ev);
            if (st != hipSuccess) { (void)hipGetLastError(); SAC_HIP(hipStreamWaitEvent(s, b->spec_ev, 0)); }
        } else {
            if (c == timed) SAC_HIP(hipEventRecord(t->ev_tm[0], q));
            if (launch_sample(b, t->Bt, m, pos * t->B, nullptr, q)) return -1;
            if (c == timed) SAC_HIP(hipEventRecord(t->ev_tm[1], q));
            if (launch_gather(b, b->d_idx + pos * t->B, t->B, m, b->d_slots + (size_t)pos * b->slot.slot_floats, b->slot, 1, q))
                return -1;
            if (c == timed) SAC_HIP(hipEventRecord(t->ev_tm[2], q));
        }
        // Chunk 0's two bookkeeping events go BEHIND its gather: an event record between two dependent launches of a stream
        // is a packet of its own there (the gather started 10.6 us after the draw had ended instead of ~2.5: rocprofv3
        // timeline of a 20-step call).  ev[0]: start of the device span (sac_last_loop_ms; it now excludes chunk 0's draw +
        // gather, ~12 us); b->ev[3]: the generator's state (see below -- chunk 1's draw has four steps of slack).
        // (a single-chunk call needs no b->ev[3]: nothing is drawn on the buffer's stream during the call, and the call
        //  returns behind a drained stream.  It must not leave a wait on the buffer's stream either: the NEXT call's
        //  hipStreamQuery then reports that stream busy and pays two runtime calls plus an in-stream wait in front of its
        //  first launch -- ~30 us of the first 20-step call behind single-step calls.)
        const bool more_chunks = m < n_steps;
        // (chunk 0 drawn on the trainer's stream: b->ev[3] tells the buffer's stream where the generator stands -- needed by a
        //  second chunk, and by the speculative chunk this call may leave for the next one)
        const bool will_speculate = spec_ok && !more_chunks;      // (a single-chunk call: the speculative draw follows chunk 0's, which ran on the trainer's stream)
        if (c == 0) { SAC_HIP(hipEventRecord(t->ev[0], s)); if (!use_spec && (more_chunks || will_speculate)) SAC_HIP(hipEventRecord(b->ev[3], q)); }
        if (c == 0) ht("chunk 0 draw + gather submitted");
        // (the call's last chunk on the buffer's stream: the speculative chunk goes right behind its gather and IN FRONT of
        //  the event the host polls below -- the buffer's stream then ends on a command the host has seen complete, and a
        //  device-wide synchronisation behind the call does not block on it: +16 us measured with the chunk at the very end)
        if (c > 0 && spec_ok && first + m == n_steps && launch_spec(pos + m)) return -1;
        if (c > 0) {
            // The chunk's slots must be gathered before its first step.  An in-stream wait for the buffer's stream costs the
            // trainer's stream ~5-10 us of idle time between two steps even when the gather finished long ago (a 20-step call
            // has one such boundary: 1.5 % of it).  The host is far ahead of the device, so it looks itself: once it has
            // SEEN the gather's event complete, the steps need no dependency at all.  (Bounded: a gather that is itself
            // waiting for slots takes the in-stream wait.)
            SAC_HIP(hipEventRecord(t->ev_ready[e], b->stream));
            const auto spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(400);
            hipError_t st;
            while ((st = hipEventQuery(t->ev_ready[e])) == hipErrorNotReady && std::chrono::steady_clock::now() < spin_until) { }
            if (st != hipSuccess) { (void)hipGetLastError(); SAC_HIP(hipStreamWaitEvent(s, t->ev_ready[e], 0)); }
        }
        for (int64_t i = 0; i < m; ++i) {
            t->publish_diag = (first + i == n_steps - 1);
            if (launch_step(t, b->d_slots + (size_t)(pos + i) * b->slot.slot_floats, b->slot, (int)(step0 + first + i), nullptr, step0 + first + i == 0)) return -1;
        }
        // the generator's state is one in-order sequence: everything later on the buffer's stream follows chunk 0's draw
        // (told to that stream only now, behind chunk 0's step launches: nothing of it is in front of the first step)
        if (c == 0 && !use_spec && (more_chunks || will_speculate)) SAC_HIP(hipStreamWaitEvent(b->stream, b->ev[3], 0));
        t->publish_diag = true;
        // "the trainer is done with this chunk's slots": only a call that wraps the slot ring ever asks (the next call
        // starts behind a drained stream) -- a short call keeps these records out of its stream (~2-5 us each)
        if (n_steps > LOOP_RING) {
            SAC_HIP(hipEventRecord(t->ev_done[e], s));
            SAC_REQUIRE(n_live < sac_trainer::NLOOP_EV + 4, "internal: loop chunk bookkeeping overflow");
            live[n_live++] = Live{pos, m, e};
        }
        pos += m;
        done += m;
    }
    SAC_HIP(hipEventRecord(t->ev[1], s));
    ht("all launches submitted");
    b->defer_mirror = false;
    host_rng_advance(b, t->Bt, b->deferred_batches);
    b->deferred_batches = 0;
    ht("generator mirrored on the host");
    b->loop_pos = (pos + sac_buffer::LOOP_SPEC <= LOOP_RING) ? pos : 0;
    if (spec_ok && !spec_launched && launch_spec(pos)) return -1;       // (single-chunk calls)
    b->loop_streak += 1;
    ht("next call's first chunk submitted");
    if (wait_trainer_stream(t, t->ev[1])) return -1;      // (the diagnostics are in mapped pinned memory: nothing to copy)
    ht("stream idle");
    // (the speculative chunk finished long ago, under this call's steps: ONE look at its stream lets the runtime know, so that
    //  a device-wide synchronisation behind this call does not go into a blocking wait for the buffer's stream: +16 us)
    if (b->spec_valid && hipEventQuery(b->spec_ev) != hipSuccess) (void)hipGetLastError();
    if (check_fused_abort(t, lost_out) < 0) return -1;
    t->timing_pending = true;                 // (the event intervals are read when sac_last_loop_ms asks: ~1 us each)
    t->mirror_valid = false;
    ht("return");
    return 0;
}

int sac_train_loop(sac_trainer_t *t, sac_buffer_t *b, int64_t n_steps, float *diag_first, float *diag_last) {
    SAC_REQUIRE(t && b && n_steps > 0 && n_steps < (1 << 30), "bad arguments to sac_train_loop");
    SAC_REQUIRE(b->device == t->device, "buffer and trainer live on different devices");
    SAC_REQUIRE(b->O == t->O && b->A == t->A, "buffer dims (%d,%d) do not match trainer dims (%d,%d)", b->O, b->A,
                t->O, t->A);
    SAC_HIP(hipSetDevice(t->device));
    if (t->fused && t->pend_n > 0) {          // device-batch steps nobody has verified yet: settle them first
        if (wait_trainer_stream(t)) return -1;
        if (recover_device_steps(t) < 0) return -1;
    }
    if (host_rng_sync_in(b)) return -1;            // (a bound generator: somebody else may have moved np.random)
    // (batches the stepwise interface drew ahead: the generator goes back first -- a speculative first chunk of THIS call,
    //  left by the previous loop call, is not an outside touch and stays: train_loop_run decides about it)
    if (b->ra_ahead > 0) { if (readahead_rollback(b)) return -1; }
    else b->ra_streak = 0;
    const MtState start = b->host_seen;            // the generator in front of the loop's first batch
    unsigned lost = 0;
    if (train_loop_run(t, b, n_steps, 0, &lost)) return -1;
    if (lost) {
        // The fused step gave up inside this call (a co-tenant on the GPU): the last `lost` steps were not applied and the
        // trainer is on the four-launch step now.  The generator goes back to the first unapplied step -- the state in
        // front of the loop advanced by the batches of the applied ones -- and the rest of the loop runs again: the caller
        // sees the same trajectory, index stream and final generator state as an undisturbed run.
        SAC_REQUIRE(lost <= (unsigned)n_steps, "internal: %u steps lost in a loop of %lld", lost, (long long)n_steps);
        const int64_t applied = n_steps - (int64_t)lost;
        b->spec_valid = false;                     // (whatever the failed pass left for a next call: the generator is reset)
        MtState st = start;
        host_rng_skip(b, st, t->Bt, applied);
        if (host_rng_adopt(b, st)) return -1;
        unsigned again = 0;
        if (train_loop_run(t, b, (int64_t)lost, applied, &again)) return -1;
    }
    if (diag_first) memcpy(diag_first, t->h_diag, sizeof(float) * SAC_DIAG_N);
    if (diag_last) memcpy(diag_last, t->h_diag + SAC_DIAG_N, sizeof(float) * SAC_DIAG_N);
    return 0;
}

int sac_profile_loop(sac_trainer_t *t, sac_buffer_t *b, int64_t n_steps, float out_ms[9]) {
    SAC_REQUIRE(t && b && n_steps > 0 && n_steps <= 4096 && out_ms, "bad arguments to sac_profile_loop");
    SAC_REQUIRE(t->algo == 0 && !t->gen, "sac_profile_loop instruments the SAC step of the fused kernels only");
    SAC_REQUIRE(b->device == t->device && b->O == t->O && b->A == t->A, "buffer does not match trainer");
    SAC_HIP(hipSetDevice(t->device));
    hipStream_t s = t->stream;
    t->dev.eps1 = t->dev.eps2 = nullptr;
    if (stage_batches(t, b, n_steps)) return -1;
    constexpr int NE = 7;            // e0 A e1 B e2 C e3 (nothing) e4 D e5 (nothing) e6
    std::vector<hipEvent_t> ev((size_t)n_steps * NE);
    for (auto &e : ev) SAC_HIP(hipEventCreate(&e));
    for (int64_t i = 0; i < n_steps; ++i) {
        t->publish_diag = (i == n_steps - 1);
        if (launch_step(t, b->d_slots + (size_t)i * b->slot.slot_floats, b->slot, (int)i, &ev[(size_t)i * NE])) return -1;
    }
    t->publish_diag = true;
    SAC_HIP(hipStreamSynchronize(s));
    if (check_fused_abort(t)) {
        sac::set_error("the fused step gave up during the profiling pass (is another process using this GPU?); the trainer "
                       "is on the four-launch step now: profile again");
        return -3;
    }
    // interval k = launch k between two event records; the empty interval e5->e6 measures what an
    // event pair costs by itself and is subtracted from the kernel intervals (slot 5 of out_ms, once
    // k_policy_bwd, is the second empty interval and reads 0)
    double acc[NE - 1] = {0, 0, 0, 0, 0, 0};
    for (int64_t i = 0; i < n_steps; ++i)
        for (int k = 0; k < NE - 1; ++k) {
            float ms = 0.f;
            SAC_HIP(hipEventElapsedTime(&ms, ev[(size_t)i * NE + k], ev[(size_t)i * NE + k + 1]));
            acc[k] += ms;
        }
    float tot = 0.f;
    SAC_HIP(hipEventElapsedTime(&tot, ev[0], ev[(size_t)n_steps * NE - 1]));
    for (auto &e : ev) (void)hipEventDestroy(e);
    SAC_HIP(hipEventElapsedTime(&out_ms[0], b->ev[0], b->ev[1]));
    SAC_HIP(hipEventElapsedTime(&out_ms[1], b->ev[1], b->ev[2]));
    const double empty = acc[NE - 2] / (double)n_steps;
    for (int k = 0; k < 5; ++k) {
        const double v = acc[k] / (double)n_steps - empty;
        out_ms[2 + k] = (float)(v > 0 ? v : 0);
    }
    out_ms[7] = (float)empty;
    out_ms[8] = tot;
    t->mirror_valid = false;
    return 0;
}

int sac_sync(sac_trainer_t *t) {
    SAC_REQUIRE(t, "null trainer");
    SAC_HIP(hipSetDevice(t->device));
    if (wait_trainer_stream(t)) return -1;
    return recover_device_steps(t) < 0 ? -1 : 0;
}

// Experiment (bench.py --replicas-per-gpu): confine this trainer's launches to the CUs of the XCDs in `xcd_mask` (bit k =
// XCD k).  The fused step needs all its 16 * NB workgroups resident at once: a confined trainer keeps it only if its
// CUs can hold them (batch 128 on half the chip) -- and then takes no part in the gate that serialises fused launches of
// different trainers: the CALLER promises that confined fused trainers own disjoint XCDs.  Otherwise it switches to the
// four-launch step.
extern "C" int sac_make_xcd_mask_stream(hipStream_t *out, unsigned xcd_mask);
int sac_trainer_set_xcd_mask(sac_trainer_t *t, unsigned xcd_mask) {
    SAC_REQUIRE(t != nullptr, "null trainer");
    SAC_HIP(hipSetDevice(t->device));
    SAC_HIP(hipStreamSynchronize(t->stream));
    hipStream_t ns = nullptr;
    if (sac_make_xcd_mask_stream(&ns, xcd_mask)) return -1;
    sac::stream_unregister(t->stream);       // (drained above: a buffer that remembers it finds no step in flight)
    SAC_HIP(hipStreamDestroy(t->stream));
    t->stream = ns;
    sac::stream_register(ns);
    t->pend_n = 0;
    int cus = 0;
    SAC_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, t->device));
    const int mine = (cus / 8) * __builtin_popcount(xcd_mask & 0xffu);
    if (t->fused) {
        FusedGate &G = g_gate[t->device & 63];
        std::lock_guard<std::mutex> lk(G.mu);
        if (!t->gate_exempt) G.live -= 1;
        if (t->abc_grid <= mine && (xcd_mask & 0xffu) != 0xffu) {
            t->gate_exempt = true;
        } else if ((xcd_mask & 0xffu) == 0xffu) {
            t->gate_exempt = false;
            G.live += 1;
        } else {
            t->gate_exempt = false;
            t->fused = false;
            set_abort_ptrs(t, nullptr);
        }
    }
    return 0;
}
int sac_trainer_set_xcd(sac_trainer_t *t, int xcd) {
    SAC_REQUIRE(xcd >= 0 && xcd < 8, "bad XCD index %d", xcd);
    return sac_trainer_set_xcd_mask(t, 1u << xcd);
}

// 1 while this trainer runs the fused two-launch step (k_abc + k_dw_adam), 0 for the four-launch step
int sac_trainer_is_fused(const sac_trainer_t *t) { return (t && t->fused) ? 1 : 0; }
int sac_trainer_step_kind(const sac_trainer_t *t) { return !t ? -1 : (t->gen ? 3 : (t->fused ? (t->chain_bwd ? 4 : 1) : (t->chain ? 2 : 0))); }

int sac_last_loop_ms(sac_trainer_t *t, float *total_ms, float *sample_ms, float *gather_ms, float *steps_ms) {
    SAC_REQUIRE(t, "null trainer");
    if (t->timing_pending) {
        t->timing_pending = false;
        SAC_HIP(hipSetDevice(t->device));
        SAC_HIP(hipEventElapsedTime(&t->last_ms[1], t->ev_tm[0], t->ev_tm[1]));
        SAC_HIP(hipEventElapsedTime(&t->last_ms[2], t->ev_tm[1], t->ev_tm[2]));
        SAC_HIP(hipEventElapsedTime(&t->last_ms[3], t->ev[0], t->ev[1]));
        t->last_ms[0] = t->last_ms[3];        // sampling + gathering of later chunks overlap the steps
    }
    if (total_ms) *total_ms = t->last_ms[0];
    if (sample_ms) *sample_ms = t->last_ms[1];
    if (gather_ms) *gather_ms = t->last_ms[2];
    if (steps_ms) *steps_ms = t->last_ms[3];
    return 0;
}

int64_t sac_debug_fetch(sac_trainer_t *t, const char *name, float *out, int64_t cap) {
    if (!t || !name || !out) { sac::set_error("bad arguments to sac_debug_fetch"); return -2; }
    if (hipSetDevice(t->device) != hipSuccess) { sac::set_error("hipSetDevice failed"); return -1; }
    if (t->gen) return gen_debug_fetch(t, std::string(name), out, cap);
    const int B = t->B, A = t->A;
    const Dev &d = t->dev;
    const std::string nm(name);
    auto fetch = [&](const float *src, int64_t n, std::vector<float> &h) -> int {
        h.resize((size_t)n);
        SAC_HIP(hipMemcpyAsync(h.data(), src, sizeof(float) * n, hipMemcpyDeviceToHost, t->stream));
        SAC_HIP(hipStreamSynchronize(t->stream));
        return 0;
    };
    std::vector<float> h;
    if (nm == "diag_trace") {
        const int64_t n = (int64_t)DIAG_TRACE_CAP * SAC_DIAG_N < cap ? (int64_t)DIAG_TRACE_CAP * SAC_DIAG_N : cap;
        if (fetch(d.diag_trace, n, h)) return -1;
        memcpy(out, h.data(), sizeof(float) * n);
        return n;
    }
    struct Row16 { const char *n; const float *p; };
    const Row16 r16[] = {{"a_new", d.anew}, {"mu", d.mu}, {"log_std", d.ls}, {"a_next", d.a2}, {"z", d.z}};
    for (auto &e : r16)
        if (nm == e.n) {
            const int Bt = t->Bt;           // (the rows of the true batch)
            if (cap < (int64_t)Bt * A) { sac::set_error("buffer too small"); return -2; }
            if (fetch(e.p, 16LL * Bt, h)) return -1;
            for (int b = 0; b < Bt; ++b) for (int a = 0; a < A; ++a) out[(size_t)b * A + a] = h[(size_t)b * 16 + a];
            return (int64_t)Bt * A;
        }
    struct Vec { const char *n; const float *p; };
    const Vec vecs[] = {{"log_pi", d.logpi}, {"log_pi_next", d.logpi2}, {"q1", d.q}, {"q2", d.q + B},
                        {"q1_new", d.q + 2 * (size_t)B}, {"q2_new", d.q + 3 * (size_t)B},
                        {"tq1", d.q + 4 * (size_t)B}, {"tq2", d.q + 5 * (size_t)B}, {"q_target", d.y}};
    for (auto &e : vecs)
        if (nm == e.n) {
            const int Bt = t->Bt;
            if (cap < Bt) { sac::set_error("buffer too small"); return -2; }
            if (fetch(e.p, Bt, h)) return -1;
            memcpy(out, h.data(), sizeof(float) * Bt);
            return Bt;
        }
    const char *gn[3] = {"g_policy", "g_qf1", "g_qf2"};
    for (int i = 0; i < 3; ++i)
        if (nm == gn[i]) {
            const int64_t n = sac_param_count(t, i);
            if (cap < n) { sac::set_error("buffer too small"); return -2; }
            if (download_padded(t, i, t->net[i].G, out)) return -1;
            return n;
        }
    sac::set_error("unknown debug tensor '%s'", name);
    return -2;
}

#ifdef SAC_STAMPS
int sac_fetch_stamps(sac_trainer_t *t, unsigned long long *out) {
    SAC_HIP(hipSetDevice(t->device));
    SAC_HIP(hipStreamSynchronize(t->stream));
    SAC_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(sac::g_stamps), sizeof(unsigned long long) * 5 * 512 * 16));
    return 0;
}
#endif

int sac_policy_mirror(sac_trainer_t *t) {
    SAC_REQUIRE(t, "null trainer");
    const int64_t n = sac_param_count(t, SAC_NET_POLICY);
    t->h_policy.resize((size_t)n);
    if (sac_get_params(t, SAC_NET_POLICY, t->h_policy.data(), n)) return -1;
    t->mirror_valid = true;
    return 0;
}

int sac_policy_act(sac_trainer_t *t, const float *obs, int deterministic, const float *eps, float *act) {
    SAC_REQUIRE(t && obs && act, "bad arguments to sac_policy_act");
    SAC_REQUIRE(deterministic || eps || t->algo == 1, "stochastic acting needs the N(0,1) draw (eps)");
    if (t->algo == 1) deterministic = 1;        // TanhMlpPolicy: tanh(last_fc); exploration noise is the caller's strategy
    if (!t->mirror_valid && sac_policy_mirror(t)) return -1;
    const int O = t->O, A = t->A;
    int hs[gen::GMAXL], nh = 2;
    hs[0] = t->HP[0]; hs[1] = t->HP[1];
    if (t->gen) { nh = t->gen->Lp; for (int i = 0; i < nh; ++i) hs[i] = t->gen->hp[i]; }
    // the flat vector: per hidden layer W [out][in] then b; last_fc (mean); last_fc_log_std (absent for TD3)
    const float *p = t->h_policy.data();
    thread_local std::vector<float> hin, hout;
    hin.assign(obs, obs + O);
    int dprev = O;
    for (int l = 0; l < nh; ++l) {
        const float *W = p, *b = W + (size_t)hs[l] * dprev;
        hout.resize((size_t)hs[l]);
        for (int n = 0; n < hs[l]; ++n) {
            float s = b[n];
            for (int k = 0; k < dprev; ++k) s += W[(size_t)n * dprev + k] * hin[(size_t)k];
            hout[(size_t)n] = s > 0.f ? s : 0.f;
        }
        hin.swap(hout);
        p = b + hs[l];
        dprev = hs[l];
    }
    const int H2 = dprev;
    const float *h2 = hin.data();
    const float *Wm = p, *bm = Wm + (size_t)A * H2, *Ws = bm + A, *bs = Ws + (size_t)A * H2;
    for (int a = 0; a < A; ++a) {
        float m = bm[a], ls = 0.f;
        for (int k = 0; k < H2; ++k) m += Wm[(size_t)a * H2 + k] * h2[k];
        if (deterministic) act[a] = tanhf(m);
        else {
            ls = bs[a];
            for (int k = 0; k < H2; ++k) ls += Ws[(size_t)a * H2 + k] * h2[k];
            ls = fminf(fmaxf(ls, LOG_SIG_MIN), LOG_SIG_MAX);
            act[a] = tanhf(m + expf(ls) * eps[a]);
        }
    }
    return 0;
}

}  // extern "C"
