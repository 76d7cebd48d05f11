// SAC gradient step on MI355X (gfx950): five row-block / tile-owner kernels per step.
//
// Replaces rlkit SACTrainer.train_from_torch (+ np_to_pytorch_batch, soft_update_from_to); call
// sites /root/reference/util/rlkit_utils.py:64-106, /root/reference/util/rlkit_custom.py:238.
// Normative step order: SURVEY.md Appendix A (lines 1-18), ordering O1, logging quirk Q1.
//
// Decomposition (DESIGN.md "Kernels"): batch rows are independent through forward and backward-dX,
// so a workgroup owns a 16-row block (one MFMA 16x16x4 M-tile) and chains whole layers through LDS;
// the only all-to-all seams are (a) mean(log_pi) -> alpha, (b) min over twin nets, (c) the weight
// gradient's contraction over the batch.  Each seam is a kernel boundary (cheaper than an in-launch
// grid barrier on 8 XCDs); the weight-gradient kernel is tile-owner parallel and applies Adam and
// the Polyak update in its epilogue, so gradients never round-trip through HBM.
//
//   K1 k_policy_fwd   2*B/16 WGs   pi(s), pi(s') : 3 layers + tanh-Gaussian head, log_pi
//   K2 k_q_fwd        6*B/16 WGs   Q1,Q2 on (s,a), (s,a_new); T1,T2 on (s',a')  (+ alpha Adam step)
//   K3 k_q_bwd        4*B/16 WGs   critic dL/dh (2 nets), actor dQ/da (2 nets)
//   K4 k_policy_bwd     B/16 WGs   head gradient (reparameterised), dL/dh
//   K5 k_dw_adam      ~250  WGs    dW = dY^T X over the batch (MFMA), Adam, Polyak, diagnostics
#include "sac_common.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace sac {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int H = 256;            // hidden width (every shipped variant.json)
constexpr float LOG_SIG_MAX = 2.0f, LOG_SIG_MIN = -20.0f, TANH_EPS = 1e-6f;
constexpr float ADAM_B1 = 0.9f, ADAM_B2 = 0.999f, ADAM_EPS = 1e-8f;
constexpr int DIAG_TRACE_CAP = 4096;

struct Ctl {                       // device-resident step state (read by every kernel at entry)
    long long n_train_steps_total; // rlkit _n_train_steps_total
    long long adam_t;              // optimizer step count (all four optimizers step together)
    int loop_pos;                  // step index inside the current sac_train_loop / 0 for sac_step
    unsigned ticket;               // K5 arrival counter
    float log_alpha, a_m, a_v, alpha, alpha_loss;
    int pad[3];
};

struct Layer {                     // one nn.Linear in the padded device layout
    int N, K, Np, Kp;
    long long offW, offB;          // in P / M / V / G : W [Np][Kp], b [Np]
    long long offWt;               // in PT : W^T [Kp + 16][Np]
};

struct Net {
    float *P = nullptr, *M = nullptr, *V = nullptr, *PT = nullptr, *G = nullptr;
    long long nP = 0, nPT = 0;
    Layer L[3];
};

// everything the step kernels need, passed by value
struct Dev {
    int B, O, A, KP, KQ, NH, NB;   // KP = pad16(O), KQ = pad16(O+A), NH = pad16(2A), NB = B/16
    float discount, reward_scale, tau, target_entropy, alpha_lr;
    int period, auto_alpha;
    unsigned long long noise_seed;
    Ctl *ctl;
    // nets: 0 policy, 1 qf1, 2 qf2, 3 tqf1, 4 tqf2
    const float *P[5];
    const float *PT[3];
    Layer LP[3], LQ[3];
    // policy activations (s rows) feature-major [256][B]; per-row head values row-major [B][16]
    float *PH1T, *PH2T, *mu, *ls, *lsok, *z, *anew, *epsv, *logpi, *a2, *logpi2, *part_logpi;
    // Q forward: passes 0..3 keep h1/h2 feature-major; q values for all 6 passes
    float *QH1T, *QH2T, *q;
    // backward
    float *y, *dq16T, *dQH2T, *dQH1T, *da, *dheadT, *dPH2T, *dPH1T;
    // diagnostics
    float *diag_first, *diag_last, *diag_trace;
    // caller-supplied noise (NULL => counter-based device stream)
    const float *eps1, *eps2;
};

struct DwJob {
    const float *dYT, *XT;         // rows n0.. of dY^T [.][B]; rows k0.. of X^T [.][B]
    float *P, *M, *V, *PT, *G;     // layer bases (W part)
    float *bias, *mb, *vb, *gb;    // non-null only for the k0 == 0 strip
    float *TP, *Tbias;             // Polyak target (W base / bias), or null
    int N, K, n0, k0, ldp, ldt;
    float lr;
    int xt_from_slot;              // XT is an offset into the current minibatch slot (saT)
    long long xt_off;
};

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int lds_off(int row, int k, int KL) {
    return row * KL + ((((k >> 2) ^ (row & 15)) << 2) | (k & 3));
}

// acc[t] += X[16 x 16*KS] * W[cols n_base + t*n_stride .. +16][.]^T  for this wave's NT column tiles.
// X: LDS row-block, row stride KL (multiple of 64), 16-B chunks XOR-swizzled by row.
// W: global [n][ldw] (ldw multiple of 4).  Contraction index order is permuted (lane group g owns
// k = 16S + 4g + i), identically for both operands.
template <int NT>
__device__ __forceinline__ void gemm_tiles(const float *X, int KL, int S0, int S1, const float *__restrict__ W,
                                           int ldw, int n_base, int n_stride, f32x4 (&acc)[NT]) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    const float *xrow = X + r * KL;
    const float *wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wp[t] = W + (size_t)(n_base + t * n_stride + r) * ldw + 4 * g;
#pragma unroll 2
    for (int S = S0; S < S1; ++S) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(xrow + 4 * ((4 * S + g) ^ r));
        f32x4 b[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t] = *reinterpret_cast<const f32x4 *>(wp[t] + 16 * S);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[t][i], acc[t], 0, 0, 0);
        }
    }
}

// one [16 x 16*NTT] output with the contraction split over the 4 waves; result (summed) lands in
// LDS `out` as row-major [16][ldo] (+ bias).  `red` = 4*NTT*256 floats of scratch.
template <int NTT>
__device__ __forceinline__ void gemm_splitk(const float *X, int KL, int KS, const float *__restrict__ W, int ldw,
                                            const float *__restrict__ bias, float *red, float *out, int ldo) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x4 acc[NTT];
#pragma unroll
    for (int t = 0; t < NTT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int per = (KS + 3) >> 2;
    const int s0 = wave * per, s1 = (s0 + per < KS) ? s0 + per : KS;
    if (s0 < s1) gemm_tiles<NTT>(X, KL, s0, s1, W, ldw, 0, 16, acc);
#pragma unroll
    for (int t = 0; t < NTT; ++t)
        *reinterpret_cast<f32x4 *>(red + ((wave * NTT + t) * 64 + lane) * 4) = acc[t];
    __syncthreads();
    for (int e = threadIdx.x; e < NTT * 256; e += 256) {
        const int t = e >> 8, l = (e >> 2) & 63, i = e & 3;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[((w * NTT + t) * 64 + l) * 4 + i];
        const int row = 4 * (l >> 4) + i, col = 16 * t + (l & 15);
        out[row * ldo + col] = s + (bias ? bias[col] : 0.f);
    }
    __syncthreads();
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

// fill a swizzled LDS row-block [16][KL] from row-major global rows (two sources concatenated)
__device__ __forceinline__ void load_rows_cat(float *X, int KL, int Kfill, const float *__restrict__ s0, int n0,
                                              int ld0, const float *__restrict__ s1, int n1, int ld1) {
    for (int e = threadIdx.x; e < RB * Kfill; e += 256) {
        const int row = e / Kfill, k = e - row * Kfill;
        float v = 0.f;
        if (k < n0) v = s0[row * ld0 + k];
        else if (k < n0 + n1) v = s1[row * ld1 + (k - n0)];
        X[lds_off(row, k, KL)] = v;
    }
}

// epilogue of a hidden layer: bias + relu from the accumulators into the next LDS row-block, and
// (optionally) the feature-major copy [n][B] for the weight-gradient kernel.
template <int NT>
__device__ __forceinline__ void hidden_epilogue(const f32x4 (&acc)[NT], int n_base, int n_stride,
                                                const float *__restrict__ bias, float *Xn, int KL, float *outT,
                                                int B, int row0) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n_base + t * n_stride + c;
        const float bv = bias[n];
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = fmaxf(acc[t][i] + bv, 0.f);
            Xn[lds_off(4 * g + i, n, KL)] = v[i];
        }
        if (outT) *reinterpret_cast<f32x4 *>(outT + (size_t)n * B + row0 + 4 * g) = v;
    }
}

// ------------------------------------------------------------------------------------------
// K1: policy forward on s (blocks [0,NB)) and s' (blocks [NB,2NB))
// ------------------------------------------------------------------------------------------
template <int NTH>
__global__ __launch_bounds__(256) void k_policy_fwd(Dev d, const float *__restrict__ slots, SlotLayout SL,
                                                    int n_slots) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int B = d.B, O = d.O, A = d.A;
    const int KL0 = (d.KP + 63) & ~63;
    float *X0 = lds;                     // [16][KL0]
    float *X1 = X0 + RB * KL0;           // [16][256]
    float *X2 = X1 + RB * H;             // [16][256]
    float *HD = X2 + RB * H;             // [16][NH]
    float *red = HD + RB * 32;           // split-K scratch 4*NTH*256
    const Ctl ctl = *d.ctl;
    const float *S = slots + (size_t)(ctl.loop_pos % n_slots) * SL.slot_floats;
    const bool next = blockIdx.x >= (unsigned)d.NB;
    const int rb = next ? blockIdx.x - d.NB : blockIdx.x;
    const int row0 = rb * RB;
    const int wave = threadIdx.x >> 6;
    const float *P = d.P[0];

    load_rows_cat(X0, KL0, d.KP, S + (next ? SL.off_nobs : SL.off_obs) + (size_t)row0 * O, O, O, nullptr, 0, 0);
    __syncthreads();
    {
        f32x4 acc[4] = {};
        gemm_tiles<4>(X0, KL0, 0, d.KP >> 4, P + d.LP[0].offW, d.LP[0].Kp, 64 * wave, 16, acc);
        hidden_epilogue<4>(acc, 64 * wave, 16, P + d.LP[0].offB, X1, H, next ? nullptr : d.PH1T, B, row0);
    }
    __syncthreads();
    {
        f32x4 acc[4] = {};
        gemm_tiles<4>(X1, H, 0, H >> 4, P + d.LP[1].offW, H, 64 * wave, 16, acc);
        hidden_epilogue<4>(acc, 64 * wave, 16, P + d.LP[1].offB, X2, H, next ? nullptr : d.PH2T, B, row0);
    }
    __syncthreads();
    gemm_splitk<NTH>(X2, H, H >> 4, P + d.LP[2].offW, H, P + d.LP[2].offB, red, HD, 32);

    // tanh-Gaussian head: thread = (row, a)
    const int row = threadIdx.x >> 4, a = threadIdx.x & 15;
    const int grow = row0 + row;
    float lp = 0.f;
    if (a < A) {
        const float mean = HD[row * 32 + a];
        const float raw = HD[row * 32 + A + a];
        const float lstd = fminf(fmaxf(raw, LOG_SIG_MIN), LOG_SIG_MAX);
        const float stdv = expf(lstd);
        const float *epp = next ? d.eps2 : d.eps1;
        const float eps = epp ? epp[grow * A + a]
                              : philox_normal(d.noise_seed, (unsigned long long)ctl.n_train_steps_total,
                                              (unsigned)(grow * 16 + a), next ? 1u : 0u);
        const float zz = __fadd_rn(mean, __fmul_rn(stdv, eps));          // TanhNormal.rsample
        const float act = tanhf(zz);
        const float dd = __fsub_rn(zz, mean);                            // Normal.log_prob(z)
        const float var = __fmul_rn(stdv, stdv);
        const float nlp = -(dd * dd) / (2.0f * var) - logf(stdv) - 0.91893853320467274178f;
        lp = nlp - logf(1.0f - act * act + TANH_EPS);
        if (!next) {
            d.mu[grow * 16 + a] = mean;
            d.ls[grow * 16 + a] = lstd;
            d.lsok[grow * 16 + a] = (raw >= LOG_SIG_MIN && raw <= LOG_SIG_MAX) ? 1.f : 0.f;
            d.z[grow * 16 + a] = zz;
            d.anew[grow * 16 + a] = act;
            d.epsv[grow * 16 + a] = eps;
        } else {
            d.a2[grow * 16 + a] = act;
        }
    } else if (!next) {
        d.anew[grow * 16 + a] = 0.f;
    } else {
        d.a2[grow * 16 + a] = 0.f;
    }
    const float lsum = group16_sum(lp);
    if (a == 0) (next ? d.logpi2 : d.logpi)[grow] = lsum;
    if (!next) {
        // block partial of sum(log_pi) in a fixed order (deterministic alpha)
        __syncthreads();
        if (a == 0) red[row] = lsum;
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = 0.f;
            for (int i = 0; i < RB; ++i) s += red[i];
            d.part_logpi[rb] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------
// K2: six Q forward passes.  pass = blockIdx / NB:
//   0 Q1(s,a) 1 Q2(s,a) 2 Q1(s,a_new) 3 Q2(s,a_new) 4 T1(s',a') 5 T2(s',a')
// Block 0 also performs the alpha Adam step (SURVEY Appendix A lines 4-6).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_q_fwd(Dev d, const float *__restrict__ slots, SlotLayout SL, int n_slots) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int B = d.B, O = d.O, A = d.A;
    const int KL0 = (d.KQ + 63) & ~63;
    float *X0 = lds;
    float *X1 = X0 + RB * KL0;
    float *X2 = X1 + RB * H;
    const Ctl ctl = *d.ctl;
    const float *S = slots + (size_t)(ctl.loop_pos % n_slots) * SL.slot_floats;
    const int pass = blockIdx.x / d.NB, rb = blockIdx.x - pass * d.NB;
    const int row0 = rb * RB;
    const int wave = threadIdx.x >> 6;
    const int net = (pass < 4) ? 1 + (pass & 1) : 3 + (pass & 1);
    const float *P = d.P[net];

    if (blockIdx.x == 0 && threadIdx.x == 0) {
        Ctl *c = d.ctl;
        if (d.auto_alpha) {
            float s = 0.f;
            for (int i = 0; i < d.NB; ++i) s += d.part_logpi[i];
            const float mean_lp = s / (float)B + d.target_entropy;      // mean(log_pi + H)
            const float la = c->log_alpha;
            // torch: -(log_alpha * x).mean(); the mean's running sum starts at +0, so log_alpha == 0 logs -0.0
            c->alpha_loss = -((la * mean_lp) + 0.0f);
            const float gr = -mean_lp;                                   // d alpha_loss / d log_alpha
            const double t = (double)(ctl.adam_t + 1);
            const float m = c->a_m + (1.0f - ADAM_B1) * (gr - c->a_m);
            const float v = c->a_v * ADAM_B2 + (1.0f - ADAM_B2) * gr * gr;
            const double bc1 = 1.0 - pow((double)ADAM_B1, t), bc2 = 1.0 - pow((double)ADAM_B2, t);
            const float step_size = (float)((double)d.alpha_lr / bc1);
            const float denom = sqrtf(v) / (float)sqrt(bc2) + ADAM_EPS;
            const float nla = la + (-step_size * m) / denom;
            c->a_m = m; c->a_v = v; c->log_alpha = nla;
            c->alpha = expf(nla);
        } else {
            c->alpha = 1.0f;
            c->alpha_loss = 0.0f;
        }
    }

    const float *obs = S + ((pass >= 4) ? SL.off_nobs : SL.off_obs) + (size_t)row0 * O;
    const float *act;
    int lda;
    if (pass < 2) { act = S + SL.off_act + (size_t)row0 * A; lda = A; }
    else if (pass < 4) { act = d.anew + (size_t)row0 * 16; lda = 16; }
    else { act = d.a2 + (size_t)row0 * 16; lda = 16; }
    load_rows_cat(X0, KL0, d.KQ, obs, O, O, act, A, lda);
    __syncthreads();
    float *h1T = (pass < 4) ? d.QH1T + (size_t)pass * H * B : nullptr;
    float *h2T = (pass < 4) ? d.QH2T + (size_t)pass * H * B : nullptr;
    {
        f32x4 acc[4] = {};
        gemm_tiles<4>(X0, KL0, 0, d.KQ >> 4, P + d.LQ[0].offW, d.LQ[0].Kp, 64 * wave, 16, acc);
        hidden_epilogue<4>(acc, 64 * wave, 16, P + d.LQ[0].offB, X1, H, h1T, B, row0);
    }
    __syncthreads();
    {
        f32x4 acc[4] = {};
        gemm_tiles<4>(X1, H, 0, H >> 4, P + d.LQ[1].offW, H, 64 * wave, 16, acc);
        hidden_epilogue<4>(acc, 64 * wave, 16, P + d.LQ[1].offB, X2, H, h2T, B, row0);
    }
    __syncthreads();
    // last_fc: q[row] = h2[row] . w3 + b3   (N = 1: VALU dot, 16 lanes per row)
    const int row = threadIdx.x >> 4, part = threadIdx.x & 15;
    const float *w3 = P + d.LQ[2].offW;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int k = part + 16 * j;
        s += X2[lds_off(row, k, H)] * w3[k];
    }
    s = group16_sum(s);
    if (part == 0) d.q[(size_t)pass * B + row0 + row] = s + P[d.LQ[2].offB];
}

// ------------------------------------------------------------------------------------------
// K3: Q backward.  pass 0/1: critic Q1/Q2 (dL/dh kept for dW); pass 2/3: actor path through
// Q1/Q2 down to d/da_new (input gradient only).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_q_bwd(Dev d, const float *__restrict__ slots, SlotLayout SL, int n_slots) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int B = d.B;
    float *X2 = lds;                 // dL/dh2 row-block [16][256]
    float *X1 = X2 + RB * H;         // dL/dh1 row-block (actor)
    float *red = X1 + RB * H;        // 1024 floats
    __shared__ float s_dq[RB];
    const Ctl ctl = *d.ctl;
    const float *S = slots + (size_t)(ctl.loop_pos % n_slots) * SL.slot_floats;
    const int pass = blockIdx.x / d.NB, rb = blockIdx.x - pass * d.NB;
    const int row0 = rb * RB;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool critic = pass < 2;
    const int qi = pass & 1;               // which twin
    const float *P = d.P[1 + qi];
    const float *PT = d.PT[1 + qi];
    const float invB = 1.0f / (float)B;

    if (threadIdx.x < RB) {
        const int r = row0 + threadIdx.x;
        float dq;
        if (critic) {
            const float tmin = fminf(d.q[4 * (size_t)B + r], d.q[5 * (size_t)B + r]);
            const float tq = tmin - ctl.alpha * d.logpi2[r];
            const float yv = d.reward_scale * S[SL.off_rew + r] + (1.0f - S[SL.off_term + r]) * d.discount * tq;
            if (qi == 0) d.y[r] = yv;
            dq = 2.0f * (d.q[(size_t)qi * B + r] - yv) * invB;
            d.dq16T[(size_t)qi * 16 * B + r] = dq;                  // row 0 of the padded [16][B]
        } else {
            const float mine = d.q[(size_t)(2 + qi) * B + r], other = d.q[(size_t)(3 - qi) * B + r];
            const float sel = (mine < other) ? 1.0f : ((mine == other) ? 0.5f : 0.0f);   // torch.min backward
            dq = -invB * sel;
        }
        s_dq[threadIdx.x] = dq;
    }
    __syncthreads();
    // dL/dh2 = dq * w3 * relu'(h2)   (thread = feature k, 4-row groups)
    {
        const float *h2T = d.QH2T + (size_t)pass * H * B;
        const float *w3 = P + d.LQ[2].offW;
        float *outT = critic ? d.dQH2T + (size_t)qi * H * B : nullptr;
        const int k = threadIdx.x;
        const float wk = w3[k];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const f32x4 hv = *reinterpret_cast<const f32x4 *>(h2T + (size_t)k * B + row0 + 4 * qd);
            f32x4 gv;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                gv[i] = (hv[i] > 0.f) ? s_dq[4 * qd + i] * wk : 0.f;
                X2[lds_off(4 * qd + i, k, H)] = gv[i];
            }
            if (outT) *reinterpret_cast<f32x4 *>(outT + (size_t)k * B + row0 + 4 * qd) = gv;
        }
    }
    __syncthreads();
    // dL/dh1 = (dL/dh2 . W2) * relu'(h1)
    {
        f32x4 acc[4] = {};
        gemm_tiles<4>(X2, H, 0, H >> 4, PT + d.LQ[1].offWt, H, 64 * wave, 16, acc);
        const float *h1T = d.QH1T + (size_t)pass * H * B;
        float *outT = critic ? d.dQH1T + (size_t)qi * H * B : nullptr;
        const int c = lane & 15, g = lane >> 4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n = 64 * wave + 16 * t + c;
            const f32x4 hv = *reinterpret_cast<const f32x4 *>(h1T + (size_t)n * B + row0 + 4 * g);
            f32x4 gv;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                gv[i] = (hv[i] > 0.f) ? acc[t][i] : 0.f;
                if (!critic) X1[lds_off(4 * g + i, n, H)] = gv[i];
            }
            if (outT) *reinterpret_cast<f32x4 *>(outT + (size_t)n * B + row0 + 4 * g) = gv;
        }
    }
    if (critic) return;
    __syncthreads();
    // d/da_new = dL/dh1 . W1[:, O:O+A]   -> da[qi][row][16]
    gemm_splitk<1>(X1, H, H >> 4, PT + d.LQ[0].offWt + (size_t)d.O * H, H, nullptr, red, X2, 16);
    d.da[(size_t)qi * B * 16 + (size_t)row0 * 16 + threadIdx.x] = X2[threadIdx.x];
}

// ------------------------------------------------------------------------------------------
// K4: policy backward (actor loss = mean(alpha*log_pi - min Q)), analytic head gradient.
//   dL/dz      = da*(1-a^2) + (alpha/B) * 2a(1-a^2)/(1-a^2+1e-6)
//   dL/dmu     = dL/dz                      (the Normal terms cancel exactly under rsample)
//   dL/dlogstd = dL/dz * std*eps - alpha/B  (masked by the clamp)
// ------------------------------------------------------------------------------------------
template <int NTH>
__global__ __launch_bounds__(256) void k_policy_bwd(Dev d) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int B = d.B, A = d.A;
    float *XH = lds;                 // [16][64] head gradient row-block
    float *X2 = XH + RB * 64;        // [16][256]
    const Ctl ctl = *d.ctl;
    const int rb = blockIdx.x, row0 = rb * RB;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float *PT = d.PT[0];
    const float invB = 1.0f / (float)B;

    for (int e = threadIdx.x; e < RB * 64; e += 256) XH[e] = 0.f;
    __syncthreads();
    {
        const int row = threadIdx.x >> 4, a = threadIdx.x & 15;
        const int gi = (row0 + row) * 16 + a;
        if (a < A) {
            const float act = d.anew[gi];
            const float om = 1.0f - act * act;
            const float dav = d.da[gi] + d.da[(size_t)B * 16 + gi];
            const float dz = dav * om + (ctl.alpha * invB) * (2.0f * act * om / (om + TANH_EPS));
            const float stdv = expf(d.ls[gi]);
            const float dls = (dz * stdv * d.epsv[gi] - ctl.alpha * invB) * d.lsok[gi];
            XH[lds_off(row, a, 64)] = dz;
            XH[lds_off(row, A + a, 64)] = dls;
            d.dheadT[(size_t)a * B + row0 + row] = dz;
            d.dheadT[(size_t)(A + a) * B + row0 + row] = dls;
        }
    }
    __syncthreads();
    const int c = lane & 15, g = lane >> 4;
    {
        f32x4 acc[4] = {};
        gemm_tiles<4>(XH, 64, 0, NTH, PT + d.LP[2].offWt, d.LP[2].Np, 64 * wave, 16, acc);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n = 64 * wave + 16 * t + c;
            const f32x4 hv = *reinterpret_cast<const f32x4 *>(d.PH2T + (size_t)n * B + row0 + 4 * g);
            f32x4 gv;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                gv[i] = (hv[i] > 0.f) ? acc[t][i] : 0.f;
                X2[lds_off(4 * g + i, n, H)] = gv[i];
            }
            *reinterpret_cast<f32x4 *>(d.dPH2T + (size_t)n * B + row0 + 4 * g) = gv;
        }
    }
    __syncthreads();
    {
        f32x4 acc[4] = {};
        gemm_tiles<4>(X2, H, 0, H >> 4, PT + d.LP[1].offWt, H, 64 * wave, 16, acc);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n = 64 * wave + 16 * t + c;
            const f32x4 hv = *reinterpret_cast<const f32x4 *>(d.PH1T + (size_t)n * B + row0 + 4 * g);
            f32x4 gv;
#pragma unroll
            for (int i = 0; i < 4; ++i) gv[i] = (hv[i] > 0.f) ? acc[t][i] : 0.f;
            *reinterpret_cast<f32x4 *>(d.dPH1T + (size_t)n * B + row0 + 4 * g) = gv;
        }
    }
}

// ------------------------------------------------------------------------------------------
// K5: weight gradients + Adam + Polyak, tile-owner parallel.  One WG owns a 16 (out) x 64 (in)
// tile of one layer: dW = sum_b dY[b][n] X[b][k] with the batch split over the 4 waves (MFMA
// 16x16x4, both operands read feature-major so every lane load is 16 B), reduced through LDS,
// then the owner applies torch.optim.Adam's update in place, refreshes the transposed copy and
// (every target_update_period steps) the Polyak average of the target net.  The extra last block
// computes the step's diagnostics.  The last block to arrive advances the step counters.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void adam_update(float &p, float &m, float &v, float g, float step_size, float bc2s) {
    m = m + (1.0f - ADAM_B1) * (g - m);
    v = v * ADAM_B2 + (1.0f - ADAM_B2) * g * g;
    const float denom = sqrtf(v) / bc2s + ADAM_EPS;
    p = p + (-step_size * m) / denom;
}

__device__ void block_stats(const float *x, int n, int stride, int width, float *out4, double *sh) {
    // Mean / population Std / Max / Min over x[i*stride + j], i<n, j<width
    double s = 0.0, s2 = 0.0;
    float mx = -INFINITY, mn = INFINITY;
    const int total = n * width;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int i = e / width, j = e - i * width;
        const float v = x[i * stride + j];
        s += v; s2 += (double)v * v;
        mx = fmaxf(mx, v); mn = fminf(mn, v);
    }
    __syncthreads();
    sh[threadIdx.x] = s; sh[256 + threadIdx.x] = s2; sh[512 + threadIdx.x] = mx; sh[768 + threadIdx.x] = mn;
    __syncthreads();
    if (threadIdx.x == 0) {
        double S = 0, S2 = 0, MX = -INFINITY, MN = INFINITY;
        for (int i = 0; i < 256; ++i) {
            S += sh[i]; S2 += sh[256 + i];
            MX = fmax(MX, sh[512 + i]); MN = fmin(MN, sh[768 + i]);
        }
        const double mean = S / total;
        double var = S2 / total - mean * mean;
        if (var < 0) var = 0;
        out4[0] = (float)mean; out4[1] = (float)sqrt(var); out4[2] = (float)MX; out4[3] = (float)MN;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_dw_adam(Dev d, const DwJob *__restrict__ jobs, int njobs,
                                                 const float *__restrict__ slots, SlotLayout SL, int n_slots) {
    __shared__ __attribute__((aligned(16))) float red[4 * 4 * 64 * 4];   // 16 KB (also diag scratch)
    __shared__ float redb[4 * 16];
    __shared__ float s_sc[2];
    __shared__ unsigned s_last;
    const int B = d.B;
    const Ctl ctl = *d.ctl;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    const float *S = slots + (size_t)(ctl.loop_pos % n_slots) * SL.slot_floats;

    if ((int)blockIdx.x < njobs) {
        const DwJob J = jobs[blockIdx.x];
        if (threadIdx.x == 0) {
            const double t = (double)(ctl.adam_t + 1);
            const double bc1 = 1.0 - pow((double)ADAM_B1, t), bc2 = 1.0 - pow((double)ADAM_B2, t);
            s_sc[0] = (float)((double)J.lr / bc1);
            s_sc[1] = (float)sqrt(bc2);
        }
        const float *XT = (J.xt_from_slot ? S + J.xt_off : J.XT) + (size_t)J.k0 * B;
        const float *YT = J.dYT + (size_t)J.n0 * B;
        f32x4 acc[4] = {};
        float bsum = 0.f;
        const int per = (B / 16) / 4;                 // 16-row chunks of the batch per wave
        const int rem = (B / 16) - 4 * per;
        const int s0 = wave * per + (wave < rem ? wave : rem);
        const int s1 = s0 + per + (wave < rem ? 1 : 0);
        const float *yp = YT + (size_t)r * B + 4 * g;
        const float *xp = XT + (size_t)r * B + 4 * g;
        for (int sI = s0; sI < s1; ++sI) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(yp + 16 * sI);
            f32x4 b[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) b[t] = *reinterpret_cast<const f32x4 *>(xp + (size_t)16 * t * B + 16 * sI);
            bsum += (a[0] + a[1]) + (a[2] + a[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[t][i], acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4 *>(red + ((wave * 4 + t) * 64 + lane) * 4) = acc[t];
        bsum += __shfl_xor(bsum, 16);
        bsum += __shfl_xor(bsum, 32);
        if (g == 0) redb[wave * 16 + r] = bsum;
        __syncthreads();
        const float step_size = s_sc[0], bc2s = s_sc[1];
        const bool polyak = (J.TP != nullptr) && (ctl.n_train_steps_total % d.period == 0);
        // owner of tile t == wave: lane (c = r, g) holds rows n0+4g+i, col k0 + 16*wave + c
        {
            f32x4 gsum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) gsum += *reinterpret_cast<const f32x4 *>(red + ((w * 4 + wave) * 64 + lane) * 4);
            const int k = J.k0 + 16 * wave + r;
            f32x4 pn = {0.f, 0.f, 0.f, 0.f};
            bool anyv = false;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = J.n0 + 4 * g + i;
                if (n < J.N && k < J.K) {
                    const size_t o = (size_t)n * J.ldp + k;
                    float p = J.P[o], m = J.M[o], v = J.V[o];
                    adam_update(p, m, v, gsum[i], step_size, bc2s);
                    J.P[o] = p; J.M[o] = m; J.V[o] = v;
                    if (J.G) J.G[o] = gsum[i];
                    if (polyak) J.TP[o] = J.TP[o] * (1.0f - d.tau) + p * d.tau;
                    pn[i] = p;
                    anyv = true;
                }
            }
            if (J.PT && anyv && k < J.K)
                *reinterpret_cast<f32x4 *>(J.PT + (size_t)k * J.ldt + J.n0 + 4 * g) = pn;
        }
        if (J.bias && threadIdx.x < 16) {
            const int n = J.n0 + threadIdx.x;
            if (n < J.N) {
                const float gb = (redb[threadIdx.x] + redb[16 + threadIdx.x]) + (redb[32 + threadIdx.x] + redb[48 + threadIdx.x]);
                float p = J.bias[n], m = J.mb[n], v = J.vb[n];
                adam_update(p, m, v, gb, step_size, bc2s);
                J.bias[n] = p; J.mb[n] = m; J.vb[n] = v;
                if (J.gb) J.gb[n] = gb;
                if (polyak) J.Tbias[n] = J.Tbias[n] * (1.0f - d.tau) + p * d.tau;
            }
        }
    } else {
        // ---- diagnostics block (SURVEY Appendix A line 17) ----
        double *sh = reinterpret_cast<double *>(red);      // 1024 doubles = 8 KB
        __shared__ float dg[SAC_DIAG_N];
        if (threadIdx.x < SAC_DIAG_N) dg[threadIdx.x] = 0.f;
        __syncthreads();
        // losses
        double l1 = 0, l2 = 0, lpl = 0, lal = 0;
        for (int i = threadIdx.x; i < B; i += 256) {
            const float yv = d.y[i];
            const float e1 = d.q[i] - yv, e2 = d.q[(size_t)B + i] - yv;
            l1 += (double)e1 * e1; l2 += (double)e2 * e2;
            const float qn = fminf(d.q[2 * (size_t)B + i], d.q[3 * (size_t)B + i]);
            lpl += (double)(d.logpi[i] - qn);
            lal += (double)(ctl.alpha * d.logpi[i] - qn);
        }
        sh[threadIdx.x] = l1; sh[256 + threadIdx.x] = l2; sh[512 + threadIdx.x] = lpl; sh[768 + threadIdx.x] = lal;
        __syncthreads();
        if (threadIdx.x < 4) {
            double s = 0;
            for (int i = 0; i < 256; ++i) s += sh[256 * threadIdx.x + i];
            dg[SAC_D_QF1_LOSS + threadIdx.x] = (float)(s / B);
        }
        __syncthreads();
        block_stats(d.q, B, 1, 1, dg + SAC_D_Q1_MEAN, sh);
        block_stats(d.q + B, B, 1, 1, dg + SAC_D_Q2_MEAN, sh);
        block_stats(d.y, B, 1, 1, dg + SAC_D_QT_MEAN, sh);
        block_stats(d.logpi, B, 1, 1, dg + SAC_D_LOGPI_MEAN, sh);
        block_stats(d.mu, B, 16, d.A, dg + SAC_D_MU_MEAN, sh);
        block_stats(d.ls, B, 16, d.A, dg + SAC_D_LOGSTD_MEAN, sh);
        if (threadIdx.x == 0) { dg[SAC_D_ALPHA] = ctl.alpha; dg[SAC_D_ALPHA_LOSS] = ctl.alpha_loss; }
        __syncthreads();
        if (threadIdx.x < SAC_DIAG_N) {
            const float v = dg[threadIdx.x];
            d.diag_last[threadIdx.x] = v;
            if (ctl.loop_pos == 0) d.diag_first[threadIdx.x] = v;
            if (ctl.loop_pos < DIAG_TRACE_CAP) d.diag_trace[(size_t)ctl.loop_pos * SAC_DIAG_N + threadIdx.x] = v;
        }
    }
    // ---- last arriver advances the counters (every block has read ctl by now) ----
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned tk = atomicAdd(&d.ctl->ticket, 1u);
        s_last = (tk == (unsigned)njobs) ? 1u : 0u;
        if (s_last) {
            d.ctl->ticket = 0u;
            d.ctl->n_train_steps_total = ctl.n_train_steps_total + 1;
            d.ctl->adam_t = ctl.adam_t + 1;
            d.ctl->loop_pos = ctl.loop_pos + 1;
            __threadfence();
        }
    }
}

}  // namespace sac

// ==========================================================================================
// host side
// ==========================================================================================
using namespace sac;

struct sac_trainer {
    sac_config_t cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    int B = 0, O = 0, A = 0, KP = 0, KQ = 0, NH = 0, NB = 0;
    Net net[5];
    Dev dev{};
    DwJob *d_jobs = nullptr; int njobs = 0;
    float *ws = nullptr; int64_t ws_floats = 0;      // all activations / gradients
    float *ext_slot = nullptr; SlotLayout ext_layout{};
    float *d_eps = nullptr;                           // [2][B*A]
    float *d_diag = nullptr;                          // first[32] last[32] trace[CAP][32]
    Ctl *d_ctl = nullptr;
    void *h_stage = nullptr; size_t stage_bytes = 0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    float last_ms[4] = {0, 0, 0, 0};
    std::vector<float> h_policy;                      // host mirror for acting
    bool mirror_valid = false;
    size_t lds_pf = 0, lds_qf = 0, lds_qb = 0, lds_pb = 0;
};

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

int alloc_zero(float **p, long long n, hipStream_t s) {
    SAC_HIP(hipMalloc(p, sizeof(float) * (size_t)n));
    SAC_HIP(hipMemsetAsync(*p, 0, sizeof(float) * (size_t)n, s));
    return 0;
}

// flat nn.Linear layout <-> padded device layout (host vectors)
struct FlatMap { int nl; int N[4], K[4]; };   // logical layers in the flat vector

FlatMap flat_map(const sac_trainer *t, int netid) {
    FlatMap f{};
    if (netid == SAC_NET_POLICY) {
        f.nl = 4;
        f.N[0] = H; f.K[0] = t->O; f.N[1] = H; f.K[1] = H; f.N[2] = t->A; f.K[2] = H; f.N[3] = t->A; f.K[3] = H;
    } else {
        f.nl = 3;
        f.N[0] = H; f.K[0] = t->O + t->A; f.N[1] = H; f.K[1] = H; f.N[2] = 1; f.K[2] = H;
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
        for (int n = 0; n < f.N[l]; ++n)
            for (int k = 0; k < f.K[l]; ++k) fn(fi++, dl, n + nshift, k, false);
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

int launch_step(sac_trainer *t, const float *slots, const SlotLayout &SL, int n_slots) {
    const Dev &d = t->dev;
    hipStream_t s = t->stream;
    const int NB = t->NB;
    if (t->NH == 16) hipLaunchKernelGGL(k_policy_fwd<1>, dim3(2 * NB), dim3(256), t->lds_pf, s, d, slots, SL, n_slots);
    else hipLaunchKernelGGL(k_policy_fwd<2>, dim3(2 * NB), dim3(256), t->lds_pf, s, d, slots, SL, n_slots);
    hipLaunchKernelGGL(k_q_fwd, dim3(6 * NB), dim3(256), t->lds_qf, s, d, slots, SL, n_slots);
    hipLaunchKernelGGL(k_q_bwd, dim3(4 * NB), dim3(256), t->lds_qb, s, d, slots, SL, n_slots);
    if (t->NH == 16) hipLaunchKernelGGL(k_policy_bwd<1>, dim3(NB), dim3(256), t->lds_pb, s, d);
    else hipLaunchKernelGGL(k_policy_bwd<2>, dim3(NB), dim3(256), t->lds_pb, s, d);
    hipLaunchKernelGGL(k_dw_adam, dim3(t->njobs + 1), dim3(256), 0, s, d, t->d_jobs, t->njobs, slots, SL, n_slots);
    SAC_HIP(hipGetLastError());
    return 0;
}

// same five launches with HIP events between them (profiling pass only)
int launch_step_timed(sac_trainer *t, const float *slots, const SlotLayout &SL, int n_slots, hipEvent_t *ev) {
    const Dev &d = t->dev;
    hipStream_t s = t->stream;
    const int NB = t->NB;
    SAC_HIP(hipEventRecord(ev[0], s));
    if (t->NH == 16) hipLaunchKernelGGL(k_policy_fwd<1>, dim3(2 * NB), dim3(256), t->lds_pf, s, d, slots, SL, n_slots);
    else hipLaunchKernelGGL(k_policy_fwd<2>, dim3(2 * NB), dim3(256), t->lds_pf, s, d, slots, SL, n_slots);
    SAC_HIP(hipEventRecord(ev[1], s));
    hipLaunchKernelGGL(k_q_fwd, dim3(6 * NB), dim3(256), t->lds_qf, s, d, slots, SL, n_slots);
    SAC_HIP(hipEventRecord(ev[2], s));
    hipLaunchKernelGGL(k_q_bwd, dim3(4 * NB), dim3(256), t->lds_qb, s, d, slots, SL, n_slots);
    SAC_HIP(hipEventRecord(ev[3], s));
    if (t->NH == 16) hipLaunchKernelGGL(k_policy_bwd<1>, dim3(NB), dim3(256), t->lds_pb, s, d);
    else hipLaunchKernelGGL(k_policy_bwd<2>, dim3(NB), dim3(256), t->lds_pb, s, d);
    SAC_HIP(hipEventRecord(ev[4], s));
    hipLaunchKernelGGL(k_dw_adam, dim3(t->njobs + 1), dim3(256), 0, s, d, t->d_jobs, t->njobs, slots, SL, n_slots);
    SAC_HIP(hipEventRecord(ev[5], s));
    SAC_HIP(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" {

int sac_profile_loop(sac_trainer_t *t, sac_buffer_t *b, int64_t n_steps, float out_ms[8]) {
    SAC_REQUIRE(t && b && n_steps > 0 && n_steps <= 4096 && out_ms, "bad arguments to sac_profile_loop");
    SAC_REQUIRE(b->device == t->device && b->O == t->O && b->A == t->A, "buffer does not match trainer");
    SAC_HIP(hipSetDevice(t->device));
    hipStream_t s = t->stream;
    t->dev.eps1 = t->dev.eps2 = nullptr;
    if (ensure_slots(b, t->B, n_steps)) return -1;
    SAC_HIP(hipEventRecord(b->ev[0], b->stream));
    if (launch_sample(b, t->B, n_steps)) return -1;
    SAC_HIP(hipEventRecord(b->ev[1], b->stream));
    if (launch_gather(b, b->d_idx, t->B, n_steps, b->d_slots, b->slot, 1)) return -1;
    SAC_HIP(hipEventRecord(b->ev[2], b->stream));
    SAC_HIP(hipStreamWaitEvent(s, b->ev[2], 0));
    int zero = 0;
    SAC_HIP(hipMemcpyAsync(&t->d_ctl->loop_pos, &zero, sizeof(int), hipMemcpyHostToDevice, s));
    std::vector<hipEvent_t> ev((size_t)n_steps * 6);
    for (auto &e : ev) SAC_HIP(hipEventCreate(&e));
    for (int64_t i = 0; i < n_steps; ++i)
        if (launch_step_timed(t, b->d_slots, b->slot, (int)n_steps, &ev[(size_t)i * 6])) return -1;
    SAC_HIP(hipStreamSynchronize(s));
    double acc[5] = {0, 0, 0, 0, 0};
    for (int64_t i = 0; i < n_steps; ++i)
        for (int k = 0; k < 5; ++k) {
            float ms = 0.f;
            SAC_HIP(hipEventElapsedTime(&ms, ev[(size_t)i * 6 + k], ev[(size_t)i * 6 + k + 1]));
            acc[k] += ms;
        }
    float tot = 0.f;
    SAC_HIP(hipEventElapsedTime(&tot, ev[0], ev[(size_t)n_steps * 6 - 1]));
    for (auto &e : ev) (void)hipEventDestroy(e);
    SAC_HIP(hipEventElapsedTime(&out_ms[0], b->ev[0], b->ev[1]));
    SAC_HIP(hipEventElapsedTime(&out_ms[1], b->ev[1], b->ev[2]));
    for (int k = 0; k < 5; ++k) out_ms[2 + k] = (float)(acc[k] / (double)n_steps);
    out_ms[7] = tot;
    t->mirror_valid = false;
    return 0;
}

int sac_trainer_create(sac_trainer_t **out, const sac_config_t *cfg) {
    SAC_REQUIRE(out && cfg, "null argument to sac_trainer_create");
    *out = nullptr;
    SAC_REQUIRE(sac_device_count() > 0, "no HIP device visible: libsac_hip has no CPU fallback");
    SAC_REQUIRE(cfg->hidden == H, "hidden size %d unsupported (only 256, as in every shipped variant.json)", cfg->hidden);
    SAC_REQUIRE(cfg->obs_dim > 0 && cfg->act_dim > 0 && cfg->act_dim <= 16,
                "unsupported dims obs=%d act=%d (act_dim must be in 1..16)", cfg->obs_dim, cfg->act_dim);
    SAC_REQUIRE(cfg->batch > 0 && cfg->batch % 16 == 0, "batch size %d must be a positive multiple of 16", cfg->batch);
    SAC_REQUIRE(cfg->target_update_period > 0, "target_update_period must be positive");
    SAC_HIP(hipSetDevice(cfg->device));
    sac_trainer *t = new sac_trainer();
    t->cfg = *cfg; t->device = cfg->device;
    t->B = cfg->batch; t->O = cfg->obs_dim; t->A = cfg->act_dim;
    t->KP = round_up(t->O, 16); t->KQ = round_up(t->O + t->A, 16); t->NH = round_up(2 * t->A, 16);
    t->NB = t->B / 16;
    SAC_HIP(hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking));
    for (auto &e : t->ev) SAC_HIP(hipEventCreate(&e));
    hipStream_t s = t->stream;
    const int B = t->B;

    const int shp[3][2] = {{H, t->O}, {H, H}, {2 * t->A, H}};
    const int shq[3][2] = {{H, t->O + t->A}, {H, H}, {1, H}};
    for (int i = 0; i < 5; ++i) {
        Net &n = t->net[i];
        build_layers(n, i == 0 ? shp : shq, 3);
        if (alloc_zero(&n.P, n.nP, s)) return -1;
        if (i < 3) {
            if (alloc_zero(&n.M, n.nP, s) || alloc_zero(&n.V, n.nP, s) || alloc_zero(&n.PT, n.nPT, s) ||
                alloc_zero(&n.G, n.nP, s)) return -1;
        }
    }
    // workspace carve
    Dev &d = t->dev;
    std::vector<std::pair<float **, long long>> parts = {
        {&d.PH1T, (long long)H * B}, {&d.PH2T, (long long)H * B},
        {&d.mu, 16LL * B}, {&d.ls, 16LL * B}, {&d.lsok, 16LL * B}, {&d.z, 16LL * B}, {&d.anew, 16LL * B},
        {&d.epsv, 16LL * B}, {&d.logpi, B}, {&d.a2, 16LL * B}, {&d.logpi2, B}, {&d.part_logpi, round_up(t->NB, 64)},
        {&d.QH1T, 4LL * H * B}, {&d.QH2T, 4LL * H * B}, {&d.q, 6LL * B},
        {&d.y, B}, {&d.dq16T, 2LL * 16 * B}, {&d.dQH2T, 2LL * H * B}, {&d.dQH1T, 2LL * H * B}, {&d.da, 2LL * 16 * B},
        {&d.dheadT, (long long)t->NH * B}, {&d.dPH2T, (long long)H * B}, {&d.dPH1T, (long long)H * B}};
    long long tot = 0;
    for (auto &p : parts) tot += round_up64(p.second, 64);
    if (alloc_zero(&t->ws, tot, s)) return -1;
    t->ws_floats = tot;
    tot = 0;
    for (auto &p : parts) { *p.first = t->ws + tot; tot += round_up64(p.second, 64); }
    t->ext_layout = make_slot_layout(B, t->O, t->A);
    if (alloc_zero(&t->ext_slot, t->ext_layout.slot_floats, s)) return -1;
    if (alloc_zero(&t->d_eps, 2LL * B * t->A, s)) return -1;
    if (alloc_zero(&t->d_diag, (long long)SAC_DIAG_N * (2 + DIAG_TRACE_CAP), s)) return -1;
    SAC_HIP(hipMalloc(&t->d_ctl, sizeof(Ctl)));
    SAC_HIP(hipMemsetAsync(t->d_ctl, 0, sizeof(Ctl), s));

    d.B = B; d.O = t->O; d.A = t->A; d.KP = t->KP; d.KQ = t->KQ; d.NH = t->NH; d.NB = t->NB;
    d.discount = cfg->discount; d.reward_scale = cfg->reward_scale; d.tau = cfg->soft_target_tau;
    d.target_entropy = std::isnan(cfg->target_entropy) ? -(float)t->A : cfg->target_entropy;
    d.alpha_lr = cfg->policy_lr; d.period = cfg->target_update_period;
    d.auto_alpha = cfg->use_automatic_entropy_tuning; d.noise_seed = cfg->noise_seed;
    d.ctl = t->d_ctl;
    for (int i = 0; i < 5; ++i) d.P[i] = t->net[i].P;
    for (int i = 0; i < 3; ++i) d.PT[i] = t->net[i].PT;
    for (int l = 0; l < 3; ++l) { d.LP[l] = t->net[0].L[l]; d.LQ[l] = t->net[1].L[l]; }
    d.diag_first = t->d_diag; d.diag_last = t->d_diag + SAC_DIAG_N; d.diag_trace = t->d_diag + 2 * SAC_DIAG_N;
    d.eps1 = d.eps2 = nullptr;

    // weight-gradient job table
    std::vector<DwJob> jobs;
    auto add_layer = [&](int netid, int l, const float *dYT, const float *XT, int from_slot, float lr) {
        Net &n = t->net[netid];
        const Layer &L = n.L[l];
        for (int n0 = 0; n0 < L.Np; n0 += 16)
            for (int k0 = 0; k0 < L.Kp; k0 += 64) {
                DwJob j{};
                j.dYT = dYT; j.XT = XT; j.xt_from_slot = from_slot; j.xt_off = t->ext_layout.off_saT;
                j.P = n.P + L.offW; j.M = n.M + L.offW; j.V = n.V + L.offW; j.G = n.G + L.offW;
                j.PT = n.PT + L.offWt;
                j.ldp = L.Kp; j.ldt = L.Np; j.N = L.N; j.K = L.K; j.n0 = n0; j.k0 = k0; j.lr = lr;
                if (k0 == 0) { j.bias = n.P + L.offB; j.mb = n.M + L.offB; j.vb = n.V + L.offB; j.gb = n.G + L.offB; }
                if (netid == 1 || netid == 2) {
                    j.TP = t->net[netid + 2].P + L.offW;
                    j.Tbias = t->net[netid + 2].P + L.offB;
                }
                jobs.push_back(j);
            }
    };
    // big (256x256) layers first: they are the longest jobs
    add_layer(0, 1, d.dPH2T, d.PH1T, 0, cfg->policy_lr);
    add_layer(1, 1, d.dQH2T, d.QH1T, 0, cfg->qf_lr);
    add_layer(2, 1, d.dQH2T + (size_t)H * B, d.QH1T + (size_t)H * B, 0, cfg->qf_lr);
    add_layer(0, 0, d.dPH1T, nullptr, 1, cfg->policy_lr);
    add_layer(1, 0, d.dQH1T, nullptr, 1, cfg->qf_lr);
    add_layer(2, 0, d.dQH1T + (size_t)H * B, nullptr, 1, cfg->qf_lr);
    add_layer(0, 2, d.dheadT, d.PH2T, 0, cfg->policy_lr);
    add_layer(1, 2, d.dq16T, d.QH2T, 0, cfg->qf_lr);
    add_layer(2, 2, d.dq16T + (size_t)16 * B, d.QH2T + (size_t)H * B, 0, cfg->qf_lr);
    t->njobs = (int)jobs.size();
    SAC_HIP(hipMalloc(&t->d_jobs, sizeof(DwJob) * jobs.size()));
    SAC_HIP(hipMemcpyAsync(t->d_jobs, jobs.data(), sizeof(DwJob) * jobs.size(), hipMemcpyHostToDevice, s));

    const int KL0p = round_up(t->KP, 64), KL0q = round_up(t->KQ, 64);
    const int nth = t->NH / 16;
    t->lds_pf = sizeof(float) * (size_t)(RB * KL0p + 2 * RB * H + RB * 32 + 4 * nth * 256);
    t->lds_qf = sizeof(float) * (size_t)(RB * KL0q + 2 * RB * H);
    t->lds_qb = sizeof(float) * (size_t)(2 * RB * H + 1024);
    t->lds_pb = sizeof(float) * (size_t)(RB * 64 + RB * H);
    SAC_REQUIRE(t->lds_pf <= 64 * 1024 && t->lds_qf <= 64 * 1024,
                "observation too wide for the 64 KB LDS row-block budget (obs_dim=%d)", t->O);
    SAC_HIP(hipStreamSynchronize(s));
    *out = t;
    return 0;
}

int sac_trainer_destroy(sac_trainer_t *t) {
    if (!t) return 0;
    (void)hipSetDevice(t->device);
    (void)hipStreamSynchronize(t->stream);
    for (auto &n : t->net)
        for (float *p : {n.P, n.M, n.V, n.PT, n.G}) (void)hipFree(p);
    (void)hipFree(t->ws); (void)hipFree(t->ext_slot); (void)hipFree(t->d_eps); (void)hipFree(t->d_diag);
    (void)hipFree(t->d_ctl); (void)hipFree(t->d_jobs);
    if (t->h_stage) (void)hipHostFree(t->h_stage);
    for (auto &e : t->ev) if (e) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(t->stream);
    delete t;
    return 0;
}

int64_t sac_param_count(const sac_trainer_t *t, int net) {
    if (!t || net < 0 || net > 4) return -1;
    return flat_count(flat_map(t, net));
}

static int upload_padded(sac_trainer *t, int net, const float *flat, float *devbuf, bool also_transposed) {
    Net &n = t->net[net];
    std::vector<float> P((size_t)n.nP, 0.f), PT;
    if (also_transposed) PT.assign((size_t)n.nPT, 0.f);
    for_each_param(t, net, [&](int64_t fi, int l, int nn, int k, bool is_b) {
        const Layer &L = n.L[l];
        if (is_b) P[L.offB + nn] = flat[fi];
        else {
            P[L.offW + (size_t)nn * L.Kp + k] = flat[fi];
            if (also_transposed) PT[L.offWt + (size_t)k * L.Np + nn] = flat[fi];
        }
    });
    SAC_HIP(hipMemcpyAsync(devbuf, P.data(), sizeof(float) * P.size(), hipMemcpyHostToDevice, t->stream));
    if (also_transposed)
        SAC_HIP(hipMemcpyAsync(n.PT, PT.data(), sizeof(float) * PT.size(), hipMemcpyHostToDevice, t->stream));
    SAC_HIP(hipStreamSynchronize(t->stream));
    return 0;
}

static int download_padded(sac_trainer *t, int net, const float *devbuf, float *flat) {
    Net &n = t->net[net];
    std::vector<float> P((size_t)n.nP);
    SAC_HIP(hipMemcpyAsync(P.data(), devbuf, sizeof(float) * P.size(), hipMemcpyDeviceToHost, t->stream));
    SAC_HIP(hipStreamSynchronize(t->stream));
    for_each_param(t, net, [&](int64_t fi, int l, int nn, int k, bool is_b) {
        const Layer &L = n.L[l];
        flat[fi] = is_b ? P[L.offB + nn] : P[L.offW + (size_t)nn * L.Kp + k];
    });
    return 0;
}

int sac_set_params(sac_trainer_t *t, int net, const float *flat, int64_t n) {
    SAC_REQUIRE(t && flat && net >= 0 && net <= 4, "bad arguments to sac_set_params");
    SAC_REQUIRE(n == sac_param_count(t, net), "net %d expects %lld parameters, got %lld", net,
                (long long)sac_param_count(t, net), (long long)n);
    SAC_HIP(hipSetDevice(t->device));
    t->mirror_valid = false;
    return upload_padded(t, net, flat, t->net[net].P, net < 3);
}

int sac_get_params(sac_trainer_t *t, int net, float *flat, int64_t n) {
    SAC_REQUIRE(t && flat && net >= 0 && net <= 4, "bad arguments to sac_get_params");
    SAC_REQUIRE(n == sac_param_count(t, net), "net %d holds %lld parameters, buffer has %lld", net,
                (long long)sac_param_count(t, net), (long long)n);
    SAC_HIP(hipSetDevice(t->device));
    return download_padded(t, net, t->net[net].P, flat);
}

int sac_set_opt_state(sac_trainer_t *t, int net, const float *m, const float *v, int64_t n) {
    SAC_REQUIRE(t && m && v && net >= 0 && net <= 2, "bad arguments to sac_set_opt_state (trained nets are 0..2)");
    SAC_REQUIRE(n == sac_param_count(t, net), "size mismatch in sac_set_opt_state");
    SAC_HIP(hipSetDevice(t->device));
    if (upload_padded(t, net, m, t->net[net].M, false)) return -1;
    return upload_padded(t, net, v, t->net[net].V, false);
}

int sac_get_opt_state(sac_trainer_t *t, int net, float *m, float *v, int64_t n) {
    SAC_REQUIRE(t && m && v && net >= 0 && net <= 2, "bad arguments to sac_get_opt_state (trained nets are 0..2)");
    SAC_REQUIRE(n == sac_param_count(t, net), "size mismatch in sac_get_opt_state");
    SAC_HIP(hipSetDevice(t->device));
    if (download_padded(t, net, t->net[net].M, m)) return -1;
    return download_padded(t, net, t->net[net].V, v);
}

int sac_set_scalars(sac_trainer_t *t, const double sc[6]) {
    SAC_REQUIRE(t && sc, "bad arguments to sac_set_scalars");
    SAC_HIP(hipSetDevice(t->device));
    Ctl c;
    memset(&c, 0, sizeof(c));
    c.log_alpha = (float)sc[0]; c.a_m = (float)sc[1]; c.a_v = (float)sc[2];
    c.adam_t = (long long)sc[3]; c.n_train_steps_total = (long long)sc[4];
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
    sc[0] = c.log_alpha; sc[1] = c.a_m; sc[2] = c.a_v; sc[3] = (double)c.adam_t;
    sc[4] = (double)c.n_train_steps_total; sc[5] = c.alpha;
    return 0;
}

int sac_step(sac_trainer_t *t, const float *obs, const float *act, const float *rew, const float *term,
             const float *next_obs, const float *eps1, const float *eps2, float *diag) {
    SAC_REQUIRE(t && obs && act && rew && term && next_obs, "null batch pointer in sac_step");
    SAC_REQUIRE((eps1 == nullptr) == (eps2 == nullptr), "eps1 and eps2 must both be given or both be NULL");
    SAC_HIP(hipSetDevice(t->device));
    const int B = t->B, O = t->O, A = t->A;
    const SlotLayout &L = t->ext_layout;
    hipStream_t s = t->stream;
    // np_to_pytorch_batch: host fp32 -> pinned -> HBM slot (row-major part), + feature-major saT
    const size_t nfl = (size_t)B * (2 * O + A + 2) + (size_t)L.KQ64 * B + (eps1 ? 2 * (size_t)B * A : 0);
    if (ensure_stage_t(t, sizeof(float) * nfl)) return -1;
    float *st = (float *)t->h_stage;
    float *so = st, *sa = so + (size_t)B * O, *sr = sa + (size_t)B * A, *stt = sr + B, *sn = stt + B,
          *sT = sn + (size_t)B * O, *se = sT + (size_t)L.KQ64 * B;
    memcpy(so, obs, sizeof(float) * B * O); memcpy(sa, act, sizeof(float) * B * A);
    memcpy(sr, rew, sizeof(float) * B); memcpy(stt, term, sizeof(float) * B);
    memcpy(sn, next_obs, sizeof(float) * B * O);
    memset(sT, 0, sizeof(float) * (size_t)L.KQ64 * B);
    for (int b = 0; b < B; ++b) {
        for (int k = 0; k < O; ++k) sT[(size_t)k * B + b] = obs[(size_t)b * O + k];
        for (int k = 0; k < A; ++k) sT[(size_t)(O + k) * B + b] = act[(size_t)b * A + k];
    }
    float *E = t->ext_slot;
    SAC_HIP(hipMemcpyAsync(E + L.off_obs, so, sizeof(float) * B * O, hipMemcpyHostToDevice, s));
    SAC_HIP(hipMemcpyAsync(E + L.off_act, sa, sizeof(float) * B * A, hipMemcpyHostToDevice, s));
    SAC_HIP(hipMemcpyAsync(E + L.off_rew, sr, sizeof(float) * B, hipMemcpyHostToDevice, s));
    SAC_HIP(hipMemcpyAsync(E + L.off_term, stt, sizeof(float) * B, hipMemcpyHostToDevice, s));
    SAC_HIP(hipMemcpyAsync(E + L.off_nobs, sn, sizeof(float) * B * O, hipMemcpyHostToDevice, s));
    SAC_HIP(hipMemcpyAsync(E + L.off_saT, sT, sizeof(float) * (size_t)L.KQ64 * B, hipMemcpyHostToDevice, s));
    if (eps1) {
        memcpy(se, eps1, sizeof(float) * B * A); memcpy(se + (size_t)B * A, eps2, sizeof(float) * B * A);
        SAC_HIP(hipMemcpyAsync(t->d_eps, se, sizeof(float) * 2 * B * A, hipMemcpyHostToDevice, s));
        t->dev.eps1 = t->d_eps; t->dev.eps2 = t->d_eps + (size_t)B * A;
    } else {
        t->dev.eps1 = t->dev.eps2 = nullptr;
    }
    int zero = 0;
    SAC_HIP(hipMemcpyAsync(&t->d_ctl->loop_pos, &zero, sizeof(int), hipMemcpyHostToDevice, s));
    if (launch_step(t, t->ext_slot, L, 1)) return -1;
    if (diag) SAC_HIP(hipMemcpyAsync(diag, t->dev.diag_last, sizeof(float) * SAC_DIAG_N, hipMemcpyDeviceToHost, s));
    SAC_HIP(hipStreamSynchronize(s));
    t->mirror_valid = false;
    return 0;
}

int sac_train_loop(sac_trainer_t *t, sac_buffer_t *b, int64_t n_steps, float *diag_first, float *diag_last) {
    SAC_REQUIRE(t && b && n_steps > 0, "bad arguments to sac_train_loop");
    SAC_REQUIRE(b->device == t->device, "buffer and trainer live on different devices");
    SAC_REQUIRE(b->O == t->O && b->A == t->A, "buffer dims (%d,%d) do not match trainer dims (%d,%d)", b->O, b->A,
                t->O, t->A);
    SAC_HIP(hipSetDevice(t->device));
    hipStream_t s = t->stream;
    t->dev.eps1 = t->dev.eps2 = nullptr;
    // 1) indices for every step, 2) one gather launch -> slots; both on the buffer's stream
    if (ensure_slots(b, t->B, n_steps)) return -1;
    SAC_HIP(hipEventRecord(b->ev[0], b->stream));
    if (launch_sample(b, t->B, n_steps)) return -1;
    SAC_HIP(hipEventRecord(b->ev[1], b->stream));
    if (launch_gather(b, b->d_idx, t->B, n_steps, b->d_slots, b->slot, 1)) return -1;
    SAC_HIP(hipEventRecord(b->ev[2], b->stream));
    SAC_HIP(hipStreamWaitEvent(s, b->ev[2], 0));
    // 3) the steps
    int zero = 0;
    SAC_HIP(hipMemcpyAsync(&t->d_ctl->loop_pos, &zero, sizeof(int), hipMemcpyHostToDevice, s));
    SAC_HIP(hipEventRecord(t->ev[0], s));
    for (int64_t i = 0; i < n_steps; ++i)
        if (launch_step(t, b->d_slots, b->slot, (int)n_steps)) return -1;
    SAC_HIP(hipEventRecord(t->ev[1], s));
    if (diag_first) SAC_HIP(hipMemcpyAsync(diag_first, t->dev.diag_first, sizeof(float) * SAC_DIAG_N, hipMemcpyDeviceToHost, s));
    if (diag_last) SAC_HIP(hipMemcpyAsync(diag_last, t->dev.diag_last, sizeof(float) * SAC_DIAG_N, hipMemcpyDeviceToHost, s));
    SAC_HIP(hipStreamSynchronize(s));
    SAC_HIP(hipEventElapsedTime(&t->last_ms[1], b->ev[0], b->ev[1]));
    SAC_HIP(hipEventElapsedTime(&t->last_ms[2], b->ev[1], b->ev[2]));
    SAC_HIP(hipEventElapsedTime(&t->last_ms[3], t->ev[0], t->ev[1]));
    SAC_HIP(hipEventElapsedTime(&t->last_ms[0], b->ev[0], t->ev[1]));
    t->mirror_valid = false;
    return 0;
}

int sac_sync(sac_trainer_t *t) {
    SAC_REQUIRE(t, "null trainer");
    SAC_HIP(hipSetDevice(t->device));
    SAC_HIP(hipStreamSynchronize(t->stream));
    return 0;
}

int sac_last_loop_ms(sac_trainer_t *t, float *total_ms, float *sample_ms, float *gather_ms, float *steps_ms) {
    SAC_REQUIRE(t, "null trainer");
    if (total_ms) *total_ms = t->last_ms[0];
    if (sample_ms) *sample_ms = t->last_ms[1];
    if (gather_ms) *gather_ms = t->last_ms[2];
    if (steps_ms) *steps_ms = t->last_ms[3];
    return 0;
}

int64_t sac_debug_fetch(sac_trainer_t *t, const char *name, float *out, int64_t cap) {
    if (!t || !name || !out) { sac::set_error("bad arguments to sac_debug_fetch"); return -2; }
    if (hipSetDevice(t->device) != hipSuccess) { sac::set_error("hipSetDevice failed"); return -1; }
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
            if (cap < (int64_t)B * A) { sac::set_error("buffer too small"); return -2; }
            if (fetch(e.p, 16LL * B, h)) return -1;
            for (int b = 0; b < B; ++b) for (int a = 0; a < A; ++a) out[(size_t)b * A + a] = h[(size_t)b * 16 + a];
            return (int64_t)B * A;
        }
    struct Vec { const char *n; const float *p; };
    const Vec vecs[] = {{"log_pi", d.logpi}, {"log_pi_next", d.logpi2}, {"q1", d.q}, {"q2", d.q + B},
                        {"q1_new", d.q + 2 * (size_t)B}, {"q2_new", d.q + 3 * (size_t)B},
                        {"tq1", d.q + 4 * (size_t)B}, {"tq2", d.q + 5 * (size_t)B}, {"q_target", d.y}};
    for (auto &e : vecs)
        if (nm == e.n) {
            if (cap < B) { sac::set_error("buffer too small"); return -2; }
            if (fetch(e.p, B, h)) return -1;
            memcpy(out, h.data(), sizeof(float) * B);
            return B;
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
    SAC_REQUIRE(deterministic || eps, "stochastic acting needs the N(0,1) draw (eps)");
    if (!t->mirror_valid && sac_policy_mirror(t)) return -1;
    const int O = t->O, A = t->A;
    const float *p = t->h_policy.data();
    const float *W0 = p, *b0 = W0 + (size_t)H * O, *W1 = b0 + H, *b1 = W1 + (size_t)H * H;
    const float *Wm = b1 + H, *bm = Wm + (size_t)A * H, *Ws = bm + A, *bs = Ws + (size_t)A * H;
    float h1[H], h2[H];
    for (int n = 0; n < H; ++n) {
        float s = b0[n];
        for (int k = 0; k < O; ++k) s += W0[(size_t)n * O + k] * obs[k];
        h1[n] = s > 0.f ? s : 0.f;
    }
    for (int n = 0; n < H; ++n) {
        float s = b1[n];
        for (int k = 0; k < H; ++k) s += W1[(size_t)n * H + k] * h1[k];
        h2[n] = s > 0.f ? s : 0.f;
    }
    for (int a = 0; a < A; ++a) {
        float m = bm[a], ls = bs[a];
        for (int k = 0; k < H; ++k) { m += Wm[(size_t)a * H + k] * h2[k]; ls += Ws[(size_t)a * H + k] * h2[k]; }
        if (deterministic) act[a] = tanhf(m);
        else {
            ls = fminf(fmaxf(ls, LOG_SIG_MIN), LOG_SIG_MAX);
            act[a] = tanhf(m + expf(ls) * eps[a]);
        }
    }
    return 0;
}

}  // extern "C"
