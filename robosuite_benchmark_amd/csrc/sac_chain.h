// k_chain: launches A and B of the four-launch step as ONE launch, for batches whose column split is 1
// (B >= 1024; included by sac_trainer.hip, namespace sac).  BASELINE config 3: Door, batch 1024.
//
// Why: with one workgroup per (network pass, 16-row block) and the whole 256-wide layers in it (column split 1) no
// partial sums cross workgroups any more, so the seam between launch A (policy forward, Q(s,a) forward) and launch B
// (tanh-Gaussian head, Q(s,a_new), target nets, actor tail) is not a seam at all: a workgroup can run the policy, take
// its own head, and go straight on into the Q nets.  What launch B looked like at batch 1024 (in-kernel stamps,
// scratch/stamps_big.py): the Q(s,a_new) blocks -- forward + actor tail -- end at 15.2 us, the target-net blocks at 9.5:
// half the chip idles for 6 us; each launch spends ~3.3 us in its prologue before its first 256 x 256 GEMM and ~2.5 us at
// the dispatch boundary.  Here a 16-row block is four work items of similar length, one workgroup each (4 * NB = 256
// workgroups at batch 1024: one per CU):
//
//   item 0, 1  (P_i)   pi(s) -> head (a_new, log pi) -> Q_i(s, a_new) -> unit actor gradient dQ_i/da      ~880 MFMAs/wave
//   item 2     (N)     pi(s') -> head (a', log pi') -> T1(s', a') -> T2(s', a')                             ~960
//   item 3     (C)     Q1(s, a) -> Q2(s, a)   (activations kept feature-major for the weight-gradient launch)  ~640
//
// (pi(s) is computed by both P items: 320 MFMAs per wave for not having a cross-workgroup hand-off.)  The only all-to-all
// seam left in front of the backward launch is the entropy coefficient (mean log pi over the batch), so the step is
// three launches: k_chain, k_bwd (column split 1), k_dw_adam.  Everything k_bwd / k_dw_adam / the diagnostics read is
// written in the layouts of the four-launch kernels at column split 1 (qpart[pass][row], dapart[twin][row][16], the
// per-row head values, the feature-major activations), so those launches are used unchanged.
//
// Arithmetic: the policy pass, the head and the target / critic passes run the MFMA sequences of k_fwd_a / k_fwd_b at
// column split 1 (bit-identical).  Q_i(s, a_new)'s first layer is contracted over [obs | a_new] directly -- the
// four-launch step takes launch A's pre-activations of Q_i(s, a) and adds W1[:, action] (a_new - a), which needs the
// other pass's result -- so q_new and what follows agree with the four-launch step to rounding (~1e-7 relative), not
// bit for bit; against the oracle both sit within the same tolerances (tests/test_gpu_large_batch.py).
#pragma once

namespace chain {

// one 256-wide layer for 16 rows: wave w owns columns 64 w .. 64 w + 63 (four tiles).  Two rings, used alternately: the
// first layers' and the 256 x 256 layers' (DB chunks deep; 8 measured the same as 4: the waves are not waiting for weights)
constexpr int DB = 4;
using Ring = WRing<4, 4>;
using RingB = WRing<4, DB>;

// chunks [S_begin, S_end) of a KS-chunk contraction (S_begin a multiple of the ring depth): gemm_ring's loop -- the same
// MFMA order -- cut in two so that the NEXT layer's ring can be requested in straight-line code between the pieces, one
// microsecond before this layer's last MFMA (a request inside the loop would be a conditional load: rule 6 of DESIGN.md)
template <int D>
__device__ __forceinline__ void gemm_span(WRing<4, D> &R, const float *X, int KL, int S_begin, int S_end, int KS, f32x4 (&acc)[4]) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    const float *xrow = X + r * KL;
    f32x4 a_cur = ld4(xrow + 4 * ((4 * S_begin + g) ^ r));
    for (int S0 = S_begin; S0 < S_end; S0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int S = S0 + u;
            if (S < S_end) {
                const int Sn = (S + 1 < KS) ? S + 1 : S;
                const f32x4 a_nxt = ld4(xrow + 4 * ((4 * Sn + g) ^ r));
                SB();
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[i], R.b[u][t][i], acc[t], 0, 0, 0);
                SB();
                if (S + D < KS) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) R.b[u][t] = ld4(R.wp[t] + 256 * (S + D));
                }
                SB();
                a_cur = a_nxt;
            }
        }
    }
}

// the 256 x 256 layers (16 chunks): chunks 0-7, then `between` (the next ring's requests), then chunks 8-15
#define CHAIN_GEMM16(R, X, ACC, BETWEEN)                 \
    do {                                                 \
        chain::gemm_span(R, X, H, 0, 8, 16, ACC);        \
        SB();                                            \
        BETWEEN;                                         \
        SB();                                            \
        chain::gemm_span(R, X, H, 8, 16, 16, ACC);       \
    } while (0)

// A first layer (ring A: its first four chunks were requested long ago) with the request of the 256 x 256 layer behind it
// (ring B).  Narrow (at most four chunks: obs + act <= 64 columns): B's requests go out FIRST and the GEMM is straight-line
// code, so its waits count exactly and leave them in flight; wide: a loop with a run-time trip count, whose back-edge
// would drain whatever is in flight (rule 4 of DESIGN.md) -- B is requested behind it.
template <bool WIDE>
__device__ __forceinline__ void first_layer(Ring &A_, const float *X0, int KL, int KS, f32x4 (&acc)[4], RingB &B_) {
    if constexpr (!WIDE) {
        B_.fill(H >> 4);
        SB();
        gemm_straight(A_, X0, KL, KS, acc);
    } else {
        gemm_ring(A_, X0, KL, KS, acc);
        SB();
        B_.fill(H >> 4);
    }
    SB();
}

__device__ __forceinline__ void bias4(const float *P, const Layer &L, float (&bv)[4]) {
    const int wave = threadIdx.x >> 6, c = threadIdx.x & 15;
#pragma unroll
    for (int t = 0; t < 4; ++t) bv[t] = P[L.offB + 64 * wave + 16 * t + c];
}

// relu(acc + b) of this wave's four tiles -> the [16][256] LDS block Xn; the values stay in `keep`
__device__ __forceinline__ void to_lds(const f32x4 (&acc)[4], const float (&bv)[4], float *Xn, f32x4 (&keep)[4]) {
    hidden_epilogue<4>(acc, 64 * (threadIdx.x >> 6), 16, bv, Xn, H, keep);
}

// q = h2 . w3 (without the bias: the consumers add it, as for the four-launch step's partials) -> out[row]
__device__ __forceinline__ void q_value(const float (&w3)[16], const float *XS, float *out) {
    const int row = threadIdx.x >> 4, a = threadIdx.x & 15;
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) s += XS[lds_off(row, a + 16 * u, H)] * w3[u];
    s = group16_sum(s);
    if (a == 0) out[row] = s;
}
__device__ __forceinline__ void load_w3(const float *P, const Layer &L2, float (&w3)[16]) {
    const int a = threadIdx.x & 15;
#pragma unroll
    for (int u = 0; u < 16; ++u) w3[u] = P[L2.offW + frag_off(0, a + 16 * u, H)];
}

// One Q-net pass over the rows in X0 ([obs | action], KQ columns): ring A holds the first layer (requested by the
// caller), ring B takes the 256 x 256 layer; `next` is run between chunks 11 and 12 of the second layer: the caller's
// requests for whatever follows.  Leaves relu(h2) in XS, relu(h1) in X1 / keep1, relu(h2) values in keep2.
template <int ST, bool WIDE, typename Next>
__device__ __forceinline__ void q_pass(const Dev &d, const float *PQ, Ring &A_, RingB &B_, const float *X0, int KLQ, float *X1,
                                       float *XS, f32x4 (&keep1)[4], f32x4 (&keep2)[4], Next &&next) {
    const int wave = threadIdx.x >> 6;
    float bv[4];
    bias4(PQ, d.LQ[0], bv);
    B_.init(PQ + d.LQ[1].offW, H, 64 * wave, 16);
    {
        f32x4 acc[4] = {};
        first_layer<WIDE>(A_, X0, KLQ, d.KQ >> 4, acc, B_);
        to_lds(acc, bv, X1, keep1);
    }
    bias4(PQ, d.LQ[1], bv);
    lds_barrier();
    STAMP(0, ST);
    {
        f32x4 acc[4] = {};
        CHAIN_GEMM16(B_, X1, acc, next());
        to_lds(acc, bv, XS, keep2);
    }
    lds_barrier();
    STAMP(0, ST + 1);
}

}  // namespace chain

// grid: 4 * NB workgroups (NB even).  b % 8 -> item (two slots each: blocks that share an XCD under round-robin placement
// run the same item, i.e. stream the same networks), row-block 2 (b / 8) + (b & 1).
template <int NTH, bool WIDE>
__global__ __launch_bounds__(256) void k_chain(Dev d, const float *__restrict__ S, SlotLayout SL, StepArg sa) {
    kernarg_prefetch<sizeof(Dev) + 8 + sizeof(SlotLayout) + sizeof(StepArg)>();
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int B = d.B, O = d.O, A = d.A;
    const int KLQ = (d.KQ + 63) & ~63;
    float *X0 = lds;                     // [16][KLQ]  input rows: obs (policy) / cat(obs, action) (Q nets)
    float *X1 = X0 + RB * KLQ;           // [16][256]
    float *XS = X1 + RB * H;             // [16][256]
    float *red = XS + RB * H;            // 4*NTH*256 floats of split-K scratch (>= 1024)
    float *HD = red + 4 * NTH * 256;     // [16][32] head pre-activations of the row-block
    const int xr = blockIdx.x & 7, item = xr >> 1, rb = 2 * (blockIdx.x >> 3) + (xr & 1);
    if (rb >= d.NB) return;
    const int row0 = rb * RB;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int row = threadIdx.x >> 4, a = threadIdx.x & 15, grow = row0 + row;
    const int am = (a < A) ? a : 0;
    STAMP(0, 0);
    chain::Ring RA;
    chain::RingB RBm;                    // two weight rings, used alternately: one feeds the MFMAs, the other is being filled
    f32x4 keep1[4], keep2[4];
    float w3[16];

    if (item == 3) {
        // ---- C: Q1(s, a), Q2(s, a); activations feature-major for the weight-gradient launch ----
        RowRegs<32> rows;
        rows.issue(d.KQ, S + SL.off_obs + (size_t)row0 * O, O, O, S + SL.off_act + (size_t)row0 * A, A, A, d.KP);
        RA.init(d.P[1] + d.LQ[0].offW, d.LQ[0].Kp, 64 * wave, 16);
        RA.fill(d.KQ >> 4);
        SB();
        rows.commit(X0, KLQ, d.KQ, O, d.KP, A);
        lds_barrier();
        STAMP(0, 1);
        // Q1: the second twin's first layer and its q weights are requested inside the first twin's second layer
        chain::q_pass<2, WIDE>(d, d.P[1], RA, RBm, X0, KLQ, X1, XS, keep1, keep2, [&] {
            RA.init(d.P[2] + d.LQ[0].offW, d.LQ[0].Kp, 64 * wave, 16);
            RA.fill(d.KQ >> 4);
            chain::load_w3(d.P[1], d.LQ[2], w3);
        });
        store_features<4>(keep1, 64 * wave, 16, d.QH1T, B, row0);
        store_features<4>(keep2, 64 * wave, 16, d.QH2T, B, row0);
        chain::q_value(w3, XS, d.qpart + row0);
        lds_barrier();                   // (X1 / XS are reused by the second twin)
        chain::q_pass<6, WIDE>(d, d.P[2], RA, RBm, X0, KLQ, X1, XS, keep1, keep2, [&] { chain::load_w3(d.P[2], d.LQ[2], w3); });
        store_features<4>(keep1, 64 * wave, 16, d.QH1T + (size_t)H * B, B, row0);
        store_features<4>(keep2, 64 * wave, 16, d.QH2T + (size_t)H * B, B, row0);
        chain::q_value(w3, XS, d.qpart + (size_t)B + row0);
        STAMP(0, 12);
        return;
    }

    // ---- P0, P1, N: the policy on s (P) or s' (N) ----
    const int side = (item == 2) ? 1 : 0;
    const bool own = (item == 0);        // P0 stores what the policy pass on s leaves for the backward / weight-gradient launches
    const float *PP = d.P[0];
    const float *PQ = d.P[(item == 2) ? 3 : 1 + item];       // the Q net that follows the head: Q_i (P_i) / T1 (N)
    WRing<NTH, 4> rh;
    {
        RowRegs<32> rows;
        rows.issue(d.KP, S + (side ? SL.off_nobs : SL.off_obs) + (size_t)row0 * O, O, O, nullptr, 0, 0, 0);
        RA.init(PP + d.LP[0].offW, d.LP[0].Kp, 64 * wave, 16);
        RA.fill(d.KP >> 4);
        SB();
        rows.commit(X0, KLQ, d.KP, O, 0, 0);
        lds_barrier();
        STAMP(0, 1);
    }
    {
        float bv[4];
        chain::bias4(PP, d.LP[0], bv);
        RBm.init(PP + d.LP[1].offW, H, 64 * wave, 16);
        {
            f32x4 acc[4] = {};
            chain::first_layer<WIDE>(RA, X0, KLQ, d.KP >> 4, acc, RBm);
            chain::to_lds(acc, bv, X1, keep1);
        }
        chain::bias4(PP, d.LP[1], bv);
        lds_barrier();
        STAMP(0, 2);
        {
            f32x4 acc[4] = {};
            // inside the policy's second layer: the head's weights and the first layer of the Q net that follows
            CHAIN_GEMM16(RBm, X1, acc, {
                rh.init(PP + d.LP[2].offW, H, 0, 16, 4 * wave);
                rh.fill(4);
                RA.init(PQ + d.LQ[0].offW, d.LQ[0].Kp, 64 * wave, 16);
                RA.fill(d.KQ >> 4);
            });
            chain::to_lds(acc, bv, XS, keep2);
        }
        if (own) {
            store_features<4>(keep1, 64 * wave, 16, d.PH1T, B, row0);
            store_features<4>(keep2, 64 * wave, 16, d.PH2T, B, row0);
        }
        lds_barrier();
        STAMP(0, 3);
    }
    {   // head pre-activations: each wave contracts its 64 columns of h2, summed through LDS in wave order (the order
        // in which the four-launch step adds them)
        f32x4 acc[NTH] = {};
        gemm_ring(rh, XS, H, 4, acc, 4 * wave);
        splitk_reduce<NTH>(acc, nullptr, red, HD, 32);
    }
    STAMP(0, 4);
    // ---- tanh-Gaussian head (k_fwd_b's arithmetic) ----
    const float *epp = side ? d.eps2 : d.eps1;
    float mean = 0.f, raw = 0.f, lstd = 0.f, stdv = 1.f, eps = 0.f, zz = 0.f, act = 0.f, lp = 0.f;
    if (a < A) {
        mean = HD[row * 32 + a] + PP[d.LP[2].offB + am];
        raw = HD[row * 32 + A + a] + PP[d.LP[2].offB + A + am];
        lstd = fminf(fmaxf(raw, LOG_SIG_MIN), LOG_SIG_MAX);
        stdv = expf(lstd);
        eps = epp ? epp[grow * A + am]
                  : philox_normal(d.noise_seed, (unsigned long long)sa.step_now, (unsigned)(grow * 16 + a), side ? 1u : 0u);
        zz = __fadd_rn(mean, __fmul_rn(stdv, eps));
        act = tanhf(zz);
        const float dd = __fsub_rn(zz, mean);
        const float var = __fmul_rn(stdv, stdv);
        const float nlp = -(dd * dd) / (2.0f * var) - logf(stdv) - 0.91893853320467274178f;
        lp = nlp - logf(1.0f - act * act + TANH_EPS);
    }
    const float lsum = group16_sum(lp);
    if (own) {
        if (a < A) {
            d.mu[grow * 16 + a] = mean;
            d.ls[grow * 16 + a] = lstd;
            d.lsok[grow * 16 + a] = (raw >= LOG_SIG_MIN && raw <= LOG_SIG_MAX) ? 1.f : 0.f;
            d.z[grow * 16 + a] = zz;
            d.epsv[grow * 16 + a] = eps;
        }
        d.anew[grow * 16 + a] = act;                          // (0 beyond A)
        if (a == 0) { d.logpi[grow] = lsum; red[row] = (grow < d.Bt) ? lsum : 0.f; }
    } else if (item == 2) {
        d.a2[grow * 16 + a] = act;
        if (a == 0) d.logpi2[grow] = lsum;
    }
    // the Q nets' input: [obs | 0 | action | 0]: the observation columns are in place, the action chunk is written now
    // (columns O .. KP-1 are zero from the commit; the policy's GEMM never read beyond KP)
    X0[lds_off(row, d.KP + a, KLQ)] = (a < A) ? act : 0.f;
    lds_barrier();
    STAMP(0, 5);
    if (own && threadIdx.x == 0) {       // this row-block's sum(log pi), fixed order
        float s = 0.f;
        for (int i = 0; i < RB; ++i) s += red[i];
        d.part_logpi[rb] = s;
    }

    if (item == 2) {
        // ---- N: T1(s', a'), T2(s', a') ----
        chain::q_pass<6, WIDE>(d, d.P[3], RA, RBm, X0, KLQ, X1, XS, keep1, keep2, [&] {
            RA.init(d.P[4] + d.LQ[0].offW, d.LQ[0].Kp, 64 * wave, 16);
            RA.fill(d.KQ >> 4);
            chain::load_w3(d.P[3], d.LQ[2], w3);
        });
        chain::q_value(w3, XS, d.qpart + (size_t)4 * B + row0);
        lds_barrier();
        chain::q_pass<8, WIDE>(d, d.P[4], RA, RBm, X0, KLQ, X1, XS, keep1, keep2, [&] { chain::load_w3(d.P[4], d.LQ[2], w3); });
        chain::q_value(w3, XS, d.qpart + (size_t)5 * B + row0);
        STAMP(0, 12);
        return;
    }

    // ---- P_i: Q_i(s, a_new) and the UNIT input gradient dQ_i/da (see k_fwd_b) ----
    const int qi = item;
    const float *PT = d.PT[1 + qi];
    WRing<1, 4> ra;
    // inside Q_i's second layer: the transposed second layer of the tail (ring A is free by then) and the q weights
    chain::q_pass<6, WIDE>(d, PQ, RA, RBm, X0, KLQ, X1, XS, keep1, keep2, [&] { chain::load_w3(PQ, d.LQ[2], w3); });
    // the tail's transposed second layer goes through the deep ring too (free again now)
    RBm.init(PT + d.LQ[1].offWt, H, 64 * wave, 16);
    RBm.fill(H >> 4);
    SB();
    chain::q_value(w3, XS, d.qpart + (size_t)(2 + qi) * B + row0);
    // dq/dh2 = w3 * relu'(h2), in place (own elements)
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int off = lds_off(row, a + 16 * u, H);
        XS[off] = (XS[off] > 0.f) ? w3[u] : 0.f;
    }
    lds_barrier();
    {   // dq/dh1 = dq/dh2 . W2, masked in place; the action rows of W1^T are requested inside it
        f32x4 acc[4] = {};
        CHAIN_GEMM16(RBm, XS, acc, {
            ra.init(PT + d.LQ[0].offWt, H, d.KP, 16, 4 * wave);
            ra.fill(4);
        });
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int off = lds_off(4 * g + i, 64 * wave + 16 * t + c, H);
                X1[off] = (X1[off] > 0.f) ? acc[t][i] : 0.f;
            }
    }
    lds_barrier();
    STAMP(0, 10);
    {   // dq/da = dq/dh1 . W1[:, action columns]  (contraction split over the waves)
        f32x4 acc[1] = {};
        gemm_ring(ra, X1, H, 4, acc, 4 * wave);
        splitk_reduce<1>(acc, nullptr, red, d.dapart + ((size_t)qi * B + row0) * 16, 16);
    }
    STAMP(0, 12);
}
#undef CHAIN_GEMM16

// ---- the eight-wave variant's helpers (see k_chain8 below) ----
namespace chain8 {

constexpr int CW = 8, CT = 16 / CW, CF = 16 * CT;     // waves, 16-column tiles per wave (2), features per wave (32)
constexpr int DB = 4;
using Ring = WRing<CT, 4>;
using RingB = WRing<CT, DB>;

template <int D>
__device__ __forceinline__ void gemm_span(WRing<CT, D> &R, const float *X, int KL, int S_begin, int S_end, int KS, f32x4 (&acc)[CT]) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    const float *xrow = X + r * KL;
    f32x4 a_cur = ld4(xrow + 4 * ((4 * S_begin + g) ^ r));
    for (int S0 = S_begin; S0 < S_end; S0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int S = S0 + u;
            if (S < S_end) {
                const int Sn = (S + 1 < KS) ? S + 1 : S;
                const f32x4 a_nxt = ld4(xrow + 4 * ((4 * Sn + g) ^ r));
                SB();
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < CT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[i], R.b[u][t][i], acc[t], 0, 0, 0);
                SB();
                if (S + D < KS) {
#pragma unroll
                    for (int t = 0; t < CT; ++t) R.b[u][t] = ld4(R.wp[t] + 256 * (S + D));
                }
                SB();
                a_cur = a_nxt;
            }
        }
    }
}

#define CHAIN8_GEMM16(R, X, ACC, BETWEEN)                \
    do {                                                 \
        chain8::gemm_span(R, X, H, 0, 8, 16, ACC);       \
        SB();                                            \
        BETWEEN;                                         \
        SB();                                            \
        chain8::gemm_span(R, X, H, 8, 16, 16, ACC);      \
    } while (0)

template <bool WIDE>
__device__ __forceinline__ void first_layer(Ring &A_, const float *X0, int KL, int KS, f32x4 (&acc)[CT], RingB &B_) {
    if constexpr (!WIDE) {
        B_.fill(H >> 4);
        SB();
        gemm_straight(A_, X0, KL, KS, acc);
    } else {
        gemm_ring(A_, X0, KL, KS, acc);
        SB();
        B_.fill(H >> 4);
    }
    SB();
}

__device__ __forceinline__ void bias4(const float *P, const Layer &L, float (&bv)[CT]) {
    const int wave = threadIdx.x >> 6, c = threadIdx.x & 15;
#pragma unroll
    for (int t = 0; t < CT; ++t) bv[t] = P[L.offB + CF * wave + 16 * t + c];
}

__device__ __forceinline__ void to_lds(const f32x4 (&acc)[CT], const float (&bv)[CT], float *Xn, f32x4 (&keep)[CT]) {
    hidden_epilogue<CT>(acc, CF * (threadIdx.x >> 6), 16, bv, Xn, H, keep);
}

// (the first four waves: thread = (row, 16 lanes), as in k_chain)
template <bool SC1 = false>        // SC1: another workgroup of this launch reads the values (k_chain8<.., BWD>)
__device__ __forceinline__ void q_value(const float (&w3)[16], const float *XS, float *out) {
    const int row = (threadIdx.x & 255) >> 4, a = threadIdx.x & 15;
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) s += XS[lds_off(row, a + 16 * u, H)] * w3[u];
    s = group16_sum(s);
    if (a == 0 && threadIdx.x < 256) { if constexpr (SC1) st_sc1(out + row, s); else out[row] = s; }
}
__device__ __forceinline__ void load_w3(const float *P, const Layer &L2, float (&w3)[16]) {
    const int a = threadIdx.x & 15;
#pragma unroll
    for (int u = 0; u < 16; ++u) w3[u] = P[L2.offW + frag_off(0, a + 16 * u, H)];
}

// split-K epilogue of a product that the FIRST FOUR waves computed (splitk_reduce's sum, in its order)
template <int NTT, bool SC1 = false>
__device__ __forceinline__ void reduce4(const f32x4 (&acc)[NTT], float *red, float *out, int ldo) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {
#pragma unroll
        for (int t = 0; t < NTT; ++t) st4(red + ((wave * NTT + t) * 64 + lane) * 4, acc[t]);
    }
    lds_barrier();
    for (int e = threadIdx.x; e < NTT * 256; e += 64 * CW) {
        const int t = e >> 8, l = (e >> 2) & 63, i = e & 3;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[((w * NTT + t) * 64 + l) * 4 + i];
        float *o = out + (4 * (l >> 4) + i) * ldo + 16 * t + (l & 15);
        if constexpr (SC1) st_sc1(o, s); else *o = s;
    }
    lds_barrier();
}

template <int ST, bool WIDE, typename Next>
__device__ __forceinline__ void q_pass(const Dev &d, const float *PQ, Ring &A_, RingB &B_, const float *X0, int KLQ, float *X1,
                                       float *XS, f32x4 (&keep1)[CT], f32x4 (&keep2)[CT], Next &&next) {
    const int wave = threadIdx.x >> 6;
    float bv[CT];
    bias4(PQ, d.LQ[0], bv);
    B_.init(PQ + d.LQ[1].offW, H, CF * wave, 16);
    {
        f32x4 acc[CT] = {};
        first_layer<WIDE>(A_, X0, KLQ, d.KQ >> 4, acc, B_);
        to_lds(acc, bv, X1, keep1);
    }
    bias4(PQ, d.LQ[1], bv);
    lds_barrier();
    STAMP(0, ST);
    {
        f32x4 acc[CT] = {};
        CHAIN8_GEMM16(B_, X1, acc, next());
        to_lds(acc, bv, XS, keep2);
    }
    lds_barrier();
    STAMP(0, ST + 1);
}

}  // namespace chain8

// EIGHT waves per workgroup (round 3, second half): the same items, each wave with TWO 16-column tiles of a 256-wide layer
// instead of four -- two waves per SIMD, one's LDS reads / address arithmetic / epilogues under the other's MFMAs (what
// paid in the general step's matrix-product kernel, sac_general.h).  Every tile's MFMA sequence is k_chain's: the results
// are k_chain's bit for bit; the per-row sections (row staging, head, q values) and the two split-K products (head, action
// gradient: four partials, summed in k_chain's order) run on the first four waves.
// BWD (round 3's last hours): the BACKWARD launch inside this one.  The only all-to-all seam between the forward and the
// backward pass is the entropy coefficient -- NB row-block sums of log pi -- and with 4 * NB <= CUs every workgroup of the launch
// is resident, so the seam can be an in-launch hand-off (sac_fused.h's protocol: write-through stores, one relaxed agent-scope
// counter per producer group, a 50-ms give-up that makes the weight-gradient launch apply nothing and the host fall back to
// k_chain8 + k_bwd8): every backward block needs the target values, i.e. starts behind item N's forward, the longest -- so the
// three blocks of a row-block go to three different items, each behind its own hand-off: item C (the shortest) publishes its
// Q2(s, a) pass and runs Q1's critic block once item N has published T1 / T2 / log pi'; item N runs Q2's critic block on item
// C's activations; item P0 runs the policy block once P1 has published Q2(s, a_new) and dQ2/da; all behind the log-pi sums of
// every P0 item.  (With both critic blocks in item C the launch took 45.7 us: they run one after the other there.)  Same
// blocks, same arithmetic as k_bwd8: bit-identical to k_chain8 + k_bwd8.
// grid: 4 * NB workgroups (NB even).  b % 8 -> item (two slots each: blocks that share an XCD under round-robin placement
// run the same item, i.e. stream the same networks), row-block 2 (b / 8) + (b & 1).
template <int NTH, bool WIDE, bool BWD = false>
__global__ __launch_bounds__(512) void k_chain8(Dev d, const float *__restrict__ S, SlotLayout SL, StepArg sa) {
    kernarg_prefetch<sizeof(Dev) + 8 + sizeof(SlotLayout) + sizeof(StepArg)>();
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int B = d.B, O = d.O, A = d.A;
    const int KLQ = (d.KQ + 63) & ~63;
    float *X0 = lds;                     // [16][KLQ]  input rows: obs (policy) / cat(obs, action) (Q nets)
    float *X1 = X0 + RB * KLQ;           // [16][256]
    float *XS = X1 + RB * H;             // [16][256]
    float *red = XS + RB * H;            // 4*NTH*256 floats of split-K scratch (>= 1024)
    float *HD = red + 4 * NTH * 256;     // [16][32] head pre-activations of the row-block
    const int xr = blockIdx.x & 7, item = xr >> 1, rb = 2 * (blockIdx.x >> 3) + (xr & 1);
    if (rb >= d.NB) return;
    // hand-off counters (BWD): tq[rb] <- item N, ac[rb] <- item P1, lp <- every P0 item (units of the launch number)
    unsigned *cnt_qa = d.cnt + (size_t)2 * d.NB * CNT_STRIDE, *cnt_tq = d.cnt + (size_t)3 * d.NB * CNT_STRIDE,
             *cnt_ac = d.cnt + (size_t)4 * d.NB * CNT_STRIDE, *cnt_lp = d.cnt + (size_t)5 * d.NB * CNT_STRIDE;
    __shared__ int s_ok;
    // test hook (the give-up path must work on hardware): on the launch the host marks, item N of row-block 0 leaves without publishing
    if (BWD && (sa.pad2 & 1u) && blockIdx.x == 4) return;
    const int row0 = rb * RB;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const bool lo = threadIdx.x < 256;                        // the waves that run the per-row sections
    const int row = (threadIdx.x & 255) >> 4, a = threadIdx.x & 15, grow = row0 + row;
    const int am = (a < A) ? a : 0;
    STAMP(0, 0);
    chain8::Ring RA;
    chain8::RingB RBm;                    // two weight rings, used alternately: one feeds the MFMAs, the other is being filled
    f32x4 keep1[chain8::CT], keep2[chain8::CT];
    float w3[16];

    if (item == 3) {
        // ---- C: Q1(s, a), Q2(s, a); activations feature-major for the weight-gradient launch ----
        RowRegs<32> rows;
        if (lo) rows.issue(d.KQ, S + SL.off_obs + (size_t)row0 * O, O, O, S + SL.off_act + (size_t)row0 * A, A, A, d.KP);
        RA.init(d.P[1] + d.LQ[0].offW, d.LQ[0].Kp, chain8::CF * wave, 16);
        RA.fill(d.KQ >> 4);
        SB();
        if (lo) rows.commit(X0, KLQ, d.KQ, O, d.KP, A);
        lds_barrier();
        STAMP(0, 1);
        // Q1: the second twin's first layer and its q weights are requested inside the first twin's second layer
        chain8::q_pass<2, WIDE>(d, d.P[1], RA, RBm, X0, KLQ, X1, XS, keep1, keep2, [&] {
            RA.init(d.P[2] + d.LQ[0].offW, d.LQ[0].Kp, chain8::CF * wave, 16);
            RA.fill(d.KQ >> 4);
            chain8::load_w3(d.P[1], d.LQ[2], w3);
        });
        store_features<chain8::CT>(keep1, chain8::CF * wave, 16, d.QH1T, B, row0);
        store_features<chain8::CT>(keep2, chain8::CF * wave, 16, d.QH2T, B, row0);
        chain8::q_value(w3, XS, d.qpart + row0);
        lds_barrier();                   // (X1 / XS are reused by the second twin)
        chain8::q_pass<6, WIDE>(d, d.P[2], RA, RBm, X0, KLQ, X1, XS, keep1, keep2, [&] { chain8::load_w3(d.P[2], d.LQ[2], w3); });
        // (BWD: the second twin's activations and q values are item N's operands -- it runs that twin's critic backward)
        store_features<chain8::CT, BWD>(keep1, chain8::CF * wave, 16, d.QH1T + (size_t)H * B, B, row0);
        store_features<chain8::CT, BWD>(keep2, chain8::CF * wave, 16, d.QH2T + (size_t)H * B, B, row0);
        chain8::q_value<BWD>(w3, XS, d.qpart + (size_t)B + row0);
        STAMP(0, 12);
        if constexpr (BWD) {
            handoff_publish(cnt_qa + (size_t)rb * CNT_STRIDE);
            // the target values of this row-block and the log-pi sums of all: then Q1's critic backward block
            handoff_wait_multi<16>(2, cnt_tq + (size_t)rb * CNT_STRIDE, sa.seq, cnt_lp, (unsigned)d.NB * sa.seq, cnt_lp, (unsigned)d.NB * sa.seq,
                               d.abort_flag, &s_ok);
            lds_barrier();
            if (!s_ok) return;
            bwd8::critic_block<true>(d, S, SL, sa, 0, rb);
        }
        return;
    }

    // ---- P0, P1, N: the policy on s (P) or s' (N) ----
    const int side = (item == 2) ? 1 : 0;
    const bool own = (item == 0);        // P0 stores what the policy pass on s leaves for the backward / weight-gradient launches
    const float *PP = d.P[0];
    const float *PQ = d.P[(item == 2) ? 3 : 1 + item];       // the Q net that follows the head: Q_i (P_i) / T1 (N)
    WRing<NTH, 4> rh;
    {
        RowRegs<32> rows;
        if (lo) rows.issue(d.KP, S + (side ? SL.off_nobs : SL.off_obs) + (size_t)row0 * O, O, O, nullptr, 0, 0, 0);
        RA.init(PP + d.LP[0].offW, d.LP[0].Kp, chain8::CF * wave, 16);
        RA.fill(d.KP >> 4);
        SB();
        if (lo) rows.commit(X0, KLQ, d.KP, O, 0, 0);
        lds_barrier();
        STAMP(0, 1);
    }
    {
        float bv[chain8::CT];
        chain8::bias4(PP, d.LP[0], bv);
        RBm.init(PP + d.LP[1].offW, H, chain8::CF * wave, 16);
        {
            f32x4 acc[chain8::CT] = {};
            chain8::first_layer<WIDE>(RA, X0, KLQ, d.KP >> 4, acc, RBm);
            chain8::to_lds(acc, bv, X1, keep1);
        }
        chain8::bias4(PP, d.LP[1], bv);
        lds_barrier();
        STAMP(0, 2);
        {
            f32x4 acc[chain8::CT] = {};
            // inside the policy's second layer: the head's weights (first four waves) and the first layer of the Q net that follows
            CHAIN8_GEMM16(RBm, X1, acc, {
                if (wave < 4) { rh.init(PP + d.LP[2].offW, H, 0, 16, 4 * wave); rh.fill(4); }
                RA.init(PQ + d.LQ[0].offW, d.LQ[0].Kp, chain8::CF * wave, 16);
                RA.fill(d.KQ >> 4);
            });
            chain8::to_lds(acc, bv, XS, keep2);
        }
        if (own) {
            store_features<chain8::CT>(keep1, chain8::CF * wave, 16, d.PH1T, B, row0);
            store_features<chain8::CT>(keep2, chain8::CF * wave, 16, d.PH2T, B, row0);
        }
        lds_barrier();
        STAMP(0, 3);
    }
    {   // head pre-activations: each wave contracts its 64 columns of h2, summed through LDS in wave order (the order
        // in which the four-launch step adds them)
        f32x4 acc[NTH] = {};
        if (wave < 4) gemm_ring(rh, XS, H, 4, acc, 4 * wave);
        chain8::reduce4<NTH>(acc, red, HD, 32);
    }
    STAMP(0, 4);
    // ---- tanh-Gaussian head (k_fwd_b's arithmetic) ----
    const float *epp = side ? d.eps2 : d.eps1;
    float mean = 0.f, raw = 0.f, lstd = 0.f, stdv = 1.f, eps = 0.f, zz = 0.f, act = 0.f, lp = 0.f;
    if (a < A) {
        mean = HD[row * 32 + a] + PP[d.LP[2].offB + am];
        raw = HD[row * 32 + A + a] + PP[d.LP[2].offB + A + am];
        lstd = fminf(fmaxf(raw, LOG_SIG_MIN), LOG_SIG_MAX);
        stdv = expf(lstd);
        eps = epp ? epp[grow * A + am]
                  : philox_normal(d.noise_seed, (unsigned long long)sa.step_now, (unsigned)(grow * 16 + a), side ? 1u : 0u);
        zz = __fadd_rn(mean, __fmul_rn(stdv, eps));
        act = tanhf(zz);
        const float dd = __fsub_rn(zz, mean);
        const float var = __fmul_rn(stdv, stdv);
        const float nlp = -(dd * dd) / (2.0f * var) - logf(stdv) - 0.91893853320467274178f;
        lp = nlp - logf(1.0f - act * act + TANH_EPS);
    }
    const float lsum = group16_sum(lp);
    if (own && lo) {
        if (a < A) {
            d.mu[grow * 16 + a] = mean;
            d.ls[grow * 16 + a] = lstd;
            d.lsok[grow * 16 + a] = (raw >= LOG_SIG_MIN && raw <= LOG_SIG_MAX) ? 1.f : 0.f;
            d.z[grow * 16 + a] = zz;
            d.epsv[grow * 16 + a] = eps;
        }
        d.anew[grow * 16 + a] = act;                          // (0 beyond A)
        if (a == 0) { d.logpi[grow] = lsum; red[row] = (grow < d.Bt) ? lsum : 0.f; }
    } else if (item == 2 && lo) {
        d.a2[grow * 16 + a] = act;
        if (a == 0) { if constexpr (BWD) st_sc1(d.logpi2 + grow, lsum); else d.logpi2[grow] = lsum; }
    }
    // the Q nets' input: [obs | 0 | action | 0]: the observation columns are in place, the action chunk is written now
    // (columns O .. KP-1 are zero from the commit; the policy's GEMM never read beyond KP)
    if (lo) X0[lds_off(row, d.KP + a, KLQ)] = (a < A) ? act : 0.f;
    lds_barrier();
    STAMP(0, 5);
    if (own && threadIdx.x == 0) {       // this row-block's sum(log pi), fixed order
        float s = 0.f;
        for (int i = 0; i < RB; ++i) s += red[i];
        if constexpr (BWD) {
            st_sc1(d.part_logpi + rb, s);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(cnt_lp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            d.part_logpi[rb] = s;
        }
    }

    if (item == 2) {
        // ---- N: T1(s', a'), T2(s', a') ----
        chain8::q_pass<6, WIDE>(d, d.P[3], RA, RBm, X0, KLQ, X1, XS, keep1, keep2, [&] {
            RA.init(d.P[4] + d.LQ[0].offW, d.LQ[0].Kp, chain8::CF * wave, 16);
            RA.fill(d.KQ >> 4);
            chain8::load_w3(d.P[3], d.LQ[2], w3);
        });
        chain8::q_value<BWD>(w3, XS, d.qpart + (size_t)4 * B + row0);
        lds_barrier();
        chain8::q_pass<8, WIDE>(d, d.P[4], RA, RBm, X0, KLQ, X1, XS, keep1, keep2, [&] { chain8::load_w3(d.P[4], d.LQ[2], w3); });
        chain8::q_value<BWD>(w3, XS, d.qpart + (size_t)5 * B + row0);
        STAMP(0, 12);
        if constexpr (BWD) {
            handoff_publish(cnt_tq + (size_t)rb * CNT_STRIDE);
            // item C's Q2(s, a) pass of this row-block (long out) and the log-pi sums of all: then Q2's critic backward block
            handoff_wait_multi<16>(2, cnt_qa + (size_t)rb * CNT_STRIDE, sa.seq, cnt_lp, (unsigned)d.NB * sa.seq, cnt_lp, (unsigned)d.NB * sa.seq,
                               d.abort_flag, &s_ok);
            lds_barrier();
            if (!s_ok) return;
            bwd8::critic_block<true>(d, S, SL, sa, 1, rb);
        }
        return;
    }

    // ---- P_i: Q_i(s, a_new) and the UNIT input gradient dQ_i/da (see k_fwd_b) ----
    const int qi = item;
    const float *PT = d.PT[1 + qi];
    WRing<1, 4> ra;
    // inside Q_i's second layer: the transposed second layer of the tail (ring A is free by then) and the q weights
    chain8::q_pass<6, WIDE>(d, PQ, RA, RBm, X0, KLQ, X1, XS, keep1, keep2, [&] { chain8::load_w3(PQ, d.LQ[2], w3); });
    // the tail's transposed second layer goes through the deep ring too (free again now)
    RBm.init(PT + d.LQ[1].offWt, H, chain8::CF * wave, 16);
    RBm.fill(H >> 4);
    SB();
    chain8::q_value<BWD>(w3, XS, d.qpart + (size_t)(2 + qi) * B + row0);
    // dq/dh2 = w3 * relu'(h2), in place (own elements)
    if (lo) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int off = lds_off(row, a + 16 * u, H);
            XS[off] = (XS[off] > 0.f) ? w3[u] : 0.f;
        }
    }
    lds_barrier();
    {   // dq/dh1 = dq/dh2 . W2, masked in place; the action rows of W1^T are requested inside it
        f32x4 acc[chain8::CT] = {};
        CHAIN8_GEMM16(RBm, XS, acc, {
            if (wave < 4) { ra.init(PT + d.LQ[0].offWt, H, d.KP, 16, 4 * wave); ra.fill(4); }
        });
#pragma unroll
        for (int t = 0; t < chain8::CT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int off = lds_off(4 * g + i, chain8::CF * wave + 16 * t + c, H);
                X1[off] = (X1[off] > 0.f) ? acc[t][i] : 0.f;
            }
    }
    lds_barrier();
    STAMP(0, 10);
    {   // dq/da = dq/dh1 . W1[:, action columns]  (contraction split over the waves)
        f32x4 acc[1] = {};
        if (wave < 4) gemm_ring(ra, X1, H, 4, acc, 4 * wave);
        chain8::reduce4<1, BWD>(acc, red, d.dapart + ((size_t)qi * B + row0) * 16, 16);
    }
    STAMP(0, 12);
    if constexpr (BWD) {
        if (qi == 1) { handoff_publish(cnt_ac + (size_t)rb * CNT_STRIDE); return; }
        // P0: the other twin's q_new and action gradient, the log-pi sums of all row-blocks: then the policy backward block
        handoff_wait_multi<16>(2, cnt_ac + (size_t)rb * CNT_STRIDE, sa.seq, cnt_lp, (unsigned)d.NB * sa.seq, cnt_lp, (unsigned)d.NB * sa.seq,
                           d.abort_flag, &s_ok);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        if (!s_ok) return;
        bwd8::policy_block<NTH, true>(d, sa, rb);
    }
}
#undef CHAIN8_GEMM16
