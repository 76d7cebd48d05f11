// k_bwd8: the backward launch at column split 1 (batch >= 1024) on EIGHT waves per workgroup -- two per SIMD, each with two
// 16-column tiles of the 256-wide layers instead of four (see k_chain8, sac_chain.h).  SAC only.  The same blocks as
// k_bwd<NTH, 1> (critic_bwd_block / policy_bwd_block at SP = 1): every tile's MFMA sequence and every scalar expression is
// theirs, so the results are theirs bit for bit; the per-row / per-feature sections run on the first four waves.
// Included by sac_trainer.hip (namespace sac).
#pragma once

namespace bwd8 {

constexpr int CT = 2, CF = 32;      // 16-column tiles per wave, features per wave

// XW (k_chain8<.., BWD = true>, sac_chain.h): the block runs INSIDE the forward launch, behind in-launch hand-offs -- what
// another workgroup of the launch wrote (target q values, log pi', the other twin's q_new and action gradient, the row-block
// sums of log pi) is read through agent-scope loads, and the entropy step sums those sums from vector loads (the scalar
// cache may hold the previous launch's words); same values, same order: same bits.
template <bool XW>
__device__ __forceinline__ float xload(const float *p) { return XW ? ld_sc1(p) : *p; }

// alpha_step with the row-block sums read by vector agent-scope loads, summed in alpha_step's order (NB <= 64 per pass)
__device__ __forceinline__ AlphaStep alpha_step_xw(const Ctl *ctl, const float *part_logpi, int NB, int B, float target_entropy,
                                                   float lr, int auto_alpha, double bc1, double bc2s) {
    AlphaStep r;
    const float la = sload(&ctl->log_alpha), m0 = sload(&ctl->a_m), v0 = sload(&ctl->a_v);
    if (!auto_alpha) { r.alpha = 1.0f; r.alpha_loss = 0.0f; r.log_alpha = la; r.m = m0; r.v = v0; return r; }
    const int lane = threadIdx.x & 63;
    float sum = 0.f;
    for (int i0 = 0; i0 < NB; i0 += 64) {
        const float mine = ld_sc1(part_logpi + (i0 + lane < NB ? i0 + lane : 0));
#pragma unroll
        for (int i = 0; i < 64; ++i)
            if (i0 + i < NB) sum += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), i));
    }
    const float mean_lp = sum / (float)B + target_entropy;
    r.alpha_loss = -((la * mean_lp) + 0.0f);
    const float gr = -mean_lp;
    r.m = m0 + ADAM_1MB1 * (gr - m0);
    r.v = v0 * ADAM_B2 + ADAM_1MB2 * gr * gr;
    const float step_size = (float)((double)lr / bc1);
    const float denom = sqrtf(r.v) / (float)bc2s + 1e-8f;
    r.log_alpha = la + (-step_size * r.m) / denom;
    r.alpha = expf(r.log_alpha);
    return r;
}

template <bool XW = false>
__device__ __forceinline__ void critic_block(const Dev &d, const float *__restrict__ S, const SlotLayout &SL, const StepArg &sa,
                                             int qi, int rb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int B = d.B, NB = d.NB;
    float *X2 = lds;                 // dL/dh2 row-block [16][256]
    __shared__ float s_dq[RB];
    const int row0 = rb * RB;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const bool lo = threadIdx.x < 256;
    const float *P = d.P[1 + qi];
    const float *PT = d.PT[1 + qi];
    const float invB = 1.0f / (float)d.Bt;
    const float *h2T = d.QH2T + (size_t)qi * H * B;
    const float *h1T = d.QH1T + (size_t)qi * H * B;
    const int n0 = CF * wave;
    const long long oB3 = d.LQ[2].offB;

    const float b3a = sload(d.P[3] + oB3), b3b = sload(d.P[4] + oB3), b3q = sload(P + oB3);
    float qa = 0.f, qb = 0.f, qq = 0.f, in_c = 0.f, in_r = 0.f, in_t = 0.f;
    if (threadIdx.x < RB) {
        const int r = row0 + threadIdx.x;
        qa = xload<XW>(d.qpart + (size_t)4 * B + r);
        qb = xload<XW>(d.qpart + (size_t)5 * B + r);
        qq = xload<XW>(d.qpart + (size_t)qi * B + r);
        in_c = xload<XW>(d.logpi2 + r);
        in_r = S[SL.off_rew + r]; in_t = S[SL.off_term + r];
    }
    const int k = threadIdx.x & 255;                         // (first four waves: thread = feature k)
    const float wk = P[d.LQ[2].offW + frag_off(0, k, H)];
    f32x4 h2v[4];
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) h2v[qd] = ld4(h2T + frag_off(k, row0 + 4 * qd, B));
    SB();
    WRing<CT, 4> r1;
    r1.init(PT + d.LQ[1].offWt, H, n0, 16);
    r1.fill(H >> 4);
    SB();
    f32x4 h1v[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) h1v[t] = ld4(h1T + frag_off(n0 + 16 * t + c, row0 + 4 * g, B));
    SB();
    const float alpha = XW ? alpha_step_xw(d.ctl, d.part_logpi, NB, d.Bt, d.target_entropy, d.alpha_lr, d.auto_alpha, sa.bc1, sa.bc2s).alpha
                           : alpha_step(d.ctl, d.part_logpi, NB, d.Bt, d.target_entropy, d.alpha_lr, d.auto_alpha, sa.bc1, sa.bc2s).alpha;
    USE_FROM_HERE(qa); USE_FROM_HERE(qb); USE_FROM_HERE(qq);
    USE_FROM_HERE(in_c); USE_FROM_HERE(in_r); USE_FROM_HERE(in_t);
    float va = 0.f, vb = 0.f, vq = 0.f, yv = 0.f, dq = 0.f;
    if (threadIdx.x < RB) {
        va = qa + b3a;                                                   // T1(s',a')
        vb = qb + b3b;                                                   // T2(s',a')
        vq = qq + b3q;                                                   // Q_i(s,a)
        const float tq = fminf(va, vb) - alpha * in_c;
        yv = bellman_target(d.reward_scale, in_r, in_t, d.discount, tq);
        dq = (row0 + (int)threadIdx.x < d.Bt) ? 2.0f * (vq - yv) * invB : 0.f;
        s_dq[threadIdx.x] = dq;
    }
    lds_barrier();
    f32x4 gv2[4];
    if (lo) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                gv2[qd][i] = (h2v[qd][i] > 0.f) ? s_dq[4 * qd + i] * wk : 0.f;
                X2[lds_off(4 * qd + i, k, H)] = gv2[qd][i];
            }
        }
    }
    lds_barrier();
    {
        f32x4 acc[CT] = {};
        gemm_ring(r1, X2, H, H >> 4, acc);
        if (threadIdx.x < RB) {
            const int r = row0 + threadIdx.x;
            d.q[(size_t)qi * B + r] = vq;
            d.dq16T[(size_t)qi * 16 * B + frag_off(0, r, B)] = dq;
            if (qi == 0) { d.y[r] = yv; d.q[4 * (size_t)B + r] = va; d.q[5 * (size_t)B + r] = vb; }
        }
        if (lo) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) st4(d.dQH2T + (size_t)qi * H * B + frag_off(k, row0 + 4 * qd, B), gv2[qd]);
        }
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            f32x4 gv;
#pragma unroll
            for (int i = 0; i < 4; ++i) gv[i] = (h1v[t][i] > 0.f) ? acc[t][i] : 0.f;
            st4(d.dQH1T + (size_t)qi * H * B + frag_off(n0 + 16 * t + c, row0 + 4 * g, B), gv);
        }
    }
}

template <int NTH, bool XW = false>
__device__ __forceinline__ void policy_block(const Dev &d, const StepArg &sa, int rb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int B = d.B, A = d.A;
    float *XH = lds;                 // [16][64] head gradient row-block
    float *X2 = XH + RB * 64;        // [16][256] dL/dh2
    const int row0 = rb * RB;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;
    const bool lo = threadIdx.x < 256;
    const float *PT = d.PT[0];
    const float invB = 1.0f / (float)d.Bt;
    const int n0 = CF * wave;
    const long long oB3 = d.LQ[2].offB;

    const int row = (threadIdx.x & 255) >> 4, a = threadIdx.x & 15;
    const int gi = (row0 + row) * 16 + a;
    const float b3a = sload(d.P[1] + oB3), b3b = sload(d.P[2] + oB3);
    float act = 0.f, dap0 = 0.f, dap1 = 0.f, lsv = 0.f, epv = 0.f, okv = 0.f;
    const float qa = xload<XW>(d.qpart + (size_t)2 * B + row0 + row), qb = xload<XW>(d.qpart + (size_t)3 * B + row0 + row);
    if (a < A) {
        act = d.anew[gi];
        dap0 = xload<XW>(d.dapart + gi); dap1 = xload<XW>(d.dapart + (size_t)B * 16 + gi);
        lsv = d.ls[gi]; epv = d.epsv[gi]; okv = d.lsok[gi];
    }
    SB();
    WRing<CT> rh;
    rh.init(PT + d.LP[2].offWt, d.LP[2].Np, n0, 16);
    rh.fill(NTH);
    f32x4 h2v[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) h2v[t] = ld4(d.PH2T + frag_off(n0 + 16 * t + c, row0 + 4 * g, B));
    SB();
    WRing<CT, 4> r1;
    r1.init(PT + d.LP[1].offWt, H, n0, 16);
    r1.fill(H >> 4);
    SB();
    f32x4 h1v[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) h1v[t] = ld4(d.PH1T + frag_off(n0 + 16 * t + c, row0 + 4 * g, B));
    SB();
    for (int e = threadIdx.x; e < RB * 64; e += 512) XH[e] = 0.f;
    const float alpha = XW ? alpha_step_xw(d.ctl, d.part_logpi, d.NB, d.Bt, d.target_entropy, d.alpha_lr, d.auto_alpha, sa.bc1, sa.bc2s).alpha
                           : alpha_step(d.ctl, d.part_logpi, d.NB, d.Bt, d.target_entropy, d.alpha_lr, d.auto_alpha, sa.bc1, sa.bc2s).alpha;
    lds_barrier();
    USE_FROM_HERE(act); USE_FROM_HERE(lsv); USE_FROM_HERE(epv); USE_FROM_HERE(okv);
    USE_FROM_HERE(dap0); USE_FROM_HERE(dap1);
    float qnew1 = 0.f, qnew2 = 0.f, dz = 0.f, dls = 0.f;
    {
        const float va = qa + b3a, vb = qb + b3b;                                    // Q1, Q2(s, a_new)
        const float sel1 = (va < vb) ? 1.0f : ((va == vb) ? 0.5f : 0.0f);
        const float dq1 = -invB * sel1, dq2 = -invB * (1.0f - sel1);
        qnew1 = va; qnew2 = vb;
        if (lo && a < A && row0 + row < d.Bt) {
            const float da = actor_da(dap0, dq1, dap1, dq2);
            const float om = 1.0f - act * act;
            const float alpha_invB = __fmul_rn(alpha, invB);
            dz = actor_dz(da, om, alpha_invB, act);
            const float stdv = expf(lsv);
            dls = actor_dls(dz, stdv, epv, alpha_invB, okv);
            XH[lds_off(row, A + a, 64)] = dls;
            XH[lds_off(row, a, 64)] = dz;
        }
    }
    lds_barrier();
    f32x4 gk2[CT];
    {
        f32x4 acc[CT] = {};
        gemm_ring(rh, XH, 64, NTH, acc);
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int n = n0 + 16 * t + c;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                gk2[t][i] = (h2v[t][i] > 0.f) ? acc[t][i] : 0.f;
                X2[lds_off(4 * g + i, n, H)] = gk2[t][i];
            }
        }
    }
    lds_barrier();
    {
        f32x4 acc[CT] = {};
        gemm_ring(r1, X2, H, H >> 4, acc);
        if (lo) {
            if (a == 0) { d.q[2 * (size_t)B + row0 + row] = qnew1; d.q[3 * (size_t)B + row0 + row] = qnew2; }
            if (a < A) {
                d.dheadT[frag_off(a, row0 + row, B)] = dz;
                d.dheadT[frag_off(A + a, row0 + row, B)] = dls;
            }
        }
        store_features<CT>(gk2, n0, 16, d.dPH2T, B, row0);
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            f32x4 gv;
#pragma unroll
            for (int i = 0; i < 4; ++i) gv[i] = (h1v[t][i] > 0.f) ? acc[t][i] : 0.f;
            st4(d.dPH1T + frag_off(n0 + 16 * t + c, row0 + 4 * g, B), gv);
        }
    }
}

}  // namespace bwd8

// the same block -> work map as k_bwd<NTH, 1> (three classes: critic Q1, critic Q2, policy)
template <int NTH>
__global__ __launch_bounds__(512) void k_bwd8(Dev d, const float *__restrict__ S, SlotLayout SL, StepArg sa, int compact) {
    kernarg_prefetch<sizeof(Dev) + 8 + sizeof(SlotLayout) + sizeof(StepArg)>();
    int cls, b;
    if (compact) { cls = (blockIdx.x & 7) >> 1; b = 2 * (blockIdx.x >> 3) + (blockIdx.x & 1); }
    else { cls = blockIdx.x % 3; b = blockIdx.x / 3; }
    if (cls > 2) return;
    if (cls < 2) bwd8::critic_block<false>(d, S, SL, sa, cls, b);
    else bwd8::policy_block<NTH>(d, sa, b);
}
