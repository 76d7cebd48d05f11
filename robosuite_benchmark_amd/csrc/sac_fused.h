// k_abc: launches A, B and C of the SAC step as ONE launch (included by sac_trainer.hip, namespace sac).
//
// Why: a step is four dependent launches of ~6 us each, and every launch pays (i) ~2.2 us of dispatch boundary,
// (ii) ~1.3 us until its first data returns and (iii) a cold pass over its weights (the L2s are invalidated at every
// dispatch; a CU takes in ~50-70 GB/s from the Infinity Cache) before any MFMA runs.  The seams A -> B and B -> C are
// not all-to-all: launch B needs the four head partials of its own 16-row block, launch C the q / d/da partials of its
// own row-block plus 16 scalars (the row-block sums of log pi, for the entropy coefficient).  Here the same 16*NB
// workgroups run all three phases and hand those few KB to each other through HBM inside the launch:
//
//   critic chain (twin i, row-block rb, column part p)      policy chain (side s, rb, p)
//   A  Q_i(s, a): first layer (z kept in registers),         A  pi(side): first layer, 64-column slice, head partial
//      64-column slice (W2 slice streamed once, staged           --> head[side][rb] published
//      transposed in LDS), q partial
//   -- wait head[s][rb] ---------------------------------    -- wait head[s'][rb] ---------------------------------
//   B  tanh-Gaussian head on s; Q_i(s, a_new): first layer    B  head on s'; T_{side+1}(s', a'): first layer, slice,
//      = z + W1[:, act] (a_new - a), slice (second, L2-warm      q partial
//      pass over the same W2 slice), q partial; unit
//      gradient dQ_i/da from the staged slice (partials)
//   -- wait phase B of all 16 blocks of rb, and the 16 row-block sums of log pi -------------------------------------
//   C  critic backward of Q_i (dL/dh kept for dW)             C  side 0: policy backward; side 1: done
//
// While a block waits, the weights of its next phase are already in flight (they do not depend on the hand-off), so
// the cold pass of phases B and C hides behind the wait instead of following a dispatch boundary.
//
// Hand-off protocol (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"):
// producer: payload with agent-scope write-through stores (sc1) -> every storing wave s_waitcnt vmcnt(0) -> workgroup
// barrier -> ONE lane: agent-scope atomic add on a counter; consumer: one lane polls the counter with sc1 loads (plus
// s_sleep), workgroup barrier, then loads.  Counters are monotonic (targets are multiples of the launch number:
// no reset, no ABA).  Every handed-off datum lives in 128-B lines that ONE producer block writes and nobody reads
// before the counter says so, so neither a CU's L1 nor an XCD's L2 can hold a stale copy of them (both are
// invalidated at dispatch); 4-byte consumer loads are sc1 (L1-bypassing) on top of that.  The one single-word hand-off -- a
// row-block's sum of log pi, 16 words for the entropy coefficient of every phase C -- travels as {launch number, value} in ONE
// 8-byte store instead (tagged_publish / tagged_wait: no drain, no counter; the poll returns the value).
//
// Residency: the 16*NB blocks (<= 256, ~106 KB of LDS each: one per CU) must all be resident -- true when the launch
// has the chip to itself.  Nothing spins forever: a wait gives up after 50 ms (s_memrealtime), raises the sticky abort
// word, every other wait sees it and leaves, launch D then applies NOTHING (the step's only writer of weights, Adam
// state, targets and the entropy coefficient), and the host reports the error and falls back to the four-launch step.
// Requirements checked by the host: SAC, column split 4 (batch <= 256), narrow first layers (obs_dim <= 112),
// at least 16*NB CUs.  Fused launches of different trainers in one process are serialised by the host.
#pragma once

constexpr int CNT_STRIDE = 32;                                  // unsigned per counter: one 128-B line each
constexpr unsigned long long HANDOFF_TIMEOUT_TICKS = 5000000ull;   // s_memrealtime ticks (100 MHz): 50 ms
constexpr int FUSED_RED = 2048;                                 // floats of split-K scratch

// one lane: wait until *cnt has reached target (wrap-safe); false = abort (ours or somebody else's)
// (SLEEP: s_sleep units of 64 cycles between polls -- 1 for the short waits of k_abc; the long waits of k_chain8<.., BWD>, where
//  half the chip polls for microseconds, back off further)
template <int SLEEP = 1>
__device__ __forceinline__ int handoff_wait(const unsigned *cnt, unsigned target, unsigned *abort_flag) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const unsigned v = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)(v - target) >= 0) return 1;
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return 0;
        if (__builtin_amdgcn_s_memrealtime() - t0 > HANDOFF_TIMEOUT_TICKS) {
            __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return 0;
        }
        __builtin_amdgcn_s_sleep(SLEEP);
    }
}

// Wave 0: lanes 0 .. n-1 wait for one counter each -- ONE memory round trip per poll iteration for all of them
// (three waits one after the other cost three round trips: +0.8 us measured).  *s_ok = 1 iff every counter got there.
template <int SLEEP = 1>
__device__ __forceinline__ void handoff_wait_multi(int n, const unsigned *c0, unsigned t0, const unsigned *c1, unsigned t1,
                                                   const unsigned *c2, unsigned t2, unsigned *abort_flag, int *s_ok) {
    if (threadIdx.x < 64) {
        const int l = threadIdx.x;
        const unsigned *c = (l == 0) ? c0 : ((l == 1) ? c1 : c2);
        const unsigned t = (l == 0) ? t0 : ((l == 1) ? t1 : t2);
        int ok = 1;
        if (l < n) ok = handoff_wait<SLEEP>(c, t, abort_flag);
        const unsigned long long all = __ballot(ok != 0);
        if (l == 0) *s_ok = (all == ~0ull) ? 1 : 0;
    }
}

// A single value handed over as ONE 8-byte word {launch number, value}: the datum is its own signal.  The producer neither drains
// its stores nor touches a counter (a wave that waits for its write-through stores to be acknowledged is ~0.4 us late at the
// workgroup's next barrier -- measured on the block that publishes a row-block's sum of log pi, which sits on the step's
// critical path); the consumer's poll returns the value with the tag.
__device__ __forceinline__ void tagged_publish(unsigned long long *p, float v, unsigned seq) {
    const unsigned long long w = ((unsigned long long)seq << 32) | (unsigned long long)__builtin_bit_cast(unsigned, v);
    __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int SLEEP = 1>
__device__ __forceinline__ int tagged_wait(const unsigned long long *p, unsigned seq, unsigned *abort_flag, float *v) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const unsigned long long w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(w >> 32) == seq) { *v = __builtin_bit_cast(float, (unsigned)w); return 1; }
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return 0;
        if (__builtin_amdgcn_s_memrealtime() - t0 > HANDOFF_TIMEOUT_TICKS) {
            __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return 0;
        }
        __builtin_amdgcn_s_sleep(SLEEP);
    }
}
// handoff_wait_multi for up to two counters (lanes 0, 1) PLUS the NB tagged row-block sums of log pi (lanes 16 .. 16 + NB - 1; values to
// s_lp[0 .. NB-1] in LDS): still one memory round trip per poll iteration for all of them
// (peek: what multi_lp_peek loaded for this lane some time ago -- an early look that usually finds everything there, so that the
//  wait costs no memory round trip: 0.6-0.8 us at the start of the critic chain's phase C, whose conditions have been true for a
//  microsecond by then; 0 = no early look)
__device__ __forceinline__ void handoff_wait_multi_lp(int n, const unsigned *c0, unsigned t0, const unsigned *c1, unsigned t1,
                                                      const unsigned long long *lp_tag, int NB, unsigned seq, unsigned *abort_flag,
                                                      int *s_ok, float *s_lp, unsigned long long peek = 0ull) {
    if (threadIdx.x < 64) {
        const int l = threadIdx.x;
        int ok = 1;
        if (l < n) {
            const unsigned t = l == 0 ? t0 : t1;
            if ((int)((unsigned)peek - t) < 0) ok = handoff_wait(l == 0 ? c0 : c1, t, abort_flag);
        } else if (l >= 16 && l < 16 + NB) {
            float v = __builtin_bit_cast(float, (unsigned)peek);
            if ((unsigned)(peek >> 32) != seq) ok = tagged_wait(lp_tag + (l - 16), seq, abort_flag, &v);
            s_lp[l - 16] = v;
        }
        const unsigned long long all = __ballot(ok != 0);
        if (l == 0) *s_ok = (all == ~0ull) ? 1 : 0;
    }
}
// the early look for handoff_wait_multi_lp: wave 0's lanes load their counter / tagged word (issued in front of a publish of the
// caller's own, whose drain covers the round trip)
__device__ __forceinline__ unsigned long long multi_lp_peek(int n, const unsigned *c0, const unsigned *c1, const unsigned long long *lp_tag, int NB) {
    unsigned long long w = 0ull;
    if (threadIdx.x < 64) {
        const int l = threadIdx.x;
        if (l < n) w = __hip_atomic_load(l == 0 ? c0 : c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (l >= 16 && l < 16 + NB) w = __hip_atomic_load(lp_tag + (l - 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return w;
}

// every wave: its stores have left; then ONE lane signals for the whole workgroup
__device__ __forceinline__ void handoff_publish(unsigned *cnt) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// alpha_step with the row-block sums taken from LDS (handoff_wait_multi_lp left them there: they were published earlier in THIS
// launch); the sum runs in the same order as alpha_step's, so launch D's diagnostics block computes bit-identical state from
// the same words (the float copies in d.part_logpi)
__device__ __forceinline__ AlphaStep alpha_step_v(const Ctl *ctl, const float *part_logpi, int NB, int B, float target_entropy,
                                                  float lr, int auto_alpha, double bc1, double bc2s) {
    AlphaStep r;
    const float la = sload(&ctl->log_alpha), m0 = sload(&ctl->a_m), v0 = sload(&ctl->a_v);
    if (!auto_alpha) { r.alpha = 1.0f; r.alpha_loss = 0.0f; r.log_alpha = la; r.m = m0; r.v = v0; return r; }
    const int lane = threadIdx.x & 63;
    const float mine = part_logpi[lane < NB ? lane : 0];      // (LDS)
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i < NB) sum += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), i));
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

// Wide first layers (obs_dim > 112; Wipe: 379): recomputing the whole layer in each of the four column parts of a
// row-block is 4 tiles x K/16 chunks x 4 MFMAs per wave and pass (Wipe: 384 MFMAs, ~6 us, twice per chain).  Here a part
// computes ITS quarter of the layer -- this wave: tile `wave` of quarter `part`, one accumulator, the same MFMA sequence
// per tile as the four-tile GEMM: bit-identical -- publishes the pre-activations z = acc + b as one 1-KB tile (MFMA C
// layout, write-through) and, once the four parts of (chain, rb) have signalled, loads this wave's four tiles of the
// whole layer (columns 64 wave + 16 t: quarter `wave`, tile t).  One in-launch hand-off (~1.3 us) for 3/4 of the MFMAs.
// Returns false when the wait gave up.
template <int D>
__device__ __forceinline__ bool split_first_layer(WRing<1, D> &rq, const float *X0, int KLQ, int KS, float bq, float *zx_blk,
                                                  unsigned *cnt, unsigned target, unsigned *abort_flag, int *s_ok, int part,
                                                  f32x4 (&z)[4]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x4 acc[1] = {};
    gemm_ring<false, false>(rq, X0, KLQ, KS, acc);
    f32x4 zq;
#pragma unroll
    for (int i = 0; i < 4; ++i) zq[i] = acc[0][i] + bq;
    st4_sc1(zx_blk + ((part * 4 + wave) * 64 + lane) * 4, zq);
    handoff_publish(cnt);
    if (threadIdx.x == 0) *s_ok = handoff_wait(cnt, target, abort_flag);
    lds_barrier();
    if (!*s_ok) return false;
#pragma unroll
    for (int t = 0; t < 4; ++t) z[t] = ld4(zx_blk + ((wave * 4 + t) * 64 + lane) * 4);
    return true;
}

// MODE M_TD3_CRITIC (round 3): the TD3 critic pass as the same launch.  Critic chain: phase A (Q_i(s, a)), no phase B, phase C
// (critic backward, target without entropy term) behind the target partials.  Policy chain on s' (net 1): the TARGET
// policy, head = tanh(mean) + clipped smoothing noise, T2; policy chain on s (net 0): the online policy's phase A only when
// an actor pass follows this launch (sa.pad2 bit 2: its head partials and activations are what that pass reads), then T1.
// No log pi anywhere, no policy backward (the actor pass keeps its own launches).
template <int NTH, bool WIDE, int MODE = M_SAC>
__global__ __launch_bounds__(256) void k_abc(Dev d, const float *__restrict__ S, SlotLayout SL, StepArg sa) {
    kernarg_prefetch<sizeof(Dev) + 8 + sizeof(SlotLayout) + sizeof(StepArg)>();
    constexpr int SP = 4, SW = 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int s_ok;
    __shared__ float s_dq[RB];
    __shared__ float s_lp[16];          // the row-block sums of log pi (phase C's entropy coefficient)
    const int B = d.B, O = d.O, A = d.A, NB = d.NB;
    const int KLQ = (d.KQ + 63) & ~63;
    float *X0 = lds;                     // [16][KLQ]  input rows (phase A), cat(obs, action) (phase B)
    float *X1 = X0 + RB * KLQ;           // [16][256]
    float *XS = X1 + RB * H;             // [16][64]
    float *red = XS + RB * SW;           // FUSED_RED floats of split-K scratch
    float *WL = red + FUSED_RED;         // [256][WLD]  critic chain: W2 slice, transposed (phase A -> phase B's tail)
    // XCD-aware map (speed only): b % 8 in {0,1} -> critic chain of Q1, {2,3} -> Q2, {4,5} -> policy chain on s (then T1,
    // then the policy backward), {6,7} -> policy chain on s' (then T2); the 16 blocks of row-block rb are 16 rb .. 16 rb + 15
#ifndef SAC_XR_XOR
#define SAC_XR_XOR 4                 // blocks b % 8 in {0..3}: the policy chains (first link of the critical path: dispatched first)
#endif
    const int xq = blockIdx.x >> 3, xr = (blockIdx.x & 7) ^ SAC_XR_XOR;      // (SAC_XR_XOR: placement experiments only)
    const bool isq = xr < 4;
    const int net = (xr >> 1) & 1;                            // critic chain: twin; policy chain: side
    const int bi = 2 * xq + (xr & 1);
    const int part = bi & 3, rb = bi >> 2;
    const int row0 = rb * RB;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int row = threadIdx.x >> 4, a = threadIdx.x & 15, grow = row0 + row;
    const int n0 = SW * part + 16 * wave;                     // this wave's tile of the split layers
    // counters (one per 128-B line): head[side][rb] (policy chain, phase A), qa[rb] (critic chain, phase A), tq[rb] (policy
    // chains, phase B), ac[rb] (critic chain, phase B incl. the actor tail); the NB row-block sums of log pi need none: tagged words
    unsigned *cnt_head = d.cnt, *cnt_qa = d.cnt + (size_t)2 * NB * CNT_STRIDE, *cnt_tq = d.cnt + (size_t)3 * NB * CNT_STRIDE,
             *cnt_ac = d.cnt + (size_t)4 * NB * CNT_STRIDE;
    unsigned *cnt_zx = d.cnt + (size_t)(5 * NB + 2) * CNT_STRIDE;   // [6 chains][NB]: the exchange of the split first layers (WIDE)
    unsigned long long *lp_tag = reinterpret_cast<unsigned long long *>(d.cnt + (size_t)(11 * NB + 2) * CNT_STRIDE);   // [NB] {launch, sum of log pi}
    const unsigned seq = sa.seq;
    const bool own_s = isq && net == 0 && part == 0, own_n = !isq && net == 0 && part == 0;
    // test hook (tests/test_gpu_fused_step.py: the give-up path must work on hardware): on the launch the host marks,
    // one producer block leaves without publishing, so its consumers run into the hand-off timeout
    if ((sa.pad2 & 1u) && blockIdx.x == (4 ^ SAC_XR_XOR)) return;
    STAMP(0, 0);
    // =========================================================================================================
    // phase A
    // =========================================================================================================
    float eps = 0.f;                     // the N(0,1) draw of phase B's rsample (made in phase A, below)
    f32x4 keep1[4], zkeep[4];
    float bv1[1], w3[4];
    unsigned peek = 0u;                  // (thread 0 of a critic-chain block: early look at head[s][rb])
    // TD3 critic pass: the online policy on s runs only when an actor pass follows (it reads this chain's head partials)
    const bool skip_a = (MODE == M_TD3_CRITIC) && !isq && net == 0 && !(sa.pad2 & 4u);
    if (!skip_a) {
        const float *P = isq ? d.P[1 + net] : ((MODE == M_TD3_CRITIC && net == 1) ? d.P[5] : d.P[0]);
        const Layer L0 = isq ? d.LQ[0] : d.LP[0], L1 = isq ? d.LQ[1] : d.LP[1], L2 = isq ? d.LQ[2] : d.LP[2];
        const int K0 = isq ? d.KQ : d.KP;
        const float *obs = S + ((!isq && net) ? SL.off_nobs : SL.off_obs) + (size_t)row0 * O;
        RowRegs<WIDE ? 32 : 8> rows;
        rows.issue(K0, obs, O, O, S + SL.off_act + (size_t)row0 * A, isq ? A : 0, A, d.KP);
        WRing<4, WIDE ? 1 : RD0> r0;
        WRing<1, 8> rq;                                       // WIDE: this wave's ONE tile of this part's quarter (split_first_layer)
        constexpr int PRE0 = 2;
        float bv0[4], bq = 0.f;
        if constexpr (WIDE) {
            rq.init(P + L0.offW, L0.Kp, 64 * part + 16 * wave, 16);
            rq.fill(K0 >> 4);
            bq = P[L0.offB + 64 * part + 16 * wave + c];
        } else {
            r0.init(P + L0.offW, L0.Kp, 64 * wave, 16);
            r0.fill_part(K0 >> 4, 0, PRE0);
#pragma unroll
            for (int t = 0; t < 4; ++t) bv0[t] = P[L0.offB + 64 * wave + 16 * t + c];
        }
        SB();
        WRing<1, 8> r1;
        r1.init(P + L1.offW, H, n0, 16);
        if constexpr (WIDE) r1.fill(H >> 4);
        WRing<NTH, 1> rh;                                     // policy chain: head rows x this wave's 16 columns
#define ABC_LATE_REQUESTS()                                                                             \
        do {                                                                                            \
            bv1[0] = P[L1.offB + n0 + c];                                                               \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) w3[u] = 0.f;                                  \
            if (!isq) {                                                                                 \
                rh.init(P + L2.offW, H, 0, 16, 4 * part + wave);                                        \
                rh.fill(1);                                                                             \
            } else {                                                                                    \
                _Pragma("unroll") for (int u = 0; u < 4; ++u) w3[u] = P[L2.offW + frag_off(0, SW * part + a + 16 * u, H)]; \
            }                                                                                           \
            SB();                                                                                       \
        } while (0)
        if constexpr (WIDE) ABC_LATE_REQUESTS();
        // The N(0,1) draw of phase B's rsample depends on nothing but (seed, step, row, action): it is computed HERE, behind
        // the block's first requests, in the ~1.3 us it waits for their data anyway (in phase B it would sit on the
        // critical path: the head partials are usually there before this chain is).
        {
            const int side0 = isq ? 0 : 1;
            const float *epp0 = side0 ? d.eps2 : d.eps1;
            if (a < A && !epp0) eps = philox_normal(d.noise_seed, (unsigned long long)sa.step_now, (unsigned)(grow * 16 + a), side0 ? 1u : 0u);
        }
        rows.commit(X0, KLQ, K0, O, d.KP, isq ? A : 0);
        lds_barrier();
        if constexpr (WIDE) {   // first layer: this part's quarter, then the exchange with the other three parts
            const int chain = isq ? net : 2 + net;
            if (!split_first_layer(rq, X0, KLQ, K0 >> 4, bq, d.zx + (size_t)(chain * NB + rb) * (RB * H),
                                   cnt_zx + (size_t)(chain * NB + rb) * CNT_STRIDE, 4u * seq, d.abort_flag, &s_ok, part, zkeep)) return;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    keep1[t][i] = fmaxf(zkeep[t][i], 0.f);
                    X1[lds_off(4 * g + i, 64 * wave + 16 * t + c, H)] = keep1[t][i];
                }
        } else {   // first layer, all 256 features (recomputed by the 4 blocks of this row-block)
            f32x4 acc[4] = {};
            gemm_straight_pf(r0, X0, KLQ, K0 >> 4, acc, r1, H >> 4, PRE0);
            // the critic chain keeps the PRE-activation z = W1 [s, a] + b in registers: the layer is linear in the action,
            // so phase B gets Q_i(s, a_new)'s first layer as z + W1[:, action chunk] (a_new - a)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) zkeep[t][i] = acc[t][i] + bv0[t];
            hidden_epilogue<4>(acc, 64 * wave, 16, bv0, X1, H, keep1);
        }
        lds_barrier();
        if constexpr (!WIDE) ABC_LATE_REQUESTS();
#undef ABC_LATE_REQUESTS
        {   // this block's 64 columns of the 256x256 layer; the critic chain also leaves the slice in LDS, transposed
            f32x4 acc[1] = {};
            if (isq) gemm_ring<true>(r1, X1, H, H >> 4, acc, 0, WL + 16 * wave);
            else gemm_ring(r1, X1, H, H >> 4, acc);
            // feature-major copies: operands of the weight-gradient launch, and (sc1) of the backward phase of the
            // sibling blocks of this row-block
            // (the policy chain's copies are stored behind its head publish below: the head is the first link of the
            //  step's critical path, and these 20 KB would sit in the drain in front of the signal)
            if (isq && wave == part) store_features<4, true>(keep1, 64 * wave, 16, d.QH1T + (size_t)net * H * B, B, row0);
            slice_epilogue<1, true>(acc, bv1, wave, XS, isq ? d.QH2T + (size_t)net * H * B : nullptr, n0, B, row0);
        }
        lds_barrier();
        if (!isq) {     // partial head pre-activations over these 64 columns
            f32x4 acc[NTH] = {};
            gemm_ring(rh, XS, SW, 1, acc, wave);
            splitk_reduce<NTH, true>(acc, nullptr, red, d.headpart + ((size_t)(net * NB + rb) * SP + part) * (RB * 32), 32);
        } else {        // partial Q_i(s, a) over these 64 columns
            // (early look at the head counter of phase B: the policy chains publish ~1 us before this chain gets here, so the
            //  wait below usually finds this answer waiting and costs no round trip)
            if (MODE == M_SAC && threadIdx.x == 0) peek = __hip_atomic_load(cnt_head + (size_t)rb * CNT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float s = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) s += XS[lds_off(row, a + 16 * u, SW)] * w3[u];
            s = group16_sum(s);
            if (a == 0) st_sc1(d.qpart2 + ((size_t)(net * NB + rb) * SP + part) * 32 + row, s);
        }
    }
    if (skip_a) {
        // (what phase A would have left for phase B: the noise draw; and the exchange counter of this chain's split first
        //  layer keeps in step with the launch number -- counters count in units of it)
        if (a < A && !d.eps2) eps = philox_normal(d.noise_seed, (unsigned long long)sa.step_now, (unsigned)(grow * 16 + a), 1u);
        if (WIDE && threadIdx.x == 0) __hip_atomic_fetch_add(cnt_zx + (size_t)(2 * NB + rb) * CNT_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    STAMP(0, 1);
    // (the critic chain's phase-A results -- q partial, h2 slice -- are read by its siblings' backward phase: it signals
    //  them behind the head wait below, when their stores have long left, instead of draining them here)
    if (!isq && !skip_a) handoff_publish(cnt_head + (size_t)(net * NB + rb) * CNT_STRIDE);
    // (TD3 critic pass: the critic chain has no phase B; its phase-A results are signalled to its siblings' backward here)
    if (MODE == M_TD3_CRITIC && isq) handoff_publish(cnt_qa + (size_t)rb * CNT_STRIDE);
    STAMP(0, 2);
    if (!isq && net == 0 && !skip_a) {  // pi(s)'s activations for the weight-gradient launch and the policy backward (ordered by tq)
        if (wave == part) store_features<4, true>(keep1, 64 * wave, 16, d.PH1T, B, row0);
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = XS[lds_off(4 * g + i, 16 * wave + c, SW)];
        st4_sc1(d.PH2T + frag_off(n0 + c, row0 + 4 * g, B), v);
    }

    // =========================================================================================================
    // phase B: requests that do not depend on the hand-off first, then the wait
    // =========================================================================================================
    WRing<1, 8> rc;                      // critic chain: the transposed W2 slice of the critic backward (phase C)
    unsigned long long peek_c = 0ull;    // critic chain, wave 0: early look at phase C's counters / tagged words (multi_lp_peek)
    if (MODE == M_SAC || !isq) {         // (TD3 critic pass: the critic chain goes straight on to phase C)
    const int p4 = isq ? net : 2 + net;                       // Q1, Q2 on (s, a_new) | T1, T2 on (s', a')
    const int side = isq ? 0 : 1, pass = 2 + p4;
    const float *PQ = d.P[1 + p4];
    const int KS0 = d.KQ >> 4, lo0 = (isq && !WIDE) ? KS0 - 1 : 0;   // critic chain: only the action chunk of the first layer
    // The N(0,1) draw of rsample does not depend on the hand-off: it is computed in front of the wait.  Then the wait
    // itself, with NOTHING of this block in flight (a poll queued behind a CU's own weight requests returns only when
    // they have: +1-2 us), and only then the weight requests -- they land under the head math.
    const float *epp = side ? d.eps2 : d.eps1;
    const int am = (a < A) ? a : 0;
    const float *PH = (MODE == M_TD3_CRITIC) ? d.P[5] : d.P[0];       // head bias: target policy / online policy
    const float hbm = PH[d.LP[2].offB + am], hbr = (MODE == M_SAC) ? PH[d.LP[2].offB + A + am] : 0.f;
    if (a < A && epp) eps = epp[grow * A + am];            // (caller-supplied noise; the device stream's draw was made at entry)
    // (a handful of small loads that do not depend on the hand-off either: the s' rows of the target net / the batch action)
    RowRegs<WIDE ? 32 : 8> rows2;
    float abat = 0.f;
    if (isq) abat = S[SL.off_act + (size_t)grow * A + ((a < A) ? a : 0)];
    else rows2.issue(d.KQ, S + SL.off_nobs + (size_t)row0 * O, O, O, nullptr, 0, 0, 0);
    if (threadIdx.x == 0)
        s_ok = (isq && (int)(peek - 4u * seq) >= 0) ? 1 : handoff_wait(cnt_head + (size_t)(side * NB + rb) * CNT_STRIDE, 4u * seq, d.abort_flag);
    if (isq) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // phase A's stores (and two small loads): ~1 us old
    lds_barrier();
    if (!s_ok) return;
    if (isq && threadIdx.x == 0) __hip_atomic_fetch_add(cnt_qa + (size_t)rb * CNT_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    STAMP(0, 3);
    float hm[SP], hr[SP];
    {
        const float *hp = d.headpart + (size_t)(side * NB + rb) * SP * (RB * 32) + row * 32;
#pragma unroll
        for (int p = 0; p < SP; ++p) {
            hm[p] = ld_sc1(hp + p * (RB * 32) + am);
            hr[p] = (MODE == M_SAC) ? ld_sc1(hp + p * (RB * 32) + A + am) : 0.f;      // (TD3 heads have no log-std rows)
        }
    }
    f32x4 acc0[4];
    float bv0b[4];
    // (wide first layers: a refilling ring -- for the critic chain it starts AT the action chunk and walks one chunk; the
    //  policy chains, whose target nets run the whole layer, split it over the four parts: split_first_layer)
    const int so0 = (WIDE && isq) ? KS0 - 1 : 0, ks0 = (WIDE && isq) ? 1 : KS0;
    WRing<4, WIDE ? RD : RD0> q0;
    WRing<1, 8> rq2;
    float bq2 = 0.f;
    if (isq) {
#pragma unroll
        for (int t = 0; t < 4; ++t) { acc0[t] = zkeep[t]; bv0b[t] = 0.f; }
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc0[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (!WIDE) {
#pragma unroll
            for (int t = 0; t < 4; ++t) bv0b[t] = PQ[d.LQ[0].offB + 64 * wave + 16 * t + c];
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) bv0b[t] = 0.f;
        }
    }
    if constexpr (WIDE) {
        if (isq) {
            q0.init(PQ + d.LQ[0].offW, d.LQ[0].Kp, 64 * wave, 16, so0);
            q0.fill(ks0);
        } else {
            rq2.init(PQ + d.LQ[0].offW, d.LQ[0].Kp, 64 * part + 16 * wave, 16);
            rq2.fill(KS0);
            bq2 = PQ[d.LQ[0].offB + 64 * part + 16 * wave + c];
        }
    } else {
        q0.init(PQ + d.LQ[0].offW, d.LQ[0].Kp, 64 * wave, 16, so0);
        q0.fill_part(KS0, 0, RD0, lo0);
    }
    WRing<1, 8> q1;
    q1.init(PQ + d.LQ[1].offW, H, n0, 16);
    q1.fill(H >> 4);
    bv1[0] = PQ[d.LQ[1].offB + n0 + c];
#pragma unroll
    for (int u = 0; u < 4; ++u) w3[u] = PQ[d.LQ[2].offW + frag_off(0, SW * part + a + 16 * u, H)];
    SB();
    if (!isq) rows2.commit(X0, KLQ, d.KQ, O, 0, 0, d.KP, d.KP + 16);     // (the head writes the action chunk)
    // ---- tanh-Gaussian head on this block's rows (every block of the row-block computes the same) ----
    float lp = 0.f, mean = 0.f, raw = 0.f, lstd = 0.f, stdv = 1.f, zz = 0.f, act = 0.f;
    USE_FROM_HERE(hm[0]);
    if (a < A) {
        mean = hm[0]; raw = hr[0];
#pragma unroll
        for (int p = 1; p < SP; ++p) { mean += hm[p]; raw += hr[p]; }     // fixed order
        mean += hbm; raw += hbr;
        if constexpr (MODE == M_SAC) {
            lstd = fminf(fmaxf(raw, LOG_SIG_MIN), LOG_SIG_MAX);
            stdv = expf(lstd);
            zz = __fadd_rn(mean, __fmul_rn(stdv, eps));                  // TanhNormal.rsample
            act = tanhf(zz);
            const float dd = __fsub_rn(zz, mean);                            // Normal.log_prob(z)
            const float var = __fmul_rn(stdv, stdv);
            const float nlp = -(dd * dd) / (2.0f * var) - logf(stdv) - 0.91893853320467274178f;
            lp = nlp - logf(1.0f - act * act + TANH_EPS);
        } else {
            // TD3 target smoothing (k_fwd_b<M_TD3_CRITIC>): a' + clamp(N(0,1) * sigma, +-clip); the sum is NOT re-clipped
            zz = tanhf(mean);
            act = zz + fminf(fmaxf(eps * d.td3_sigma, -d.td3_clip), d.td3_clip);
        }
    }
    // the whole action chunk (0 beyond A); the critic chain contracts the difference to the batch action
    X0[lds_off(row, d.KP + a, KLQ)] = (a < A) ? (isq ? act - abat : act) : 0.f;
    const float lsum = group16_sum(lp);
    if (own_s && a == 0) red[row] = (grow < d.Bt) ? lsum : 0.f;          // (pad rows carry no weight)
    lds_barrier();
    float lsum_blk = 0.f;
    if (own_s && threadIdx.x == 0) {          // this row-block's sum(log_pi), fixed order
        for (int i = 0; i < RB; ++i) lsum_blk += red[i];
    }
    STAMP(0, 4);
    // ---- Q / target-Q net on cat(obs, action) ----
    if constexpr (WIDE) {
        if (isq) {
            gemm_ring(q0, X0, KLQ, ks0, acc0, so0);
            hidden_epilogue<4>(acc0, 64 * wave, 16, bv0b, X1, H, keep1);
        } else {
            if (!split_first_layer(rq2, X0, KLQ, KS0, bq2, d.zx + (size_t)((4 + net) * NB + rb) * (RB * H),
                                   cnt_zx + (size_t)((4 + net) * NB + rb) * CNT_STRIDE, 4u * seq, d.abort_flag, &s_ok, part, acc0)) return;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) X1[lds_off(4 * g + i, 64 * wave + 16 * t + c, H)] = fmaxf(acc0[t][i], 0.f);
        }
    } else {
        gemm_straight(q0, X0, KLQ, KS0, acc0, lo0);
        hidden_epilogue<4>(acc0, 64 * wave, 16, bv0b, X1, H, keep1);
    }
    WRing<1, 4> ra;                                          // W1^T action rows: this wave's 64 first-hidden features
    if (isq) {
        SB();
        ra.init(d.PT[1 + net] + d.LQ[0].offWt, H, d.KP, 16, 4 * wave);      // rows KP.. of W1^T = the action columns
        ra.fill(4);
        SB();
    }
    lds_barrier();
    {
        f32x4 acc[1] = {};
        gemm_ring(q1, X1, H, H >> 4, acc);
        // (every load of the forward part has been requested: now the head's global results)
        if (own_s) {
            if (a < A) {
                d.mu[grow * 16 + a] = mean;
                d.ls[grow * 16 + a] = lstd;
                d.z[grow * 16 + a] = zz;
                // what the policy backward of this row-block needs from the head, as ONE 16-byte write-through store per
                // (row, action) -- four dword sc1 stores cost ~6x as much per byte, and this block is on the critical path
                st4_sc1(d.hv4 + (size_t)(grow * 16 + a) * 4,
                        f32x4{act, lstd, eps, (raw >= LOG_SIG_MIN && raw <= LOG_SIG_MAX) ? 1.f : 0.f});
            }
            d.anew[grow * 16 + a] = act;                     // (0 beyond A)
            if (a == 0) d.logpi[grow] = lsum;
            if (threadIdx.x == 0) { d.part_logpi[rb] = lsum_blk; tagged_publish(lp_tag + rb, lsum_blk, seq); }   // (float copy: launch D)
        } else if (own_n) {
            d.a2[grow * 16 + a] = act;
            if (MODE == M_SAC && a == 0) { d.logpi2[grow] = lsum; st_sc1(d.logpi2p + rb * 32 + row, lsum); }
        }
        slice_epilogue<1>(acc, bv1, wave, XS, nullptr, n0, B, row0);
    }
    lds_barrier();
    {
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) s += XS[lds_off(row, a + 16 * u, SW)] * w3[u];
        s = group16_sum(s);
        if (a == 0) st_sc1(d.qpart2 + ((size_t)(pass * NB + rb) * SP + part) * 32 + row, s);
    }
    STAMP(0, 5);
    // critic chain: the transposed W2 slice of the critic backward (phase C) is requested HERE, in front of the actor
    // tail -- it depends on no hand-off, and its first eight k-chunks land while the tail runs
    if (isq) {
        rc.init(d.PT[1 + net] + d.LQ[1].offWt, H, n0, 16);
        rc.fill(H >> 4);
    }
    if (isq) {
        // ---- actor path: UNIT input gradient of Q_i(s, a_new) (dq = 1), partial over this block's columns ----
#pragma unroll
        for (int u = 0; u < 4; ++u) {                        // dq/dh2 = w3 * relu'(h2), in place (own elements)
            const int off = lds_off(row, a + 16 * u, SW);
            XS[off] = (XS[off] > 0.f) ? w3[u] : 0.f;
        }
        lds_barrier();
        {   // partial dq/dh1 over ALL first-hidden features = dq/dh2[:, slice] . W2[slice, :] (staged in phase A), masked
            f32x4 acc[4] = {};
            gemm_lds_rows<4, 4>(WL, 64 * wave, XS, SW, acc);
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
        {   // partial dq/da = dq/dh1 . W1[:, O:O+A]  (contraction split over the waves)
            f32x4 acc[1] = {};
            gemm_ring(ra, X1, H, 4, acc, 4 * wave);
            splitk_reduce<1, true>(acc, nullptr, red, d.dapart + (((size_t)net * SP + part) * B + row0) * 16, 16);
        }
    }
    STAMP(0, 6);
    // (the critic chain's early look at what its phase C waits for -- the target partials, its siblings' phase A, the sums of
    //  log pi: all out for a microsecond by now -- rides in the drain of this publish)
    if (MODE == M_SAC && isq) peek_c = multi_lp_peek(2, cnt_tq + (size_t)rb * CNT_STRIDE, cnt_qa + (size_t)rb * CNT_STRIDE, lp_tag, NB);
    handoff_publish((isq ? cnt_ac : cnt_tq) + (size_t)rb * CNT_STRIDE);
    STAMP(0, 7);
    }   // phase B
    if (MODE == M_TD3_CRITIC) {
        if (!isq) return;                // (the policy chains of a TD3 critic pass have no backward)
        rc.init(d.PT[1 + net] + d.LQ[1].offWt, H, n0, 16);
        rc.fill(H >> 4);
    }

    // =========================================================================================================
    // phase C.  Critic chain: critic backward of Q_i as soon as the TARGET partials of the row-block are out (it does
    // not need the actor tails).  Policy chains (both: s and s' share it, 32 of the 64 dL/dh1 features each): policy
    // backward once the eight actor tails of the row-block are out.  Everything that does not depend on those hand-offs
    // -- transposed weights, activation masks, the entropy coefficient -- is requested / computed in front of the wait.
    // =========================================================================================================
    const float invB = 1.0f / (float)d.Bt;                   // means run over the true batch; pad rows get no gradient
    const long long oB3 = d.LQ[2].offB;
    if (isq) {
        float *X2 = lds;                                     // dL/dh2 row-block [16][256]
        const float *P = d.P[1 + net], *PT = d.PT[1 + net];
        const float *h2T = d.QH2T + (size_t)net * H * B, *h1T = d.QH1T + (size_t)net * H * B;
        const int k = threadIdx.x;
        // the three counters in ONE round trip, with nothing of this block in flight; the target partials have usually been
        // out for a microsecond by now (the policy chains have no actor tail)
        handoff_wait_multi_lp(2, cnt_tq + (size_t)rb * CNT_STRIDE, 8u * seq, cnt_qa + (size_t)rb * CNT_STRIDE, 8u * seq, lp_tag,
                              (MODE == M_SAC) ? NB : 0, seq, d.abort_flag, &s_ok, s_lp, peek_c);
        lds_barrier();
        if (!s_ok) return;
        STAMP(0, 8);
        const float wk = P[d.LQ[2].offW + frag_off(0, k, H)];
        const float b3a = sload(d.P[3] + oB3), b3b = sload(d.P[4] + oB3), b3q = sload(P + oB3);
        float qa[SP], qb[SP], qq[SP], in_c = 0.f, in_r = 0.f, in_t = 0.f;
#pragma unroll
        for (int p = 0; p < SP; ++p) { qa[p] = 0.f; qb[p] = 0.f; qq[p] = 0.f; }
        if (threadIdx.x < RB) {
            const int r = row0 + threadIdx.x;
#pragma unroll
            for (int p = 0; p < SP; ++p) {
                qa[p] = ld_sc1(d.qpart2 + ((size_t)(4 * NB + rb) * SP + p) * 32 + threadIdx.x);
                qb[p] = ld_sc1(d.qpart2 + ((size_t)(5 * NB + rb) * SP + p) * 32 + threadIdx.x);
                qq[p] = ld_sc1(d.qpart2 + ((size_t)(net * NB + rb) * SP + p) * 32 + threadIdx.x);
            }
            if constexpr (MODE == M_SAC) in_c = ld_sc1(d.logpi2p + rb * 32 + threadIdx.x);
            in_r = S[SL.off_rew + r]; in_t = S[SL.off_term + r];
        }
        f32x4 h2v[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) h2v[qd] = ld4(h2T + frag_off(k, row0 + 4 * qd, B));
        f32x4 h1v[1];
        h1v[0] = ld4(h1T + frag_off(n0 + c, row0 + 4 * g, B));
        float alpha = 0.f;                                                   // (TD3: no entropy term in the target)
        if constexpr (MODE == M_SAC)
            alpha = alpha_step_v(d.ctl, s_lp, NB, d.Bt, d.target_entropy, d.alpha_lr, d.auto_alpha, sa.bc1, sa.bc2s).alpha;
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
        STAMP(1, 0);
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
        STAMP(1, 1);
        {   // dL/dh1[:, this block's features] = (dL/dh2 . W2)[:, slice] * relu'(h1)
            f32x4 acc[1] = {};
            gemm_ring(rc, X2, H, H >> 4, acc);
#ifdef SAC_STAMPS
            { float probe = acc[0][0]; asm volatile("" ::"v"(probe)); STAMP(1, 2); }
#endif
            if (threadIdx.x < RB && part == 0) {
                const int r = row0 + threadIdx.x;
                d.q[(size_t)net * B + r] = vq;
                d.dq16T[(size_t)net * 16 * B + frag_off(0, r, B)] = dq;      // row 0 of the padded [16][B]
                if (net == 0) { d.y[r] = yv; d.q[4 * (size_t)B + r] = va; d.q[5 * (size_t)B + r] = vb; }
            }
            if ((k / SW) == part) {
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) st4(d.dQH2T + (size_t)net * H * B + frag_off(k, row0 + 4 * qd, B), gv2[qd]);
            }
            f32x4 gv;
#pragma unroll
            for (int i = 0; i < 4; ++i) gv[i] = (h1v[0][i] > 0.f) ? acc[0][i] : 0.f;
            st4(d.dQH1T + (size_t)net * H * B + frag_off(n0 + c, row0 + 4 * g, B), gv);
        }
        STAMP(0, 9);
    } else {
        // policy backward (actor loss = mean(alpha*log_pi - min Q)), analytic head gradient: see policy_bwd_block.
        // Block (side, rb, part) owns dL/dh1 features 64 part + 32 side .. + 32: waves {0,1} / {2,3} contract the first /
        // second half of the 256 dL/dh2 features for the two 16-feature tiles, the halves meet in LDS.
        float *XH = lds;                 // [16][64] head gradient row-block
        float *X2 = XH + RB * 64;        // [16][256] dL/dh2 (recomputed by the 8 blocks of the row-block)
        const float *PT = d.PT[0];
        const int gi = grow * 16 + a;
        const int nt = SW * part + 32 * net + 16 * (wave & 1), kh = wave >> 1;   // this wave's tile / half of the contraction
        WRing<4> rh;
        rh.init(PT + d.LP[2].offWt, d.LP[2].Np, 64 * wave, 16);
        rh.fill(NTH);
        WRing<1, 8> r1;
        r1.init(PT + d.LP[1].offWt, H, nt, 16, 8 * kh);
        r1.fill(8);
        const float b3a = sload(d.P[1] + oB3), b3b = sload(d.P[2] + oB3);
        SB();
        for (int e = threadIdx.x; e < RB * 64; e += 256) XH[e] = 0.f;
        handoff_wait_multi_lp(1, cnt_tq + (size_t)rb * CNT_STRIDE, 8u * seq, nullptr, 0u, lp_tag, NB, seq,
                              d.abort_flag, &s_ok, s_lp);      // the policy chains' phases A + B (pi(s)'s activations), the log-pi sums
        lds_barrier();
        if (!s_ok) return;
        f32x4 h2v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) h2v[t] = ld4(d.PH2T + frag_off(64 * wave + 16 * t + c, row0 + 4 * g, B));
        f32x4 h1v[1];
        h1v[0] = ld4(d.PH1T + frag_off(nt + c, row0 + 4 * g, B));
        const float alpha = alpha_step_v(d.ctl, s_lp, NB, d.Bt, d.target_entropy, d.alpha_lr, d.auto_alpha, sa.bc1, sa.bc2s).alpha;
        if (threadIdx.x == 0) s_ok = handoff_wait(cnt_ac + (size_t)rb * CNT_STRIDE, 8u * seq, d.abort_flag);   // the eight actor tails
        lds_barrier();
        if (!s_ok) return;
        STAMP(0, 8);
        float actv = 0.f, dap[2 * SP], qa[SP], qb[SP], lsv = 0.f, epv = 0.f, okv = 0.f;
#pragma unroll
        for (int p = 0; p < 2 * SP; ++p) dap[p] = 0.f;
#pragma unroll
        for (int p = 0; p < SP; ++p) {                           // Q1, Q2(s, a_new) partials of this row
            qa[p] = ld_sc1(d.qpart2 + ((size_t)(2 * NB + rb) * SP + p) * 32 + row);
            qb[p] = ld_sc1(d.qpart2 + ((size_t)(3 * NB + rb) * SP + p) * 32 + row);
        }
        if (a < A) {
#pragma unroll
            for (int p = 0; p < 2 * SP; ++p) dap[p] = ld_sc1(d.dapart + (size_t)p * B * 16 + gi);
            const f32x4 hv = ld4(d.hv4 + (size_t)gi * 4);     // (first touched here, behind the counter: exclusive 1-KB blocks)
            actv = hv[0]; lsv = hv[1]; epv = hv[2]; okv = hv[3];
        }
        float qnew1 = 0.f, qnew2 = 0.f, dz = 0.f, dls = 0.f;
        {
            float va = qa[0], vb = qb[0];
#pragma unroll
            for (int p = 1; p < SP; ++p) { va += qa[p]; vb += qb[p]; }                   // fixed order
            va += b3a; vb += b3b;                                                        // Q1, Q2(s, a_new)
            // torch.min backward: the smaller one takes the gradient, a tie splits it
            const float sel1 = (va < vb) ? 1.0f : ((va == vb) ? 0.5f : 0.0f);
            const float dq1 = -invB * sel1, dq2 = -invB * (1.0f - sel1);
            qnew1 = va; qnew2 = vb;
            if (a < A && grow < d.Bt) {
                float da1 = dap[0], da2 = dap[SP];
#pragma unroll
                for (int p = 1; p < SP; ++p) { da1 += dap[p]; da2 += dap[SP + p]; }      // fixed order
                const float da = actor_da(da1, dq1, da2, dq2);
                const float om = 1.0f - actv * actv;
                const float alpha_invB = __fmul_rn(alpha, invB);
                dz = actor_dz(da, om, alpha_invB, actv);
                const float stdv2 = expf(lsv);
                dls = actor_dls(dz, stdv2, epv, alpha_invB, okv);
                XH[lds_off(row, A + a, 64)] = dls;
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
            f32x4 acc[1] = {};
            gemm_ring(r1, X2, H, 8, acc, 8 * kh);
            if (net == 0 && part == 0) {
                if (a == 0) { d.q[2 * (size_t)B + grow] = qnew1; d.q[3 * (size_t)B + grow] = qnew2; }
                if (a < A) {
                    d.dheadT[frag_off(a, grow, B)] = dz;
                    d.dheadT[frag_off(A + a, grow, B)] = dls;
                }
            }
            if (net == 0 && wave == part) store_features<4>(gk2, 64 * wave, 16, d.dPH2T, B, row0);
            // the two halves of the contraction meet in LDS (fixed order: first half + second half)
            float *hx = X2 + RB * H;         // 2 tiles x 64 lanes x 4 floats behind the row-block
            if (kh == 1) st4(hx + ((wave & 1) * 64 + lane) * 4, acc[0]);
            lds_barrier();
            if (kh == 0) {
                const f32x4 o = ld4(hx + ((wave & 1) * 64 + lane) * 4);
                f32x4 gv;
#pragma unroll
                for (int i = 0; i < 4; ++i) gv[i] = (h1v[0][i] > 0.f) ? (acc[0][i] + o[i]) : 0.f;
                st4(d.dPH1T + frag_off(nt + c, row0 + 4 * g, B), gv);
            }
        }
        STAMP(0, 9);
    }
}
