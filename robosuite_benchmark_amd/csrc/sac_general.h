// The SAC step for network shapes beyond the fast kernels' two hidden layers of at most 256 units
// (variant['policy_kwargs'|'qf_kwargs']['hidden_sizes'], /root/reference/util/arguments.py:98,104 -> FlattenMlp /
// TanhGaussianPolicy at /root/reference/util/rlkit_utils.py:64-97): any depth, any width.  Included by sac_trainer.hip
// inside namespace sac (device part); the host part is sac_general_host.h.
//
// Same step, same order, as a sequence of launches instead of fused row-block kernels:
//   k_g_gemm     ONE kernel for every matrix product of the step -- layer forward (X W^T + b, relu), backward through a
//                layer (dY W, masked by the relu of the layer below), weight gradient (dY^T X, the bias gradient as an
//                extra column of ones) -- as a table of jobs per launch: a launch carries every network's job of one
//                stage (Q1, Q2 and both targets' layer l are ONE launch).  64 x 64 output tiles, fp32 MFMA 16x16x4,
//                operands staged through LDS with generic strides.
//   k_g_head     the policy's head layer (2A outputs) and the tanh-Gaussian head on both policy passes (rsample, log-prob); the
//                Q nets' input rows [obs|act ; obs|a_new ; next_obs|a']; its last workgroup: mean(log_pi) -> Adam step on log_alpha
//   k_g_loss     the Q nets' last layers (one output), min over the twin nets, Bellman target, the loss gradients of the four
//                Q passes that have one, and the backward pass through those last layers
//   k_g_polgrad  dQ/da through the action columns of the Q nets' first layers, head gradient of the reparameterised actor loss,
//                backward pass through the head layer
//   k_g_diag     the diagnostics vector (Adam and the Polyak average run in the weight-gradient launch's epilogue: a tile's
//                owner updates the parameters it owns)
// Weights live in nn.Linear layout (W [out][in] row-major, then b), activations row-major [row][feature]; rows = the
// TRUE batch (no row-block padding).  2 Lp + 2 Lq + 2 launches per step (10 for two hidden layers; + one for the diagnostics on the steps somebody reads
// them; the layers of a handful of outputs ride in k_g_head / k_g_loss / k_g_polgrad and their TD3 counterparts): this path is for
// shapes the reference can be configured with but no shipped variant uses -- the shipped ones take the fused kernels.
#pragma once

namespace gen {

constexpr int GMAXL = 8;            // layers of one network: up to 7 hidden layers + the output layer
constexpr int GMAXJ = 24;           // jobs of one launch (the weight-gradient launch: 3 networks x GMAXL)
constexpr int GT = 64;              // output tile (rows and columns)
constexpr int GLD = GT + 4;         // LDS row stride of a staged operand chunk
constexpr int GK = 64;              // reduction chunk
constexpr int GW = 8;               // waves of a workgroup: two per SIMD (one wave's LDS reads and address arithmetic under the other's MFMAs)
constexpr int GE = GT * GK / (64 * GW);   // values of an operand's chunk per thread (8)
constexpr int GNT = 16 / GW;        // 16 x 16 output tiles per wave (2): row tile wave & 3, column tiles GNT (wave >> 2) + t

struct GemmJob {
    const float *A;                 // A(m, r) = A[m sa_m + r sa_r]
    const float *Bm;                // B(n, r) = Bm[n sb_n + r sb_r]
    float *C;                       // C(m, n) = sum_r A(m, r) B(n, r)  [+ bias(n)]  [relu]  [0 where mask(m, n) <= 0]
    const float *bias, *mask;
    float *c_ones;                  // ones_col: B gets a column n == N of ones, whose sums go here (c_ones[m]): a bias gradient
    long long sa_m, sa_r, sb_n, sb_r, ldc, ldmask;
    long long a_off, b_off;         // a_slot / b_slot: the operand lives in the step's minibatch slot, at this offset (floats)
    int M, N, R;
    int relu, ones_col, tiles_n, tile0, a_slot, b_slot;
    int a_vec, b_vec, pad_;         // the operand may be fetched in 16-byte pieces (alignment and extents checked by the host)
    // weight-gradient jobs: the tile's owner applies torch.optim.Adam (and the Polyak average of the target) in its epilogue --
    // C is the layer's slice of the gradient vector, these are the same slice of the parameter / moment / target vectors
    // (aX: the weights, bX: the bias behind the column of ones); null: a plain product
    float *aP, *aM, *aV, *aTP, *bP, *bM, *bV, *bTP;
    float lr; int pad2_;
};
struct GemmStage {                  // kernel argument: the jobs of one launch (device array) + their first tiles
    const GemmJob *jobs;
    const float *S;                 // the step's minibatch slot
    int stamp;                      // (diagnostic build, -DSAC_STAMPS: this launch records its in-kernel timeline)
    // split reduction (launches with few tiles: a 256-row batch leaves most of the chip idle): `splitk` workgroups share a
    // tile, each takes a contiguous range of the reduction's chunks and leaves its partial tile in `scratch`; the LAST one
    // to finish adds the partials in the fixed order 0 .. splitk-1 (deterministic whatever the arrival order) and runs the
    // epilogue.  tile_cnt: one arrival counter per tile, back at 0 when the launch ends.
    int splitk;
    float *scratch;
    unsigned *tile_cnt;
    // the optimizer step of a weight-gradient launch (host state, as StepArg): bias corrections, Polyak this step?, keep the
    // gradient vector (sac_debug_fetch "g_*": the steps whose caller reads the diagnostics)?
    double bc1, bc2s;
    float tau; int polyak, keep_grad, pad3_;
    int njobs, ntiles;
    int tile0[GMAXJ];
};

// everything the elementwise kernels need
struct GDev {
    int n, O, A, NI;                // rows (true batch); NI: noise index stride per row (16 up to 16 actions, else 64)
    int ldq;                        // O + A
    float discount, reward_scale, tau, target_entropy, alpha_lr;
    int period, auto_alpha;
    unsigned long long noise_seed;
    Ctl *ctl;
    float *XQ;                      // the Q nets' input rows [3n][O + A]
    const float *HD;                // head pre-activations [2n][2A]: mean | log-std
    float *mu, *ls, *ok, *epsv, *anew, *a2, *logpi, *logpi2;
    unsigned *done;                 // workgroups of the head launch that have finished (the last one takes the entropy step)
    const float *QO[4];             // q1 [2n] (s,a | s,a_new), q2 [2n], target q1 [n], target q2 [n]
    float *DQ[2];                   // dL/dq of the critic rows | of the actor rows [2n]
    float *y, *qn;                  // Bellman target [n]; min Q(s, a_new) [n]
    const float *DA[2];             // actor-loss gradient w.r.t. a_new through Q1 / Q2 [n][A]
    float *DHD;                     // head gradient [n][2A]
    // SAC, the layers of one or two dozen outputs ride in the elementwise kernels (no matrix-product launch for them):
    const float *PHl; int KPl;      // the policy's last hidden activations [2n][KPl]
    const float *Wh, *bh;           // its merged head layer [2A][KPl], [2A]
    float *HDw;                     // (HD as this step's head kernel writes it)
    const float *QHl[2], *THl[2];   // last hidden activations of Q1, Q2 [2n][KQl] and of the targets [n][KQl]
    int KQl;
    const float *Wl[4], *bl[4];     // last layers (one output) of Q1, Q2, T1, T2: [KQl], [1]
    float *QOw[4];
    float *dQZl[2];                 // dL/d(last hidden) of Q1, Q2 [2n][KQl]
    const float *dQZ0[2]; int HQ0;  // dL/d(first hidden) of Q1, Q2 [2n][HQ0] (the actor rows start at n)
    const float *W1q[2];            // their first layers [HQ0][O + A]
    float *DAw[2];
    float *dPZl;                    // dL/d(the policy's last hidden) [n][KPl]
    float *diag_first, *diag_last, *diag_trace, *diag_dev;
    const float *eps1, *eps2;
    // TD3 (algo 1; rlkit TD3Trainer): XQ holds [(s, a) ; (s', a~)]; QO = {Q1, Q2 on (s, a), target Q1, Q2 on (s', a~)} [n] each
    int algo;
    float td3_sigma, td3_clip;
    const float *HDT, *HDP;         // head pre-activations [n][A]: target policy on s' / online policy on s
    float *XA, *pa;                 // the actor pass: Q1's input rows [obs | tanh(mean)] [n][O + A]; the policy action [n][A]
    const float *QA;                // Q1(s, pi(s)) [n]
    float *DQA;                     // its loss gradient (-1/n) [n]
    const float *DAa;               // dL/da through Q1 [n][A]
    float *DHP;                     // head gradient [n][A]
    // TD3, the small layers inside the elementwise kernels (as SAC's above; Wh / bh / KPl / PHl / dPZl / W1q[0] / Wl / bl / QOw / KQl /
    // QHl / THl / dQZl are shared): the target policy's last hidden activations and head [A][KPl]; Q1's hidden activations on
    // (s, pi(s)) and their gradients
    const float *PHTl, *WhT, *bhT;
    float *HDTw, *HDPw;
    const float *AHl; float *dAZl;  // [n][KQl]
    const float *dAZ0;              // [n][HQ0]
    float *QAw, *DAaw;
};

// ------------------------------------------------------------------------------------------
// the matrix-product kernel
// ------------------------------------------------------------------------------------------
// A workgroup (eight waves: two per SIMD) owns a 64 x 64 tile of C; wave w the row tile w & 3 and two of the four column
// tiles (two 16 x 16 MFMA accumulators).  The reduction runs in chunks of 64: each thread fetches 8 values per operand (a wave reads 256 contiguous bytes along whichever
// direction the operand is contiguous in), the chunk goes to LDS as [slow][fast] -- [row][r] for an operand that is
// contiguous along the reduction, [r][row] otherwise: conflict-free writes either way -- and the next chunk's loads are in
// flight while this one's 32 MFMAs per wave run.  The MFMA with index i of k-group (q, g) contracts r = 16 q + 4 g + i for
// both operands, so an operand stored [row][r] is read with one 16-byte LDS load per four MFMAs.
// One operand's share of a chunk in one thread: 8 values v[k] at chunk coordinates (slow, fast) -- for an operand that is
// contiguous along the reduction (RC) that is (row, r), else (r, row); the chunk sits in LDS as [slow][fast].
//   scalar map: slow = wave + 8 k, fast = lane               (8 dword loads, a wave reads 256 contiguous bytes)
//   vector map: slow = (tid >> 4) + 32 (k >> 2), fast = 4 (tid & 15) + (k & 3)
//               (2 loads of 16 bytes, 2 LDS writes of 16 bytes: when base, row stride and extent along `fast` allow it)
// Loads are UNCONDITIONAL from clamped indices -- a row beyond the matrix repeats the last one (its products land in
// outputs that are never stored), the reduction's padding is zeroed in the edge chunk only -- and go through explicitly
// GLOBAL pointers.  Both matter: a conditional load is a branch whose merge point waits for the data, and a generic
// pointer (these come out of a table read with scalar loads) makes a flat load, which counts on lgkmcnt too, so that
// every wait for an LDS read of the chunk being multiplied also waited for the next chunk's loads.  A FULL chunk is
// loaded at uniform base + constant 32-bit lane offsets (offsets are 32-bit: checked at creation).
template <bool RC>
struct GOperand {
    const float *P;
    int sx, sr, X, R, x0, ones_x;     // strides (floats), rows, reduction length, the tile's first row, local row of ones (-1: none)
    bool vec;
    unsigned off[GE];
    float v[GE];
    __device__ __forceinline__ int slow_of(int k) const {
        const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        return vec ? (tid >> 4) + 4 * GW * (k >> 2) : wv + GW * k;
    }
    __device__ __forceinline__ int fast_of(int k) const {
        const int tid = threadIdx.x;
        return vec ? 4 * (tid & 15) + (k & 3) : (tid & 63);
    }
    __device__ __forceinline__ unsigned addr(int row_l, int r) const {       // clamped element offset
        return (unsigned)(min(x0 + row_l, X - 1) * sx + min(r, R - 1) * sr);
    }
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int k = 0; k < GE; ++k) off[k] = RC ? addr(slow_of(k), fast_of(k)) : addr(fast_of(k), slow_of(k));
        // (vector map, k = 4 j: the 16 bytes at off[4 j]; an operand that is contiguous along its rows clamps whole vectors)
        if (vec && !RC) {
#pragma unroll
            for (int j = 0; j < GE / 4; ++j) off[4 * j] = (unsigned)(min(x0 + fast_of(4 * j), X - 4) * sx + min(slow_of(4 * j), R - 1) * sr);
        }
    }
    __device__ __forceinline__ void fetch(int s) {
        const int r0 = GK * s;
        if (r0 + GK <= R) {
            const float *cb = P + (long long)r0 * sr;                       // uniform
            if (vec) {
#pragma unroll
                for (int j = 0; j < GE / 4; ++j) {
                    const f32x4 q = *(const __attribute__((address_space(1))) f32x4 *)(uintptr_t)(cb + off[4 * j]);
                    v[4 * j] = q[0]; v[4 * j + 1] = q[1]; v[4 * j + 2] = q[2]; v[4 * j + 3] = q[3];
                }
            } else {
#pragma unroll
                for (int k = 0; k < GE; ++k) v[k] = ld1g(cb + off[k]);
            }
            return;
        }
        // the edge chunk: reduction indices clamped (their products are zeroed by fix); whole vectors stay inside R (R % 4 == 0)
        if (vec) {
#pragma unroll
            for (int j = 0; j < GE / 4; ++j) {
                const unsigned o = RC ? (unsigned)(min(x0 + slow_of(4 * j), X - 1) * sx + min(r0 + fast_of(4 * j), R - 4) * sr)
                                      : (unsigned)(min(x0 + fast_of(4 * j), X - 4) * sx + min(r0 + slow_of(4 * j), R - 1) * sr);
                const f32x4 q = *(const __attribute__((address_space(1))) f32x4 *)(uintptr_t)(P + o);
                v[4 * j] = q[0]; v[4 * j + 1] = q[1]; v[4 * j + 2] = q[2]; v[4 * j + 3] = q[3];
            }
        } else {
#pragma unroll
            for (int k = 0; k < GE; ++k) v[k] = ld1g(P + (RC ? addr(slow_of(k), r0 + fast_of(k)) : addr(fast_of(k), r0 + slow_of(k))));
        }
    }
    // behind the loads' arrival, in front of the LDS writes: the row of ones, the reduction's zero padding.  (A vector that
    // was clamped back into the matrix holds OTHER elements than its coordinates say: rows beyond X only, whose products
    // are never stored -- except the row of ones, which is set here by coordinate.)
    __device__ __forceinline__ void fix(int s) {
        const int r0 = GK * s;
        if (ones_x >= 0) {
#pragma unroll
            for (int k = 0; k < GE; ++k) if ((RC ? slow_of(k) : fast_of(k)) == ones_x) v[k] = 1.0f;
        }
        if (r0 + GK > R) {
#pragma unroll
            for (int k = 0; k < GE; ++k) if (r0 + (RC ? fast_of(k) : slow_of(k)) >= R) v[k] = 0.f;
        }
    }
    __device__ __forceinline__ void write(float *lds) const {
        if (vec) {
#pragma unroll
            for (int j = 0; j < GE / 4; ++j) st4(lds + slow_of(4 * j) * GLD + fast_of(4 * j), f32x4{v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]});
        } else {
#pragma unroll
            for (int k = 0; k < GE; ++k) lds[slow_of(k) * GLD + fast_of(k)] = v[k];
        }
    }
};

// A_RC / B_RC (compile time: the jobs of a launch share them): the operand is contiguous along the reduction -- forward
// (true, true), backward through a layer (true, false), weight gradient (false, false).  Only the thread -> element map
// and the LDS layout depend on it; the addresses always use the job's strides.
template <bool A_RC, bool B_RC>
__global__ __launch_bounds__(64 * GW) void k_g_gemm(GemmStage T) {
    __shared__ __attribute__((aligned(16))) float As[GT * GLD], Bs[GT * GLD];
    const int splitk = T.splitk, tile_lin = (int)blockIdx.x / splitk, ks = (int)blockIdx.x - tile_lin * splitk;
    int li = 0;
#pragma unroll
    for (int q = 1; q < GMAXJ; ++q) li = (tile_lin >= T.tile0[q]) ? q : li;
    union { GemmJob J; unsigned long long w[sizeof(GemmJob) / 8]; } ud;
    {
        const __attribute__((address_space(4))) unsigned long long *src =
            (const __attribute__((address_space(4))) unsigned long long *)(uintptr_t)(T.jobs + li);
#pragma unroll
        for (int q = 0; q < (int)(sizeof(GemmJob) / 8); ++q) ud.w[q] = src[q];
    }
    const GemmJob &J = ud.J;
    const float *const Ap = J.a_slot ? T.S + J.a_off : J.A, *const Bp = J.b_slot ? T.S + J.b_off : J.Bm;
    const int tile = tile_lin - J.tile0;
    const int m0 = GT * (tile / J.tiles_n), n0 = GT * (tile % J.tiles_n);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int rt = wave & 3, ct0 = GNT * (wave >> 2);          // this wave's row tile and its first column tile
    constexpr bool a_rc = A_RC, b_rc = B_RC;
    GOperand<a_rc> oa;
    GOperand<b_rc> ob;
    oa.P = Ap; oa.sx = (int)J.sa_m; oa.sr = (int)J.sa_r; oa.X = J.M; oa.R = J.R; oa.x0 = m0; oa.ones_x = -1; oa.vec = J.a_vec != 0;
    ob.P = Bp; ob.sx = (int)J.sb_n; ob.sr = (int)J.sb_r; ob.X = J.N; ob.R = J.R; ob.x0 = n0; ob.vec = J.b_vec != 0;
    ob.ones_x = (J.ones_col && n0 <= J.N && J.N < n0 + GT) ? J.N - n0 : -1;
    const bool has_ones = ob.ones_x >= 0;
    oa.init(); ob.init();
#ifdef SAC_STAMPS
#define GSTAMP(i) do { if (T.stamp) STAMP(0, i); } while (0)
#else
#define GSTAMP(i) do { } while (0)
#endif
    f32x4 acc[GNT] = {};
    const int nSall = (J.R + GK - 1) / GK;
    const int sfirst = (ks * nSall) / splitk, nS = ((ks + 1) * nSall) / splitk;       // this workgroup's chunks [sfirst, nS)
    GSTAMP(0);
    oa.fetch(sfirst); ob.fetch(sfirst);
    GSTAMP(1);
    for (int s = sfirst; s < nS; ++s) {
        if (s > sfirst) __syncthreads();
        if (s == 1) GSTAMP(2);
        oa.fix(s); ob.fix(s);
        oa.write(As); ob.write(Bs);
        __syncthreads();
        if (s == 1) GSTAMP(3);
        if (s + 1 < nS) { oa.fetch(s + 1); ob.fetch(s + 1); }
        if (s == 1) GSTAMP(4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 a, b[GNT];
            if constexpr (a_rc) a = ld4(As + (16 * rt + c) * GLD + 16 * q + 4 * g);
            else {
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = As[(16 * q + 4 * g + i) * GLD + 16 * rt + c];
            }
            if constexpr (b_rc) {
#pragma unroll
                for (int t = 0; t < GNT; ++t) b[t] = ld4(Bs + (16 * (ct0 + t) + c) * GLD + 16 * q + 4 * g);
            } else {
#pragma unroll
                for (int t = 0; t < GNT; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) b[t][i] = Bs[(16 * q + 4 * g + i) * GLD + 16 * (ct0 + t) + c];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < GNT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[t][i], acc[t], 0, 0, 0);
        }
        if (s == 1) { asm volatile("" :: "v"(acc[0][0]), "v"(acc[GNT - 1][3])); GSTAMP(5); }
    }
    GSTAMP(6);
    if (splitk > 1) {
        __shared__ unsigned am_last;
        float *mine = T.scratch + ((size_t)tile_lin * splitk + ks) * (GT * GT);
#pragma unroll
        for (int t = 0; t < GNT; ++t) st4_sc1(mine + (t * 64 * GW + tid) * 4, acc[t]);
        // (the hand-off protocol of the fused step, sac_fused.h: write-through stores, every wave waits for its own, a
        //  workgroup barrier, ONE relaxed agent-scope increment -- a __threadfence() / an acq_rel atomic here writes the
        //  whole L2 back and invalidates it: 27 us per launch, measured)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(T.tile_cnt + tile_lin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            am_last = (old == (unsigned)splitk - 1u) ? 1u : 0u;
            if (am_last) __hip_atomic_store(T.tile_cnt + tile_lin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!am_last) return;
        // (all partials requested at once -- fetched one dependent load at a time this tail took 28 us -- then added in order.
        //  Plain loads are coherent here: the partials were written through (sc1) and fenced before their counter
        //  increment, and this workgroup touches these lines for the first time in a launch that began with invalidated
        //  caches)
        const float *all = T.scratch + (size_t)tile_lin * splitk * (GT * GT);
        f32x4 part[4][GNT];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int t = 0; t < GNT; ++t)
                part[k][t] = *(const __attribute__((address_space(1))) f32x4 *)(uintptr_t)(all + (size_t)(k < splitk ? k : 0) * (GT * GT) + (t * 64 * GW + tid) * 4);
        SB();
#pragma unroll
        for (int t = 0; t < GNT; ++t) {
            acc[t] = part[0][t];
#pragma unroll
            for (int k = 1; k < 4; ++k) if (k < splitk) acc[t] += part[k][t];
        }
    }
    if constexpr (!A_RC && !B_RC) {
        if (J.aP) {     // a weight-gradient tile: Adam (+ Polyak) on the parameters it owns; the gradient never round-trips through HBM
            const unsigned ldc = (unsigned)J.ldc;
            const bool polyak = T.polyak && J.aTP;
            const float step_size = (float)((double)J.lr / T.bc1), bc2s = (float)T.bc2s;
            float pv[GNT][4], mv[GNT][4], vv[GNT][4], tv[GNT][4];
#pragma unroll
            for (int t = 0; t < GNT; ++t) {
                const int n = n0 + 16 * (ct0 + t) + c;
                const bool ones = has_ones && n == J.N;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned mc = (unsigned)min(m0 + 16 * rt + 4 * g + i, J.M - 1);
                    const unsigned off = ones ? mc : mc * ldc + (unsigned)min(n, J.N - 1);
                    pv[t][i] = ld1g((ones ? J.bP : J.aP) + off);
                    mv[t][i] = ld1g((ones ? J.bM : J.aM) + off);
                    vv[t][i] = ld1g((ones ? J.bV : J.aV) + off);
                    tv[t][i] = polyak ? ld1g((ones ? J.bTP : J.aTP) + off) : 0.f;
                }
            }
            SB();
#pragma unroll
            for (int t = 0; t < GNT; ++t) {
                const int n = n0 + 16 * (ct0 + t) + c;
                const bool ones = has_ones && n == J.N;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = m0 + 16 * rt + 4 * g + i;
                    if (m >= J.M || !(n < J.N || ones)) continue;
                    const unsigned off = ones ? (unsigned)m : (unsigned)m * ldc + (unsigned)n;
                    float p = pv[t][i], mm = mv[t][i], v = vv[t][i];
                    adam_update(p, mm, v, acc[t][i], step_size, bc2s);
                    typedef __attribute__((address_space(1))) float gfloat;
                    *(gfloat *)(uintptr_t)((ones ? J.bP : J.aP) + off) = p;
                    *(gfloat *)(uintptr_t)((ones ? J.bM : J.aM) + off) = mm;
                    *(gfloat *)(uintptr_t)((ones ? J.bV : J.aV) + off) = v;
                    if (T.keep_grad) *(gfloat *)(uintptr_t)((ones ? J.c_ones : J.C) + off) = acc[t][i];
                    if (polyak) *(gfloat *)(uintptr_t)((ones ? J.bTP : J.aTP) + off) = tv[t][i] * (1.0f - T.tau) + p * T.tau;
                }
            }
            GSTAMP(7);
            return;
        }
    }
    // epilogue: every load up front (clamped, unconditional, pinned in front of the arithmetic), stores through global
    // pointers at 32-bit offsets, one predicate per element -- written with early-outs and conditional loads it compiled to
    // a branch and a full wait per element: 2.2 us of a forward launch, 4.7 us of a masked one (in-kernel stamps)
    float bv[GNT], mk[GNT][4];
    const unsigned ldc = (unsigned)J.ldc, ldm = (unsigned)J.ldmask;
#pragma unroll
    for (int t = 0; t < GNT; ++t) {
        const unsigned nc = (unsigned)min(n0 + 16 * (ct0 + t) + c, J.N - 1);
        bv[t] = ld1g((J.bias ? J.bias : Bp) + (J.bias ? nc : 0u));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned mc = (unsigned)min(m0 + 16 * rt + 4 * g + i, J.M - 1);
            mk[t][i] = ld1g((J.mask ? J.mask : Bp) + (J.mask ? mc * ldm + nc : 0u));
        }
    }
    SB();
    const bool has_bias = J.bias != nullptr, has_mask = J.mask != nullptr;
    float *const Cg = J.C;
#pragma unroll
    for (int t = 0; t < GNT; ++t) {
        const int n = n0 + 16 * (ct0 + t) + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + 16 * rt + 4 * g + i;
            float v = acc[t][i] + (has_bias ? bv[t] : 0.f);
            if (J.relu) v = fmaxf(v, 0.f);
            if (has_mask) v = (mk[t][i] > 0.f) ? v : 0.f;
            if (m < J.M && n < J.N) *(__attribute__((address_space(1))) float *)(uintptr_t)(Cg + ((unsigned)m * ldc + (unsigned)n)) = v;
        }
    }
    if (has_ones) {
#pragma unroll
        for (int t = 0; t < GNT; ++t)
            if (n0 + 16 * (ct0 + t) + c == J.N) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = m0 + 16 * rt + 4 * g + i;
                    if (m < J.M) *(__attribute__((address_space(1))) float *)(uintptr_t)(J.c_ones + m) = acc[t][i];
                }
            }
    }
    GSTAMP(7);
}

// ------------------------------------------------------------------------------------------
// elementwise kernels
// ------------------------------------------------------------------------------------------
// TanhGaussianPolicy.forward(reparameterize=True, return_log_prob=True) INCLUDING its head layer, both passes: one workgroup
// per GRW rows (rows 0 .. n-1: policy(s), n .. 2n-1: policy(s')).  The head layer -- mean and log-std rows merged, 2A <= 32 outputs
// -- is no matrix-product launch of its own: thread (kg, j) = (t >> 4, t & 15) contracts the inputs k = kg, kg + 16, ... for
// outputs j (mean) and A + j (log-std), the sixteen partials meet in LDS in a fixed order, and threads 0 .. 15 go on as
// (row, action).  The kernel also assembles the Q nets' input rows -- the torch.cat of
// FlattenMlp: [obs|act ; obs|a_new ; next_obs|a'] -- and its LAST workgroup to finish takes the entropy step (SURVEY
// Appendix A lines 4-6: alpha_loss = -mean(log_alpha (log_pi + H)), one Adam step on log_alpha, alpha = exp(.) post-step;
// the sum runs in a fixed order).  Rows written by other workgroups are read back through agent-scope loads.
constexpr int GRW = 4;      // rows per workgroup of the kernels that hold a small layer (weights loaded once per GRW rows)
constexpr int GCK = 256;    // their reduction chunk: one column per thread
// A layer of at most 32 outputs on rows r0 .. r0 + GRW - 1 of X [nrows][K] (a policy's head layer), without a matrix-product
// launch: thread (kg, j) = (t >> 4, t & 15) contracts the inputs k = kg, kg + 16, ... for output j (and, `two`, output A + j: the
// log-std row of SAC's merged heads); the sixteen partials per output are left in part[row][kg][j | 16 + j] for the caller to
// add in a fixed order.  Every weight is loaded once for the GRW rows.  The reduction runs in chunks of GCK inputs
// staged through LDS: a thread fetches ONE column of the chunk -- its GRW activations and the outputs' weights, every load
// independent and coalesced, the next chunk's in flight under this chunk's arithmetic -- so a chunk costs one memory
// latency (a loop that loaded where it multiplied, sixteen strided rows per wave-load and a latency per iteration, took
// 12 us at 512 inputs under rocprofv3).
// (NJ: compile-time bound of the outputs, 8 / 16 / 32 -- the unrolled per-output code is what these kernels' run time is made of)
template <int NJ>
__device__ __forceinline__ void small_layer_rows(const float *X, int K, const float *W, int A, bool two, int r0, int nrows,
                                                 float (*part)[16][33], float (*xs)[GCK], float (*ws)[GCK + 1]) {
    const int a = threadIdx.x & 15, kg = threadIdx.x >> 4;
    const int t = threadIdx.x, nj = two ? 2 * A : A;
    float sm[GRW], sr[GRW], rx[GRW], rw[NJ];
#pragma unroll
    for (int q = 0; q < GRW; ++q) { sm[q] = 0.f; sr[q] = 0.f; }
    // (loads are unconditional with a clamped column -- a lane-dependent condition around a load is a branch per load, and with
    //  64-bit index arithmetic these kernels were thousands of instructions long: one wave per SIMD executes them end to end, which
    //  is what their 12-15 us were; the activation of a column beyond K is zeroed instead, so its weights never count.  Offsets fit
    //  32 bits: checked at creation.)
    int xoff[GRW];
#pragma unroll
    for (int q = 0; q < GRW; ++q) xoff[q] = (r0 + q < nrows ? r0 + q : r0) * K;
    auto fetch = [&](int k0) {
        const int k = k0 + t;
        const bool in = k < K;
        const int kc = in ? k : K - 1;
#pragma unroll
        for (int q = 0; q < GRW; ++q) { const float v = X[xoff[q] + kc]; rx[q] = in ? v : 0.f; }
#pragma unroll
        for (int j = 0; j < NJ; ++j) if (j < nj) rw[j] = W[j * K + kc];
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += GCK) {
        __syncthreads();                              // (the previous chunk's readers are done)
#pragma unroll
        for (int q = 0; q < GRW; ++q) xs[q][t] = rx[q];
#pragma unroll
        for (int j = 0; j < NJ; ++j) if (j < nj) ws[j][t] = rw[j];
        __syncthreads();
        if (k0 + GCK < K) fetch(k0 + GCK);
        const float *wmr = ws[a < A ? a : 0], *wrr = ws[two ? A + (a < A ? a : 0) : 0];
#pragma unroll
        for (int i = 0; i < GCK / 16; ++i) {
            const int kk = kg + 16 * i;
            const float wmv = wmr[kk], wrv = wrr[kk];
#pragma unroll
            for (int q = 0; q < GRW; ++q) {
                const float xv = xs[q][kk];
                sm[q] = fmaf(xv, wmv, sm[q]);
                sr[q] = fmaf(xv, wrv, sr[q]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < GRW; ++q) { part[q][kg][a] = sm[q]; part[q][kg][16 + a] = sr[q]; }
    __syncthreads();
}

template <int MA>       // MA: 8 for up to eight actions, else 16
__global__ __launch_bounds__(256) void k_g_head(GDev d, const float *__restrict__ S, SlotLayout SL, StepArg sa) {
    const int n = d.n, O = d.O, A = d.A, ldq = d.ldq;
    __shared__ float part[GRW][16][33];
    __shared__ float xs[GRW][GCK];
    __shared__ float ws[32][GCK + 1];
    {
        const int w = 2 * O + A, tot = n * w;               // (32-bit: checked at creation)
        for (int e = (int)(blockIdx.x * 256 + threadIdx.x); e < tot; e += (int)(gridDim.x * 256)) {
            const int b = e / w, k = e - b * w;
            if (k < O) {
                const float v = S[SL.off_obs + (long long)b * O + k];
                d.XQ[(long long)b * ldq + k] = v;
                d.XQ[(long long)(n + b) * ldq + k] = v;
            } else if (k < 2 * O) {
                d.XQ[(long long)(2 * n + b) * ldq + (k - O)] = S[SL.off_nobs + (long long)b * O + (k - O)];
            } else {
                d.XQ[(long long)b * ldq + O + (k - 2 * O)] = S[SL.off_act + (long long)b * A + (k - 2 * O)];
            }
        }
    }
    const int r0 = blockIdx.x * GRW, a = threadIdx.x & 15;
    small_layer_rows<2 * MA>(d.PHl, d.KPl, d.Wh, A, true, r0, 2 * n, part, xs, ws);
    const int rr = threadIdx.x >> 4, r = r0 + rr;
    if (rr < GRW && r < 2 * n) {
        const int side = r >= n ? 1 : 0, b = r - side * n;
        const float *epp = side ? d.eps2 : d.eps1;
        float lp = 0.f;
        if (a < A) {
            float mean = part[rr][0][a], raw = part[rr][0][16 + a];
#pragma unroll
            for (int q = 1; q < 16; ++q) { mean += part[rr][q][a]; raw += part[rr][q][16 + a]; }      // fixed order
            mean += d.bh[a]; raw += d.bh[A + a];
            d.HDw[(long long)r * 2 * A + a] = mean; d.HDw[(long long)r * 2 * A + A + a] = raw;
            const float lstd = fminf(fmaxf(raw, LOG_SIG_MIN), LOG_SIG_MAX);
            const float stdv = expf(lstd);
            const float eps = epp ? epp[(long long)b * A + a]
                                  : philox_normal(d.noise_seed, (unsigned long long)sa.step_now, (unsigned)(b * d.NI + a), side ? 1u : 0u);
            const float zz = __fadd_rn(mean, __fmul_rn(stdv, eps));            // TanhNormal.rsample
            const float act = tanhf(zz);
            const float dd = __fsub_rn(zz, mean);                               // Normal.log_prob(z) - log(1 - a^2 + eps)
            const float var = __fmul_rn(stdv, stdv);
            const float nlp = -(dd * dd) / (2.0f * var) - logf(stdv) - 0.91893853320467274178f;
            lp = nlp - logf(1.0f - act * act + TANH_EPS);
            const long long gi = (long long)b * A + a;
            if (!side) {
                d.mu[gi] = mean; d.ls[gi] = lstd; d.ok[gi] = (raw >= LOG_SIG_MIN && raw <= LOG_SIG_MAX) ? 1.0f : 0.0f;
                d.epsv[gi] = eps; d.anew[gi] = act;
                d.XQ[(long long)(n + b) * ldq + O + a] = act;
            } else {
                d.a2[gi] = act;
                d.XQ[(long long)(2 * n + b) * ldq + O + a] = act;
            }
        }
        const float lsum = group16_sum(lp);
        if (a == 0) st_sc1((side ? d.logpi2 : d.logpi) + b, lsum);
    }
    // ---- the last workgroup: mean(log_pi) -> the entropy coefficient ----
    __shared__ float red[256];
    __shared__ unsigned am_last;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (log pi left through sc1 stores: see k_g_gemm's split tail)
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(d.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        am_last = (old == gridDim.x - 1) ? 1u : 0u;
        if (am_last) __hip_atomic_store(d.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // for the next step
    }
    __syncthreads();
    if (!am_last) return;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += ld_sc1(d.logpi + i);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    Ctl *c = d.ctl;
    if (!d.auto_alpha) { c->alpha = 1.0f; c->alpha_loss = 0.0f; return; }
    const float la = c->log_alpha, m0 = c->a_m, v0 = c->a_v;
    const float mean_lp = red[0] / (float)n + d.target_entropy;
    const float gr = -mean_lp;
    const float m = m0 + ADAM_1MB1 * (gr - m0);
    const float v = v0 * ADAM_B2 + ADAM_1MB2 * gr * gr;
    const float step_size = (float)((double)d.alpha_lr / sa.bc1);
    const float denom = sqrtf(v) / (float)sa.bc2s + 1e-8f;
    const float nla = la + (-step_size * m) / denom;
    c->alpha_loss = -((la * mean_lp) + 0.0f);
    c->log_alpha = nla; c->a_m = m; c->a_v = v; c->alpha = expf(nla);
}

// One workgroup per batch row b.  The Q nets' LAST layers (one output each: a dot product over the last hidden layer -- Q1, Q2
// on (s, a) and (s, a_new), the targets on (s', a')), then the min over the twin nets (actor loss, target), the Bellman target
// and dL/dq of the passes that carry a gradient -- critic rows 2 (q - y) / n, actor rows -1/n to the smaller of Q1, Q2(s, a_new)
// (torch.min: a tie splits it) -- and the backward pass through those last layers: dL/dh = dq w where h > 0.  Three launches
// (a matrix product with ONE output column, this kernel, a matrix product with a reduction of length one) as one.
__global__ __launch_bounds__(256) void k_g_loss(GDev d, const float *__restrict__ S, SlotLayout SL) {
    const int b = blockIdx.x, n = d.n, K = d.KQl;
    __shared__ float red[4][8];
    __shared__ float s_dq[4];
    const float *h[6] = {d.QHl[0] + (long long)b * K, d.QHl[0] + (long long)(n + b) * K, d.QHl[1] + (long long)b * K,
                         d.QHl[1] + (long long)(n + b) * K, d.THl[0] + (long long)b * K, d.THl[1] + (long long)b * K};
    const float *w[6] = {d.Wl[0], d.Wl[0], d.Wl[1], d.Wl[1], d.Wl[2], d.Wl[3]};
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // (thread 0's scalars are requested with the first operands: they would otherwise cost a memory latency behind the reduction)
    float p_alpha = 0.f, p_lp2 = 0.f, p_rew = 0.f, p_term = 0.f, p_b[4] = {0.f, 0.f, 0.f, 0.f};
    if (threadIdx.x == 0) {
        p_alpha = d.ctl->alpha; p_lp2 = d.logpi2[b]; p_rew = S[SL.off_rew + b]; p_term = S[SL.off_term + b];
#pragma unroll
        for (int q = 0; q < 4; ++q) p_b[q] = d.bl[q][0];
    }
#pragma unroll 4
    for (int k = threadIdx.x; k < K; k += 256) {
#pragma unroll
        for (int q = 0; q < 6; ++q) acc[q] = fmaf(h[q][k], w[q][k], acc[q]);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[q] += __shfl_xor(acc[q], o);
        if (lane == 0) red[wave][q] = acc[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float v[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) v[q] = ((red[0][q] + red[1][q]) + (red[2][q] + red[3][q]));      // fixed order
        const float q1 = v[0] + p_b[0], qa = v[1] + p_b[0], q2 = v[2] + p_b[1], qb = v[3] + p_b[1];
        const float t1 = v[4] + p_b[2], t2 = v[5] + p_b[3];
        const float invB = 1.0f / (float)n;
        const float tq = fminf(t1, t2) - p_alpha * p_lp2;
        const float y = bellman_target(d.reward_scale, p_rew, p_term, d.discount, tq);
        d.QOw[0][b] = q1; d.QOw[0][n + b] = qa; d.QOw[1][b] = q2; d.QOw[1][n + b] = qb; d.QOw[2][b] = t1; d.QOw[3][b] = t2;
        d.y[b] = y;
        d.qn[b] = fminf(qa, qb);
        const float sel1 = (qa < qb) ? 1.0f : ((qa == qb) ? 0.5f : 0.0f);
        const float g0 = 2.0f * (q1 - y) * invB, g1 = 2.0f * (q2 - y) * invB, g2 = -invB * sel1, g3 = -invB * (1.0f - sel1);
        d.DQ[0][b] = g0; d.DQ[1][b] = g1; d.DQ[0][n + b] = g2; d.DQ[1][n + b] = g3;
        s_dq[0] = g0; s_dq[1] = g2; s_dq[2] = g1; s_dq[3] = g3;          // in the order of h[0 .. 3]
    }
    __syncthreads();
    float *o[4] = {d.dQZl[0] + (long long)b * K, d.dQZl[0] + (long long)(n + b) * K, d.dQZl[1] + (long long)b * K,
                   d.dQZl[1] + (long long)(n + b) * K};
    for (int k = threadIdx.x; k < K; k += 256) {
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q][k] = (h[q][k] > 0.f) ? s_dq[q] * w[q][k] : 0.f;
    }
}

// One workgroup per GRW batch rows: the actor-loss gradient w.r.t. a_new through the FIRST layers of Q1 and Q2 (the action columns
// of W1: a dot product over the first hidden layer per action), d/d(mean, log_std) of mean(alpha log_pi - min Q) through
// a = tanh(mean + std eps), and the backward pass through the policy's head layer (2A terms per hidden unit, masked).  Three
// launches (a matrix product with A output columns, the elementwise kernel, a matrix product with a reduction of length 2A) as one.
template <int MA>       // MA: 8 for up to eight actions, else 16
__global__ __launch_bounds__(256) void k_g_polgrad(GDev d) {
    const int b0 = blockIdx.x * GRW, n = d.n, A = d.A, H0 = d.HQ0, ldq = d.ldq, t = threadIdx.x;
    __shared__ float part[GRW][16][33];
    __shared__ float dhd[GRW][32];
    __shared__ float gs[GRW][2][GCK];          // dL/d(first hidden) of Q1 | Q2, the actor rows of this workgroup, one chunk
    __shared__ float was[2][GCK][17];          // the action columns of W1 (Q1 | Q2), one chunk of hidden units
    const int a = t & 15, hg = t >> 4;
    // Everything that does not depend on dQ/da is requested FIRST (the kernel is a chain of dependent phases, each of which
    // would otherwise start with a cold memory latency of its own: 14.7 us under rocprofv3 for a few hundred kFLOP): the head
    // layer's weights and the relu masks of this thread's first hidden unit of the last phase, and what the sixteen
    // (row, action) threads per row need from the head kernel.
    const int K = d.KPl, nj = 2 * A;
    float wv[2 * MA], hm[GRW];
    {
        const int k = t < K ? t : 0;
#pragma unroll
        for (int j = 0; j < 2 * MA; ++j) wv[j] = (j < nj) ? d.Wh[j * K + k] : 0.f;
#pragma unroll
        for (int q = 0; q < GRW; ++q) hm[q] = d.PHl[(b0 + q < n ? b0 + q : b0) * K + k];
    }
    const int rr = t >> 4;
    const bool head_thread = rr < GRW && b0 + rr < n && a < A;
    float p_act = 0.f, p_ls = 0.f, p_eps = 0.f, p_ok = 0.f, p_alpha = 0.f;
    if (head_thread) {
        const long long e = (long long)(b0 + rr) * A + a;
        p_act = d.anew[e]; p_ls = d.ls[e]; p_eps = d.epsv[e]; p_ok = d.ok[e]; p_alpha = d.ctl->alpha;
    }
    {   // dQ/da: chunks of GCK hidden units through LDS (a thread fetches one hidden unit: 2 GRW gradients, 2 A weights)
        // (the A action columns of a chunk's GCK rows of W1 are fetched as the flat sequence (row, column): consecutive lanes read
        //  consecutive columns of a row, then the next row -- a wave-load touches ~64 / A rows.  One row per lane, the first
        //  cut, touched 64 rows per wave-load: 64 cache lines per instruction, 8 us of the kernel's 15.)
        // (unconditional loads with clamped rows, 32-bit offsets, the (row, column) of a lane's elements stepped without divisions:
        //  see small_layer_rows -- a hidden unit beyond H0 has a zero gradient, so whatever weight is loaded for it never counts)
        float s0[GRW], s1[GRW], rg[GRW][2], rwa[2][MA];
        const int q256 = 256 / A, r256 = 256 - q256 * A, hl_t = t / A, j_t = t - hl_t * A;
        const float *w0a = d.W1q[0] + d.O, *w1a = d.W1q[1] + d.O;
        int goff[GRW];
#pragma unroll
        for (int q = 0; q < GRW; ++q) { s0[q] = 0.f; s1[q] = 0.f; goff[q] = (n + (b0 + q < n ? b0 + q : b0)) * H0; }
        auto fetch = [&](int h0) {
            const int hh = h0 + t;
            const bool in = hh < H0;
            const int hc = in ? hh : H0 - 1;
#pragma unroll
            for (int q = 0; q < GRW; ++q) {
                const float v0 = d.dQZ0[0][goff[q] + hc], v1 = d.dQZ0[1][goff[q] + hc];
                rg[q][0] = in ? v0 : 0.f; rg[q][1] = in ? v1 : 0.f;
            }
            int hl = hl_t, j = j_t;                         // element e = t + 256 m of the chunk's GCK * A: (row e / A, column e % A)
#pragma unroll
            for (int m = 0; m < MA; ++m) {
                if (m < A) {
                    const int hr = h0 + hl < H0 ? h0 + hl : H0 - 1;
                    rwa[0][m] = w0a[hr * ldq + j]; rwa[1][m] = w1a[hr * ldq + j];
                }
                j += r256; hl += q256;
                if (j >= A) { j -= A; hl += 1; }
            }
        };
        fetch(0);
        for (int h0 = 0; h0 < H0; h0 += GCK) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < GRW; ++q) { gs[q][0][t] = rg[q][0]; gs[q][1][t] = rg[q][1]; }
            {
                int hl = hl_t, j = j_t;
#pragma unroll
                for (int m = 0; m < MA; ++m) {
                    if (m < A) { was[0][hl][j] = rwa[0][m]; was[1][hl][j] = rwa[1][m]; }
                    j += r256; hl += q256;
                    if (j >= A) { j -= A; hl += 1; }
                }
            }
            __syncthreads();
            if (h0 + GCK < H0) fetch(h0 + GCK);
#pragma unroll
            for (int i = 0; i < GCK / 16; ++i) {
                const int hl = hg + 16 * i;
                const float w0v = was[0][hl][a], w1v = was[1][hl][a];
#pragma unroll
                for (int q = 0; q < GRW; ++q) {
                    s0[q] = fmaf(gs[q][0][hl], w0v, s0[q]);
                    s1[q] = fmaf(gs[q][1][hl], w1v, s1[q]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < GRW; ++q) { part[q][hg][a] = s0[q]; part[q][hg][16 + a] = s1[q]; }
    }
    if (t < GRW * 32) dhd[t >> 5][t & 31] = 0.f;
    __syncthreads();
    if (head_thread) {
        const int b = b0 + rr;
        float da0 = part[rr][0][a], da1 = part[rr][0][16 + a];
#pragma unroll
        for (int q = 1; q < 16; ++q) { da0 += part[rr][q][a]; da1 += part[rr][q][16 + a]; }          // fixed order
        const long long e = (long long)b * A + a;
        d.DAw[0][e] = da0; d.DAw[1][e] = da1;
        const float alpha_invB = __fmul_rn(p_alpha, 1.0f / (float)n);
        const float da = da0 + da1;
        const float act = p_act;
        const float om = 1.0f - act * act;
        const float dz = actor_dz(da, om, alpha_invB, act);
        const float dls = actor_dls(dz, expf(p_ls), p_eps, alpha_invB, p_ok);
        d.DHD[(long long)b * 2 * A + a] = dz;
        d.DHD[(long long)b * 2 * A + A + a] = dls;
        dhd[rr][a] = dz; dhd[rr][A + a] = dls;
    }
    __syncthreads();
    // backward through the head layer: a thread owns hidden unit k -- its 2A weights and GRW masks in one round of loads (the
    // first unit's came in at the start; the next unit's are requested before this one's arithmetic)
    for (int k = t; k < K; k += 256) {
        float wc[2 * MA], hc[GRW];
#pragma unroll
        for (int j = 0; j < 2 * MA; ++j) wc[j] = wv[j];
#pragma unroll
        for (int q = 0; q < GRW; ++q) hc[q] = hm[q];
        if (k + 256 < K) {
#pragma unroll
            for (int j = 0; j < 2 * MA; ++j) wv[j] = (j < nj) ? d.Wh[j * K + k + 256] : 0.f;
#pragma unroll
            for (int q = 0; q < GRW; ++q) hm[q] = d.PHl[(b0 + q < n ? b0 + q : b0) * K + k + 256];
        }
        float sacc[GRW];
#pragma unroll
        for (int q = 0; q < GRW; ++q) sacc[q] = 0.f;
#pragma unroll
        for (int j = 0; j < 2 * MA; ++j)
            if (j < nj) {
#pragma unroll
                for (int q = 0; q < GRW; ++q) sacc[q] = fmaf(dhd[q][j], wc[j], sacc[q]);
            }
#pragma unroll
        for (int q = 0; q < GRW; ++q)
            if (b0 + q < n) d.dPZl[(long long)(b0 + q) * K + k] = (hc[q] > 0.f) ? sacc[q] : 0.f;
    }
}

__device__ void diag_block(const GDev &d, const StepArg &sa);
__device__ void td3_diag_block(const GDev &d, const StepArg &sa);

// the step's diagnostics (one workgroup): launched on the steps somebody reads them (SAC) / behind every pass (TD3, whose
// vector keeps the most recent value of each entry)
__global__ __launch_bounds__(256) void k_g_diag(GDev d, StepArg sa) {
    if (d.algo == 1) td3_diag_block(d, sa);
    else diag_block(d, sa);
}

// the diagnostics vector (SURVEY Appendix A line 17); one workgroup, sums in double
__device__ void diag_block(const GDev &d, const StepArg &sa) {
    constexpr int NQ = 28;
    // 0-3 q1 (sum, sum sq, max, min)  4-7 q2  8-11 y  12-15 log_pi  16 (q1-y)^2  17 (q2-y)^2  18 log_pi - q_new
    // 19 alpha log_pi - q_new  20-23 mu  24-27 log_std
    double acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = ((q & 3) == 2 && (q < 16 || q >= 20)) ? -INFINITY : (((q & 3) == 3 && (q < 16 || q >= 20)) ? INFINITY : 0.0);
    const int n = d.n, A = d.A;
    const float alpha = d.ctl->alpha;
    auto stat = [&](int q0, float v) {
        acc[q0] += (double)v; acc[q0 + 1] += (double)v * (double)v;
        acc[q0 + 2] = fmax(acc[q0 + 2], (double)v); acc[q0 + 3] = fmin(acc[q0 + 3], (double)v);
    };
    for (int i = threadIdx.x; i < n; i += 256) {
        const float q1 = d.QO[0][i], q2 = d.QO[1][i], y = d.y[i], lp = d.logpi[i], qn = d.qn[i];
        stat(0, q1); stat(4, q2); stat(8, y); stat(12, lp);
        const float e1 = q1 - y, e2 = q2 - y;
        acc[16] += (double)(e1 * e1); acc[17] += (double)(e2 * e2);
        acc[18] += (double)(lp - qn); acc[19] += (double)(alpha * lp - qn);
        for (int a = 0; a < A; ++a) { stat(20, d.mu[(long long)i * A + a]); stat(24, d.ls[(long long)i * A + a]); }
    }
    __shared__ double part[4][NQ];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const bool stq = q < 16 || q >= 20;
        const int kind = stq ? (q & 3) : 0;          // 0/1 sum, 2 max, 3 min
        double v = acc[q];
        for (int o = 32; o > 0; o >>= 1) {
            const double w = __shfl_xor(v, o);
            v = kind == 2 ? fmax(v, w) : (kind == 3 ? fmin(v, w) : v + w);
        }
        if (lane == 0) part[wave][q] = v;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double tot[NQ];
    for (int q = 0; q < NQ; ++q) {
        const bool stq = q < 16 || q >= 20;
        const int kind = stq ? (q & 3) : 0;
        double v = part[0][q];
        for (int w = 1; w < 4; ++w) v = kind == 2 ? fmax(v, part[w][q]) : (kind == 3 ? fmin(v, part[w][q]) : v + part[w][q]);
        tot[q] = v;
    }
    float out[SAC_DIAG_N];
    for (int i = 0; i < SAC_DIAG_N; ++i) out[i] = 0.f;
    out[SAC_D_QF1_LOSS] = (float)(tot[16] / n); out[SAC_D_QF2_LOSS] = (float)(tot[17] / n);
    out[SAC_D_POLICY_LOSS] = (float)(tot[18] / n); out[SAC_D_ACTOR_LOSS] = (float)(tot[19] / n);
    const int base[6] = {0, 4, 8, 12, 20, 24};
    for (int s = 0; s < 6; ++s) {
        const double cnt = s < 4 ? (double)n : (double)n * A;
        const double mean = tot[base[s]] / cnt;
        double var = tot[base[s] + 1] / cnt - mean * mean;
        if (var < 0) var = 0;
        out[SAC_D_Q1_MEAN + 4 * s + 0] = (float)mean; out[SAC_D_Q1_MEAN + 4 * s + 1] = (float)sqrt(var);
        out[SAC_D_Q1_MEAN + 4 * s + 2] = (float)tot[base[s] + 2]; out[SAC_D_Q1_MEAN + 4 * s + 3] = (float)tot[base[s] + 3];
    }
    out[SAC_D_ALPHA] = alpha; out[SAC_D_ALPHA_LOSS] = d.ctl->alpha_loss;
    float *const dlast = (sa.pad2 & 2u) ? d.diag_last : d.diag_dev;
    for (int i = 0; i < 30; ++i) {
        dlast[i] = out[i];
        if (sa.loop_pos == 0) d.diag_first[i] = out[i];
        if (sa.loop_pos < DIAG_TRACE_CAP) d.diag_trace[(size_t)sa.loop_pos * SAC_DIAG_N + i] = out[i];
    }
}

// ------------------------------------------------------------------------------------------
// TD3 (rlkit TD3Trainer, /root/reference/util/rlkit_utils.py:107-135) on the same launches: a critic pass every step, an
// actor pass on policy steps (and, without updates, on the steps whose statistics somebody reads)
// ------------------------------------------------------------------------------------------
// critic pass: the Q nets' input rows [(s, a) ; (s', a~)], a~ = tanh(target policy(s')) + clamp(N(0,1) sigma, +-clip)
// (the sum is NOT clipped to the action range)
template <int MA>
__global__ __launch_bounds__(256) void k_g_td3_head(GDev d, const float *__restrict__ S, SlotLayout SL, StepArg sa) {
    const int n = d.n, O = d.O, A = d.A, ldq = d.ldq;
    __shared__ float part[GRW][16][33];
    __shared__ float xs[GRW][GCK];
    __shared__ float ws[32][GCK + 1];
    {
        const int w = 2 * O + A, tot = n * w;
        for (int e = (int)(blockIdx.x * 256 + threadIdx.x); e < tot; e += (int)(gridDim.x * 256)) {
            const int b = e / w, k = e - b * w;
            if (k < O) d.XQ[(long long)b * ldq + k] = S[SL.off_obs + (long long)b * O + k];
            else if (k < 2 * O) d.XQ[(long long)(n + b) * ldq + (k - O)] = S[SL.off_nobs + (long long)b * O + (k - O)];
            else d.XQ[(long long)b * ldq + O + (k - 2 * O)] = S[SL.off_act + (long long)b * A + (k - 2 * O)];
        }
    }
    // the target policy's head layer (A outputs) on the GRW rows of this workgroup, then thread = (row, action)
    const int r0 = blockIdx.x * GRW, a = threadIdx.x & 15, rr = threadIdx.x >> 4, b = r0 + rr;
    small_layer_rows<MA>(d.PHTl, d.KPl, d.WhT, A, false, r0, n, part, xs, ws);
    if (rr < GRW && b < n && a < A) {
        float mean = part[rr][0][a];
#pragma unroll
        for (int q = 1; q < 16; ++q) mean += part[rr][q][a];                                  // fixed order
        mean += d.bhT[a];
        d.HDTw[(long long)b * A + a] = mean;
        const float eps = d.eps2 ? d.eps2[(long long)b * A + a]
                                 : philox_normal(d.noise_seed, (unsigned long long)sa.step_now, (unsigned)(b * d.NI + a), 1u);
        const float act = tanhf(mean) + fminf(fmaxf(eps * d.td3_sigma, -d.td3_clip), d.td3_clip);
        d.a2[(long long)b * A + a] = act;
        d.XQ[(long long)(n + b) * ldq + O + a] = act;
    }
}

// One workgroup per batch row b: the LAST layers of Q1, Q2 on (s, a) and of their targets on (s', a~) (one output each), then
// y = reward_scale r + (1 - d) discount min(T1, T2)(s', a~), dL/dq_i = 2 (q_i - y) / n, and the backward pass through
// those last layers (k_g_loss's structure).
__global__ __launch_bounds__(256) void k_g_td3_loss(GDev d, const float *__restrict__ S, SlotLayout SL) {
    const int b = blockIdx.x, n = d.n, K = d.KQl;
    __shared__ float red[4][4];
    __shared__ float s_dq[2];
    const float *h[4] = {d.QHl[0] + (long long)b * K, d.QHl[1] + (long long)b * K, d.THl[0] + (long long)b * K, d.THl[1] + (long long)b * K};
    const float *w[4] = {d.Wl[0], d.Wl[1], d.Wl[2], d.Wl[3]};
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float p_rew = 0.f, p_term = 0.f, p_b[4] = {0.f, 0.f, 0.f, 0.f};
    if (threadIdx.x == 0) {
        p_rew = S[SL.off_rew + b]; p_term = S[SL.off_term + b];
#pragma unroll
        for (int q = 0; q < 4; ++q) p_b[q] = d.bl[q][0];
    }
#pragma unroll 4
    for (int k = threadIdx.x; k < K; k += 256) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = fmaf(h[q][k], w[q][k], acc[q]);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[q] += __shfl_xor(acc[q], o);
        if (lane == 0) red[wave][q] = acc[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = ((red[0][q] + red[1][q]) + (red[2][q] + red[3][q])) + p_b[q];      // fixed order
        const float invB = 1.0f / (float)n;
        const float y = bellman_target(d.reward_scale, p_rew, p_term, d.discount, fminf(v[2], v[3]));
#pragma unroll
        for (int q = 0; q < 4; ++q) d.QOw[q][b] = v[q];
        d.y[b] = y;
        const float g0 = 2.0f * (v[0] - y) * invB, g1 = 2.0f * (v[1] - y) * invB;
        d.DQ[0][b] = g0; d.DQ[1][b] = g1;
        s_dq[0] = g0; s_dq[1] = g1;
    }
    __syncthreads();
    float *o[2] = {d.dQZl[0] + (long long)b * K, d.dQZl[1] + (long long)b * K};
    for (int k = threadIdx.x; k < K; k += 256) {
#pragma unroll
        for (int q = 0; q < 2; ++q) o[q][k] = (h[q][k] > 0.f) ? s_dq[q] * w[q][k] : 0.f;
    }
}

// actor pass: the online policy's head layer (A outputs) and Q1's input rows [obs | tanh(policy(s))]; the loss -mean Q1 has the
// gradient -1/n on every row
template <int MA>
__global__ __launch_bounds__(256) void k_g_td3_ahead(GDev d, const float *__restrict__ S, SlotLayout SL) {
    const int n = d.n, O = d.O, A = d.A, ldq = d.ldq;
    __shared__ float part[GRW][16][33];
    __shared__ float xs[GRW][GCK];
    __shared__ float ws[32][GCK + 1];
    for (int e = (int)(blockIdx.x * 256 + threadIdx.x); e < n * O; e += (int)(gridDim.x * 256)) {
        const int b = e / O, k = e - b * O;
        d.XA[(long long)b * ldq + k] = S[SL.off_obs + (long long)b * O + k];
    }
    const int r0 = blockIdx.x * GRW, a = threadIdx.x & 15, rr = threadIdx.x >> 4, b = r0 + rr;
    small_layer_rows<MA>(d.PHl, d.KPl, d.Wh, A, false, r0, n, part, xs, ws);
    if (rr < GRW && b < n && a < A) {
        float mean = part[rr][0][a];
#pragma unroll
        for (int q = 1; q < 16; ++q) mean += part[rr][q][a];                                  // fixed order
        mean += d.bh[a];
        d.HDPw[(long long)b * A + a] = mean;
        const float act = tanhf(mean);
        d.pa[(long long)b * A + a] = act;
        d.XA[(long long)b * ldq + O + a] = act;
        if (a == 0) d.DQA[b] = -1.0f / (float)n;
    }
}

// actor pass: Q1's last layer on (s, pi(s)) (one output per row: Q1(s, pi(s)), the actor loss's statistic) and the backward pass
// through it for the loss -mean Q1: dL/dh = -1/n w where h > 0.  One workgroup per batch row.
__global__ __launch_bounds__(256) void k_g_td3_qa(GDev d, int backward) {
    const int b = blockIdx.x, n = d.n, K = d.KQl;
    __shared__ float red[4];
    const float *h = d.AHl + (long long)b * K, *w = d.Wl[0];
    float acc = 0.f;
#pragma unroll 4
    for (int k = threadIdx.x; k < K; k += 256) acc = fmaf(h[k], w[k], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) d.QAw[b] = ((red[0] + red[1]) + (red[2] + red[3])) + d.bl[0][0];      // fixed order
    if (!backward) return;
    const float g = -1.0f / (float)n;
    float *o = d.dAZl + (long long)b * K;
    for (int k = threadIdx.x; k < K; k += 256) o[k] = (h[k] > 0.f) ? g * w[k] : 0.f;
}

// actor pass, one workgroup per GRW batch rows: dL/da through the action columns of Q1's first layer, dL/d(pre-tanh) = dL/da (1 - a^2),
// and the backward pass through the policy's head layer (k_g_polgrad's structure with one critic and A head rows)
template <int MA>
__global__ __launch_bounds__(256) void k_g_td3_polgrad(GDev d) {
    const int b0 = blockIdx.x * GRW, n = d.n, A = d.A, H0 = d.HQ0, ldq = d.ldq, t = threadIdx.x;
    __shared__ float part[GRW][16][33];
    __shared__ float dhd[GRW][16];
    __shared__ float gs[GRW][GCK];
    __shared__ float was[GCK][17];
    const int a = t & 15, hg = t >> 4;
    const int K = d.KPl;
    float wv[MA], hm[GRW];
    {
        const int k = t < K ? t : 0;
#pragma unroll
        for (int j = 0; j < MA; ++j) wv[j] = (j < A) ? d.Wh[j * K + k] : 0.f;
#pragma unroll
        for (int q = 0; q < GRW; ++q) hm[q] = d.PHl[(b0 + q < n ? b0 + q : b0) * K + k];
    }
    const int rr = t >> 4;
    const bool head_thread = rr < GRW && b0 + rr < n && a < A;
    float p_act = 0.f;
    if (head_thread) p_act = d.pa[(long long)(b0 + rr) * A + a];
    {
        float s0[GRW], rg[GRW], rwa[MA];
        const int q256 = 256 / A, r256 = 256 - q256 * A, hl_t = t / A, j_t = t - hl_t * A;
        const float *w0a = d.W1q[0] + d.O;
        int goff[GRW];
#pragma unroll
        for (int q = 0; q < GRW; ++q) { s0[q] = 0.f; goff[q] = (b0 + q < n ? b0 + q : b0) * H0; }
        auto fetch = [&](int h0) {
            const int hh = h0 + t;
            const bool in = hh < H0;
            const int hc = in ? hh : H0 - 1;
#pragma unroll
            for (int q = 0; q < GRW; ++q) { const float v = d.dAZ0[goff[q] + hc]; rg[q] = in ? v : 0.f; }
            int hl = hl_t, j = j_t;                         // element e = t + 256 m of the chunk's GCK * A: (row e / A, column e % A)
#pragma unroll
            for (int m = 0; m < MA; ++m) {
                if (m < A) rwa[m] = w0a[(h0 + hl < H0 ? h0 + hl : H0 - 1) * ldq + j];
                j += r256; hl += q256;
                if (j >= A) { j -= A; hl += 1; }
            }
        };
        fetch(0);
        for (int h0 = 0; h0 < H0; h0 += GCK) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < GRW; ++q) gs[q][t] = rg[q];
            {
                int hl = hl_t, j = j_t;
#pragma unroll
                for (int m = 0; m < MA; ++m) {
                    if (m < A) was[hl][j] = rwa[m];
                    j += r256; hl += q256;
                    if (j >= A) { j -= A; hl += 1; }
                }
            }
            __syncthreads();
            if (h0 + GCK < H0) fetch(h0 + GCK);
#pragma unroll
            for (int i = 0; i < GCK / 16; ++i) {
                const int hl = hg + 16 * i;
                const float w0v = was[hl][a];
#pragma unroll
                for (int q = 0; q < GRW; ++q) s0[q] = fmaf(gs[q][hl], w0v, s0[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < GRW; ++q) part[q][hg][a] = s0[q];
    }
    if (t < GRW * 16) dhd[t >> 4][t & 15] = 0.f;
    __syncthreads();
    if (head_thread) {
        float da = part[rr][0][a];
#pragma unroll
        for (int q = 1; q < 16; ++q) da += part[rr][q][a];                                    // fixed order
        const long long e = (long long)(b0 + rr) * A + a;
        d.DAaw[e] = da;
        const float g = da * (1.0f - p_act * p_act);
        d.DHP[e] = g;
        dhd[rr][a] = g;
    }
    __syncthreads();
    for (int k = t; k < K; k += 256) {
        float wc[MA], hc[GRW];
#pragma unroll
        for (int j = 0; j < MA; ++j) wc[j] = wv[j];
#pragma unroll
        for (int q = 0; q < GRW; ++q) hc[q] = hm[q];
        if (k + 256 < K) {
#pragma unroll
            for (int j = 0; j < MA; ++j) wv[j] = (j < A) ? d.Wh[j * K + k + 256] : 0.f;
#pragma unroll
            for (int q = 0; q < GRW; ++q) hm[q] = d.PHl[(b0 + q < n ? b0 + q : b0) * K + k + 256];
        }
        float sacc[GRW];
#pragma unroll
        for (int q = 0; q < GRW; ++q) sacc[q] = 0.f;
#pragma unroll
        for (int j = 0; j < MA; ++j)
            if (j < A) {
#pragma unroll
                for (int q = 0; q < GRW; ++q) sacc[q] = fmaf(dhd[q][j], wc[j], sacc[q]);
            }
#pragma unroll
        for (int q = 0; q < GRW; ++q)
            if (b0 + q < n) d.dPZl[(long long)(b0 + q) * K + k] = (hc[q] > 0.f) ? sacc[q] : 0.f;
    }
}

// TD3's statistics in the slots of the SAC vector (sac_hip.h): sa.pad bit 0 = the critic part (every step), bit 1 = the policy
// part (policy / statistics steps).  The device copy keeps the most recent value of every entry; a launch whose caller
// reads the diagnostics copies the whole vector out at its end.
__device__ void td3_diag_block(const GDev &d, const StepArg &sa) {
    constexpr int NS = 6;        // q1, q2, y, be1, be2, policy action
    const int n = d.n, A = d.A, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, loop_pos = sa.loop_pos;
    double sm[NS], sq[NS], lsum = 0;
    float mx[NS], mn[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) { sm[q] = 0; sq[q] = 0; mx[q] = -INFINITY; mn[q] = INFINITY; }
    auto acc1 = [&](int q, float v) { sm[q] += v; sq[q] += (double)v * v; mx[q] = fmaxf(mx[q], v); mn[q] = fminf(mn[q], v); };
    if (sa.pad & 1)
        for (int i = threadIdx.x; i < n; i += 256) {
            const float yv = d.y[i], q1 = d.QO[0][i], q2 = d.QO[1][i];
            acc1(0, q1); acc1(1, q2); acc1(2, yv); acc1(3, (q1 - yv) * (q1 - yv)); acc1(4, (q2 - yv) * (q2 - yv));
        }
    if (sa.pad & 2) {
        for (int i = threadIdx.x; i < n; i += 256) lsum += (double)d.QA[i];
        for (int e = threadIdx.x; e < n * A; e += 256) acc1(5, d.pa[e]);
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
    __shared__ double sh[4][32];
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NS; ++q) { sh[wave][q] = sm[q]; sh[wave][6 + q] = sq[q]; sh[wave][12 + q] = mx[q]; sh[wave][18 + q] = mn[q]; }
        sh[wave][24] = lsum;
    }
    __syncthreads();
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
            for (int w = 0; w < 4; ++w) { s += sh[w][q]; s2 += sh[w][6 + q]; MX = fmax(MX, sh[w][12 + q]); MN = fmin(MN, sh[w][18 + q]); }
            const double cnt = (q < 5) ? (double)n : (double)n * A;
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
        for (int w = 0; w < 4; ++w) s += sh[w][24];
        put(SAC_D_POLICY_LOSS, (float)(-s / n));
    }
    if (sa.pad2 & 2u) {
        __syncthreads();
        if (threadIdx.x < SAC_DIAG_N) d.diag_last[threadIdx.x] = ld_sc1(d.diag_dev + threadIdx.x);
    }
}

}  // namespace gen
