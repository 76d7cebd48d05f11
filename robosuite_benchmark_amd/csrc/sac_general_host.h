// Host side of the general-shape SAC step (sac_general.h): network tables, the job lists of the matrix-product
// launches, parameter transfer, the step's launch sequence.  Included by sac_trainer.hip behind struct sac_trainer.
#pragma once

struct GenLayer { int N = 0, K = 0; long long offW = 0, offB = 0; };
struct GenNet {
    int nl = 0;
    GenLayer L[gen::GMAXL];
    long long n = 0, nflat = 0;             // floats of the device vector (every W starts on a 16-byte boundary) / of the flat vector
    float *P = nullptr, *M = nullptr, *V = nullptr, *G = nullptr;
    std::vector<long long> perm;            // flat (sac_get_params) index -> index in the device vector
};
constexpr int GEN_MAX_JOBS = 16 * gen::GMAXJ;
struct GenStage { int kind = 0, mode = 0; size_t base = 0; gen::GemmStage gs{}; };   // mode: 0 forward, 1 backward, 2 weight gradients   // kind 0: a k_g_gemm launch; else see launch
enum { GS_GEMM = 0, GS_HEAD, GS_LOSS, GS_POLGRAD, GS_DIAG, GS_TD3_HEAD, GS_TD3_LOSS, GS_TD3_AHEAD, GS_TD3_POLGRAD, GS_TD3_QA };

struct sac_general {
    int n = 0, O = 0, A = 0, Lp = 0, Lq = 0;
    int hp[gen::GMAXL] = {}, hq[gen::GMAXL] = {};
    GenNet net[6];                                    // 5: TD3's target policy
    gen::GDev dev{};
    // TD3: the critic pass (every step), the actor pass (policy steps), the actor pass's forward half (statistics steps)
    std::vector<GenStage> td3_critic, td3_actor, td3_stats;
    char *arena = nullptr;
    gen::GemmJob *d_jobs = nullptr;
    float *d_scratch = nullptr; unsigned *d_tile_cnt = nullptr;      // split reductions (sac_general.h: GemmStage::splitk)
    std::vector<GenStage> stages;
};

namespace {

struct GenBump {
    std::vector<std::pair<void **, size_t>> req;
    size_t total = 0;
    template <typename T> void want(T **p, long long count) {
        req.emplace_back(reinterpret_cast<void **>(p), total);
        total += ((size_t)count * sizeof(T) + 255) & ~(size_t)255;
    }
};

void gen_shape_net(GenNet &N, const int *hidden, int nh, int in_dim, int out_dim, bool merged_heads, int A) {
    long long off = 0, flat = 0;
    int d = in_dim;
    N.nl = nh + 1;
    for (int l = 0; l <= nh; ++l) {
        GenLayer &L = N.L[l];
        L.N = (l < nh) ? hidden[l] : out_dim; L.K = d;
        off = (off + 3) & ~3LL;                              // (16-byte loads of the matrix-product kernel)
        L.offW = off; off += (long long)L.N * L.K;
        L.offB = off; off += L.N;
        flat += (long long)L.N * L.K + L.N;
        d = L.N;
    }
    N.n = (off + 3) & ~3LL; N.nflat = flat;
    // flat layout: per layer W then b, packed; the policy's two heads are two layers there (last_fc, last_fc_log_std) and
    // ONE layer of 2A outputs here ([W_mean ; W_log_std] then [b_mean ; b_log_std])
    N.perm.resize((size_t)flat);
    long long fi = 0;
    for (int l = 0; l <= nh; ++l) {
        const GenLayer &L = N.L[l];
        if (l == nh && merged_heads) {
            const long long AK = (long long)A * L.K;
            for (long long i = 0; i < AK; ++i) N.perm[(size_t)fi++] = L.offW + i;               // last_fc.weight
            for (int a = 0; a < A; ++a) N.perm[(size_t)fi++] = L.offB + a;                      // last_fc.bias
            for (long long i = 0; i < AK; ++i) N.perm[(size_t)fi++] = L.offW + AK + i;          // last_fc_log_std.weight
            for (int a = 0; a < A; ++a) N.perm[(size_t)fi++] = L.offB + A + a;                  // last_fc_log_std.bias
        } else {
            for (long long i = 0; i < (long long)L.N * L.K; ++i) N.perm[(size_t)fi++] = L.offW + i;
            for (int i = 0; i < L.N; ++i) N.perm[(size_t)fi++] = L.offB + i;
        }
    }
}

gen::GemmJob gen_job_zero() { gen::GemmJob J; memset(&J, 0, sizeof(J)); return J; }

// Y = act(X W^T + b): X [M][K], W [N][K]
gen::GemmJob gen_fwd(const float *X, int M, const float *W, const float *b, int N, int K, float *Y, int relu) {
    gen::GemmJob J = gen_job_zero();
    J.A = X; J.sa_m = K; J.sa_r = 1;
    J.Bm = W; J.sb_n = K; J.sb_r = 1;
    J.C = Y; J.ldc = N; J.bias = b; J.relu = relu;
    J.M = M; J.N = N; J.R = K;
    return J;
}
// dX[:, c0 : c0 + ncols] = dY W[:, c0 : c0 + ncols], zero where mask <= 0: dY [M][Nl], W [Nl][K]
gen::GemmJob gen_bwd(const float *dY, int M, int Nl, const float *W, int K, int c0, int ncols, float *dX, int lddx,
                     const float *mask, int ldmask) {
    gen::GemmJob J = gen_job_zero();
    J.A = dY; J.sa_m = Nl; J.sa_r = 1;
    J.Bm = W + c0; J.sb_n = 1; J.sb_r = K;
    J.C = dX; J.ldc = lddx; J.mask = mask; J.ldmask = ldmask;
    J.M = M; J.N = ncols; J.R = Nl;
    return J;
}
// dW = dY^T X over `rows` rows, db = column sums of dY (dY [rows][Nl], X [rows][K]) for layer L of net N, whose owner tiles
// apply Adam at learning rate lr and the Polyak average into `target` (null: none) in their epilogue
gen::GemmJob gen_dw(const float *dY, const float *X, int rows, const GenNet &N, const GenLayer &L, float lr, const GenNet *target) {
    gen::GemmJob J = gen_job_zero();
    J.A = dY; J.sa_m = 1; J.sa_r = L.N;
    J.Bm = X; J.sb_n = 1; J.sb_r = L.K;
    J.C = N.G + L.offW; J.ldc = L.K; J.c_ones = N.G + L.offB; J.ones_col = 1;
    J.M = L.N; J.N = L.K; J.R = rows;
    J.aP = N.P + L.offW; J.aM = N.M + L.offW; J.aV = N.V + L.offW; J.aTP = target ? target->P + L.offW : nullptr;
    J.bP = N.P + L.offB; J.bM = N.M + L.offB; J.bV = N.V + L.offB; J.bTP = target ? target->P + L.offB : nullptr;
    J.lr = lr;
    return J;
}

// builds the launch lists of a trainer: matrix-product launches as tables of jobs + the small kernels in between
struct GenPlanner {
    std::vector<gen::GemmJob> jobs;
    std::vector<GenStage> *list = nullptr;
    GenStage cur;
    long long scratch_floats = 0;
    int max_tiles = 0;
    bool overflow = false;                  // a launch with more jobs than its table holds (checked by gen_commit_plan)
    void begin(int mode) {
        cur = GenStage{};
        cur.kind = GS_GEMM; cur.mode = mode; cur.base = jobs.size();
        for (int q = 0; q < gen::GMAXJ; ++q) cur.gs.tile0[q] = 1 << 30;
    }
    // may an operand be fetched in 16-byte pieces?  rc (contiguous along the reduction): base, row stride and R multiples of
    // four floats; else (contiguous along its rows): base, reduction stride and the row count
    static int vec_ok(const float *p, long long slot_off, bool in_slot, long long s_row, long long s_red, int rows, int R, bool rc) {
        const bool aligned = in_slot ? (slot_off % 4 == 0) : ((reinterpret_cast<uintptr_t>(p) & 15) == 0);
        if (getenv("SAC_GEN_NO_VEC")) return 0;              // (A/B comparisons)
        return (int)(rc ? (aligned && s_red == 1 && s_row % 4 == 0 && R % 4 == 0 && R >= 4)
                        : (aligned && s_row == 1 && s_red % 4 == 0 && rows % 4 == 0 && rows >= 4));
    }
    void add(gen::GemmJob J) {
        if (cur.gs.njobs >= gen::GMAXJ) { overflow = true; return; }
        J.a_vec = vec_ok(J.A, J.a_off, J.a_slot != 0, J.sa_m, J.sa_r, J.M, J.R, cur.mode != 2);
        J.b_vec = vec_ok(J.Bm, J.b_off, J.b_slot != 0, J.sb_n, J.sb_r, J.N, J.R, cur.mode == 0);
        J.tiles_n = (J.N + (J.ones_col ? 1 : 0) + gen::GT - 1) / gen::GT;
        J.tile0 = cur.gs.ntiles;
        cur.gs.tile0[cur.gs.njobs++] = J.tile0;
        cur.gs.ntiles += ((J.M + gen::GT - 1) / gen::GT) * J.tiles_n;
        jobs.push_back(J);
    }
    void end() {
        // few tiles and a long reduction: split the reduction over up to four workgroups per tile (never the weight-gradient
        // launch: many tiles, four chunks)
        int min_chunks = 1 << 30;
        for (size_t i = cur.base; i < jobs.size(); ++i) min_chunks = std::min(min_chunks, (jobs[i].R + gen::GK - 1) / gen::GK);
        int sk = 1;
        if (cur.mode != 2 && !getenv("SAC_GEN_NO_SPLITK"))
            while (sk < 4 && 2 * sk <= min_chunks && cur.gs.ntiles * 2 * sk <= 512) sk *= 2;      // (up to two workgroups per CU)
        cur.gs.splitk = sk;
        if (sk > 1) scratch_floats = std::max(scratch_floats, (long long)cur.gs.ntiles * sk * gen::GT * gen::GT);
        max_tiles = std::max(max_tiles, cur.gs.ntiles);
        list->push_back(cur);
    }
    void one(int mode, const gen::GemmJob &J) { begin(mode); add(J); end(); }
    void plain(int kind) { GenStage s; s.kind = kind; list->push_back(s); }
};

// the planner's jobs to the device, the split reductions' scratch, the lists' device pointers
int gen_commit_plan(sac_trainer *t, GenPlanner &pl, std::initializer_list<std::vector<GenStage> *> lists) {
    sac_general *g = t->gen;
    SAC_REQUIRE(pl.jobs.size() <= (size_t)GEN_MAX_JOBS && !pl.overflow, "internal: %zu matrix-product jobs%s", pl.jobs.size(),
                pl.overflow ? " (a launch's job table overflowed)" : "");
    SAC_HIP(hipMemcpyAsync(g->d_jobs, pl.jobs.data(), sizeof(gen::GemmJob) * pl.jobs.size(), hipMemcpyHostToDevice, t->stream));
    if (pl.scratch_floats) SAC_HIP(hipMalloc(reinterpret_cast<void **>(&g->d_scratch), sizeof(float) * (size_t)pl.scratch_floats));
    SAC_HIP(hipMalloc(reinterpret_cast<void **>(&g->d_tile_cnt), sizeof(unsigned) * (size_t)(pl.max_tiles + 1)));
    SAC_HIP(hipMemsetAsync(g->d_tile_cnt, 0, sizeof(unsigned) * (size_t)(pl.max_tiles + 1), t->stream));
    SAC_HIP(hipStreamSynchronize(t->stream));
    for (auto *L : lists)
        for (auto &s : *L)
            if (s.kind == GS_GEMM) { s.gs.jobs = g->d_jobs + s.base; s.gs.scratch = g->d_scratch; s.gs.tile_cnt = g->d_tile_cnt; }
    return 0;
}

int gen_build(sac_trainer *t, const int *hp, int np_, const int *hq, int nq_) {
    sac_general *g = new sac_general();
    t->gen = g;
    const int n = t->Bt, O = t->O, A = t->A;
    g->n = n; g->O = O; g->A = A; g->Lp = np_; g->Lq = nq_;
    for (int i = 0; i < np_; ++i) g->hp[i] = hp[i];
    for (int i = 0; i < nq_; ++i) g->hq[i] = hq[i];
    gen_shape_net(g->net[0], hp, np_, O, 2 * A, true, A);
    for (int i = 1; i < 5; ++i) gen_shape_net(g->net[i], hq, nq_, O + A, 1, false, A);
    const int Lp = np_, Lq = nq_, ldq = O + A;
    {   // the matrix-product kernel indexes its operands with 32-bit offsets
        long long widest = ldq;
        for (int i = 0; i < np_; ++i) widest = hp[i] > widest ? hp[i] : widest;
        for (int i = 0; i < nq_; ++i) widest = hq[i] > widest ? hq[i] : widest;
        SAC_REQUIRE(3LL * n * widest < (1LL << 31) && widest * widest < (1LL << 31),
                    "batch %d x layer width %lld is beyond the general step's 32-bit operand offsets", n, widest);
    }

    GenBump bump;
    for (int i = 0; i < 5; ++i) {
        bump.want(&g->net[i].P, g->net[i].n);
        if (i < 3) { bump.want(&g->net[i].M, g->net[i].n); bump.want(&g->net[i].V, g->net[i].n); bump.want(&g->net[i].G, g->net[i].n); }
    }
    gen::GDev &d = g->dev;
    float *PH[gen::GMAXL] = {}, *dPZ[gen::GMAXL] = {}, *HD = nullptr;
    float *QH[2][gen::GMAXL] = {}, *TH[2][gen::GMAXL] = {}, *dQZ[2][gen::GMAXL] = {}, *QO[4] = {}, *DA[2] = {};
    bump.want(&d.XQ, 3LL * n * ldq); bump.want(&d.done, 64);
    for (int l = 0; l < Lp; ++l) { bump.want(&PH[l], 2LL * n * hp[l]); bump.want(&dPZ[l], (long long)n * hp[l]); }
    bump.want(&HD, 2LL * n * 2 * A);
    float **rowsA[] = {&d.mu, &d.ls, &d.ok, &d.epsv, &d.anew, &d.a2};
    for (float **p : rowsA) bump.want(p, (long long)n * A);
    bump.want(&d.logpi, n); bump.want(&d.logpi2, n); bump.want(&d.y, n); bump.want(&d.qn, n);
    for (int k = 0; k < 2; ++k) {
        for (int l = 0; l < Lq; ++l) {
            bump.want(&QH[k][l], 2LL * n * hq[l]); bump.want(&TH[k][l], (long long)n * hq[l]); bump.want(&dQZ[k][l], 2LL * n * hq[l]);
        }
        bump.want(&QO[k], 2LL * n); bump.want(&QO[2 + k], n); bump.want(&d.DQ[k], 2LL * n); bump.want(&DA[k], (long long)n * A);
    }
    bump.want(&d.DHD, (long long)n * 2 * A);
    // shared with the entry points of sac_trainer.hip: host-batch slot, caller-supplied noise, diagnostics, entropy state
    t->ext_layout = make_slot_layout(t->Bt, O, A);
    bump.want(&t->ext_slot, t->ext_layout.slot_floats);
    bump.want(&t->d_eps, 2LL * t->B * A);
    bump.want(&t->d_diag, (long long)SAC_DIAG_N * (2 + DIAG_TRACE_CAP));
    bump.want(&t->d_ctl, 1);
    bump.want(&g->d_jobs, GEN_MAX_JOBS);
    {
        const size_t bytes = (bump.total + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        SAC_HIP(hipMalloc(reinterpret_cast<void **>(&g->arena), bytes));
        SAC_HIP(hipMemsetAsync(g->arena, 0, bytes, t->stream));
        for (auto &r : bump.req) *r.first = g->arena + r.second;
    }
    const sac_config_t &cfg = t->cfg;
    d.n = n; d.O = O; d.A = A; d.NI = 16; d.ldq = ldq;
    d.discount = cfg.discount; d.reward_scale = cfg.reward_scale; d.tau = cfg.soft_target_tau;
    d.target_entropy = std::isnan(cfg.target_entropy) ? -(float)A : cfg.target_entropy;
    d.alpha_lr = cfg.policy_lr; d.period = cfg.target_update_period; d.auto_alpha = cfg.use_automatic_entropy_tuning;
    d.noise_seed = cfg.noise_seed; d.ctl = t->d_ctl; d.HD = HD;
    for (int k = 0; k < 4; ++k) d.QO[k] = QO[k];
    for (int k = 0; k < 2; ++k) d.DA[k] = DA[k];
    d.diag_first = t->d_diag_host; d.diag_last = t->d_diag_host + SAC_DIAG_N; d.diag_trace = t->d_diag + 2 * SAC_DIAG_N;
    d.diag_dev = t->d_diag;
    d.eps1 = d.eps2 = nullptr;
    const float lrs[3] = {cfg.policy_lr, cfg.qf_lr, cfg.qf_lr};

    // ---- the launch sequence ----
    GenPlanner pl;
    pl.list = &g->stages;
    auto begin = [&](int mode) { pl.begin(mode); };
    auto add = [&](const gen::GemmJob &J) { pl.add(J); };
    auto end = [&]() { pl.end(); };
    auto plain = [&](int kind) { pl.plain(kind); };
    auto Wp = [&](int net, int l) { return g->net[net].P + g->net[net].L[l].offW; };
    auto Bp = [&](int net, int l) { return g->net[net].P + g->net[net].L[l].offB; };

    const SlotLayout &XL = t->ext_layout;               // (the buffer's slots have the same layout: make_slot_layout(batch, O, A))
    for (int l = 0; l < Lp; ++l) {                       // policy trunk on [s ; s']: the first layer reads the slot's rows
        const GenLayer &L = g->net[0].L[l];
        begin(0);
        if (l == 0) {
            gen::GemmJob Js = gen_fwd(nullptr, n, Wp(0, 0), Bp(0, 0), L.N, L.K, PH[0], 1), Jn = Js;
            Js.a_slot = Jn.a_slot = 1; Js.a_off = XL.off_obs; Jn.a_off = XL.off_nobs; Jn.C = PH[0] + (long long)n * L.N;
            add(Js); add(Jn);
        } else {
            add(gen_fwd(PH[l - 1], 2 * n, Wp(0, l), Bp(0, l), L.N, L.K, PH[l], 1));
        }
        end();
    }
    // (the head layer -- 2A outputs -- is computed inside the head kernel, the Q nets' last layers -- one output -- inside the
    //  loss kernel together with their backward pass, the action columns of the Q nets' first layers and the head layer's
    //  backward pass inside the policy-gradient kernel: five matrix-product launches fewer per step than one per layer and
    //  direction, each of which had a handful of output columns or a reduction of a handful of terms)
    d.PHl = PH[Lp - 1]; d.KPl = hp[Lp - 1]; d.Wh = Wp(0, Lp); d.bh = Bp(0, Lp); d.HDw = HD;
    for (int k = 0; k < 2; ++k) {
        d.QHl[k] = QH[k][Lq - 1]; d.THl[k] = TH[k][Lq - 1]; d.dQZl[k] = dQZ[k][Lq - 1];
        d.dQZ0[k] = dQZ[k][0]; d.W1q[k] = Wp(1 + k, 0); d.DAw[k] = DA[k];
    }
    d.KQl = hq[Lq - 1]; d.HQ0 = hq[0]; d.dPZl = dPZ[Lp - 1];
    for (int k = 0; k < 4; ++k) { d.Wl[k] = Wp(1 + k, Lq); d.bl[k] = Bp(1 + k, Lq); d.QOw[k] = QO[k]; }
    plain(GS_HEAD);
    for (int l = 0; l < Lq; ++l) {                       // Q1, Q2 on [(s,a) ; (s,a_new)], targets on (s',a'): the hidden layers
        begin(0);
        for (int k = 0; k < 2; ++k) {
            const GenLayer &L = g->net[1 + k].L[l];
            float *out = l < Lq ? QH[k][l] : QO[k];
            add(gen_fwd(l == 0 ? d.XQ : QH[k][l - 1], 2 * n, Wp(1 + k, l), Bp(1 + k, l), L.N, L.K, out, l < Lq));
        }
        for (int k = 0; k < 2; ++k) {
            const GenLayer &L = g->net[3 + k].L[l];
            float *out = l < Lq ? TH[k][l] : QO[2 + k];
            add(gen_fwd(l == 0 ? d.XQ + 2LL * n * ldq : TH[k][l - 1], n, Wp(3 + k, l), Bp(3 + k, l), L.N, L.K, out, l < Lq));
        }
        end();
    }
    plain(GS_LOSS);
    for (int j = Lq - 1; j >= 1; --j) {                  // backward through hidden layer j of Q1, Q2 (critic and actor rows)
        begin(1);
        for (int k = 0; k < 2; ++k) {
            const GenLayer &L = g->net[1 + k].L[j];
            add(gen_bwd(dQZ[k][j], 2 * n, L.N, Wp(1 + k, j), L.K, 0, L.K, dQZ[k][j - 1], L.K, QH[k][j - 1], L.K));
        }
        end();
    }
    plain(GS_POLGRAD);
    for (int j = Lp - 1; j >= 1; --j) {                  // backward through the policy's hidden layers (rows of s only)
        const GenLayer &L = g->net[0].L[j];
        begin(1); add(gen_bwd(dPZ[j], n, L.N, Wp(0, j), L.K, 0, L.K, dPZ[j - 1], L.K, PH[j - 1], L.K)); end();
    }
    begin(2);                                             // every weight gradient of the step
    for (int l = Lp; l >= 0; --l) {
        const GenLayer &L = g->net[0].L[l];
        gen::GemmJob Jw = gen_dw(l == Lp ? d.DHD : dPZ[l], l == 0 ? nullptr : PH[l - 1], n, g->net[0], L, lrs[0], nullptr);
        if (l == 0) { Jw.b_slot = 1; Jw.b_off = XL.off_obs; }
        add(Jw);
    }
    for (int k = 0; k < 2; ++k)
        for (int l = Lq; l >= 0; --l) {
            const GenLayer &L = g->net[1 + k].L[l];
            add(gen_dw(l == Lq ? d.DQ[k] : dQZ[k][l], l == 0 ? d.XQ : QH[k][l - 1], n, g->net[1 + k], L, lrs[1 + k], &g->net[3 + k]));
        }
    end();
    plain(GS_DIAG);
    return gen_commit_plan(t, pl, {&g->stages});
}

// TD3 (td3_trainer_create_mlp): nets 0 policy (one head of A outputs, tanh), 1, 2 critics, 3, 4 their targets, 5 target policy
int gen_build_td3(sac_trainer *t, const td3_config_t *c, const int *hp, int np_, const int *hq, int nq_) {
    sac_general *g = new sac_general();
    t->gen = g;
    const int n = t->Bt, O = t->O, A = t->A;
    g->n = n; g->O = O; g->A = A; g->Lp = np_; g->Lq = nq_;
    for (int i = 0; i < np_; ++i) g->hp[i] = hp[i];
    for (int i = 0; i < nq_; ++i) g->hq[i] = hq[i];
    gen_shape_net(g->net[0], hp, np_, O, A, false, A);
    gen_shape_net(g->net[5], hp, np_, O, A, false, A);
    for (int i = 1; i < 5; ++i) gen_shape_net(g->net[i], hq, nq_, O + A, 1, false, A);
    const int Lp = np_, Lq = nq_, ldq = O + A;
    {
        long long widest = ldq;
        for (int i = 0; i < np_; ++i) widest = hp[i] > widest ? hp[i] : widest;
        for (int i = 0; i < nq_; ++i) widest = hq[i] > widest ? hq[i] : widest;
        SAC_REQUIRE(3LL * n * widest < (1LL << 31) && widest * widest < (1LL << 31),
                    "batch %d x layer width %lld is beyond the general step's 32-bit operand offsets", n, widest);
    }
    GenBump bump;
    for (int i = 0; i < 6; ++i) {
        bump.want(&g->net[i].P, g->net[i].n);
        if (i < 3) { bump.want(&g->net[i].M, g->net[i].n); bump.want(&g->net[i].V, g->net[i].n); bump.want(&g->net[i].G, g->net[i].n); }
    }
    gen::GDev &d = g->dev;
    float *PHT[gen::GMAXL] = {}, *PHP[gen::GMAXL] = {}, *dPZ[gen::GMAXL] = {}, *HDT = nullptr, *HDP = nullptr;
    float *QH[2][gen::GMAXL] = {}, *TH[2][gen::GMAXL] = {}, *dQZ[2][gen::GMAXL] = {}, *QO[4] = {};
    float *AH[gen::GMAXL] = {}, *dAZ[gen::GMAXL] = {}, *QA = nullptr, *DAa = nullptr;
    bump.want(&d.XQ, 2LL * n * ldq); bump.want(&d.XA, (long long)n * ldq); bump.want(&d.done, 64);
    for (int l = 0; l < Lp; ++l) { bump.want(&PHT[l], (long long)n * hp[l]); bump.want(&PHP[l], (long long)n * hp[l]); bump.want(&dPZ[l], (long long)n * hp[l]); }
    bump.want(&HDT, (long long)n * A); bump.want(&HDP, (long long)n * A);
    bump.want(&d.a2, (long long)n * A); bump.want(&d.pa, (long long)n * A); bump.want(&d.DHP, (long long)n * A); bump.want(&DAa, (long long)n * A);
    bump.want(&d.y, n); bump.want(&QA, n); bump.want(&d.DQA, n);
    for (int k = 0; k < 2; ++k) {
        for (int l = 0; l < Lq; ++l) { bump.want(&QH[k][l], (long long)n * hq[l]); bump.want(&TH[k][l], (long long)n * hq[l]); bump.want(&dQZ[k][l], (long long)n * hq[l]); }
        bump.want(&QO[k], n); bump.want(&QO[2 + k], n); bump.want(&d.DQ[k], n);
    }
    for (int l = 0; l < Lq; ++l) { bump.want(&AH[l], (long long)n * hq[l]); bump.want(&dAZ[l], (long long)n * hq[l]); }
    t->ext_layout = make_slot_layout(t->Bt, O, A);
    bump.want(&t->ext_slot, t->ext_layout.slot_floats);
    bump.want(&t->d_eps, 2LL * t->B * A);
    bump.want(&t->d_diag, (long long)SAC_DIAG_N * (2 + DIAG_TRACE_CAP));
    bump.want(&t->d_ctl, 1);
    bump.want(&g->d_jobs, GEN_MAX_JOBS);
    {
        const size_t bytes = (bump.total + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        SAC_HIP(hipMalloc(reinterpret_cast<void **>(&g->arena), bytes));
        SAC_HIP(hipMemsetAsync(g->arena, 0, bytes, t->stream));
        for (auto &r : bump.req) *r.first = g->arena + r.second;
    }
    d.n = n; d.O = O; d.A = A; d.NI = 16; d.ldq = ldq; d.algo = 1;
    d.discount = c->discount; d.reward_scale = c->reward_scale; d.tau = c->tau; d.period = 1; d.auto_alpha = 0;
    d.noise_seed = c->noise_seed; d.ctl = t->d_ctl;
    d.td3_sigma = c->target_policy_noise; d.td3_clip = c->target_policy_noise_clip;
    d.HDT = HDT; d.HDP = HDP; d.QA = QA; d.DAa = DAa;
    for (int k = 0; k < 4; ++k) d.QO[k] = QO[k];
    d.diag_first = t->d_diag_host; d.diag_last = t->d_diag_host + SAC_DIAG_N; d.diag_trace = t->d_diag + 2 * SAC_DIAG_N;
    d.diag_dev = t->d_diag;
    d.eps1 = d.eps2 = nullptr;
    GenPlanner pl;
    const SlotLayout &XL = t->ext_layout;
    auto Wp = [&](int net, int l) { return g->net[net].P + g->net[net].L[l].offW; };
    auto Bp = [&](int net, int l) { return g->net[net].P + g->net[net].L[l].offB; };
    // a policy (net) forward on the slot's rows at `off`: the trunk into PHx (the head layer -- A outputs -- rides in the kernel
    // behind it, like every layer of a handful of outputs: see gen_build)
    auto policy_fwd = [&](int net, long long off, float **PHx) {
        for (int l = 0; l < Lp; ++l) {
            const GenLayer &L = g->net[net].L[l];
            gen::GemmJob J = gen_fwd(l == 0 ? nullptr : PHx[l - 1], n, Wp(net, l), Bp(net, l), L.N, L.K, PHx[l], 1);
            if (l == 0) { J.a_slot = 1; J.a_off = off; }
            pl.one(0, J);
        }
    };
    d.KPl = hp[Lp - 1]; d.KQl = hq[Lq - 1]; d.HQ0 = hq[0];
    d.PHTl = PHT[Lp - 1]; d.WhT = Wp(5, Lp); d.bhT = Bp(5, Lp); d.HDTw = HDT;
    d.PHl = PHP[Lp - 1]; d.Wh = Wp(0, Lp); d.bh = Bp(0, Lp); d.HDPw = HDP; d.dPZl = dPZ[Lp - 1];
    for (int k = 0; k < 2; ++k) { d.QHl[k] = QH[k][Lq - 1]; d.THl[k] = TH[k][Lq - 1]; d.dQZl[k] = dQZ[k][Lq - 1]; }
    for (int k = 0; k < 4; ++k) { d.Wl[k] = Wp(1 + k, Lq); d.bl[k] = Bp(1 + k, Lq); d.QOw[k] = QO[k]; }
    d.AHl = AH[Lq - 1]; d.dAZl = dAZ[Lq - 1]; d.dAZ0 = dAZ[0]; d.W1q[0] = Wp(1, 0); d.QAw = QA; d.DAaw = DAa;
    // ---- the critic pass ----
    pl.list = &g->td3_critic;
    policy_fwd(5, XL.off_nobs, PHT);
    pl.plain(GS_TD3_HEAD);
    for (int l = 0; l < Lq; ++l) {                       // Q1, Q2 on (s, a), their targets on (s', a~): the hidden layers
        pl.begin(0);
        for (int k = 0; k < 2; ++k) {
            const GenLayer &L = g->net[1 + k].L[l];
            pl.add(gen_fwd(l == 0 ? d.XQ : QH[k][l - 1], n, Wp(1 + k, l), Bp(1 + k, l), L.N, L.K, l < Lq ? QH[k][l] : QO[k], l < Lq));
        }
        for (int k = 0; k < 2; ++k) {
            const GenLayer &L = g->net[3 + k].L[l];
            pl.add(gen_fwd(l == 0 ? d.XQ + (long long)n * ldq : TH[k][l - 1], n, Wp(3 + k, l), Bp(3 + k, l), L.N, L.K, l < Lq ? TH[k][l] : QO[2 + k], l < Lq));
        }
        pl.end();
    }
    pl.plain(GS_TD3_LOSS);
    for (int j = Lq - 1; j >= 1; --j) {
        pl.begin(1);
        for (int k = 0; k < 2; ++k) {
            const GenLayer &L = g->net[1 + k].L[j];
            pl.add(gen_bwd(dQZ[k][j], n, L.N, Wp(1 + k, j), L.K, 0, L.K, dQZ[k][j - 1], L.K, QH[k][j - 1], L.K));
        }
        pl.end();
    }
    pl.begin(2);
    for (int k = 0; k < 2; ++k)
        for (int l = Lq; l >= 0; --l) {
            const GenLayer &L = g->net[1 + k].L[l];
            pl.add(gen_dw(l == Lq ? d.DQ[k] : dQZ[k][l], l == 0 ? d.XQ : QH[k][l - 1], n, g->net[1 + k], L, c->qf_learning_rate, &g->net[3 + k]));
        }
    pl.end();
    pl.plain(GS_DIAG);
    // ---- the actor pass (and its forward half for statistics steps) ----
    for (int full = 1; full >= 0; --full) {
        pl.list = full ? &g->td3_actor : &g->td3_stats;
        policy_fwd(0, XL.off_obs, PHP);
        pl.plain(GS_TD3_AHEAD);
        for (int l = 0; l < Lq; ++l) {                   // Q1(s, policy(s)) through the updated qf1: the hidden layers
            const GenLayer &L = g->net[1].L[l];
            pl.one(0, gen_fwd(l == 0 ? d.XA : AH[l - 1], n, Wp(1, l), Bp(1, l), L.N, L.K, AH[l], 1));
        }
        pl.plain(GS_TD3_QA);                             // its last layer (and, on policy steps, the backward pass through it)
        pl.list->back().mode = full;
        if (full) {
            for (int j = Lq - 1; j >= 1; --j) {
                const GenLayer &L = g->net[1].L[j];
                pl.one(1, gen_bwd(dAZ[j], n, L.N, Wp(1, j), L.K, 0, L.K, dAZ[j - 1], L.K, AH[j - 1], L.K));
            }
            pl.plain(GS_TD3_POLGRAD);
            for (int j = Lp - 1; j >= 1; --j) {
                const GenLayer &L = g->net[0].L[j];
                pl.one(1, gen_bwd(dPZ[j], n, L.N, Wp(0, j), L.K, 0, L.K, dPZ[j - 1], L.K, PHP[j - 1], L.K));
            }
            pl.begin(2);
            for (int l = Lp; l >= 0; --l) {
                const GenLayer &L = g->net[0].L[l];
                gen::GemmJob Jw = gen_dw(l == Lp ? d.DHP : dPZ[l], l == 0 ? nullptr : PHP[l - 1], n, g->net[0], L, c->policy_learning_rate, &g->net[5]);
                if (l == 0) { Jw.b_slot = 1; Jw.b_off = XL.off_obs; }
                pl.add(Jw);
            }
            pl.end();
        }
        pl.plain(GS_DIAG);
    }
    return gen_commit_plan(t, pl, {&g->td3_critic, &g->td3_actor, &g->td3_stats});
}

void gen_destroy(sac_general *g) {
    if (!g) return;
    if (g->arena) (void)hipFree(g->arena);
    if (g->d_scratch) (void)hipFree(g->d_scratch);
    if (g->d_tile_cnt) (void)hipFree(g->d_tile_cnt);
    delete g;
}

int gen_upload(sac_trainer *t, int net, const float *flat, float *dst) {
    const GenNet &N = t->gen->net[net];
    std::vector<float> h((size_t)N.n, 0.f);
    for (long long i = 0; i < N.nflat; ++i) h[(size_t)N.perm[(size_t)i]] = flat[i];
    SAC_HIP(hipMemcpyAsync(dst, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice, t->stream));
    SAC_HIP(hipStreamSynchronize(t->stream));
    return 0;
}

int gen_download(sac_trainer *t, int net, const float *src, float *flat) {
    const GenNet &N = t->gen->net[net];
    std::vector<float> h((size_t)N.n);
    SAC_HIP(hipMemcpyAsync(h.data(), src, sizeof(float) * h.size(), hipMemcpyDeviceToHost, t->stream));
    SAC_HIP(hipStreamSynchronize(t->stream));
    for (long long i = 0; i < N.nflat; ++i) flat[i] = h[(size_t)N.perm[(size_t)i]];
    return 0;
}

// one launch list on minibatch slot S
// (polyak: the weight-gradient launch of this list also soft-updates the targets of the nets it trains)
int gen_run_list(sac_trainer *t, const std::vector<GenStage> &list, const float *S, const SlotLayout &SL, const StepArg &sa,
                 bool polyak) {
    sac_general *g = t->gen;
    hipStream_t s = t->stream;
    gen::GDev &d = g->dev;
    const int n = g->n, A = g->A;
    auto blocks = [](long long work, int cap) { const long long b = (work + 255) / 256; return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b)); };
    // (kernels with a thread per (row, action) need ceil(rows / 16) workgroups; their copy of the slot's rows into the input
    //  matrices spreads over up to 128)
    auto head_grid = [&](int rows, long long copy_elems) {
        const unsigned need = (unsigned)((rows + 15) / 16), copy = blocks(copy_elems / 4, 128);
        return dim3(need > copy ? need : copy);
    };
    for (const GenStage &st : list) {
        switch (st.kind) {
        case GS_GEMM: {
            gen::GemmStage gs = st.gs;
            gs.S = S;
            gs.bc1 = sa.bc1; gs.bc2s = sa.bc2s; gs.tau = d.tau; gs.polyak = polyak ? 1 : 0; gs.keep_grad = (sa.pad2 & 2u) ? 1 : 0;
#ifdef SAC_STAMPS
            { static const char *e = getenv("SAC_GEN_STAMP_STAGE"); gs.stamp = (e && atoi(e) == (int)(&st - list.data())) ? 1 : 0; }
#endif
            const dim3 grid(gs.ntiles * gs.splitk);
            if (st.mode == 0) hipLaunchKernelGGL((gen::k_g_gemm<true, true>), grid, dim3(64 * gen::GW), 0, s, gs);
            else if (st.mode == 1) hipLaunchKernelGGL((gen::k_g_gemm<true, false>), grid, dim3(64 * gen::GW), 0, s, gs);
            else hipLaunchKernelGGL((gen::k_g_gemm<false, false>), grid, dim3(64 * gen::GW), 0, s, gs);
            break;
        }
        // (SAC: one workgroup per row / per GRW rows -- the kernels hold the layers of a handful of outputs, see gen_build)
        case GS_HEAD: if (A <= 8) hipLaunchKernelGGL(gen::k_g_head<8>, dim3((2 * n + gen::GRW - 1) / gen::GRW), dim3(256), 0, s, d, S, SL, sa);
            else hipLaunchKernelGGL(gen::k_g_head<16>, dim3((2 * n + gen::GRW - 1) / gen::GRW), dim3(256), 0, s, d, S, SL, sa); break;
        case GS_LOSS: hipLaunchKernelGGL(gen::k_g_loss, dim3(n), dim3(256), 0, s, d, S, SL); break;
        case GS_POLGRAD: if (A <= 8) hipLaunchKernelGGL(gen::k_g_polgrad<8>, dim3((n + gen::GRW - 1) / gen::GRW), dim3(256), 0, s, d);
            else hipLaunchKernelGGL(gen::k_g_polgrad<16>, dim3((n + gen::GRW - 1) / gen::GRW), dim3(256), 0, s, d);
            break;
        case GS_DIAG:       // (SAC: on the steps whose diagnostics somebody reads -- the first and the last of a loop, single steps)
            if (t->algo == 1 || (sa.pad2 & 2u) || sa.loop_pos == 0) hipLaunchKernelGGL(gen::k_g_diag, dim3(1), dim3(256), 0, s, d, sa);
            break;
        case GS_TD3_HEAD: if (A <= 8) hipLaunchKernelGGL(gen::k_g_td3_head<8>, dim3((n + gen::GRW - 1) / gen::GRW), dim3(256), 0, s, d, S, SL, sa);
            else hipLaunchKernelGGL(gen::k_g_td3_head<16>, dim3((n + gen::GRW - 1) / gen::GRW), dim3(256), 0, s, d, S, SL, sa);
            break;
        case GS_TD3_LOSS: hipLaunchKernelGGL(gen::k_g_td3_loss, dim3(n), dim3(256), 0, s, d, S, SL); break;
        case GS_TD3_AHEAD: if (A <= 8) hipLaunchKernelGGL(gen::k_g_td3_ahead<8>, dim3((n + gen::GRW - 1) / gen::GRW), dim3(256), 0, s, d, S, SL);
            else hipLaunchKernelGGL(gen::k_g_td3_ahead<16>, dim3((n + gen::GRW - 1) / gen::GRW), dim3(256), 0, s, d, S, SL);
            break;
        case GS_TD3_QA: hipLaunchKernelGGL(gen::k_g_td3_qa, dim3(n), dim3(256), 0, s, d, st.mode); break;
        case GS_TD3_POLGRAD: if (A <= 8) hipLaunchKernelGGL(gen::k_g_td3_polgrad<8>, dim3((n + gen::GRW - 1) / gen::GRW), dim3(256), 0, s, d);
            else hipLaunchKernelGGL(gen::k_g_td3_polgrad<16>, dim3((n + gen::GRW - 1) / gen::GRW), dim3(256), 0, s, d);
            break;
        }
    }
    SAC_HIP(hipGetLastError());
    return 0;
}

// one step on minibatch slot S.  SAC: the launch list built by gen_build.  TD3 (rlkit TD3Trainer.train_from_torch, as
// launch_step_td3): the critic pass every step -- its Adam launch also soft-updates the critics' targets on policy steps --
// and the actor pass on policy steps (n_train_steps_total % policy_and_target_update_period == 0: Q1(s, policy(s)) through the
// ALREADY UPDATED qf1, policy backward, policy Adam + soft update of the target policy); want_stats on another step: the
// actor pass's forward half only, for Policy Loss / Policy Action (rlkit recomputes them for the epoch statistics).
int gen_launch_step(sac_trainer *t, const float *S, const SlotLayout &SL, int j, bool want_stats) {
    sac_general *g = t->gen;
    SAC_REQUIRE(SL.off_obs == t->ext_layout.off_obs && SL.off_nobs == t->ext_layout.off_nobs && SL.Bt == g->n,
                "internal: minibatch slot layout differs from the one the general step was built for");
    g->dev.eps1 = t->dev.eps1; g->dev.eps2 = t->dev.eps2;
    const double tt = (double)(t->adam_t + 1);
    StepArg sa{t->n_train_steps_total, t->adam_t + 1, j, 0, 1.0 - std::pow(0.9, tt), std::sqrt(1.0 - std::pow(0.999, tt))};
    sa.pad2 = t->publish_diag ? 2u : 0u;
    if (t->algo == 0) {
        if (gen_run_list(t, g->stages, S, SL, sa, (t->n_train_steps_total % g->dev.period) == 0)) return -1;
    } else {
        const bool pstep = (t->n_train_steps_total % t->td3_period) == 0, actor = pstep || want_stats;
        const double tp = (double)(t->adam_t_pi + 1);
        StepArg sp{t->n_train_steps_total, t->adam_t_pi + 1, j, 2, 1.0 - std::pow(0.9, tp), std::sqrt(1.0 - std::pow(0.999, tp))};
        sp.pad2 = sa.pad2;
        sa.pad = 1;
        if (gen_run_list(t, g->td3_critic, S, SL, sa, pstep)) return -1;
        if (actor && gen_run_list(t, pstep ? g->td3_actor : g->td3_stats, S, SL, sp, true)) return -1;
        if (pstep) t->adam_t_pi += 1;
    }
    t->n_train_steps_total += 1;
    t->adam_t += 1;
    return 0;
}

// sac_debug_fetch on a general trainer: the same names, rows of the true batch
int64_t gen_debug_fetch(sac_trainer *t, const std::string &nm, float *out, int64_t cap) {
    sac_general *g = t->gen;
    const gen::GDev &d = g->dev;
    const int64_t n = g->n, nA = (int64_t)g->n * g->A;
    struct V { const char *name; const float *p; int64_t cnt; };
    if (t->algo == 1) {
        const V tv[] = {{"a_next", d.a2, nA}, {"a_new", d.pa, nA}, {"q1", d.QO[0], n}, {"q2", d.QO[1], n}, {"tq1", d.QO[2], n},
                        {"tq2", d.QO[3], n}, {"q_target", d.y, n}, {"q1_new", d.QA, n},
                        {"diag_trace", d.diag_trace, (int64_t)DIAG_TRACE_CAP * SAC_DIAG_N}};
        for (const V &v : tv)
            if (nm == v.name) {
                const int64_t cnt = nm == "diag_trace" && cap < v.cnt ? cap : v.cnt;
                if (cap < cnt) { sac::set_error("buffer too small"); return -2; }
                if (hipMemcpyAsync(out, v.p, sizeof(float) * cnt, hipMemcpyDeviceToHost, t->stream) != hipSuccess ||
                    hipStreamSynchronize(t->stream) != hipSuccess) { sac::set_error("copy failed in sac_debug_fetch"); return -1; }
                return cnt;
            }
    }
    const V vs[] = {{"a_new", d.anew, nA}, {"mu", d.mu, nA}, {"log_std", d.ls, nA}, {"a_next", d.a2, nA},
                    {"log_pi", d.logpi, n}, {"log_pi_next", d.logpi2, n}, {"q1", d.QO[0], n}, {"q2", d.QO[1], n},
                    {"q1_new", d.QO[0] + n, n}, {"q2_new", d.QO[1] + n, n}, {"tq1", d.QO[2], n}, {"tq2", d.QO[3], n},
                    {"q_target", d.y, n}, {"diag_trace", d.diag_trace, (int64_t)DIAG_TRACE_CAP * SAC_DIAG_N}};
    for (const V &v : vs)
        if (t->algo == 0 && nm == v.name) {
            const int64_t cnt = nm == "diag_trace" && cap < v.cnt ? cap : v.cnt;
            if (cap < cnt) { sac::set_error("buffer too small"); return -2; }
            if (hipMemcpyAsync(out, v.p, sizeof(float) * cnt, hipMemcpyDeviceToHost, t->stream) != hipSuccess ||
                hipStreamSynchronize(t->stream) != hipSuccess) { sac::set_error("copy failed in sac_debug_fetch"); return -1; }
            return cnt;
        }
    const char *gn[3] = {"g_policy", "g_qf1", "g_qf2"};
    for (int i = 0; i < 3; ++i)
        if (nm == gn[i]) {
            if (cap < g->net[i].nflat) { sac::set_error("buffer too small"); return -2; }
            if (gen_download(t, i, g->net[i].G, out)) return -1;
            return g->net[i].nflat;
        }
    sac::set_error("unknown debug tensor '%s'", nm.c_str());
    return -2;
}

}  // namespace
