// HBM replay buffer: SoA fp32 storage, device MT19937 index generation (bit-exact with NumPy's
// legacy np.random.randint) and the LDS-staged row gather into contiguous minibatch slots.
//
// Replaces rlkit EnvReplayBuffer (call sites /root/reference/util/rlkit_utils.py:139-142,
// /root/reference/util/rlkit_custom.py:207,230,235-236).  Index algorithm: SURVEY.md Appendix B.
#include "sac_common.h"

#include <cstring>
#include <mutex>
#include <unordered_set>
#include <vector>

namespace sac {

static thread_local std::string g_err;
void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}
const char *last_error() { return g_err.c_str(); }

// ------------------------------------------------------------------------------------------
// k_mt_randint: ONE workgroup of four waves walks the MT19937 stream in order.
//   Twist: element i of the new state needs old mt[i], old mt[i+1] and old mt[i+397] (i < 227) or NEW mt[i-227]
//   (i >= 227), so the 624 words fall into three dependent groups -- [0,227) from the old state alone, [227,454) from the
//   first group, [454,624) from the second (word 623 also takes the new word 0) -- each a single pass of <= 227 threads.
//   Old and new state are two LDS arrays, so a group needs one barrier, not a read barrier and a write barrier.
//   Draws: tempering, masking and rejection are thread-parallel over the 624 words (three per thread); accepted draws
//   are compacted in draw order -- a ballot prefix inside a wave, the twelve (pass, wave) counts through LDS -- so output
//   j is exactly the j-th accepted draw and the stream stops right after the draw that produced the last output.
// Round 1 ran this on one wave (ten 64-wide passes per twist, two barriers each): 4.4 ns per index against ~1 here; the
// draw of a loop's first chunk sits in front of its first step.
// ------------------------------------------------------------------------------------------
// Output j goes to out[(j / bt) * bp + j % bt]: batches of bt indices stored at a stride of bp >= bt (the row-block
// padding of a batch whose size is not a multiple of 16; the pad entries keep a valid index and are never drawn).
__global__ __launch_bounds__(256) void k_mt_randint(MtState *st, uint32_t rng, uint32_t mask, int64_t count,
                                                    int64_t *__restrict__ out, int bt, int bp) {
    __shared__ uint32_t mt[2][MT_N];
    __shared__ int s_cnt[2][12];
    __shared__ int s_newpos;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < MT_N; i += 256) mt[0][i] = st->mt[i];
    int pos = st->pos, cur = 0;
    __syncthreads();
    constexpr int G = MT_N - MT_M;                 // 227
    auto mix = [](uint32_t hi, uint32_t lo, uint32_t far) {
        const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
        return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    };
    int64_t produced = 0;
    for (int it = 0;; ++it) {
        if (pos >= MT_N) {
            const uint32_t *o = mt[cur];
            uint32_t *n = mt[cur ^ 1];
            if (tid < G) n[tid] = mix(o[tid], o[tid + 1], o[tid + MT_M]);
            __syncthreads();
            if (tid < G) { const int i = G + tid; n[i] = mix(o[i], o[i + 1], n[i - G]); }
            __syncthreads();
            if (tid < MT_N - 2 * G) { const int i = 2 * G + tid; n[i] = mix(o[i], (i == MT_N - 1) ? n[0] : o[i + 1], n[i - G]); }
            __syncthreads();
            cur ^= 1;
            pos = 0;
        }
        const uint32_t *m = mt[cur];
        uint32_t v[3];
        bool ok[3];
        int before[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int i = p * 256 + tid;
            const bool valid = i < MT_N && i >= pos;
            uint32_t y = valid ? m[i] : 0u;
            y ^= y >> 11;
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= y >> 18;
            v[p] = y & mask;
            ok[p] = valid && (v[p] <= rng);
            const unsigned long long bal = __ballot(ok[p]);
            before[p] = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) s_cnt[it & 1][p * 4 + wave] = __popcll(bal);
        }
        __syncthreads();
        int total = 0, base[3] = {0, 0, 0};
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const int c = s_cnt[it & 1][q];
#pragma unroll
            for (int p = 0; p < 3; ++p) base[p] += (q < p * 4 + wave) ? c : 0;
            total += c;
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int64_t slot = produced + base[p] + before[p];
            if (ok[p] && slot < count) {
                out[(bt == bp) ? slot : (slot / bt) * bp + slot % bt] = (int64_t)v[p];
                if (slot == count - 1) s_newpos = p * 256 + tid + 1;
            }
        }
        if (produced + total >= count) {       // (uniform: every thread sums the same twelve counts)
            __syncthreads();
            pos = s_newpos;
            break;
        }
        produced += total;
        pos = MT_N;
    }
    for (int i = tid; i < MT_N; i += 256) st->mt[i] = mt[cur][i];
    if (tid == 0) st->pos = pos;
}

// ------------------------------------------------------------------------------------------
// k_gather: persistent 256-thread workgroups, each moving 16-row blocks of minibatch slots:
//   HBM rows (16-B aligned, padded stride) --dwordx4--> registers --> LDS tile --dwordx4--> slot.
// The LDS tile lets the unpadded (B,O) output be written as full 16-B-per-lane coalesced stores
// and lets the same rows be re-emitted feature-major (saT[KQ][B], fragment-major: frag_off) for the weight-gradient kernel.
// Three-stage software pipeline per workgroup (indices of block n+2, rows of block n+1, write-out
// of block n), so a workgroup always has a row burst in flight while it stores: the gather is
// bound by bytes in flight per CU, not by the two dependent round trips (index -> row) per block.
// Barriers wait for LDS only (a __syncthreads() would drain the prefetched rows).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LOAD_IDX / ISSUE_ROWS work on plain local arrays with static indices (structs passed by reference
// into lambdas end up in scratch memory, and a scratch store of an in-flight load is a wait).
#define GATHER_LOAD_IDX(BLK, IO, IA, IS)                                                          \
    do {                                                                                          \
        const int _b = (BLK);                                                                     \
        const int _sl = _b / blocks_per_slot;                                                     \
        const int64_t _base = (_b < nblk32) ? (int64_t)_sl * B + (_b - _sl * blocks_per_slot) * RB : 0; \
        _Pragma("unroll") for (int i = 0; i < NIT; ++i) IO[i] = (int)idx[_base + (ro[i] >= 0 ? ro[i] : 0)]; \
        IA = (int)idx[_base + (ra >= 0 ? ra : 0)];                                                \
        IS = (int)idx[_base + (rs >= 0 ? rs : 0)];                                                \
    } while (0)
// the indices become "known" only here: otherwise the compiler hoists idx * stride up behind the
// index load issued one phase earlier and waits (vmcnt(0)) in the middle of the write-out
#define GATHER_ISSUE_ROWS(IO, IA, IS, DO, DN, DA_, DS, DS2)                                       \
    do {                                                                                          \
        _Pragma("unroll") for (int i = 0; i < NIT; ++i) asm volatile("" : "+v"(IO[i]));            \
        asm volatile("" : "+v"(IA));                                                              \
        asm volatile("" : "+v"(IS));                                                              \
        _Pragma("unroll") for (int i = 0; i < NIT; ++i) {                                         \
            const int64_t _src = (int64_t)IO[i] * Ost + 4 * qo[i];                                \
            DO[i] = *reinterpret_cast<const float4 *>(rv.obs + _src);                             \
            DN[i] = *reinterpret_cast<const float4 *>(rv.nobs + _src);                            \
        }                                                                                         \
        DA_ = *reinterpret_cast<const float4 *>(rv.act + (int64_t)IA * Ast + 4 * qa);             \
        DS = rv.rew[IS];  /* both unconditional: a conditional load is a branch whose merge */    \
        DS2 = rv.term[IS]; /* point waits for the data */                                         \
    } while (0)

template <int NIT>      // obs 16-B chunks per thread: ceil(16 * (Ost/4) / 256)
__global__ __launch_bounds__(256) void k_gather(ReplayView rv, const int64_t *__restrict__ idx, int B,
                                                int64_t n_blocks_total, float *__restrict__ slots,
                                                SlotLayout L, int write_saT) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int Ost = rv.Ost, Ast = rv.Ast, O = rv.O, A = rv.A;
    float *t_obs = lds;                       // [16][Ost]
    float *t_nobs = t_obs + RB * Ost;         // [16][Ost]
    float *t_act = t_nobs + RB * Ost;         // [16][Ast]
    const int tid = threadIdx.x;
    const int blocks_per_slot = B / RB;
    const int oc = Ost >> 2, ac = Ast >> 2;   // 16-B chunks per row
    const int nblk32 = (int)n_blocks_total;   // launch_gather caps the block count below 2^30
    // this thread's fixed share of a block: obs chunks c_i = tid + 256 i, one act chunk, one scalar
    int ro[NIT], qo[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int c = tid + 256 * i;
        ro[i] = (c < RB * oc) ? c / oc : -1;
        qo[i] = (c < RB * oc) ? c - ro[i] * oc : 0;
    }
    const int ra = (tid < RB * ac) ? tid / ac : -1, qa = (tid < RB * ac) ? tid - ra * ac : 0;
    const int rs = (tid < 2 * RB) ? (tid & (RB - 1)) : -1;         // tid < 16: reward, 16..31: terminal

    // The element -> (row, column) maps of the write-out are the same for every block: for narrow rows
    // (NIT == 1: one pass per thread) they are computed once, outside the block loop (the integer
    // divisions by the runtime row length otherwise dominate the write-out's instruction count).
    // (feature-major copy: O + A <= 64 + 16 features x 4 row groups = up to 320 elements -> two passes of 256 threads)
    int mo[4] = {0, 0, 0, 0}, ma[4] = {0, 0, 0, 0}, ms[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    const bool has_o = tid < ((RB * O) >> 2), has_a = tid < ((RB * A) >> 2);
    const bool has_s[2] = {tid < (O + A) * 4, tid + 256 < (O + A) * 4};
    if constexpr (NIT == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int eo = 4 * tid + j, ro_ = eo / O;
            mo[j] = has_o ? ro_ * Ost + (eo - ro_ * O) : 0;
            const int ea = 4 * tid + j, ra_ = ea / A;
            ma[j] = has_a ? ra_ * Ast + (ea - ra_ * A) : 0;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int f = (tid + 256 * ps) >> 2, r = 4 * (tid & 3) + j;
                ms[ps][j] = has_s[ps] ? ((f < O) ? r * Ost + f : 2 * RB * Ost + r * Ast + (f - O)) : 0;   // t_act follows t_nobs
            }
        }
    }

    // stage C: registers of one block -> LDS tile -> slot
    auto write_out = [&](int blk, const float4 (&Do)[NIT], const float4 (&Dn)[NIT], const float4 &Da, float Ds,
                         float Ds2) {
        const int slot = blk / blocks_per_slot;
        const int row0 = (blk - slot * blocks_per_slot) * RB;
        float *S = slots + (int64_t)slot * L.slot_floats;
#pragma unroll
        for (int i = 0; i < NIT; ++i)
            if (ro[i] >= 0) {
                *reinterpret_cast<float4 *>(t_obs + ro[i] * Ost + 4 * qo[i]) = Do[i];
                *reinterpret_cast<float4 *>(t_nobs + ro[i] * Ost + 4 * qo[i]) = Dn[i];
            }
        if (ra >= 0) *reinterpret_cast<float4 *>(t_act + ra * Ast + 4 * qa) = Da;
        if (rs >= 0) S[(tid < RB ? L.off_rew : L.off_term) + row0 + rs] = (tid < RB) ? Ds : Ds2;
        lds_barrier();
        if constexpr (NIT == 1) {
            if (has_o) {
                float4 a, b;
                a.x = t_obs[mo[0]]; a.y = t_obs[mo[1]]; a.z = t_obs[mo[2]]; a.w = t_obs[mo[3]];
                b.x = t_nobs[mo[0]]; b.y = t_nobs[mo[1]]; b.z = t_nobs[mo[2]]; b.w = t_nobs[mo[3]];
                *reinterpret_cast<float4 *>(S + L.off_obs + (int64_t)row0 * O + 4 * tid) = a;
                *reinterpret_cast<float4 *>(S + L.off_nobs + (int64_t)row0 * O + 4 * tid) = b;
            }
            if (has_a) {
                float4 a;
                a.x = t_act[ma[0]]; a.y = t_act[ma[1]]; a.z = t_act[ma[2]]; a.w = t_act[ma[3]];
                *reinterpret_cast<float4 *>(S + L.off_act + (int64_t)row0 * A + 4 * tid) = a;
            }
#pragma unroll
            for (int ps = 0; ps < 2; ++ps)
                if (write_saT && has_s[ps]) {
                    float4 v;
                    v.x = lds[ms[ps][0]]; v.y = lds[ms[ps][1]]; v.z = lds[ms[ps][2]]; v.w = lds[ms[ps][3]];
                    const int f = (tid + 256 * ps) >> 2, frow = (f < O) ? f : L.KA + (f - O);
                    *reinterpret_cast<float4 *>(S + L.off_saT + frag_off(frow, row0 + 4 * (tid & 3), B)) = v;   // fragment-major
                }
            lds_barrier();                         // tile free for the next block
            return;
        }
        {   // LDS -> contiguous row-major slot (16*O floats = 64*O bytes, 16-B aligned)
            float *dst = S + L.off_obs + (int64_t)row0 * O;
            float *dstn = S + L.off_nobs + (int64_t)row0 * O;
            const int n4 = (RB * O) >> 2;
            for (int c = tid; c < n4; c += 256) {
                float4 a, b;
                float *pa = &a.x, *pb = &b.x;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = 4 * c + j;
                    const int r = e / O, k = e - r * O;
                    pa[j] = t_obs[r * Ost + k];
                    pb[j] = t_nobs[r * Ost + k];
                }
                *reinterpret_cast<float4 *>(dst + 4 * c) = a;
                *reinterpret_cast<float4 *>(dstn + 4 * c) = b;
            }
            float *dsta = S + L.off_act + (int64_t)row0 * A;
            const int a4 = (RB * A) >> 2;
            for (int c = tid; c < a4; c += 256) {
                float4 a;
                float *pa = &a.x;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = 4 * c + j;
                    const int r = e / A, k = e - r * A;
                    pa[j] = t_act[r * Ast + k];
                }
                *reinterpret_cast<float4 *>(dsta + 4 * c) = a;
            }
        }
        if (write_saT) {   // LDS -> feature-major saT[f][row0 .. row0+15]
            float *T = S + L.off_saT;
            for (int c = tid; c < (O + A) * 4; c += 256) {
                const int f = c >> 2, q = c & 3;
                float4 v;
                float *pv = &v.x;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * q + j;
                    pv[j] = (f < O) ? t_obs[r * Ost + f] : t_act[r * Ast + (f - O)];
                }
                *reinterpret_cast<float4 *>(T + frag_off((f < O) ? f : L.KA + (f - O), row0 + 4 * q, B)) = v;
            }
        }
        lds_barrier();                             // tile free for the next block
    };

    // Two register sets, ping-pong (no copies: copying an in-flight load's destination is a wait).
    const int stride = gridDim.x;
    int io0[NIT], io1[NIT], ia0, ia1, is0, is1;
    float4 oA[NIT], nA[NIT], aA, oB[NIT], nB[NIT], aB;
    float sA, s2A, sB, s2B;
    int blk = blockIdx.x;
    GATHER_LOAD_IDX(blk, io0, ia0, is0);
    GATHER_LOAD_IDX(blk + stride, io1, ia1, is1);
    GATHER_ISSUE_ROWS(io0, ia0, is0, oA, nA, aA, sA, s2A);
    for (; blk < nblk32; blk += 2 * stride) {
        GATHER_LOAD_IDX(blk + 2 * stride, io0, ia0, is0);          // stage A: indices two blocks ahead
        GATHER_ISSUE_ROWS(io1, ia1, is1, oB, nB, aB, sB, s2B);     // stage B: next block's rows into flight
        __builtin_amdgcn_sched_barrier(0);
        write_out(blk, oA, nA, aA, sA, s2A);                       // stage C
        if (blk + stride < nblk32) {
            GATHER_LOAD_IDX(blk + 3 * stride, io1, ia1, is1);
            GATHER_ISSUE_ROWS(io0, ia0, is0, oA, nA, aA, sA, s2A);
            __builtin_amdgcn_sched_barrier(0);
            write_out(blk + stride, oB, nB, aB, sB, s2B);
        }
    }
}
#undef GATHER_LOAD_IDX
#undef GATHER_ISSUE_ROWS

int ensure_stage(sac_buffer *b, size_t bytes) {
    if (b->stage_bytes >= bytes) return 0;
    if (b->h_stage) SAC_HIP(hipHostFree(b->h_stage));
    b->h_stage = nullptr;
    b->stage_bytes = 0;
    SAC_HIP(hipHostMalloc(&b->h_stage, bytes, hipHostMallocDefault));
    b->stage_bytes = bytes;
    return 0;
}

// Index and slot buffers only ever grow: a loop of 20 steps behind one of 2000 (or the other way round) must not
// pay hipFree + hipMalloc + a memset inside its call (rlkit_custom.py:233-240 runs this loop once per epoch).
int ensure_idx(sac_buffer *b, int64_t n) {
    if (b->idx_cap >= n) return 0;
    if (b->d_idx) { SAC_HIP(hipStreamSynchronize(b->stream)); SAC_HIP(hipFree(b->d_idx)); }
    b->d_idx = nullptr;
    b->idx_cap = 0;
    SAC_HIP(hipMalloc(&b->d_idx, sizeof(int64_t) * n));
    SAC_HIP(hipMemsetAsync(b->d_idx, 0, sizeof(int64_t) * n, b->stream));      // pad entries: a valid index
    b->idx_cap = n;
    return 0;
}

int ensure_slots(sac_buffer *b, int B, int64_t n_slots) {
    SlotLayout L = make_slot_layout(B, b->O, b->A);
    const int64_t need = L.slot_floats * n_slots;
    if (b->spec_valid && (b->slots_cap < need || b->slot.Bt != B || !b->in_loop) && loop_spec_drop(b)) return -1;
    if (b->slots_cap < need) {
        if (b->d_slots) { SAC_HIP(hipStreamSynchronize(b->stream)); SAC_HIP(hipFree(b->d_slots)); }
        b->d_slots = nullptr;
        b->slots_cap = 0;
        SAC_HIP(hipMalloc(&b->d_slots, sizeof(float) * need));
        // padding rows of saT (between and after obs / act) are contracted by the weight-gradient kernel: keep them 0
        SAC_HIP(hipMemsetAsync(b->d_slots, 0, sizeof(float) * need, b->stream));
        b->slots_cap = need;
    } else if (b->slot.Bt != B) {
        // another batch size = another layout: EVERY slot of the allocation may hold rows of the old one where the
        // new one has padding, so the whole allocation is cleared (not just the slots this call asks for)
        SAC_HIP(hipMemsetAsync(b->d_slots, 0, sizeof(float) * b->slots_cap, b->stream));
        if (b->d_idx) SAC_HIP(hipMemsetAsync(b->d_idx, 0, sizeof(int64_t) * b->idx_cap, b->stream));
    }
    if (b->slot.Bt != B || b->slot.slot_floats != L.slot_floats) b->loop_pos = 0;
    b->slot = L;
    b->n_slots = n_slots;
    return 0;
}

// `batch` indices per batch, stored at a stride of round_up(batch, 16).  dst == null: into b->d_idx + idx_offset
// (idx_offset in elements of that padded layout; the buffer is grown as needed when idx_offset == 0).
int readahead_rollback(sac_buffer *b) {
    if (!b->in_loop) b->loop_streak = 0;         // (an outside touch: the next loop call is not "in a row")
    if (loop_spec_drop(b)) return -1;
    b->ra_streak = 0;
    if (b->ra_ahead <= 0) return 0;
    const int consumed = b->ra_chunk - (int)b->ra_ahead;
    // the speculative batches that nobody asked for: their slots hold no batch any more
    for (int64_t j = 0; j < b->ra_ahead; ++j) b->ring_token[(b->ring_next + j) % sac_buffer::NRING] = -1;
    b->ra_ahead = 0;
    SAC_HIP(hipMemcpyAsync(b->d_rng, b->d_rng_saved, sizeof(MtState), hipMemcpyDeviceToDevice, b->stream));
    if (consumed > 0) {      // the generator as `consumed` random_batch calls leave it (the indices themselves are not needed)
        b->ra_internal = true;
        const int rc = launch_sample(b, b->ra_batch, consumed, 0, b->d_ra_scratch, b->stream);
        b->ra_internal = false;
        if (rc) return -1;
    }
    return 0;
}

// ---- streams of live trainers -------------------------------------------------------------------------------
// A buffer remembers the stream its device-batch steps run on (step_stream: slot-release events are recorded there).
// That stream belongs to a trainer, which may be destroyed, re-created for another batch size or moved to a CU-masked
// stream while the buffer lives on: before the buffer touches the handle it asks whether the stream still exists.  (A
// trainer drains its stream before giving it up, so a dead stream has no step in flight: its slots are simply free.)
static std::mutex g_streams_mu;
static std::unordered_set<hipStream_t> g_streams;
void stream_register(hipStream_t s) { std::lock_guard<std::mutex> lk(g_streams_mu); g_streams.insert(s); }
void stream_unregister(hipStream_t s) { std::lock_guard<std::mutex> lk(g_streams_mu); g_streams.erase(s); }
bool stream_is_live(hipStream_t s) { std::lock_guard<std::mutex> lk(g_streams_mu); return g_streams.count(s) != 0; }

void forget_dead_step_stream(sac_buffer *b) {
    if (!b->step_stream || stream_is_live(b->step_stream)) return;
    b->step_stream = nullptr;
    b->multi_stream = false;                 // (events of per-step mode recorded on the dead stream have long fired)
    b->waited_stream = nullptr; b->waited_chunk_token = -1;
    for (int i = 0; i < sac_buffer::NRING; ++i) b->ring_in_use[i] = false;
    for (int i = 0; i < 4; ++i) b->free4_seq[i] = -1;
    b->free_waited_seq = b->step_seq - 1;
}

// ---- the generator bound to a host-resident state (sac_rng_bind_host) -------------------------------------
// Sequential MT19937 + masked rejection, the same stream k_mt_randint walks on the device.
static inline void mt_twist_host(uint32_t *mt) {
    auto mix = [](uint32_t hi, uint32_t lo, uint32_t far) {
        const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
        return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    };
    int i = 0;
    for (; i < MT_N - MT_M; ++i) mt[i] = mix(mt[i], mt[i + 1], mt[i + MT_M]);
    for (; i < MT_N - 1; ++i) mt[i] = mix(mt[i], mt[i + 1], mt[i + (MT_M - MT_N)]);
    mt[MT_N - 1] = mix(mt[MT_N - 1], mt[0], mt[MT_M - 1]);
}

static void mt_skip_accepted(MtState &st, uint32_t rng, uint32_t mask, int64_t count) {
    int pos = st.pos;
    while (count > 0) {
        if (pos >= MT_N) { mt_twist_host(st.mt); pos = 0; }
        uint32_t y = st.mt[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        count -= ((y & mask) <= rng) ? 1 : 0;
    }
    st.pos = pos;
}

static int upload_rng(sac_buffer *b, const MtState &s) {
    SAC_HIP(hipMemcpyAsync(b->d_rng, &s, sizeof(MtState), hipMemcpyHostToDevice, b->stream));
    SAC_HIP(hipStreamSynchronize(b->stream));      // (s may be a local; rare path: only when somebody else moved the stream)
    return 0;
}

// the speculative batches nobody asked for are void (the generator is about to be replaced: nothing to restore)
static void readahead_drop(sac_buffer *b) {
    b->ra_streak = 0;
    for (int64_t j = 0; j < b->ra_ahead; ++j) b->ring_token[(b->ring_next + j) % sac_buffer::NRING] = -1;
    b->ra_ahead = 0;
}

// a speculative first chunk of the next loop call is void: the device generator goes back to the mirror
int loop_spec_drop(sac_buffer *b) {
    if (!b->spec_valid) return 0;
    b->spec_valid = false;
    return upload_rng(b, b->host_seen);
}

int host_rng_sync_in(sac_buffer *b) {
    if (!b->host_key) return 0;
    const int32_t hp = *b->host_pos;
    if (hp == b->host_seen.pos && memcmp(b->host_key, b->host_seen.mt, sizeof(uint32_t) * MT_N) == 0) return 0;
    SAC_REQUIRE(hp >= 0 && hp <= MT_N, "bound host generator holds an invalid position %d", (int)hp);
    b->spec_valid = false; b->loop_streak = 0;   // (the generator is replaced below anyway)
    readahead_drop(b);
    memcpy(b->host_seen.mt, b->host_key, sizeof(uint32_t) * MT_N);
    b->host_seen.pos = hp;
    return upload_rng(b, b->host_seen);
}

// the state `st` as n_batches draws of `batch` indices from a buffer of this size leave it
void host_rng_skip(const sac_buffer *b, MtState &st, int batch, int64_t n_batches) {
    if (b->size <= 1) return;                          // (size 1: NumPy consumes no draws)
    const uint32_t rng = (uint32_t)(b->size - 1);
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    mt_skip_accepted(st, rng, mask, (int64_t)batch * n_batches);
}

// host_seen mirrors the generator ALWAYS (bound or not: ~5 ns per index, beside a device draw of ~1 us per batch); a bound
// generator also writes the new state through to the caller's words
void host_rng_advance(sac_buffer *b, int batch, int64_t n_batches) {
    host_rng_skip(b, b->host_seen, batch, n_batches);
    if (b->host_key) {
        memcpy(b->host_key, b->host_seen.mt, sizeof(uint32_t) * MT_N);
        *b->host_pos = b->host_seen.pos;
    }
}

int host_rng_adopt(sac_buffer *b, const MtState &s) {
    b->host_seen = s;
    if (b->host_key) {
        memcpy(b->host_key, s.mt, sizeof(uint32_t) * MT_N);
        *b->host_pos = s.pos;
    }
    return upload_rng(b, s);
}

int launch_sample(sac_buffer *b, int batch, int64_t n_batches, int64_t idx_offset, int64_t *dst, hipStream_t on) {
    if (!b->ra_internal) {
        SAC_REQUIRE(b->size > 0, "random_batch on an empty replay buffer");
        if (host_rng_sync_in(b)) return -1;
        if (readahead_rollback(b)) return -1;
    }
    hipStream_t q = on ? on : b->stream;
    const int bp = round_up(batch, RB);
    const int64_t count = (int64_t)batch * n_batches, padded = (int64_t)bp * n_batches;
    SAC_REQUIRE(b->size > 0, "random_batch on an empty replay buffer");
    SAC_REQUIRE(b->size - 1 <= 0xffffffffLL, "replay buffers above 2^32 slots are not supported");
    if (!dst) {
        if (idx_offset == 0 && ensure_idx(b, padded)) return -1;
        SAC_REQUIRE(idx_offset + padded <= b->idx_cap, "index buffer too small for this offset");
        dst = b->d_idx + idx_offset;
    }
    const uint32_t rng = (uint32_t)(b->size - 1);
    if (rng == 0) {     // NumPy: no draws consumed, all zeros
        SAC_HIP(hipMemsetAsync(dst, 0, sizeof(int64_t) * padded, q));
        return 0;
    }
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    hipLaunchKernelGGL(k_mt_randint, dim3(1), dim3(256), 0, q, b->d_rng, rng, mask, count, dst, batch, bp);
    SAC_HIP(hipGetLastError());
    // (behind the launch: the device draws meanwhile; a loop mirrors all its chunks in one go once everything is queued)
    if (!b->ra_internal) {
        if (b->defer_mirror) b->deferred_batches += n_batches;
        else host_rng_advance(b, batch, n_batches);
    }
    return 0;
}

int launch_gather(sac_buffer *b, const int64_t *d_idx, int batch, int64_t n_batches, float *d_slots,
                  const SlotLayout &L, int write_saT, hipStream_t on) {
    hipStream_t q = on ? on : b->stream;
    SAC_REQUIRE(batch > 0 && batch % RB == 0, "batch size %d must be a positive multiple of %d", batch, RB);
    const int64_t nblk = (int64_t)(batch / RB) * n_batches;
    SAC_REQUIRE(nblk < (1LL << 30), "too many rows in one gather launch");
    const int grid = (int)(nblk < 1024 ? nblk : 1024);          // 4 persistent workgroups per CU (VGPR-limited)
    const size_t lds = sizeof(float) * (size_t)(2 * RB * b->Ost + RB * b->Ast);
    SAC_REQUIRE(lds <= 64 * 1024, "observation rows too wide for the gather tile (%zu B LDS)", lds);
    const int nit = (RB * (b->Ost >> 2) + 255) / 256;            // obs chunks per thread
    SAC_REQUIRE(nit <= 8, "observation rows too wide for the gather kernel (obs_dim %d)", b->O);
#define SAC_GATHER_LAUNCH(N)                                                                               \
    hipLaunchKernelGGL(k_gather<N>, dim3(grid), dim3(256), lds, q, b->view(), d_idx, batch, nblk, \
                       d_slots, L, write_saT)
    if (nit <= 1) SAC_GATHER_LAUNCH(1);
    else if (nit <= 2) SAC_GATHER_LAUNCH(2);
    else if (nit <= 4) SAC_GATHER_LAUNCH(4);
    else SAC_GATHER_LAUNCH(8);
#undef SAC_GATHER_LAUNCH
    SAC_HIP(hipGetLastError());
    return 0;
}

}  // namespace sac

using namespace sac;

// one contiguous ring segment [at, at+n) <- staged rows.
// Asynchronous ingest (the reference inserts the 2 500 exploration steps of an epoch between rollouts and training,
// /root/reference/util/rlkit_custom.py:223-231): rows are packed into one of TWO pinned staging buffers and handed to
// the copy engine with hipMemcpyAsync on the buffer's stream; the call returns as soon as the copies are ENQUEUED.
// While the DMA of one staging buffer runs, the host packs the next chunk into the other one (or goes back to
// stepping environments).  Everything that reads the storage (index draw + gather, sac_buffer_read) runs on the same
// in-order stream, so it sees the inserted rows without any host-side wait; a staging buffer is reused only after the
// event recorded behind its copies has fired.  The caller's arrays are free for reuse when the call returns (they
// have been copied into pinned memory by then).
template <typename Src>
static int add_segment(sac_buffer *b, int64_t at, int64_t n, const Src *obs, const Src *act, const Src *rew,
                       const Src *nobs, const uint8_t *term) {
    const int O = b->O, A = b->A, Ost = b->Ost, Ast = b->Ast;
    const int64_t CH = sac_buffer::INGEST_ROWS;
    const size_t per_row = sizeof(float) * (size_t)(2 * Ost + Ast + 2);
    for (int k = 0; k < 2; ++k)
        if (!b->h_ing[k]) {
            SAC_HIP(hipHostMalloc(&b->h_ing[k], per_row * CH, hipHostMallocDefault));
            SAC_HIP(hipEventCreateWithFlags(&b->ing_free[k], hipEventDisableTiming));
        }
    for (int64_t done = 0; done < n; done += CH) {
        const int64_t m = (n - done < CH) ? n - done : CH;
        const int k = (int)(b->ing_next++ & 1);
        if (b->ing_busy[k]) { SAC_HIP(hipEventSynchronize(b->ing_free[k])); b->ing_busy[k] = false; }
        float *so = (float *)b->h_ing[k], *sn = so + m * Ost, *sa = sn + m * Ost, *sr = sa + m * Ast, *st = sr + m;
        for (int64_t i = 0; i < m; ++i) {
            const Src *po = obs + (done + i) * O, *pn = nobs + (done + i) * O, *pa = act + (done + i) * A;
            for (int kk = 0; kk < O; ++kk) { so[i * Ost + kk] = (float)po[kk]; sn[i * Ost + kk] = (float)pn[kk]; }
            for (int kk = O; kk < Ost; ++kk) { so[i * Ost + kk] = 0.f; sn[i * Ost + kk] = 0.f; }
            for (int kk = 0; kk < A; ++kk) sa[i * Ast + kk] = (float)pa[kk];
            for (int kk = A; kk < Ast; ++kk) sa[i * Ast + kk] = 0.f;
            sr[i] = (float)rew[done + i];
            st[i] = term[done + i] ? 1.0f : 0.0f;
        }
        const int64_t r0 = at + done;
        SAC_HIP(hipMemcpyAsync(b->obs + r0 * Ost, so, sizeof(float) * m * Ost, hipMemcpyHostToDevice, b->stream));
        SAC_HIP(hipMemcpyAsync(b->nobs + r0 * Ost, sn, sizeof(float) * m * Ost, hipMemcpyHostToDevice, b->stream));
        SAC_HIP(hipMemcpyAsync(b->act + r0 * Ast, sa, sizeof(float) * m * Ast, hipMemcpyHostToDevice, b->stream));
        SAC_HIP(hipMemcpyAsync(b->rew + r0, sr, sizeof(float) * m, hipMemcpyHostToDevice, b->stream));
        SAC_HIP(hipMemcpyAsync(b->term + r0, st, sizeof(float) * m, hipMemcpyHostToDevice, b->stream));
        SAC_HIP(hipEventRecord(b->ing_free[k], b->stream));
        b->ing_busy[k] = true;
    }
    return 0;
}

template <typename Src>
static int add_impl(sac_buffer *b, int64_t n, const Src *obs, const Src *act, const Src *rew, const Src *nobs,
                    const uint8_t *term) {
    SAC_REQUIRE(b != nullptr, "null buffer");
    SAC_REQUIRE(n >= 0 && obs && act && rew && nobs && term, "bad arguments to sac_buffer_add");
    SAC_HIP(hipSetDevice(b->device));
    if (readahead_rollback(b)) return -1;        // (batches gathered ahead saw the old rows and the old size)
    const int O = b->O, A = b->A;
    int64_t i = 0;
    if (n > b->capacity) {      // only the last `capacity` samples survive; keep ring arithmetic exact
        const int64_t skip = n - b->capacity;
        b->top = (b->top + skip) % b->capacity;
        b->size = b->capacity;
        i = skip;
    }
    while (i < n) {
        const int64_t room = b->capacity - b->top;
        const int64_t m = (n - i < room) ? n - i : room;
        if (add_segment<Src>(b, b->top, m, obs + i * O, act + i * A, rew + i, nobs + i * O, term + i)) return -1;
        b->top = (b->top + m) % b->capacity;
        b->size = (b->size + m < b->capacity) ? b->size + m : b->capacity;
        i += m;
    }
    return 0;
}

extern "C" {

const char *sac_last_error(void) { return sac::last_error(); }
const char *sac_version(void) { return "sac_hip 0.2 (gfx950)"; }

int sac_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int buffer_build(sac_buffer *b, int64_t capacity, int obs_dim, int act_dim, int device);

int sac_buffer_create(sac_buffer_t **out, int64_t capacity, int obs_dim, int act_dim, int device) {
    SAC_REQUIRE(out != nullptr, "null out pointer");
    *out = nullptr;
    SAC_REQUIRE(capacity > 0 && obs_dim > 0 && act_dim > 0, "bad replay buffer shape (%lld, %d, %d)",
                (long long)capacity, obs_dim, act_dim);
    SAC_REQUIRE(sac_device_count() > 0, "no HIP device visible: libsac_hip has no CPU fallback");
    SAC_HIP(hipSetDevice(device));
    sac_buffer *b = new sac_buffer();
    if (buffer_build(b, capacity, obs_dim, act_dim, device)) {     // (the error message is already set)
        sac_buffer_destroy(b);
        return -1;
    }
    *out = b;
    return 0;
}

static int buffer_build(sac_buffer *b, int64_t capacity, int obs_dim, int act_dim, int device) {
    b->device = device;
    b->capacity = capacity;
    b->O = obs_dim; b->A = act_dim;
    b->Ost = round_up(obs_dim, 4); b->Ast = round_up(act_dim, 4);
    SAC_HIP(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    SAC_HIP(hipMalloc(&b->obs, sizeof(float) * capacity * b->Ost));
    SAC_HIP(hipMalloc(&b->nobs, sizeof(float) * capacity * b->Ost));
    SAC_HIP(hipMalloc(&b->act, sizeof(float) * capacity * b->Ast));
    SAC_HIP(hipMalloc(&b->rew, sizeof(float) * capacity));
    SAC_HIP(hipMalloc(&b->term, sizeof(float) * capacity));
    SAC_HIP(hipMalloc(&b->d_rng, sizeof(MtState)));
    for (auto &e : b->ev) { SAC_HIP(hipEventCreate(&e)); SAC_HIP(hipEventRecord(e, b->stream)); }     // (first record = signal set-up)
    return sac_rng_seed(b, 5489u);
}

int sac_buffer_destroy(sac_buffer_t *b) {
    if (!b) return 0;
    (void)hipSetDevice(b->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    if (b->spec_ev) (void)hipEventDestroy(b->spec_ev);
    for (void *p : {(void *)b->obs, (void *)b->nobs, (void *)b->act, (void *)b->rew, (void *)b->term,
                    (void *)b->d_rng, (void *)b->d_idx, (void *)b->d_slots})
        (void)hipFree(p);
    if (b->h_stage) (void)hipHostFree(b->h_stage);
    for (int k = 0; k < 2; ++k) {
        if (b->h_ing[k]) (void)hipHostFree(b->h_ing[k]);
        if (b->ing_free[k]) (void)hipEventDestroy(b->ing_free[k]);
    }
    (void)hipFree(b->d_ring); (void)hipFree(b->d_ring_idx);
    (void)hipFree(b->d_rng_saved); (void)hipFree(b->d_ra_scratch);
    for (auto &e : b->ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : b->ring_ready) if (e) (void)hipEventDestroy(e);
    for (auto &e : b->ring_free) if (e) (void)hipEventDestroy(e);
    for (auto &e : b->free4) if (e) (void)hipEventDestroy(e);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
    return 0;
}

// Stream confined to the CUs of the XCDs in `xcd_mask` (bit k = XCD k; experiments: independent runs side by side on one
// GPU, bench.py --replicas-per-gpu).  CU-mask bits are dealt round-robin over the 8 XCDs, so XCD k owns bits k, k + 8, ...
int sac_make_xcd_mask_stream(hipStream_t *out, unsigned xcd_mask) {
    int cus = 0, dev = 0;
    SAC_HIP(hipGetDevice(&dev));
    SAC_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    SAC_REQUIRE((xcd_mask & 0xffu) != 0 && (xcd_mask >> 8) == 0 && cus >= 8 && cus <= 512, "bad XCD mask 0x%x", xcd_mask);
    uint32_t mask[16] = {0};
    for (int i = 0; i < cus; ++i)
        if ((xcd_mask >> (i & 7)) & 1u) mask[i >> 5] |= 1u << (i & 31);
    SAC_HIP(hipExtStreamCreateWithCUMask(out, (uint32_t)((cus + 31) / 32), mask));
    return 0;
}
int sac_make_xcd_stream(hipStream_t *out, int xcd) {
    SAC_REQUIRE(xcd >= 0 && xcd < 8, "bad XCD index %d", xcd);
    return sac_make_xcd_mask_stream(out, 1u << xcd);
}

int sac_buffer_set_xcd_mask(sac_buffer_t *b, unsigned xcd_mask) {
    SAC_REQUIRE(b != nullptr, "null buffer");
    SAC_HIP(hipSetDevice(b->device));
    if (readahead_rollback(b)) return -1;
    SAC_HIP(hipStreamSynchronize(b->stream));
    hipStream_t ns = nullptr;
    if (sac_make_xcd_mask_stream(&ns, xcd_mask)) return -1;
    SAC_HIP(hipStreamDestroy(b->stream));
    b->stream = ns;
    return 0;
}
int sac_buffer_set_xcd(sac_buffer_t *b, int xcd) {
    SAC_REQUIRE(xcd >= 0 && xcd < 8, "bad XCD index %d", xcd);
    return sac_buffer_set_xcd_mask(b, 1u << xcd);
}

int64_t sac_buffer_size(const sac_buffer_t *b) { return b ? b->size : -1; }
int64_t sac_buffer_top(const sac_buffer_t *b) { return b ? b->top : -1; }
int64_t sac_buffer_capacity(const sac_buffer_t *b) { return b ? b->capacity : -1; }

int sac_buffer_add(sac_buffer_t *b, int64_t n, const float *obs, const float *act, const float *rew,
                   const float *next_obs, const uint8_t *term) {
    return add_impl<float>(b, n, obs, act, rew, next_obs, term);
}
int sac_buffer_add_f64(sac_buffer_t *b, int64_t n, const double *obs, const double *act, const double *rew,
                       const double *next_obs, const uint8_t *term) {
    return add_impl<double>(b, n, obs, act, rew, next_obs, term);
}

// 1 while inserted rows are still on their way to HBM, 0 once every enqueued insert has landed (never blocks)
int sac_buffer_ingest_pending(sac_buffer_t *b) {
    SAC_REQUIRE(b != nullptr, "null buffer");
    for (int k = 0; k < 2; ++k)
        if (b->ing_busy[k]) {
            const hipError_t e = hipEventQuery(b->ing_free[k]);
            if (e == hipErrorNotReady) return 1;
            if (e != hipSuccess) { sac::set_error("hipEventQuery failed: %s", hipGetErrorString(e)); return -1; }
            b->ing_busy[k] = false;
        }
    return 0;
}

// block until every enqueued insert has landed (nobody has to call this: readers are ordered behind the inserts on
// the buffer's stream; it exists for measurements and for callers that want the PCIe transfer done at a known point)
int sac_buffer_ingest_wait(sac_buffer_t *b) {
    SAC_REQUIRE(b != nullptr, "null buffer");
    SAC_HIP(hipSetDevice(b->device));
    for (int k = 0; k < 2; ++k)
        if (b->ing_busy[k]) { SAC_HIP(hipEventSynchronize(b->ing_free[k])); b->ing_busy[k] = false; }
    return 0;
}

// Storage rows [start, start + n) -> dense host arrays (checkpointing; the reference's buffer is plain
// NumPy arrays one can save, rlkit get_snapshot() returns {} for it -- SURVEY.md 8b "Snapshot contract").
int sac_buffer_read(sac_buffer_t *b, int64_t start, int64_t n, float *obs, float *act, float *rew, float *next_obs,
                    uint8_t *term) {
    SAC_REQUIRE(b && obs && act && rew && next_obs && term, "bad arguments to sac_buffer_read");
    SAC_REQUIRE(start >= 0 && n >= 0 && start + n <= b->capacity, "rows [%lld, %lld) outside the buffer",
                (long long)start, (long long)(start + n));
    if (n == 0) return 0;
    SAC_HIP(hipSetDevice(b->device));
    const int O = b->O, A = b->A;
    if (sac::ensure_stage(b, sizeof(float) * (size_t)n)) return -1;
    hipStream_t s = b->stream;
    SAC_HIP(hipMemcpy2DAsync(obs, sizeof(float) * O, b->obs + start * b->Ost, sizeof(float) * b->Ost, sizeof(float) * O,
                             (size_t)n, hipMemcpyDeviceToHost, s));
    SAC_HIP(hipMemcpy2DAsync(next_obs, sizeof(float) * O, b->nobs + start * b->Ost, sizeof(float) * b->Ost,
                             sizeof(float) * O, (size_t)n, hipMemcpyDeviceToHost, s));
    SAC_HIP(hipMemcpy2DAsync(act, sizeof(float) * A, b->act + start * b->Ast, sizeof(float) * b->Ast, sizeof(float) * A,
                             (size_t)n, hipMemcpyDeviceToHost, s));
    SAC_HIP(hipMemcpyAsync(rew, b->rew + start, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, s));
    SAC_HIP(hipMemcpyAsync(b->h_stage, b->term + start, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, s));
    SAC_HIP(hipStreamSynchronize(s));
    const float *tf = static_cast<const float *>(b->h_stage);
    for (int64_t i = 0; i < n; ++i) term[i] = tf[i] != 0.f ? 1 : 0;
    return 0;
}

// Restore the ring cursor after re-inserting saved rows in storage order.
int sac_buffer_set_cursor(sac_buffer_t *b, int64_t top, int64_t size) {
    SAC_REQUIRE(b != nullptr, "null buffer");
    SAC_REQUIRE(top >= 0 && top < b->capacity && size >= 0 && size <= b->capacity, "bad cursor (top %lld, size %lld)",
                (long long)top, (long long)size);
    SAC_HIP(hipSetDevice(b->device));
    if (readahead_rollback(b)) return -1;
    if (size < b->size) {      // pad entries of the index buffers must stay valid rows: back to row 0
        SAC_HIP(hipSetDevice(b->device));
        if (b->d_idx) SAC_HIP(hipMemsetAsync(b->d_idx, 0, sizeof(int64_t) * b->idx_cap, b->stream));
        if (b->d_ring_idx)
            SAC_HIP(hipMemsetAsync(b->d_ring_idx, 0, sizeof(int64_t) * (size_t)b->ring_layout.B * sac_buffer::NRING, b->stream));
    }
    b->top = top; b->size = size;
    return 0;
}

static int adopt_state(sac_buffer *b, const MtState &s) { return sac::host_rng_adopt(b, s); }

int sac_rng_seed(sac_buffer_t *b, uint32_t seed) {
    SAC_REQUIRE(b != nullptr, "null buffer");
    SAC_HIP(hipSetDevice(b->device));
    if (readahead_rollback(b)) return -1;
    MtState s;
    memset(&s, 0, sizeof(s));
    s.mt[0] = seed;
    for (int i = 1; i < MT_N; ++i) s.mt[i] = 1812433253u * (s.mt[i - 1] ^ (s.mt[i - 1] >> 30)) + (uint32_t)i;
    s.pos = MT_N;
    return adopt_state(b, s);
}

int sac_rng_bind_host(sac_buffer_t *b, uint32_t *key, int32_t *pos) {
    SAC_REQUIRE(b != nullptr && (key == nullptr) == (pos == nullptr), "bad arguments to sac_rng_bind_host");
    SAC_HIP(hipSetDevice(b->device));
    if (host_rng_sync_in(b)) return -1;          // (a pending outside change of the old binding is adopted first)
    if (readahead_rollback(b)) return -1;
    if (!key) { b->host_key = nullptr; b->host_pos = nullptr; return 0; }
    SAC_REQUIRE(*pos >= 0 && *pos <= MT_N, "host generator holds an invalid position %d", (int)*pos);
    b->host_key = key; b->host_pos = pos;
    memcpy(b->host_seen.mt, key, sizeof(uint32_t) * MT_N);
    b->host_seen.pos = *pos;
    return upload_rng(b, b->host_seen);
}

int sac_rng_get_state(sac_buffer_t *b, uint32_t key[624], int32_t *pos) {
    SAC_REQUIRE(b && key && pos, "bad arguments to sac_rng_get_state");
    SAC_HIP(hipSetDevice(b->device));
    if (b->host_key) {       // bound: the host words ARE the state behind the batches handed out (no device round trip,
        if (host_rng_sync_in(b)) return -1;      // and the read-ahead stays)
        memcpy(key, b->host_seen.mt, sizeof(uint32_t) * MT_N);
        *pos = b->host_seen.pos;
        return 0;
    }
    // private stream: the DEVICE generator's own words (behind the batches the caller was GIVEN: the speculation goes
    // back first) -- and they must be what the host mirror says they are
    if (readahead_rollback(b)) return -1;
    MtState s;
    SAC_HIP(hipMemcpyAsync(&s, b->d_rng, sizeof(s), hipMemcpyDeviceToHost, b->stream));
    SAC_HIP(hipStreamSynchronize(b->stream));
    SAC_REQUIRE(s.pos == b->host_seen.pos && memcmp(s.mt, b->host_seen.mt, sizeof(uint32_t) * MT_N) == 0,
                "internal: the device generator (pos %d) and its host mirror (pos %d) have diverged", (int)s.pos,
                (int)b->host_seen.pos);
    memcpy(key, s.mt, sizeof(uint32_t) * MT_N);
    *pos = s.pos;
    return 0;
}

int sac_rng_set_state(sac_buffer_t *b, const uint32_t key[624], int32_t pos) {
    SAC_REQUIRE(b && key && pos >= 0 && pos <= MT_N, "bad arguments to sac_rng_set_state");
    SAC_HIP(hipSetDevice(b->device));
    if (readahead_rollback(b)) return -1;
    MtState s;
    memset(&s, 0, sizeof(s));
    memcpy(s.mt, key, sizeof(uint32_t) * MT_N);
    s.pos = pos;
    return adopt_state(b, s);
}

int sac_sample_indices(sac_buffer_t *b, int batch, int64_t n_batches, int64_t *idx_out) {
    SAC_REQUIRE(b && batch > 0 && n_batches > 0, "bad arguments to sac_sample_indices");
    SAC_HIP(hipSetDevice(b->device));
    if (launch_sample(b, batch, n_batches)) return -1;
    if (idx_out)        // (device layout: stride round_up(batch, 16) per batch)
        SAC_HIP(hipMemcpy2DAsync(idx_out, sizeof(int64_t) * batch, b->d_idx, sizeof(int64_t) * round_up(batch, RB),
                                 sizeof(int64_t) * batch, (size_t)n_batches, hipMemcpyDeviceToHost, b->stream));
    SAC_HIP(hipStreamSynchronize(b->stream));
    return 0;
}

static int copy_slot_out(sac_buffer *b, int64_t slot, float *obs, float *act, float *rew, float *term,
                         float *nobs) {
    const SlotLayout &L = b->slot;
    const float *S = b->d_slots + slot * L.slot_floats;
    const int B = L.Bt;                 // the rows that were asked for (the slot may hold row-block padding behind them)
    if (obs) SAC_HIP(hipMemcpyAsync(obs, S + L.off_obs, sizeof(float) * B * L.O, hipMemcpyDeviceToHost, b->stream));
    if (act) SAC_HIP(hipMemcpyAsync(act, S + L.off_act, sizeof(float) * B * L.A, hipMemcpyDeviceToHost, b->stream));
    if (rew) SAC_HIP(hipMemcpyAsync(rew, S + L.off_rew, sizeof(float) * B, hipMemcpyDeviceToHost, b->stream));
    if (term) SAC_HIP(hipMemcpyAsync(term, S + L.off_term, sizeof(float) * B, hipMemcpyDeviceToHost, b->stream));
    if (nobs) SAC_HIP(hipMemcpyAsync(nobs, S + L.off_nobs, sizeof(float) * B * L.O, hipMemcpyDeviceToHost, b->stream));
    return 0;
}

int sac_random_batch(sac_buffer_t *b, int batch, float *obs, float *act, float *rew, float *term, float *next_obs,
                     int64_t *idx_out) {
    SAC_REQUIRE(b && batch > 0, "bad arguments to sac_random_batch");
    SAC_HIP(hipSetDevice(b->device));
    if (ensure_slots(b, batch, 1)) return -1;          // (first: a layout change clears the index buffer too)
    if (launch_sample(b, batch, 1)) return -1;
    if (launch_gather(b, b->d_idx, b->slot.B, 1, b->d_slots, b->slot, 1)) return -1;
    if (copy_slot_out(b, 0, obs, act, rew, term, next_obs)) return -1;
    if (idx_out) SAC_HIP(hipMemcpyAsync(idx_out, b->d_idx, sizeof(int64_t) * batch, hipMemcpyDeviceToHost, b->stream));
    SAC_HIP(hipStreamSynchronize(b->stream));
    return 0;
}

// ---- device-resident batches (the stepwise interface without a PCIe round trip per step) ----------------
static int ensure_ring(sac_buffer *b, int batch) {
    if (b->d_ring && b->ring_layout.Bt == batch) return 0;
    if (readahead_rollback(b)) return -1;
    SAC_HIP(hipStreamSynchronize(b->stream));
    forget_dead_step_stream(b);
    if (b->step_stream) SAC_HIP(hipStreamSynchronize(b->step_stream));       // (steps still reading the old ring)
    if (b->d_ring) { SAC_HIP(hipFree(b->d_ring)); SAC_HIP(hipFree(b->d_ring_idx)); b->d_ring = nullptr; b->d_ring_idx = nullptr; }
    b->ring_layout = make_slot_layout(batch, b->O, b->A);
    const size_t nfl = (size_t)b->ring_layout.slot_floats * sac_buffer::NRING;
    SAC_HIP(hipMalloc(&b->d_ring, sizeof(float) * nfl));
    SAC_HIP(hipMemsetAsync(b->d_ring, 0, sizeof(float) * nfl, b->stream));    // saT padding rows stay 0
    SAC_HIP(hipMalloc(&b->d_ring_idx, sizeof(int64_t) * (size_t)b->ring_layout.B * sac_buffer::NRING));
    SAC_HIP(hipMemsetAsync(b->d_ring_idx, 0, sizeof(int64_t) * (size_t)b->ring_layout.B * sac_buffer::NRING, b->stream));
    if (!b->d_rng_saved) SAC_HIP(hipMalloc(&b->d_rng_saved, sizeof(MtState)));
    if (b->ra_scratch_cap < (int64_t)b->ring_layout.B * sac_buffer::RA_MAX) {
        if (b->d_ra_scratch) SAC_HIP(hipFree(b->d_ra_scratch));
        b->ra_scratch_cap = (int64_t)b->ring_layout.B * sac_buffer::RA_MAX;
        SAC_HIP(hipMalloc(&b->d_ra_scratch, sizeof(int64_t) * b->ra_scratch_cap));
    }
    if (const char *e = getenv("SAC_READAHEAD")) b->ra_enabled = atoi(e) != 0;
    b->waited_chunk_token = -1; b->waited_stream = nullptr;
    b->step_stream = nullptr; b->multi_stream = false;       // (a fresh ring: no step of any trainer is in flight on it)
    for (int i = 0; i < sac_buffer::NRING; ++i) {
        b->ring_token[i] = -1; b->ring_in_use[i] = false;
        b->ring_first[i] = i; b->ring_chunk_token[i] = -1;
        if (!b->ring_ready[i]) SAC_HIP(hipEventCreateWithFlags(&b->ring_ready[i], hipEventDisableTiming));
        if (!b->ring_free[i]) SAC_HIP(hipEventCreateWithFlags(&b->ring_free[i], hipEventDisableTiming));
    }
    for (int i = 0; i < 4; ++i) {
        if (!b->free4[i]) SAC_HIP(hipEventCreateWithFlags(&b->free4[i], hipEventDisableTiming));
        b->free4_seq[i] = -1;
    }
    b->free_waited_seq = b->step_seq - 1;
    return 0;
}

// the slot's previous batch may still be read by a step in flight on a trainer's stream: the buffer's stream waits
static int wait_slot_free(sac_buffer *b, int slot) {
    if (!b->ring_in_use[slot]) return 0;
    forget_dead_step_stream(b);
    if (!b->ring_in_use[slot]) return 0;
    if (b->multi_stream) {
        SAC_HIP(hipStreamWaitEvent(b->stream, b->ring_free[slot], 0));
    } else if (b->free_waited_seq < b->slot_seq[slot]) {
        const int64_t m = b->slot_seq[slot] | 15;
        const int idx = (int)((m >> 4) & 3);
        if (b->free4_seq[idx] == m) {                   // the usual case: recorded ~48 steps ago, long fired
            SAC_HIP(hipStreamWaitEvent(b->stream, b->free4[idx], 0));
            b->free_waited_seq = m;
        } else {                                        // no event at or behind that step yet: record one now
            SAC_HIP(hipEventRecord(b->ring_free[slot], b->step_stream));
            SAC_HIP(hipStreamWaitEvent(b->stream, b->ring_free[slot], 0));
            b->free_waited_seq = b->step_seq - 1;
        }
    }
    b->ring_in_use[slot] = false;
    return 0;
}

int sac_random_batch_device(sac_buffer_t *b, int batch, int64_t *token) {
    SAC_REQUIRE(b && batch > 0 && token, "bad arguments to sac_random_batch_device");
    SAC_HIP(hipSetDevice(b->device));
    if (ensure_ring(b, batch)) return -1;
    SAC_REQUIRE(b->size > 0, "random_batch on an empty replay buffer");
    if (host_rng_sync_in(b)) return -1;      // (bound generator: somebody else moved np.random -> the speculation is void)
    b->loop_streak = 0;
    if (loop_spec_drop(b)) return -1;        // (a loop call's speculative next chunk: this draw comes first)
    const int64_t n = b->ring_next;
    const int slot = (int)(n % sac_buffer::NRING);
    if (b->ra_ahead > 0) {                   // drawn and gathered ahead by an earlier call: hand it out
        b->ra_ahead -= 1;
        b->ra_streak += 1;
        b->ring_next = n + 1;
        *token = n;
        host_rng_advance(b, batch, 1);
        return 0;
    }
    // how many batches this call draws: 1 until the caller has asked twice in a row with nothing in between, then 2, 4, 8, 16
    // (never across the end of the ring: the chunk's slots and indices are contiguous)
    int k = 1;
    if (b->ra_enabled && !b->multi_stream) {
        k = b->ra_streak >= 16 ? 16 : (b->ra_streak >= 8 ? 8 : (b->ra_streak >= 4 ? 4 : (b->ra_streak >= 2 ? 2 : 1)));
        if (k > sac_buffer::RA_MAX) k = sac_buffer::RA_MAX;
        if (k > sac_buffer::NRING - slot) k = sac_buffer::NRING - slot;
    }
    for (int j = 0; j < k; ++j)
        if (wait_slot_free(b, slot + j)) return -1;
    int64_t *didx = b->d_ring_idx + (size_t)slot * b->ring_layout.B;
    if (k > 1) SAC_HIP(hipMemcpyAsync(b->d_rng_saved, b->d_rng, sizeof(MtState), hipMemcpyDeviceToDevice, b->stream));
    b->ra_internal = true;
    const int rc = launch_sample(b, batch, k, 0, didx);
    b->ra_internal = false;
    if (rc) return -1;
    if (launch_gather(b, didx, b->ring_layout.B, k, b->d_ring + (size_t)slot * b->ring_layout.slot_floats, b->ring_layout, 1)) return -1;
    SAC_HIP(hipEventRecord(b->ring_ready[slot], b->stream));
    for (int j = 0; j < k; ++j) {
        b->ring_token[slot + j] = n + j;
        b->ring_first[slot + j] = slot;
        b->ring_chunk_token[slot + j] = n;
    }
    b->ra_chunk = k; b->ra_batch = batch; b->ra_ahead = k - 1;
    b->ra_streak += 1;
    b->ring_next = n + 1;
    *token = n;
    host_rng_advance(b, batch, 1);           // (the batch HANDED OUT: the host words never run ahead of the caller)
    return 0;
}

// slot of a live token, or -1 (error set)
int sac_ring_slot_of(sac_buffer_t *b, int64_t token) {
    const int slot = (int)(token % sac_buffer::NRING);
    if (!b || token < 0 || !b->d_ring || b->ring_token[slot] != token) {
        sac::set_error("device batch %lld has expired: a batch stays valid until %d more have been drawn",
                       (long long)token, sac_buffer::NRING);
        return -1;
    }
    return slot;
}

int sac_read_batch_device(sac_buffer_t *b, int64_t token, float *obs, float *act, float *rew, float *term,
                          float *next_obs, int64_t *idx_out) {
    SAC_REQUIRE(b != nullptr, "null buffer");
    const int slot = sac_ring_slot_of(b, token);
    if (slot < 0) return -1;
    SAC_HIP(hipSetDevice(b->device));
    const SlotLayout &L = b->ring_layout;
    const float *S = b->d_ring + (size_t)slot * L.slot_floats;
    const int B = L.Bt;
    hipStream_t s = b->stream;          // in order behind the gather that filled the slot
    if (obs) SAC_HIP(hipMemcpyAsync(obs, S + L.off_obs, sizeof(float) * B * L.O, hipMemcpyDeviceToHost, s));
    if (act) SAC_HIP(hipMemcpyAsync(act, S + L.off_act, sizeof(float) * B * L.A, hipMemcpyDeviceToHost, s));
    if (rew) SAC_HIP(hipMemcpyAsync(rew, S + L.off_rew, sizeof(float) * B, hipMemcpyDeviceToHost, s));
    if (term) SAC_HIP(hipMemcpyAsync(term, S + L.off_term, sizeof(float) * B, hipMemcpyDeviceToHost, s));
    if (next_obs) SAC_HIP(hipMemcpyAsync(next_obs, S + L.off_nobs, sizeof(float) * B * L.O, hipMemcpyDeviceToHost, s));
    if (idx_out) SAC_HIP(hipMemcpyAsync(idx_out, b->d_ring_idx + (size_t)slot * L.B, sizeof(int64_t) * B, hipMemcpyDeviceToHost, s));
    SAC_HIP(hipStreamSynchronize(s));
    return 0;
}

int sac_gather(sac_buffer_t *b, const int64_t *idx, int batch, float *obs, float *act, float *rew, float *term,
               float *next_obs) {
    SAC_REQUIRE(b && idx && batch > 0, "bad arguments to sac_gather");
    SAC_HIP(hipSetDevice(b->device));
    if (readahead_rollback(b)) return -1;        // (this call reuses the gather slots: nothing speculative may sit in them)
    for (int i = 0; i < batch; ++i)
        SAC_REQUIRE(idx[i] >= 0 && idx[i] < b->size, "index %lld out of range [0, %lld)", (long long)idx[i],
                    (long long)b->size);
    if (ensure_slots(b, batch, 1)) return -1;
    if (ensure_idx(b, b->slot.B)) return -1;
    SAC_HIP(hipMemsetAsync(b->d_idx, 0, sizeof(int64_t) * b->slot.B, b->stream));
    SAC_HIP(hipMemcpyAsync(b->d_idx, idx, sizeof(int64_t) * batch, hipMemcpyHostToDevice, b->stream));
    if (launch_gather(b, b->d_idx, b->slot.B, 1, b->d_slots, b->slot, 1)) return -1;
    if (copy_slot_out(b, 0, obs, act, rew, term, next_obs)) return -1;
    SAC_HIP(hipStreamSynchronize(b->stream));
    return 0;
}

int sac_sample_gather_device(sac_buffer_t *b, int batch, int64_t n_batches, float kernel_ms[2]) {
    SAC_REQUIRE(b && batch > 0 && n_batches > 0, "bad arguments to sac_sample_gather_device");
    SAC_HIP(hipSetDevice(b->device));
    if (ensure_slots(b, batch, n_batches)) return -1;
    SAC_HIP(hipEventRecord(b->ev[0], b->stream));
    if (launch_sample(b, batch, n_batches)) return -1;
    SAC_HIP(hipEventRecord(b->ev[1], b->stream));
    if (launch_gather(b, b->d_idx, b->slot.B, n_batches, b->d_slots, b->slot, 1)) return -1;
    SAC_HIP(hipEventRecord(b->ev[2], b->stream));
    SAC_HIP(hipStreamSynchronize(b->stream));
    if (kernel_ms) {
        SAC_HIP(hipEventElapsedTime(&kernel_ms[0], b->ev[0], b->ev[1]));
        SAC_HIP(hipEventElapsedTime(&kernel_ms[1], b->ev[1], b->ev[2]));
    }
    return 0;
}

int sac_read_slot(sac_buffer_t *b, int64_t slot, float *obs, float *act, float *rew, float *term, float *next_obs,
                  int64_t *idx_out) {
    SAC_REQUIRE(b && slot >= 0 && slot < b->n_slots, "slot %lld out of range", (long long)slot);
    SAC_HIP(hipSetDevice(b->device));
    if (copy_slot_out(b, slot, obs, act, rew, term, next_obs)) return -1;
    if (idx_out)
        SAC_HIP(hipMemcpyAsync(idx_out, b->d_idx + slot * b->slot.B, sizeof(int64_t) * b->slot.Bt,
                               hipMemcpyDeviceToHost, b->stream));
    SAC_HIP(hipStreamSynchronize(b->stream));
    return 0;
}

}  // extern "C"
