// Shared host/device definitions for libsac_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <cstdio>
#include <cstdarg>

#include "../../include/sac_hip.h"

namespace sac {

void set_error(const char *fmt, ...);

#define SAC_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            sac::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return -1;                                                                        \
        }                                                                                     \
    } while (0)

#define SAC_REQUIRE(cond, ...)                                                                \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            sac::set_error(__VA_ARGS__);                                                      \
            return -2;                                                                        \
        }                                                                                     \
    } while (0)

constexpr int RB = 16;            // rows of one row-block (one MFMA 16x16x4 M tile)
constexpr int MT_N = 624;
constexpr int MT_M = 397;

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline int64_t round_up64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// Weight matrices AND the feature-major activations ([feature][batch]: rows = features, k = batch row) live in HBM in
// FRAGMENT-MAJOR order: the 16x16 block (row tile, k-chunk) of a [rows][ld] matrix is
// 1 KB contiguous, in MFMA operand order -- lane l = (row & 15) + 16 * ((k & 15) >> 2) holds the four floats k & 3.
// One wave-instruction of the weight stream (a 16x16 fragment, 16 B per lane) then reads 8 whole 128-B lines instead
// of 16 half lines: a kernel's first pass over freshly written weights (cold L2, every step) is bound by the number
// of cache lines per instruction -- scratch/vmem_wall.hip: 66 GB/s per CU for 1 KB-contiguous instructions against
// 33 GB/s for 16 rows x 64 B.
static inline __host__ __device__ size_t frag_off(int row, int k, int ld) {
    return ((size_t)(row >> 4) * (ld >> 4) + (k >> 4)) * 256 + (size_t)(((row & 15) + 16 * ((k & 15) >> 2)) * 4 + (k & 3));
}

// Layout of one minibatch slot in HBM (floats).  Written by the gather kernel, read by the step
// kernels.  Row-major part == what random_batch returns; saT is the feature-major copy
// [KQ][B] of the Q-net input that the weight-gradient kernel contracts over the batch.  The Q-net input is laid
// out [obs | 0-pad to KA | act | 0-pad to KQ] with KA = round_up(O, 16): the action sits in a k-chunk (16 columns)
// of its own, so a first-layer GEMM can be split into its observation part and its action part.
struct SlotLayout {
    int B, O, A;                  // B: rows of the slot = the batch size rounded up to a multiple of 16 (row-blocks)
    int Bt;                       // rows that hold a sampled transition (the batch size asked for); the rest is padding
                                  // (it repeats buffer row 0 and carries zero weight in every mean of the step)
    int KA;                       // first action row of saT / action column of the Q-net input: round_up(O, 16)
    int KQ;                       // KA + 16
    int KQ64;                     // rows allocated for saT: round_up(KQ, 64) (64-wide k strips)
    int64_t off_obs, off_act, off_rew, off_term, off_nobs, off_saT;
    int64_t slot_floats;          // multiple of 64 floats (256 B)
};

static inline SlotLayout make_slot_layout(int batch, int O, int A) {
    SlotLayout L;
    const int B = round_up(batch, RB);
    L.B = B; L.Bt = batch; L.O = O; L.A = A;
    L.KA = round_up(O, 16);
    L.KQ = L.KA + 16;
    L.KQ64 = round_up(L.KQ, 64);
    int64_t off = 0;
    L.off_obs = off;  off += round_up64((int64_t)B * O, 64);
    L.off_act = off;  off += round_up64((int64_t)B * A, 64);
    L.off_rew = off;  off += round_up64(B, 64);
    L.off_term = off; off += round_up64(B, 64);
    L.off_nobs = off; off += round_up64((int64_t)B * O, 64);
    L.off_saT = off;  off += (int64_t)L.KQ64 * B;
    L.slot_floats = round_up64(off, 64);
    return L;
}

struct MtState {
    uint32_t mt[MT_N];
    int32_t pos;
    int32_t pad[3];
};

}  // namespace sac

// device-side view of the replay storage
struct ReplayView {
    const float *obs, *act, *rew, *term, *nobs;
    int O, A, Ost, Ast;           // row strides (floats), multiples of 4
    int64_t capacity;
};

struct sac_buffer {
    int device = 0;
    hipStream_t stream = nullptr;
    int64_t capacity = 0, size = 0, top = 0;
    int O = 0, A = 0, Ost = 0, Ast = 0;
    float *obs = nullptr, *act = nullptr, *rew = nullptr, *term = nullptr, *nobs = nullptr;
    sac::MtState *d_rng = nullptr;
    // sampled indices + gathered slots of the last batched draw
    int64_t *d_idx = nullptr; int64_t idx_cap = 0;
    float *d_slots = nullptr; int64_t slots_cap = 0;      // floats
    sac::SlotLayout slot{};
    int64_t n_slots = 0;
    // pinned staging (read-out); double-buffered pinned staging of the asynchronous ingest (sac_buffer_add*)
    void *h_stage = nullptr; size_t stage_bytes = 0;
    static constexpr int64_t INGEST_ROWS = 8192;     // rows per staging buffer
    void *h_ing[2] = {nullptr, nullptr};
    hipEvent_t ing_free[2] = {nullptr, nullptr};     // recorded behind the copies that read staging buffer k
    bool ing_busy[2] = {false, false};
    uint64_t ing_next = 0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // device-resident batches of the stepwise interface (sac_random_batch_device): a ring of NRING slots; batch
    // number n lives in slot n % NRING until batch n + NRING is drawn
    static constexpr int NRING = 64;                 // (a token stays valid until at least NRING - RA_MAX = 48 more batches have been drawn)
    float *d_ring = nullptr; int64_t *d_ring_idx = nullptr;
    sac::SlotLayout ring_layout{};
    int64_t ring_next = 0;                           // number of batches drawn so far
    int64_t ring_token[NRING];                       // batch number held by each slot (-1: none)
    hipEvent_t ring_ready[NRING] = {}, ring_free[NRING] = {};
    bool ring_in_use[NRING] = {};                    // a step was launched on the slot and nobody has waited for it since
    // "the trainer is done with this slot" is signalled by ONE event per sixteen steps (two runtime calls per step was a
    // fifth of the stepwise interface's host time): step k (launch sequence) leaves its number in the slot, every sixteenth
    // step records free4[(k >> 4) & 3] (four events cover the ring of 64); a draw that reuses a slot waits for the oldest event recorded at or behind the
    // slot's step (the per-slot event ring_free[] is the fall-back when no such event exists)
    hipEvent_t free4[4] = {nullptr, nullptr, nullptr, nullptr};
    int64_t free4_seq[4] = {-1, -1, -1, -1};
    int64_t step_seq = 0, free_waited_seq = -1;
    int64_t slot_seq[NRING] = {};
    // Read-ahead of the stepwise interface: when random_batch_device is called again and again with nothing in between that
    // touches the generator or the buffer (the reference's loop: rlkit_custom.py:234-238), the call draws and gathers the
    // next RA batches too, in the same two launches, and the following calls just hand them out.  Anything that reads or
    // changes the generator, the buffer's rows or its size first ROLLS the speculation BACK: the generator's state saved
    // in front of the chunk is restored and advanced by the batches that were handed out (one small draw), so the index
    // stream stays NumPy's, bit for bit, whatever the caller interleaves.
    static constexpr int RA_MAX = 16;
    int64_t ra_ahead = 0;                            // batches drawn + gathered beyond ring_next
    int ra_chunk = 0, ra_batch = 0;                  // size of the chunk they belong to; the batch size it was drawn for
    int ra_streak = 0;                               // consecutive random_batch_device calls with nothing in between
    bool ra_internal = false;                        // (a draw of the read-ahead itself is in progress)
    bool ra_enabled = true;
    sac::MtState *d_rng_saved = nullptr;
    int64_t *d_ra_scratch = nullptr; int64_t ra_scratch_cap = 0;
    int ring_first[NRING] = {};                      // slot whose ring_ready event covers this slot (first of its chunk)
    int64_t ring_chunk_token[NRING] = {};            // token of that first batch
    hipStream_t waited_stream = nullptr;             // the step stream has already waited for chunks up to waited_chunk_token
    int64_t waited_chunk_token = -1;
    hipStream_t step_stream = nullptr;               // the stream the steps are launched on (one trainer per buffer; a second
    bool multi_stream = false;                       // stream switches to one event per step)
    // host_seen: the generator's state behind the batches HANDED OUT, mirrored on the host -- every draw on the device is
    // repeated by the same sequential draw here (~5 ns per index).  It is what a fused step that gave up is replayed from
    // (sac_train_loop), and what binds the generator to a HOST-resident MT19937 state (sac_rng_bind_host: np.random's own
    // state struct): the new state is written straight into the caller's words, so np.random is right after every
    // random_batch without a device round trip -- and a change made by anybody else (np.random.seed, a host consumer) is
    // noticed by comparing those words with host_seen in front of the next draw.
    uint32_t *host_key = nullptr; int32_t *host_pos = nullptr;
    sac::MtState host_seen{};
    bool defer_mirror = false;                       // sac_train_loop: the mirror follows once all launches are queued
    int64_t deferred_batches = 0;
    // Loop calls that follow each other with nothing in between that touches the generator, the rows or the slots (a
    // driver that trains in slices; the bracket of a benchmark): behind its last draw a call draws and gathers the NEXT
    // call's first chunk into the slots behind its own (on this buffer's stream, under its last steps), so that the next
    // call's first step has no draw and no gather in front of it (~15 us of a call).  Speculation: the host mirror stays at
    // the batches handed out; anything else that comes first takes it back (loop_spec_drop: the device generator is
    // re-uploaded from the mirror).
    static constexpr int LOOP_SPEC = 4;              // batches of a speculative first chunk (= the loop plan's first chunk)
    bool spec_valid = false;
    int spec_batch = 0;                              // the batch size it was drawn for
    int64_t spec_size = 0, spec_pos = 0;             // buffer size at the draw; first slot of the chunk in the loop's slot ring
    hipEvent_t spec_ev = nullptr;                    // behind its gather (buffer's stream)
    int64_t loop_pos = 0;                            // where the next loop call starts in the slot ring
    int loop_streak = 0;                             // loop calls in a row without an outside touch
    bool in_loop = false;                            // (a loop call is submitting: its own draws are not outside touches)
    ReplayView view() const { return ReplayView{obs, act, rew, term, nobs, O, A, Ost, Ast, capacity}; }
};

// slot of a live device batch (sac_random_batch_device token), or -1 with the error set (internal)
extern "C" int sac_ring_slot_of(sac_buffer *b, int64_t token);
extern "C" int sac_make_xcd_stream(hipStream_t *out, int xcd);
extern "C" int sac_make_xcd_mask_stream(hipStream_t *out, unsigned xcd_mask);

namespace sac {
int ensure_stage(sac_buffer *b, size_t bytes);
int ensure_idx(sac_buffer *b, int64_t n);
int ensure_slots(sac_buffer *b, int B, int64_t n_slots);
// launches on `on` (null: b->stream); indices stay in b->d_idx, slots in b->d_slots
int launch_sample(sac_buffer *b, int batch, int64_t n_batches, int64_t idx_offset = 0, int64_t *dst = nullptr,
                  hipStream_t on = nullptr);
int launch_gather(sac_buffer *b, const int64_t *d_idx, int batch, int64_t n_batches, float *d_slots,
                  const SlotLayout &L, int write_saT, hipStream_t on = nullptr);
// undo the stepwise interface's read-ahead (see sac_buffer::ra_ahead); to be called in front of anything that reads or
// changes the generator's state, the buffer's rows or its size
int readahead_rollback(sac_buffer *b);
int loop_spec_drop(sac_buffer *b);                   // take a speculative first chunk back (see sac_buffer::spec_valid)
// bound generator (sac_rng_bind_host): adopt a host state somebody else changed (in front of a draw) / mirror the draws
// of n_batches x batch indices on the host state (behind it)
int host_rng_sync_in(sac_buffer *b);
void host_rng_advance(sac_buffer *b, int batch, int64_t n_batches);
void host_rng_skip(const sac_buffer *b, MtState &st, int batch, int64_t n_batches);
int host_rng_adopt(sac_buffer *b, const MtState &s);      // a new generator state: mirror, bound host words, device
// streams of live trainers (a buffer remembers the stream its steps run on: it must not outlive the trainer there)
void stream_register(hipStream_t s);
void stream_unregister(hipStream_t s);
bool stream_is_live(hipStream_t s);
void forget_dead_step_stream(sac_buffer *b);
}  // namespace sac
