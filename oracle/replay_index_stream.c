/* Oracle (TEST INFRASTRUCTURE, never shipped): scalar C restatement of the
 * index stream behind EnvReplayBuffer.random_batch.
 *
 * Call sites it follows: /root/reference/scripts/train.py:112 (np.random.seed),
 * /root/reference/util/rlkit_custom.py:235 (random_batch) -> rlkit
 * SimpleReplayBuffer.random_batch = np.random.randint(0, size, batch_size).
 * Algorithm: SURVEY.md Appendix B (MT19937 + masked rejection on 32-bit draws).
 * Pinned against numpy.random.RandomState in tests/test_oracle_index_stream.py.
 *
 * Build:  gcc -O2 -shared -fPIC -o oracle/liboracle_index.so oracle/replay_index_stream.c
 */
#include <stdint.h>

#define MT_N 624
#define MT_M 397

typedef struct {
    uint32_t mt[MT_N];
    int32_t pos;
} oracle_mt_t;

void oracle_mt_seed(oracle_mt_t *s, uint32_t seed) {
    s->mt[0] = seed;
    for (int i = 1; i < MT_N; ++i)
        s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->pos = MT_N;
}

static void twist(oracle_mt_t *s) {
    uint32_t *mt = s->mt;
    for (int i = 0; i < MT_N; ++i) {
        uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % MT_N] & 0x7fffffffu);
        mt[i] = mt[(i + MT_M) % MT_N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    s->pos = 0;
}

uint32_t oracle_mt_next(oracle_mt_t *s) {
    if (s->pos >= MT_N) twist(s);
    uint32_t y = s->mt[s->pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* np.random.randint(0, size, count) on the legacy stream; returns draws consumed. */
int64_t oracle_randint(oracle_mt_t *s, uint64_t size, int64_t count, int64_t *out) {
    int64_t draws = 0;
    if (size == 0) return -1;
    uint32_t rng = (uint32_t)(size - 1);
    if (rng == 0) {
        for (int64_t i = 0; i < count; ++i) out[i] = 0;
        return 0;
    }
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    for (int64_t i = 0; i < count; ++i) {
        uint32_t v;
        do { v = oracle_mt_next(s) & mask; ++draws; } while (v > rng);
        out[i] = (int64_t)v;
    }
    return draws;
}
