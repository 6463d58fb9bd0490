// Host-side restatement of CPython's random.sample(range(n), k) -- the call rpn_util._apply_sampling makes twice per training image
// (rpn_util.py:324-350: `random.sample(range(num_pos), num_pos - 128)`, `random.sample(range(num_neg), num_neg + num_pos - 256)`).
// In the interpreter that call draws ~num_neg Mersenne-Twister words through `_randbelow` and a Python-level pool shuffle:
// 7-80 ms per image for the 20 000-60 000 negative anchors of a 600x1000 / 600x1500 frame, twenty times the training step
// itself.  The draw MUST stay the global `random` stream's (which anchors are sampled is part of the reference's
// behaviour), so this file replays the same generator (MT19937, `genrand_uint32` of CPython's _randommodule.c: the reference
// implementation of Matsumoto & Nishimura) and the same selection algorithm (CPython 3.10 Lib/random.py: sample(), the
// `_randbelow_with_getrandbits` rejection loop, both the "pool" and the "set" branch) on the generator's OWN state: the caller passes
// `random.getstate()`'s 624 words + index and writes the advanced state back with `random.setstate()`.  Pure host code: no HIP call.
#include "common.h"
#include <stdint.h>
#include <stdlib.h>
#include <vector>

namespace {

constexpr int MT_N = 624, MT_M = 397;
constexpr uint32_t MATRIX_A = 0x9908b0dfu, UPPER_MASK = 0x80000000u, LOWER_MASK = 0x7fffffffu;

struct Mt {
    uint32_t* mt;
    int index;
    uint32_t next() {
        if (index >= MT_N) {
            static const uint32_t mag01[2] = {0u, MATRIX_A};
            int kk;
            for (kk = 0; kk < MT_N - MT_M; ++kk) {
                const uint32_t y = (mt[kk] & UPPER_MASK) | (mt[kk + 1] & LOWER_MASK);
                mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ mag01[y & 1u];
            }
            for (; kk < MT_N - 1; ++kk) {
                const uint32_t y = (mt[kk] & UPPER_MASK) | (mt[kk + 1] & LOWER_MASK);
                mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ mag01[y & 1u];
            }
            const uint32_t y = (mt[MT_N - 1] & UPPER_MASK) | (mt[0] & LOWER_MASK);
            mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ mag01[y & 1u];
            index = 0;
        }
        uint32_t y = mt[index++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    // random.Random._randbelow_with_getrandbits(n), 1 <= n < 2^31: k = n.bit_length(); r = getrandbits(k) until r < n
    uint32_t below(uint32_t n) {
        const int bits = 32 - __builtin_clz(n);
        uint32_t r = next() >> (32 - bits);
        while (r >= n) r = next() >> (32 - bits);
        return r;
    }
};

}  // namespace

extern "C" int frcnn_host_mt_sample_range(uint32_t* mt_state, int32_t* mt_index, int n, int k, int use_pool, int32_t* out) {
    if (!mt_state || !mt_index || n < 0 || k < 0 || k > n || (k > 0 && !out)) return frcnn::fail(FRCNN_E_ARG, "host_mt_sample_range: need 0 <= k <= n and the generator state");
    if (*mt_index < 0 || *mt_index > MT_N) return frcnn::fail(FRCNN_E_ARG, "host_mt_sample_range: generator index out of range");
    Mt g = {mt_state, *mt_index};
    if (use_pool) {                                        // n <= setsize: pool = list(range(n)); result[i] = pool[j]; pool[j] = pool[n - i - 1]
        std::vector<int32_t> pool((size_t)n);
        for (int i = 0; i < n; ++i) pool[i] = i;
        for (int i = 0; i < k; ++i) {
            const uint32_t j = g.below((uint32_t)(n - i));
            out[i] = pool[j];
            pool[j] = pool[n - i - 1];
        }
    } else {                                               // selected = set(); draw j = randbelow(n) until it is new
        std::vector<uint64_t> seen(((size_t)n + 63) / 64, 0);
        for (int i = 0; i < k; ++i) {
            uint32_t j = g.below((uint32_t)n);
            while (seen[j >> 6] >> (j & 63) & 1) j = g.below((uint32_t)n);
            seen[j >> 6] |= 1ull << (j & 63);
            out[i] = (int32_t)j;
        }
    }
    *mt_index = g.index;
    return FRCNN_OK;
}
