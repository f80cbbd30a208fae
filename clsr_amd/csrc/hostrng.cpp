// Host-side replay of CPython's `random` module (MT19937) for the input pipeline (no GPU work here).
//
// The reference iterator shuffles the cached list of parsed train lines with random.shuffle on every pass and draws
// the in-batch negatives with random.randint (io/sequential_iterator.py:249-261, 612-634).  To keep the batch
// stream bit-identical while leaving the interpreter out of a 1M-element shuffle, these entry points continue the
// generator from a state exported by random.getstate() and hand the advanced state back for random.setstate().
//   getrandbits(k <= 32) == genrand_uint32() >> (32 - k);  _randbelow(n): k = n.bit_length(), redraw while r >= n;
//   shuffle(x): for i = len-1 .. 1: j = _randbelow(i + 1); swap(x[i], x[j])        (CPython 3.10 Lib/random.py)
#include "common.h"
#include "clsr_hip.h"

namespace {
struct MT {
  uint32_t* mt;
  int idx;
  inline uint32_t next() {
    if (idx >= 624) {
      static const uint32_t mag01[2] = {0u, 0x9908b0dfu};
      int kk = 0;
      uint32_t y;
      for (; kk < 624 - 397; ++kk) {
        y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
        mt[kk] = mt[kk + 397] ^ (y >> 1) ^ mag01[y & 1u];
      }
      for (; kk < 623; ++kk) {
        y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
        mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ mag01[y & 1u];
      }
      y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
      mt[623] = mt[396] ^ (y >> 1) ^ mag01[y & 1u];
      idx = 0;
    }
    uint32_t y = mt[idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
  inline uint32_t below(uint32_t n) {  // random._randbelow_with_getrandbits, n >= 1
    int k = 0;
    for (uint32_t v = n; v; v >>= 1) ++k;
    uint32_t r = next() >> (32 - k);
    while (r >= n) r = next() >> (32 - k);
    return r;
  }
};
}  // namespace

// perm (n entries) is shuffled in place exactly like random.shuffle(list) would; key[624] / *pos = MT state.
extern "C" int clsr_host_mt_shuffle(unsigned* key, int* pos, long* perm, long n) {
  CLSR_CHECK_ARG(key && pos && perm && n >= 0 && *pos >= 0 && *pos <= 624);
  CLSR_CHECK_SUPPORTED(n < (1L << 31));
  MT g{key, *pos};
  for (long i = n - 1; i >= 1; --i) {
    const long j = g.below((uint32_t)(i + 1));
    const long t = perm[i];
    perm[i] = perm[j];
    perm[j] = t;
  }
  *pos = g.idx;
  return CLSR_OK;
}

// In-batch negative sampling of one training batch (sequential_iterator.py:612-634): for every line i, ngs draws
// j = randint(0, n-1), redrawn while items[j] == items[i]; src[i*(ngs+1)] = i, then the accepted j's.
extern "C" int clsr_host_mt_sample_negatives(unsigned* key, int* pos, const long* items, long n, int ngs,
                                             long* src) {
  CLSR_CHECK_ARG(key && pos && items && src && n >= 2 && ngs >= 1 && *pos >= 0 && *pos <= 624);
  CLSR_CHECK_SUPPORTED(n < (1L << 31));
  MT g{key, *pos};
  for (long i = 0; i < n; ++i) {
    long* row = src + i * (ngs + 1);
    row[0] = i;
    const long own = items[i];
    int count = 0;
    long guard = 0;
    while (count < ngs) {
      const long j = g.below((uint32_t)n);
      if (items[j] == own) {
        if (++guard > (1L << 24)) {   // a batch whose items are all equal would never terminate in the reference
          clsr_set_error("clsr_host_mt_sample_negatives: every draw equals the positive item of line %ld", i);
          return CLSR_EINVAL;
        }
        continue;
      }
      row[++count] = j;
    }
  }
  *pos = g.idx;
  return CLSR_OK;
}
