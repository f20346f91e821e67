// Host-side iBOT block-mask sampler in C++ with CPython's own random stream (the reference samples the masks of
// every step in pure Python on the training thread, LT/_methods/dinov2/utils.py:41-152 `MaskingGenerator._mask / __call__` and
// `create_collated_masks` -- 13 ms for 256 crops).  This is the same algorithm, drawing from the SAME Mersenne-Twister stream:
// the caller hands over `random.getstate()` (624 state words + position), the draws below reproduce CPython's
//   random.random()      genrand_res53: (a >> 5, b >> 6) of two 32-bit outputs            (Modules/_randommodule.c)
//   random.uniform(a,b)  a + (b - a) * random()                                            (Lib/random.py)
//   random.randint(a,b)  a + _randbelow_with_getrandbits(b - a + 1): k = n.bit_length(), r = getrandbits(k) until r < n
//   random.shuffle(x)    for i in reversed(range(1, len(x))): j = _randbelow(i + 1); swap
// bit for bit, and the advanced state goes back through `random.setstate`, so `random.seed(s)` followed by this sampler yields
// exactly the masks (and leaves exactly the stream position) the reference's Python loop would.  math.exp / math.sqrt / round()
// are libm's exp / sqrt and round-half-even (nearbyint in the default rounding mode), as in CPython.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/lt_amd.h"

void lt_set_error(const char* fmt, ...);

namespace {

struct MT {
  uint32_t* mt;
  int pos;
  uint32_t next() {
    constexpr int N = 624, M = 397;
    constexpr uint32_t MATRIX_A = 0x9908b0dfU, UPPER = 0x80000000U, LOWER = 0x7fffffffU;
    if (pos >= N) {
      int kk;
      uint32_t y;
      for (kk = 0; kk < N - M; kk++) {
        y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
        mt[kk] = mt[kk + M] ^ (y >> 1) ^ ((y & 1U) ? MATRIX_A : 0U);
      }
      for (; kk < N - 1; kk++) {
        y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
        mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ ((y & 1U) ? MATRIX_A : 0U);
      }
      y = (mt[N - 1] & UPPER) | (mt[0] & LOWER);
      mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ ((y & 1U) ? MATRIX_A : 0U);
      pos = 0;
    }
    uint32_t y = mt[pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680U;
    y ^= (y << 15) & 0xefc60000U;
    y ^= (y >> 18);
    return y;
  }
  double random() {
    const uint32_t a = next() >> 5, b = next() >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
  }
  double uniform(double a, double b) { return a + (b - a) * random(); }
  uint32_t randbelow(uint32_t n) {   // n >= 1
    int k = 0;
    for (uint32_t t = n; t; t >>= 1) ++k;          // n.bit_length()
    uint32_t r = next() >> (32 - k);
    while (r >= n) r = next() >> (32 - k);
    return r;
  }
  int randint(int a, int b) { return a + (int)randbelow((uint32_t)(b - a + 1)); }
};

// MaskingGenerator._mask (utils.py:77-101)
int try_block(MT& rng, uint8_t* grid, int H, int W, int min_patches, int budget, double la0, double la1) {
  int gained = 0;
  for (int attempt = 0; attempt < 10; ++attempt) {
    const double area = rng.uniform((double)min_patches, (double)budget);
    const double aspect = exp(rng.uniform(la0, la1));
    const int bh = (int)nearbyint(sqrt(area * aspect));
    const int bw = (int)nearbyint(sqrt(area / aspect));
    if (bw < W && bh < H) {
      const int y0 = rng.randint(0, H - bh), x0 = rng.randint(0, W - bw);
      int masked = 0;
      for (int i = y0; i < y0 + bh; ++i)
        for (int j = x0; j < x0 + bw; ++j) masked += grid[i * W + j];
      const int fresh = bh * bw - masked;
      if (0 < fresh && fresh <= budget) {
        for (int i = y0; i < y0 + bh; ++i)
          for (int j = x0; j < x0 + bw; ++j) grid[i * W + j] = 1;
        gained += fresh;
      }
    }
    if (gained > 0) break;
  }
  return gained;
}

}  // namespace

extern "C" int lt_sample_block_masks(uint32_t* mt_state, int* mt_pos, const double* ratio_edges, int n_masked_crops, int n_crops, int H, int W,
                                     int max_num_patches, int min_num_patches, double log_aspect_min, double log_aspect_max, uint8_t* masks) {
  if (!mt_state || !mt_pos || !ratio_edges || !masks || n_masked_crops < 0 || n_crops < n_masked_crops || H <= 0 || W <= 0 || *mt_pos < 0 ||
      *mt_pos > 624) {
    lt_set_error("lt_sample_block_masks: bad arguments");
    return LT_ERR_INVALID;
  }
  MT rng{mt_state, *mt_pos};
  const int P = H * W;
  std::vector<uint8_t> tmp((size_t)n_crops * P, 0);
  // create_collated_masks (utils.py:104-123): masked crops first, each with its own target count, then the unmasked ones
  for (int i = 0; i < n_masked_crops; ++i) {
    const int target = (int)((double)P * rng.uniform(ratio_edges[i], ratio_edges[i + 1]));
    uint8_t* grid = tmp.data() + (size_t)i * P;
    int done = 0;
    while (done < target) {   // MaskingGenerator.__call__ (utils.py:103-117)
      int budget = target - done;
      if (budget > max_num_patches) budget = max_num_patches;
      const int got = try_block(rng, grid, H, W, min_num_patches, budget, log_aspect_min, log_aspect_max);
      if (got == 0) break;
      done += got;
    }
  }
  // random.shuffle(masks_list)
  std::vector<int> order(n_crops);
  for (int i = 0; i < n_crops; ++i) order[i] = i;
  for (int i = n_crops - 1; i >= 1; --i) {
    const int j = (int)rng.randbelow((uint32_t)(i + 1));
    const int t = order[i]; order[i] = order[j]; order[j] = t;
  }
  for (int i = 0; i < n_crops; ++i) memcpy(masks + (size_t)i * P, tmp.data() + (size_t)order[i] * P, P);
  *mt_pos = rng.pos;
  return LT_OK;
}
