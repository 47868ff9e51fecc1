// Host side of a batch launch: are the rows permutations of 0..p-1?  (The reference never asks -- its orderings come
// from its own samplers, ls_spa/ls_spa.py:375-456 -- but this library takes them through a C ABI, and a repeated
// index makes a permuted Gram matrix singular, an index out of range a read outside the Gram matrix.)
//
// The check is on the launch path of every batch, and at small p the host's share of a batch is what bounds the step
// (a group of 2048 orderings at p = 100: 94 us of stamping against 200 us of GPU work, round 5).  For 8 <= p <= 128 on an
// AVX2 host a row is reduced to a 128-bit set -- 1 << f for eight entries a time, OR-ed -- and is a permutation iff no
// entry is out of range and the set is full (p entries, p distinct values).  Anything else takes the stamp loop.
#include <cstddef>
#include <cstdint>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace lsspa {

static bool rows_stamped(const int32_t* perms, int B, int p, std::vector<int32_t>& mark) {
  mark.assign(p, 0);
  for (int s = 0; s < B; ++s) {
    const int32_t* row = perms + (size_t)s * p;
    const int32_t stamp = s + 1;
    for (int j = 0; j < p; ++j) {
      const uint32_t f = (uint32_t)row[j];
      if (f >= (uint32_t)p || mark[f] == stamp) return false;
      mark[f] = stamp;
    }
  }
  return true;
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) static bool rows_as_sets_avx2(const int32_t* perms, int B, int p) {
  const __m256i one = _mm256_set1_epi64x(1), c63 = _mm256_set1_epi64x(63);
  const __m256i top = _mm256_set1_epi32(p - 1);
  const uint64_t full_lo = p >= 64 ? ~0ull : ((1ull << p) - 1);
  const uint64_t full_hi = p <= 64 ? 0ull : (p == 128 ? ~0ull : ((1ull << (p - 64)) - 1));
  __m256i out_of_range = _mm256_setzero_si256();
  uint64_t incomplete = 0;
  for (int s = 0; s < B; ++s) {
    const int32_t* row = perms + (size_t)s * p;
    __m256i lo = _mm256_setzero_si256(), hi = _mm256_setzero_si256();
    for (int j0 = 0; j0 < p; j0 += 8) {
      const int j = j0 + 8 <= p ? j0 : p - 8;      // the last load overlaps the one before: a set does not mind
      const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(row + j));
      out_of_range = _mm256_or_si256(out_of_range, _mm256_xor_si256(_mm256_max_epu32(v, top), top));
      const __m256i h[2] = {_mm256_cvtepu32_epi64(_mm256_castsi256_si128(v)),
                            _mm256_cvtepu32_epi64(_mm256_extracti128_si256(v, 1))};
      for (int q = 0; q < 2; ++q) {
        const __m256i bit = _mm256_sllv_epi64(one, _mm256_and_si256(h[q], c63));
        const __m256i upper = _mm256_cmpgt_epi64(h[q], c63);      // in-range entries: 64 <= f < 128
        hi = _mm256_or_si256(hi, _mm256_and_si256(bit, upper));
        lo = _mm256_or_si256(lo, _mm256_andnot_si256(upper, bit));
      }
    }
    alignas(32) uint64_t a[4], b[4];
    _mm256_store_si256(reinterpret_cast<__m256i*>(a), lo);
    _mm256_store_si256(reinterpret_cast<__m256i*>(b), hi);
    incomplete |= ((a[0] | a[1]) | (a[2] | a[3])) ^ full_lo;
    incomplete |= ((b[0] | b[1]) | (b[2] | b[3])) ^ full_hi;
  }
  return _mm256_testz_si256(out_of_range, out_of_range) && incomplete == 0;
}
#endif

bool all_permutations(const int32_t* perms, int B, int p, std::vector<int32_t>& mark) {
#if defined(__x86_64__)
  if (p >= 8 && p <= 128 && __builtin_cpu_supports("avx2")) return rows_as_sets_avx2(perms, B, p);
#endif
  return rows_stamped(perms, B, p, mark);
}

// the stamp loop by itself: what the fast form is tested against (lsspa_debug_check_perms)
bool all_permutations_plain(const int32_t* perms, int B, int p, std::vector<int32_t>& mark) {
  return rows_stamped(perms, B, p, mark);
}

}  // namespace lsspa

