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


// ---------------------------------------------------------------------------------------------------------------
// The row argsort of the QMC samplers (ls_spa/ls_spa.py:375-397 argsort of Sobol' points, :400-456 of projected normals;
// here ls_spa/_samplers.py) on a few native threads.  numpy.argsort(points, axis=1) is 0.8 us a row of 100 on one thread
// and three quarters of what the sampler's helper thread does for a small problem, whose public call waits for
// orderings half of its time -- and a second PYTHON thread for the sorts costs the driver's thread more waits for the
// interpreter lock than it saves (measured twice, DESIGN_HISTORY.md).  Threads of this library hold no such lock.
// A row whose keys are all different has ONE argsort, so any correct sort reproduces numpy's; rows with equal keys
// (numpy's order among them is its sort's own business, and the reference's results inherit it) or a NaN are reported
// and left to numpy by the caller.  Keys spread over their range (uniform points, projected normals): one counting pass
// into 4 p buckets between the row's minimum and maximum leaves them almost in order, an insertion pass finishes --
// correct whatever the buckets did, since the last pass is a full sort.
#include <algorithm>
#include <cmath>
#include <thread>

namespace lsspa {

// false: the row has equal keys or a NaN (out is then unspecified)
static bool argsort_row(const double* key, int p, int32_t* out, std::vector<int32_t>& cnt, std::vector<int32_t>& tmp) {
  if (p == 1) {
    out[0] = 0;
    return key[0] == key[0];
  }
  double mn = key[0], mx = key[0];
  bool nan = false;
  for (int j = 0; j < p; ++j) {
    const double v = key[j];
    nan |= !(v == v);
    mn = v < mn ? v : mn;
    mx = v > mx ? v : mx;
  }
  if (nan || !(mx > mn) || !std::isfinite(mx - mn)) return false;
  const int nb = 4 * p;
  const double inv = (double)(nb - 1) / (mx - mn);
  cnt.assign((size_t)nb + 1, 0);
  tmp.resize((size_t)p);
  for (int j = 0; j < p; ++j) {
    int b = (int)((key[j] - mn) * inv);
    b = b < 0 ? 0 : (b > nb - 1 ? nb - 1 : b);
    tmp[j] = b;
    ++cnt[b + 1];
  }
  for (int b = 0; b < nb; ++b) cnt[b + 1] += cnt[b];
  for (int j = 0; j < p; ++j) out[cnt[tmp[j]]++] = j;
  for (int i = 1; i < p; ++i) {       // the insertion pass over the bucket order
    const int32_t x = out[i];
    const double kx = key[x];
    int j = i - 1;
    while (j >= 0 && key[out[j]] > kx) {
      out[j + 1] = out[j];
      --j;
    }
    out[j + 1] = x;
  }
  for (int i = 1; i < p; ++i)
    if (key[out[i]] == key[out[i - 1]]) return false;
  return true;
}

// out [B][p] = argsort of every row of keys [B][p]; redo [B] = 1 where the row is left to the caller (equal keys, NaN).
// Returns the number of such rows.  Rows are independent: the block is cut over up to `threads` threads.
int64_t argsort_rows_host(const double* keys, int64_t B, int p, int32_t* out, uint8_t* redo, int threads) {
  auto range = [&](int64_t lo, int64_t hi) {
    std::vector<int32_t> cnt, tmp;
    for (int64_t s = lo; s < hi; ++s) redo[s] = argsort_row(keys + s * p, p, out + s * p, cnt, tmp) ? 0 : 1;
  };
  int nt = threads < 1 ? 1 : (threads > 16 ? 16 : threads);
  nt = (int)std::min<int64_t>(nt, std::max<int64_t>(1, B / 64));
  if (nt == 1) {
    range(0, B);
  } else {
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(range, B * t / nt, B * (t + 1) / nt);
    range(0, B / nt);                  // the caller's thread takes the first share
    for (auto& t : th) t.join();
  }
  int64_t n = 0;
  for (int64_t s = 0; s < B; ++s) n += redo[s];
  return n;
}

}  // namespace lsspa

// ---------------------------------------------------------------------------------------------------------------
// The 'argsort' ordering source as a thread of this library: Sobol' points by the generator's own recurrence and their
// row argsort, ahead of the loop that consumes them, with no interpreter in the way (the Python helper thread this
// replaces shares the interpreter lock with the driver's thread: the public call of a small problem waited for
// orderings a third of its time).  The reference: experiments/ground_truth_medium.py:56-60 --
// np.argsort(qmc.Sobol(p, seed).random(n), axis=1).  SciPy's engine is what defines the stream (direction numbers,
// scramble, first point): the caller hands over its direction numbers sv [p][bits], its state before the first step
// q0 [p] and its scale, read off a SciPy engine it has built and CHECKED against that engine's output
// (ls_spa/_samplers.py, _DirectSobol.make); point number i is (q0 XOR the direction numbers of the bits of the Gray
// code of i) * scale, and state i is state i - 1 XOR direction number ctz(i) -- the engine's draw loop.  Ordering number
// g of the run belongs to rank g mod world: every state is stepped through (a row of XORs), only a rank's own rows are
// scaled and sorted.  Rows with equal keys are reported (position and number): the caller sorts those with numpy.
#include <condition_variable>
#include <deque>
#include <mutex>
#include <stdexcept>
#include <string>

namespace lsspa {

struct SobolSampler {
  int p = 0, bits = 0, rank = 0, world = 1, block = 256;
  double scale = 0.0;
  int64_t limit = 0, ahead = 0, ahead_unasked = 0;
  std::vector<uint64_t> sv;      // [bits][p]: direction number b of every dimension side by side
  std::vector<uint64_t> q0;
  struct Part {
    int64_t first = 0, n = 0, used = 0, own_used = 0;     // orderings [first, first + n) of the run; `used` handed out
    bool complete = false;
    std::vector<int32_t> rows;                             // this rank's rows, in order
    std::vector<uint8_t> redo;
  };
  std::deque<Part> parts;        // in the run's order; the producers fill them in, the consumer takes complete ones off the front
  int64_t next_first = 0;        // the first ordering of the next block to be claimed
  int64_t taken = 0;             // orderings handed out
  bool stop = false, asked = false;
  std::string error;
  std::mutex m;
  std::condition_variable cv;
  std::vector<std::thread> th;

  int64_t own_in(int64_t lo, int64_t hi) const {           // orderings lo <= g < hi that belong to this rank
    const int64_t first = lo + ((rank - lo) % world + world) % world;
    return first >= hi ? 0 : (hi - 1 - first) / world + 1;
  }
  // one producer: claims the next block, steps the generator through it from the block's own first state (a block's
  // states depend on nothing but their numbers), sorts this rank's rows, marks the block complete
  void work() {
    try {
      std::vector<double> keys;
      std::vector<uint64_t> state((size_t)p);
      for (;;) {
        Part* part = nullptr;
        {
          std::unique_lock<std::mutex> lk(m);
          cv.wait(lk, [&] { return stop || next_first >= limit || next_first - taken < (asked ? ahead : ahead_unasked); });
          if (stop || next_first >= limit) return;
          parts.emplace_back();
          part = &parts.back();          // (a deque's elements stay where they are when others are added or taken off)
          part->first = next_first;
          part->n = std::min<int64_t>(block, limit - next_first);
          next_first += part->n;
        }
        const int64_t lo = part->first, n = part->n, n_own = own_in(lo, lo + n);
        if (lo + n - 1 > 0 && (64 - __builtin_clzll((unsigned long long)(lo + n - 1))) > bits)
          throw std::runtime_error("more orderings than the generator has points");
        // the state of ordering lo: q0 XOR the direction numbers of the bits of the Gray code of lo
        state = q0;
        const uint64_t gray = (uint64_t)lo ^ ((uint64_t)lo >> 1);
        for (int b = 0; b < bits; ++b)
          if ((gray >> b) & 1) {
            const uint64_t* d = sv.data() + (size_t)b * p;
            for (int j = 0; j < p; ++j) state[j] ^= d[j];
          }
        keys.resize((size_t)std::max<int64_t>(n_own, 1) * p);
        int64_t k = 0;
        for (int64_t i = lo; i < lo + n; ++i) {
          if (i > lo) {
            const uint64_t* d = sv.data() + (size_t)__builtin_ctzll((unsigned long long)i) * p;
            for (int j = 0; j < p; ++j) state[j] ^= d[j];
          }
          if ((i - rank) % world == 0) {
            double* row = keys.data() + (size_t)k * p;
            for (int j = 0; j < p; ++j) row[j] = (double)state[j] * scale;
            ++k;
          }
        }
        std::vector<int32_t> rows((size_t)n_own * p);
        std::vector<uint8_t> redo((size_t)n_own, 0);
        if (n_own > 0) argsort_rows_host(keys.data(), n_own, p, rows.data(), redo.data(), 1);
        {
          std::lock_guard<std::mutex> lk(m);
          part->rows = std::move(rows);
          part->redo = std::move(redo);
          part->complete = true;
        }
        cv.notify_all();
      }
    } catch (const std::exception& e) {
      std::lock_guard<std::mutex> lk(m);
      error = e.what();
      stop = true;
      cv.notify_all();
    }
  }
};

SobolSampler* sobol_sampler_create(int p, int bits, const uint64_t* sv_pb /*[p][bits]*/, const uint64_t* q0, double scale,
                                   int64_t limit, int block, int64_t ahead, int64_t ahead_unasked, int threads, int rank,
                                   int world) {
  auto* s = new SobolSampler();
  s->p = p;
  s->bits = bits;
  s->rank = rank;
  s->world = world;
  s->block = block;
  s->scale = scale;
  s->limit = limit;
  s->ahead = std::max<int64_t>(ahead, block);
  s->ahead_unasked = std::max<int64_t>(std::min<int64_t>(ahead_unasked, s->ahead), 1);
  s->sv.resize((size_t)bits * p);
  for (int j = 0; j < p; ++j)
    for (int b = 0; b < bits; ++b) s->sv[(size_t)b * p + j] = sv_pb[(size_t)j * bits + b];
  s->q0.assign(q0, q0 + p);
  const int nt = threads < 1 ? 1 : (threads > 16 ? 16 : threads);
  if (limit > 0)
    for (int t = 0; t < nt; ++t) s->th.emplace_back([s] { s->work(); });
  return s;
}

void sobol_sampler_destroy(SobolSampler* s) {
  if (!s) return;
  {
    std::lock_guard<std::mutex> lk(s->m);
    s->stop = true;
  }
  s->cv.notify_all();
  for (auto& t : s->th)
    if (t.joinable()) t.join();
  delete s;
}

// The next `count` orderings of the run: *n_taken of them exist; this rank's rows of them go to out ([cap][p], in
// order; *n_own of them); redo_pos / redo_id: positions in `out` and numbers in the run of the rows with equal keys.
// Returns 0, 1 (out too small) or 2 (a producer failed: *err points at its message).
int sobol_sampler_take(SobolSampler* s, int64_t count, int32_t* out, int64_t cap, int64_t* n_taken, int64_t* n_own,
                       int64_t* redo_pos, int64_t* redo_id, int64_t* n_redo, const char** err) {
  *n_taken = *n_own = *n_redo = 0;
  std::unique_lock<std::mutex> lk(s->m);
  if (!s->asked) {
    s->asked = true;
    s->cv.notify_all();
  }
  int64_t need = count;
  while (need > 0) {
    s->cv.wait(lk, [&] {
      return !s->error.empty() || (!s->parts.empty() && s->parts.front().complete) ||
             (s->parts.empty() && s->next_first >= s->limit);
    });
    if (!s->error.empty()) {
      if (err) *err = s->error.c_str();
      return 2;
    }
    if (s->parts.empty()) break;
    SobolSampler::Part& P = s->parts.front();
    const int64_t lo = P.first + P.used, use = std::min<int64_t>(P.n - P.used, need);
    const int64_t k = s->own_in(lo, lo + use);
    if (*n_own + k > cap) return 1;
    std::copy(P.rows.begin() + (size_t)P.own_used * s->p, P.rows.begin() + (size_t)(P.own_used + k) * s->p,
              out + (size_t)*n_own * s->p);
    for (int64_t r = 0; r < k; ++r)
      if (P.redo[(size_t)(P.own_used + r)]) {
        const int64_t first_own = lo + ((s->rank - lo) % s->world + s->world) % s->world;
        redo_pos[*n_redo] = *n_own + r;
        redo_id[*n_redo] = first_own + r * s->world;
        ++*n_redo;
      }
    *n_own += k;
    *n_taken += use;
    need -= use;
    P.used += use;
    P.own_used += k;
    s->taken += use;
    if (P.used == P.n) s->parts.pop_front();
    s->cv.notify_all();
  }
  return 0;
}

}  // namespace lsspa
