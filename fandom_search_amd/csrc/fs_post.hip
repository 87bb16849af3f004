// fs_post.hip -- everything between the scan bitmap and the output records.
//
//   quad bitmap --popcount/scan--> quad list --verify--> hits (window p, gram g,
//   work w) --scan--> compact hit list --new-word counts/scan--> records
//
// Reference semantics reproduced here (file:line in /root/reference):
//   search.py:182-184  keep candidates with distance < threshold: in the exact
//                      regime those are the script windows whose n vector ids
//                      equal the fan window's; NearestFilter keeps the first N
//                      of them in script order (ties in a stable sort)
//   search.py:189-190  Levenshtein of the script span text against
//                      str(list of fan tokens)
//   search.py:192-218  n word records per match
//   search.py:224-226  per fan word the FIRST record of minimal combined
//                      distance, output sorted by word index
// A window that crosses a work boundary is not a window of the reference
// (windows are built per file, search.py:170-173); verify drops those.
//
// No kernel here needs a host round trip: element counts live in the device
// status block, every kernel is a grid-stride loop over a device-side count,
// producers clamp to their buffer capacity and the host re-runs with larger
// buffers if a total exceeded it.
#include "fs_internal.h"

namespace {

constexpr int kScanBlocks = 256;
constexpr int kScanThreads = 256;

struct NSrc {               // element count, known on the host or on the device
  const uint32_t* ptr;      // device count (clamped to cap, times mult) ...
  uint32_t mult, cap;
  uint32_t fixed;           // ... or a host constant when ptr == nullptr
  __device__ uint32_t get() const {
    if (!ptr) return fixed;
    uint32_t v = *ptr;
    if (v > cap) v = cap;
    return v * mult;
  }
};

// ---- block-wide helpers ----------------------------------------------------

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}

// exclusive scan of one value per thread over a 256-thread block; returns the
// thread's exclusive prefix, *total = block sum.  s_w: 4 words of LDS.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_w, uint32_t* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t inc = wave_incl_scan(v, lane);
  if (lane == 63) s_w[w] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < kScanThreads / 64; ++i) {
    const uint32_t x = s_w[i];
    if (i < w) base += x;
    tot += x;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// ---- three-kernel exclusive scan over f(0..n) -------------------------------

template <class F>
__global__ __launch_bounds__(kScanThreads) void k_scan_reduce(F f, NSrc ns, uint32_t* bsum) {
  __shared__ uint32_t s_w[4];
  const uint32_t n = ns.get();
  const uint32_t chunk = (n + kScanBlocks - 1) / kScanBlocks;
  const uint32_t lo = blockIdx.x * chunk;
  uint32_t hi = lo + chunk;
  if (hi > n) hi = n;
  uint32_t acc = 0;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += kScanThreads) acc += f(i);
  uint32_t tot;
  block_excl_scan(acc, s_w, &tot);
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(kScanBlocks) void k_scan_mid(uint32_t* bsum, uint32_t* total) {
  __shared__ uint32_t s_w[4];
  uint32_t tot;
  const uint32_t v = bsum[threadIdx.x];
  const uint32_t ex = block_excl_scan(v, s_w, &tot);
  bsum[threadIdx.x] = ex;
  if (threadIdx.x == 0) *total = tot;
}

template <class F>
__global__ __launch_bounds__(kScanThreads) void k_scan_down(F f, NSrc ns, const uint32_t* bsum,
                                                            uint32_t* out) {
  __shared__ uint32_t s_w[4];
  const uint32_t n = ns.get();
  const uint32_t chunk = (n + kScanBlocks - 1) / kScanBlocks;
  const uint32_t lo = blockIdx.x * chunk;
  uint32_t hi = lo + chunk;
  if (hi > n) hi = n;
  uint32_t carry = bsum[blockIdx.x];
  for (uint32_t t = lo; t < hi; t += kScanThreads) {
    const uint32_t i = t + threadIdx.x;
    const uint32_t v = i < hi ? f(i) : 0;
    uint32_t tot;
    const uint32_t ex = block_excl_scan(v, s_w, &tot);
    if (i < hi) out[i] = carry + ex;
    carry += tot;
  }
}

template <class F>
int device_scan(F f, NSrc ns, uint32_t* out, uint32_t* bsum, uint32_t* total, hipStream_t s) {
  hipLaunchKernelGGL(k_scan_reduce<F>, dim3(kScanBlocks), dim3(kScanThreads), 0, s, f, ns, bsum);
  hipLaunchKernelGGL(k_scan_mid, dim3(1), dim3(kScanBlocks), 0, s, bsum, total);
  hipLaunchKernelGGL(k_scan_down<F>, dim3(kScanBlocks), dim3(kScanThreads), 0, s, f, ns, bsum, out);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

// ---- count functors -----------------------------------------------------------

struct PopcountF {
  const uint64_t* qbm;
  __device__ uint32_t operator()(uint32_t i) const { return __popcll(qbm[i]); }
};
struct IsHitF {
  const uint32_t* hg;
  __device__ uint32_t operator()(uint32_t i) const { return hg[i] != FS_NONE; }
};
struct MatchCountF {
  const uint32_t* hit_g;
  const uint32_t* gcnt;
  __device__ uint32_t operator()(uint32_t h) const { return gcnt[hit_g[h]]; }
};
// fan words first covered by hit h: the last min(n, p_h - p_{h-1}) words of its
// window (hits are in ascending position; windows of different works are at
// least n apart because a window never crosses a work boundary)
struct NewWordsF {
  const uint32_t* hit_p;
  uint32_t n;
  __device__ uint32_t operator()(uint32_t h) const {
    if (h == 0) return n;
    const uint32_t d = hit_p[h] - hit_p[h - 1];
    return d < n ? d : n;
  }
};

// ---- kernels --------------------------------------------------------------------

__global__ void k_expand(const uint64_t* __restrict__ qbm, uint32_t n_words,
                         const uint32_t* __restrict__ off1, uint32_t* __restrict__ qpos,
                         uint32_t qcap) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_words;
       i += gridDim.x * blockDim.x) {
    uint64_t bits = qbm[i];
    uint32_t idx = off1[i];
    while (bits) {
      const int b = __ffsll((unsigned long long)bits) - 1;
      bits &= bits - 1;
      if (idx < qcap) qpos[idx] = i * 256u + 4u * (uint32_t)b;
      ++idx;
    }
  }
}

// first index with work_off[idx] > p, minus one
__device__ __forceinline__ uint32_t work_of(const uint64_t* work_off, uint32_t n_works, uint64_t p) {
  uint32_t lo = 0, hi = n_works;   // invariant: work_off[lo] <= p < work_off[hi] (p < n_tok)
  while (hi - lo > 1) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (work_off[mid] <= p) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ void k_verify(CorpusDev c, GramIndexDev g, const uint32_t* __restrict__ qpos,
                         NSrc nq4, uint32_t* __restrict__ hg, uint32_t* __restrict__ hw,
                         fs_status* st) {
  const uint32_t total = nq4.get();
  const uint32_t slot_mask = (1u << g.log2_slots) - 1;
  uint32_t positives = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += gridDim.x * blockDim.x) {
    const uint64_t p = (uint64_t)qpos[i >> 2] + (i & 3);
    uint32_t gram = FS_NONE;
    if (p + g.n <= c.n_tok) {
      const uint32_t* t = c.tok + p;
      uint32_t h = fs_premix(t[0]);
      for (int k = 1; k < g.n; ++k) h = fs_fold(h, fs_premix(t[k]));
      h = fs_finish(h);
      const uint32_t word = g.filter[fs_bloom_word(h, g.log2_words)];
      const uint32_t mask = fs_bloom_mask(h);
      if ((word & mask) == mask) {
        ++positives;
        uint32_t slot = fs_table_slot(h, g.log2_slots);
        for (;;) {
          const uint32_t e = g.table[slot];
          if (e == 0) break;
          const uint32_t* s = g.stok + g.gpos[(size_t)(e - 1) * g.nn];
          bool same = true;
          for (int k = 0; k < g.n; ++k) same = same && (s[k] == t[k]);
          if (same) { gram = e - 1; break; }
          slot = (slot + 1) & slot_mask;
        }
      }
      if (gram != FS_NONE) {
        const uint32_t w = work_of(c.work_off, c.n_works, p);
        if (p + g.n > c.work_off[w + 1]) gram = FS_NONE;   // crosses into the next work
        else hw[i] = w;
      }
    }
    hg[i] = gram;
  }
  // statistics: filter-positive windows
  for (int d = 32; d > 0; d >>= 1) positives += __shfl_xor(positives, d);
  if ((threadIdx.x & 63) == 0 && positives) atomicAdd(&st->n_cand_windows, positives);
}

__global__ void k_compact(const uint32_t* __restrict__ qpos, NSrc nq4,
                          const uint32_t* __restrict__ hg, const uint32_t* __restrict__ hw,
                          const uint32_t* __restrict__ hoff, uint32_t hcap,
                          uint32_t* __restrict__ hit_p, uint32_t* __restrict__ hit_g,
                          uint32_t* __restrict__ hit_w) {
  const uint32_t total = nq4.get();
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += gridDim.x * blockDim.x) {
    const uint32_t gram = hg[i];
    if (gram == FS_NONE) continue;
    const uint32_t idx = hoff[i];
    if (idx >= hcap) continue;
    hit_p[idx] = qpos[i >> 2] + (i & 3);
    hit_g[idx] = gram;
    hit_w[idx] = hw[i];
  }
}

// Levenshtein.distance(match_str, fan_context), search.py:189-190:
//   match_str   = script words s .. s+n-1 joined by single spaces
//   fan_context = '[' + ', '.join(fan token texts) + ']'
// unit costs over code points.  One thread, operands in scratch.
__device__ uint32_t lev_device(const GramIndexDev& g, uint32_t s, const uint32_t* fan_sid,
                               const uint32_t* chars, const uint64_t* coff, uint32_t n_str,
                               fs_status* st) {
  uint32_t a[FS_LEV_MAX], b[FS_LEV_MAX];
  uint16_t row[FS_LEV_MAX + 1];
  uint32_t la = 0, lb = 0;
  bool ok = true;
  for (int k = 0; k < g.n && ok; ++k) {
    if (k) { if (la < FS_LEV_MAX) a[la] = ' '; ++la; }
    for (uint64_t c = g.soff[s + k]; c < g.soff[s + k + 1]; ++c) {
      if (la < FS_LEV_MAX) a[la] = g.schars[c];
      ++la;
    }
  }
  if (lb < FS_LEV_MAX) b[lb] = '[';
  ++lb;
  for (int k = 0; k < g.n; ++k) {
    if (k) {
      if (lb < FS_LEV_MAX) b[lb] = ','; ++lb;
      if (lb < FS_LEV_MAX) b[lb] = ' '; ++lb;
    }
    const uint32_t sid = fan_sid[k];
    if (sid >= n_str) { st->bad_string = 1; ok = false; break; }
    for (uint64_t c = coff[sid]; c < coff[sid + 1]; ++c) {
      if (lb < FS_LEV_MAX) b[lb] = chars[c];
      ++lb;
    }
  }
  if (lb < FS_LEV_MAX) b[lb] = ']';
  ++lb;
  if (!ok) return 0;
  if (la > FS_LEV_MAX || lb > FS_LEV_MAX) { st->lev_overflow = 1; return 0; }
  for (uint32_t j = 0; j <= lb; ++j) row[j] = (uint16_t)j;
  for (uint32_t x = 1; x <= la; ++x) {
    uint32_t diag = row[0];
    row[0] = (uint16_t)x;
    const uint32_t ca = a[x - 1];
    for (uint32_t j = 1; j <= lb; ++j) {
      const uint32_t up = row[j];
      uint32_t best = diag + (ca != b[j - 1] ? 1u : 0u);
      if (up + 1 < best) best = up + 1;
      const uint32_t left = row[j - 1];
      if (left + 1 < best) best = left + 1;
      diag = up;
      row[j] = (uint16_t)best;
    }
  }
  return row[lb];
}

// Per (gram, rank) Levenshtein table for corpora whose string id == vector id:
// the fan text of a hit is then a function of the gram alone.
__global__ void k_levtab(GramIndexDev g, CorpusDev c, uint32_t* __restrict__ levtab,
                         fs_status* st) {
  const uint32_t total = g.n_grams * (uint32_t)g.nn;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += gridDim.x * blockDim.x) {
    const uint32_t gram = i / g.nn, r = i % g.nn;
    uint32_t v = 0;
    if (r < g.gcnt[gram]) {
      const uint32_t s = g.gpos[i];
      const uint32_t first = g.gpos[(size_t)gram * g.nn];
      v = lev_device(g, s, g.stok + first, c.chars, c.coff, c.n_str, st);
    }
    levtab[i] = v;
  }
}

// Per match Levenshtein when fan tokens carry their own string ids.
__global__ void k_matchlev(GramIndexDev g, CorpusDev c, const uint32_t* __restrict__ hit_p,
                           const uint32_t* __restrict__ hit_g, const uint32_t* __restrict__ moff,
                           NSrc nh_nn, uint32_t mcap, uint32_t* __restrict__ mlev,
                           fs_status* st) {
  const uint32_t total = nh_nn.get();
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += gridDim.x * blockDim.x) {
    const uint32_t h = i / g.nn, r = i % g.nn;
    const uint32_t gram = hit_g[h];
    if (r >= g.gcnt[gram]) continue;
    const uint32_t idx = moff[h] + r;
    if (idx >= mcap) continue;
    const uint32_t s = g.gpos[(size_t)gram * g.nn + r];
    mlev[idx] = lev_device(g, s, c.str + hit_p[h], c.chars, c.coff, c.n_str, st);
  }
}

// One thread per (hit, k): the k-th fan word first covered by hit h.  Its record
// is the first minimum of dist*lev over every (hit, rank) whose window covers
// the word, in the reference's insertion order: ascending window position, then
// NearestFilter rank (search.py:176-218, 224-225).
__global__ void k_rows(GramIndexDev g, CorpusDev c, const uint32_t* __restrict__ hit_p,
                       const uint32_t* __restrict__ hit_g, const uint32_t* __restrict__ hit_w,
                       const uint32_t* __restrict__ roff, const uint32_t* __restrict__ levtab,
                       const uint32_t* __restrict__ moff, const uint32_t* __restrict__ mlev,
                       const uint32_t* n_hits_ptr, uint32_t hcap, uint32_t rcap,
                       fs_row* __restrict__ rows) {
  uint32_t nh = *n_hits_ptr;
  if (nh > hcap) nh = hcap;
  const uint32_t n = g.n;
  const uint32_t total = nh * n;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += gridDim.x * blockDim.x) {
    const uint32_t h = i / n, k = i % n;
    const uint32_t p = hit_p[h];
    uint32_t cnt = n;
    if (h) { const uint32_t d = p - hit_p[h - 1]; if (d < n) cnt = d; }
    if (k >= cnt) continue;
    const uint32_t ridx = roff[h] + k;
    if (ridx >= rcap) continue;
    const uint32_t x = p + n - cnt + k;           // global token position of the word
    bool have = false;
    fs_row best;
    for (uint32_t h2 = h; h2 < nh; ++h2) {
      const uint32_t p2 = hit_p[h2];
      if (p2 > x) break;
      const uint32_t gram = hit_g[h2];
      const uint32_t m = g.gcnt[gram];
      for (uint32_t r = 0; r < m; ++r) {
        const uint32_t s = g.gpos[(size_t)gram * g.nn + r];
        const uint32_t lv = levtab ? levtab[(size_t)gram * g.nn + r] : mlev[moff[h2] + r];
        const double dist = g.selfdist[s];
        const double comb = __dmul_rn(dist, (double)lv);
        if (!have || comb < best.comb) {
          have = true;
          best.orig_ix = s + (x - p2);
          best.lev = lv;
          best.dist = dist;
          best.comb = comb;
        }
      }
    }
    const uint32_t w = hit_w[h];
    best.work = w;
    best.fan_ix = (uint32_t)((uint64_t)x - c.work_off[w]);
    rows[ridx] = best;
  }
}

__global__ void k_count_matches(const uint32_t* __restrict__ hit_g,
                                const uint32_t* __restrict__ gcnt, const uint32_t* n_hits_ptr,
                                uint32_t hcap, fs_status* st) {
  uint32_t nh = *n_hits_ptr;
  if (nh > hcap) nh = hcap;
  uint32_t acc = 0;
  for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h < nh; h += gridDim.x * blockDim.x)
    acc += gcnt[hit_g[h]];
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(&st->n_matches, acc);
}

}  // namespace

int fs_launch_levtab(fs_index* ix, fs_corpus* c, hipStream_t s) {
  const size_t total = (size_t)ix->n_grams * ix->cfg.nearest_n;
  FS_TRY(c->d_levtab.reserve(total));
  if (total) {
    const uint32_t blocks = (uint32_t)((total + 255) / 256);
    hipLaunchKernelGGL(k_levtab, dim3(blocks > 1024 ? 1024 : blocks), dim3(256), 0, s,
                       ix->gram_dev(), c->dev(), c->d_levtab.p, ix->d_status.p);
    FS_HIP(hipGetLastError());
  }
  return FS_OK;
}

int fs_launch_post(fs_index* ix, fs_corpus* c, uint32_t n_bm_words, uint32_t qcap,
                   uint32_t hcap, uint32_t mcap, uint32_t rcap, fs_row* d_rows, hipStream_t s) {
  const GramIndexDev g = ix->gram_dev();
  const CorpusDev cd = c->dev();
  fs_status* st = ix->d_status.p;
  uint32_t* bsum = ix->w_bsum.p;
  const uint32_t nn = ix->cfg.nearest_n;
  const uint32_t n = ix->cfg.window_size;
  const bool use_levtab = !c->has_str;
  const int grid = ix->num_cu * 4;

  // 1. quads
  FS_TRY(device_scan(PopcountF{ix->w_qbm.p}, NSrc{nullptr, 0, 0, n_bm_words}, ix->w_off1.p, bsum,
                     &st->n_quads, s));
  if (n_bm_words) {
    hipLaunchKernelGGL(k_expand, dim3(grid), dim3(256), 0, s, ix->w_qbm.p, n_bm_words,
                       ix->w_off1.p, ix->w_qpos.p, qcap);
  }
  // 2. verify the four windows of every quad
  const NSrc nq4{&st->n_quads, 4, qcap, 0};
  hipLaunchKernelGGL(k_verify, dim3(grid), dim3(256), 0, s, cd, g, ix->w_qpos.p, nq4, ix->w_hg.p,
                     ix->w_hw.p, st);
  FS_TRY(device_scan(IsHitF{ix->w_hg.p}, nq4, ix->w_hoff.p, bsum, &st->n_hits, s));
  hipLaunchKernelGGL(k_compact, dim3(grid), dim3(256), 0, s, ix->w_qpos.p, nq4, ix->w_hg.p,
                     ix->w_hw.p, ix->w_hoff.p, hcap, ix->w_hit_p.p, ix->w_hit_g.p, ix->w_hit_w.p);
  // 3. matches
  const NSrc nh{&st->n_hits, 1, hcap, 0};
  if (use_levtab) {
    hipLaunchKernelGGL(k_count_matches, dim3(grid), dim3(256), 0, s, ix->w_hit_g.p, ix->d_gcnt.p,
                       &st->n_hits, hcap, st);
  } else {
    FS_TRY(device_scan(MatchCountF{ix->w_hit_g.p, ix->d_gcnt.p}, nh, ix->w_moff.p, bsum,
                       &st->n_matches, s));
    const NSrc nh_nn{&st->n_hits, nn, hcap, 0};
    hipLaunchKernelGGL(k_matchlev, dim3(grid), dim3(256), 0, s, g, cd, ix->w_hit_p.p,
                       ix->w_hit_g.p, ix->w_moff.p, nh_nn, mcap, ix->w_mlev.p, st);
  }
  // 4. records
  FS_TRY(device_scan(NewWordsF{ix->w_hit_p.p, n}, nh, ix->w_roff.p, bsum, &st->n_rows, s));
  hipLaunchKernelGGL(k_rows, dim3(grid), dim3(256), 0, s, g, cd, ix->w_hit_p.p, ix->w_hit_g.p,
                     ix->w_hit_w.p, ix->w_roff.p, use_levtab ? c->d_levtab.p : nullptr,
                     ix->w_moff.p, ix->w_mlev.p, &st->n_hits, hcap, rcap, d_rows);
  FS_HIP(hipGetLastError());
  return FS_OK;
}
