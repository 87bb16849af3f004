// fs_post.hip -- everything between the scan's output and the records: three launches
// behind the scan on the direct path, five on the bitmap paths, no host round trip.
//
//   k_verify_direct          direct path: the scan's candidate records (per wave range)
//                            -> positions in LDS -> (n-gram id, work id) or FS_NONE per
//                            candidate, dense arrays in position order
//   k_reduce<SubTileCount>   bitmap paths: per-block partial sums of the scan's sub-tile
//                            counts (the 8-tokens-per-lane scan writes them itself)
//   k_expand<TPL>            bitmap -> candidate window positions, in position
//                            order (block prefix from the partial sums + an
//                            in-block scan; one thread expands one bitmap word)
//   k_verify                 candidate -> (n-gram id, work id) or FS_NONE: exact
//                            id-for-id comparison with the script n-gram found
//                            through the hash-and-displace table; windows that
//                            cross a work boundary are dropped
//   k_hitrows                per candidate (hit?, fan words first covered) and the
//                            combined distance of its best rank, kept for k_rows,
//                            and their per-block partial sums
//   k_rows                   per 256-candidate tile a block scan gives every hit
//                            its record offset, then one thread per record: the
//                            first minimum of dist*lev over all hits covering
//                            the word (each hit offers its best rank)
// plus the Levenshtein kernels (k_levtab: per (n-gram, rank) once per string table;
// k_strbest: a lane per hit when fan tokens carry their own strings, k_strrec its string
// records; k_matchlev + k_cbest: the wave-per-pair form behind FS_STR_FAST=0), the
// wire-record unpack, the `format` histogram and corpus-build helpers.
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
// (windows are built per file, search.py:170-173).
//
// Element counts live in the device status block; every kernel sizes its loops
// from there, producers clamp to their buffer capacity and the host re-runs
// with larger buffers if a total exceeded it.
#include "fs_device.h"

#include <stdlib.h>

#include <algorithm>

namespace {

using namespace fsdev;

// ---- count functors -----------------------------------------------------------

struct SubTileCountF {
  const uint32_t* qcnt;
  __device__ uint32_t operator()(uint32_t i) const { return qcnt[i]; }
};

// per candidate: low word 1 if it is a hit, high word = number of fan words the
// hit is the first to cover: the last min(n, p - p_prev_hit) words of its window
// (hits are in ascending position; a hit of another work is at least n away
// because no window crosses a work boundary)
struct HitRowsF {
  const uint32_t* cpos;
  const uint32_t* cg;
  uint32_t n;
  __device__ uint64_t operator()(uint32_t c) const {
    if (cg[c] == FS_NONE) return 0;
    const uint32_t p = cpos[c];
    uint32_t cnt = n;
    for (uint32_t j = c; j-- > 0;) {
      const uint32_t d = p - cpos[j];
      if (d >= n) break;
      if (cg[j] != FS_NONE) { cnt = d; break; }
    }
    return 1ull | ((uint64_t)cnt << 32);
  }
};

// (hits, records) partial sums; the per-candidate value is kept for k_rows so that
// the backward walk is done once
__global__ __launch_bounds__(kThreads) void k_hitrows(HitRowsF f, NSrc ns,
                                                      const fs_best* __restrict__ best_tab,
                                                      int best_per_cand,
                                                      uint64_t* __restrict__ hv,
                                                      double* __restrict__ hcomb,
                                                      uint64_t* __restrict__ bsum) {
  __shared__ uint64_t s_w[4];
  uint32_t lo, hi;
  chunk_of_block(ns.get(), &lo, &hi);
  uint64_t acc = 0;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += kThreads) {
    // what this candidate offers to the words it covers: the combined distance of its
    // best rank, +inf when it is not a hit.  k_rows then walks 8-byte neighbours
    // instead of chasing candidate -> n-gram -> 32-byte record for every covering hit.
    const uint32_t gram = f.cg[i];
    hcomb[i] = gram == FS_NONE ? __longlong_as_double(0x7FF0000000000000ll)
                               : best_tab[best_per_cand ? i : gram].comb;
    const uint64_t v = f(i);
    hv[i] = v;
    acc += v;
  }
  uint64_t tot;
  block_excl_scan(acc, s_w, &tot);
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// ---- kernels --------------------------------------------------------------------

// TPL = tokens per lane of the scan that wrote the bitmap.
//   TPL 4  four ballot words per sub-tile of 256 tokens, bit L of word j <-> window
//          256 i + 4 L + j.  Candidates are numbered in position order (L, then j):
//          a window's index is the sub-tile's base + the set bits of all words below
//          lane L + the set bits of the earlier words at lane L.
//   TPL 8  eight words per sub-tile of 512 tokens in natural order, bit L of word j
//          <-> window 512 i + 64 j + L.
// One thread per word.  (Verifying inside this kernel measured slower than the separate
// k_verify launch, 110 against 84 us per C2 step in round 1: the thread that decodes a ballot
// word then runs the dependent-load chain of each of its candidates one after the other.)
template <int TPL>
__global__ __launch_bounds__(kThreads) void k_expand(const uint64_t* __restrict__ qbm,
                                                     const uint32_t* __restrict__ qcnt,
                                                     uint32_t n_sub,
                                                     const uint32_t* __restrict__ bsum,
                                                     uint32_t* __restrict__ cpos, uint32_t ccap,
                                                     fs_status* st) {
  constexpr int SUBS = kThreads / TPL;          // sub-tiles per block iteration
  __shared__ uint32_t s_w[4];
  __shared__ uint32_t s_base[SUBS];
  uint32_t lo, hi;
  chunk_of_block(n_sub, &lo, &hi);
  const int j = threadIdx.x % TPL, sl = threadIdx.x / TPL;
  // loads of one tile: requested before the barriers of the prefix sums, so that
  // they are in flight together with the partial sums
  uint32_t cnt = 0;
  uint64_t b[TPL] = {};
  auto load_tile = [&](uint32_t t0) {
    const uint32_t sub = t0 + sl;
    const bool live = sub < hi;
    cnt = (live && j == 0) ? qcnt[sub] : 0;
    if (live) {
      const uint4* w = reinterpret_cast<const uint4*>(qbm + (size_t)sub * TPL);   // 16-B aligned
#pragma unroll
      for (int h = 0; h < TPL / 2; ++h) {
        const uint4 q = w[h];
        b[2 * h] = q.x | ((uint64_t)q.y << 32);
        b[2 * h + 1] = q.z | ((uint64_t)q.w << 32);
      }
    }
  };
  if (lo < hi) load_tile(lo);
  uint32_t carry = block_prefix(bsum, s_w);
  if (blockIdx.x == kNB && threadIdx.x == 0) st->n_cands = carry;    // the extra block: total
  for (uint32_t t0 = lo; t0 < hi; t0 += SUBS) {
    if (t0 != lo) load_tile(t0);
    const uint32_t sub = t0 + sl;
    const bool live = sub < hi;
    uint32_t tile_total;
    const uint32_t ex = block_excl_scan(cnt, s_w, &tile_total);
    if (j == 0) s_base[sl] = carry + ex;
    __syncthreads();
    if (live) {
      uint64_t mine = b[0];
#pragma unroll
      for (int k = 1; k < TPL; ++k) mine = j == k ? b[k] : mine;
      const uint64_t mine_all = mine;
      (void)mine_all;
      const uint32_t base = s_base[sl];
      while (mine) {
        const int L = __ffsll((unsigned long long)mine) - 1;
        mine &= mine - 1;
        const uint64_t below = (1ull << L) - 1;
        uint32_t idx = base;
        uint32_t p;
        if constexpr (TPL == 8) {
#pragma unroll
          for (int k = 0; k < TPL; ++k)
            if (k < j) idx += __popcll(b[k]);
          idx += __popcll(mine_all & below);
          p = sub * 512u + 64u * (uint32_t)j + (uint32_t)L;
        } else {
#pragma unroll
          for (int k = 0; k < TPL; ++k) {
            idx += __popcll(b[k] & below);
            if (k < j) idx += (uint32_t)((b[k] >> L) & 1);
          }
          p = sub * (64u * TPL) + (uint32_t)TPL * (uint32_t)L + (uint32_t)j;
        }
        if (idx < ccap) cpos[idx] = p;
      }
    }
    __syncthreads();
    carry += tile_total;
  }
}

// Verification of the candidate list, one thread per candidate.
__global__ __launch_bounds__(kThreads) void k_verify(CorpusDev c, GramIndexDev g,
                                                     const uint32_t* __restrict__ cpos,
                                                     NSrc nc, uint32_t* __restrict__ cg,
                                                     uint32_t* __restrict__ cw,
                                                     uint32_t* __restrict__ bmatch) {
  __shared__ uint32_t s_w[4];
  const uint32_t total = nc.get();
  uint32_t matches = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += gridDim.x * blockDim.x) {
    uint32_t w = 0, kept = 0;
    const uint32_t gram = verify_window(c, g, cpos[i], &w, &kept);
    cg[i] = gram;
    if (gram != FS_NONE) { cw[i] = w; matches += kept; }
  }
  uint32_t tot;
  block_excl_scan(matches, s_w, &tot);
  if (threadIdx.x == 0) bmatch[blockIdx.x] = tot;
}

// Verification on the direct path (fs_scan_extra): the scan left, per wave range, a list
// of records {(position / 8) << 8 | flag byte, rank of the first candidate in the range}
// and info[range] = {records, candidates}.  Block b takes chunk b: its candidates start
// at the sum of the chunk sums before it.  Stage 1 turns the chunk's records (four wave
// ranges, back to back over the threads) into (position, index) pairs in LDS; stage 2
// verifies one candidate per thread exactly like k_verify.  (Verifying inside the
// record loop instead measured 36 us against 10 us: the staged form keeps full waves
// on the dependent-load chains.)  Writes the dense per-candidate arrays of the later
// kernels; the extra block publishes the total.
// (launch bounds: all kNB blocks must be resident at once, i.e. eight waves per SIMD)
__global__ __launch_bounds__(kThreads, 8) void k_verify_direct(CorpusDev c, GramIndexDev g,
                                                            const uint32_t* __restrict__ bsum,
                                                            const uint2* __restrict__ recs,
                                                            const uint2* __restrict__ info,
                                                            uint32_t capw, uint32_t ccap,
                                                            uint32_t* __restrict__ cpos,
                                                            uint32_t* __restrict__ cg,
                                                            uint32_t* __restrict__ cw,
                                                            uint32_t* __restrict__ bmatch,
                                                            fs_status* st) {
  __shared__ uint32_t s_w[4];
  // requested before the barriers of the prefix sum
  uint2 inf[4] = {};
  if (blockIdx.x < kNB) {
    const uint4* src = reinterpret_cast<const uint4*>(info + (size_t)blockIdx.x * 4);
    const uint4 a = src[0], b = src[1];
    inf[0] = make_uint2(a.x, a.y); inf[1] = make_uint2(a.z, a.w);
    inf[2] = make_uint2(b.x, b.y); inf[3] = make_uint2(b.z, b.w);
  }
  const uint32_t base = block_prefix(bsum, s_w);
  if (blockIdx.x == kNB) {                       // the extra block: its prefix is the total
    if (threadIdx.x == 0) st->n_cands = base;
    return;
  }
  // the records of the chunk's four wave ranges, back to back over the block's threads
  uint32_t nr[4], roff[5], coff[4];
  roff[0] = 0; coff[0] = base;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    nr[k] = inf[k].x;
    if (nr[k] > capw) {                          // the host repeats the search with longer lists
      if (threadIdx.x == 0) atomicMax(&st->max_recs, nr[k]);
      nr[k] = capw;
    }
    roff[k + 1] = roff[k] + nr[k];
    if (k < 3) coff[k + 1] = coff[k] + inf[k].y;
  }
  const uint2* lists = recs + (size_t)blockIdx.x * 4 * capw;
  // stage 1: records -> positions of the chunk's candidates, in candidate order, in LDS
  // (1024 at a time; a slot stays FS_NONE when its record did not fit the list)
  __shared__ uint32_t s_p[1024];
  const uint32_t n_c = coff[3] + inf[3].y - base;          // candidates of the chunk
  uint32_t matches = 0;
  for (uint32_t c0 = 0; c0 < n_c; c0 += 1024) {
    const uint32_t m = n_c - c0 < 1024 ? n_c - c0 : 1024;
    for (uint32_t t = threadIdx.x; t < m; t += kThreads) s_p[t] = FS_NONE;
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < roff[4]; t += kThreads) {
      const int k = (t >= roff[1]) + (t >= roff[2]) + (t >= roff[3]);
      const uint32_t r = t - (k == 0 ? 0u : k == 1 ? roff[1] : k == 2 ? roff[2] : roff[3]);
      const uint32_t cb = k == 0 ? coff[0] : k == 1 ? coff[1] : k == 2 ? coff[2] : coff[3];
      const uint2 rec = lists[(size_t)k * capw + r];
      uint32_t flags = rec.x & 0xFFu;
      const uint32_t p0 = (rec.x >> 8) << 3;
      uint32_t li = cb + rec.y - base;                      // index inside the chunk
      while (flags) {
        const int bb = __ffs(flags) - 1;
        flags &= flags - 1;
        if (li >= c0 && li - c0 < m) s_p[li - c0] = p0 + (uint32_t)bb;
        ++li;
      }
    }
    __syncthreads();
    // stage 2: one candidate per thread, as in k_verify.  Every index below the total
    // is written (the later kernels read them all): a candidate whose record was cut
    // off is marked as no hit, and the host repeats the search with longer lists.
    for (uint32_t t = threadIdx.x; t < m; t += kThreads) {
      const uint32_t i = base + c0 + t;
      if (i >= ccap) continue;
      const uint32_t p = s_p[t];
      uint32_t w = 0, kept = 0, gram = FS_NONE;
      if (p != FS_NONE) gram = verify_window(c, g, p, &w, &kept);
      cpos[i] = p;
      cg[i] = gram;
      if (gram != FS_NONE) { cw[i] = w; matches += kept; }
    }
    __syncthreads();
  }
  uint32_t tot;
  block_excl_scan(matches, s_w, &tot);
  if (threadIdx.x == 0) bmatch[blockIdx.x] = tot;
}

// Per (gram, rank) Levenshtein table: the distance of script window gpos[gram][rank]
// against the strings whose ids are the gram's vector ids.  For corpora whose string id ==
// vector id the fan text of a hit is a function of the gram alone; corpora with string ids
// of their own use the table for the hits whose tokens all carry string id == vector id
// (k_matchlev).  One wave per entry.  `tolerant` (string ids of their own): an id without
// a string or an over-long text gives FS_NONE instead of an error.
__global__ __launch_bounds__(256) void k_levtab(GramIndexDev g, CorpusDev c,
                                                uint32_t* __restrict__ levtab, fs_status* st,
                                                bool tolerant) {
  __shared__ uint32_t s_a[4][FS_LEV_MAX + 2], s_b[4][FS_LEV_MAX + 2];
  __shared__ fs_status s_st[4];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t total = g.n_grams * (uint32_t)g.nn;
  for (uint32_t i = blockIdx.x * 4 + wave; i < total; i += gridDim.x * 4) {
    const uint32_t gram = i / g.nn, r = i % g.nn;
    uint32_t v = 0;
    if (r < g.gcnt[gram]) {                              // wave-uniform
      const uint32_t first = g.gpos[(size_t)gram * g.nn];
      if (tolerant) {
        if (lane == 0) { s_st[wave].bad_string = 0; s_st[wave].lev_overflow = 0; }
        __builtin_amdgcn_wave_barrier();
        v = lev_wave(g, g.gpos[i], g.stok + first, c.chars, c.coff, c.n_str, &s_st[wave], s_a[wave],
                     s_b[wave]);
        __builtin_amdgcn_wave_barrier();
        if (s_st[wave].bad_string | s_st[wave].lev_overflow) v = FS_NONE;
      } else {
        v = lev_wave(g, g.gpos[i], g.stok + first, c.chars, c.coff, c.n_str, st, s_a[wave], s_b[wave]);
      }
    }
    if (lane == 0) levtab[i] = v;
    __builtin_amdgcn_wave_barrier();
  }
}

// Per (candidate, rank) Levenshtein when fan tokens carry their own string ids.  64
// candidates per wave and step, NWAVES apart; first one lane per candidate: not a hit ->
// nothing; every token of the window with string id == vector id -> the table's values for
// all its ranks; the rest one (candidate, rank) at a time by the whole wave.
__global__ __launch_bounds__(256) void k_matchlev(GramIndexDev g, CorpusDev c,
                                                  const uint32_t* __restrict__ cpos,
                                                  const uint32_t* __restrict__ cg, NSrc nc,
                                                  const uint32_t* __restrict__ levtab,
                                                  uint32_t* __restrict__ mlev, fs_status* st) {
  __shared__ uint32_t s_a[4][FS_LEV_MAX + 2], s_b[4][FS_LEV_MAX + 2];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t total = nc.get();
  const uint32_t gw = blockIdx.x * 4 + wave, NWAVES = gridDim.x * 4;
  for (uint32_t t0 = 0; (uint64_t)t0 * NWAVES < total; t0 += 64) {
    const uint64_t cl = (uint64_t)(t0 + lane) * NWAVES + gw;
    bool live = false;
    if (cl < total) {
      const uint32_t gram = cg[cl];
      live = gram != FS_NONE;
      if (live && levtab) {
        const uint32_t p = cpos[cl];
        bool same = true;
        for (int k = 0; k < g.n; ++k) same = same && c.str[p + k] == c.tok[p + k];
        if (same) {
          const uint32_t m = g.gcnt[gram];
          bool all = true;
          for (uint32_t r = 0; r < m; ++r) all = all && levtab[(size_t)gram * g.nn + r] != FS_NONE;
          if (all) {
            for (uint32_t r = 0; r < m; ++r) mlev[cl * g.nn + r] = levtab[(size_t)gram * g.nn + r];
            live = false;
          }
        }
      }
    }
    uint64_t todo = __ballot(live);
    while (todo) {
      const uint32_t cand = (t0 + (uint32_t)(__ffsll((unsigned long long)todo) - 1)) * NWAVES + gw;
      todo &= todo - 1;
      const uint32_t gram = cg[cand], m = g.gcnt[gram], p = cpos[cand];
      for (uint32_t r = 0; r < m; ++r) {                 // wave-uniform
        const uint32_t s = g.gpos[(size_t)gram * g.nn + r];
        const uint32_t v = lev_wave(g, s, c.str + p, c.chars, c.coff, c.n_str, st, s_a[wave], s_b[wave]);
        if (lane == 0) mlev[(size_t)cand * g.nn + r] = v;
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
}

// the string table as classes: {length (capped at 255), classes of code points 0 .. 14}
__global__ void k_strrec(const uint32_t* __restrict__ chars, const uint64_t* __restrict__ coff,
                         uint32_t n_str, StrFast F, uint4* __restrict__ rec) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_str) return;
  const uint64_t a = coff[i], b = coff[i + 1];
  const uint64_t len = b - a;
  uint32_t w[4] = {len < 255 ? (uint32_t)len : 255u, 0u, 0u, 0u};
  for (uint32_t j = 0; j < 15 && j < len; ++j) {
    const uint32_t cl = cls_of(F, chars[a + j]);
    w[(j + 1) >> 2] |= cl << (8 * ((j + 1) & 3));
  }
  rec[i] = make_uint4(w[0], w[1], w[2], w[3]);
}

// Batches with string ids: the record every hit offers (best_of_ranks over its own Levenshtein
// distances).  64 consecutive candidates per wave and step, a lane each: no hit -> nothing;
// every token of the window with string id == vector id -> the n-gram's record of this string
// table (k_gbest); the rest queue up in LDS and are worked off 64 at a time, a lane per hit,
// all its ranks (lev_lane), so that the lanes of the expensive part are all busy.
__global__ __launch_bounds__(256) void k_strbest(GramIndexDev g, CorpusDev c, StrFast F,
                                                 const uint32_t* __restrict__ cpos,
                                                 const uint32_t* __restrict__ cg, NSrc nc,
                                                 const uint32_t* __restrict__ levtab,
                                                 const fs_best* __restrict__ gbest,
                                                 fs_best* __restrict__ cbest, fs_status* st) {
  __shared__ uint32_t s_q[4][128];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t total = nc.get();
  const uint32_t gw = blockIdx.x * 4 + wave, NWAVES = gridDim.x * 4;
  uint32_t head = 0, tail = 0;                             // (wave-uniform) ring of candidate indices
  auto work_off = [&](uint32_t count) {                    // the `count` <= 64 oldest of the queue
    if ((uint32_t)lane < count) {
      const uint32_t i = s_q[wave][(head + lane) & 127];
      const uint32_t gram = cg[i], m = g.gcnt[gram], p = cpos[i];
      fs_best b;
      b.s = 0; b.lev = 0; b.dist = 0.0; b.comb = 0.0; b.pad = 0.0;
      for (uint32_t r = 0; r < m; ++r) {
        const uint32_t s = g.gpos[(size_t)gram * g.nn + r];
        const uint32_t lev = lev_lane(g, c, F, s, c.str + p, st);
        const double dist = g.selfdist[s];
        const double comb = __dmul_rn(dist, (double)lev);
        if (r == 0 || comb < b.comb) { b.s = s; b.lev = lev; b.dist = dist; b.comb = comb; }
      }
      cbest[i] = b;
    }
    head += count;
  };
  for (uint64_t blk = gw; blk * 64 < total; blk += NWAVES) {
    const uint64_t il = blk * 64 + lane;
    bool slow = false;
    if (il < total) {
      const uint32_t gram = cg[il];
      if (gram != FS_NONE) {
        slow = true;
        if (levtab) {
          const uint32_t p = cpos[il];
          bool same = true;
          for (int k = 0; k < g.n; ++k) same = same && c.str[p + k] == c.tok[p + k];
          if (same) {
            const uint32_t m = g.gcnt[gram];
            bool all = true;
            for (uint32_t r = 0; r < m; ++r) all = all && levtab[(size_t)gram * g.nn + r] != FS_NONE;
            if (all) {
              const uint4* src = reinterpret_cast<const uint4*>(gbest + gram);
              uint4* dst = reinterpret_cast<uint4*>(cbest + il);
              dst[0] = src[0]; dst[1] = src[1];
              slow = false;
            }
          }
        }
      }
    }
    const uint64_t sb = __ballot(slow);
    if (slow)
      s_q[wave][(tail + __builtin_amdgcn_mbcnt_hi((uint32_t)(sb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sb, 0u))) & 127] = (uint32_t)il;
    tail += (uint32_t)__popcll(sb);
    __builtin_amdgcn_wave_barrier();
    if (tail - head >= 64) work_off(64);
    __builtin_amdgcn_wave_barrier();
  }
  if (tail != head) work_off(tail - head);
}

// The record a hit offers to every fan word of its window: the first minimum of
// dist*lev over its NearestFilter ranks (all ranks of one hit precede all ranks
// of the next in the reference's insertion order, so the per-word first minimum
// over (hit, rank) pairs equals the first minimum over hits of this record).
//   levs: lev[rank] of the gram (levtab mode) or of the candidate (mlev mode)
__device__ __forceinline__ fs_best best_of_ranks(const GramIndexDev& g, uint32_t gram,
                                                 const uint32_t* levs) {
  fs_best b;
  const uint32_t m = g.gcnt[gram];
  b.s = 0; b.lev = 0; b.dist = 0.0; b.comb = 0.0;
  for (uint32_t r = 0; r < m; ++r) {
    const uint32_t s = g.gpos[(size_t)gram * g.nn + r];
    const double dist = g.selfdist[s];
    const double comb = __dmul_rn(dist, (double)levs[r]);
    if (r == 0 || comb < b.comb) { b.s = s; b.lev = levs[r]; b.dist = dist; b.comb = comb; }
  }
  return b;
}

__global__ void k_gbest(GramIndexDev g, const uint32_t* __restrict__ levtab,
                        fs_best* __restrict__ gbest) {
  for (uint32_t gram = blockIdx.x * blockDim.x + threadIdx.x; gram < g.n_grams;
       gram += gridDim.x * blockDim.x)
    gbest[gram] = best_of_ranks(g, gram, levtab + (size_t)gram * g.nn);
}

__global__ void k_cbest(GramIndexDev g, const uint32_t* __restrict__ cg,
                        const uint32_t* __restrict__ mlev, NSrc nc, fs_best* __restrict__ cbest) {
  const uint32_t total = nc.get();
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += gridDim.x * blockDim.x) {
    const uint32_t gram = cg[i];
    if (gram != FS_NONE) cbest[i] = best_of_ranks(g, gram, mlev + (size_t)i * g.nn);
  }
}

// Records.  Per 256-candidate tile a block scan gives every hit the offset of
// the fan words it is the first to cover; the tile's records are then computed
// one per thread.  A word's record is the first minimum of dist*lev over every
// hit whose window covers the word, in the reference's insertion order
// (ascending window position; search.py:176-218, 224-225), each hit offering
// its best rank (best_of_ranks).
// WIRE: 0 = 32-byte fs_row records, 16 / 8 = wire records (include/fandom_search.h)
template <int WIRE>
__global__ __launch_bounds__(kThreads) void k_rows(GramIndexDev g, CorpusDev c,
                                                   const uint32_t* __restrict__ cpos,
                                                   const uint32_t* __restrict__ cg,
                                                   const uint32_t* __restrict__ cw,
                                                   const uint64_t* __restrict__ hv,
                                                   const double* __restrict__ hcomb,
                                                   const uint64_t* __restrict__ bsum,
                                                   const uint32_t* __restrict__ bmatch,
                                                   const fs_best* __restrict__ best_tab,
                                                   int best_per_cand, NSrc nc, uint32_t rcap,
                                                   fs_row* __restrict__ rows, fs_status* st,
                                                   fs_status* host_st, uint64_t* count_out) {
  __shared__ uint64_t s_w[4];
  __shared__ uint32_t s_w32[4];
  __shared__ uint32_t s_roff[kThreads];
  const uint32_t NC = nc.get();
  const uint32_t n = g.n;
  uint32_t lo, hi;
  chunk_of_block(NC, &lo, &hi);
  // first tile's values requested before the barriers of the prefix sums
  uint64_t v_first = lo + threadIdx.x < hi ? hv[lo + threadIdx.x] : 0;
  uint64_t carry = block_prefix(bsum, s_w);
  if (blockIdx.x == kNB) {                 // the extra block: its prefix is the total
    const uint64_t total = carry;
    uint32_t m = 0, mt;
    for (int i = threadIdx.x; i < kNB; i += kThreads) m += bmatch[i];
    block_excl_scan(m, s_w32, &mt);
    if (threadIdx.x == 0) {
      // every earlier kernel of the chain has finished: publish the totals and
      // flags straight into the caller's pinned block (no copy node afterwards)
      fs_status out = *st;
      out.n_hits = (uint32_t)total;
      out.n_rows = (uint32_t)(total >> 32);
      out.n_matches = mt;
      *st = out;
      *host_st = out;
      if (count_out) *count_out = out.n_rows;    // FS_ROWS_HEADER
    }
  }
  for (uint32_t t0 = lo; t0 < hi; t0 += kThreads) {
    const uint32_t i = t0 + threadIdx.x;
    const uint64_t v = t0 == lo ? v_first : (i < hi ? hv[i] : 0);
    uint64_t tile_total;
    const uint64_t ex = block_excl_scan(v, s_w, &tile_total);
    s_roff[threadIdx.x] = (uint32_t)(ex >> 32);
    __syncthreads();
    const uint32_t tile_rows = (uint32_t)(tile_total >> 32);
    const uint32_t rbase = (uint32_t)(carry >> 32);
    for (uint32_t r = threadIdx.x; r < tile_rows; r += kThreads) {
      // owner: last tile entry whose exclusive offset is <= r (a hit: entries
      // that are not hits share the offset of the hit that follows them)
      uint32_t a = 0, b = kThreads;
      while (b - a > 1) {
        const uint32_t mid = (a + b) >> 1;
        if (s_roff[mid] <= r) a = mid; else b = mid;
      }
      const uint32_t own = t0 + a;
      const uint32_t k = r - s_roff[a];
      const uint32_t ridx = rbase + r;
      if (ridx >= rcap) continue;
      const uint32_t p = cpos[own];
      const uint32_t w = cw[own];
      const uint64_t wbase = c.work_off[w];
      // words first covered by `own`: the last cnt of its window; cnt from the
      // next entry's offset
      const uint32_t next_off = a + 1 < kThreads ? s_roff[a + 1] : tile_rows;
      const uint32_t cnt = next_off - s_roff[a];
      const uint32_t x = p + n - cnt + k;          // global token position of the word
      // first minimum of the combined distance over the hits covering x (candidates
      // own, own+1, ... while they start at or before x; non-hits carry +inf)
      uint32_t jbest = own, pbest = p;
      double cbest = hcomb[own];
      for (uint32_t j = own + 1; j < NC; ++j) {
        const uint32_t p2 = cpos[j];
        if (p2 > x) break;
        const double cj = hcomb[j];
        if (cj < cbest) { cbest = cj; jbest = j; pbest = p2; }
      }
      const fs_best bb = best_tab[best_per_cand ? jbest : cg[jbest]];
      fs_row out;
      const uint32_t koff = x - pbest;
      out.orig_ix = bb.s + koff;
      out.lev = bb.lev;
      out.dist = bb.dist;
      out.comb = bb.comb;
      out.work = w;
      out.fan_ix = (uint32_t)((uint64_t)x - wbase);
      if (WIRE == 16) {
        // 16-byte wire record {work, fan_ix, orig_ix, lev | k << 16}: k = offset of the
        // word inside the matched window, so that dist = selfdist[orig_ix - k]
        uint4 q;
        q.x = out.work; q.y = out.fan_ix; q.z = out.orig_ix; q.w = out.lev | (koff << 16);
        reinterpret_cast<uint4*>(rows)[ridx] = q;
        continue;
      }
      if (WIRE == 8) {
        // 8-byte wire record {token position, orig_ix | k << 18 | lev << 22}
        reinterpret_cast<uint2*>(rows)[ridx] = make_uint2(x, out.orig_ix | (koff << 18) | (out.lev << 22));
        continue;
      }
      // one 32-byte record = two 16-byte stores (row buffers are 16-byte aligned)
      uint4 q0, q1;
      q0.x = out.work; q0.y = out.fan_ix; q0.z = out.orig_ix; q0.w = out.lev;
      const uint64_t db = (uint64_t)__double_as_longlong(out.dist);
      const uint64_t cb = (uint64_t)__double_as_longlong(out.comb);
      q1.x = (uint32_t)db; q1.y = (uint32_t)(db >> 32); q1.z = (uint32_t)cb; q1.w = (uint32_t)(cb >> 32);
      uint4* dst = reinterpret_cast<uint4*>(rows + ridx);
      dst[0] = q0; dst[1] = q1;
    }
    __syncthreads();
    carry += tile_total;
  }
}

// 16-byte wire records -> full records (on the rank that gathers them)
__global__ void k_unpack(const uint4* __restrict__ packed, uint64_t n,
                         const double* __restrict__ selfdist, fs_row* __restrict__ rows) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const uint4 q = packed[i];
    const uint32_t lev = q.w & 0xFFFFu, k = q.w >> 16;
    const double dist = selfdist[q.z - k];
    const double comb = __dmul_rn(dist, (double)lev);
    uint4 q0, q1;
    q0.x = q.x; q0.y = q.y; q0.z = q.z; q0.w = lev;
    const uint64_t db = (uint64_t)__double_as_longlong(dist);
    const uint64_t cb = (uint64_t)__double_as_longlong(comb);
    q1.x = (uint32_t)db; q1.y = (uint32_t)(db >> 32); q1.z = (uint32_t)cb; q1.w = (uint32_t)(cb >> 32);
    uint4* dst = reinterpret_cast<uint4*>(rows + i);
    dst[0] = q0; dst[1] = q1;
  }
}

// 8-byte wire records -> full records: the work is the last one starting at or
// before the token position (binary search over the batch's offsets)
__global__ void k_unpack8(const uint2* __restrict__ packed, uint64_t n,
                          const uint64_t* __restrict__ work_off, uint32_t n_works,
                          const double* __restrict__ selfdist, fs_row* __restrict__ rows) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const uint2 q = packed[i];
    const uint32_t x = q.x, orig = q.y & 0x3FFFFu, k = (q.y >> 18) & 0xFu, lev = q.y >> 22;
    uint32_t lo = 0, hi = n_works;               // work_off[lo] <= x < work_off[hi]
    while (hi - lo > 1) {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      if (work_off[mid] <= x) lo = mid; else hi = mid;
    }
    const double dist = selfdist[orig - k];
    const double comb = __dmul_rn(dist, (double)lev);
    uint4 q0, q1;
    q0.x = lo; q0.y = (uint32_t)(x - work_off[lo]); q0.z = orig; q0.w = lev;
    const uint64_t db = (uint64_t)__double_as_longlong(dist);
    const uint64_t cb = (uint64_t)__double_as_longlong(comb);
    q1.x = (uint32_t)db; q1.y = (uint32_t)(db >> 32); q1.z = (uint32_t)cb; q1.w = (uint32_t)(cb >> 32);
    uint4* dst = reinterpret_cast<uint4*>(rows + i);
    dst[0] = q0; dst[1] = q1;
  }
}

// ---- `format` aggregation -----------------------------------------------------
// first[word][b] += 1 where b is the first threshold with comb <= t_b (n_thr if none)
template <class GetIx, class GetComb>
__device__ __forceinline__ void hist_body(GetIx ix_of, GetComb comb_of, uint64_t n_rows,
                                          uint64_t n_script, const double* thr, uint32_t n_thr,
                                          uint32_t* first) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t o = ix_of(i);
    if (o >= n_script) continue;
    const double c = comb_of(i);
    uint32_t b = n_thr;
    for (uint32_t t = 0; t < n_thr; ++t)
      if (c <= thr[t]) { b = t; break; }
    atomicAdd(&first[(size_t)o * (n_thr + 1) + b], 1u);
  }
}

__global__ void k_hist_cols(const uint32_t* __restrict__ orig_ix, const double* __restrict__ comb,
                            uint64_t n_rows, uint64_t n_script, const double* __restrict__ thr,
                            uint32_t n_thr, uint32_t* __restrict__ first) {
  hist_body([&](uint64_t i) { return orig_ix[i]; }, [&](uint64_t i) { return comb[i]; }, n_rows,
            n_script, thr, n_thr, first);
}

__global__ void k_hist_rows(const fs_row* __restrict__ rows, uint64_t n_rows, uint64_t n_script,
                            const double* __restrict__ thr, uint32_t n_thr,
                            uint32_t* __restrict__ first) {
  hist_body([&](uint64_t i) { return rows[i].orig_ix; }, [&](uint64_t i) { return rows[i].comb; },
            n_rows, n_script, thr, n_thr, first);
}

// counts[word][b] = records with comb <= t_b = sum of first[word][0..b]; column
// n_thr = all records of the word
__global__ void k_hist_cumulate(uint32_t* __restrict__ counts, uint64_t n_script, uint32_t n_thr) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_script) return;
  uint32_t* c = counts + w * (n_thr + 1);
  uint32_t acc = 0;
  for (uint32_t b = 0; b <= n_thr; ++b) { acc += c[b]; c[b] = acc; }
}

// work of the first token of every 256-token block and where that work ends
// (corpus build time)
__global__ void k_blk_work(const uint64_t* __restrict__ work_off, uint32_t n_works,
                           uint32_t n_blocks, uint2* __restrict__ blk_work,
                           uint4* __restrict__ blk4) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_blocks) return;
  const uint64_t p = (uint64_t)b * 256;
  uint32_t lo = 0, hi = n_works;   // work_off[lo] <= p < work_off[hi]
  while (hi - lo > 1) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (work_off[mid] <= p) lo = mid; else hi = mid;
  }
  blk_work[b] = make_uint2(lo, (uint32_t)work_off[lo + 1]);      // a batch holds < 2^32 tokens
  blk4[b] = make_uint4(lo, (uint32_t)work_off[lo], (uint32_t)work_off[lo + 1],
                       (uint32_t)work_off[lo + 2 < n_works ? lo + 2 : n_works]);
}

// validation of an uploaded batch: largest embedding row id + 1, "any OOV id",
// largest string id + 1 (string id == vector id when str is null)
__global__ __launch_bounds__(256) void k_corpus_check(const uint32_t* __restrict__ tok,
                                                      const uint32_t* __restrict__ str,
                                                      uint32_t n_tok, uint32_t* __restrict__ check) {
  uint32_t max_row = 0, oov = 0, max_str = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_tok; i += gridDim.x * blockDim.x) {
    const uint32_t id = tok[i];
    if (id & FS_OOV_FLAG) oov = 1; else max_row = max(max_row, id + 1);
    const uint32_t sid = str ? str[i] : id;
    if (str || !(id & FS_OOV_FLAG)) max_str = max(max_str, sid + 1);
  }
  for (int d = 32; d > 0; d >>= 1) {
    max_row = max(max_row, (uint32_t)__shfl_xor(max_row, d));
    max_str = max(max_str, (uint32_t)__shfl_xor(max_str, d));
    oov |= __shfl_xor(oov, d);
  }
  __shared__ uint32_t s_m[3][4];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (lane == 0) { s_m[0][wave] = max_row; s_m[1][wave] = oov; s_m[2][wave] = max_str; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t a = 0, b = 0, c = 0;
    for (int w = 0; w < 4; ++w) { a = max(a, s_m[0][w]); b |= s_m[1][w]; c = max(c, s_m[2][w]); }
    if (a) atomicMax(&check[0], a);
    if (b) atomicOr(&check[1], b);
    if (c) atomicMax(&check[2], c);
  }
}

}  // namespace

int fs_launch_corpus_check(const uint32_t* tok, const uint32_t* str, uint32_t n_tok,
                           uint32_t* check, hipStream_t s) {
  if (!n_tok) return FS_OK;
  uint32_t blocks = (n_tok + 256 * 16 - 1) / (256 * 16);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(k_corpus_check, dim3(blocks), dim3(256), 0, s, tok, str, n_tok, check);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

int fs_launch_histogram(const uint32_t* d_orig, const double* d_comb, const fs_row* d_rows,
                        uint64_t n_rows, uint64_t n_script, const double* d_thr, uint32_t n_thr,
                        uint32_t* d_counts, hipStream_t s) {
  FS_HIP(hipMemsetAsync(d_counts, 0, n_script * (n_thr + 1) * sizeof(uint32_t), s));
  if (n_rows) {
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n_rows + 255) / 256, 4096);
    if (d_rows)
      hipLaunchKernelGGL(k_hist_rows, dim3(blocks), dim3(256), 0, s, d_rows, n_rows, n_script, d_thr,
                         n_thr, d_counts);
    else
      hipLaunchKernelGGL(k_hist_cols, dim3(blocks), dim3(256), 0, s, d_orig, d_comb, n_rows,
                         n_script, d_thr, n_thr, d_counts);
  }
  if (n_script)
    hipLaunchKernelGGL(k_hist_cumulate, dim3((uint32_t)((n_script + 255) / 256)), dim3(256), 0, s,
                       d_counts, n_script, n_thr);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

int fs_launch_blk_work(const uint64_t* work_off, uint32_t n_works, uint32_t n_blocks,
                       uint2* blk_work, uint4* blk4, hipStream_t s) {
  if (!n_blocks || !n_works) return FS_OK;
  hipLaunchKernelGGL(k_blk_work, dim3((n_blocks + 255) / 256), dim3(256), 0, s, work_off, n_works,
                     n_blocks, blk_work, blk4);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

int fs_launch_strrec(fs_index* ix, fs_corpus* c, hipStream_t s) {
  FS_TRY(c->d_strrec.reserve(c->n_str + 1));
  if (c->n_str) {
    const StrFast F{ix->d_pat.p, ix->d_clsmap.p, ix->n_cls, ix->str_punct, nullptr};
    hipLaunchKernelGGL(k_strrec, dim3((uint32_t)((c->n_str + 255) / 256)), dim3(256), 0, s, c->d_chars.p,
                       c->d_coff.p, (uint32_t)c->n_str, F, c->d_strrec.p);
    FS_HIP(hipGetLastError());
  }
  return FS_OK;
}

int fs_launch_levtab(fs_index* ix, fs_corpus* c, hipStream_t s) {
  const size_t total = (size_t)ix->n_grams * ix->cfg.nearest_n;
  FS_TRY(c->d_levtab.reserve(total));
  FS_TRY(c->d_gbest.reserve(ix->n_grams));
  if (total) {
    const uint32_t blocks = (uint32_t)((total + 3) / 4);
    hipLaunchKernelGGL(k_levtab, dim3(blocks > 4096 ? 4096 : blocks), dim3(256), 0, s,
                       ix->gram_dev(), c->dev(), c->d_levtab.p, ix->cur->d_status.p, c->has_str);
    const uint32_t gb = (ix->n_grams + 255) / 256;
    hipLaunchKernelGGL(k_gbest, dim3(gb > 1024 ? 1024 : gb), dim3(256), 0, s, ix->gram_dev(),
                       c->d_levtab.p, c->d_gbest.p);
    FS_HIP(hipGetLastError());
  }
  return FS_OK;
}

// bitmap + counts -> candidate positions (w_cpos), n_cands in the status block
// counted: the scan already wrote the chunk sums and cleared the status block
int fs_launch_expand(fs_index* ix, fs_corpus* c, uint32_t n_sub, uint32_t ccap, int tpl,
                     hipStream_t s, bool counted) {
  uint32_t* bsum32 = ix->cur->w_bsum.p;
  if (!counted)
    hipLaunchKernelGGL((k_reduce<SubTileCountF, uint32_t>), dim3(kNB), dim3(kThreads), 0, s,
                       SubTileCountF{ix->cur->w_qcnt.p}, NSrc{nullptr, 0, 0, n_sub}, bsum32,
                       ix->cur->d_status.p);
  (void)c;
#define FS_EXPAND(T)                                                                          \
  hipLaunchKernelGGL((k_expand<T>), dim3(kNB + 1), dim3(kThreads), 0, s, ix->cur->w_qbm.p,    \
                     ix->cur->w_qcnt.p, n_sub, bsum32, ix->cur->w_cpos.p, ccap, ix->cur->d_status.p)
  if (tpl == 8) FS_EXPAND(8); else FS_EXPAND(4);
#undef FS_EXPAND
  FS_HIP(hipGetLastError());
  return FS_OK;
}

// verified candidates (w_cpos, w_cg, w_cw) + best records -> output rows
int fs_launch_unpack(fs_index* ix, const void* packed, uint64_t n, fs_row* rows, hipStream_t s) {
  if (!n) return FS_OK;
  const uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(k_unpack, dim3(blocks), dim3(256), 0, s,
                     reinterpret_cast<const uint4*>(packed), n, ix->d_selfdist.p, rows);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

int fs_launch_unpack8(fs_index* ix, const void* packed, uint64_t n, const uint64_t* work_off,
                      uint64_t n_works, fs_row* rows, hipStream_t s) {
  if (!n) return FS_OK;
  const uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(k_unpack8, dim3(blocks), dim3(256), 0, s,
                     reinterpret_cast<const uint2*>(packed), n, work_off, (uint32_t)n_works,
                     ix->d_selfdist.p, rows);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

// wire: 0 (fs_row), 16 or 8 bytes per record
int fs_launch_rows(fs_index* ix, fs_corpus* c, const fs_best* best_tab, int best_per_cand,
                   uint32_t ccap, uint32_t rcap, fs_row* d_rows, int wire, fs_status* host_st,
                   hipStream_t s, uint64_t* count_out) {
  fs_status* st = ix->cur->d_status.p;
  const NSrc nc{&st->n_cands, 1, ccap, 0};
  uint32_t* bmatch = ix->cur->w_bsum.p + kNB;
  uint64_t* bsum64 = ix->cur->w_bsum64.p;
  hipLaunchKernelGGL(k_hitrows, dim3(kNB), dim3(kThreads), 0, s,
                     HitRowsF{ix->cur->w_cpos.p, ix->cur->w_cg.p, ix->cfg.window_size}, nc, best_tab,
                     best_per_cand, ix->cur->w_hv.p, ix->cur->w_hcomb.p, bsum64);
#define FS_ROWS(P)                                                                           \
  hipLaunchKernelGGL((k_rows<P>), dim3(kNB + 1), dim3(kThreads), 0, s, ix->gram_dev(), c->dev(),   \
                     ix->cur->w_cpos.p, ix->cur->w_cg.p, ix->cur->w_cw.p, ix->cur->w_hv.p,   \
                     ix->cur->w_hcomb.p, bsum64, bmatch, best_tab, best_per_cand, nc, rcap,  \
                     d_rows, st, host_st, count_out)
  if (wire == 16) FS_ROWS(16); else if (wire == 8) FS_ROWS(8); else FS_ROWS(0);
#undef FS_ROWS
  FS_HIP(hipGetLastError());
  return FS_OK;
}

int fs_launch_post(fs_index* ix, fs_corpus* c, uint32_t n_sub, int tpl, uint32_t ccap,
                   uint32_t rcap, fs_row* d_rows, int wire, fs_status* host_st,
                   hipStream_t s, const fs_scan_extra& scan, uint64_t* count_out) {
  const GramIndexDev g = ix->gram_dev();
  const CorpusDev cd = c->dev();
  fs_status* st = ix->cur->d_status.p;
  uint32_t* bmatch = ix->cur->w_bsum.p + kNB;
  const bool per_cand = c->has_str;
  const NSrc nc{&st->n_cands, 1, ccap, 0};
  if (scan.direct) {
    // the scan wrote candidate records per wave range: no expand kernel
    hipLaunchKernelGGL(k_verify_direct, dim3(kNB + 1), dim3(kThreads), 0, s, cd, g, scan.bsum,
                       scan.recs, scan.info, scan.capw, ccap, ix->cur->w_cpos.p,
                       ix->cur->w_cg.p, ix->cur->w_cw.p, bmatch, st);
  } else {
    FS_TRY(fs_launch_expand(ix, c, n_sub, ccap, tpl, s, scan.counted));
    hipLaunchKernelGGL(k_verify, dim3(kNB), dim3(kThreads), 0, s, cd, g, ix->cur->w_cpos.p, nc,
                       ix->cur->w_cg.p, ix->cur->w_cw.p, bmatch);
  }
  if (per_cand && c->strrec_ready && ix->strfast_ok && ix->sw.str_fast) {
    const StrFast F{ix->d_pat.p, ix->d_clsmap.p, ix->n_cls, ix->str_punct, c->d_strrec.p};
    hipLaunchKernelGGL(k_strbest, dim3(kNB), dim3(kThreads), 0, s, g, cd, F, ix->cur->w_cpos.p,
                       ix->cur->w_cg.p, nc,
                       c->levtab_ready && ix->sw.str_levtab ? (const uint32_t*)c->d_levtab.p : nullptr,
                       c->d_gbest.p, ix->cur->w_cbest.p, st);
  } else if (per_cand) {
    hipLaunchKernelGGL(k_matchlev, dim3(kNB), dim3(kThreads), 0, s, g, cd, ix->cur->w_cpos.p,
                       ix->cur->w_cg.p, nc,
                       c->levtab_ready && ix->sw.str_levtab ? (const uint32_t*)c->d_levtab.p : nullptr,
                       ix->cur->w_mlev.p, st);
    hipLaunchKernelGGL(k_cbest, dim3(kNB), dim3(kThreads), 0, s, g, ix->cur->w_cg.p, ix->cur->w_mlev.p, nc,
                       ix->cur->w_cbest.p);
  }
  FS_HIP(hipGetLastError());
  return fs_launch_rows(ix, c, per_cand ? ix->cur->w_cbest.p : c->d_gbest.p, per_cand ? 1 : 0, ccap,
                        rcap, d_rows, wire, host_st, s, count_out);
}
