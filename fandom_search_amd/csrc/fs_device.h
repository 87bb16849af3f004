// fs_device.h -- device helpers shared by fs_post.hip and fs_lsh.hip.
#pragma once
#include "fs_internal.h"
#include <type_traits>

namespace fsdev {

constexpr int kNB = FS_CHUNKS;        // blocks of every chunked kernel (and partial sums)
constexpr int kThreads = 256;

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

// ---- wave-wide helpers (DPP) -------------------------------------------------

// sum over the wave, complete in lane 63 (six DPP adds)
__device__ __forceinline__ uint32_t wave_sum_lane63(uint32_t x) {
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, false);   // row_half_mirror
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xF, 0xF, false);   // row_mirror
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1, 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2, 3
  return x;
}

// inclusive prefix sum over the wave (four row_shr adds inside each row of 16 lanes,
// then the two row broadcasts)
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t x) {
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);    // row_shr:1
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);    // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);    // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);    // row_shr:8
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1, 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2, 3
  return x;
}

// word >> (bits 8B .. 8B+4 of h): the shift takes the low five bits of the selected byte
template <int B>
__device__ __forceinline__ uint32_t shr_by_byte(uint32_t word, uint32_t h) {
  uint32_t r;
  if constexpr (B == 0)
    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(h), "v"(word));
  else if constexpr (B == 1)
    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(h), "v"(word));
  else if constexpr (B == 2)
    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(h), "v"(word));
  else
    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(h), "v"(word));
  return r;
}


// ---- block-wide helpers ----------------------------------------------------

template <class V>
__device__ __forceinline__ V wave_incl_scan(V v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    V t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}

// exclusive scan of one value per thread over a 256-thread block; returns the
// thread's exclusive prefix, *total = block sum.  s_w: 4 values of LDS.
template <class V>
__device__ __forceinline__ V block_excl_scan(V v, V* s_w, V* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const V inc = wave_incl_scan(v, lane);
  if (lane == 63) s_w[w] = inc;
  __syncthreads();
  V base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < kThreads / 64; ++i) {
    const V x = s_w[i];
    if (i < w) base += x;
    tot += x;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// Sum of the partial sums of all blocks before this one: block b reads b values.
// Kernels that also need the grand total are launched with kNB + 1 blocks; the
// extra block (index kNB) has an empty chunk, its prefix is the total, and it does
// the bookkeeping beside the others instead of in front of block 0's work.
template <class V>
__device__ __forceinline__ V block_prefix(const V* __restrict__ bsum, V* s_w) {
  V pre = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x && i < kNB; i += kThreads) pre += bsum[i];
  V t_pre;
  block_excl_scan(pre, s_w, &t_pre);
  return t_pre;
}

__device__ __forceinline__ void chunk_of_block(uint32_t n, uint32_t* lo, uint32_t* hi) {
  const uint32_t chunk = (n + kNB - 1) / kNB;
  const uint64_t l = (uint64_t)blockIdx.x * chunk;
  *lo = l < n ? (uint32_t)l : n;
  const uint64_t h = l + chunk;
  *hi = h < n ? (uint32_t)h : n;
}

// `zero` (optional): the per-search status block, cleared here because this is the
// first kernel of the chain that may touch it (saves a memset node per search)
template <class F, class V>
__global__ __launch_bounds__(kThreads) void k_reduce(F f, NSrc ns, V* __restrict__ bsum,
                                                     fs_status* zero) {
  __shared__ V s_w[4];
  if (zero && blockIdx.x == 0 && threadIdx.x == 0) {
    zero->n_cands = 0; zero->n_hits = 0; zero->n_matches = 0; zero->n_rows = 0;
    zero->max_recs = 0; zero->lev_overflow = 0; zero->bad_string = 0; zero->max_rows = 0;
    zero->lsh_pending = 0;
  }
  uint32_t lo, hi;
  chunk_of_block(ns.get(), &lo, &hi);
  V acc = 0;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += kThreads) acc += f(i);
  V tot;
  block_excl_scan(acc, s_w, &tot);
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}


// work of token p and the end of that work: one 8-byte read of the block table
// unless a work boundary lies between the block's first token and p
__device__ __forceinline__ uint32_t work_of_token(const CorpusDev& c, uint64_t p, uint64_t* end) {
  const uint2 e = c.blk_work[p >> 8];
  uint32_t w = e.x;
  uint64_t we = e.y;
  while (we <= p) { ++w; we = c.work_off[w + 1]; }
  *end = we;
  return w;
}

// ids of a window: two (n <= 8) or four 16-byte loads.  The window start is only
// 4-byte aligned; gfx950 global loads accept that (unaligned access mode), and
// the token buffers are padded, so reading up to 16 ids is always in bounds.
struct Ids16 { uint32_t v[16]; };
__device__ __forceinline__ void load_ids(const uint32_t* p, int n, Ids16* out) {
  __builtin_memcpy(out->v, p, 32);
  if (n > 8) __builtin_memcpy(out->v + 8, p + 8, 32);
}

// Exact check of the window at token p: the script n-gram with the same n vector
// ids (found through the open-addressing table, compared id for id), or FS_NONE;
// FS_NONE also for a window that crosses a work boundary (windows are built per
// file, search.py:170-173).  *w = the work, *kept = NearestFilter entries it has.
__device__ __forceinline__ uint32_t verify_window(const CorpusDev& c, const GramIndexDev& g,
                                                  uint64_t p, uint32_t* w_out, uint32_t* kept) {
  if (p + g.n > c.n_tok) return FS_NONE;
  const uint32_t slot_mask = (1u << g.log2_slots) - 1;
  // two independent chains of two reads each: ids -> table entry (which carries the
  // script n-gram's ids), and block -> work
  Ids16 f;
  load_ids(c.tok + p, g.n, &f);
  uint64_t work_end;
  const uint32_t w = work_of_token(c, p, &work_end);
  const bool inside = p + g.n <= work_end;
  uint32_t h = 0;
#pragma unroll
  for (int k = 0; k < FS_MAX_WINDOW; ++k)
    if (k < g.n) h ^= fs_rotl(fs_premix(f.v[k]), fs_rot_of(g.n - 1 - k));
  // hash-and-displace: the bucket's displacement seed sends every script n-gram of the
  // bucket to a slot of its own, so one entry read decides (hit, or not a script
  // n-gram).  Only buckets holding n-grams with identical 32-bit hashes carry the
  // overflow flag and continue by linear probing.
  const uint32_t d = g.disp[fs_table_bucket(h, g.log2_buckets)];
  uint32_t slot = fs_table_slot_d(h, d & FS_DISP_MASK, g.log2_slots);
  const int quads = g.tstride >> 2;              // 16-byte pieces of an entry: 2 (n <= 6) .. 5
  for (;;) {
    // {gram + 1 (0 = empty), occurrences kept, ids[n], pad}
    const uint4* e = reinterpret_cast<const uint4*>(g.table + (size_t)slot * g.tstride);
    uint32_t ew[20];
#pragma unroll
    for (int q = 0; q < 5; ++q)
      if (q < quads) {
        const uint4 t = e[q];
        ew[4 * q] = t.x; ew[4 * q + 1] = t.y; ew[4 * q + 2] = t.z; ew[4 * q + 3] = t.w;
      }
    if (ew[0] == 0) return FS_NONE;
    bool same = true;
#pragma unroll
    for (int k = 0; k < FS_MAX_WINDOW; ++k)
      if (k < g.n) same = same && (ew[2 + k] == f.v[k]);
    if (same) {
      if (!inside) return FS_NONE;
      *w_out = w;
      *kept = ew[1];
      return ew[0] - 1;
    }
    if (!(d & FS_DISP_OVERFLOW)) return FS_NONE;
    slot = (slot + 1) & slot_mask;
  }
}

// Levenshtein.distance(match_str, fan_context), search.py:189-190:
//   match_str   = script words s .. s+n-1 joined by single spaces
//   fan_context = '[' + ', '.join(fan token texts) + ']'
// unit costs over code points.  One thread; the fan operand and one DP row live in
// scratch (FS_LEV_MAX code points), the script operand is walked word by word.
__device__ inline uint32_t lev_device(const GramIndexDev& g, uint32_t s, const uint32_t* fan_sid,
                                      const uint32_t* chars, const uint64_t* coff, uint32_t n_str,
                                      fs_status* st) {
  uint32_t b[FS_LEV_MAX];
  uint16_t row[FS_LEV_MAX + 1];
  uint32_t lb = 0;
  if (lb < FS_LEV_MAX) b[lb] = '[';
  ++lb;
  for (int k = 0; k < g.n; ++k) {
    if (k) {
      if (lb < FS_LEV_MAX) b[lb] = ','; ++lb;
      if (lb < FS_LEV_MAX) b[lb] = ' '; ++lb;
    }
    const uint32_t sid = fan_sid[k];
    if (sid >= n_str) { st->bad_string = 1; return 0; }
    for (uint64_t c = coff[sid]; c < coff[sid + 1]; ++c) {
      if (lb < FS_LEV_MAX) b[lb] = chars[c];
      ++lb;
    }
  }
  if (lb < FS_LEV_MAX) b[lb] = ']';
  ++lb;
  const uint64_t la = g.soff[s + g.n] - g.soff[s] + (uint64_t)(g.n - 1);
  if (la > FS_LEV_MAX || lb > FS_LEV_MAX) { st->lev_overflow = 1; return 0; }
  for (uint32_t j = 0; j <= lb; ++j) row[j] = (uint16_t)j;
  uint32_t x = 0;
  for (int k = 0; k < g.n; ++k) {
    // characters of script word k, preceded by the joining space
    const uint64_t c0 = g.soff[s + k], c1 = g.soff[s + k + 1];
    for (uint64_t c = k ? c0 - 1 : c0; c < c1; ++c) {
      const uint32_t ca = (k && c == c0 - 1) ? (uint32_t)' ' : g.schars[c];
      ++x;
      uint32_t diag = row[0];
      row[0] = (uint16_t)x;
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
  }
  return row[lb];
}

// The same distance computed by a whole wave (all 64 lanes must call it with the
// same arguments).  Both operands are assembled in LDS (`s_a`, `s_b`: FS_LEV_MAX
// code points each, private to the wave); when the shorter one has at most 64 code
// points the distance comes from Myers' bit-vector recurrence: lane j holds
// pattern[j], __ballot(pattern == c) is the match mask of text character c, and
// the column update is a dozen 64-bit operations.  Longer patterns take the
// scratch DP of lev_device on lane 0.
__device__ inline uint32_t lev_wave(const GramIndexDev& g, uint32_t s, const uint32_t* fan_sid,
                                    const uint32_t* chars, const uint64_t* coff, uint32_t n_str,
                                    fs_status* st, uint32_t* s_a, uint32_t* s_b) {
  const int lane = threadIdx.x & 63;
  const int n = g.n;
  // lane k < n: where word k of either side starts and how long it is (all offsets
  // requested together; the n <= 16 words of a window)
  uint64_t a0 = 0, b0 = 0;
  uint32_t wla = 0, wlb = 0;
  bool bad = false;
  if (lane < n) {
    a0 = g.soff[s + lane];
    const uint64_t a1 = g.soff[s + lane + 1];
    const uint32_t sid = fan_sid[lane];
    bad = sid >= n_str;
    if (!bad) {
      b0 = coff[sid];
      wlb = (uint32_t)(coff[sid + 1] - b0);
    }
    wla = (uint32_t)(a1 - a0);
  }
  if (__any(bad)) { if (lane == 0) st->bad_string = 1; return 0; }
  const uint32_t ia = wave_incl_scan_dpp(wla), ib = wave_incl_scan_dpp(wlb);
  const uint32_t la = (uint32_t)(n - 1) + (uint32_t)__builtin_amdgcn_readlane((int)ia, 63);
  const uint32_t lb = 2u + 2u * (uint32_t)(n - 1) + (uint32_t)__builtin_amdgcn_readlane((int)ib, 63);
  if (la > FS_LEV_MAX || lb > FS_LEV_MAX) { if (lane == 0) st->lev_overflow = 1; return 0; }
  // operands into LDS: a = script words joined by ' ', b = '[' + ', '.join(fan) + ']';
  // lane j takes code points j, j + 64, ... of each operand (one load per code point, all
  // of a lane's requests independent of each other)
  const uint32_t pa = (uint32_t)lane + (ia - wla);            // first position of word `lane` in a
  const uint32_t pb = 1u + 2u * (uint32_t)lane + (ib - wlb);  //                              in b
  const uint32_t a0lo = (uint32_t)a0, a0hi = (uint32_t)(a0 >> 32);
  const uint32_t b0lo = (uint32_t)b0, b0hi = (uint32_t)(b0 >> 32);
  for (uint32_t j = lane; j < la; j += 64) {
    uint64_t at = 0;
    bool in_word = false;
    for (int k = 0; k < n; ++k) {
      const uint32_t st_k = (uint32_t)__builtin_amdgcn_readlane((int)pa, k);
      const uint32_t len_k = (uint32_t)__builtin_amdgcn_readlane((int)wla, k);
      const uint64_t src = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)a0lo, k) |
                           ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)a0hi, k) << 32);
      if (j - st_k < len_k) { at = src + (j - st_k); in_word = true; }
    }
    s_a[j] = in_word ? g.schars[at] : (uint32_t)' ';
  }
  for (uint32_t j = lane; j < lb; j += 64) {
    uint32_t ch = j == 0 ? '[' : ']';
    uint64_t at = 0;
    bool in_word = false;
    for (int k = 0; k < n; ++k) {
      const uint32_t st_k = (uint32_t)__builtin_amdgcn_readlane((int)pb, k);
      const uint32_t len_k = (uint32_t)__builtin_amdgcn_readlane((int)wlb, k);
      const uint64_t src = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)b0lo, k) |
                           ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)b0hi, k) << 32);
      if (j - st_k < len_k) { at = src + (j - st_k); in_word = true; }
      else if (k + 1 < n && j == st_k + len_k) ch = ',';
      else if (k + 1 < n && j == st_k + len_k + 1) ch = ' ';
    }
    s_b[j] = in_word ? chars[at] : ch;
  }
  __builtin_amdgcn_wave_barrier();
  // the pattern sits in the lanes (at most 64 code points), the text is walked column by column
  // -- a column is ~25 scalar instructions, and a CU has one scalar unit for all its waves: with
  // both operands within 64 the LONGER one is the pattern (the distance is symmetric)
  const bool a_pat = (la <= 64 && lb <= 64) ? la >= lb : la <= lb;
  const uint32_t* pat = a_pat ? s_a : s_b;
  const uint32_t* txt = a_pat ? s_b : s_a;
  const uint32_t m = a_pat ? la : lb, t = a_pat ? lb : la;
  if (m == 0) return t;
  if (t == 0) return m;
  if (m > 64) {
    uint32_t r = 0;
    if (lane == 0) r = lev_device(g, s, fan_sid, chars, coff, n_str, st);
    return __shfl(r, 0);
  }
  const uint32_t pc = lane < (int)m ? pat[lane] : 0xFFFFFFFFu;    // never equal to a code point
  uint64_t pv = ~0ull, mv = 0;
  uint32_t score = m;
  // (the text 64 code points at a time in a register, a lane each, and read out lane by lane: an
  // LDS read per column was most of a column's time.  The column's +1 / -1 -- bit m - 1 of ph and
  // of mh -- is not counted column by column: the word that holds the bit goes into lane i of a
  // register, two vector instructions where the test, the select and the add were eight scalar
  // ones, and the lanes are counted once per 64 columns)
  const uint32_t bit = (m - 1) & 31u;
  auto columns = [&](auto hi_word) {
    constexpr bool HI = decltype(hi_word)::value;
    for (uint32_t base = 0; base < t; base += 64) {
      const uint32_t tv = base + (uint32_t)lane < t ? txt[base + lane] : 0u;
      const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t - base < 64u ? t - base : 64u));
      uint32_t accp = 0, accm = 0;
      for (uint32_t i = 0; i < cnt; ++i) {
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)tv, (int)i);
        const uint64_t eq = __ballot(pc == c);
        const uint64_t xv = eq | mv;
        const uint64_t xh = (((eq & pv) + pv) ^ pv) | eq;
        uint64_t ph = mv | ~(xh | pv);
        uint64_t mh = pv & xh;
        // (gfx9: one SGPR per vector instruction, so the lane number travels in m0, which is
        // put back: the compiler does not track it through an asm statement)
        uint32_t m0_keep;
        asm("s_mov_b32 %2, m0\n\ts_mov_b32 m0, %5\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\ts_mov_b32 m0, %2"
            : "+v"(accp), "+v"(accm), "=&s"(m0_keep)
            : "s"((uint32_t)(HI ? ph >> 32 : ph)), "s"((uint32_t)(HI ? mh >> 32 : mh)), "s"(i));
        ph = (ph << 1) | 1ull;
        mh <<= 1;
        pv = mh | ~(xv | ph);
        mv = ph & xv;
      }
      const bool in = (uint32_t)lane < cnt;
      score += (uint32_t)__popcll(__ballot(in && ((accp >> bit) & 1u)));
      score -= (uint32_t)__popcll(__ballot(in && ((accm >> bit) & 1u)));
    }
  };
  if (m > 32) columns(std::true_type{}); else columns(std::false_type{});
  return score;
}

// ---- lane-per-pair Levenshtein (batches with string ids) ------------------------------
// class of a code point in the script's alphabet (StrFast), 0: not a script character
__device__ __forceinline__ uint32_t cls_of(const StrFast& F, uint32_t cp) {
  uint32_t lo = 0, hi = F.n_cls;                 // clsmap[lo - 1] < cp <= clsmap[hi - 1] or hi = n
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (F.clsmap[mid] < cp) lo = mid + 1; else hi = mid;
  }
  return lo < F.n_cls && F.clsmap[lo] == cp ? lo + 1 : 0u;
}

// Levenshtein.distance(match_str, fan_context) of search.py:189-190 by ONE lane: Myers'
// bit-vector recurrence over the script window's text (the pattern, at most 64 code points,
// as 7 bit planes of character classes: the match mask of a fan character is seven xnor/and
// pairs, no table and no load inside the loop), the fan text '[' + ', '.join(words) + ']'
// streamed class by class from the string records.  Words of more than 15 code points read
// their text; windows of more than 64 code points take the scratch DP of lev_device.
__device__ inline uint32_t lev_lane(const GramIndexDev& g, const CorpusDev& c, const StrFast& F,
                                    uint32_t s, const uint32_t* __restrict__ sid, fs_status* st) {
  const uint4* P4 = reinterpret_cast<const uint4*>(F.pat + 8 * (size_t)s);
  const uint4 t0 = P4[0], t1 = P4[1], t2 = P4[2], t3 = P4[3];
  const uint32_t la = t3.z;
  if (la > 64) return lev_device(g, s, sid, c.chars, c.coff, c.n_str, st);
  const uint32_t plo[7] = {t0.x, t0.z, t1.x, t1.z, t2.x, t2.z, t3.x};
  const uint32_t phi[7] = {t0.y, t0.w, t1.y, t1.w, t2.y, t2.w, t3.y};
  const unsigned long long last = 1ull << (la - 1);
  unsigned long long pv = ~0ull, mv = 0ull;
  uint32_t score = la;
  auto step = [&](uint32_t cl) {
    uint32_t elo = 0xFFFFFFFFu, ehi = 0xFFFFFFFFu;
#pragma unroll
    for (int b = 0; b < 7; ++b) {
      const uint32_t m = 0u - ((cl >> b) & 1u);
      elo &= ~(plo[b] ^ m);
      ehi &= ~(phi[b] ^ m);
    }
    const unsigned long long eq = (unsigned long long)elo | ((unsigned long long)ehi << 32);
    const unsigned long long xv = eq | mv;
    const unsigned long long xh = (((eq & pv) + pv) ^ pv) | eq;
    unsigned long long ph = mv | ~(xh | pv);
    unsigned long long mh = pv & xh;
    score += (ph & last) ? 1u : 0u;
    score -= (mh & last) ? 1u : 0u;
    ph = (ph << 1) | 1ull;
    mh <<= 1;
    pv = mh | ~(xv | ph);
    mv = ph & xv;
  };
  step(F.punct & 0xFFu);                                   // '['
  uint4 nxt = F.strrec[sid[0]];
  for (int k = 0; k < g.n; ++k) {
    uint4 cur = nxt;
    const uint32_t id = sid[k];
    if (k + 1 < g.n) nxt = F.strrec[sid[k + 1]];           // (requested before this word is walked)
    if (k) { step((F.punct >> 8) & 0xFFu); step((F.punct >> 16) & 0xFFu); }   // ', '
    const uint32_t len = cur.x & 0xFFu;
    if (len <= 15) {
      for (uint32_t j = 0; j < len; ++j) {
        cur.x = __builtin_amdgcn_alignbit(cur.y, cur.x, 8);
        cur.y = __builtin_amdgcn_alignbit(cur.z, cur.y, 8);
        cur.z = __builtin_amdgcn_alignbit(cur.w, cur.z, 8);
        cur.w >>= 8;
        step(cur.x & 0xFFu);
      }
    } else {
      for (uint64_t a = c.coff[id]; a < c.coff[id + 1]; ++a) step(cls_of(F, c.chars[a]));
    }
  }
  step(F.punct >> 24);                                     // ']'
  // (the 8-byte wire records hold the distance in ten bits: more is refused like an
  // over-long operand of the other Levenshtein kernels)
  if (score > 1023u) st->lev_overflow = 1;
  return score;
}

// lev_lane with the window's string ids in registers and every word's string record requested
// up front, together with the script window's bit planes: one level of loads where lev_lane
// walks id -> record word by word (k_lsh_lev: a lane per kept match, nothing else to hide a
// chain of 2n loads behind).
__device__ inline uint32_t lev_lane_ids(const GramIndexDev& g, const CorpusDev& c, const StrFast& F,
                                        uint32_t s, const Ids16& sid, fs_status* st) {
  const uint4* P4 = reinterpret_cast<const uint4*>(F.pat + 8 * (size_t)s);
  const uint4 t0 = P4[0], t1 = P4[1], t2 = P4[2], t3 = P4[3];
  uint4 rec[FS_MAX_WINDOW];
#pragma unroll
  for (int k = 0; k < FS_MAX_WINDOW; ++k)
    if (k < g.n) rec[k] = F.strrec[sid.v[k]];
  const uint32_t la = t3.z;
  if (la > 64) return lev_device(g, s, sid.v, c.chars, c.coff, c.n_str, st);
  const uint32_t plo[7] = {t0.x, t0.z, t1.x, t1.z, t2.x, t2.z, t3.x};
  const uint32_t phi[7] = {t0.y, t0.w, t1.y, t1.w, t2.y, t2.w, t3.y};
  const unsigned long long last = 1ull << (la - 1);
  unsigned long long pv = ~0ull, mv = 0ull;
  uint32_t score = la;
  auto step = [&](uint32_t cl) {
    uint32_t elo = 0xFFFFFFFFu, ehi = 0xFFFFFFFFu;
#pragma unroll
    for (int b = 0; b < 7; ++b) {
      const uint32_t m = 0u - ((cl >> b) & 1u);
      elo &= ~(plo[b] ^ m);
      ehi &= ~(phi[b] ^ m);
    }
    const unsigned long long eq = (unsigned long long)elo | ((unsigned long long)ehi << 32);
    const unsigned long long xv = eq | mv;
    const unsigned long long xh = (((eq & pv) + pv) ^ pv) | eq;
    unsigned long long ph = mv | ~(xh | pv);
    unsigned long long mh = pv & xh;
    score += (ph & last) ? 1u : 0u;
    score -= (mh & last) ? 1u : 0u;
    ph = (ph << 1) | 1ull;
    mh <<= 1;
    pv = mh | ~(xv | ph);
    mv = ph & xv;
  };
  step(F.punct & 0xFFu);                                   // '['
  // (one word: `cur` its record; the loop over the words below is written out so that the
  // records stay in registers)
  auto word = [&](uint4 cur, uint32_t id, bool first) {
    if (!first) { step((F.punct >> 8) & 0xFFu); step((F.punct >> 16) & 0xFFu); }   // ', '
    const uint32_t len = cur.x & 0xFFu;
    if (len <= 15) {
      for (uint32_t j = 0; j < len; ++j) {
        cur.x = __builtin_amdgcn_alignbit(cur.y, cur.x, 8);
        cur.y = __builtin_amdgcn_alignbit(cur.z, cur.y, 8);
        cur.z = __builtin_amdgcn_alignbit(cur.w, cur.z, 8);
        cur.w >>= 8;
        step(cur.x & 0xFFu);
      }
    } else {
      for (uint64_t a = c.coff[id]; a < c.coff[id + 1]; ++a) step(cls_of(F, c.chars[a]));
    }
  };
#define FS_LEV_WORD(K) if (K < g.n) word(rec[K], sid.v[K], K == 0)
  FS_LEV_WORD(0); FS_LEV_WORD(1); FS_LEV_WORD(2); FS_LEV_WORD(3); FS_LEV_WORD(4); FS_LEV_WORD(5);
  FS_LEV_WORD(6); FS_LEV_WORD(7); FS_LEV_WORD(8); FS_LEV_WORD(9); FS_LEV_WORD(10); FS_LEV_WORD(11);
  FS_LEV_WORD(12); FS_LEV_WORD(13); FS_LEV_WORD(14); FS_LEV_WORD(15);
#undef FS_LEV_WORD
  static_assert(FS_MAX_WINDOW == 16, "the words are written out");
  step(F.punct >> 24);                                     // ']'
  if (score > 1023u) st->lev_overflow = 1;
  return score;
}

}  // namespace fsdev
