// fs_ranges.h -- "candidates of a wave range -> output records" inside the scan kernel
// (k_scan_rows, fs_scan.hip), and the hand-off that puts the records into place.
//
// Reference semantics (file:line in /root/reference), as in fs_post.hip:
//   search.py:182-184  kept candidates = script windows with the fan window's ids
//   search.py:192-218  n word records per match
//   search.py:224-226  per fan word the FIRST record of minimal dist*lev over the
//                      windows covering it, output ascending by word index
//
// A wave range is the contiguous run of 512-token sub-tiles one wave scans.  The record
// of fan word x depends only on the hits among windows x-n+1 .. x, so a range's records
// follow from its own candidates plus the n-1 windows in front of it (the "halo",
// verified unconditionally): no data crosses ranges.  Per round of at most 64-(n-1)
// candidates (one per lane, position order) the wave
//   1. verifies every candidate: ids -> hash -> displacement seed (a byte in LDS) -> the
//      64-byte entry of the batch table (fs_hash.h), compared id for id, and the work
//      boundary from the token block's entry (two levels of loads, every load of a level
//      requested before any is used); the entry carries the best rank's record of the
//      n-gram for this batch (k_ctab)
//   2. compacts the hits behind the <= n-1 hits carried over from the last round
//   3. emits the records of the words in [E, F): E = words done so far, F = the first
//      position whose hit status is not known yet (the next round's first candidate, or
//      the scan front).  Hit j is the first to cover the words
//      [max(p_j, p_{j-1} + n), p_j + n); one lane per word looks at the <= n-1 later hits
//      that also cover it for the first minimum of the combined distance
//   4. keeps the hits that may still cover words >= F.
// Records go to a staging area of `caprow` records per range (position order inside a
// range = output order) and from there into place: finish_rows below, or k_compact.
// The records never depend on timing: values and order are functions of the token
// positions alone.
#pragma once
#include "fs_device.h"

namespace fsdev {

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr uint32_t FS_NO_LDS = 0xFFFFFFFFu;
constexpr int kHitPad = 72;     // 64 hits + the n-1 <= 7 slots the first-minimum walk may look at

// A hit is one 16-byte LDS record {combined distance of the best rank (two words), window
// position relative to the range | Levenshtein distance << 22, script position of the best
// rank}: what the record loop needs of the <= n hits covering a word comes with one
// ds_read_b128 each, all requested together.  (Position: p - a + 8 < 2^22, the n-1 <= 7 halo
// windows in front of the range included -- checked at launch; Levenshtein distance <= 1023,
// the most lev_lane / lev_device report.)
constexpr uint32_t kHitPosBits = 22;
struct alignas(16) RangeLds {
  uint4 hit[kHitPad];         // hits: {comb lo, comb hi, (p - a + 8) | lev << 22, s}
  uint2 wb[64];               //       {work, first token of that work}
  uint32_t lo[64];            // hit -> (first word it emits in this round) - (its first record)
  uint8_t owner[64 * 8];      // record of the round -> hit
};

struct RangeState {
  uint32_t E;          // words below E are done
  uint32_t hc;         // hits carried over from the last round, at S.*[0 .. hc)
  uint32_t rows_run;   // records of the range so far
  uint32_t hits_run;   // hits inside the range (not the halo)
  uint32_t match_acc;  // per lane: (window, script window) pairs of its hits
  // the range's first 128 records stay in registers (lane L: records L and L + 64, in the
  // staged form) when the launch itself puts the records into place: they go from here to
  // their final place, never through the staging area
  uint4 k0, k1;
  uint32_t pool_base;  // a slice's round (below): its first record in the workgroup's pool
};

// ---- shared rounds ----------------------------------------------------------------
// At the end of its range a wave holds the candidates its last sub-tiles queued; what does
// not fit its next round it may post as *slices* -- runs of queued records that fit one round
// each -- for whichever wave of the workgroup has nothing left to do (its own included).
// A slice is a range boundary like any other: the n-1 windows in front of its first
// candidate are verified again (the hits a serial continuation would have carried), it owns
// the words [P, F) from its first candidate to the next slice's, and its records go to a
// pool of the workgroup (allocated when their number is known), from where the hand-off
// puts them behind the owner's own.  Same records, same order as the serial rounds: a
// record is a function of the hits covering its word.
constexpr uint32_t kCoopPerWave = 6;      // slices a wave may post
constexpr uint32_t kCoopWaves = 16;
constexpr uint32_t kCoopSlots = kCoopPerWave * kCoopWaves;
struct CoopSlice {
  uint32_t r0, r1;     // records [r0, r1) of the owner's queue (at most 64)
  uint32_t P, F;       // the slice's words
  uint32_t a;          // first token of the owner's range (queued positions are relative to it)
  uint32_t base, rows; // out: its records in the pool
  uint32_t hits, pairs;// out: statistics
  uint32_t pad;
};
struct alignas(16) CoopLds {
  CoopSlice slice[kCoopSlots];            // wave w posts into [w * kCoopPerWave, ...)
  uint32_t ring[kCoopSlots + kCoopWaves]; // slot + 1 in posting order (0: not posted yet)
  uint32_t head, tail;                    // tickets taken, slices posted
  uint32_t posted;                        // waves that are through with their range
  uint32_t pool;                          // records allocated in the pool
  uint32_t stat[16];                      // finish_rows of the last workgroup: the statistics its last waves gather
};

// FS_DIAG & 2: time per phase of the rounds, summed per wave range, in ticks of the 100 MHz
// constant clock: {candidates picked, ids + block entry arrived, table line arrived, hits
// stored, records emitted, carried hits moved}
struct RoundClock {
  bool on;
  uint32_t t0, t1, t2, t3, t4, t5;
  uint32_t last;
};
template <int PHASE>
__device__ __forceinline__ void round_lap(RoundClock& k, bool drain) {
  if (!k.on) return;
  if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint32_t now = (uint32_t)__builtin_amdgcn_s_memrealtime();
  uint32_t& t = PHASE == 0 ? k.t0 : PHASE == 1 ? k.t1 : PHASE == 2 ? k.t2 : PHASE == 3 ? k.t3 : PHASE == 4 ? k.t4 : k.t5;
  t += now - k.last;
  k.last = now;
}

struct RangeOut {
  uint8_t* stage;      // caprow records per range
  uint32_t caprow;
  int wire;            // what the caller gets: 0 fs_row, 16 / 8 wire records (staged: 8 for 8, else 16)
  uint8_t* xstage;     // shared rounds: xpool records per workgroup (nullptr: none)
  uint32_t xpool;
};

// One round: lane i < 64 - (N-1) holds candidate window position `p` (ascending over the
// lanes; FS_NONE = none).  F = first position whose hit status is not known after this
// round; a = first token of the range.  A candidate lies inside the token buffer (the
// scan masks windows that run past it; a halo window is in front of the range).
// STR (batches whose fan tokens carry string ids of their own, search.py:151 vs :166): a hit
// whose tokens do not all have string id == vector id cannot take the table's record; its
// lane works out the Levenshtein distance of every kept occurrence (lev_lane, fs_device.h)
// and the first minimum of dist * lev right here.  `sf`: the character classes;
// `give_up`: the host word that sends the search through the chained kernels (set when a
// text is too long for this path: they report it).
template <int N, bool STR = false>
__device__ __forceinline__ void range_round(const CorpusDev& c, const GramIndexDev& g,
                                            uint32_t disp_off, RangeLds& S,
                                            uint32_t p, uint32_t F, uint32_t a,
                                            uint32_t range_id, const RangeOut& out,
                                            RangeState& R, const StrFast* sf,
                                            uint32_t* give_up, bool keep_regs, RoundClock& clk,
                                            uint32_t* pool = nullptr) {
  const int lane = threadIdx.x & 63;
  if (F < R.E) F = R.E;
  round_lap<0>(clk, false);
  // 1. verification, one candidate per lane, two levels of loads: ids + work of the
  // token block, then (seed from LDS) the 64-byte entry of the n-gram, which holds the
  // ids and the batch's best record.  Every load of a level is requested before
  // anything of that level is looked at.
  bool hit = false;
  uint32_t kept = 0, w = 0, wbase = 0, bs = 0, blev = 0;
  double comb = 0.0;
  if (p != FS_NONE) {
    uint32_t f[8];
    uint32_t sid[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint4 bw;
    if constexpr (STR) {
      // (the string ids with the same requests as the vector ids; the buffer is padded)
      const uint4 t0 = *reinterpret_cast<const uint4*>(c.str + p);
      sid[0] = t0.x; sid[1] = t0.y; sid[2] = t0.z; sid[3] = t0.w;
      if constexpr (N > 4) {
        const uint4 t1 = *reinterpret_cast<const uint4*>(c.str + p + 4);
        sid[4] = t1.x; sid[5] = t1.y; sid[6] = t1.z; sid[7] = t1.w;
      }
    }
    {
      const uint4 q0 = *reinterpret_cast<const uint4*>(c.tok + p);   // unaligned 16-byte loads;
      f[0] = q0.x; f[1] = q0.y; f[2] = q0.z; f[3] = q0.w;            // the buffer is padded
      if constexpr (N > 6) {
        const uint4 q1 = *reinterpret_cast<const uint4*>(c.tok + p + 4);
        f[4] = q1.x; f[5] = q1.y; f[6] = q1.z; f[7] = q1.w;
      } else if constexpr (N > 4) {
        const uint2 q1 = *reinterpret_cast<const uint2*>(c.tok + p + 4);
        f[4] = q1.x; f[5] = q1.y; f[6] = 0; f[7] = 0;
      } else {
        f[4] = f[5] = f[6] = f[7] = 0;
      }
      bw = c.blk4[p >> 8];
      asm volatile("" : "+v"(bw.x), "+v"(bw.y), "+v"(bw.z), "+v"(bw.w), "+v"(f[0]));
    }
    round_lap<1>(clk, true);
    uint32_t h = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) h ^= fs_rotl(fs_premix(f[k]), fs_rot_of(N - 1 - k));
    const uint32_t bucket = fs_table_bucket(h, g.log2_buckets);
    // the seed: a byte of the workgroup's dynamic LDS (offset disp_off; an LDS read, not
    // a load through a generic pointer) or, with too many buckets for LDS, of memory
    uint32_t d;
    if (disp_off != FS_NO_LDS) {
      extern __shared__ __attribute__((aligned(16))) uint8_t fs_dyn_lds_bytes[];
      d = fs_dyn_lds_bytes[disp_off + bucket];
    } else {
      d = g.disp8[bucket];
      asm volatile("" : "+v"(d));   // (keeps the two reads apart: merged, they become one
                                    // load through a generic pointer, slower than either)
    }
    if (d == FS_DISP8_WIDE) d = g.disp[bucket];
    // entry: {n-gram + 1, kept, id0, id1 | id2..id5 | id6, id7, s, lev | dist, comb}
    uint4 q0, q1, q2, q3;
    auto differ = [&]() {
      uint32_t diff = (q0.z ^ f[0]) | (N > 1 ? q0.w ^ f[1] : 0u);
      if constexpr (N > 2) diff |= q1.x ^ f[2];
      if constexpr (N > 3) diff |= q1.y ^ f[3];
      if constexpr (N > 4) diff |= q1.z ^ f[4];
      if constexpr (N > 5) diff |= q1.w ^ f[5];
      if constexpr (N > 6) diff |= q2.x ^ f[6];
      if constexpr (N > 7) diff |= q2.y ^ f[7];
      asm volatile("" : "+v"(diff));       // one compare of the OR, not one compare per id
      return diff;
    };
    uint32_t slot = fs_table_slot_d(h, d & FS_DISP_MASK, g.log2_slots);
    bool same = false;
    for (;;) {
      const uint4* e = c.ctab + 4 * (size_t)slot;
      q0 = e[0]; q1 = e[1]; q2 = e[2]; q3 = e[3];
      asm volatile("" : "+v"(q0.x), "+v"(q1.x), "+v"(q2.x), "+v"(q3.x));
      same = (q0.x != 0) & (differ() == 0);
      // one probe decides unless the bucket holds n-grams with colliding 32-bit hashes
      // (flagged: those continue by linear probing)
      if (same | (q0.x == 0) | !(d & FS_DISP_OVERFLOW)) break;
      slot = (slot + 1) & ((1u << g.log2_slots) - 1);
    }
    round_lap<2>(clk, true);
    kept = q0.y; bs = q2.z; blev = q2.w;
    comb = __longlong_as_double((long long)(q3.z | ((uint64_t)q3.w << 32)));
    // the work of p: the block's first work or the one behind it, else walk on
    w = bw.x; wbase = bw.y;
    uint32_t we = bw.z;                               // a batch holds < 2^32 tokens
    if (we <= p) {
      ++w; wbase = we; we = bw.w;
      while (we <= p) { ++w; wbase = we; we = (uint32_t)c.work_off[w + 1]; }
    }
    hit = same & (p + N <= we);
    if constexpr (STR) {
      uint32_t differ_s = 0;
#pragma unroll
      for (int k = 0; k < N; ++k) differ_s |= sid[k] ^ f[k];
      if (hit && (differ_s != 0 || blev == FS_NONE)) {
        const uint32_t gram = q0.x - 1;
        fs_status tmp;
        tmp.bad_string = 0; tmp.lev_overflow = 0;
        // (the distinct occurrences: `kept` counts a window once per table without UniqueFilter)
        const uint32_t ranks = g.gcnt[gram];
        for (uint32_t r = 0; r < ranks; ++r) {
          const uint32_t sw = g.gpos[(size_t)gram * g.nn + r];
          const uint32_t lev = lev_lane(g, c, *sf, sw, c.str + p, &tmp);
          const double cb = __dmul_rn(g.selfdist[sw], (double)lev);
          if (r == 0 || cb < comb) { bs = sw; blev = lev; comb = cb; }
        }
        if (tmp.bad_string | tmp.lev_overflow)
          __hip_atomic_store(give_up, FS_WAIT_GAVE_UP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  // 2. hits behind the carried ones, in position order
  const uint64_t hb = __ballot(hit);
  const uint32_t hidx = R.hc + __builtin_amdgcn_mbcnt_hi((uint32_t)(hb >> 32),
                                   __builtin_amdgcn_mbcnt_lo((uint32_t)hb, 0));
  const uint32_t pbias = a - 8u;                     // position field of a hit: p - pbias
  constexpr uint32_t kPosMask = (1u << kHitPosBits) - 1u;
  if (hit) {
    const uint64_t cbits = (uint64_t)__double_as_longlong(comb);
    S.hit[hidx] = make_uint4((uint32_t)cbits, (uint32_t)(cbits >> 32), (p - pbias) | (blev << kHitPosBits), bs);
    S.wb[hidx] = make_uint2(w, wbase);
    if (p >= a) R.match_acc += kept;                 // a halo hit belongs to the range before
  }
  const uint32_t nh = R.hc + (uint32_t)__popcll(hb);
  R.hits_run += (uint32_t)__popcll(__ballot(hit && p >= a));
  wave_sync();
  round_lap<3>(clk, false);
  // 3. words [E, F): lane j = hit j
  uint32_t cnt = 0, lo = 0, pj = 0;
  if ((uint32_t)lane < nh) {
    pj = (S.hit[lane].z & kPosMask) + pbias;
    uint32_t first = pj;
    if (lane > 0) {
      const uint32_t pv = (S.hit[lane - 1].z & kPosMask) + pbias + N;
      if (pv > first) first = pv;
    }
    lo = first > R.E ? first : R.E;
    uint32_t hi = pj + N;
    if (hi > F) hi = F;
    cnt = hi > lo ? hi - lo : 0;
  }
  const uint32_t inc = wave_incl_scan_dpp(cnt);
  const uint32_t excl = inc - cnt;
  const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
  S.lo[lane] = lo - excl;                              // word of record r of hit j: r + lo[j]
#pragma unroll
  for (uint32_t k = 0; k < (uint32_t)N; ++k)
    if (k < cnt) S.owner[excl + k] = (uint8_t)lane;
  if (pool) {                                          // a slice: its records' place in the pool
    uint32_t b = 0;
    if (lane == 0) b = __hip_atomic_fetch_add(pool, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
    R.pool_base = b;
    R.rows_run = b;
  }
  wave_sync();
  // record r of the range is made by lane r % 64 (so that lane L makes the records L and
  // L + 64 it may keep in registers)
  for (uint32_t rr = ((uint32_t)lane - R.rows_run) & 63u; rr < tot; rr += 64) {
    const uint32_t j = S.owner[rr];
    // one level of LDS reads: the word, the owner's work, the <= n hits that may cover the word
    const uint32_t x = rr + S.lo[j];
    const uint2 wbj = S.wb[j];
    uint4 hs[N];
#pragma unroll
    for (uint32_t t = 0; t < (uint32_t)N; ++t) hs[t] = S.hit[j + t];
    double best = __longlong_as_double((long long)(hs[0].x | ((uint64_t)hs[0].y << 32)));
    uint32_t bz = hs[0].z, bsp = hs[0].w;
#pragma unroll
    for (uint32_t t = 1; t < (uint32_t)N; ++t) {       // the <= n-1 later hits that cover x too
      const double cj = __longlong_as_double((long long)(hs[t].x | ((uint64_t)hs[t].y << 32)));
      const bool ok = (j + t < nh) & ((hs[t].z & kPosMask) + pbias <= x) & (cj < best);
      best = ok ? cj : best;
      bz = ok ? hs[t].z : bz;
      bsp = ok ? hs[t].w : bsp;
    }
    const uint32_t ridx = R.rows_run + rr;
    if (ridx >= out.caprow) continue;
    const uint32_t koff = x - ((bz & kPosMask) + pbias);
    const uint32_t orig = bsp + koff, lev = bz >> kHitPosBits;
    // 16-byte wire record, also when the caller wants fs_row: the distances of a record are
    // functions of the matched script window (its distance to itself, times the Levenshtein
    // distance) and are filled in where the record is put into place; 8-byte form in x, y
    uint4 qv;
    if (out.wire == 8) { qv.x = x; qv.y = orig | (koff << 18) | (lev << 22); qv.z = 0; qv.w = 0; }
    else { qv.x = wbj.x; qv.y = x - wbj.y; qv.z = orig; qv.w = lev | (koff << 16); }
    if (keep_regs && ridx < 128) {
      if (ridx < 64) R.k0 = qv; else R.k1 = qv;
      continue;
    }
    const size_t at = (size_t)range_id * out.caprow + ridx;
    if (out.wire == 8) reinterpret_cast<uint2*>(out.stage)[at] = make_uint2(qv.x, qv.y);
    else if (out.wire == 16 || !pool) reinterpret_cast<uint4*>(out.stage)[at] = qv;
    else {
      // a slice's record for a caller of fs_row: in its final form (the hand-off copies it)
      const double dist = g.selfdist[orig - koff];
      const double cmb = __dmul_rn(dist, (double)lev);
      const uint64_t db = (uint64_t)__double_as_longlong(dist), cb = (uint64_t)__double_as_longlong(cmb);
      uint4* d4 = reinterpret_cast<uint4*>(out.stage) + 2 * at;
      d4[0] = make_uint4(qv.x, qv.y, qv.z, lev);
      d4[1] = make_uint4((uint32_t)db, (uint32_t)(db >> 32), (uint32_t)cb, (uint32_t)(cb >> 32));
    }
  }
  round_lap<4>(clk, false);
  R.rows_run += tot;
  // 4. hits that may cover words >= F move to the front
  const bool keep = (uint32_t)lane < nh && pj + N > F;
  const uint64_t kb = __ballot(keep);
  const uint32_t first_keep = kb ? (uint32_t)(__ffsll((unsigned long long)kb) - 1) : nh;
  if (kb) {                                                        // wave-uniform
    uint4 th = make_uint4(0, 0, 0, 0);
    uint2 tw = make_uint2(0, 0);
    if (keep) { th = S.hit[lane]; tw = S.wb[lane]; }
    wave_sync();
    if (keep) {
      const uint32_t dd = lane - first_keep;
      S.hit[dd] = th; S.wb[dd] = tw;
    }
  }
  R.hc = nh - first_keep;
  R.E = F;
  wave_sync();
  round_lap<5>(clk, false);
}

// ---- records into place ------------------------------------------------------------
// One staged record -> the caller's format.  The two halves are separate so that the
// loads need not wait for the record's final index (the in-launch finish learns it last).
struct StagedRec {
  uint4 q;           // the staged record (8-byte ones in q.x, q.y)
  double dist;
  bool have;
};

__device__ __forceinline__ StagedRec fetch_staged(const uint8_t* __restrict__ stage, int wire,
                                                  const double* __restrict__ selfdist, size_t at, bool have) {
  StagedRec v;
  v.have = have; v.dist = 0.0; v.q = make_uint4(0, 0, 0, 0);
  if (!have) return v;
  if (wire == 8) {
    const uint2 t = reinterpret_cast<const uint2*>(stage)[at];
    v.q.x = t.x; v.q.y = t.y;
  } else {
    v.q = reinterpret_cast<const uint4*>(stage)[at];
    // fs_row: the distance is the matched script window's distance to itself, the
    // combined distance its product with the Levenshtein distance (as fs_rows_unpack)
    if (wire == 0) v.dist = selfdist[v.q.z - (v.q.w >> 16)];
  }
  return v;
}

// the same for a record that stayed in registers (RangeState::k0 / k1)
__device__ __forceinline__ StagedRec kept_staged(const uint4& q, int wire,
                                                 const double* __restrict__ selfdist, bool have) {
  StagedRec v;
  v.have = have; v.dist = 0.0; v.q = q;
  if (have && wire == 0) v.dist = selfdist[q.z - (q.w >> 16)];
  return v;
}

typedef uint32_t fs_v4u __attribute__((ext_vector_type(4)));
typedef uint32_t fs_v2u __attribute__((ext_vector_type(2)));
// (nt: streaming stores -- the records are written once and read by somebody else)
__device__ __forceinline__ void put16(uint8_t* base, size_t idx16, const uint4& q, bool nt) {
  fs_v4u v = {q.x, q.y, q.z, q.w};
  fs_v4u* d = reinterpret_cast<fs_v4u*>(base) + idx16;
  if (nt) __builtin_nontemporal_store(v, d); else *d = v;
}
__device__ __forceinline__ void put8(uint8_t* base, size_t idx8, uint32_t x, uint32_t y, bool nt) {
  fs_v2u v = {x, y};
  fs_v2u* d = reinterpret_cast<fs_v2u*>(base) + idx8;
  if (nt) __builtin_nontemporal_store(v, d); else *d = v;
}

__device__ __forceinline__ void store_staged(uint8_t* __restrict__ rows, int wire, const StagedRec& v, size_t dst,
                                             bool nt = false) {
  if (!v.have) return;
  if (wire == 8) {
    put8(rows, dst, v.q.x, v.q.y, nt);
  } else if (wire == 16) {
    put16(rows, dst, v.q, nt);
  } else {
    const uint32_t lev = v.q.w & 0xFFFFu;
    const double comb = __dmul_rn(v.dist, (double)lev);
    const uint64_t db = (uint64_t)__double_as_longlong(v.dist), cb = (uint64_t)__double_as_longlong(comb);
    put16(rows, 2 * dst, make_uint4(v.q.x, v.q.y, v.q.z, lev), nt);
    put16(rows, 2 * dst + 1, make_uint4((uint32_t)db, (uint32_t)(db >> 32), (uint32_t)cb, (uint32_t)(cb >> 32)), nt);
  }
}

// A slice's records lie in the workgroup's pool in the caller's format (the wave that made
// them has looked the distances up): the hand-off copies bytes.
struct PoolRec { uint4 a, b; };
__device__ __forceinline__ uint32_t pool_rec_bytes(int wire) { return wire == 8 ? 8u : wire == 16 ? 16u : 32u; }
__device__ __forceinline__ PoolRec pool_load(const uint8_t* __restrict__ pool, int wire, size_t idx) {
  PoolRec r;
  r.a = make_uint4(0, 0, 0, 0); r.b = r.a;
  if (wire == 8) { const uint2 t = reinterpret_cast<const uint2*>(pool)[idx]; r.a.x = t.x; r.a.y = t.y; }
  else if (wire == 16) r.a = reinterpret_cast<const uint4*>(pool)[idx];
  else { r.a = reinterpret_cast<const uint4*>(pool)[2 * idx]; r.b = reinterpret_cast<const uint4*>(pool)[2 * idx + 1]; }
  return r;
}
__device__ __forceinline__ void pool_store(uint8_t* __restrict__ rows, int wire, const PoolRec& r, size_t dst, bool nt) {
  if (wire == 8) put8(rows, dst, r.a.x, r.a.y, nt);
  else if (wire == 16) put16(rows, dst, r.a, nt);
  else { put16(rows, 2 * dst, r.a, nt); put16(rows, 2 * dst + 1, r.b, nt); }
}

// ---- records into place inside the launch ------------------------------------------
// Workgroup b of k_scan_rows holds the wave ranges [b * waves, (b + 1) * waves).
// When its waves are done it publishes its record count as one 8-byte granule
// {epoch, count} (the data is the flag; epoch = launch number of the lane, so nothing is
// cleared between launches) and its statistics as four more, sums the granules of the
// workgroups in front of it and copies its staged records to their final place (every
// lane has requested its first two records before it starts to wait).  The last
// workgroup has then seen every other one and publishes totals and status.
//
// No atomic read-modify-write anywhere (one word serves about 88 of them per microsecond:
// a ticket per workgroup or a counter update per wave would cost more than the whole
// hand-off).  A workgroup waits only for workgroups with a smaller index.  The records
// do not depend on timing or placement; that the wait ends relies on workgroups being
// started in index order (every earlier one is then running or done).  The wait is
// bounded: should it ever give up, the search is flagged and the host runs it again
// through the chained kernels of fs_post.hip, which have no in-launch hand-off.
struct RowSync {
  unsigned long long* gran;  // [n_blocks] {epoch << 32 | records of the workgroup}
  unsigned long long* sgran; // [n_blocks][4] {epoch << 32 | hits, pairs, candidates, max records of a range}
  uint32_t epoch;            // >= 1
  uint32_t n_blocks;
  uint32_t spin_limit;       // polls of one batch of granules before the wait gives up (FS_WAIT_SPINS; tests: 0)
  uint32_t wait_ticks;       // ... and the time it may take, in ticks of the 100 MHz constant clock: a few
                             // times what the launch needs for its scan (fs_row_sync), not "about a second"
  // the other finish (searches overlapped on several lanes: a waiting workgroup would hold
  // its CU): per-range and per-workgroup counts go to memory and k_compact, the next
  // kernel, puts the records into place
  uint4* rinfo;              // nullptr: finish inside the launch
  uint4* csum;
  uint32_t* cmax;
};

struct RowFinal {
  uint8_t* rows;           // the caller's buffer
  uint32_t rcap;           // records it holds
  fs_status* st;           // device status block
  fs_status* host_st;      // pinned host copy; the word behind it: set when a wait gave up
  uint64_t* count_out;     // FS_ROWS_HEADER
  bool fresh;              // nothing before this kernel wrote *st
  bool nt;                 // streaming stores for the records
};

// FS_DIAG & 2: stamps of the hand-off (workgroup together, counts known to the wave, to the workgroup)
struct FinStamps {
  bool on;
  unsigned long long t0, t1, t2;
};

// All threads of the workgroup call this once every wave has staged its records.
//   range_id   the wave's range (blockIdx.x * waves + wave)
//   my_rows    (wave-uniform) records of the wave's range; stats of the wave: hits, pairs,
//              candidates
//   s_cnt      LDS, 6 * n_waves + 2 words
__device__ __forceinline__ void finish_rows(const RowSync& sy, const RowFinal& fin,
                                            const RangeOut& out, const double* __restrict__ selfdist,
                                            uint32_t range_id,
                                            uint32_t my_rows, uint32_t hits, uint32_t pairs,
                                            uint32_t cands, uint32_t* s_cnt,
                                            const RangeState* kept,
                                            const CoopLds* coop, uint32_t my_slices,
                                            uint32_t* pool_need, uint32_t* s_stat, FinStamps& stamps) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  const uint32_t L = blockIdx.x;
  const uint32_t staged = my_rows < out.caprow ? my_rows : out.caprow;
  if (sy.rinfo) {                      // k_compact finishes
    if (lane == 0) {
      sy.rinfo[range_id] = make_uint4(my_rows, hits, pairs, cands);
      s_cnt[4 * wave] = my_rows; s_cnt[4 * wave + 1] = hits; s_cnt[4 * wave + 2] = pairs;
      s_cnt[4 * wave + 3] = cands;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint4 t = make_uint4(0, 0, 0, 0);
      uint32_t mx = 0;
      for (uint32_t i = 0; i < n_waves; ++i) {
        t.x += s_cnt[4 * i]; t.y += s_cnt[4 * i + 1]; t.z += s_cnt[4 * i + 2]; t.w += s_cnt[4 * i + 3];
        mx = s_cnt[4 * i] > mx ? s_cnt[4 * i] : mx;
      }
      sy.csum[L] = t;
      sy.cmax[L] = mx;
    }
    return;
  }
  // the staged records of this wave have reached memory before anybody reads them back
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // shared rounds: whoever worked a slice off has added its records to the count of the wave
  // that posted it (s_cnt[owner], LDS atomics; cleared at the start of the launch) and its hits
  // and pairs to its own statistics; this wave adds its own records
  const uint32_t slot0 = wave * kCoopPerWave;
  if (lane == 0) {
    if (coop) __hip_atomic_fetch_add(&s_cnt[wave], staged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else s_cnt[wave] = staged;
    s_cnt[n_waves + 2 + 4 * wave] = hits; s_cnt[n_waves + 3 + 4 * wave] = pairs;
    s_cnt[n_waves + 4 + 4 * wave] = cands; s_cnt[n_waves + 5 + 4 * wave] = my_rows;
  }
  // this lane's first two records: in registers since they were made, or requested from the
  // staging area now; stored once their place is known
  const size_t sbase = (size_t)range_id * out.caprow;
  StagedRec r0, r1;
  if (kept) {
    r0 = kept_staged(kept->k0, out.wire, selfdist, (uint32_t)lane < staged);
    r1 = kept_staged(kept->k1, out.wire, selfdist, (uint32_t)lane + 64 < staged);
  } else {
    r0 = fetch_staged(out.stage, out.wire, selfdist, sbase + lane, (uint32_t)lane < staged);
    r1 = fetch_staged(out.stage, out.wire, selfdist, sbase + lane + 64, (uint32_t)lane + 64 < staged);
  }
  __syncthreads();
  if (stamps.on) stamps.t0 = __builtin_amdgcn_s_memrealtime();      // every wave of the workgroup is here
  // the records of this wave's first slice, from the workgroup's pool: asked for now (two per
  // lane), stored once their place is known
  PoolRec p0, p1;
  bool hp0 = false, hp1 = false;
  const size_t xbase = coop ? (size_t)blockIdx.x * out.xpool : 0;
  if (coop) {
    if (my_slices) {
      const uint32_t rows0 = coop->slice[slot0].rows, base0 = coop->slice[slot0].base;
      hp0 = (uint32_t)lane < rows0 && base0 + lane < out.xpool;
      hp1 = (uint32_t)lane + 64 < rows0 && base0 + lane + 64 < out.xpool;
      if (hp0) p0 = pool_load(out.xstage, out.wire, xbase + base0 + lane);
      if (hp1) p1 = pool_load(out.xstage, out.wire, xbase + base0 + lane + 64);
    }
    if (threadIdx.x == 0 && coop->pool > out.xpool)      // the pool was too small: the host grows it
      __hip_atomic_store(pool_need, coop->pool, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  bool gave_up = false;
  const unsigned long long tag = (unsigned long long)sy.epoch << 32;
  const uint32_t uwave = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
  // records of the workgroup and of the waves in front of this one: lane i looks at wave i
  const uint32_t cw = (uint32_t)lane < n_waves ? s_cnt[lane] : 0u;
  const uint32_t cincl = wave_incl_scan_dpp(cw);
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)cincl, 63);
  const uint32_t in_front = (uint32_t)__builtin_amdgcn_readlane((int)(cincl - cw), (int)uwave);
  if (wave == 0) {
    // five granules, no ordering between them: each carries the epoch.  The count first
    // (the workgroups behind wait for it), the statistics once they are summed.
    if (lane == 0) __hip_atomic_store(sy.gran + L, tag | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t h = 0, pr = 0, cd = 0, mr = 0;
    if ((uint32_t)lane < n_waves) {
      h = s_cnt[n_waves + 2 + 4 * lane]; pr = s_cnt[n_waves + 3 + 4 * lane];
      cd = s_cnt[n_waves + 4 + 4 * lane]; mr = s_cnt[n_waves + 5 + 4 * lane];
    }
    h = wave_sum_lane63(h); pr = wave_sum_lane63(pr); cd = wave_sum_lane63(cd);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      const uint32_t o = (uint32_t)__shfl_xor((int)mr, d);
      mr = o > mr ? o : mr;
    }
    if (lane == 63) {
      __hip_atomic_store(sy.sgran + 4 * (size_t)L + 0, tag | h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sy.sgran + 4 * (size_t)L + 1, tag | pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sy.sgran + 4 * (size_t)L + 2, tag | cd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sy.sgran + 4 * (size_t)L + 3, tag | mr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // records of the workgroups in front: every granule, once it carries this launch's
  // epoch.  Wave w takes the granules [64 w, 64 w + 64), and so on in steps of the
  // workgroup: the waves poll side by side, one round trip when everybody is done.
  uint32_t pre = 0;
  const uint64_t t_wait0 = __builtin_amdgcn_s_memrealtime();
  for (uint32_t i0 = wave * 64; i0 < L; i0 += n_waves * 64) {
    const uint32_t i = i0 + lane;
    unsigned long long v = 0;
    bool ok = i >= L;
    for (uint32_t spins = 0; !__all(ok); ++spins) {
      if (!ok) {
        v = __hip_atomic_load(sy.gran + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = (uint32_t)(v >> 32) == sy.epoch;
      }
      if (spins > 16) __builtin_amdgcn_s_sleep(1);
      if (spins >= sy.spin_limit) { gave_up = true; break; }
      // (the clock every 32nd poll: the workgroups in front are co-resident and publish when
      // their scan is through; a wait of several scans' length means one of them is not running)
      if ((spins & 31) == 31 && __builtin_amdgcn_s_memrealtime() - t_wait0 > sy.wait_ticks) { gave_up = true; break; }
    }
    pre += (i < L && ok) ? (uint32_t)v : 0u;
  }
  pre = wave_sum_lane63(pre);
  if (lane == 63) s_cnt[5 * n_waves + 2 + wave] = pre;
  if (stamps.on) stamps.t1 = __builtin_amdgcn_s_memrealtime();      // the counts in front are known to this wave
  __syncthreads();
  if (stamps.on) stamps.t2 = __builtin_amdgcn_s_memrealtime();      // ... and to every wave of the workgroup
  // this wave's records: from the registers / the staging area to their place
  uint32_t before = 0;                                   // records of the workgroups in front
  for (uint32_t i = 0; i < n_waves; ++i) before += s_cnt[5 * n_waves + 2 + i];
  const uint32_t first = before + in_front;
  const uint32_t lim = first < fin.rcap ? fin.rcap - first : 0u;      // records of this wave the caller's buffer holds
  const uint32_t n = staged < lim ? staged : lim;
  if ((uint32_t)lane < n) store_staged(fin.rows, out.wire, r0, (size_t)first + lane, fin.nt);
  if ((uint32_t)lane + 64 < n) store_staged(fin.rows, out.wire, r1, (size_t)first + lane + 64, fin.nt);
  for (uint32_t i = lane + 128; i < n; i += 64)
    store_staged(fin.rows, out.wire, fetch_staged(out.stage, out.wire, selfdist, sbase + i, true), (size_t)first + i, fin.nt);
  if (coop) {                       // the records of the wave's slices, from the workgroup's pool
    if (hp0 && staged + lane < lim) pool_store(fin.rows, out.wire, p0, (size_t)first + staged + lane, fin.nt);
    if (hp1 && staged + lane + 64 < lim) pool_store(fin.rows, out.wire, p1, (size_t)first + staged + lane + 64, fin.nt);
    uint32_t off = staged;
    for (uint32_t k = 0; k < my_slices; ++k) {
      const uint32_t rows_k = coop->slice[slot0 + k].rows, base_k = coop->slice[slot0 + k].base;
      for (uint32_t i = lane + (k ? 0u : 128u); i < rows_k; i += 64)
        if (off + i < lim && base_k + i < out.xpool)
          pool_store(fin.rows, out.wire, pool_load(out.xstage, out.wire, xbase + base_k + i), (size_t)first + off + i, fin.nt);
      off += rows_k;
    }
  }
  // The search's status: the record count from the last workgroup (it has seen every other
  // one), the statistics from the first -- it waits for nobody's count, so once its records
  // are in place its waves ask for the statistics granules of all workgroups (every one again
  // until it is there) while the others are still busy with their hand-off.
  const bool last = L + 1 == sy.n_blocks;
  if (last && threadIdx.x == 0) {
    const uint32_t n_rows = before + total;
    if (fin.fresh) {
      fin.st->max_recs = 0; fin.st->lev_overflow = 0; fin.st->bad_string = 0; fin.st->lsh_pending = 0;
      fin.host_st->max_recs = 0; fin.host_st->lev_overflow = 0; fin.host_st->bad_string = 0; fin.host_st->lsh_pending = 0;
    } else {
      fin.host_st->max_recs = fin.st->max_recs; fin.host_st->lev_overflow = fin.st->lev_overflow;
      fin.host_st->bad_string = fin.st->bad_string; fin.host_st->lsh_pending = fin.st->lsh_pending;
    }
    fin.st->n_rows = n_rows;
    fin.host_st->n_rows = n_rows;
    if (fin.count_out) *fin.count_out = n_rows;
  }
  if (L == 0) {
    const uint32_t stat_waves = n_waves < 4 ? n_waves : 4u;
    if (wave < stat_waves) {
      uint32_t h = 0, pr = 0, cd = 0, mx = 0;
      for (uint32_t i = wave * 64 + lane; i < sy.n_blocks; i += stat_waves * 64) {
        const unsigned long long* sg = sy.sgran + 4 * (size_t)i;
        unsigned long long a = 0, b = 0, c2 = 0, d2 = 0;
        for (uint32_t spins = 0;; ++spins) {
          // the four granules with two 16-byte loads in flight together (device-coherent: sc1;
          // each granule carries its own epoch, so the pair need not be one access)
          fs_v4u q0, q1;
          asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                       : "=&v"(q0), "=&v"(q1) : "v"(sg) : "memory");
          a = q0.x | ((unsigned long long)q0.y << 32); b = q0.z | ((unsigned long long)q0.w << 32);
          c2 = q1.x | ((unsigned long long)q1.y << 32); d2 = q1.z | ((unsigned long long)q1.w << 32);
          if ((uint32_t)(a >> 32) == sy.epoch && (uint32_t)(b >> 32) == sy.epoch &&
              (uint32_t)(c2 >> 32) == sy.epoch && (uint32_t)(d2 >> 32) == sy.epoch) break;
          if (spins > 16) __builtin_amdgcn_s_sleep(8);
          if (spins >= sy.spin_limit) { gave_up = true; break; }
          if ((spins & 31) == 31 && __builtin_amdgcn_s_memrealtime() - t_wait0 > sy.wait_ticks) { gave_up = true; break; }
        }
        h += (uint32_t)a; pr += (uint32_t)b; cd += (uint32_t)c2;
        mx = (uint32_t)d2 > mx ? (uint32_t)d2 : mx;
      }
      h = wave_sum_lane63(h); pr = wave_sum_lane63(pr); cd = wave_sum_lane63(cd);
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)mx, d);
        mx = o > mx ? o : mx;
      }
      if (lane == 63) { s_stat[4 * wave] = h; s_stat[4 * wave + 1] = pr; s_stat[4 * wave + 2] = cd; s_stat[4 * wave + 3] = mx; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t h = 0, pr = 0, cd = 0, mx = 0;
      for (uint32_t k = 0; k < stat_waves; ++k) {
        h += s_stat[4 * k]; pr += s_stat[4 * k + 1]; cd += s_stat[4 * k + 2];
        mx = s_stat[4 * k + 3] > mx ? s_stat[4 * k + 3] : mx;
      }
      const uint32_t over = mx > out.caprow ? mx : 0;
      fin.st->n_hits = h; fin.st->n_matches = pr; fin.st->n_cands = cd; fin.st->max_rows = over;
      fin.host_st->n_hits = h; fin.host_st->n_matches = pr; fin.host_st->n_cands = cd; fin.host_st->max_rows = over;
    }
  }
  if (__any(gave_up) && lane == 0) {
    // whoever gives up flags the search, in a word of its own that nothing else writes
    __hip_atomic_store(reinterpret_cast<uint32_t*>(fin.host_st + 1), FS_WAIT_GAVE_UP, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace fsdev
