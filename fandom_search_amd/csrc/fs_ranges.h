// fs_ranges.h -- one round of "candidates of a wave range -> output records", shared by
// k_ranges (fs_ranges.hip: candidates from the scan's record lists in global memory) and
// k_scan_rows (fs_scan.hip: the scan kernel itself, candidates from its LDS queue).
// The head of fs_ranges.hip describes the scheme.
#pragma once
#include "fs_device.h"

namespace fsdev {

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct alignas(16) RangeLds {
  double comb[64];          // hits: combined distance of the best rank
  uint32_t cand[64];        // candidates of the round; [RS] = first one of the next round
  uint32_t p[64];           // hits: window position
  uint32_t slot[64];        //       table slot (-> sbest)
  uint32_t w[64];           //       work
  uint32_t wbase[64];       //       first token of that work
  uint32_t lo[64];          //       first word the hit emits in this round
  uint16_t owner[64 * 8];   // word -> hit | k << 8
};

struct RangeState {
  uint32_t E;          // words below E are done
  uint32_t hc;         // hits carried over from the last round, at S.*[0 .. hc)
  uint32_t rows_run;   // records of the range so far
  uint32_t hits_run;   // hits inside the range (not the halo)
  uint32_t match_acc;  // per lane: (window, script window) pairs of its hits
};

struct RangeOut {
  uint8_t* stage;      // caprow records per range
  uint32_t caprow;
  int wire;            // 0: fs_row, 16 / 8: wire records
};

// One round over the candidates S.cand[0 .. m) (window positions in ascending order,
// FS_NONE = no candidate), m <= 64 - (N-1).  F = first position whose hit status is not
// known after this round; a = first token of the range.
template <int N>
__device__ __forceinline__ void range_round(const CorpusDev& c, const GramIndexDev& g,
                                            const fs_best* __restrict__ sbest, RangeLds& S,
                                            uint32_t m, uint32_t F, uint32_t a,
                                            uint32_t range_id, const RangeOut& out,
                                            RangeState& R) {
  constexpr int TS = (2 + N + 3) & ~3;             // words per table entry
  const int lane = threadIdx.x & 63;
  const uint32_t slot_mask = (1u << g.log2_slots) - 1;
  if (F < R.E) F = R.E;
  // 2. verification, one candidate per lane
  const uint32_t p = (uint32_t)lane < m ? S.cand[lane] : FS_NONE;
  bool hit = false;
  uint32_t slot = 0, kept = 0, w = 0, wbase = 0;
  double comb = 0.0;
  if (p != FS_NONE && (uint64_t)p + N <= c.n_tok) {
    uint32_t f[8];
    __builtin_memcpy(f, c.tok + p, 32);            // the buffer is padded
    const uint2 bw = c.blk_work[p >> 8];
    uint32_t h = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) h ^= fs_rotl(fs_premix(f[k]), fs_rot_of(N - 1 - k));
    const uint32_t d = g.disp[fs_table_bucket(h, g.log2_buckets)];
    w = bw.x;
    uint64_t we = bw.y;
    while (we <= p) { ++w; we = c.work_off[w + 1]; }
    const bool inside = (uint64_t)p + N <= we;
    slot = fs_table_slot_d(h, d & FS_DISP_MASK, g.log2_slots);
    for (;;) {
      const uint4* e = reinterpret_cast<const uint4*>(g.table + (size_t)slot * TS);
      uint32_t ew[TS];
#pragma unroll
      for (int qd = 0; qd < TS / 4; ++qd) {
        const uint4 t = e[qd];
        ew[4 * qd] = t.x; ew[4 * qd + 1] = t.y; ew[4 * qd + 2] = t.z; ew[4 * qd + 3] = t.w;
      }
      comb = sbest[slot].comb;                     // beside the entry, not behind it
      wbase = (uint32_t)c.work_off[w];
      if (ew[0] == 0) break;
      bool same = true;
#pragma unroll
      for (int k = 0; k < N; ++k) same = same && (ew[2 + k] == f[k]);
      if (same) { hit = inside; kept = ew[1]; break; }
      if (!(d & FS_DISP_OVERFLOW)) break;
      slot = (slot + 1) & slot_mask;
    }
  }
  // 3. hits behind the carried ones, in position order
  const uint64_t hb = __ballot(hit);
  const uint32_t hidx = R.hc + __builtin_amdgcn_mbcnt_hi((uint32_t)(hb >> 32),
                                 __builtin_amdgcn_mbcnt_lo((uint32_t)hb, 0));
  if (hit) {
    S.p[hidx] = p; S.slot[hidx] = slot; S.w[hidx] = w; S.wbase[hidx] = wbase;
    S.comb[hidx] = comb;
    if (p >= a) R.match_acc += kept;                 // a halo hit belongs to the range before
  }
  const uint32_t nh = R.hc + (uint32_t)__popcll(hb);
  R.hits_run += (uint32_t)__popcll(__ballot(hit && p >= a));
  wave_sync();
  // 4. words [E, F): lane j = hit j
  uint32_t cnt = 0;
  if ((uint32_t)lane < nh) {
    const uint32_t pj = S.p[lane];
    uint32_t first = pj;
    if (lane > 0) {
      const uint32_t pv = S.p[lane - 1] + N;
      if (pv > first) first = pv;
    }
    const uint32_t lo = first > R.E ? first : R.E;
    uint32_t hi = pj + N;
    if (hi > F) hi = F;
    cnt = hi > lo ? hi - lo : 0;
    S.lo[lane] = lo;
  }
  const uint32_t inc = wave_incl_scan_dpp(cnt);
  const uint32_t excl = inc - cnt;
  const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
#pragma unroll
  for (uint32_t k = 0; k < (uint32_t)N; ++k)
    if (k < cnt) S.owner[excl + k] = (uint16_t)((uint32_t)lane | (k << 8));
  wave_sync();
  for (uint32_t rr = lane; rr < tot; rr += 64) {
    const uint32_t o = S.owner[rr];
    const uint32_t j = o & 0xFFu, k = o >> 8;
    const uint32_t x = S.lo[j] + k;
    double best = S.comb[j];
    uint32_t bj = j;
    for (uint32_t jn = j + 1; jn < nh; ++jn) {     // at most n-1 later hits cover x
      if (S.p[jn] > x) break;
      const double cj = S.comb[jn];
      if (cj < best) { best = cj; bj = jn; }
    }
    const uint32_t ridx = R.rows_run + rr;
    if (ridx >= out.caprow) continue;
    const fs_best sb = sbest[S.slot[bj]];
    const uint32_t koff = x - S.p[bj];
    const uint32_t orig = sb.s + koff;
    const size_t at = (size_t)range_id * out.caprow + ridx;
    if (out.wire == 8) {
      reinterpret_cast<uint2*>(out.stage)[at] = make_uint2(x, orig | (koff << 18) | (sb.lev << 22));
    } else if (out.wire == 16) {
      uint4 qv;
      qv.x = S.w[j]; qv.y = x - S.wbase[j]; qv.z = orig; qv.w = sb.lev | (koff << 16);
      reinterpret_cast<uint4*>(out.stage)[at] = qv;
    } else {
      uint4 q0, q1;
      q0.x = S.w[j]; q0.y = x - S.wbase[j]; q0.z = orig; q0.w = sb.lev;
      const uint64_t db = (uint64_t)__double_as_longlong(sb.dist);
      const uint64_t cb = (uint64_t)__double_as_longlong(sb.comb);
      q1.x = (uint32_t)db; q1.y = (uint32_t)(db >> 32); q1.z = (uint32_t)cb; q1.w = (uint32_t)(cb >> 32);
      uint4* dst = reinterpret_cast<uint4*>(out.stage) + 2 * at;
      dst[0] = q0; dst[1] = q1;
    }
  }
  R.rows_run += tot;
  // 5. hits that may cover words >= F move to the front
  const bool keep = (uint32_t)lane < nh && S.p[lane] + N > F;
  const uint64_t kb = __ballot(keep);
  const uint32_t first_keep = kb ? (uint32_t)(__ffsll((unsigned long long)kb) - 1) : nh;
  uint32_t tp = 0, ts = 0, tw = 0, tb = 0;
  double tc = 0.0;
  if (keep) { tp = S.p[lane]; ts = S.slot[lane]; tw = S.w[lane]; tb = S.wbase[lane]; tc = S.comb[lane]; }
  wave_sync();
  if (keep) {
    const uint32_t dd = lane - first_keep;
    S.p[dd] = tp; S.slot[dd] = ts; S.w[dd] = tw; S.wbase[dd] = tb; S.comb[dd] = tc;
  }
  R.hc = nh - first_keep;
  R.E = F;
  wave_sync();
}

}  // namespace fsdev
