// fs_lsh.hip -- the general pipeline: the reference's algorithm as written.
//
// Every fan window is hashed with all H random-binary-projection tables, every
// bucket candidate is scored with the cosine distance, NearestFilter keeps the N
// nearest, the threshold keeps those below `distance_threshold`
// (/root/reference/search.py:112-123, 176-184 and NearPy's Engine).
// Used whenever the exact n-gram proof does not hold: window sizes for which one
// substituted token can stay within the threshold (n = 8, 10 on the synthetic
// table), out-of-vocabulary tokens, vector tables with near-duplicate rows.
//
//   RandomBinaryProjections.hash_vector   window_keys: per-token projection
//        tables A[k][v][c] (k_atab), window projection = sum over k in order,
//        key bit = (p > 0.0); canonical arithmetic of DESIGN.md section 3
//   Engine.store_vector                   fs_lsh_build: script window keys on
//        the device, CSR buckets per table (ascending window index) on the host
//   Engine.neighbours                     lsh_neighbours: bucket entries of
//        table 0, 1, ... in order, UniqueFilter by script window, cosine
//        distance (canonical), stable NearestFilter(N), threshold
//
// The kernels that use them:
//   k_lsh_scan     one block per 256-window sub-tile: keys for all 256 windows
//                  (threads = projection columns, coalesced reads of A rows), then
//                  the (window, bucket candidate) pairs of the sub-tile are dealt out
//                  evenly over the threads and each asks "within the threshold?"; the
//                  per-window answers go out in the scan's bitmap format, so k_expand /
//                  k_rows of fs_post.hip are shared
//   k_lsh_sift     one lane per flagged window, where one slot at most may differ and no
//                  OOV id is involved: wildcard-key Bloom test, the n-gram's record of
//                  this string table (k_lsh_gramtab), the exact one-slot map; what is
//                  left goes onto the pending list
//   k_lsh_verify   one wave per pending window (lsh_window): keys again, the full
//                  neighbours list, Levenshtein per kept match, best rank ->
//                  per-candidate record for k_rows
//   k_lsh_gramtab  lsh_window once per script n-gram and string table
//   k_share_scan   (round 5) in k_lsh_scan's place on tables whose vectors are not unit length:
//                  the windows that can have a script window within the threshold at all, by the
//                  shares of the squared norms their agreeing slots hold ("the share rule" below)
//
// A candidate's exact distance is skipped only when a sound upper bound on its
// cosine is already below 1 - threshold: too few identical slots for the table's
// c_max (integer test), or, slot by slot, the partial canonical sum plus the
// Cauchy-Schwarz bound of the remaining slots (window_distance).  Skipped
// candidates can never be in the output, so the result equals the oracle's, which
// computes every distance.
#include "fs_device.h"

#include <hip/hip_ext.h>

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <type_traits>
#include <vector>

using namespace fsdev;

namespace {

struct LshDev {
  const double* atab;      // [n][V][C]
  const double* nt;        // [n][D][C] normals transposed
  const uint32_t* boff;    // [H][2^B + 1]
  const uint32_t* bids;    // [H][W]
  const double* ss;        // [W] sum of q over script window
  const fs_swin* sw;       // [W] first-slot record per script window
  const double* q;         // [V]
  const float* emb;        // [V][D]
  const uint32_t* stok;    // script vector ids
  const float* atab32;     // [n][V][Cp] float32 copy of atab (rows padded to Cp = 4*ceil(C/4)
                           // floats with zeros, 16-byte aligned), or nullptr
  const float* amax;       // [n][V] >= max_c |atab[k][v][c]|
  const float* nt32;       // [n][D][Cp] float32 copy of nt (rows padded like atab32): the projection row of an
                           // out-of-vocabulary vector is the sum of the rows of its (up to three) hot positions
  const float* ntmax;      // [n][D] >= max_c |nt[k][d][c]|
  float bound_scale;       // n * 2^-22 (times a test factor)
  int m_min;               // fewer id-identical slots than this cannot reach the threshold
  int diag;                // diagnostics: 1 = skip candidate walk, 2 = skip key computation,
                           // 3 = walk the buckets but skip the distances
  int serial_neighbours;   // FS_LSH_SERIAL=1: k_lsh_verify walks the buckets on one lane (cross-check)
  const double* gtab;      // [n_srow][V] g(script row, table row), or nullptr
  const int32_t* sidx;     // [V] row of gtab for a table id, -1 if not a script word
  const uint2* emap;       // one-slot-wildcard keys (over the vector ids, or the component ids: emap_comp) -> distinct
                           // script n-gram: 2^log2_emap buckets of four {key, gram + 1} (k_lsh_batch), or nullptr
  int log2_emap, emap_comp;
  const uint32_t* skeys;   // [W][H] LSH keys of the script windows
  const fs_spos* spos;     // [n_script] {q, pair-table row, id} of every script token (k_spos), or nullptr
  const uint32_t* selflev; // [W] Levenshtein of script window w against the strings of its own ids for
                           // this batch's string table (FS_NONE: compute), or nullptr
  const uint32_t* wild;    // one-slot-wildcard keys of the script windows (fs_hash.h), or nullptr
  int log2_wild;
  const uint32_t* wild_tok;// the ids the keys are made of: nullptr = the vector ids, else the
                           // component ids of the batch's tokens (tables with near-synonyms)
  const uint2* wmap;       // the same keys as an exact map: 2^log2_wmap buckets of four {key, script window + 1}, or nullptr
  int log2_wmap;
  // the share rule (fs_build_share): component ids under the angular relation of the table's vectors and
  // of the script's tokens, the script windows' subset keys, and its constants
  const uint32_t* compa;   // [V], or nullptr: the rule is not in use
  const uint64_t* ssig;    // [W] the script windows' component signatures (share_pair_possible)
  const uint32_t* sharef;  // 2^log2_sharef filter words
  const uint2* smap;       // the same keys as an exact map: 2^log2_smap buckets of four {key, list} (k_share_scan)
  const uint4* slists;     // the lists of the map: a list's length (.x), then its script windows with their signature words
                           // {window, word's low half, high half, 0} (the map names the first of those)
  int log2_smap;
  int log2_sharef, share_flags;
  unsigned long long* share_cnt;   // diagnostics (FS_SHARE_COUNT=1): k_share_scan's counters, or nullptr
  const uint2* oovmap;     // the script's out-of-vocabulary vectors (share_comp): 2^log2_oovmap {key, component}, or nullptr
  int log2_oovmap;
  float share_lim;         // <= 1 - phi: the share of a window's squared norm its near slots must hold
  double share_scale;      // squared norms as integers: floor(q * share_scale) <= 2^20
  double share_phi, share_tau, share_gamma;
  uint32_t V, W;
  int n, H, B, D, C, Cp, nn, unique;
  double thr, cmax;
};

// ---- canonical per-token quantities ------------------------------------------

__device__ __forceinline__ void oov_hot(uint32_t id, int D, uint32_t* a, uint32_t* b, uint32_t* c) {
  const uint32_t code = id & ~FS_OOV_FLAG;
  *c = code % D; *b = (code / D) % D; *a = code / ((uint32_t)D * D);
}

// A[k][id][c]
__device__ __forceinline__ double a_value(const LshDev& L, int k, uint32_t id, int c) {
  if (!(id & FS_OOV_FLAG)) return L.atab[((size_t)k * L.V + id) * L.C + c];
  uint32_t a, b, cc;
  oov_hot(id, L.D, &a, &b, &cc);
  const double* nt = L.nt + (size_t)k * L.D * L.C + c;
  double acc = __dadd_rn(0.0, nt[(size_t)a * L.C]);
  if (b != a) acc = __dadd_rn(acc, nt[(size_t)b * L.C]);
  if (cc != b) acc = __dadd_rn(acc, nt[(size_t)cc * L.C]);
  return acc;
}

// Float32 projection row (four columns from `colc`) of slot k's vector: the table row, or for
// an out-of-vocabulary id the sum of the rows of its distinct hot positions, in a_value's
// order.  *m gets >= max_c |row| added, *terms the number of float32 addends behind the row
// (1, or up to 3): what the decision bound of the float32 keys is made of.
__device__ __forceinline__ float4 row32(const LshDev& L, int k, uint32_t id, int colc) {
  if (!(id & FS_OOV_FLAG)) return *reinterpret_cast<const float4*>(L.atab32 + ((size_t)k * L.V + id) * L.Cp + colc);
  uint32_t a, b, c;
  oov_hot(id, L.D, &a, &b, &c);
  const float* base = L.nt32 + (size_t)k * L.D * L.Cp + colc;
  float4 r = *reinterpret_cast<const float4*>(base + (size_t)a * L.Cp);
  if (b != a) {
    const float4 t = *reinterpret_cast<const float4*>(base + (size_t)b * L.Cp);
    r.x = __fadd_rn(r.x, t.x); r.y = __fadd_rn(r.y, t.y); r.z = __fadd_rn(r.z, t.z); r.w = __fadd_rn(r.w, t.w);
  }
  if (c != b) {
    const float4 t = *reinterpret_cast<const float4*>(base + (size_t)c * L.Cp);
    r.x = __fadd_rn(r.x, t.x); r.y = __fadd_rn(r.y, t.y); r.z = __fadd_rn(r.z, t.z); r.w = __fadd_rn(r.w, t.w);
  }
  return r;
}
__device__ __forceinline__ void row32_bound(const LshDev& L, int k, uint32_t id, float* m, int* terms) {
  if (!(id & FS_OOV_FLAG)) { *m += L.amax[(size_t)k * L.V + id]; *terms += 1; return; }
  uint32_t a, b, c;
  oov_hot(id, L.D, &a, &b, &c);
  const float* mx = L.ntmax + (size_t)k * L.D;
  *m += mx[a]; *terms += 1;
  if (b != a) { *m += mx[b]; *terms += 1; }
  if (c != b) { *m += mx[c]; *terms += 1; }
}

__device__ __forceinline__ double q_of(const LshDev& L, uint32_t id) {
  if (!(id & FS_OOV_FLAG)) return L.q[id];
  uint32_t a, b, c;
  oov_hot(id, L.D, &a, &b, &c);
  return 1.0 + (b != a ? 1.0 : 0.0) + (c != b ? 1.0 : 0.0);
}

// g(u, v) = seqsum_d e_u[d] * e_v[d], u != v; u is a script token.  For table
// rows the sum was computed once per index (k_gtab, same order of operations).
__device__ double g_of(const LshDev& L, uint32_t u, uint32_t v) {
  const bool uo = u & FS_OOV_FLAG, vo = v & FS_OOV_FLAG;
  if (!uo && !vo) {
    if (L.gtab) {
      const int32_t r = L.sidx[u];
      if (r >= 0) return L.gtab[(size_t)r * L.V + v];
    }
    const float* eu = L.emb + (size_t)u * L.D;
    const float* ev = L.emb + (size_t)v * L.D;
    double acc = 0.0;
    for (int d = 0; d < L.D; ++d) acc = __dadd_rn(acc, __dmul_rn((double)eu[d], (double)ev[d]));
    return acc;
  }
  if (uo && vo) {
    uint32_t ua[3], va[3];
    oov_hot(u, L.D, &ua[0], &ua[1], &ua[2]);
    oov_hot(v, L.D, &va[0], &va[1], &va[2]);
    double acc = 0.0;
    for (int i = 0; i < 3; ++i) {
      if (i && ua[i] == ua[i - 1]) continue;
      bool hit = false;
      for (int j = 0; j < 3; ++j) hit = hit || va[j] == ua[i];
      if (hit) acc = __dadd_rn(acc, 1.0);
    }
    return acc;
  }
  const uint32_t row = uo ? v : u, oov = uo ? u : v;
  uint32_t h[3];
  oov_hot(oov, L.D, &h[0], &h[1], &h[2]);
  const float* e = L.emb + (size_t)row * L.D;
  double acc = 0.0;
  for (int i = 0; i < 3; ++i) {
    if (i && h[i] == h[i - 1]) continue;
    acc = __dadd_rn(acc, (double)e[h[i]]);
  }
  return acc;
}

// key h from the per-64-column ballots bal[]: bit j of the key string is column
// h*B + j, first column = most significant bit
__device__ __forceinline__ uint32_t assemble_key(const uint64_t* bal, int h, int B) {
  const int start = h * B, word = start >> 6, off = start & 63;
  uint64_t field = bal[word] >> off;
  if (off + B > 64) field |= bal[word + 1] << (64 - off);
  const uint32_t f = (uint32_t)field & ((1u << B) - 1);
  return __brev(f) >> (32 - B);
}

// CosineDistance of fan window f[] to script window s (canonical), with the
// sound skips described in the file header.  Returns false when skipped or NaN.
// qf: q of the fan window's slots (LDS; lsh_neighbours_wave), or nullptr.  With it, where the
// ids were compared (stage 0), a slot that holds the same id on both sides needs no load at
// all (its q is the fan side's) and only the slots that differ fetch the script side: its id,
// q and the pair-table entry -- at n = 10 two levels of loads for the one slot instead of two
// per slot.
__device__ __forceinline__ bool window_distance_rest(const LshDev& L, uint32_t s, const fs_swin& sw, int same,
                                                     uint32_t diff, const uint32_t* f, const double* qf,
                                                     double ff, double rff, double* out);
__device__ bool window_distance(const LshDev& L, uint32_t s, const uint32_t* f, const double* qf,
                                double ff, double rff, double* out) {
  // stage 0: integer only.  With all table norms in [sqrt(q_min), sqrt(q_max)] and
  // no OOV vector involved, m identical slots bound the cosine by
  // (m q_max + (n-m) c_max q_max) / (n q_min); m_min is the smallest m for which that
  // reaches 1 - threshold (host side, lsh_dev).
  int same = -1;
  uint32_t diff = 0xFFFFFFFFu;                  // bit k: slot k holds different ids (all: not compared)
  // (the window's record requested with its ids: one level for the two)
  fs_swin sw = L.sw[s];
  if (L.m_min > 0) {
    // (the window's ids requested together: stok is padded by a window)
    const uint4* sp = reinterpret_cast<const uint4*>(L.stok + s);
    same = 0;
    diff = 0;
    uint32_t anyoov = 0;
#pragma unroll
    for (int q4 = 0; q4 < FS_MAX_WINDOW / 4; ++q4) {
      if (4 * q4 >= L.n) break;
      const uint4 t = sp[q4];
      const uint32_t u4[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (4 * q4 + k < L.n) {
          const bool eq = u4[k] == f[4 * q4 + k];
          same += eq;
          diff |= eq ? 0u : 1u << (4 * q4 + k);
          anyoov |= u4[k] | f[4 * q4 + k];
        }
    }
    asm volatile("" : "+v"(sw.ss), "+v"(sw.rss), "+v"(sw.qu0), "+v"(sw.u0), "+v"(sw.r0));
    if (anyoov & FS_OOV_FLAG) { same = -1; diff = 0xFFFFFFFFu; }
    else if (L.m_min > 0 && same < L.m_min) return false;
  }
  return window_distance_rest(L, s, sw, same, diff, f, qf, ff, rff, out);
}

// (stage 1 of window_distance, behind the window's record and the comparison of the ids)
__device__ __forceinline__ bool window_distance_rest(const LshDev& L, uint32_t s, const fs_swin& sw, int same,
                                                     uint32_t diff, const uint32_t* f, const double* qf,
                                                     double ff, double rff, double* out) {
  // stage 1: the canonical sum SF slot by slot, leaving as soon as the slots still to
  // come cannot lift it to the threshold.  By Cauchy-Schwarz the remaining slots add
  // at most sqrt(SS_rem * FF_rem) (SS_rem, FF_rem = squared norms of the remaining
  // slots), so  SF_k + sqrt(SS_rem FF_rem) < (1 - thr - 1e-6) sqrt(SS) sqrt(FF)  proves
  // distance > thr + 1e-6, far outside the rounding of the canonical expression.  A
  // bucket collision between unrelated windows leaves after its first slot, which
  // costs one 32-byte record of the script window and one pair-table entry.
  const double norm = __dmul_rn(sw.rss, rff);
  if (same == L.n) {
    // identical ids in every slot: the canonical sum adds q(u_k) in slot order from 0.0,
    // which is how k_ss computed sw.ss -- the same bits, no load
    const double d = __dsub_rn(1.0, __ddiv_rn(sw.ss, norm));
    if (d != d) return false;
    *out = d;
    return true;
  }
  const double need = (1.0 - L.thr - 1e-6) * norm * (1.0 - 1e-9);
  double sf = 0.0, ssr = sw.ss, ffr = ff;
  for (int k = 0; k < L.n; ++k) {
    double g, qu, qv;
    if (qf && !(diff >> k & 1u)) {
      // the same id on both sides: g = q(u) = q(v), the fan side's (the same bits: q is a
      // function of the id)
      qu = qv = g = qf[k];
    } else {
      const uint32_t u = k ? L.stok[s + k] : sw.u0, v = f[k];
      qu = k ? q_of(L, u) : sw.qu0;
      if (u == v) g = qu;
      else if (k == 0 && sw.r0 >= 0 && !(v & FS_OOV_FLAG)) g = L.gtab[(size_t)sw.r0 * L.V + v];
      else g = g_of(L, u, v);
      qv = u == v ? qu : qf ? qf[k] : q_of(L, v);
    }
    sf = __dadd_rn(sf, g);
    if (k + 1 < L.n) {
      ssr -= qu;
      ffr -= qv;
      const double t = need - sf;
      const double rem = fmax(ssr, 0.0) * fmax(ffr, 0.0) * (1.0 + 1e-9);
      if (t > 0.0 && t * t > rem) return false;
    }
  }
  const double d = __dsub_rn(1.0, __ddiv_rn(sf, norm));
  if (d != d) return false;
  *out = d;
  return true;
}

// window_distance for k_lsh_batch, the window size at compile time.  The same first level of
// loads (the window's record and ids); a window that differs from the fan window in three slots
// or fewer -- every real neighbour -- then fetches what the canonical sum needs of those slots
// together: their 16-byte {q, pair-table row, id} records (k_spos) in one level, the pair-table
// entries in the next, where window_distance goes id -> q, id -> row -> entry slot after slot.
// The same arithmetic in the same order; everything else (bucket collisions of unrelated
// windows, which leave after a slot or two; OOV ids) takes window_distance's own loop.
template <int N>
__device__ __forceinline__ bool window_distance_flat(const LshDev& L, uint32_t s, const uint32_t* f,
                                                     const double* qf, double ff, double rff, double* out) {
  const fs_swin sw = L.sw[s];
  const uint4* sp = reinterpret_cast<const uint4*>(L.stok + s);
  uint32_t u[4 * ((N + 3) / 4)];
#pragma unroll
  for (int q4 = 0; q4 < (N + 3) / 4; ++q4) {
    const uint4 t = sp[q4];
    u[4 * q4] = t.x; u[4 * q4 + 1] = t.y; u[4 * q4 + 2] = t.z; u[4 * q4 + 3] = t.w;
  }
  uint32_t diff = 0, anyoov = 0;
  int same = 0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const bool eq = u[k] == f[k];
    same += eq;
    diff |= eq ? 0u : 1u << k;
    anyoov |= u[k] | f[k];
  }
  if (anyoov & FS_OOV_FLAG) return window_distance_rest(L, s, sw, -1, 0xFFFFFFFFu, f, qf, ff, rff, out);
  if (L.m_min > 0 && same < L.m_min) return false;
  if (!L.spos || !L.gtab || N - same > 3)
    return window_distance_rest(L, s, sw, L.m_min > 0 ? same : -1, L.m_min > 0 ? diff : 0xFFFFFFFFu, f, qf, ff, rff, out);
  const double norm = __dmul_rn(sw.rss, rff);
  if (same == N) {
    const double d = __dsub_rn(1.0, __ddiv_rn(sw.ss, norm));
    if (d != d) return false;
    *out = d;
    return true;
  }
  // the (at most three) slots that differ
  int kd[3];
  uint32_t rest = diff;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    kd[i] = rest ? __ffs((int)rest) - 1 : -1;
    rest &= rest - 1;
  }
  uint4 rec[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (kd[i] >= 0) rec[i] = *reinterpret_cast<const uint4*>(L.spos + s + kd[i]);
  double gd[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    gd[i] = 0.0;
    if (kd[i] >= 0) {
      const int32_t row = (int32_t)rec[i].z;
      gd[i] = row >= 0 ? L.gtab[(size_t)row * L.V + f[kd[i]]] : g_of(L, rec[i].w, f[kd[i]]);
    }
  }
  double sf = 0.0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    double g = qf[k];
    g = k == kd[0] ? gd[0] : g; g = k == kd[1] ? gd[1] : g; g = k == kd[2] ? gd[2] : g;
    sf = __dadd_rn(sf, g);
  }
  const double d = __dsub_rn(1.0, __ddiv_rn(sf, norm));
  if (d != d) return false;
  *out = d;
  return true;
}

// Engine.neighbours + threshold.  ANY: stop at the first candidate within the
// threshold (return 1).  Otherwise fill top_s/top_d (capacity nn) with the kept
// matches in NearestFilter order and return their number.
template <bool ANY>
__device__ int lsh_neighbours(const LshDev& L, const uint32_t* keys, const uint32_t* f,
                              uint32_t* top_s, double* top_d) {
  double ff = 0.0;
  for (int k = 0; k < L.n; ++k) ff = __dadd_rn(ff, q_of(L, f[k]));
  const double rff = __dsqrt_rn(ff);
  const uint32_t nb1 = (1u << L.B) + 1;
  int cnt = 0;
  for (int h = 0; h < L.H; ++h) {
    const uint32_t* o = L.boff + (size_t)h * nb1 + keys[h];
    const uint32_t e0 = o[0], e1 = o[1];
    for (uint32_t e = e0; e < e1; ++e) {
      const uint32_t s = L.bids[(size_t)h * L.W + e];
      if (!ANY && L.unique) {
        bool seen = false;
        for (int t = 0; t < cnt; ++t) seen = seen || top_s[t] == s;
        if (seen) continue;
      }
      double d;
      if (L.diag == 3) { cnt += s == 0xFFFFFFFFu; continue; }          // diagnostics: bucket walk only
      if (!window_distance(L, s, f, nullptr, ff, rff, &d)) continue;
      if (!(d < L.thr)) continue;
      if (ANY) return 1;
      // stable insertion: behind every entry with distance <= d
      int pos = cnt;
      while (pos > 0 && d < top_d[pos - 1]) --pos;
      if (pos >= L.nn) continue;
      const int last = cnt < L.nn ? cnt : L.nn - 1;
      for (int m = last; m > pos; --m) { top_d[m] = top_d[m - 1]; top_s[m] = top_s[m - 1]; }
      top_d[pos] = d; top_s[pos] = s;
      if (cnt < L.nn) ++cnt;
    }
  }
  return cnt;
}

// The same neighbour list computed by a whole wave (all 64 lanes call it with the same
// arguments; `keys`, `f`, `top_s`, `top_d`, `s_pre`, `s_e0` in LDS, private to the wave).
// Lane h reads the bucket range of table h; the bucket entries of table 0, 1, ... are
// numbered in order and dealt to the lanes, 64 - nn at a time, behind the <= nn entries
// kept so far: every lane fetches its script window and computes its distance (the
// dependent loads of all candidates in flight together), UniqueFilter = not equal to a
// kept entry or to an earlier lane of the round (an entry dropped earlier has the same
// distance and would be dropped again), NearestFilter = rank in the stable order
// (distance, then arrival) below nn.  Equal to lsh_neighbours<false> entry for entry.
__device__ int lsh_neighbours_wave(const LshDev& L, const uint32_t* keys, const uint32_t* f,
                                   uint32_t* top_s, double* top_d, uint32_t* s_pre,
                                   uint32_t* s_e0, double* qf) {
  const int lane = threadIdx.x & 63;
  if (lane < L.n) qf[lane] = q_of(L, f[lane]);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  double ff = 0.0;
  for (int k = 0; k < L.n; ++k) ff = __dadd_rn(ff, qf[k]);
  const double rff = __dsqrt_rn(ff);
  const uint32_t nb1 = (1u << L.B) + 1;
  uint32_t e0 = 0, cnt_h = 0;
  if (lane < L.H) {
    const uint32_t* o = L.boff + (size_t)lane * nb1 + keys[lane];
    e0 = o[0];
    cnt_h = o[1] - e0;
  }
  const uint32_t incl = wave_incl_scan_dpp(cnt_h);
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  s_pre[lane] = incl - cnt_h;                  // lanes >= H: = total
  s_e0[lane] = e0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const uint32_t nn = (uint32_t)L.nn;
  const uint32_t R = 64 - nn;                  // new candidates per round
  uint32_t kcnt = 0;
  for (uint32_t r0 = 0; r0 < total; r0 += R) {
    const uint32_t m = total - r0 < R ? total - r0 : R;
    uint32_t s = FS_NONE;
    double d = 0.0;
    bool valid = false;
    if ((uint32_t)lane < kcnt) {               // kept so far, in rank order
      s = top_s[lane]; d = top_d[lane]; valid = true;
    } else if ((uint32_t)lane - kcnt < m) {
      const uint32_t j = r0 + ((uint32_t)lane - kcnt);
      uint32_t h = 0;                          // last table with s_pre[h] <= j
#pragma unroll
      for (uint32_t step = 32; step > 0; step >>= 1)
        if (h + step < (uint32_t)L.H && s_pre[h + step] <= j) h += step;
      s = L.bids[(size_t)h * L.W + s_e0[h] + (j - s_pre[h])];
      if (L.diag != 3) valid = window_distance(L, s, f, qf, ff, rff, &d) && d < L.thr;
    }
    if (L.unique) {
      // not the window of a kept entry or of an earlier lane of this round.  Only lanes within
      // the threshold are looked at: a second entry of a window that is not has the same
      // distance and is dropped like the first (two or three lanes instead of every bucket
      // entry: the loop is scalar work, which this kernel is short of)
      uint64_t live = __ballot(valid);
      bool dup = false;
      while (live) {
        const int l = __ffsll((unsigned long long)live) - 1;
        live &= live - 1;
        const uint32_t sl = (uint32_t)__builtin_amdgcn_readlane((int)s, l);
        dup = dup || (l < lane && sl == s);
      }
      if ((uint32_t)lane >= kcnt) valid = valid && !dup;
    }
    // rank in the stable order: smaller distance first, earlier arrival first
    uint64_t vm = __ballot(valid);
    const uint32_t nvalid = (uint32_t)__popcll(vm);
    uint32_t rank = 0;
    const uint32_t dlo = (uint32_t)__double_as_longlong(d), dhi = (uint32_t)(__double_as_longlong(d) >> 32);
    while (vm) {
      const int l = __ffsll((unsigned long long)vm) - 1;
      vm &= vm - 1;
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)dlo, l);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)dhi, l);
      const double dl = __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
      rank += (dl < d || (dl == d && l < lane)) ? 1u : 0u;
    }
    __builtin_amdgcn_wave_barrier();           // every lane has read its kept entry
    if (valid && rank < nn) { top_s[rank] = s; top_d[rank] = d; }
    kcnt = nvalid < nn ? nvalid : nn;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  return (int)kcnt;
}

// ---- build kernels -----------------------------------------------------------

__global__ void k_nt(const double* __restrict__ normals, int n, int D, int C,
                     double* __restrict__ nt) {
  // nt[k][d][c] = normals[c][k*D + d]
  const size_t total = (size_t)n * D * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const size_t kd = i / C;
    nt[i] = normals[(size_t)c * n * D + kd];
  }
}

// float32 copy of nt, rows padded to Cp, and the rows' largest magnitudes (rounded up)
__global__ __launch_bounds__(256) void k_nt32(const double* __restrict__ nt, int rows, int C, int Cp,
                                              float* __restrict__ nt32, float* __restrict__ ntmax) {
  __shared__ float s_m[4];
  const int r = blockIdx.x;
  if (r >= rows) return;
  float mx = 0.0f;
  for (int c = threadIdx.x; c < Cp; c += blockDim.x) {
    const double v = c < C ? nt[(size_t)r * C + c] : 0.0;
    nt32[(size_t)r * Cp + c] = (float)v;
    mx = fmaxf(mx, __double2float_ru(fabs(v)));
  }
  for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) ntmax[r] = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
}

// A[k][v][c] = seqsum_d nt[k][d][c] * (double)E[v][d]; block = one (k, v)
__global__ __launch_bounds__(256) void k_atab(const double* __restrict__ nt,
                                              const float* __restrict__ emb, uint32_t V, int D,
                                              int C, int Cp, double* __restrict__ atab,
                                              float* __restrict__ atab32,
                                              float* __restrict__ amax) {
  __shared__ float s_m[4];
  const uint32_t v = blockIdx.x;
  const int k = blockIdx.y;
  const float* e = emb + (size_t)v * D;
  const double* ntk = nt + (size_t)k * D * C;
  float mx = 0.0f;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    double acc = 0.0;
    for (int d = 0; d < D; ++d)
      acc = __dadd_rn(acc, __dmul_rn(ntk[(size_t)d * C + c], (double)e[d]));
    const size_t r = (size_t)k * V + v;
    atab[r * C + c] = acc;
    atab32[r * Cp + c] = (float)acc;
    mx = fmaxf(mx, __double2float_ru(fabs(acc)));          // rounded up
  }
  for (int c = C + threadIdx.x; c < Cp; c += blockDim.x) atab32[((size_t)k * V + v) * Cp + c] = 0.0f;
  for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0)
    amax[(size_t)k * V + v] = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
}

// embT[d][v] = (double) E[v][d]: coalesced reads for k_gtab
__global__ void k_embT(const float* __restrict__ emb, uint32_t V, int D, float* __restrict__ embT) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  for (int d = 0; d < D; ++d) embT[(size_t)d * V + v] = emb[(size_t)v * D + d];
}

// gtab[r][v] = seqsum_d E[srow[r]][d] * E[v][d]  (canonical: mul then add, d ascending)
__global__ __launch_bounds__(256) void k_gtab(const float* __restrict__ emb,
                                              const float* __restrict__ embT, uint32_t V, int D,
                                              const uint32_t* __restrict__ srow,
                                              double* __restrict__ gtab) {
  extern __shared__ float s_u[];     // the script row, D floats
  const uint32_t r = blockIdx.y;
  const float* eu = emb + (size_t)srow[r] * D;
  for (int d = threadIdx.x; d < D; d += blockDim.x) s_u[d] = eu[d];
  __syncthreads();
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  double acc = 0.0;
  for (int d = 0; d < D; ++d)
    acc = __dadd_rn(acc, __dmul_rn((double)s_u[d], (double)embT[(size_t)d * V + v]));
  gtab[(size_t)r * V + v] = acc;
}

__global__ void k_ss(const uint32_t* __restrict__ stok, uint32_t W, LshDev L,
                     double* __restrict__ ss, fs_swin* __restrict__ sw) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= W) return;
  double acc = 0.0;
  for (int k = 0; k < L.n; ++k) acc = __dadd_rn(acc, q_of(L, stok[w + k]));
  ss[w] = acc;
  fs_swin r;
  r.ss = acc;
  r.rss = __dsqrt_rn(acc);
  r.u0 = stok[w];
  r.qu0 = q_of(L, r.u0);
  r.r0 = (L.gtab && !(r.u0 & FS_OOV_FLAG)) ? L.sidx[r.u0] : -1;
  sw[w] = r;
}

// {q, pair-table row, id} of every script token: what window_distance_flat reads per slot
__global__ void k_spos(const uint32_t* __restrict__ stok, uint32_t n_script, LshDev L, fs_spos* __restrict__ spos) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_script) return;
  fs_spos r;
  r.id = stok[i];
  r.q = q_of(L, r.id);
  r.row = (L.gtab && !(r.id & FS_OOV_FLAG)) ? L.sidx[r.id] : -1;
  spos[i] = r;
}

// keys of the windows of a token stream; one wave per window
__global__ __launch_bounds__(256) void k_keys(LshDev L, const uint32_t* __restrict__ tok,
                                              uint32_t n_windows, uint32_t* __restrict__ keys) {
  __shared__ uint64_t s_bal[4][32];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int NW = (L.C + 63) >> 6;
  for (uint32_t w = blockIdx.x * 4 + wave; w < n_windows; w += gridDim.x * 4) {
    for (int ch = 0; ch < NW; ++ch) {
      const int c = ch * 64 + lane;
      bool bit = false;
      if (c < L.C) {
        double acc = a_value(L, 0, tok[w], c);
        for (int k = 1; k < L.n; ++k) acc = __dadd_rn(acc, a_value(L, k, tok[w + k], c));
        bit = acc > 0.0;
      }
      const uint64_t b = __ballot(bit);
      if (lane == 0) s_bal[wave][ch] = b;
    }
    if (lane == 0) s_bal[wave][NW] = 0;
    __builtin_amdgcn_wave_barrier();
    if (lane < L.H) keys[(size_t)w * L.H + lane] = assemble_key(s_bal[wave], lane, L.B);
    __builtin_amdgcn_wave_barrier();
  }
}

// string id == vector id: Levenshtein of every script window against the strings of its
// own ids, one wave per window (FS_NONE where lev_wave reports a bad string or an
// overflow: the search then computes that match itself and reports the same)
__global__ __launch_bounds__(256) void k_selflev(GramIndexDev g, CorpusDev c, uint32_t W,
                                                 uint32_t* __restrict__ selflev) {
  __shared__ uint32_t s_la[4][FS_LEV_MAX + 2], s_lb[4][FS_LEV_MAX + 2];
  __shared__ uint32_t s_ids[4][FS_MAX_WINDOW];
  __shared__ fs_status s_st[4];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  for (uint32_t w = blockIdx.x * 4 + wave; w < W; w += gridDim.x * 4) {
    if (lane < g.n) s_ids[wave][lane] = g.stok[w + lane];
    if (lane == 0) { s_st[wave].bad_string = 0; s_st[wave].lev_overflow = 0; }
    __builtin_amdgcn_wave_barrier();
    const bool oov = lane < g.n && (s_ids[wave][lane] & FS_OOV_FLAG);
    uint32_t v = FS_NONE;
    if (!__any(oov)) {
      v = lev_wave(g, w, s_ids[wave], c.chars, c.coff, c.n_str, &s_st[wave], s_la[wave], s_lb[wave]);
      __builtin_amdgcn_wave_barrier();
      if (s_st[wave].bad_string | s_st[wave].lev_overflow) v = FS_NONE;
    }
    if (lane == 0) selflev[w] = v;
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- CSR buckets on the device (Engine.store_vector for every script window) ----------
// boff[h][k+1] counts the windows with key k in table h, a scan turns the counts into
// offsets, a scatter fills bids in arrival order, and every bucket is then sorted by window
// index: the reference's buckets list their windows in insertion (= ascending) order, and
// the order decides UniqueFilter's and NearestFilter's ties.
__global__ void k_bucket_count(const uint32_t* __restrict__ keys, uint32_t W, int H, uint32_t nb,
                               uint32_t* __restrict__ boff) {
  const uint64_t total = (uint64_t)W * H;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t h = (uint32_t)(i % H);
    atomicAdd(&boff[(size_t)h * (nb + 1) + keys[i] + 1], 1u);
  }
}

// one workgroup per table: counts -> offsets in place (boff[h][0] = 0), and a copy of the
// bucket starts as the scatter's cursors
__global__ __launch_bounds__(256) void k_bucket_offsets(uint32_t nb, uint32_t* __restrict__ boff,
                                                        uint32_t* __restrict__ cursor) {
  __shared__ uint32_t s_w32[4];
  uint32_t* off = boff + (size_t)blockIdx.x * (nb + 1);
  uint32_t* cur = cursor + (size_t)blockIdx.x * nb;
  uint32_t carry = 0;
  for (uint32_t b0 = 0; b0 < nb; b0 += 256) {
    const uint32_t b = b0 + threadIdx.x;
    const uint32_t v = b < nb ? off[b + 1] : 0u;
    uint32_t tot;
    const uint32_t excl = block_excl_scan(v, s_w32, &tot);
    if (b < nb) {
      off[b + 1] = carry + excl + v;
      cur[b] = carry + excl;
    }
    carry += tot;
    __syncthreads();
  }
}

__global__ void k_bucket_fill(const uint32_t* __restrict__ keys, uint32_t W, int H, uint32_t nb,
                              uint32_t* __restrict__ cursor, uint32_t* __restrict__ bids) {
  const uint64_t total = (uint64_t)W * H;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t h = (uint32_t)(i % H), w = (uint32_t)(i / H);
    const uint32_t at = atomicAdd(&cursor[(size_t)h * nb + keys[i]], 1u);
    bids[(size_t)h * W + at] = w;
  }
}

// ascending window index inside every bucket: a thread sorts a bucket of up to kSmallBucket
// entries by insertion; larger ones are listed for k_bucket_sort_big
constexpr uint32_t kSmallBucket = 48;
__global__ void k_bucket_sort(uint32_t W, int H, uint32_t nb, const uint32_t* __restrict__ boff,
                              uint32_t* __restrict__ bids, uint32_t* __restrict__ big,
                              uint32_t* __restrict__ n_big) {
  const uint64_t total = (uint64_t)nb * H;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t h = (uint32_t)(i / nb), b = (uint32_t)(i % nb);
    const uint32_t* off = boff + (size_t)h * (nb + 1) + b;
    const uint32_t e0 = off[0], m = off[1] - e0;
    if (m < 2) continue;
    if (m > kSmallBucket) { big[atomicAdd(n_big, 1u)] = (uint32_t)i; continue; }
    uint32_t* v = bids + (size_t)h * W + e0;
    for (uint32_t a = 1; a < m; ++a) {
      const uint32_t x = v[a];
      uint32_t c = a;
      while (c > 0 && v[c - 1] > x) { v[c] = v[c - 1]; --c; }
      v[c] = x;
    }
  }
}

// a large bucket (many script windows with one key: a repeated passage): one workgroup, every
// entry's place is the number of smaller entries (window indices are distinct)
__global__ __launch_bounds__(256) void k_bucket_sort_big(uint32_t W, uint32_t nb,
                                                         const uint32_t* __restrict__ boff,
                                                         uint32_t* __restrict__ bids,
                                                         const uint32_t* __restrict__ big,
                                                         const uint32_t* __restrict__ n_big,
                                                         uint32_t* __restrict__ tmp) {
  for (uint32_t j = blockIdx.x; j < *n_big; j += gridDim.x) {
    const uint32_t i = big[j], h = i / nb, b = i % nb;
    const uint32_t* off = boff + (size_t)h * (nb + 1) + b;
    const uint32_t e0 = off[0], m = off[1] - e0;
    uint32_t* v = bids + (size_t)h * W + e0;
    uint32_t* t = tmp + (size_t)h * W + e0;
    for (uint32_t a = threadIdx.x; a < m; a += blockDim.x) {
      const uint32_t x = v[a];
      uint32_t r = 0;
      for (uint32_t c = 0; c < m; ++c) r += v[c] < x;
      t[r] = x;
    }
    __syncthreads();
    for (uint32_t a = threadIdx.x; a < m; a += blockDim.x) v[a] = t[a];
    __syncthreads();
  }
}

// ---- the share rule ----------------------------------------------------------------
//
// For tables whose vectors are not unit length none of the integer prefilters applies ("at most
// one slot may differ" is false there: a window's squared norm may sit in a few slots, and the
// others may then hold anything).  What holds for any norms: call two vectors *near* when their
// cosine exceeds gamma (components of that relation over the table: compa; a vector of norm 0 is
// near nothing), let D be the slots of a window pair (F, S) whose vectors lie in different
// components, and A, B the shares of |F|^2 and |S|^2 those slots hold.  Then, slot by slot
// dot(f_k, s_k) <= |f_k||s_k| and <= gamma |f_k||s_k| on D, and by Cauchy-Schwarz on either group
//   cos(F, S) <= sqrt((1 - A)(1 - B)) + gamma sqrt(A B)  <=  sqrt(1 - A (1 - gamma^2)),
// so a pair within the threshold (cos > tau = 1 - thr - 1e-6) has A < phi and B < phi,
// phi = (1 - tau^2) / (1 - gamma^2): the slots that agree in their components hold more than
// 1 - phi of either window's squared norm.  Two sound skips come of it, both in k_lsh_scan:
//   * the gate, per fan window: the subsets M of slots that are *heavy* (hold that share of the
//     fan window) and minimal (no slot can go) are asked for in a filter that holds, for every
//     script window, the key (slots, component ids there) of every subset of its slots.  The set
//     of agreeing slots of a pair within the threshold is heavy, so it contains a minimal heavy
//     subset, and that one's key is in the filter: a window none of whose keys is there has no
//     script window within the threshold, needs no LSH keys and walks no bucket.  (FS_LSH_SHARE
//     bit 2: the filter holds the script windows' own heavy subsets only and every heavy subset
//     of the fan window is asked for -- the agreeing set is heavy on both sides.)  Squared norms
//     are integers here (floor(q * share_scale)), so "heavy" is one exact comparison however the
//     subset is summed, with the slack of the rounding on the permissive side.
//   * the test, per (fan window, bucket member): A from the fan side alone (LDS), then B and the
//     two-sided bound, in front of window_distance and its pair-table entries.
// An out-of-vocabulary fan token (at most three coordinates, all 1) is far from every script
// vector when sqrt(3) max_d |u_d| / |u| <= gamma for all of them (checked at index build);
// otherwise (share_flags bit 3) its slot counts as agreeing with anything.  Scripts with
// out-of-vocabulary tokens do not use the rule.
// Component id of a fan token under the share rule.  A table row: compa.  An out-of-vocabulary
// token is a vector of at most three ones: far from every table row of the script (checked at
// index build), and against the script's own out-of-vocabulary vectors, by the sets of hot
// positions -- equal sets are the same vector (cosine 1: that script vector's component), three
// distinct positions against another three share at most two (2/3: far, gamma >= 0.668 is
// required of such an index), and every other case involves a set of fewer than three (a hash
// that met itself): cosines 0.71 and 0.82 occur there, so a fan token with fewer than three
// distinct positions, or one that contains a two-position vector of the script, is FS_WILD: it
// counts as agreeing with anything.  (A script token with fewer than three positions has a
// component of its own that no fan token carries: fan tokens near it are all FS_WILD.)
#define FS_WILD 0xFFFFFFFEu
__device__ __forceinline__ uint32_t share_oov_lookup(const LshDev& L, uint32_t key) {
  const uint32_t mask = (1u << L.log2_oovmap) - 1u;
  for (uint32_t at = fs_mix24(key) & mask;; at = (at + 1) & mask) {
    const uint2 e = L.oovmap[at];
    if (e.y == 0u) return 0u;                      // (values are stored + 1)
    if (e.x == key) return e.y;
  }
}
__device__ __forceinline__ uint32_t share_comp(const LshDev& L, uint32_t id) {
  if (!(id & FS_OOV_FLAG)) return L.compa[id];
  if (L.share_flags & 8) return FS_WILD;
  if (!L.oovmap || L.diag == 0x1000000) return FS_NONE;   // (diagnostics 0x1000000, a wrong rule on purpose: what tools/stress_share.py must catch)
  uint32_t x, y, z;
  oov_hot(id, L.D, &x, &y, &z);
  uint32_t t;
  if (x > y) { t = x; x = y; y = t; }
  if (y > z) { t = y; y = z; z = t; }
  if (x > y) { t = x; x = y; y = t; }
  if (x == y || y == z) return FS_WILD;
  const uint32_t D = (uint32_t)L.D;
  if (share_oov_lookup(L, 0x80000000u | (x * D + y)) || share_oov_lookup(L, 0x80000000u | (x * D + z)) ||
      share_oov_lookup(L, 0x80000000u | (y * D + z)))
    return FS_WILD;
  const uint32_t c = share_oov_lookup(L, (x * D + y) * D + z);
  return c ? c - 1u : FS_NONE;
}

// The keys a fan window asks for (its minimal heavy subsets; all heavy ones under share_flags bit 2),
// into list[j * 256]: their number, or -1 when the window is not constrained (the rule says nothing
// about it, or the list is too short for its keys).
// (the subsets of the slots K .. KEND - 1 depth first: a subset's sum and minimum are its parent's and
// one operation each; what leaves is the subset's mask, 16 bits -- its key is made by whoever reads
// the list, once per subset asked for instead of once per subset)
// (a subset of a window's slots: a byte for windows of up to eight slots)
template <int N> using share_mask_t = std::conditional_t<(N <= 8), uint8_t, uint16_t>;

template <int N, int K, int KEND, uint32_t M>
struct ShareSubsets {
  static __device__ __forceinline__ void go(const uint32_t (&qi)[N], uint32_t usable, int thr, bool every,
                                            uint32_t sum, uint32_t mn, share_mask_t<N>* list, int cap, int& cnt) {
    if constexpr (K == KEND) {
      if constexpr (M != 0u) {
        const bool ask = (M & ~usable) == 0u && (int)sum >= thr && (every || (int)(sum - mn) < thr);
        if (ask) {
          if (cnt < cap) list[cnt * 256] = (share_mask_t<N>)M;
          ++cnt;
        }
      }
    } else {
      ShareSubsets<N, K + 1, KEND, M>::go(qi, usable, thr, every, sum, mn, list, cap, cnt);
      ShareSubsets<N, K + 1, KEND, (M | (1u << K))>::go(qi, usable, thr, every, sum + qi[K], qi[K] < mn ? qi[K] : mn,
                                                        list, cap, cnt);
    }
  }
};

// Windows of more than six slots, run by run (fs_share_blocks): the agreeing slots of a pair within
// the threshold hold more than `lim` of the fan window's squared norm, so in at least one run they
// hold more than `lim` of *that run's* -- the run's minimal subsets that do are asked for, with
// the slots' own numbers in the key.  (A run the rule says nothing about -- all of it slots that
// agree with anything -- leaves the window unconstrained.)
template <int N, int R>
__device__ __forceinline__ bool share_asks_run(const LshDev& L, const uint32_t (&qi)[N], const uint32_t (&wild)[N],
                                               uint32_t usable, bool every, share_mask_t<N>* list, int cap, int& cnt) {
  constexpr int K0 = fs_share_block_start(N, R), K1 = fs_share_block_start(N, R + 1);
  uint32_t all = 0, base = 0;
#pragma unroll
  for (int k = K0; k < K1; ++k) { all += qi[k]; base += wild[k]; }
  // heavy(M): sum_M qi >= thr.  (With x = q * scale real and qi = floor(x): a truly heavy M has
  // sum_M x >= lim sum x - sum_O x, so sum_M qi > lim * all - base - (K1 - K0).)
  const int thr = (int)floorf(L.share_lim * (float)all) - (int)base - (K1 - K0) - 2;
  if (thr <= 0) return false;
  ShareSubsets<N, K0, K1, 0u>::go(qi, usable, thr, every, 0u, 0xFFFFFFFFu, list, cap, cnt);
  return true;
}

// The subsets a fan window asks for -- their slots as bit masks, into list[j * 256] --: their number,
// or -1 when the window is not constrained (the rule says nothing about it, or the list is too
// short).  The key of a subset is made where it is needed (share_key_of: from the slots' terms).
template <int N>
__device__ __forceinline__ int share_asks(const LshDev& L, const uint32_t* cmp, const double* qd,
                                          share_mask_t<N>* list, int cap) {
  uint32_t qi[N], wild[N];
  uint32_t usable = 0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    qi[k] = (uint32_t)(qd[k] * L.share_scale);
    const uint32_t c = cmp[k];
    wild[k] = c == FS_WILD ? qi[k] + 1 : 0u;
    usable |= c < FS_WILD ? 1u << k : 0u;
  }
  const bool every = (L.share_flags & 4) != 0;
  int cnt = 0;
  bool ok = share_asks_run<N, 0>(L, qi, wild, usable, every, list, cap, cnt);
  if constexpr (fs_share_blocks(N) > 1) ok = ok && share_asks_run<N, 1>(L, qi, wild, usable, every, list, cap, cnt);
  if constexpr (fs_share_blocks(N) > 2) ok = ok && share_asks_run<N, 2>(L, qi, wild, usable, every, list, cap, cnt);
  return !ok || cnt > cap ? -1 : cnt;
}

template <int N>
__device__ __forceinline__ void share_terms(const uint32_t* cmp, uint32_t (&t)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) t[k] = fs_share_term(cmp[k], k);   // (a slot without a component is in no subset)
}
template <int N>
__device__ __forceinline__ uint32_t share_key_of(const uint32_t (&t)[N], uint32_t m) {
  uint32_t fold = 0;
#pragma unroll
  for (int k = 0; k < N; ++k) fold ^= ((m >> k) & 1u) ? t[k] : 0u;
  return fs_share_key(fold, m);
}

template <int N>
__device__ __forceinline__ bool share_gate(const LshDev& L, const uint32_t* cmp, const double* qd,
                                           share_mask_t<N>* list, int cap) {
  const int cnt = share_asks<N>(L, cmp, qd, list, cap);
  if (cnt < 0) return true;
  uint32_t t[N];
  share_terms<N>(cmp, t);
  bool hit = false;
  for (int j = 0; j < cnt && !hit; j += 4) {
    uint32_t h[4], wd[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) h[u] = j + u < cnt ? share_key_of<N>(t, list[(j + u) * 256]) : 0u;
#pragma unroll
    for (int u = 0; u < 4; ++u) wd[u] = j + u < cnt ? L.sharef[fs_bloom_word(h[u], L.log2_sharef)] : 0u;
#pragma unroll
    for (int u = 0; u < 4; ++u) hit = hit || (j + u < cnt && fs_bloom_test(wd[u], h[u]));
  }
  return hit;
}

// The pairs' test: false when script window s cannot be within the threshold of the fan window whose
// slots' component signatures and squared norms are sg[] / qd[] (sum ff).  A signature is a few bits
// of a hash of the component id (fs_share_sig; FS_NONE: an out-of-vocabulary token), the script
// window's n of them are one 64-bit word: slots whose signatures differ lie in different components,
// slots whose signatures agree count as agreeing.
template <int N>                  // (N = 0: the window size at run time)
__device__ __forceinline__ bool share_pair_possible(const LshDev& L, uint32_t s, uint64_t ssig, const uint32_t* sg,
                                                    const double* qd, double ff) {
  const int n = N ? N : L.n;
  const int b = fs_share_sig_bits(n);
  double af = 0.0;
  uint32_t dm = 0;
#pragma unroll
  for (int k = 0; k < n; ++k) {
    const uint32_t c = sg[k];
    const bool far = c == FS_WILD ? false : c == FS_NONE ? true : c != (uint32_t)((ssig >> (k * b)) & ((1u << b) - 1u));
    af = far ? af + qd[k] : af;
    dm |= far ? 1u << k : 0u;
  }
  if (!(ff > 0.0) || dm == 0u) return true;
  // (A >= phi: the bound is at most tau whatever B is)
  if (af >= L.share_phi * (1.0 + 1e-9) * ff) return false;
  const double A = fmin(af / ff, 1.0);
  double bs = 0.0;
#pragma unroll
  for (int k = 0; k < n; ++k)
    if ((dm >> k) & 1u) bs = bs + L.spos[s + k].q;
  const double ss = L.ss[s];
  if (!(ss > 0.0)) return true;
  const double B = fmin(bs / ss, 1.0);
  return sqrt((1.0 - A) * (1.0 - B)) + L.share_gamma * sqrt(A * B) > L.share_tau;
}

// ---- search kernels ------------------------------------------------------------

// The share rule's gate for every window of a token stream: bit w of gbm = window w may have a
// script window within the threshold.  A thread per window, 256 to a workgroup as k_lsh_scan's
// sub-tiles (which read the bits).  A kernel of its own: inside k_lsh_scan its 63 subsets cost a
// wave slot per SIMD (154 registers against 103).
constexpr int kGateCap = 32;
template <int N>
__global__ __launch_bounds__(256) void k_share_gate(CorpusDev c, LshDev L, uint64_t* __restrict__ gbm, uint32_t n_sub) {
  __shared__ uint32_t s_tok[256 + 16], s_cmp[256 + 16];
  __shared__ double s_qd[256 + 16];
  __shared__ share_mask_t<N> s_keys[kGateCap * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t sub = blockIdx.x; sub < n_sub; sub += gridDim.x) {
    const uint64_t p0 = (uint64_t)sub * 256;
    for (int i = threadIdx.x; i < 256 + N - 1; i += 256) {
      const uint32_t id = c.tok[p0 + i];
      s_tok[i] = id;
      s_cmp[i] = share_comp(L, id);
      s_qd[i] = q_of(L, id);
    }
    __syncthreads();
    bool pass = p0 + threadIdx.x + N <= c.n_tok;
    if (pass) pass = share_gate<N>(L, s_cmp + threadIdx.x, s_qd + threadIdx.x, s_keys + threadIdx.x, kGateCap);
    if (L.diag == 4) pass = false;                 // diagnostics: k_lsh_scan's cost with no window to work on
    const uint64_t b = __ballot(pass);
    if (lane == 0) gbm[(size_t)sub * 4 + wave] = b;
    __syncthreads();
  }
}

// The share rule instead of the key scan: the script windows that hold one of a fan window's keys,
// one by one -- the subset keys as an exact map (smap: buckets of four {key, list}, a full bucket
// spills into the next; slists: a key's script windows behind their number) -- through the pairs'
// test and, what is left, the canonical distance.  A pair within the threshold agrees on a heavy
// set of slots, that set contains one of the fan window's minimal heavy subsets, and the script
// window is in that key's list: every script window within the threshold is met, whatever buckets
// it shares with the fan window.  The windows flagged here are therefore a superset of
// k_lsh_scan's (it flags those with a script window within the threshold in a shared bucket); the
// kernels behind it make a window's neighbour list from its buckets and drop a window whose list
// is empty, as behind the other prefilters.  A window the rule does not constrain, or whose work
// finds no room in the workgroup's lists, is flagged as it is.
// One kernel, a workgroup per sub-tile of 256 windows (k_lsh_scan's, and its bitmap), every stage
// dealt out evenly over the 256 threads -- the work per window is very uneven (most windows end
// at the filter, a few have lists of hundreds of script windows):
//   1  a thread per window: its keys (share_asks), the filter; the keys that are there stay;
//   2  a thread per such key: the map -- (window, list) entries;
//   3  a thread per (window, script window) pair of the entries: the pairs' test, the distance.
// workgroups of k_share_scan per CU: the subsets of a window of up to eight slots are bytes (19 KB of
// LDS: seven, at 72 registers; eight measured slower), above that 16 bits (24 KB: six)
constexpr int share_scan_occupancy(int n) { return n <= 8 ? 7 : 6; }
constexpr int kEnumCap = 20;         // keys per window (six slots have at most 20 minimal heavy subsets)
constexpr int kEnumWork = 512;       // keys that are in the filter, and (window, list) entries, per sub-tile
template <int N>
__global__ __launch_bounds__(256, share_scan_occupancy(N)) void k_share_scan(CorpusDev c, LshDev L, uint64_t* __restrict__ qbm,
                                                    uint32_t* __restrict__ qcnt, uint32_t n_sub) {
  __shared__ uint32_t s_tok[256 + 16], s_cmp[256 + 16], s_sg[256 + 16];
  __shared__ double s_qd[256 + 16], s_ff[256];
  __shared__ __attribute__((aligned(16))) share_mask_t<N> s_keys[kEnumCap * 256];   // stage 1: the subsets asked for; stage 3: the entries' offsets (s_wpref)
  __shared__ uint32_t s_hit[kEnumWork], s_wstart[kEnumWork], s_wmeta[kEnumWork];
  __shared__ uint8_t s_found[256];
  __shared__ uint32_t s_w[4], s_nwork, s_ndist;
  uint32_t* s_wpref = reinterpret_cast<uint32_t*>(s_keys);                   // [kEnumWork + 1]
  static_assert((kEnumWork + 1) * 4 <= kEnumCap * 256 * (int)sizeof(share_mask_t<N>) && kEnumWork == 2 * 256, "the offsets take the subsets' place; two entries per thread");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t sub = blockIdx.x; sub < n_sub; sub += gridDim.x) {
    const uint64_t p0 = (uint64_t)sub * 256;
    for (int i = threadIdx.x; i < 256 + N - 1; i += 256) {
      const uint32_t id = c.tok[p0 + i];
      const uint32_t cm = share_comp(L, id);
      s_tok[i] = id;
      s_cmp[i] = cm;
      s_sg[i] = cm >= FS_WILD ? cm : fs_share_sig(cm, N);
      s_qd[i] = q_of(L, id);
    }
    if (threadIdx.x == 0) { s_nwork = 0; s_ndist = 0; }
    s_found[threadIdx.x] = 0;
    __syncthreads();
    // stage 1
    uint32_t hc = 0;
    bool flag = false;
    if (p0 + threadIdx.x + N <= c.n_tok && L.diag != 4) {
      double ff = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) ff = __dadd_rn(ff, s_qd[threadIdx.x + k]);
      s_ff[threadIdx.x] = ff;
      share_mask_t<N>* list = s_keys + threadIdx.x;
      int cnt = share_asks<N>(L, s_cmp + threadIdx.x, s_qd + threadIdx.x, list, kEnumCap);
      flag = cnt < 0;
      if (L.diag == 10) cnt = 0;                                  // diagnostics: the subsets only
      uint32_t t[N];
      share_terms<N>(s_cmp + threadIdx.x, t);
      for (int j = 0; j < cnt; j += 8) {
        uint32_t m[8], h[8], wd[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { m[u] = j + u < cnt ? list[(j + u) * 256] : 0u; h[u] = share_key_of<N>(t, m[u]); }
#pragma unroll
        for (int u = 0; u < 8; ++u) wd[u] = j + u < cnt ? L.sharef[fs_bloom_word(h[u], L.log2_sharef)] : 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (j + u < cnt && fs_bloom_test(wd[u], h[u])) list[hc++ * 256] = (share_mask_t<N>)m[u];   // (hc <= j + u: behind what is read)
      }
      if (L.diag == 6) hc = 0;                                    // diagnostics: no lists
    }
    {
      uint32_t n_hit;
      const uint32_t base = block_excl_scan(hc, s_w, &n_hit);
      for (uint32_t i = 0; i < hc; ++i) {
        if (base + i < (uint32_t)kEnumWork) s_hit[base + i] = threadIdx.x | (uint32_t)s_keys[i * 256 + threadIdx.x] << 8;
        else flag = true;                                         // (no room: the window goes on as it is)
      }
      if (flag) s_found[threadIdx.x] = 1;
      if (L.share_cnt) {                                          // diagnostics: what passes what (fs_index_share_counts)
        const bool in = p0 + threadIdx.x + N <= c.n_tok;
        const uint64_t b0 = __ballot(in), b1 = __ballot(hc > 0), b2 = __ballot(flag);
        if (lane == 0) {
          atomicAdd(L.share_cnt + 0, (unsigned long long)__popcll(b0));
          atomicAdd(L.share_cnt + 1, (unsigned long long)__popcll(b1));
          atomicAdd(L.share_cnt + 6, (unsigned long long)__popcll(b2));
        }
      }
      __syncthreads();
      // stage 2
      const uint32_t bmask = (1u << L.log2_smap) - 1u;
      n_hit = n_hit < (uint32_t)kEnumWork ? n_hit : (uint32_t)kEnumWork;
      for (uint32_t x = threadIdx.x; x < n_hit; x += 256) {
        const uint32_t t = s_hit[x] & 255u, hm = s_hit[x] >> 8;    // the window and the subset: its key again, from the slots' components
        uint32_t tt[N];
        share_terms<N>(s_cmp + t, tt);
        const uint32_t h = share_key_of<N>(tt, hm);
        uint32_t bkt = fs_wmap_slot(h, L.log2_smap);
        for (int probe = 0;; ++probe) {
          if (probe == 64) { s_found[t] = 1; break; }             // (never seen: the window goes on as it is)
          const uint4* bp = reinterpret_cast<const uint4*>(L.smap + 4 * (size_t)bkt);
          const uint4 a = bp[0], b = bp[1];
          const uint32_t key[4] = {a.x, a.z, b.x, b.z}, val[4] = {a.y, a.w, b.y, b.w};
          for (int e = 0; e < 4; ++e) {
            if (!val[e] || key[e] != h) continue;
            const uint32_t len = L.slists[val[e] - 1].x;           // a list: its length, then its script windows
            const uint32_t at = atomicAdd(&s_nwork, 1u);
            if (at < (uint32_t)kEnumWork && len < (1u << 24)) {
              s_wstart[at] = val[e];
              s_wmeta[at] = t << 24 | len;
            } else {
              s_found[t] = 1;                                      // (no room: the window goes on as it is)
            }
          }
          if (!val[3]) break;                                      // (not full: nothing has spilt past it)
          bkt = (bkt + 1) & bmask;
        }
      }
    }
    __syncthreads();
    // stage 3: the entries' offsets among the sub-tile's pairs, then the pairs, pair j to thread j mod 256
    const uint32_t n_work = s_nwork < (uint32_t)kEnumWork ? s_nwork : (uint32_t)kEnumWork;
    uint32_t mine[2], sum = 0;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint32_t e = threadIdx.x * 2 + u;
      mine[u] = e < n_work ? s_wmeta[e] & 0xFFFFFFu : 0u;
      sum += mine[u];
    }
    uint32_t pairs;
    uint32_t at = block_excl_scan(sum, s_w, &pairs);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint32_t e = threadIdx.x * 2 + u;
      if (e < n_work) s_wpref[e] = at;
      at += mine[u];
    }
    if (threadIdx.x == 0) {
      s_wpref[n_work] = pairs;
      if (L.share_cnt) { atomicAdd(L.share_cnt + 2, (unsigned long long)n_work); atomicAdd(L.share_cnt + 3, (unsigned long long)pairs); }
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < pairs && L.diag != 5; j += 256) {                            // (diagnostics 5: no pairs)
      uint32_t lo = 0, hi = n_work;                               // s_wpref[lo] <= j < s_wpref[hi]
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_wpref[mid] <= j) lo = mid; else hi = mid;
      }
      const uint32_t t = s_wmeta[lo] >> 24;
      if (s_found[t]) continue;                                   // the window has its answer already
      // (a list's entry: the script window and its signature word, one 16-byte load)
      const uint4 it = L.slists[s_wstart[lo] + (j - s_wpref[lo])];
      const double pff = s_ff[t];
      if (!share_pair_possible<N>(L, it.x, (uint64_t)it.z << 32 | it.y, s_sg + t, s_qd + t, pff) || L.diag == 7) continue;   // (diagnostics 7: no distances)
      // what is left needs the distance: all of the sub-tile's at once behind the loop (a distance
      // inside it holds the thread's wave for four more levels of loads in every pass)
      const uint32_t at = atomicAdd(&s_ndist, 1u);
      if (at < (uint32_t)kEnumWork) {
        s_hit[at] = t << 24 | it.x;                               // (s_hit is free since stage 2; script windows < 2^24: 2^18 at most)
      } else {
        double d;
        if (window_distance_flat<N>(L, it.x, s_tok + t, s_qd + t, pff, __dsqrt_rn(pff), &d) && d < L.thr) s_found[t] = 1;
      }
    }
    __syncthreads();
    {
      const uint32_t nd = s_ndist < (uint32_t)kEnumWork ? s_ndist : (uint32_t)kEnumWork;
      for (uint32_t x = threadIdx.x; x < nd; x += 256) {
        const uint32_t t = s_hit[x] >> 24, sw = s_hit[x] & 0xFFFFFFu;
        if (s_found[t]) continue;
        const double pff = s_ff[t];
        double d;
        if (window_distance_flat<N>(L, sw, s_tok + t, s_qd + t, pff, __dsqrt_rn(pff), &d) && d < L.thr) s_found[t] = 1;
      }
    }
    __syncthreads();
    // thread (wave j, lane l) reports window 4 l + j, as k_lsh_scan does: wave j's ballot is bitmap
    // word j of the sub-tile
    const uint64_t b = __ballot(s_found[4 * lane + wave] != 0);
    if (lane == 0) {
      qbm[(size_t)sub * 4 + wave] = b;
      s_w[wave] = __popcll(b);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      qcnt[sub] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
      if (L.share_cnt) {
        atomicAdd(L.share_cnt + 4, (unsigned long long)s_ndist);
        atomicAdd(L.share_cnt + 5, (unsigned long long)(s_w[0] + s_w[1] + s_w[2] + s_w[3]));
      }
    }
    __syncthreads();
  }
}

// 64-bit words of k_lsh_scan's first LDS array: the ballot words of 256 windows in phase 1; in phase 2
// s_ff, s_flag and behind them (384 words in) the share rule's per-token arrays
__host__ __device__ inline int lsh_scan_bal_words(int NW) { return 256 * NW > 800 ? 256 * NW : 800; }
__host__ __device__ inline int lsh_scan_pref_words(int H) { return 256 * H + 1 > 512 ? 256 * H + 1 : 512; }

__global__ __launch_bounds__(256) void k_lsh_scan(CorpusDev c, LshDev L, const uint64_t* __restrict__ gbm,
                                                  uint64_t* __restrict__ qbm,
                                                  uint32_t* __restrict__ qcnt, uint32_t n_sub) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  const int NW = (L.C + 63) >> 6;                    // ballot words per window
  // (LDS is what bounds this kernel's occupancy: 40 KB per workgroup, four per CU.  The
  // ballot words are dead once the keys are assembled and then hold the phase-2 arrays
  // s_ff and s_flag; s_bound, phase 1 only, lies where phase 2 keeps its pair offsets.)
  uint64_t* s_bal = reinterpret_cast<uint64_t*>(s_raw);                 // [256][NW]
  const int bal_words = lsh_scan_bal_words(NW);                         // (room for s_ff + s_flag + the share rule's arrays)
  uint32_t* s_key = reinterpret_cast<uint32_t*>(s_bal + bal_words);     // [256][H]
  uint32_t* s_tok = s_key + 256 * L.H;                                  // [256 + 16]
  uint32_t* s_pref = s_tok + 256 + 16;                                  // [256 * H + 1] pair offsets
  double* s_ff = reinterpret_cast<double*>(s_bal);                      // [256]  (phase 2)
  uint32_t* s_flag = reinterpret_cast<uint32_t*>(s_bal + 256);          // [256]  (phase 2)
  float* s_bound = reinterpret_cast<float*>(s_pref);                    // [256]  (phase 1)
  uint32_t* s_list = s_pref + 256;                                      // [256]  the windows phase 1 makes keys for
  // the share rule's test of the pairs: component id and squared norm per token of the sub-tile
  uint32_t* s_cmp2 = reinterpret_cast<uint32_t*>(s_bal + 384);          // [256 + 16]  (phase 2)
  double* s_qd2 = reinterpret_cast<double*>(s_cmp2 + 272);              // [256 + 16]  (phase 2)
  const bool pair_test = (L.share_flags & 2) != 0;
  __shared__ uint32_t s_cnt[4];
  __shared__ uint64_t s_gate[4];
  __shared__ uint32_t s_nlist;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int n = L.n;
  for (uint32_t sub = blockIdx.x; sub < n_sub; sub += gridDim.x) {
    const uint64_t p0 = (uint64_t)sub * 256;
    for (int i = threadIdx.x; i < 256 + n - 1; i += 256) s_tok[i] = c.tok[p0 + i];
    __syncthreads();
    {
      // the windows that need keys at all: those inside the token stream -- and, under the share
      // rule, through its gate
      bool pass = p0 + threadIdx.x + n <= c.n_tok;
      if (gbm) pass = pass && ((gbm[(size_t)sub * 4 + wave] >> lane) & 1ull);
      const uint64_t b = __ballot(pass);
      if (lane == 0) s_gate[wave] = b;
      __syncthreads();
      uint32_t before = 0;
      for (int j = 0; j < wave; ++j) before += __popcll(s_gate[j]);
      if (pass) s_list[before + __popcll(b & ((1ull << lane) - 1ull))] = threadIdx.x;
      if (threadIdx.x == 255) s_nlist = before + __popcll(b);
    }
    __syncthreads();
    const int n_list = __builtin_amdgcn_readfirstlane((int)s_nlist);
    if ((int)threadIdx.x < n_list) {
      // per window: the float32 decision bound, or -1 when the window needs float64
      const int w = (int)s_list[threadIdx.x];
      float m = 0.0f;
      int terms = 0;
      const bool f64 = L.atab32 == nullptr || (L.diag & 64);       // (diag 64: float64 for OOV windows as before round 5)
      bool oov = false;
      for (int k = 0; k < n; ++k) {
        const uint32_t id = s_tok[w + k];
        oov = oov || (id & FS_OOV_FLAG);
        if (!f64 || !(id & FS_OOV_FLAG)) row32_bound(L, k, id, &m, &terms);
      }
      // (an out-of-vocabulary slot is up to three float32 addends instead of one: the bound's
      // n becomes the number of addends)
      s_bound[w] = (L.atab32 == nullptr || (f64 && oov)) ? -1.0f : L.bound_scale * m * ((float)terms / (float)n);
      s_key[w] = oov ? 1u : 0u;                    // (phase 1 only: s_key is written behind it)
    }
    __syncthreads();
    // phase 1: a wave takes four windows at a time; lane l holds projection columns
    // 4l .. 4l+3 of each, so one 16-byte load per lane fetches a whole table row
    // (848 B at the default 210 columns) per wave instruction.  Only the sign of a
    // projection matters, so the float32 copy of the tables decides it whenever the
    // float32 sum is farther from zero than its worst-case distance to the canonical
    // float64 sum:  |s32 - s64| <= n * 2^-23 * sum_k max_c|A[k][t_k][c]|  (rounding
    // of the n table entries to float32 plus n-1 float32 additions; the float64
    // additions contribute 2^-53 terms).  A window with any column inside twice
    // that distance, or with an out-of-vocabulary token, is redone in float64.
    uint32_t* s_bits = reinterpret_cast<uint32_t*>(s_bal);           // [256][2 NW]
    for (int c0 = 0; c0 < L.C && L.diag != 2; c0 += 256) {
      const int col = c0 + 4 * lane;
      const int left = L.C - col;                                    // columns this lane owns
      const uint32_t cmask = left >= 4 ? 0xFu : left > 0 ? (1u << left) - 1 : 0u;
      const int colc = left > 0 ? col : 0;
      const bool store = (lane & 7) == 0 && (c0 >> 5) + (lane >> 3) < 2 * NW;
      for (int g = wave; 4 * g < n_list; g += 4) {
        // (four windows of the list at a time; the last group repeats the list's last window)
        int wl[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          wl[u] = __builtin_amdgcn_readfirstlane((int)s_list[4 * g + u < n_list ? 4 * g + u : n_list - 1]);
        float bnd[4];
        bool fast = true;
#pragma unroll
        for (int u = 0; u < 4; ++u) { bnd[u] = s_bound[wl[u]]; fast = fast && bnd[u] >= 0.0f; }
        uint32_t nib[4];
        bool redo[4] = {true, true, true, true};
        if (fast) {                                                  // wave-uniform
          float4 acc[4];
          // (table rows only -- the common case -- with no branch between the loads; a group
          // of four windows that holds an out-of-vocabulary token takes the rows through row32)
          const bool plain = !(s_key[wl[0]] | s_key[wl[1]] | s_key[wl[2]] | s_key[wl[3]]);
          if (plain) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
              acc[u] = *reinterpret_cast<const float4*>(L.atab32 + (size_t)s_tok[wl[u]] * L.Cp + colc);
            for (int k = 1; k < n; ++k) {
              float4 r[4];
#pragma unroll
              for (int u = 0; u < 4; ++u)
                r[u] = *reinterpret_cast<const float4*>(
                    L.atab32 + ((size_t)k * L.V + s_tok[wl[u] + k]) * L.Cp + colc);
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                acc[u].x = __fadd_rn(acc[u].x, r[u].x); acc[u].y = __fadd_rn(acc[u].y, r[u].y);
                acc[u].z = __fadd_rn(acc[u].z, r[u].z); acc[u].w = __fadd_rn(acc[u].w, r[u].w);
              }
            }
          } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
              acc[u] = row32(L, 0, s_tok[wl[u]], colc);
            for (int k = 1; k < n; ++k) {
              float4 r[4];
#pragma unroll
              for (int u = 0; u < 4; ++u)
                r[u] = row32(L, k, s_tok[wl[u] + k], colc);
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                acc[u].x = __fadd_rn(acc[u].x, r[u].x); acc[u].y = __fadd_rn(acc[u].y, r[u].y);
                acc[u].z = __fadd_rn(acc[u].z, r[u].z); acc[u].w = __fadd_rn(acc[u].w, r[u].w);
              }
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const uint32_t sure = (fabsf(acc[u].x) > bnd[u] ? 1u : 0u) | (fabsf(acc[u].y) > bnd[u] ? 2u : 0u) |
                                  (fabsf(acc[u].z) > bnd[u] ? 4u : 0u) | (fabsf(acc[u].w) > bnd[u] ? 8u : 0u);
            nib[u] = ((acc[u].x > 0.0f ? 1u : 0u) | (acc[u].y > 0.0f ? 2u : 0u) |
                      (acc[u].z > 0.0f ? 4u : 0u) | (acc[u].w > 0.0f ? 8u : 0u)) & cmask;
            redo[u] = __any((~sure & cmask) != 0u);                  // wave-uniform
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (redo[u]) {
            uint32_t bits = 0;
            for (int j = 0; j < 4; ++j) {
              if (!((cmask >> j) & 1u)) continue;
              double acc = a_value(L, 0, s_tok[wl[u]], col + j);
              for (int k = 1; k < n; ++k)
                acc = __dadd_rn(acc, a_value(L, k, s_tok[wl[u] + k], col + j));
              bits |= acc > 0.0 ? 1u << j : 0u;
            }
            nib[u] = bits;
          }
          // eight lanes -> one 32-bit piece of the window's column bit string
          uint32_t x = nib[u];
          x |= (uint32_t)__shfl_down((int)x, 1) << 4;
          x |= (uint32_t)__shfl_down((int)x, 2) << 8;
          x |= (uint32_t)__shfl_down((int)x, 4) << 16;
          if (store) s_bits[(size_t)wl[u] * 2 * NW + (c0 >> 5) + (lane >> 3)] = x;
        }
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_list * L.H; i += 256) {
      const int li = i / L.H, h = i - li * L.H, w = (int)s_list[li];
      s_key[w * L.H + h] = assemble_key(s_bal + w * NW, h, L.B);
    }
    __syncthreads();
    // phase 2: "is any bucket candidate of the window within the threshold?"  The 256 x H
    // buckets of the sub-tile hold very different numbers of candidates, so they are not
    // walked window by window: thread w looks up its window's H bucket ranges, a block
    // scan turns the sizes into offsets, and the (window, candidate) pairs of the whole
    // sub-tile are then dealt out evenly, pair j to thread j mod 256 (the bucket of a
    // pair is found by binary search over the offsets in LDS).  A pair that is within
    // the threshold sets its window's flag.
    {
      const int w = threadIdx.x;
      const bool valid = ((s_gate[w >> 6] >> (w & 63)) & 1ull) && L.diag != 1;
      if (pair_test)
        for (int i = threadIdx.x; i < 256 + n - 1; i += 256) {
          const uint32_t id = s_tok[i];
          const uint32_t cm = share_comp(L, id);
          s_cmp2[i] = cm >= FS_WILD ? cm : fs_share_sig(cm, n);
          s_qd2[i] = q_of(L, id);
        }
      const uint32_t nb1 = (1u << L.B) + 1;
      uint32_t sum = 0;
      for (int h = 0; h < L.H; ++h) {
        uint32_t e0 = 0, cntb = 0;
        if (valid) {
          const uint32_t* o = L.boff + (size_t)h * nb1 + s_key[w * L.H + h];
          e0 = o[0];
          cntb = o[1] - e0;
        }
        s_key[w * L.H + h] = e0;                 // the key is not needed again
        s_pref[w * L.H + h] = sum;               // offset inside the window, for now
        sum += cntb;
      }
      double ff = 0.0;
      for (int k = 0; k < n; ++k) ff = __dadd_rn(ff, q_of(L, s_tok[w + k]));
      s_ff[w] = ff;
      s_flag[w] = 0;
      uint32_t total;
      const uint32_t base = block_excl_scan(sum, s_cnt, &total);
      for (int h = 0; h < L.H; ++h) s_pref[w * L.H + h] += base;
      if (w == 255) s_pref[256 * L.H] = total;
      __syncthreads();
      const uint32_t n_b = 256u * (uint32_t)L.H;
      // (round 5 measured this loop two and four pairs at a time, level by level -- bucket entry,
      // the window's record, the first slot's pair-table entry, window_distance's first early exit
      // on those: 2.58 and 3.96 ms per search against 2.41 on the realistic table, where the loop
      // is 58 % of the search.  It is not the latency of one thread's chain that bounds it but
      // the number of random sectors: 134 M pairs per 2 M windows, each with a pair-table entry
      // out of a table far larger than the caches.  The window's record is not one of them: with
      // the records carried in the bucket entries (32 B, in bucket order) the kernel took 5.87 ms
      // against 5.86 ms on 4 M windows.  By switches (FS_LSH_DIAG 1, 3) on those 4 M windows:
      // keys 1.4 ms, the walk without distances 0.7 ms, the distances 2.5 ms)
      for (uint32_t j = threadIdx.x; j < total; j += 256) {
        uint32_t lo = 0, hi = n_b;               // s_pref[lo] <= j < s_pref[hi]
        while (hi - lo > 1) {
          const uint32_t mid = (lo + hi) >> 1;
          if (s_pref[mid] <= j) lo = mid; else hi = mid;
        }
        const uint32_t pw = lo / (uint32_t)L.H, ph = lo - pw * (uint32_t)L.H;
        if (s_flag[pw]) continue;                // the window has its answer already
        const uint32_t sidx = L.bids[(size_t)ph * L.W + s_key[lo] + (j - s_pref[lo])];
        if (L.diag == 3) continue;               // diagnostics: bucket walk only
        const double pff = s_ff[pw];
        if (pair_test && !share_pair_possible<0>(L, sidx, L.ssig[sidx], s_cmp2 + pw, s_qd2 + pw, pff)) continue;
        double d;
        if (window_distance(L, sidx, s_tok + pw, nullptr, pff, __dsqrt_rn(pff), &d) && d < L.thr) s_flag[pw] = 1;
      }
      __syncthreads();
    }
    // thread (wave j, lane l) reports window 4 l + j, so that wave j's ballot is bitmap
    // word j of the sub-tile
    const bool flag = s_flag[4 * lane + wave] != 0;
    const uint64_t b = __ballot(flag);
    if (lane == 0) {
      qbm[(size_t)sub * 4 + wave] = b;
      s_cnt[wave] = __popcll(b);
    }
    __syncthreads();
    if (threadIdx.x == 0) qcnt[sub] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    __syncthreads();
  }
}

// Is script window s, which has the ids of fan window f in every slot but k, within the
// threshold?  The canonical distance of window_distance with n - 1 slots known to add q(v_t);
// true also when the premise does not hold in a way that cannot be decided here (the caller
// then takes the full path).  A key collision (other slots differ) is not a neighbour.
template <int NW>
__device__ __forceinline__ bool one_slot_within(const LshDev& L, uint32_t s, int k, const Ids16& f) {
  Ids16 u;
  load_ids(L.stok + s, L.n, &u);
  uint32_t uk = 0, fk = 0;
  bool agree = true;
#pragma unroll
  for (int t = 0; t < NW; ++t)
    if (t < L.n) {
      if (t == k) { uk = u.v[t]; fk = f.v[t]; }
      else agree = agree && u.v[t] == f.v[t];
    }
  if (!agree) return false;
  if (uk == fk) return true;
  double q[NW];
#pragma unroll
  for (int t = 0; t < NW; ++t)
    if (t < L.n) q[t] = L.q[f.v[t]];
  const double g = g_of(L, uk, fk);
  const fs_swin sw = L.sw[s];
  double ff = 0.0, sf = 0.0;
#pragma unroll
  for (int t = 0; t < NW; ++t)
    if (t < L.n) {
      ff = __dadd_rn(ff, q[t]);
      sf = __dadd_rn(sf, t == k ? g : q[t]);
    }
  const double d = __dsub_rn(1.0, __ddiv_rn(sf, __dmul_rn(sw.rss, __dsqrt_rn(ff))));
  return !(d == d) || d < L.thr;
}

// One lane per candidate, in front of k_lsh_verify: most candidates end here.
//   cg[i] = FS_NONE      no neighbour within the threshold
//   cg[i] = 0            a record: cbest[i], cw[i] (the record of its n-gram, k_lsh_gramtab)
//   cg[i] = FS_PENDING   k_lsh_verify works the window out, a wave at a time
// A kernel of its own: k_lsh_verify carries the scratch arrays and registers of the neighbour
// lists and the Levenshtein code, which these steps do not need; consecutive candidates sit
// in consecutive lanes, so the per-candidate arrays move in whole cache lines.
constexpr uint32_t FS_PENDING = 0xFFFFFFFEu;
// The one-slot-wildcard keys of the window at `p` (made of the vector ids, or of the
// component ids: L.wild_tok): terms and fold for the caller, true when one of the n keys is in
// the grouped filter (three 16-byte blocks, requested together).
template <int NW>
__device__ __forceinline__ bool sift_keys(const CorpusDev& c, const LshDev& L, uint64_t p,
                                          uint32_t* term, uint32_t* fold_out, bool probe) {
  Ids16 kf;
  load_ids((L.wild_tok ? L.wild_tok : c.tok) + p, L.n, &kf);
  uint32_t fold = 0, gfold[3] = {0, 0, 0};
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    term[k] = 0;
    if (k < L.n) {
      term[k] = fs_rotl(fs_premix(kf.v[k]), fs_rot_of(L.n - 1 - k));
      fold ^= term[k];
      gfold[fs_wild_group(k, L.n)] ^= term[k];
    }
  }
  *fold_out = fold;
  if (!probe) return true;
  const uint4* wb = reinterpret_cast<const uint4*>(L.wild);
  uint4 blk[3];
#pragma unroll
  for (int X = 0; X < 3; ++X) blk[X] = wb[fs_wild_block(fold ^ gfold[X], X, L.log2_wild)];
  bool pass = false;
#pragma unroll
  for (int k = 0; k < NW; ++k)
    if (k < L.n) {
      const uint32_t h = fs_wild_fkey(fold, term[k], k);
      const int X = fs_wild_group(k, L.n);
      const uint4 q = X == 0 ? blk[0] : X == 1 ? blk[1] : blk[2];
      pass = pass || ((q.x >> fs_wild_fbit(h, 0)) & (q.y >> fs_wild_fbit(h, 1)) &
                      (q.z >> fs_wild_fbit(h, 2)) & (q.w >> fs_wild_fbit(h, 3)) & 1u);
    }
  return pass;
}

// One lane per candidate, in two stages.  Stage 1, every candidate: the wildcard-key filter
// (one level of loads behind the candidate's position and ids).  Most candidates end there --
// 88 % at n = 8, 84 % over component ids -- and the deeper steps (exact table, one-slot map:
// five to eight more levels of dependent loads) ran at a tenth of the lanes while every wave
// had a survivor to wait for.  So the survivors queue up in LDS and stage 2 takes them 256 at
// a time, a full lane each (round 4: 90 -> 40 us per C2 batch at n = 8).
// k_lsh_sift's second stage for one candidate per thread (il = FS_NONE: none; every thread of
// the workgroup calls it: it holds barriers): the per-n-gram record, the exact one-slot map, or
// onto the pending list.  Shared by k_lsh_sift and k_lsh_sift2.
struct SiftOut {
  uint32_t* cg; uint32_t* cw; fs_best* cbest;
  const unsigned long long* tab_best; const uint32_t* tab_cnt;
  uint32_t* pend; uint32_t* pend_cnt;
  uint32_t* s_pn; uint32_t* s_pbase;          // LDS words of the workgroup
};
template <int NW, bool WMAP>
__device__ __forceinline__ void sift_stage2(const CorpusDev& c, const LshDev& L, const GramIndexDev& g,
                                            const SiftOut& o, uint32_t il, uint64_t p_in, uint32_t* matches_io) {
  const int lane = threadIdx.x & 63;
  uint32_t* const cg = o.cg; uint32_t* const cw = o.cw; fs_best* const cbest = o.cbest;
  const unsigned long long* const tab_best = o.tab_best; const uint32_t* const tab_cnt = o.tab_cnt;
  uint32_t* const pend = o.pend; uint32_t* const pend_cnt = o.pend_cnt;
  uint32_t& s_pn = *o.s_pn; uint32_t& s_pbase = *o.s_pbase;
  uint32_t& matches = *matches_io;
  bool live = il != FS_NONE;
  // 2. A window with the ids of a script n-gram (and the strings of those ids) takes the
  //    n-gram's record of this string table (k_lsh_gramtab): no bucket is walked for it.
  uint32_t gram = FS_NONE;
  const uint64_t p = live ? p_in : 0;
  if (tab_cnt && live && !(L.diag & 128)) {
    uint32_t w = 0, kept = 0;
    gram = verify_window(c, g, p, &w, &kept);
    if (gram != FS_NONE) {
      const uint32_t have = tab_cnt[gram];
      if (have == 1) {
        cg[il] = FS_NONE;                             // (no neighbour within the threshold)
      } else {
        const uint4* m = reinterpret_cast<const uint4*>(tab_best + 4 * (size_t)gram);
        uint4* dst = reinterpret_cast<uint4*>(&cbest[il]);
        dst[0] = m[0]; dst[1] = m[1];
        cg[il] = 0;
        cw[il] = w;
        matches += have - 1;
      }
      live = false;
    }
  }
  // 3. Not a script n-gram itself: enumerate the script n-grams that equal the window in all
  //    slots but one (every neighbour within the threshold is one of them: m_min = n - 1) and
  //    take their canonical distances.  None within the threshold: whatever the buckets hold,
  //    nothing survives the threshold, and the window needs no LSH work.  One 32-byte bucket
  //    of the map per slot, all n requested together; a window with more than two such
  //    n-grams, or a full bucket in its way, is left to k_lsh_verify.
  if (WMAP && live && L.wild && p + L.n <= c.n_tok && !(L.diag & 256)) {
    uint32_t term[NW], fold = 0;
    sift_keys<NW>(c, L, p, term, &fold, false);
    Ids16 f;
    load_ids(c.tok + p, L.n, &f);
    uint32_t s0 = 0, s1 = 0, nh = 0;
    int k0 = 0, k1 = 0;
    bool possible = false;
#pragma unroll
    for (int k = 0; k < NW; ++k)
      if (k < L.n) {
        const uint32_t h = fs_wild_key(fold, term[k], k);
        const uint4* bp = reinterpret_cast<const uint4*>(L.wmap + 4 * (size_t)fs_wmap_slot(h, L.log2_wmap));
        const uint4 a = bp[0], b = bp[1];
        const uint32_t key[4] = {a.x, a.z, b.x, b.z}, val[4] = {a.y, a.w, b.y, b.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (val[e] && key[e] == h) {
            if (nh == 0) { s0 = val[e] - 1; k0 = k; }
            else if (nh == 1) { s1 = val[e] - 1; k1 = k; }
            ++nh;
          }
        possible = possible || val[3] != 0;       // (filled in order: the bucket is full)
      }
    possible = possible || nh > 2;
    if (L.diag & 512) possible = possible || nh > 0;                 // diagnostics: no distances here
    if (!possible && nh > 0) possible = one_slot_within<NW>(L, s0, k0, f);
    if (!possible && nh > 1) possible = one_slot_within<NW>(L, s1, k1, f);
    if (!possible) { cg[il] = FS_NONE; live = false; }
  }
  // what is left: onto the list k_lsh_verify deals out window by window (pending windows
  // come in runs, the boundary windows of one quoted passage, so dealing out blocks of
  // candidates leaves a few waves with most of the work)
  // (one addition to the list's counter per workgroup: five thousand waves adding to the one
  // address took 5 ns each, a third of the kernel)
  const uint64_t pb = __ballot(live);
  uint32_t wbase = 0;
  if (pb && lane == 0) wbase = atomicAdd(&s_pn, (uint32_t)__popcll(pb));      // LDS
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t n_p = s_pn;
    s_pbase = n_p ? atomicAdd(pend_cnt, n_p) : 0u;
    s_pn = 0;
  }
  __syncthreads();
  if (live) {
    const uint32_t base = s_pbase + (uint32_t)__builtin_amdgcn_readlane((int)wbase, 0);
    cg[il] = FS_PENDING;
    pend[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(pb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pb, 0u))] = (uint32_t)il;
  }
}

template <int NW, bool WMAP, int NN>
__global__ __launch_bounds__(256, 5) void k_lsh_sift(CorpusDev c, LshDev L, GramIndexDev g,
                                                  const uint32_t* __restrict__ cpos, NSrc nc,
                                                  uint32_t* __restrict__ cg, uint32_t* __restrict__ cw,
                                                  fs_best* __restrict__ cbest,
                                                  uint32_t* __restrict__ bmatch,
                                                  const unsigned long long* __restrict__ tab_best,
                                                  const uint32_t* __restrict__ tab_cnt,
                                                  uint32_t* __restrict__ pend,
                                                  uint32_t* __restrict__ pend_cnt) {
  __shared__ uint32_t s_w32[4];
  __shared__ uint32_t s_q[1024];         // survivors of stage 1 (candidate numbers): at most 255 + 3 * 256
  __shared__ uint32_t s_qn, s_pn, s_pbase;
  const uint32_t total = nc.get();
  const int lane = threadIdx.x & 63;
  uint32_t matches = 0;
  if (L.diag & 8192) {                         // diagnostics: the launch by itself
    if (threadIdx.x == 0) bmatch[blockIdx.x] = 0;
    return;
  }
  if (threadIdx.x == 0) { s_qn = 0; s_pn = 0; }
  __syncthreads();
  // stage 2 for one queued candidate (FS_NONE: none); every thread of the workgroup calls it
  const SiftOut so{cg, cw, cbest, tab_best, tab_cnt, pend, pend_cnt, &s_pn, &s_pbase};
  auto stage2 = [&](uint32_t il) {
    sift_stage2<NW, WMAP>(c, L, g, so, il, il != FS_NONE ? (uint64_t)cpos[il] : 0ull, &matches);
  };
  // U candidates per lane and pass, their loads level by level: positions, ids, filter blocks
  constexpr int U = NW <= 8 ? 3 : 2;
  const uint64_t pass = (uint64_t)gridDim.x * 256;
  for (uint64_t i0 = (uint64_t)blockIdx.x * 256; i0 < total; i0 += pass * U) {
    uint64_t il[U], p[U];
    bool live[U], probe[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      il[u] = i0 + (uint64_t)u * pass + threadIdx.x;
      live[u] = il[u] < total;
      p[u] = (L.wild && live[u]) ? cpos[il[u]] : 0;
    }
    // 1. (no OOV anywhere, at most one slot may differ) a window none of whose n one-slot-
    //    wildcard keys is a script window's key has no neighbour within the threshold
    if (L.wild) {
      uint32_t kf[U][NW];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        probe[u] = live[u] && p[u] + L.n <= c.n_tok;
        if (L.diag & 4096) { if (live[u]) cg[il[u]] = FS_NONE; live[u] = false; probe[u] = false; }   // diagnostics
        // (the window start is only 4-byte aligned; the buffers are padded: fs_device.h, load_ids)
        const uint4* src = reinterpret_cast<const uint4*>((L.wild_tok ? L.wild_tok : c.tok) + (probe[u] ? p[u] : 0));
#pragma unroll
        for (int q4 = 0; q4 < NW / 4; ++q4) {
          uint4 t = make_uint4(0, 0, 0, 0);
          if (q4 < 2 || L.n > 8) t = src[q4];
          kf[u][4 * q4] = t.x; kf[u][4 * q4 + 1] = t.y; kf[u][4 * q4 + 2] = t.z; kf[u][4 * q4 + 3] = t.w;
        }
      }
      const uint4* wb = reinterpret_cast<const uint4*>(L.wild);
      uint4 blk0[U], blk1[U], blk2[U];
      uint32_t fold[U];
      const int n = NN ? NN : L.n;              // (NN: the window size at compile time -- groups and rotations are constants then)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uint32_t g0 = 0, g1 = 0, g2 = 0;
        fold[u] = 0;
#pragma unroll
        for (int k = 0; k < NW; ++k)
          if (k < n) {
            const uint32_t t = fs_rotl(fs_premix(kf[u][k]), fs_rot_of(n - 1 - k));
            const int X = fs_wild_group(k, n);
            fold[u] ^= t;
            g0 ^= X == 0 ? t : 0u; g1 ^= X == 1 ? t : 0u; g2 ^= X == 2 ? t : 0u;
          }
        if (L.diag & 2048) { g0 = g1 = g2 = fold[u] ^ (uint32_t)threadIdx.x; }    // diagnostics: the same blocks for every wave
        blk0[u] = wb[fs_wild_block(fold[u] ^ g0, 0, L.log2_wild)];
        blk1[u] = wb[fs_wild_block(fold[u] ^ g1, 1, L.log2_wild)];
        blk2[u] = wb[fs_wild_block(fold[u] ^ g2, 2, L.log2_wild)];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uint32_t any = 0;                       // bit 0: one of the keys is in the filter
#pragma unroll
        for (int k = 0; k < NW; ++k)
          if (k < n) {
            const uint32_t t = fs_rotl(fs_premix(kf[u][k]), fs_rot_of(n - 1 - k));
            const uint32_t h = fs_wild_fkey(fold[u], t, k);
            const int X = fs_wild_group(k, n);
            uint4 q;
            q.x = X == 0 ? blk0[u].x : X == 1 ? blk1[u].x : blk2[u].x;
            q.y = X == 0 ? blk0[u].y : X == 1 ? blk1[u].y : blk2[u].y;
            q.z = X == 0 ? blk0[u].z : X == 1 ? blk1[u].z : blk2[u].z;
            q.w = X == 0 ? blk0[u].w : X == 1 ? blk1[u].w : blk2[u].w;
            any |= shr_by_byte<0>(q.x, h) & shr_by_byte<1>(q.y, h) & shr_by_byte<2>(q.z, h) & shr_by_byte<3>(q.w, h);
          }
        bool pass1 = (any & 1u) != 0;
        if (L.diag & 1024) pass1 = false;                                          // diagnostics: nothing survives
        if (probe[u] && !pass1) { cg[il[u]] = FS_NONE; live[u] = false; }
      }
    }
    // the survivors onto the queue (a slot per wave's worth of them).  The barrier keeps a
    // fast wave's additions to s_qn behind every wave's read of it at the end of the round
    // before (the read decides a workgroup-uniform branch around barriers)
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t sb = __ballot(live[u]);
      uint32_t base = 0;
      if (sb) {
        const int leader = __ffsll((unsigned long long)sb) - 1;
        if (lane == leader) base = atomicAdd(&s_qn, (uint32_t)__popcll(sb));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
        if (live[u])
          s_q[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(sb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sb, 0u))] = (uint32_t)il[u];
      }
    }
    __syncthreads();
    // stage 2 once 256 are queued, all of them at the end
    const bool last = i0 + pass * U >= total;
    uint32_t qn = s_qn;
    while (qn >= 256 || (last && qn > 0)) {
      const uint32_t take = qn < 256 ? qn : 256u;
      const uint32_t mine = threadIdx.x < take ? s_q[qn - take + threadIdx.x] : FS_NONE;
      __syncthreads();
      if (threadIdx.x == 0) s_qn = qn - take;
      stage2(mine);
      __syncthreads();
      qn = s_qn;
    }
  }
  uint32_t tot;
  block_excl_scan(matches, s_w32, &tot);
  if (threadIdx.x == 0) {
    bmatch[blockIdx.x] = tot;
    // (the sums are read kNB at a time: the workgroups that were not launched have none)
    for (uint32_t b = blockIdx.x + gridDim.x; b < (uint32_t)kNB; b += gridDim.x) bmatch[b] = 0;
  }
}

// k_lsh_sift2 (round 5): behind k_near_sift, which has put the candidates that pass the wildcard
// filter into one list per wave range.  Numbers them across the ranges (chunk sums: four ranges
// a chunk), writes the flat arrays the kernels behind expect -- an eighth of the entries
// k_expand used to make -- and takes k_lsh_sift's second stage for each.  Every workgroup takes
// an equal share of the numbered survivors (its place among the chunks by a search in the
// chunk sums' prefix, which each workgroup makes for itself in LDS): one pass of full waves.
// (A workgroup per chunk measured 45 us at n = 8 and 67 at n = 10: a C2 batch leaves 70 to 100
// survivors per chunk, so the deep steps ran at a third of the lanes, twice over.)
template <int NW, bool WMAP>
__global__ __launch_bounds__(256, 5) void k_lsh_sift2(CorpusDev c, LshDev L, GramIndexDev g,
                                                   const uint32_t* __restrict__ slist, uint32_t caps,
                                                   const uint32_t* __restrict__ scount,
                                                   const uint32_t* __restrict__ bsum,
                                                   uint32_t* __restrict__ cpos, uint32_t ccap,
                                                   uint32_t* __restrict__ cg, uint32_t* __restrict__ cw,
                                                   fs_best* __restrict__ cbest,
                                                   uint32_t* __restrict__ bmatch,
                                                   const unsigned long long* __restrict__ tab_best,
                                                   const uint32_t* __restrict__ tab_cnt,
                                                   uint32_t* __restrict__ pend, fs_status* st) {
  static_assert(kNB == 8 * 256, "eight chunk sums per thread");
  __shared__ uint32_t s_w32[4];
  __shared__ uint32_t s_pn, s_pbase;
  __shared__ uint32_t s_pre[kNB + 1];                 // survivors in front of chunk i
  uint32_t matches = 0;
  if (threadIdx.x == 0) s_pn = 0;
  {
    uint32_t v[8], sum = 0;
    const uint4* src = reinterpret_cast<const uint4*>(bsum + 8 * threadIdx.x);
    const uint4 a = src[0], b = src[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += v[k];
    uint32_t tot_all;
    uint32_t run = block_excl_scan(sum, s_w32, &tot_all);
#pragma unroll
    for (int k = 0; k < 8; ++k) { s_pre[8 * threadIdx.x + k] = run; run += v[k]; }
    if (threadIdx.x == 0) s_pre[kNB] = tot_all;
  }
  __syncthreads();
  const uint32_t total = s_pre[kNB];
  const SiftOut so{cg, cw, cbest, tab_best, tab_cnt, pend, &st->lsh_pending, &s_pn, &s_pbase};
  // this workgroup's share, in whole steps of 256
  const uint32_t steps = (total + 255) / 256;
  const uint32_t per = (steps + gridDim.x - 1) / gridDim.x;
  const uint32_t lo = (uint32_t)min((uint64_t)blockIdx.x * per * 256, (uint64_t)total);
  const uint32_t hi = (uint32_t)min((uint64_t)(blockIdx.x + 1) * per * 256, (uint64_t)total);
  for (uint32_t t0 = lo; t0 < hi; t0 += 256) {        // (workgroup-uniform)
    const uint32_t il = t0 + threadIdx.x;
    bool live = il < hi && il < ccap;                 // (beyond the arrays: n_cands says so, the search is repeated)
    uint32_t p = 0;
    if (live) {
      uint32_t a = 0, b = kNB;                        // the last chunk with s_pre[chunk] <= il
      while (b - a > 1) {
        const uint32_t mid = (a + b) >> 1;
        if (s_pre[mid] <= il) a = mid; else b = mid;
      }
      uint32_t off = il - s_pre[a];
      const uint4 cn = *reinterpret_cast<const uint4*>(scount + 4 * a);
      const uint32_t c0 = min(cn.x, caps), c1 = min(cn.y, caps), c2 = min(cn.z, caps);
      uint32_t r = 0;
      if (off >= c0) { off -= c0; r = 1; if (off >= c1) { off -= c1; r = 2; if (off >= c2) { off -= c2; r = 3; } } }
      p = slist[(size_t)(4 * a + r) * caps + off];
      cpos[il] = p;
    }
    sift_stage2<NW, WMAP>(c, L, g, so, live ? il : FS_NONE, p, &matches);
  }
  if (blockIdx.x == 0) {                              // for the kernels behind and the host
    uint32_t over = 0;
    for (uint32_t i = threadIdx.x; i < 4u * kNB; i += 256) over = max(over, scount[i]);
    if (threadIdx.x == 0) st->n_cands = total;
    if (over > caps) atomicMax(&st->max_recs, over);  // a range's list was too short (rare: the search is repeated)
  }
  uint32_t tot;
  block_excl_scan(matches, s_w32, &tot);
  if (threadIdx.x == 0) {
    bmatch[blockIdx.x] = tot;
    for (uint32_t b = blockIdx.x + gridDim.x; b < (uint32_t)kNB; b += gridDim.x) bmatch[b] = 0;
  }
}

// The LDS a wave needs for one window (private to the wave).
struct LshWaveLds {
  uint64_t* bal;      // [32] sign ballots of the projection columns
  uint32_t* key;      // [64]
  uint32_t* top_s;    // [64] kept matches in NearestFilter order
  double* top_d;      // [64]
  uint32_t* lev;      // [64]
  uint32_t *la, *lb;  // [FS_LEV_MAX + 2] Levenshtein operands
  uint32_t* f;        // [FS_MAX_WINDOW] vector ids of the window
  uint32_t* fs;       // [FS_MAX_WINDOW] string ids of the window
  int* n;             // [1]
  uint32_t *pre, *e0; // [64]
  double* qf;         // [FS_MAX_WINDOW] q of the window's slots
};

// What the reference returns for one fan window, worked out by a whole wave (all 64 lanes
// call it with the same arguments): keys of the window (search.py:176 -> nearpy), the
// buckets' members with their canonical distances, threshold, NearestFilter, the
// Levenshtein distance of every kept match (search.py:189-190) and the first minimum of
// dist * lev in rank order.  S.f / S.fs hold the window's vector and string ids.  Returns the
// number of kept matches; *b (lane 0) is the record when that is not 0.
// `defer`: stop behind the neighbour list (S.top_s / S.top_d hold it): the Levenshtein distances and
// the first minimum are then k_lsh_lev's, a lane per kept match.
__device__ __forceinline__ int lsh_window(const CorpusDev& c, const LshDev& L, const GramIndexDev& g,
                                          const LshWaveLds& S, fs_status* st, fs_best* b, bool defer = false) {
  const int lane = threadIdx.x & 63;
  const int NW = (L.C + 63) >> 6;
  // keys.  Fast path as in k_lsh_scan: lane l holds projection columns 4l .. 4l+3, one
  // 16-byte load per lane and slot fetches the float32 row, all slots requested together;
  // the float32 sums decide the signs when every column is farther from zero than the
  // worst-case distance to the canonical float64 sum, else (and with an OOV token) the
  // window is redone in float64.
  bool keys_done = false;
  if (L.atab32 && L.C <= 256 && !(L.diag & 8)) {
    float am = 0.0f;
    int terms = 0;
    bool oov = false;
    if (lane < L.n) {
      const uint32_t id = S.f[lane];
      oov = (id & FS_OOV_FLAG) != 0 && (L.diag & 64);              // (diag 64: float64 for OOV windows as before round 5)
      if (!oov) row32_bound(L, lane, id, &am, &terms);
    }
#pragma unroll
    for (int d = 8; d > 0; d >>= 1) { am += __shfl_xor(am, d); terms += __shfl_xor(terms, d); }   // n <= 16 lanes hold a value
    am = __shfl(am, 0);
    terms = __shfl(terms, 0);
    if (!__any(oov)) {
      const float bnd = L.bound_scale * am * ((float)terms / (float)L.n);
      const int col = 4 * lane;
      const int left = L.C - col;
      const uint32_t cmask = left >= 4 ? 0xFu : left > 0 ? (1u << left) - 1 : 0u;
      const int colc = left > 0 ? col : 0;
      // (eight rows in flight at a time; summed in slot order like k_lsh_scan)
      float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      for (int k0 = 0; k0 < L.n; k0 += 8) {
        float4 r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k0 + k < L.n)
            r[k] = row32(L, k0 + k, S.f[k0 + k], colc);
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k0 + k < L.n) {
            if (k0 + k == 0) { acc = r[0]; continue; }
            acc.x = __fadd_rn(acc.x, r[k].x); acc.y = __fadd_rn(acc.y, r[k].y);
            acc.z = __fadd_rn(acc.z, r[k].z); acc.w = __fadd_rn(acc.w, r[k].w);
          }
      }
      const uint32_t sure = (fabsf(acc.x) > bnd ? 1u : 0u) | (fabsf(acc.y) > bnd ? 2u : 0u) |
                            (fabsf(acc.z) > bnd ? 4u : 0u) | (fabsf(acc.w) > bnd ? 8u : 0u);
      if (!__any((~sure & cmask) != 0u)) {
        uint32_t x = ((acc.x > 0.0f ? 1u : 0u) | (acc.y > 0.0f ? 2u : 0u) |
                      (acc.z > 0.0f ? 4u : 0u) | (acc.w > 0.0f ? 8u : 0u)) & cmask;
        // eight lanes -> one 32-bit piece of the column bit string
        x |= (uint32_t)__shfl_down((int)x, 1) << 4;
        x |= (uint32_t)__shfl_down((int)x, 2) << 8;
        x |= (uint32_t)__shfl_down((int)x, 4) << 16;
        uint32_t* pieces = reinterpret_cast<uint32_t*>(S.bal);
        if ((lane & 7) == 0 && (lane >> 3) < 2 * NW) pieces[lane >> 3] = x;
        keys_done = true;
      }
    }
  }
  for (int ch = 0; ch < NW && !keys_done; ++ch) {
    const int col = ch * 64 + lane;
    bool bit = false;
    if (col < L.C && !(L.diag & 8)) {
      double acc = a_value(L, 0, S.f[0], col);
      for (int k = 1; k < L.n; ++k) acc = __dadd_rn(acc, a_value(L, k, S.f[k], col));
      bit = acc > 0.0;
    }
    const uint64_t b = __ballot(bit);
    if (lane == 0) S.bal[ch] = b;
  }
  if (lane == 0) S.bal[NW] = 0;
  __builtin_amdgcn_wave_barrier();
  if (lane < L.H) S.key[lane] = assemble_key(S.bal, lane, L.B);
  __builtin_amdgcn_wave_barrier();
  int cnt;
  if (L.diag & 32) {
    cnt = 0;
  } else if (L.nn <= 48 && L.H <= 64 && !L.serial_neighbours) {
    cnt = lsh_neighbours_wave(L, S.key, S.f, S.top_s, S.top_d, S.pre, S.e0, S.qf);
  } else {                                    // NearestFilter(N > 48): one lane walks the buckets
    if (lane == 0)
      S.n[0] = lsh_neighbours<false>(L, S.key, S.f, S.top_s, S.top_d);
    __builtin_amdgcn_wave_barrier();
    cnt = S.n[0];
  }
  if (cnt == 0 || defer) return cnt;
  // Levenshtein of every kept match (search.py:189-190), the wave working on one
  // match at a time
  for (int r = 0; r < cnt; ++r) {
    uint32_t lv = FS_NONE;
    if (L.selflev) {
      // a match with the same id in every slot has the same strings as the script window's
      // own ids: its distance was computed once per string table (k_selflev)
      const uint32_t sr = S.top_s[r];
      const bool differs = lane < L.n && L.stok[sr + lane] != S.f[lane];
      if (!__any(differs)) lv = L.selflev[sr];
    }
    if (lv == FS_NONE)
      lv = (L.diag & 16) ? 1u :
                        lev_wave(g, S.top_s[r], S.fs, c.chars, c.coff, c.n_str, st,
                                 S.la, S.lb);
    if (lane == 0) S.lev[r] = lv;
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) {
    b->pad = 0.0;
    for (int r = 0; r < cnt; ++r) {           // first minimum of dist * lev in rank order
      const double comb = __dmul_rn(S.top_d[r], (double)S.lev[r]);
      if (r == 0 || comb < b->comb) {
        b->s = S.top_s[r]; b->lev = S.lev[r]; b->dist = S.top_d[r]; b->comb = comb;
      }
    }
  }
  return cnt;
}

#define FS_LSH_WAVE_LDS                                                                       \
  __shared__ uint64_t s_bal[4][32];                                                           \
  __shared__ uint32_t s_key[4][64];                                                           \
  __shared__ uint32_t s_top_s[4][64];                                                         \
  __shared__ double s_top_d[4][64];                                                           \
  __shared__ uint32_t s_lev[4][64];                                                           \
  __shared__ uint32_t s_la[4][FS_LEV_MAX + 2], s_lb[4][FS_LEV_MAX + 2];                       \
  __shared__ uint32_t s_f[4][FS_MAX_WINDOW];                                                  \
  __shared__ uint32_t s_fs[4][FS_MAX_WINDOW];                                                 \
  __shared__ int s_n[4];                                                                      \
  __shared__ uint32_t s_pre[4][64], s_e0[4][64];                                              \
  __shared__ double s_qf[4][FS_MAX_WINDOW];                                                   \
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));                   \
  const LshWaveLds S{s_bal[wave], s_key[wave], s_top_s[wave], s_top_d[wave], s_lev[wave],     \
                     s_la[wave], s_lb[wave], s_f[wave], s_fs[wave], &s_n[wave], s_pre[wave],   \
                     s_e0[wave], s_qf[wave]}

// Per script n-gram, once per string table (fs_corpus_update_end, like the exact path's
// ctab): what a fan window with the n-gram's ids and the strings of those ids gets.  The
// reference's result for a window is a function of its vector (i.e. of its ids) and, for the
// Levenshtein distances, of its strings, so every such window of a batch takes this record
// (k_lsh_sift) and no bucket is walked for it.  tab_cnt = kept matches + 1.  One wave per
// n-gram.
__global__ __launch_bounds__(256, 4) void k_lsh_gramtab(CorpusDev c, LshDev L, GramIndexDev g,
                                                        unsigned long long* __restrict__ tab_best,
                                                        uint32_t* __restrict__ tab_cnt, fs_status* st) {
  FS_LSH_WAVE_LDS;
  const int lane = threadIdx.x & 63;
  for (uint32_t gram = blockIdx.x * 4 + wave; gram < g.n_grams; gram += gridDim.x * 4) {
    const uint32_t first = g.gpos[(size_t)gram * g.nn];
    if (lane < L.n) {
      const uint32_t id = g.stok[first + lane];
      S.f[lane] = id;
      S.fs[lane] = id;
    }
    __builtin_amdgcn_wave_barrier();
    fs_best b;
    const int cnt = lsh_window(c, L, g, S, st, &b);
    if (lane == 0) {
      if (cnt) {
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&b);
#pragma unroll
        for (int k = 0; k < 4; ++k) tab_best[4 * (size_t)gram + k] = src[k];
      }
      tab_cnt[gram] = (uint32_t)cnt + 1u;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// (five waves per SIMD: 94 registers with two spilt; 97 at four.  A wave per window and about ten
// levels of dependent loads: the windows in flight are what counts -- n = 10: 0.268 -> 0.249 ms per
// C2 batch; six waves, 80 registers, eight spilt, measured slower again)
template <bool DEFER>
__global__ __launch_bounds__(256, 5) void k_lsh_verify(CorpusDev c, LshDev L, GramIndexDev g,
                                                    const uint32_t* __restrict__ cpos, NSrc nc,
                                                    uint32_t* __restrict__ cg,
                                                    uint32_t* __restrict__ cw,
                                                    fs_best* __restrict__ cbest,
                                                    uint32_t* __restrict__ bmatch, fs_status* st,
                                                    const uint32_t* __restrict__ pend,
                                                    uint32_t* __restrict__ mcnt,
                                                    uint32_t* __restrict__ mtop_s,
                                                    double* __restrict__ mtop_d) {
  FS_LSH_WAVE_LDS;
  __shared__ uint32_t s_w32[4];
  const int lane = threadIdx.x & 63;
  uint32_t matches = 0;
  // k_lsh_sift has been over every candidate: what it left pending is worked out here, a wave
  // per window, the windows of its list dealt round-robin over the waves.
  const uint32_t gw = blockIdx.x * 4 + wave, NWAVES = gridDim.x * 4;
  const uint32_t n_pend = min(st->lsh_pending, nc.cap);
  {
    for (uint32_t j = gw; j < n_pend; j += NWAVES) {
      const uint32_t i = pend[j];
      const uint64_t p = cpos[i];
      bool ok = p + L.n <= c.n_tok;
      uint32_t w = 0;
      if (ok) {
        uint64_t work_end;
        w = work_of_token(c, p, &work_end);
        ok = p + L.n <= work_end;                 // a window never crosses a work boundary
      }
      if (!ok) {                                  // wave-uniform
        if (lane == 0) { cg[i] = FS_NONE; if (DEFER) mcnt[j] = 0; }
        continue;
      }
      if (lane < L.n) {
        S.f[lane] = c.tok[p + lane];
        S.fs[lane] = c.str ? c.str[p + lane] : c.tok[p + lane];
      }
      __builtin_amdgcn_wave_barrier();
      fs_best b;
      const int cnt = lsh_window(c, L, g, S, st, &b, DEFER);
      if (DEFER) {
        // the kept matches in NearestFilter order for k_lsh_lev (a lane per match there)
        if (lane < cnt) {
          mtop_s[(size_t)j * L.nn + lane] = S.top_s[lane];
          mtop_d[(size_t)j * L.nn + lane] = S.top_d[lane];
        }
        if (lane == 0) {
          mcnt[j] = (uint32_t)cnt;
          cg[i] = cnt ? FS_PENDING : FS_NONE;
          cw[i] = w;
          matches += (uint32_t)cnt;
        }
      } else if (lane == 0) {
        if (cnt) {
          cbest[i] = b;
          cg[i] = 0;
          cw[i] = w;
          matches += (uint32_t)cnt;
        } else {
          cg[i] = FS_NONE;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  uint32_t tot;
  block_excl_scan(matches, s_w32, &tot);
  if (threadIdx.x == 0) bmatch[blockIdx.x] += tot;       // (on top of k_lsh_sift's)
}

// ---- the pending windows eight at a time (round 5) ----------------------------------------
// k_lsh_verify works a window off with a whole wave: keys, bucket ranges, bucket members,
// distances, NearestFilter -- about ten levels of dependent loads for one window, during most
// of which most lanes wait (71 % of the wave cycles waiting, r04_lsh_verify_pmc_sq_after.json),
// and on a table with near-synonyms a C2 batch leaves 280 k such windows: 0.85 of the search's
// 1.05 ms.  k_lsh_batch takes EIGHT windows of the pending list per wave and walks the same
// steps level by level for all of them at once:
//   A  ids, q and bounds of the eight windows, lane per (window, slot); float32 projection rows
//      summed in slot order, four columns a lane (float64 redo per window where a sign is not
//      certain, as lsh_window)
//   B  the 8 x H bucket ranges, lane per (window, table)
//   C  the bucket members of all eight windows numbered in (window, table, entry) order and
//      dealt to the lanes 64 at a time: script window, canonical distance (window_distance);
//      those within the threshold go to the window's list in LDS in arrival order
//   D  UniqueFilter and NearestFilter per window, eight lanes a window: an entry's rank in the
//      stable order (distance, then arrival) among the entries that are not repeats
// What it keeps equals lsh_neighbours_wave's list entry for entry (NearPy: UniqueFilter keeps a
// window's first arrival, NearestFilter is a stable sort).  A window with more than kBatchCap
// members within the threshold (crowded buckets) is worked off by lsh_window behind the batch.
// The Levenshtein distances are k_lsh_lev's (a lane per kept match).
constexpr int kBatchW = 8;                // windows per wave and step
constexpr int kBatchCap = 32;             // members within the threshold kept per window
constexpr int kBatchH = 16;               // tables (number_of_hashes) this form serves
struct alignas(16) BatchLds {             // per wave
  uint64_t bal[kBatchW][6];               // sign bits of the projection columns (C <= 256), + a zero word
  double qf[kBatchW][FS_MAX_WINDOW];      // q of the windows' slots
  double ff[kBatchW], rff[kBatchW];
  double vd[kBatchW][kBatchCap];          // members within the threshold, arrival order: distance ...
  uint32_t vs[kBatchW][kBatchCap];        // ... and script window (bit 31: a repeat)
  uint32_t f[kBatchW][FS_MAX_WINDOW];     // vector ids
  uint32_t key[kBatchW][kBatchH];
  uint32_t e0[kBatchW][kBatchH];          // first entry of the window's bucket in table h
  uint32_t pre[kBatchW][kBatchH];         // entries of the window in the tables before h
  uint32_t dh[128];                       // C: (window, script window) -> a lane of the round that holds the pair
  uint32_t jj[kBatchW];                   // the window's place in the pending list (mcnt / mtop)
  uint32_t wbase[kBatchW + 1];            // entries of the windows before w
  uint32_t vn[kBatchW];                   // members within the threshold so far (may exceed kBatchCap)
  uint32_t ci[kBatchW];                   // candidate number
  uint32_t work[kBatchW];
  uint32_t ok[kBatchW];                   // 1: a window of one work; 2: its keys need float64
  float bnd[kBatchW];
};
static_assert(sizeof(double) * kBatchW * kBatchCap + sizeof(uint32_t) * kBatchW * kBatchCap >= 2400,
              "lsh_window's scratch is laid over vd / vs");

template <int N>
__global__ __launch_bounds__(256, 5) void k_lsh_batch(CorpusDev c, LshDev L, GramIndexDev g,
                                                   const uint32_t* __restrict__ cpos, uint32_t cap,
                                                   uint32_t* __restrict__ cg, uint32_t* __restrict__ cw,
                                                   uint32_t* __restrict__ bmatch, fs_status* st,
                                                   const uint32_t* __restrict__ pend,
                                                   uint32_t* __restrict__ mcnt,
                                                   uint32_t* __restrict__ mtop_s,
                                                   double* __restrict__ mtop_d,
                                                   const uint32_t* __restrict__ left,
                                                   const uint32_t* __restrict__ n_left) {
  __shared__ BatchLds s_b[4];
  __shared__ uint32_t s_w32[4];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  BatchLds& S = s_b[wave];
  uint32_t matches = 0;
  const uint32_t gw = blockIdx.x * 4 + wave, NWAVES = gridDim.x * 4;
  // (left: the windows k_lsh_enum could not finish, as places in the pending list)
  const uint32_t n_pend = left ? min(*n_left, cap) : min(st->lsh_pending, cap);
  const uint32_t nn = (uint32_t)L.nn;
  const uint32_t nb1 = (1u << L.B) + 1;
  auto sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // (few windows: fewer per wave, so that every wave has some -- a wave's eight windows take
  // a hundred microseconds when each walks sixty bucket members)
  const uint32_t per = max(1u, min((uint32_t)kBatchW, (n_pend + NWAVES - 1) / NWAVES));
  if ((L.diag & 0x100000) && left && blockIdx.x == 0 && threadIdx.x == 0) atomicMax(&st->max_rows, n_pend);   // diagnostics: windows left to the walk
  for (uint32_t j0 = gw * per; j0 < n_pend; j0 += NWAVES * per) {
    const uint32_t nw = min(per, n_pend - j0);
    // ---- A: the windows ------------------------------------------------------------------
    if (lane < kBatchW) {
      uint32_t ok = 0, w = 0, i = 0;
      uint64_t p = 0;
      uint32_t jx = 0;
      if ((uint32_t)lane < nw) {
        jx = left ? left[j0 + lane] : j0 + lane;
        i = pend[jx];
        p = cpos[i];
        if (p + N <= c.n_tok) {
          uint64_t work_end;
          w = work_of_token(c, p, &work_end);
          ok = p + N <= work_end ? 1u : 0u;       // a window never crosses a work boundary
        }
      }
      S.ci[lane] = i; S.work[lane] = w; S.ok[lane] = ok; S.vn[lane] = 0; S.jj[lane] = jx;
      S.wbase[lane] = (uint32_t)p;                // (the position, until B overwrites it)
      S.bal[lane][(L.C + 63) >> 6] = 0;
    }
    sync();
#pragma unroll
    for (int t = 0; t < kBatchW * FS_MAX_WINDOW / 64; ++t) {      // lane per (window, slot)
      const int w = (t * 64 + lane) / FS_MAX_WINDOW, k = (t * 64 + lane) % FS_MAX_WINDOW;
      float am = 0.0f;
      int terms = 0;
      uint32_t oov = 0;
      if (k < N && S.ok[w]) {
        const uint32_t id = c.tok[(uint64_t)S.wbase[w] + k];
        S.f[w][k] = id;
        S.qf[w][k] = q_of(L, id);
        oov = (L.diag & 64) ? id & FS_OOV_FLAG : 0u;                // (diag 64: float64 for OOV windows as before round 5)
        if (!oov && L.atab32) row32_bound(L, k, id, &am, &terms);
      }
      // sum / or over the window's FS_MAX_WINDOW lanes
      static_assert(FS_MAX_WINDOW == 16, "a window's slots are one row of sixteen lanes");
#pragma unroll
      for (int d = 8; d > 0; d >>= 1) { am += __shfl_xor(am, d); terms += __shfl_xor(terms, d); oov |= (uint32_t)__shfl_xor((int)oov, d); }
      if (k == 0 && S.ok[w]) {
        S.bnd[w] = L.bound_scale * am * ((float)terms / (float)N);
        if (oov || !L.atab32 || L.C > 256) S.ok[w] = 2u;
      }
    }
    sync();
    if (lane < kBatchW && S.ok[lane]) {
      double ff = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) ff = __dadd_rn(ff, S.qf[lane][k]);
      S.ff[lane] = ff;
      S.rff[lane] = __dsqrt_rn(ff);
    }
    // keys, float32: lane l holds projection columns 4l .. 4l+3; the N rows of a window
    // requested together, summed in slot order (as lsh_window)
    {
      const int col = 4 * lane;
      const int left = L.C - col;
      const uint32_t cmask = left >= 4 ? 0xFu : left > 0 ? (1u << left) - 1 : 0u;
      const int colc = left > 0 ? col : 0;
      // (the rows of two windows in flight together where the registers allow: n <= 8)
      constexpr int PAIR = N <= 8 ? 2 : 1;
      auto finish = [&](uint32_t w, float4 acc) {
        const float bnd = S.bnd[w];
        const uint32_t sure = (fabsf(acc.x) > bnd ? 1u : 0u) | (fabsf(acc.y) > bnd ? 2u : 0u) |
                              (fabsf(acc.z) > bnd ? 4u : 0u) | (fabsf(acc.w) > bnd ? 8u : 0u);
        if (__any((~sure & cmask) != 0u)) {
          if (lane == 0) S.ok[w] = 2u;            // a sign is not certain: float64 below
          return;
        }
        uint32_t x = ((acc.x > 0.0f ? 1u : 0u) | (acc.y > 0.0f ? 2u : 0u) |
                      (acc.z > 0.0f ? 4u : 0u) | (acc.w > 0.0f ? 8u : 0u)) & cmask;
        // eight lanes -> one 32-bit piece of the column bit string
        x |= (uint32_t)__shfl_down((int)x, 1) << 4;
        x |= (uint32_t)__shfl_down((int)x, 2) << 8;
        x |= (uint32_t)__shfl_down((int)x, 4) << 16;
        uint32_t* pieces = reinterpret_cast<uint32_t*>(S.bal[w]);
        if ((lane & 7) == 0) pieces[lane >> 3] = x;
      };
      for (uint32_t w0 = 0; w0 < nw; w0 += PAIR) {
        float4 r[PAIR][N];
        bool go[PAIR];
#pragma unroll
        for (int u = 0; u < PAIR; ++u) {
          go[u] = w0 + u < nw && S.ok[w0 + u] == 1u;             // (wave-uniform)
          if (go[u]) {
#pragma unroll
            for (int k = 0; k < N; ++k)
              r[u][k] = row32(L, k, S.f[w0 + u][k], colc);
          }
        }
#pragma unroll
        for (int u = 0; u < PAIR; ++u)
          if (go[u]) {
            float4 acc = r[u][0];
#pragma unroll
            for (int k = 1; k < N; ++k) {
              acc.x = __fadd_rn(acc.x, r[u][k].x); acc.y = __fadd_rn(acc.y, r[u][k].y);
              acc.z = __fadd_rn(acc.z, r[u][k].z); acc.w = __fadd_rn(acc.w, r[u][k].w);
            }
            finish(w0 + u, acc);
          }
      }
    }
    sync();
    for (uint32_t w = 0; w < nw; ++w) {           // float64 keys where needed (rare)
      if (S.ok[w] != 2u) continue;
      for (int ch = 0; ch < (L.C + 63) >> 6; ++ch) {
        const int col = ch * 64 + lane;
        bool bit = false;
        if (col < L.C) {
          double acc = a_value(L, 0, S.f[w][0], col);
          for (int k = 1; k < N; ++k) acc = __dadd_rn(acc, a_value(L, k, S.f[w][k], col));
          bit = acc > 0.0;
        }
        const uint64_t b = __ballot(bit);
        if (lane == 0) S.bal[w][ch] = b;
      }
    }
    sync();
    if (L.diag & 0x10000) {                       // diagnostics: the keys only
      if (lane < kBatchW && (uint32_t)lane < nw) { mcnt[S.jj[lane]] = 0; cg[S.ci[lane]] = FS_NONE; }
      continue;
    }
    // the keys
#pragma unroll
    for (int t = 0; t < kBatchW * kBatchH / 64; ++t) {
      const int w = (t * 64 + lane) / kBatchH, h = (t * 64 + lane) % kBatchH;
      if (h < L.H && S.ok[w]) S.key[w][h] = assemble_key(S.bal[w], h, L.B);
    }
    sync();
    // ---- B: bucket ranges, lane per (window, table) ----------------------------------------
#pragma unroll
    for (int t = 0; t < kBatchW * kBatchH / 64; ++t) {
      const int w = (t * 64 + lane) / kBatchH, h = (t * 64 + lane) % kBatchH;
      uint32_t e0 = 0, cnt_h = 0;
      if (h < L.H && S.ok[w]) {
        const uint32_t key = S.key[w][h];
        const uint32_t* o = L.boff + (size_t)h * nb1 + key;
        e0 = o[0];
        cnt_h = o[1] - e0;
      }
      // exclusive prefix inside the window's row of sixteen lanes
      uint32_t incl = cnt_h;
#pragma unroll
      for (int d = 1; d < kBatchH; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
        if (h >= d) incl += up;
      }
      S.e0[w][h] = e0;
      S.pre[w][h] = incl - cnt_h;
      if (h == kBatchH - 1) S.vn[w] = incl;       // (the window's entries, until C needs vn)
    }
    sync();
    if (lane == 0) {
      uint32_t run = 0;
#pragma unroll
      for (int w = 0; w < kBatchW; ++w) { S.wbase[w] = run; run += S.vn[w]; S.vn[w] = 0; }
      S.wbase[kBatchW] = run;
    }
    sync();
    // ---- C: the members, 64 at a time -------------------------------------------------------
    const uint32_t total = (L.diag & 0x20000) ? 0u : S.wbase[kBatchW];      // (diagnostics: no members)
    if (L.diag & 0x80000) { if (lane == 0) atomicAdd(&st->max_rows, S.wbase[kBatchW]); }   // diagnostics: count them (printed by fs_search_corpus_end)
    for (uint32_t r0 = 0; r0 < total; r0 += 64) {
      const uint32_t j = r0 + lane;
      bool valid = false;
      uint32_t w = 0, s = 0;
      double d = 0.0;
      if (j < total) {
#pragma unroll
        for (int i = 1; i < kBatchW; ++i) w += S.wbase[i] <= j ? 1u : 0u;
        const uint32_t jw = j - S.wbase[w];
        uint32_t h = 0;                           // last table with pre[w][h] <= jw
#pragma unroll
        for (uint32_t step = kBatchH / 2; step > 0; step >>= 1)
          if (h + step < (uint32_t)L.H && S.pre[w][h + step] <= jw) h += step;
        s = L.bids[(size_t)h * L.W + S.e0[w][h] + (jw - S.pre[w][h])];
      }
      // A script window comes back once per table whose bucket it shares with the fan window --
      // a dozen times for a real neighbour -- and its distance is the same every time: one lane
      // of the round works it out for all that hold the same (window, script window) pair.  The
      // lanes agree on it through a small table in LDS (a lane that finds another pair's lane
      // in its slot works its own out).
      const uint32_t pair = s * (uint32_t)kBatchW + w;
      const uint32_t hslot = (pair * 0x9E3779B1u) >> 25;
      if (j < total) S.dh[hslot] = (uint32_t)lane;
      sync();
      const int leader = j < total ? (int)S.dh[hslot] : lane;
      const bool follow = (uint32_t)__shfl((int)pair, leader) == pair && leader != lane && j < total;
      if (j < total && !follow && !(L.diag & 0x40000))                  // (diagnostics: no distances)
        valid = window_distance_flat<N>(L, s, S.f[w], S.qf[w], S.ff[w], S.rff[w], &d) && d < L.thr;
      {
        const long long db = __double_as_longlong(d);
        const int lo = __shfl((int)(uint32_t)db, leader), hi = __shfl((int)(uint32_t)(db >> 32), leader);
        const int lv = __shfl((int)valid, leader);
        if (follow) { d = __longlong_as_double(((long long)hi << 32) | (uint32_t)lo); valid = lv != 0; }
      }
      // to the window's list, arrival order: the lanes of one window are consecutive
      const uint64_t vm = __ballot(valid);
      if (vm) {
        const uint32_t first = S.wbase[w] > r0 ? S.wbase[w] - r0 : 0u;        // the window's first lane of this round
        const uint64_t below = ((1ull << lane) - 1) & ~((1ull << first) - 1);
        const uint32_t slot = S.vn[w] + (uint32_t)__popcll(vm & below);
        if (valid && slot < (uint32_t)kBatchCap) { S.vs[w][slot] = s; S.vd[w][slot] = d; }
        sync();
        // the counts: the window's last valid lane of the round adds the round's number
        const uint32_t last_lane = min(63u, S.wbase[w + 1] - 1 - r0);
        const uint64_t mine = vm & (~0ull >> (63 - last_lane)) & ~((1ull << first) - 1);
        if (valid && (mine >> lane) == 1ull) S.vn[w] = slot + 1;
      }
      sync();
    }
    // ---- D: UniqueFilter, NearestFilter: eight lanes a window -----------------------------
    {
      const int w = lane >> 3, t = lane & 7;
      const uint32_t V = min(S.vn[w], (uint32_t)kBatchCap);
      const bool over = S.vn[w] > (uint32_t)kBatchCap;
      if (L.unique && !over) {
        for (uint32_t e = t; e < V; e += 8) {
          const uint32_t se = S.vs[w][e] & 0x7FFFFFFFu;
          bool dup = false;
          for (uint32_t x = 0; x < e; ++x) dup = dup || (S.vs[w][x] & 0x7FFFFFFFu) == se;
          if (dup) S.vs[w][e] = se | 0x80000000u;
        }
      }
      sync();
      uint32_t kept = 0;
      if (!over && (uint32_t)w < nw) {
        const uint32_t jj = S.jj[w];
        for (uint32_t e = t; e < V; e += 8) {
          const uint32_t se = S.vs[w][e];
          if (se & 0x80000000u) continue;
          ++kept;
          const double de = S.vd[w][e];
          uint32_t rank = 0;
          for (uint32_t x = 0; x < V; ++x) {
            const double dx = S.vd[w][x];
            rank += (!(S.vs[w][x] & 0x80000000u) && (dx < de || (dx == de && x < e))) ? 1u : 0u;
          }
          if (rank < nn) { mtop_s[(size_t)jj * nn + rank] = se; mtop_d[(size_t)jj * nn + rank] = de; }
        }
      }
      kept += (uint32_t)__shfl_xor((int)kept, 1);
      kept += (uint32_t)__shfl_xor((int)kept, 2);
      kept += (uint32_t)__shfl_xor((int)kept, 4);
      kept = min(kept, nn);
      if (t == 0 && (uint32_t)w < nw && !over) {
        const uint32_t i = S.ci[w];
        mcnt[S.jj[w]] = kept;
        cg[i] = kept ? FS_PENDING : FS_NONE;
        cw[i] = S.work[w];
        matches += kept;
      }
    }
    sync();
    // the crowded windows, a wave each (lsh_window with its scratch laid over the lists)
    for (uint32_t w = 0; w < nw; ++w) {
      if (S.vn[w] <= (uint32_t)kBatchCap) continue;             // (wave-uniform)
      uint8_t* raw = reinterpret_cast<uint8_t*>(&S.vd[0][0]);
      LshWaveLds X;
      X.bal = reinterpret_cast<uint64_t*>(raw);                  // 32 x 8
      X.top_d = reinterpret_cast<double*>(raw + 256);            // 64 x 8
      X.qf = reinterpret_cast<double*>(raw + 768);               // 16 x 8
      X.key = reinterpret_cast<uint32_t*>(raw + 896);            // 64 x 4
      X.top_s = X.key + 64; X.pre = X.key + 128; X.e0 = X.key + 192;
      X.f = X.key + 256; X.fs = X.key + 272;                     // 16 + 16
      X.n = reinterpret_cast<int*>(X.key + 288);
      X.lev = nullptr; X.la = nullptr; X.lb = nullptr;           // (deferred: no Levenshtein here)
      static_assert(896 + 4 * 292 <= sizeof(double) * kBatchW * kBatchCap + sizeof(uint32_t) * kBatchW * kBatchCap, "scratch");
      const uint32_t idv = lane < N ? S.f[w][lane] : 0u;
      const uint32_t i = S.ci[w], wk = S.work[w];
      sync();
      if (lane < N) { X.f[lane] = idv; X.fs[lane] = idv; }
      sync();
      fs_best b;
      const int cnt = lsh_window(c, L, g, X, st, &b, true);
      const uint32_t jj = S.jj[w];
      if (lane < cnt) {
        mtop_s[(size_t)jj * nn + lane] = X.top_s[lane];
        mtop_d[(size_t)jj * nn + lane] = X.top_d[lane];
      }
      if (lane == 0) {
        mcnt[jj] = (uint32_t)cnt;
        cg[i] = cnt ? FS_PENDING : FS_NONE;
        cw[i] = wk;
        matches += (uint32_t)cnt;
      }
      sync();
    }
  }
  uint32_t tot;
  block_excl_scan(matches, s_w32, &tot);
  if (threadIdx.x == 0) bmatch[blockIdx.x] += tot;       // (on top of k_lsh_sift's)
}

// ---- the pending windows without a bucket walk (round 5) --------------------------------------
// Every script window within the threshold of a pending window equals it in all slots but one --
// by vector ids where the table's c_max proves it (m_min = n - 1), by component ids on tables
// with near-synonyms -- so its n-gram sits in an exact map under one of the window's n one-slot-
// wildcard keys (build_emap).  What LSH finds of such an n-gram is decided by the keys alone:
// the tables in which its key is the window's (the script windows' keys are kept, d_skeys); all
// its occurrences share those buckets in ascending order; and everything beyond the threshold
// comes behind everything within it in NearestFilter's stable order, so it never changes the
// list.  A C2 batch on the clustered table walks 67 bucket members per pending window -- 18.9 M
// distances, nearly all of unrelated windows -- for lists that hold one script n-gram.
//   k_lsh_pkeys   the windows' 15 keys, eight windows per wave (k_lsh_batch's first stage)
//   k_lsh_enum    a LANE per window: the n map lookups, the n-gram's occurrences, keys and
//                 canonical distance, the list written in arrival order (table, occurrence)
// Windows with more than one such n-gram, a long chain in the map, or nn > 10 go to k_lsh_batch's
// bucket walk through a list of their own (a percent or two).
constexpr int kEnumNN = 10;               // NearestFilter sizes k_lsh_enum serves
constexpr int kEnumG = 4;                 // script n-grams one slot away from a window that it takes
struct alignas(16) PkeysLds {             // per wave
  uint64_t bal[kBatchW][6];
  uint32_t f[kBatchW][FS_MAX_WINDOW];
  uint32_t pos[kBatchW], ok[kBatchW], work[kBatchW];
  float bnd[kBatchW];
};

template <int N>
__global__ __launch_bounds__(256) void k_lsh_pkeys(CorpusDev c, LshDev L, const uint32_t* __restrict__ cpos,
                                                   uint32_t cap, const uint32_t* __restrict__ pend,
                                                   const fs_status* st, uint32_t* __restrict__ pkeys,
                                                   uint32_t* __restrict__ pwork) {
  __shared__ PkeysLds s_b[4];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  PkeysLds& S = s_b[wave];
  const uint32_t gw = blockIdx.x * 4 + wave, NWAVES = gridDim.x * 4;
  const uint32_t n_pend = min(st->lsh_pending, cap);
  auto sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  const uint32_t per = max(1u, min((uint32_t)kBatchW, (n_pend + NWAVES - 1) / NWAVES));
  for (uint32_t j0 = gw * per; j0 < n_pend; j0 += NWAVES * per) {
    const uint32_t nw = min(per, n_pend - j0);
    if (lane < kBatchW) {
      uint32_t ok = 0, w = FS_NONE;
      uint64_t p = 0;
      if ((uint32_t)lane < nw) {
        p = cpos[pend[j0 + lane]];
        if (p + N <= c.n_tok) {
          uint64_t work_end;
          const uint32_t wk = work_of_token(c, p, &work_end);
          if (p + N <= work_end) { ok = 1u; w = wk; }           // a window never crosses a work boundary
        }
      }
      S.pos[lane] = (uint32_t)p; S.ok[lane] = ok; S.work[lane] = w;
      S.bal[lane][(L.C + 63) >> 6] = 0;
    }
    sync();
#pragma unroll
    for (int t = 0; t < kBatchW * FS_MAX_WINDOW / 64; ++t) {      // lane per (window, slot)
      const int w = (t * 64 + lane) / FS_MAX_WINDOW, k = (t * 64 + lane) % FS_MAX_WINDOW;
      float am = 0.0f;
      uint32_t oov = 0;
      if (k < N && S.ok[w]) {
        const uint32_t id = c.tok[(uint64_t)S.pos[w] + k];
        S.f[w][k] = id;
        oov = id & FS_OOV_FLAG;
        if (!oov && L.atab32) am = L.amax[(size_t)k * L.V + id];
      }
#pragma unroll
      for (int d = 8; d > 0; d >>= 1) { am += __shfl_xor(am, d); oov |= (uint32_t)__shfl_xor((int)oov, d); }
      if (k == 0 && S.ok[w]) {
        S.bnd[w] = L.bound_scale * am;
        if (oov || !L.atab32 || L.C > 256) S.ok[w] = 2u;
      }
    }
    sync();
    {
      const int col = 4 * lane;
      const int left = L.C - col;
      const uint32_t cmask = left >= 4 ? 0xFu : left > 0 ? (1u << left) - 1 : 0u;
      const int colc = left > 0 ? col : 0;
      for (uint32_t w = 0; w < nw; ++w) {
        if (S.ok[w] != 1u) continue;              // (wave-uniform)
        float4 r[N];
#pragma unroll
        for (int k = 0; k < N; ++k)
          r[k] = *reinterpret_cast<const float4*>(L.atab32 + ((size_t)k * L.V + S.f[w][k]) * L.Cp + colc);
        float4 acc = r[0];
#pragma unroll
        for (int k = 1; k < N; ++k) {
          acc.x = __fadd_rn(acc.x, r[k].x); acc.y = __fadd_rn(acc.y, r[k].y);
          acc.z = __fadd_rn(acc.z, r[k].z); acc.w = __fadd_rn(acc.w, r[k].w);
        }
        const float bnd = S.bnd[w];
        const uint32_t sure = (fabsf(acc.x) > bnd ? 1u : 0u) | (fabsf(acc.y) > bnd ? 2u : 0u) |
                              (fabsf(acc.z) > bnd ? 4u : 0u) | (fabsf(acc.w) > bnd ? 8u : 0u);
        if (__any((~sure & cmask) != 0u)) {
          if (lane == 0) S.ok[w] = 2u;            // a sign is not certain: float64 below
          continue;
        }
        uint32_t x = ((acc.x > 0.0f ? 1u : 0u) | (acc.y > 0.0f ? 2u : 0u) |
                      (acc.z > 0.0f ? 4u : 0u) | (acc.w > 0.0f ? 8u : 0u)) & cmask;
        x |= (uint32_t)__shfl_down((int)x, 1) << 4;
        x |= (uint32_t)__shfl_down((int)x, 2) << 8;
        x |= (uint32_t)__shfl_down((int)x, 4) << 16;
        uint32_t* pieces = reinterpret_cast<uint32_t*>(S.bal[w]);
        if ((lane & 7) == 0) pieces[lane >> 3] = x;
      }
    }
    sync();
    for (uint32_t w = 0; w < nw; ++w) {           // float64 keys where needed (rare)
      if (S.ok[w] != 2u) continue;
      for (int ch = 0; ch < (L.C + 63) >> 6; ++ch) {
        const int col = ch * 64 + lane;
        bool bit = false;
        if (col < L.C) {
          double acc = a_value(L, 0, S.f[w][0], col);
          for (int k = 1; k < N; ++k) acc = __dadd_rn(acc, a_value(L, k, S.f[w][k], col));
          bit = acc > 0.0;
        }
        const uint64_t b = __ballot(bit);
        if (lane == 0) S.bal[w][ch] = b;
      }
    }
    sync();
#pragma unroll
    for (int t = 0; t < kBatchW * kBatchH / 64; ++t) {
      const int w = (t * 64 + lane) / kBatchH, h = (t * 64 + lane) % kBatchH;
      if ((uint32_t)w < nw) pkeys[(size_t)(j0 + w) * kBatchH + h] = (h < L.H && S.ok[w]) ? assemble_key(S.bal[w], h, L.B) : 0u;
    }
    if (lane < kBatchW && (uint32_t)lane < nw) pwork[j0 + lane] = S.work[lane];
    sync();
  }
}

template <int N>
__global__ __launch_bounds__(256, 4) void k_lsh_enum(CorpusDev c, LshDev L, GramIndexDev g,
                                                  const uint32_t* __restrict__ cpos, uint32_t cap,
                                                  uint32_t* __restrict__ cg, uint32_t* __restrict__ cw,
                                                  uint32_t* __restrict__ bmatch, fs_status* st,
                                                  const uint32_t* __restrict__ pend,
                                                  const uint32_t* __restrict__ pkeys,
                                                  const uint32_t* __restrict__ pwork,
                                                  uint32_t* __restrict__ mcnt, uint32_t* __restrict__ mtop_s,
                                                  double* __restrict__ mtop_d,
                                                  uint32_t* __restrict__ left, uint32_t* __restrict__ n_left) {
  __shared__ uint32_t s_w32[4];
  const int lane = threadIdx.x & 63;
  const uint32_t n_pend = min(st->lsh_pending, cap);
  const uint32_t nn = (uint32_t)L.nn;
  uint32_t matches = 0;
  for (uint32_t j0 = blockIdx.x * 256; j0 < n_pend; j0 += gridDim.x * 256) {
    const uint32_t j = j0 + threadIdx.x;
    const bool live = j < n_pend;
    bool give_up = false;
    uint32_t work = FS_NONE, i = 0;
    uint64_t p = 0;
    if (live) { i = pend[j]; work = pwork[j]; p = cpos[i]; }
    const bool ok = live && work != FS_NONE;
    // ids (the buffers are padded: reading up to 12 is in bounds), the keys' ids, q
    uint32_t f[N], fc[N];
    double qf[N];
    {
      const uint4* a = reinterpret_cast<const uint4*>(c.tok + (ok ? p : 0));
      const uint4* b = reinterpret_cast<const uint4*>((L.emap_comp ? L.wild_tok : c.tok) + (ok ? p : 0));
#pragma unroll
      for (int q4 = 0; q4 < (N + 3) / 4; ++q4) {
        const uint4 x = a[q4], y = b[q4];
        const uint32_t xv[4] = {x.x, x.y, x.z, x.w}, yv[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * q4 + e < N) { f[4 * q4 + e] = xv[e]; fc[4 * q4 + e] = yv[e]; }
      }
    }
    uint32_t term[N], fold = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      term[k] = fs_rotl(fs_premix(fc[k]), fs_rot_of(N - 1 - k));
      fold ^= term[k];
      qf[k] = ok ? L.q[f[k]] : 0.0;             // (no OOV id on this path: the prefilters exclude them)
    }
    // the n map lookups, four requested together; the n-grams they name (kEnumG at most; a full
    // bucket's chain is followed twice)
    uint32_t gl[kEnumG];
    uint32_t gn = 0;
#pragma unroll
    for (int x = 0; x < kEnumG; ++x) gl[x] = FS_NONE;
    auto add_gram = [&](uint32_t gid) {
      bool seen = false;
#pragma unroll
      for (int x = 0; x < kEnumG; ++x) seen = seen || gl[x] == gid;
      if (seen) return;
      if (gn >= (uint32_t)kEnumG) { give_up = true; return; }
#pragma unroll
      for (int x = 0; x < kEnumG; ++x) gl[x] = gn == (uint32_t)x ? gid : gl[x];
      ++gn;
    };
#pragma unroll
    for (int k0 = 0; k0 < N; k0 += 4) {
      uint4 ba[4], bb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (k0 + u < N) {
          const uint32_t h = fs_wild_key(fold, term[k0 + u], k0 + u);
          const uint4* bp = reinterpret_cast<const uint4*>(L.emap + 4 * (size_t)fs_wmap_slot(h, L.log2_emap));
          ba[u] = bp[0]; bb[u] = bp[1];
        }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (k0 + u < N) {
          const uint32_t h = fs_wild_key(fold, term[k0 + u], k0 + u);
          uint4 a = ba[u], b = bb[u];
          uint32_t bkt = fs_wmap_slot(h, L.log2_emap);
          for (int probe = 0;; ++probe) {
            const uint32_t key[4] = {a.x, a.z, b.x, b.z}, val[4] = {a.y, a.w, b.y, b.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (val[e] && key[e] == h && ok) add_gram(val[e] - 1);
            if (!val[3] || !ok) break;            // (not full: nothing has spilt past it)
            if (probe == 2) { give_up = true; break; }
            bkt = (bkt + 1) & ((1u << L.log2_emap) - 1u);
            const uint4* bp = reinterpret_cast<const uint4*>(L.emap + 4 * (size_t)bkt);
            a = bp[0]; b = bp[1];
          }
        }
    }
    if (!ok) give_up = false;
    // per n-gram: the canonical distance to its first window (window_distance_flat's arithmetic,
    // the fan side in registers) and the tables that hold both in one bucket
    double gd[kEnumG];
    uint32_t gt[kEnumG];
#pragma unroll
    for (int x = 0; x < kEnumG; ++x) { gd[x] = 0.0; gt[x] = 0; }
    if (ok && !give_up && gn) {
      double ff = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) ff = __dadd_rn(ff, qf[k]);
      const double rff = __dsqrt_rn(ff);
      uint32_t mine[kBatchH];
#pragma unroll
      for (int q4 = 0; q4 < kBatchH / 4; ++q4) {
        const uint4 m = reinterpret_cast<const uint4*>(pkeys + (size_t)j * kBatchH)[q4];
        mine[4 * q4] = m.x; mine[4 * q4 + 1] = m.y; mine[4 * q4 + 2] = m.z; mine[4 * q4 + 3] = m.w;
      }
      for (uint32_t gi = 0; gi < gn; ++gi) {
        uint32_t gid = gl[0];
#pragma unroll
        for (int x = 1; x < kEnumG; ++x) gid = gi == (uint32_t)x ? gl[x] : gid;
        if (gid >= g.n_grams) continue;           // (cannot happen: the map holds gram ids)
        const uint32_t s0 = g.gpos[(size_t)gid * nn];
        uint32_t tables = 0;
#pragma unroll
        for (int h = 0; h < kBatchH; ++h)
          tables |= (h < L.H && L.skeys[(size_t)s0 * L.H + h] == mine[h]) ? 1u << h : 0u;
        const fs_swin sw = L.sw[s0];
        uint4 rec[N];
#pragma unroll
        for (int k = 0; k < N; ++k) rec[k] = *reinterpret_cast<const uint4*>(L.spos + s0 + k);
        int same = 0;
        uint32_t diff = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) {
          const bool eq = rec[k].w == f[k];
          same += eq;
          diff |= eq ? 0u : 1u << k;
        }
        double d = 0.0;
        bool v = !(L.m_min > 0 && same < L.m_min);
        if (v) {
          const double norm = __dmul_rn(sw.rss, rff);
          double sf;
          if (same == N) {
            sf = sw.ss;
          } else {
            double gk[N];
#pragma unroll
            for (int k = 0; k < N; ++k) {
              gk[k] = qf[k];
              if (diff >> k & 1u) gk[k] = L.gtab[(size_t)(int32_t)rec[k].z * L.V + f[k]];
            }
            sf = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) sf = __dadd_rn(sf, gk[k]);
          }
          d = __dsub_rn(1.0, __ddiv_rn(sf, norm));
          v = d == d && d < L.thr;
        }
        if (!v) tables = 0;
        if (L.unique && tables) tables &= 0u - tables;           // a script window counts where it arrives first
#pragma unroll
        for (int x = 0; x < kEnumG; ++x)
          if (gi == (uint32_t)x) { gd[x] = d; gt[x] = tables; }
      }
    }
    // NearestFilter's order: by distance, the n-grams one after the other (entries of one n-gram
    // share its distance and are in arrival order).  Two n-grams at exactly the same distance
    // would interleave by arrival: left to the bucket walk.
    auto cswap = [&](int a, int b) {
      const bool sw2 = (gt[b] != 0 && (gt[a] == 0 || gd[b] < gd[a]));
      const double da = gd[a], db = gd[b];
      const uint32_t ta = gt[a], tb2 = gt[b], ga = gl[a], gb = gl[b];
      gd[a] = sw2 ? db : da; gd[b] = sw2 ? da : db;
      gt[a] = sw2 ? tb2 : ta; gt[b] = sw2 ? ta : tb2;
      gl[a] = sw2 ? gb : ga; gl[b] = sw2 ? ga : gb;
    };
    static_assert(kEnumG == 4, "a sorting network of four");
    cswap(0, 1); cswap(2, 3); cswap(0, 2); cswap(1, 3); cswap(1, 2);
#pragma unroll
    for (int x = 0; x + 1 < kEnumG; ++x)
      if (gt[x] && gt[x + 1] && gd[x] == gd[x + 1]) give_up = true;
    uint32_t made = 0;
    if (ok && !give_up) {
      const size_t jj = (size_t)j * nn;
#pragma unroll
      for (int x = 0; x < kEnumG; ++x) {
        if (!gt[x] || made >= nn) continue;
        const uint32_t gid = gl[x], occ = g.gcnt[gid];
        uint32_t o[kEnumNN];
#pragma unroll
        for (int r = 0; r < kEnumNN; ++r) o[r] = (uint32_t)r < nn ? g.gpos[(size_t)gid * nn + r] : 0u;
        uint32_t tb = gt[x], r = 0;
#pragma unroll
        for (int e = 0; e < kEnumNN; ++e)
          if (tb && made < nn) {
            uint32_t sel = o[0];
#pragma unroll
            for (int y = 1; y < kEnumNN; ++y) sel = r == (uint32_t)y ? o[y] : sel;
            mtop_s[jj + made] = sel;
            mtop_d[jj + made] = gd[x];
            ++made;
            if (++r == occ) { r = 0; tb &= tb - 1; }
          }
      }
    }
    if (live && !give_up) {
      mcnt[j] = made;
      cg[i] = made ? FS_PENDING : FS_NONE;
      if (ok) cw[i] = work;
      matches += made;
    }
    // the windows left to the bucket walk, one addition per wave
    const uint64_t gm = __ballot(live && give_up);
    if (gm) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(n_left, (uint32_t)__popcll(gm));
      base = (uint32_t)__builtin_amdgcn_readlane((int)base, 0);
      if (live && give_up)
        left[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(gm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)gm, 0u))] = j;
    }
  }
  uint32_t tot;
  block_excl_scan(matches, s_w32, &tot);
  if (threadIdx.x == 0) bmatch[blockIdx.x] += tot;       // (on top of k_lsh_sift's)
}

// The Levenshtein distances of the matches k_lsh_verify<true> kept, and the record of every
// pending window: a lane per (window, rank) pair -- lev_lane, Myers' recurrence on one lane's
// registers against the script window's bit planes, where k_lsh_verify ran it as a wave per
// pair on the scalar unit (a ballot per column and ~20 scalar instructions: the CU's one
// scalar unit was busy more than half of that kernel's time).  A workgroup takes 128 windows at
// a time (a pair is ~4 us of one lane's time whatever else runs: the fewer rounds of pairs a
// workgroup has to make, the better -- 128 windows are one round of its 256 lanes unless they
// keep two matches each): prefix sum of their match counts, the pairs dealt out evenly over
// the threads (the window of a pair by binary search in LDS), the distances through LDS, then a
// lane per window takes the first minimum of dist * lev in rank order (search.py:224-225).
constexpr int kLevMaxN = 16;            // NearestFilter sizes this form serves (the default is 10)
constexpr uint32_t kLevWin = 128;       // windows per workgroup and step
__global__ __launch_bounds__(256) void k_lsh_lev(CorpusDev c, LshDev L, GramIndexDev g, StrFast F,
                                                 const uint32_t* __restrict__ cpos, uint32_t cap,
                                                 const uint32_t* __restrict__ pend,
                                                 const uint32_t* __restrict__ mcnt,
                                                 const uint32_t* __restrict__ mtop_s,
                                                 const double* __restrict__ mtop_d,
                                                 uint32_t* __restrict__ cg, fs_best* __restrict__ cbest,
                                                 fs_status* st) {
  __shared__ uint32_t s_w32[4];
  __shared__ uint32_t s_pref[257];
  __shared__ uint32_t s_lev[kLevWin * kLevMaxN];
  __shared__ uint8_t s_first[kLevWin * kLevMaxN], s_dr[kLevWin * kLevMaxN];
  const uint32_t n_pend = min(st->lsh_pending, cap);
  const uint32_t nn = (uint32_t)L.nn;
  for (uint32_t j0 = blockIdx.x * kLevWin; j0 < n_pend; j0 += gridDim.x * kLevWin) {
    const uint32_t j = j0 + threadIdx.x;
    const uint32_t cnt = (threadIdx.x < kLevWin && j < n_pend) ? mcnt[j] : 0u;
    // Without the UniqueFilter a script window is kept once per table that found it, and the
    // copies' Levenshtein distance is the one number: a pair per DISTINCT script window of
    // the list (first[r]: the first rank with rank r's window; dr[k]: rank of the k-th distinct
    // one), up to ten times fewer pairs
    uint32_t dcnt = 0;
    for (uint32_t r = 0; r < cnt; ++r) {
      const uint32_t sr = mtop_s[(size_t)j * nn + r];
      uint32_t fr = r;
      for (uint32_t x = 0; x < r; ++x)
        if (mtop_s[(size_t)j * nn + x] == sr) { fr = x; break; }
      s_first[threadIdx.x * kLevMaxN + r] = (uint8_t)fr;
      if (fr == r) s_dr[threadIdx.x * kLevMaxN + dcnt++] = (uint8_t)r;
    }
    uint32_t total;
    const uint32_t base = block_excl_scan(dcnt, s_w32, &total);
    s_pref[threadIdx.x] = base;
    if (threadIdx.x == 255) s_pref[256] = total;
    __syncthreads();
    for (uint32_t q0 = 0; q0 < total; q0 += 256) {
      const uint32_t q = q0 + threadIdx.x;
      if (q < total) {
        uint32_t lo = 0, hi = 256;               // s_pref[lo] <= q < s_pref[hi]
        while (hi - lo > 1) {
          const uint32_t mid = (lo + hi) >> 1;
          if (s_pref[mid] <= q) lo = mid; else hi = mid;
        }
        const uint32_t r = s_dr[lo * kLevMaxN + (q - s_pref[lo])], jj = j0 + lo;
        const uint32_t s = mtop_s[(size_t)jj * nn + r];
        const uint64_t p = cpos[pend[jj]];
        uint32_t lv = FS_NONE;
        if (L.selflev) {
          // a match with the same id in every slot has the strings of the script window's own
          // ids: its distance was computed once per string table (k_selflev)
          Ids16 u, f;
          load_ids(L.stok + s, L.n, &u);
          load_ids(c.tok + p, L.n, &f);
          bool same = true;
#pragma unroll
          for (int k = 0; k < FS_MAX_WINDOW; ++k)
            if (k < L.n) same = same && u.v[k] == f.v[k];
          if (same) lv = L.selflev[s];
        }
        if (lv == FS_NONE) {
          Ids16 sid;
          load_ids((c.str ? c.str : c.tok) + p, L.n, &sid);
          bool bad = false;
#pragma unroll
          for (int k = 0; k < FS_MAX_WINDOW; ++k)
            if (k < L.n) bad = bad || sid.v[k] >= c.n_str;
          if (bad) { st->bad_string = 1; lv = 0; }
          else lv = lev_lane_ids(g, c, F, s, sid, st);
        }
        s_lev[lo * nn + r] = lv;
      }
    }
    __syncthreads();
    if (cnt) {
      fs_best b;
      b.pad = 0.0;
      for (uint32_t r = 0; r < cnt; ++r) {          // first minimum of dist * lev in rank order
        const double d = mtop_d[(size_t)j * nn + r];
        const uint32_t lv = s_lev[threadIdx.x * nn + s_first[threadIdx.x * kLevMaxN + r]];
        const double comb = __dmul_rn(d, (double)lv);
        if (r == 0 || comb < b.comb) {
          b.s = mtop_s[(size_t)j * nn + r]; b.lev = lv; b.dist = d; b.comb = comb;
        }
      }
      const uint32_t i = pend[j];
      cbest[i] = b;
      cg[i] = 0;
    }
    __syncthreads();
  }
}

}  // namespace

// ---- host side -----------------------------------------------------------------

static LshDev lsh_dev(const fs_index* ix) {
  LshDev L;
  L.atab = ix->d_atab.p; L.nt = ix->d_nt.p; L.boff = ix->d_boff.p; L.bids = ix->d_bids.p;
  L.ss = ix->d_ss.p; L.sw = ix->d_sw.p; L.q = ix->d_q.p; L.emb = ix->d_emb.p; L.stok = ix->d_stok.p;
  L.gtab = ix->d_gtab.n > 1 ? ix->d_gtab.p : nullptr; L.sidx = ix->d_sidx.p;
  L.spos = ix->d_spos.n > 1 ? ix->d_spos.p : nullptr;
  L.share_cnt = nullptr; L.oovmap = nullptr; L.log2_oovmap = 0; L.compa = nullptr; L.ssig = nullptr; L.sharef = nullptr; L.smap = nullptr; L.slists = nullptr; L.log2_smap = 0; L.log2_sharef = 0; L.share_flags = 0;
  L.share_lim = 0.0f; L.share_scale = 0.0; L.share_phi = 1.0; L.share_tau = 0.0; L.share_gamma = 1.0;
  if (ix->share_flags) {
    L.compa = ix->d_compa.p; L.ssig = ix->d_ssig.p; L.sharef = ix->d_sharef.p;
    L.log2_sharef = ix->log2_sharef; L.share_flags = ix->share_flags;
    if (ix->d_share_cnt.n > 1) L.share_cnt = reinterpret_cast<unsigned long long*>(ix->d_share_cnt.p);
    if (ix->log2_oovmap) { L.oovmap = reinterpret_cast<const uint2*>(ix->d_oovmap.p); L.log2_oovmap = ix->log2_oovmap; }
    L.smap = reinterpret_cast<const uint2*>(ix->d_smap.p); L.slists = reinterpret_cast<const uint4*>(ix->d_slists.p); L.log2_smap = ix->log2_smap;
    L.share_gamma = ix->share_gamma;
    L.share_tau = 1.0 - ix->cfg.distance_threshold - 1e-6;
    L.share_phi = (1.0 - L.share_tau * L.share_tau) / (1.0 - L.share_gamma * L.share_gamma);
    L.share_lim = (float)((1.0 - L.share_phi) * (1.0 - 1e-6));
    L.share_scale = ldexp(1.0, 20) / std::max(ix->info.norm_max * ix->info.norm_max * (1.0 + 1e-9), 3.0);
  }
  L.emap = nullptr; L.log2_emap = 0; L.emap_comp = 0; L.skeys = ix->d_skeys.n > 1 ? ix->d_skeys.p : nullptr;
  L.atab32 = ix->d_atab32.n > 1 ? ix->d_atab32.p : nullptr; L.amax = ix->d_amax.p;
  L.nt32 = ix->d_nt32.p; L.ntmax = ix->d_ntmax.p;
  L.wild = nullptr; L.log2_wild = 0; L.wild_tok = nullptr; L.selflev = nullptr; L.wmap = nullptr; L.log2_wmap = 0;
  L.V = (uint32_t)ix->n_vec; L.W = (uint32_t)ix->n_windows;
  L.n = (int)ix->cfg.window_size; L.H = (int)ix->cfg.number_of_hashes;
  L.B = (int)ix->cfg.hash_dimensions; L.D = (int)ix->cfg.emb_dim; L.C = L.H * L.B;
  L.Cp = (L.C + 3) & ~3;
  L.nn = (int)ix->cfg.nearest_n; L.unique = ix->cfg.unique_filter ? 1 : 0;
  L.thr = ix->cfg.distance_threshold;
  L.cmax = ix->lsh_cmax;
  {
    // n * 2^-22; FS_LSH_F32_SLACK multiplies it (tests force the float64 fallback),
    // FS_LSH_F32=0 disables the float32 path
    L.bound_scale = (float)((double)ix->cfg.window_size * ldexp(1.0, -22) * ix->sw.lsh_f32_slack);
    if (!ix->sw.lsh_f32) L.atab32 = nullptr;
    L.m_min = ix->lsh_m_min;
    L.diag = ix->sw.lsh_diag;
    L.serial_neighbours = ix->sw.lsh_serial ? 1 : 0;
  }
  return L;
}

// Engine.store_vector for every script window (search.py:122-123)
// smallest number of id-identical slots with which a window pair can reach
// cos >= 1 - thr - 1e-6 (n: only identical windows)
static int lsh_m_min(const fs_index* ix) {
  const int n = (int)ix->cfg.window_size;
  const double qmin = ix->info.norm_min * ix->info.norm_min, qmax = ix->info.norm_max * ix->info.norm_max;
  if (!(qmin > 0.0) || !(ix->lsh_cmax < 1.0)) return 0;
  const double lim = (1.0 - ix->cfg.distance_threshold - 1e-6) * n * qmin * (1.0 - 1e-9);
  int m = 0;
  while (m <= n && (m + (n - m) * ix->lsh_cmax) * qmax < lim) ++m;
  return m > n ? n : m;          // n + 1 would mean "nothing can match": exact windows still do
}

// The grouped filter of one-slot-wildcard keys (fs_hash.h) over the id sequence `st` (vector ids,
// or component ids): log2 of its 16-byte blocks and the blocks, about 24 filter bits per key.
static int build_wild_filter(const std::vector<uint32_t>& st, uint64_t W, int n, std::vector<uint32_t>* out) {
  int lb = 8;
  while (lb < 26 && ((uint64_t)128 << lb) < W * n * 24) ++lb;
  out->assign((size_t)4 << lb, 0u);
  for (uint64_t w = 0; w < W; ++w) {
    uint32_t term[FS_MAX_WINDOW], fold = 0, gfold[3] = {0, 0, 0};
    for (int k = 0; k < n; ++k) {
      term[k] = fs_rotl(fs_premix(st[w + k]), fs_rot_of(n - 1 - k));
      fold ^= term[k];
      gfold[fs_wild_group(k, n)] ^= term[k];
    }
    for (int k = 0; k < n; ++k) {
      const uint32_t h = fs_wild_fkey(fold, term[k], k);
      const int X = fs_wild_group(k, n);
      uint32_t* blk = out->data() + 4 * (size_t)fs_wild_block(fold ^ gfold[X], X, lb);
      for (int i = 0; i < 4; ++i) blk[i] |= 1u << fs_wild_fbit(h, i);
    }
  }
  return lb;
}

// Tables with near-synonyms (every real embedding table): the proof that a neighbour within the
// threshold shares n or n - 1 vector ids with the window fails, but a weaker one holds.  With
// x_k = |f_k|, y_k = |s_k|, c_k = cos(f_k, s_k):
//   cos(F, S) |x| |y| = sum x_k y_k c_k = sum x_k y_k - sum d_k <= |x| |y| - sum d_k,
//   d_k = (1 - c_k) x_k y_k >= 0,
// so a record (cos > 1 - thr) needs sum d_k < thr |x| |y| <= T = thr n a_max^2, and at most ONE
// slot has d_k >= T / 2, i.e. cos(f_k, s_k) <= 1 - T / (2 |f_k| |s_k|).  Call a pair of a script
// vector and a table vector above that line *near* (unit vectors, n = 6, thr = 0.1: cos > 0.7)
// and give every table vector the id of its connected component in the graph of near pairs:
// a window can have a neighbour within the threshold only if its component ids equal a
// script window's in n - 1 slots or more.  That is the test the filters of the
// one-slot case make on vector ids (k_scan_near, the wildcard keys), here made on component
// ids; the windows that pass get the full LSH work (their per-n-gram record where their vector
// ids are a script n-gram's).  Sound: a filter only removes windows that cannot have a
// neighbour.  Not used when the components are too coarse to filter (one of them holding an
// eighth of the table or more: zero rows, hubs of tiny norm) or a side holds OOV vectors.
// The one-slot-wildcard keys of every distinct script n-gram (by vector ids: gram g's first
// window is gpos[g][0]) as an exact map key -> g, the keys made of `ids` (the script's vector
// ids, or their component ids): buckets of four {key, g + 1}, a full bucket spills into the next
// (k_lsh_batch gives a window up to the bucket walk when it meets a full one).
static int build_emap(fs_index* ix, const std::vector<uint32_t>& ids, DBuf<uint32_t>* out, int* log2_out) {
  const int n = (int)ix->cfg.window_size;
  const uint32_t nn = ix->cfg.nearest_n;
  const uint32_t G = ix->n_grams;
  *log2_out = 0;
  if (!G || !ix->d_gpos.p) return FS_OK;
  std::vector<uint32_t> gpos((size_t)G * nn);
  FS_HIP(hipMemcpy(gpos.data(), ix->d_gpos.p, gpos.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  int lm = 8;                                    // two buckets per entry: a full one (four entries) is rare
  while (lm < 26 && ((uint64_t)1 << lm) < 2 * (uint64_t)G * n) ++lm;
  std::vector<uint32_t> emap((size_t)8 << lm, 0u);
  const uint32_t mask = (1u << lm) - 1;
  for (uint32_t g = 0; g < G; ++g) {
    const uint32_t w = gpos[(size_t)g * nn];
    uint32_t term[FS_MAX_WINDOW], fold = 0;
    for (int k = 0; k < n; ++k) {
      term[k] = fs_rotl(fs_premix(ids[w + k]), fs_rot_of(n - 1 - k));
      fold ^= term[k];
    }
    for (int k = 0; k < n; ++k) {
      const uint32_t h = fs_wild_key(fold, term[k], k);
      uint32_t bkt = fs_wmap_slot(h, lm);
      for (;;) {
        uint32_t* e = emap.data() + 8 * (size_t)bkt;
        int at = 0;
        while (at < 4 && e[2 * at + 1]) ++at;
        if (at < 4) { e[2 * at] = h; e[2 * at + 1] = g + 1; break; }
        bkt = (bkt + 1) & mask;
      }
    }
  }
  FS_TRY(out->upload(emap.data(), emap.size(), ix->stream));
  FS_HIP(hipStreamSynchronize(ix->stream));
  *log2_out = lm;
  return FS_OK;
}

static int fs_build_components(fs_index* ix) {
  ix->syn_ok = false;
  const int n = (int)ix->cfg.window_size, D = (int)ix->cfg.emb_dim;
  const uint64_t V = ix->n_vec, W = ix->n_windows;
  if (!ix->sw.lsh_syn || ix->script_oov || !W || V > FS_MAX_EXACT_ID || n < 6 ||
      !(n <= 10 || n == 12) || !(ix->info.norm_max > 0.0)) return FS_OK;
  std::vector<uint32_t> st(ix->n_script);
  FS_HIP(hipMemcpyAsync(st.data(), ix->d_stok.p, ix->n_script * sizeof(uint32_t), hipMemcpyDeviceToHost, ix->stream));
  FS_HIP(hipStreamSynchronize(ix->stream));
  std::vector<uint32_t> rows_u;
  {
    std::vector<uint8_t> seen(V, 0);
    for (uint32_t id : st)
      if (!seen[id]) { seen[id] = 1; rows_u.push_back(id); }
  }
  // near pairs (script vector, table vector) from the device
  const uint32_t cap = 1u << 23;
  DBuf<float> embT;
  DBuf<uint32_t> d_rows_u, d_cnt;
  DBuf<uint2> d_pairs;
  FS_TRY(embT.reserve((size_t)V * D));
  FS_TRY(d_rows_u.upload(rows_u.data(), rows_u.size(), ix->stream));
  FS_TRY(d_cnt.reserve(1));
  FS_TRY(d_pairs.reserve(cap));
  FS_HIP(hipMemsetAsync(d_cnt.p, 0, sizeof(uint32_t), ix->stream));
  const double T = ix->cfg.distance_threshold * n * ix->info.norm_max * ix->info.norm_max * (1.0 + 1e-6);
  FS_TRY(fs_launch_near_pairs(ix->d_emb.p, V, D, d_rows_u.p, (uint32_t)rows_u.size(), ix->d_q.p, embT.p,
                              (float)(T / 2.0), -2.0f, d_pairs.p, cap, d_cnt.p, ix->stream));
  uint32_t n_pairs = 0;
  FS_HIP(hipMemcpyAsync(&n_pairs, d_cnt.p, sizeof n_pairs, hipMemcpyDeviceToHost, ix->stream));
  FS_HIP(hipStreamSynchronize(ix->stream));
  if (n_pairs > cap) return FS_OK;                 // (far too many near pairs: nothing to filter with)
  std::vector<uint2> pairs(n_pairs);
  if (n_pairs) FS_HIP(hipMemcpy(pairs.data(), d_pairs.p, (size_t)n_pairs * sizeof(uint2), hipMemcpyDeviceToHost));
  // connected components (union-find), ids dense in order of the smallest member
  std::vector<uint32_t> parent(V);
  for (uint64_t v = 0; v < V; ++v) parent[v] = (uint32_t)v;
  auto find = [&](uint32_t v) {
    while (parent[v] != v) { parent[v] = parent[parent[v]]; v = parent[v]; }
    return v;
  };
  for (const uint2& e : pairs) {
    const uint32_t a = find(e.x), b = find(e.y);
    if (a != b) parent[a > b ? a : b] = a > b ? b : a;
  }
  std::vector<uint32_t> comp(V), size;
  {
    std::vector<uint32_t> id_of(V, FS_NONE);
    for (uint64_t v = 0; v < V; ++v) {
      const uint32_t r = find((uint32_t)v);
      if (id_of[r] == FS_NONE) { id_of[r] = (uint32_t)size.size(); size.push_back(0); }
      comp[v] = id_of[r];
      ++size[comp[v]];
    }
  }
  ix->n_comp = (uint32_t)size.size();
  ix->comp_sizes = size;
  ix->comp_largest = *std::max_element(size.begin(), size.end());
  if ((uint64_t)ix->comp_largest * 8 > V && ix->comp_largest > 64) return FS_OK;
  FS_TRY(ix->d_comp.upload(comp.data(), comp.size(), ix->stream));
  // the two filters of the one-slot case, over the script's component ids
  std::vector<uint32_t> sc(st.size());
  for (size_t i = 0; i < st.size(); ++i) sc[i] = comp[st[i]];
  std::vector<uint32_t> sub(1u << fs_scan_near_log2(ix), 0u);
  const int K = fs_scan_near_k(n);
  for (uint64_t i = 0; i + K <= sc.size(); ++i) {
    uint32_t word, bit;
    fs_scan_near_bit(ix, sc.data() + i, &word, &bit);
    sub[word] |= 1u << bit;
  }
  FS_TRY(ix->d_sfilter3c.upload(sub.data(), sub.size(), ix->stream));
  std::vector<uint32_t> wild;
  const int lwild = build_wild_filter(sc, W, n, &wild);
  FS_TRY(ix->d_wildc.upload(wild.data(), wild.size(), ix->stream));
  ix->log2_wildc = lwild;
  FS_TRY(build_emap(ix, sc, &ix->d_emapc, &ix->log2_emapc));
  if (n == 6) {
    // the keys of slots 2 and 3 in a filter of their own for k_scan_near (fs_scan.hip)
    std::vector<uint32_t> keys((size_t)1 << FS_NEAR6_LOG2_WORDS, 0u);
    for (uint64_t w = 0; w < W; ++w) {
      uint32_t term[6], fold = 0;
      for (int k = 0; k < 6; ++k) {
        term[k] = fs_rotl(fs_premix(sc[w + k]), fs_rot_of(5 - k));
        fold ^= term[k];
      }
      for (int k = 2; k <= 3; ++k) {
        const uint32_t h = fs_wild_key(fold, term[k], k);
        keys[fs_bloom_word(h, FS_NEAR6_LOG2_WORDS)] |= fs_bloom_mask(h);
      }
    }
    FS_TRY(ix->d_keys6c.upload(keys.data(), keys.size(), ix->stream));
  }
  FS_HIP(hipStreamSynchronize(ix->stream));
  ix->syn_ok = true;
  return FS_OK;
}

// The share rule's index side (k_lsh_scan, "the share rule" above): the components of the angular
// relation cos > gamma over (script vector, table vector) pairs, the proof that out-of-vocabulary
// fan tokens are far from every script vector, and the filter of the script windows' subset keys.
static int fs_build_share(fs_index* ix) {
  ix->share_flags = 0;
  const int n = (int)ix->cfg.window_size, D = (int)ix->cfg.emb_dim;
  const uint64_t V = ix->n_vec, W = ix->n_windows;
  const double gamma = ix->sw.share_gamma;
  const double tau = 1.0 - ix->cfg.distance_threshold - 1e-6;
  if (!(ix->sw.lsh_share & 3) || !W || !V || V > FS_MAX_EXACT_ID || n < 2 ||
      !(ix->info.norm_max > 0.0) || !(gamma >= 0.05 && gamma <= 0.995) || !(tau > gamma + 1e-3))
    return FS_OK;
  // (a script with out-of-vocabulary tokens: share_comp's case analysis needs 2/3 to be far)
  if (ix->script_oov && !(gamma >= 0.668)) return FS_OK;
  hipStream_t s = ix->stream;
  std::vector<uint32_t> st(ix->n_script);
  FS_HIP(hipMemcpyAsync(st.data(), ix->d_stok.p, ix->n_script * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  FS_HIP(hipStreamSynchronize(s));
  std::vector<uint32_t> rows_u;
  {
    std::vector<uint8_t> seen(V, 0);
    for (uint32_t id : st)
      if (!(id & FS_OOV_FLAG) && !seen[id]) { seen[id] = 1; rows_u.push_back(id); }
  }
  const uint32_t cap = 1u << 23;
  DBuf<float> embT;
  DBuf<uint32_t> d_rows_u, d_cnt;
  DBuf<uint2> d_pairs;
  FS_TRY(embT.reserve((size_t)V * D));
  FS_TRY(d_rows_u.upload(rows_u.data(), rows_u.size(), s));
  FS_TRY(d_cnt.reserve(2));
  FS_TRY(d_pairs.reserve(cap));
  FS_HIP(hipMemsetAsync(d_cnt.p, 0, 2 * sizeof(uint32_t), s));
  FS_TRY(fs_launch_near_pairs(ix->d_emb.p, V, D, d_rows_u.p, (uint32_t)rows_u.size(), ix->d_q.p, embT.p, 0.0f,
                              (float)gamma, d_pairs.p, cap, d_cnt.p, s));
  // (out-of-vocabulary fan tokens against the script's rows; with out-of-vocabulary tokens in the
  // script also those against every row a fan token may be)
  if (ix->script_oov)
    FS_TRY(fs_launch_coordmax(ix->d_emb.p, D, nullptr, (uint32_t)V, ix->d_q.p, reinterpret_cast<int*>(d_cnt.p + 1), s));
  else
    FS_TRY(fs_launch_coordmax(ix->d_emb.p, D, d_rows_u.p, (uint32_t)rows_u.size(), ix->d_q.p,
                              reinterpret_cast<int*>(d_cnt.p + 1), s));
  uint32_t res[2] = {0, 0};
  FS_HIP(hipMemcpyAsync(res, d_cnt.p, sizeof res, hipMemcpyDeviceToHost, s));
  FS_HIP(hipStreamSynchronize(s));
  const uint32_t n_pairs = res[0];
  if (n_pairs > cap) return FS_OK;                 // (far too many near pairs: nothing to filter with)
  float kappa;
  memcpy(&kappa, &res[1], sizeof kappa);
  const bool oov_far = sqrt(3.0) * (double)kappa * (1.0 + 1e-6) <= gamma - 1e-4;
  std::vector<uint2> pairs(n_pairs);
  if (n_pairs) FS_HIP(hipMemcpy(pairs.data(), d_pairs.p, (size_t)n_pairs * sizeof(uint2), hipMemcpyDeviceToHost));
  std::vector<uint32_t> parent(V);
  for (uint64_t v = 0; v < V; ++v) parent[v] = (uint32_t)v;
  auto find = [&](uint32_t v) {
    while (parent[v] != v) { parent[v] = parent[parent[v]]; v = parent[v]; }
    return v;
  };
  for (const uint2& e : pairs) {
    const uint32_t a = find(e.x), b = find(e.y);
    if (a != b) parent[a > b ? a : b] = a > b ? b : a;
  }
  std::vector<uint32_t> comp(V), size;
  {
    std::vector<uint32_t> id_of(V, FS_NONE);
    for (uint64_t v = 0; v < V; ++v) {
      const uint32_t r = find((uint32_t)v);
      if (id_of[r] == FS_NONE) { id_of[r] = (uint32_t)size.size(); size.push_back(0); }
      comp[v] = id_of[r];
      ++size[comp[v]];
    }
  }
  ix->comp_sizes = size;                            // (fs_index_component_sizes: the angular relation's, where the rule is built)
  ix->share_comps = (uint32_t)size.size();
  ix->share_largest = *std::max_element(size.begin(), size.end());
  int flags = ix->sw.lsh_share & 47;
  if (!oov_far) {
    if (ix->script_oov) return FS_OK;               // (the script's 3-hot vectors may be near table rows: no rule)
    flags |= 8;
  }
  if ((flags & 8) || ix->script_oov) flags &= ~4;  // (a slot that agrees with anything has no share on the script's side)
  if (n > FS_MAX_WINDOW || n > 12) flags &= ~1;
  if (n > 6) flags &= ~4;                          // (run by run: the fan window's side only)
  uint64_t n_masks_all = 0;                        // subsets per script window, over its runs (fs_share_blocks)
  for (int r = 0; r < fs_share_blocks(n); ++r)
    n_masks_all += ((uint64_t)1 << (fs_share_block_start(n, r + 1) - fs_share_block_start(n, r))) - 1;
  if ((flags & 35) != 35 || W * n_masks_all > ((uint64_t)1 << 25)) flags &= ~32;   // (the enumeration needs the gate and the pairs' test)
  if (flags & 32) flags &= ~4;                     // (... and every subset of every script window in the filter)
  if (!(flags & 3)) return FS_OK;
  FS_TRY(ix->d_compa.upload(comp.data(), comp.size(), s));
  // the script's out-of-vocabulary vectors (share_comp): a component per distinct set of three
  // positions, in a map for the fan tokens; a component of its own per vector of fewer positions,
  // its pair of positions in the map so that a fan token that contains it counts as agreeing with
  // anything
  ix->log2_oovmap = 0;
  std::vector<uint32_t> oov_comp_of;               // per script token (OOV ones), by index into st
  std::vector<std::pair<uint32_t, uint32_t>> oov_entries;   // {key, component}
  auto hot_of = [&](uint32_t id, uint32_t h[3]) {
    const uint32_t code = id & ~FS_OOV_FLAG, Du = (uint32_t)D;
    h[2] = code % Du; h[1] = (code / Du) % Du; h[0] = code / (Du * Du);
    std::sort(h, h + 3);
  };
  auto q_host = [&](uint32_t id, const std::vector<double>& qv) {
    if (!(id & FS_OOV_FLAG)) return qv[id];
    uint32_t h[3];
    hot_of(id, h);
    return 1.0 + (h[1] != h[0] ? 1.0 : 0.0) + (h[2] != h[1] ? 1.0 : 0.0);
  };
  std::vector<uint32_t> sc(st.size() + FS_MAX_WINDOW, FS_NONE);
  {
    uint32_t next = (uint32_t)V;
    std::vector<std::pair<uint64_t, uint32_t>> sets;         // distinct position sets -> component
    for (size_t i = 0; i < st.size(); ++i) {
      if (!(st[i] & FS_OOV_FLAG)) { sc[i] = comp[st[i]]; continue; }
      uint32_t h[3];
      hot_of(st[i], h);
      const uint64_t set = ((uint64_t)h[0] << 40) | ((uint64_t)h[1] << 20) | h[2];
      uint32_t c = FS_NONE;
      for (const auto& e : sets)
        if (e.first == set) { c = e.second; break; }
      if (c == FS_NONE) {
        c = next++;
        sets.push_back({set, c});
        const uint32_t Du = (uint32_t)D;
        if (h[0] != h[1] && h[1] != h[2]) oov_entries.push_back({(h[0] * Du + h[1]) * Du + h[2], c});
        else if (h[0] != h[2]) oov_entries.push_back({0x80000000u | (h[0] * Du + h[2]), FS_WILD});   // two positions
      }
      sc[i] = c;
    }
    if ((uint64_t)D * D * D >= (1ull << 31)) { if (!oov_entries.empty()) return FS_OK; }
    if (!oov_entries.empty()) {
      int lo = 4;
      while (((size_t)1 << lo) < 2 * oov_entries.size()) ++lo;
      std::vector<uint32_t> m((size_t)2 << lo, 0u);
      const uint32_t mask = (1u << lo) - 1;
      for (const auto& e : oov_entries) {
        uint32_t at = fs_mix24(e.first) & mask;
        while (m[2 * at + 1]) at = (at + 1) & mask;
        m[2 * at] = e.first;
        m[2 * at + 1] = e.second + 1;                // (0: empty; FS_WILD + 1 = FS_NONE: share_comp reads it as "there")
      }
      FS_TRY(ix->d_oovmap.upload(m.data(), m.size(), s));
      ix->log2_oovmap = lo;
    }
  }
  std::vector<uint64_t> sig(W, 0);
  {
    const int b = fs_share_sig_bits(n);
    for (uint64_t w = 0; w < W; ++w)
      for (int k = 0; k < n; ++k) sig[w] |= (uint64_t)fs_share_sig(sc[w + k], n) << (k * b);
    FS_TRY(ix->d_ssig.upload(sig.data(), sig.size(), s));
  }
  if (flags & 1) {
    std::vector<double> q(V);
    FS_HIP(hipMemcpyAsync(q.data(), ix->d_q.p, V * sizeof(double), hipMemcpyDeviceToHost, s));
    FS_HIP(hipStreamSynchronize(s));
    const uint64_t keys = W * n_masks_all / ((flags & 4) ? 3 : 1);
    int lw = 10;
    while (lw < 26 && ((uint64_t)1 << lw) * 4 < keys * 3) ++lw;    // about 24 filter bits per key and more
    std::vector<uint32_t> f((size_t)1 << lw, 0u);
    const double phi = (1.0 - tau * tau) / (1.0 - gamma * gamma);
    for (uint64_t w = 0; w < W; ++w) {
      uint32_t t[FS_MAX_WINDOW];
      double qs[FS_MAX_WINDOW], all = 0.0;
      for (int k = 0; k < n; ++k) {
        t[k] = fs_share_term(sc[w + k], k);
        qs[k] = q_host(st[w + k], q);
        all += qs[k];
      }
      const double need = (1.0 - phi) * all * (1.0 - 1e-6);
      for (int r = 0; r < fs_share_blocks(n); ++r) {
        const int k0 = fs_share_block_start(n, r), k1 = fs_share_block_start(n, r + 1);
        for (uint32_t sub = 1; sub < (1u << (k1 - k0)); ++sub) {
          const uint32_t m = sub << k0;
          uint32_t fold = 0;
          double sum = 0.0;
          for (int k = k0; k < k1; ++k)
            if ((m >> k) & 1u) { fold ^= t[k]; sum += qs[k]; }
          if ((flags & 4) && sum < need) continue;   // (only the subsets that hold the share on this side too)
          const uint32_t h = fs_share_key(fold, m);
          f[fs_bloom_word(h, lw)] |= fs_bloom_mask(h);
        }
      }
    }
    FS_TRY(ix->d_sharef.upload(f.data(), f.size(), s));
    ix->log2_sharef = lw;
    if (flags & 32) {
      // the same keys as an exact map: key -> its script windows
      std::vector<uint64_t> ent;
      ent.reserve(W * n_masks_all);
      for (uint64_t w = 0; w < W; ++w) {
        uint32_t t[FS_MAX_WINDOW];
        for (int k = 0; k < n; ++k) t[k] = fs_share_term(sc[w + k], k);
        for (int r = 0; r < fs_share_blocks(n); ++r) {
          const int k0 = fs_share_block_start(n, r), k1 = fs_share_block_start(n, r + 1);
          for (uint32_t sub = 1; sub < (1u << (k1 - k0)); ++sub) {
            const uint32_t m = sub << k0;
            uint32_t fold = 0;
            for (int k = k0; k < k1; ++k)
              if ((m >> k) & 1u) fold ^= t[k];
            ent.push_back((uint64_t)fs_share_key(fold, m) << 32 | w);
          }
        }
      }
      std::sort(ent.begin(), ent.end());
      uint64_t distinct = 0;
      for (size_t i = 0; i < ent.size(); ++i) distinct += i == 0 || (ent[i] >> 32) != (ent[i - 1] >> 32);
      int lm = 8;                                  // two buckets per key: a full one (four entries) is rare
      while (lm < 26 && ((uint64_t)1 << lm) < 2 * distinct) ++lm;
      std::vector<uint32_t> smap((size_t)8 << lm, 0u);
      std::vector<uint4> lists;
      lists.reserve(ent.size() + distinct + 1);
      lists.push_back(make_uint4(0, 0, 0, 0));      // (a list is named by the index of its first script window: never 0)
      const uint32_t bmask = (1u << lm) - 1;
      for (size_t i = 0; i < ent.size();) {
        const uint32_t h = (uint32_t)(ent[i] >> 32);
        size_t e1 = i;
        while (e1 < ent.size() && (uint32_t)(ent[e1] >> 32) == h) ++e1;
        lists.push_back(make_uint4((uint32_t)(e1 - i), 0, 0, 0));       // its length, then its script windows
        const uint32_t first = (uint32_t)lists.size();
        for (size_t x = i; x < e1; ++x) {
          const uint32_t w = (uint32_t)ent[x];
          lists.push_back(make_uint4(w, (uint32_t)sig[w], (uint32_t)(sig[w] >> 32), 0));
        }
        uint32_t bkt = fs_wmap_slot(h, lm);
        for (;;) {
          uint32_t* e = smap.data() + 8 * (size_t)bkt;
          int at = 0;
          while (at < 4 && e[2 * at + 1]) ++at;
          if (at < 4) { e[2 * at] = h; e[2 * at + 1] = first; break; }
          bkt = (bkt + 1) & bmask;
        }
        i = e1;
      }
      FS_TRY(ix->d_smap.upload(smap.data(), smap.size(), s));
      FS_TRY(ix->d_slists.upload(reinterpret_cast<const uint32_t*>(lists.data()), lists.size() * 4, s));
      ix->log2_smap = lm;
    }
  }
  if (getenv("FS_SHARE_COUNT")) {
    FS_TRY(ix->d_share_cnt.reserve(16));
    FS_HIP(hipMemsetAsync(ix->d_share_cnt.p, 0, 16 * sizeof(uint32_t), s));
  }
  FS_HIP(hipStreamSynchronize(s));
  ix->share_gamma = gamma;
  ix->share_flags = flags | 16;                    // (bit 4: in use, whatever else is set)
  return FS_OK;
}

int fs_lsh_build(fs_index* ix) {
  if (ix->lsh_ready) return FS_OK;
  ix->lsh_m_min = lsh_m_min(ix);
  ix->near8 = fs_scan_near8_wanted(ix);
  if ((int)ix->cfg.window_size - ix->lsh_m_min == 1 && !ix->script_oov && ix->cfg.window_size >= 4 &&
      ix->n_vec <= FS_MAX_EXACT_ID) {
    // a neighbour differs from the window in at most one slot: one bit per script 3-gram
    // for the integer prefilter (k_scan_near, fs_scan.hip)
    std::vector<uint32_t> st(ix->n_script);
    FS_HIP(hipMemcpyAsync(st.data(), ix->d_stok.p, ix->n_script * sizeof(uint32_t), hipMemcpyDeviceToHost,
                          ix->stream));
    FS_HIP(hipStreamSynchronize(ix->stream));
    std::vector<uint32_t> sub(1u << fs_scan_near_log2(ix), 0u);
    for (uint64_t i = 0; i + 3 <= ix->n_script; ++i) {
      uint32_t word, bit;
      fs_scan_near_bit(ix, st.data() + i, &word, &bit);
      sub[word] |= 1u << bit;
    }
    FS_TRY(ix->d_sfilter3.upload(sub.data(), sub.size(), ix->stream));
    // ... and the n one-slot-wildcard keys of every script window, about 24 filter bits
    // per key (k_lsh_verify drops a window none of whose keys is present)
    const int n = (int)ix->cfg.window_size;
    const uint64_t W = ix->n_windows;
    std::vector<uint32_t> wild;
    const int lwild = build_wild_filter(st, W, n, &wild);
    FS_TRY(ix->d_wild.upload(wild.data(), wild.size(), ix->stream));
    ix->log2_wild = lwild;
    // ... and as an exact map, one entry per distinct n-gram (its first window) and slot
    {
      std::vector<uint32_t> first;                       // first window of every distinct n-gram
      {
        std::vector<uint32_t> order(W);
        for (uint64_t w = 0; w < W; ++w) order[w] = (uint32_t)w;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
          return std::lexicographical_compare(st.begin() + a, st.begin() + a + n, st.begin() + b, st.begin() + b + n);
        });
        for (uint64_t i = 0; i < W; ++i)
          if (i == 0 || !std::equal(st.begin() + order[i], st.begin() + order[i] + n, st.begin() + order[i - 1]))
            first.push_back(order[i]);
      }
      // buckets of four {key, window + 1}, about one entry per bucket; a full bucket spills
      // into the next one (the kernel gives a window up when it meets a full bucket)
      int lm = 8;
      while (lm < 26 && ((uint64_t)1 << lm) < first.size() * (uint64_t)n) ++lm;
      std::vector<uint32_t> wmap((size_t)8 << lm, 0u);
      const uint32_t mask = (1u << lm) - 1;
      for (uint32_t w : first) {
        uint32_t term[FS_MAX_WINDOW], fold = 0;
        for (int k = 0; k < n; ++k) {
          term[k] = fs_rotl(fs_premix(st[w + k]), fs_rot_of(n - 1 - k));
          fold ^= term[k];
        }
        for (int k = 0; k < n; ++k) {
          const uint32_t h = fs_wild_key(fold, term[k], k);
          uint32_t bkt = fs_wmap_slot(h, lm);
          for (;;) {
            uint32_t* e = wmap.data() + 8 * (size_t)bkt;
            int at = 0;
            while (at < 4 && e[2 * at + 1]) ++at;
            if (at < 4) { e[2 * at] = h; e[2 * at + 1] = w + 1; break; }
            bkt = (bkt + 1) & mask;
          }
        }
      }
      FS_TRY(ix->d_wmap.upload(wmap.data(), wmap.size(), ix->stream));
      ix->log2_wmap = lm;
    }
    FS_TRY(build_emap(ix, st, &ix->d_emap, &ix->log2_emap));
    FS_HIP(hipStreamSynchronize(ix->stream));
  }
  if ((int)ix->cfg.window_size - ix->lsh_m_min > 1) FS_TRY(fs_build_components(ix));
  // (where neither integer prefilter applies the search is k_lsh_scan: the share rule is for it)
  if (((int)ix->cfg.window_size - ix->lsh_m_min > 1 || ix->script_oov || ix->cfg.window_size < 4 ||
       ix->n_vec > FS_MAX_EXACT_ID) && !ix->syn_ok)
    FS_TRY(fs_build_share(ix));
  if (!ix->d_normals.p) { fs_set_error("normals are required for the LSH pipeline"); return FS_E_INVALID; }
  hipStream_t s = ix->stream;
  const int n = (int)ix->cfg.window_size, D = (int)ix->cfg.emb_dim;
  const int H = (int)ix->cfg.number_of_hashes, B = (int)ix->cfg.hash_dimensions, C = H * B;
  const uint64_t V = ix->n_vec, W = ix->n_windows;
  FS_TRY(ix->d_nt.reserve((size_t)n * D * C));
  FS_TRY(ix->d_atab.reserve((size_t)n * V * C));
  const int Cp = (C + 3) & ~3;
  FS_TRY(ix->d_atab32.reserve((size_t)n * V * Cp + 4));
  FS_TRY(ix->d_amax.reserve((size_t)n * V + 1));
  FS_TRY(ix->d_ss.reserve(W));
  FS_TRY(ix->d_sw.reserve(W));
  hipLaunchKernelGGL(k_nt, dim3(1024), dim3(256), 0, s, ix->d_normals.p, n, D, C, ix->d_nt.p);
  FS_TRY(ix->d_nt32.reserve((size_t)n * D * Cp + 4));
  FS_TRY(ix->d_ntmax.reserve((size_t)n * D + 1));
  hipLaunchKernelGGL(k_nt32, dim3((uint32_t)(n * D)), dim3(256), 0, s, ix->d_nt.p, n * D, C, Cp, ix->d_nt32.p, ix->d_ntmax.p);
  if (V)
    hipLaunchKernelGGL(k_atab, dim3((uint32_t)V, n), dim3(256), 0, s, ix->d_nt.p, ix->d_emb.p,
                       (uint32_t)V, D, C, Cp, ix->d_atab.p, ix->d_atab32.p, ix->d_amax.p);
  FS_HIP(hipGetLastError());
  // pair dot products g(script row, table row): one 8-byte lookup per window slot
  // instead of D multiply-adds when a candidate's exact distance is needed.  Capped
  // at 64 GiB of the 288 GB HBM; beyond that g is computed on the fly.
  {
    std::vector<uint32_t> stok_h(ix->n_script);
    FS_HIP(hipMemcpyAsync(stok_h.data(), ix->d_stok.p, ix->n_script * sizeof(uint32_t),
                          hipMemcpyDeviceToHost, s));
    FS_HIP(hipStreamSynchronize(s));
    std::vector<int32_t> sidx(std::max<uint64_t>(V, 1), -1);
    std::vector<uint32_t> srow;
    for (uint32_t id : stok_h)
      if (!(id & FS_OOV_FLAG) && sidx[id] < 0) { sidx[id] = (int32_t)srow.size(); srow.push_back(id); }
    FS_TRY(ix->d_sidx.upload(sidx.data(), sidx.size(), s));
    const uint64_t bytes = (uint64_t)srow.size() * V * sizeof(double);
    if (!srow.empty() && V && bytes <= (64ull << 30) && !ix->sw.lsh_no_gtab) {
      DBuf<uint32_t> d_srow;
      DBuf<float> embT;
      FS_TRY(d_srow.upload(srow.data(), srow.size(), s));
      FS_TRY(embT.reserve((size_t)V * D));
      FS_TRY(ix->d_gtab.reserve((size_t)srow.size() * V));
      hipLaunchKernelGGL(k_embT, dim3((uint32_t)((V + 255) / 256)), dim3(256), 0, s, ix->d_emb.p,
                         (uint32_t)V, D, embT.p);
      for (size_t r0 = 0; r0 < srow.size(); r0 += 32768) {        // grid.y limit
        const uint32_t rows = (uint32_t)std::min<size_t>(32768, srow.size() - r0);
        hipLaunchKernelGGL(k_gtab, dim3((uint32_t)((V + 255) / 256), rows), dim3(256),
                           D * sizeof(float), s, ix->d_emb.p, embT.p, (uint32_t)V, D,
                           d_srow.p + r0, ix->d_gtab.p + r0 * V);
      }
      FS_HIP(hipGetLastError());
      FS_HIP(hipStreamSynchronize(s));
    }
  }
  const uint32_t nb = 1u << B;
  FS_TRY(ix->d_boff.reserve((size_t)H * (nb + 1)));
  FS_TRY(ix->d_bids.reserve((size_t)H * std::max<uint64_t>(W, 1)));
  FS_HIP(hipMemsetAsync(ix->d_boff.p, 0, (size_t)H * (nb + 1) * sizeof(uint32_t), s));
  if (W) {
    LshDev L = lsh_dev(ix);
    hipLaunchKernelGGL(k_ss, dim3((uint32_t)((W + 255) / 256)), dim3(256), 0, s, ix->d_stok.p,
                       (uint32_t)W, L, ix->d_ss.p, ix->d_sw.p);
    FS_TRY(ix->d_spos.reserve(ix->n_script + FS_MAX_WINDOW));
    FS_HIP(hipMemsetAsync(ix->d_spos.p, 0, (ix->n_script + FS_MAX_WINDOW) * sizeof(fs_spos), s));
    hipLaunchKernelGGL(k_spos, dim3((uint32_t)((ix->n_script + 255) / 256)), dim3(256), 0, s, ix->d_stok.p,
                       (uint32_t)ix->n_script, L, ix->d_spos.p);
    // script window keys and their CSR buckets, all on the device
    DBuf<uint32_t> d_cursor, d_big, d_tmp;
    DBuf<uint32_t>& d_keys = ix->d_skeys;          // (kept: k_lsh_batch compares a window's keys with a script window's)
    FS_TRY(d_keys.reserve(W * H));
    FS_TRY(d_cursor.reserve((size_t)H * nb));
    FS_TRY(d_big.reserve((size_t)H * nb / kSmallBucket + (size_t)H * W / kSmallBucket + 2));
    FS_TRY(d_tmp.reserve((size_t)H * W));
    hipLaunchKernelGGL(k_keys, dim3((uint32_t)std::min<uint64_t>((W + 3) / 4, 4096)), dim3(256), 0,
                       s, L, ix->d_stok.p, (uint32_t)W, d_keys.p);
    const uint32_t gb = (uint32_t)std::min<uint64_t>((W * H + 255) / 256, 4096);
    uint32_t* n_big = d_big.p;                 // [0] = count, list behind it
    FS_HIP(hipMemsetAsync(n_big, 0, sizeof(uint32_t), s));
    hipLaunchKernelGGL(k_bucket_count, dim3(gb), dim3(256), 0, s, d_keys.p, (uint32_t)W, H, nb,
                       ix->d_boff.p);
    hipLaunchKernelGGL(k_bucket_offsets, dim3((uint32_t)H), dim3(256), 0, s, nb, ix->d_boff.p,
                       d_cursor.p);
    hipLaunchKernelGGL(k_bucket_fill, dim3(gb), dim3(256), 0, s, d_keys.p, (uint32_t)W, H, nb,
                       d_cursor.p, ix->d_bids.p);
    hipLaunchKernelGGL(k_bucket_sort, dim3((uint32_t)std::min<uint64_t>(((uint64_t)nb * H + 255) / 256, 4096)),
                       dim3(256), 0, s, (uint32_t)W, H, nb, ix->d_boff.p, ix->d_bids.p, d_big.p + 1, n_big);
    hipLaunchKernelGGL(k_bucket_sort_big, dim3(256), dim3(256), 0, s, (uint32_t)W, nb, ix->d_boff.p,
                       ix->d_bids.p, d_big.p + 1, n_big, d_tmp.p);
    FS_HIP(hipGetLastError());
    FS_HIP(hipStreamSynchronize(s));           // the scratch buffers die with this scope
  }
  FS_HIP(hipStreamSynchronize(s));
  ix->lsh_ready = true;
  return FS_OK;
}

int fs_launch_lsh_scan(fs_index* ix, const CorpusDev& c, uint64_t* qbm, uint32_t* qcnt,
                       uint32_t n_sub, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
  if (!n_sub) return FS_OK;
  const LshDev L = lsh_dev(ix);
  const int NW = (L.C + 63) >> 6;
  const size_t lds = (size_t)lsh_scan_bal_words(NW) * 8 + (size_t)256 * L.H * 4 + (256 + 16) * 4 +
                     ((size_t)lsh_scan_pref_words(L.H) + 1) * 4;
  FS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lsh_scan),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / (lds + 64)));
  const uint32_t blocks = std::min<uint32_t>(n_sub, ix->num_cu * per_cu);
  const uint64_t* gbm = nullptr;
  if ((L.share_flags & 32) && L.n >= 2 && L.n <= 12) {
    // the share rule by itself: the script windows behind every window's keys
    const uint32_t sblocks = std::min<uint32_t>(n_sub, ix->num_cu * (uint32_t)share_scan_occupancy(L.n));
    switch (L.n) {
#define FS_SHARE_CASE(NN) \
      case NN: hipExtLaunchKernelGGL(k_share_scan<NN>, dim3(sblocks), dim3(256), 0, s, e0, e1, 0u, c, L, qbm, qcnt, n_sub); break;
      FS_SHARE_CASE(2) FS_SHARE_CASE(3) FS_SHARE_CASE(4) FS_SHARE_CASE(5) FS_SHARE_CASE(6) FS_SHARE_CASE(7)
      FS_SHARE_CASE(8) FS_SHARE_CASE(9) FS_SHARE_CASE(10) FS_SHARE_CASE(11) FS_SHARE_CASE(12)
#undef FS_SHARE_CASE
      default: break;
    }
    FS_HIP(hipGetLastError());
    return FS_OK;
  }
  if ((L.share_flags & 1) && L.n >= 2 && L.n <= 12) {
    // the share rule's gate first: the windows that need keys at all
    FS_TRY(ix->cur->w_gate.reserve((size_t)n_sub * 4));
    const uint32_t gblocks = std::min<uint32_t>(n_sub, ix->num_cu * 4);
    uint64_t* g = ix->cur->w_gate.p;
    switch (L.n) {
#define FS_SHARE_CASE(NN) \
      case NN: hipExtLaunchKernelGGL(k_share_gate<NN>, dim3(gblocks), dim3(256), 0, s, e0, nullptr, 0u, c, L, g, n_sub); break;
      FS_SHARE_CASE(2) FS_SHARE_CASE(3) FS_SHARE_CASE(4) FS_SHARE_CASE(5) FS_SHARE_CASE(6) FS_SHARE_CASE(7)
      FS_SHARE_CASE(8) FS_SHARE_CASE(9) FS_SHARE_CASE(10) FS_SHARE_CASE(11) FS_SHARE_CASE(12)
#undef FS_SHARE_CASE
      default: break;
    }
    FS_HIP(hipGetLastError());
    if (ix->prof.on) fs_prof_mark(ix, s, "k_share_gate");
    e0 = nullptr;
    gbm = g;
  }
  hipExtLaunchKernelGGL(k_lsh_scan, dim3(blocks), dim3(256), (uint32_t)lds, s, e0, e1, 0u, c, L, gbm, qbm,
                        qcnt, n_sub);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

int fs_launch_selflev(fs_index* ix, fs_corpus* c, hipStream_t s) {
  const uint32_t W = (uint32_t)ix->n_windows;
  FS_TRY(c->d_selflev.reserve(W + 1));
  if (W) {
    hipLaunchKernelGGL(k_selflev, dim3(std::min<uint32_t>((W + 3) / 4, 4096)), dim3(256), 0, s,
                       ix->gram_dev(), c->dev(), W, c->d_selflev.p);
    FS_HIP(hipGetLastError());
  }
  return FS_OK;
}

// the strings of the batch's table against every script n-gram, once per string table; the
// status block of lane 0 collects string errors (bad_string, lev_overflow) for the caller
int fs_launch_lsh_gramtab(fs_index* ix, fs_corpus* c, hipStream_t s) {
  if (!ix->n_grams) return FS_OK;
  LshDev L = lsh_dev(ix);
  if (c->selflev_ready) L.selflev = c->d_selflev.p;
  FS_TRY(c->d_gramtab_best.reserve(4 * (size_t)ix->n_grams));
  FS_TRY(c->d_gramtab_cnt.reserve(ix->n_grams));
  const uint32_t blocks = (ix->n_grams + 3) / 4;
  hipLaunchKernelGGL(k_lsh_gramtab, dim3(blocks > kNB ? kNB : blocks), dim3(256), 0, s, c->dev(), L,
                     ix->gram_dev(), c->d_gramtab_best.p, c->d_gramtab_cnt.p, ix->cur->d_status.p);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

namespace {
__global__ void k_comp_map(const uint32_t* __restrict__ tok, uint32_t n, const uint32_t* __restrict__ comp,
                           uint32_t n_vec, uint32_t* __restrict__ out) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t t = tok[i];
    out[i] = t < n_vec ? comp[t] : 0u;
  }
}
}  // namespace

// component ids of a batch's tokens (tables with near-synonyms), the scan's pad included
int fs_launch_comp_map(fs_index* ix, fs_corpus* c, hipStream_t s) {
  const uint64_t n = c->n_tok + fs_scan_pad_tokens();
  FS_TRY(c->d_ctok.reserve(n));
  hipLaunchKernelGGL(k_comp_map, dim3(2048), dim3(256), 0, s, (const uint32_t*)c->d_tok.p, (uint32_t)n,
                     (const uint32_t*)ix->d_comp.p, (uint32_t)ix->n_vec, c->d_ctok.p);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

// The wildcard-key filter that stands in front of the LSH work of a search of `c` (nullptr: none),
// and the ids its keys are made of (nullptr: the vector ids).
void fs_lsh_wild_of(const fs_index* ix, const fs_corpus* c, const uint32_t** wild, int* log2_wild,
                    const uint32_t** wild_tok) {
  *wild = nullptr; *log2_wild = 0; *wild_tok = nullptr;
  // tables with near-synonyms: the wildcard keys over component ids (no one-slot map: a
  // neighbour may differ from the window in every vector id)
  if (fs_lsh_prefilter_mode(ix, c) == 2 && ix->sw.lsh_wild) {
    *wild = ix->d_wildc.p;
    *log2_wild = ix->log2_wildc;
    *wild_tok = c->d_ctok.p;
  }
  // at most one slot may differ and no OOV id anywhere: the wildcard-key filter applies
  if (ix->sw.lsh_wild && ix->d_wild.p && !c->has_oov && !ix->script_oov &&
      (int)ix->cfg.window_size - ix->lsh_m_min == 1) {
    *wild = ix->d_wild.p;
    *log2_wild = ix->log2_wild;
    *wild_tok = nullptr;
  }
}

// near: the candidates come from k_near_sift's lists (the wildcard filter applied):
// k_lsh_sift2 numbers them and takes the second stage, instead of k_lsh_sift over k_expand's list
int fs_launch_lsh_verify(fs_index* ix, fs_corpus* c, uint32_t ccap, hipStream_t s, const fs_near_lists* near) {
  LshDev L = lsh_dev(ix);
  if (c->selflev_ready && !c->has_str) L.selflev = c->d_selflev.p;
  // the wildcard-key filter of this search, if any (fs_lsh_wild_of)
  fs_lsh_wild_of(ix, c, &L.wild, &L.log2_wild, &L.wild_tok);
  // every neighbour within the threshold equals the window in all slots but one (by vector
  // ids, or by component ids): the exact map enumerates them, no bucket is walked (k_lsh_batch)
  if (ix->sw.lsh_emap && L.wild) {
    if (L.wild_tok && ix->d_emapc.p && ix->log2_emapc) {
      L.emap = reinterpret_cast<const uint2*>(ix->d_emapc.p); L.log2_emap = ix->log2_emapc; L.emap_comp = 1;
    } else if (!L.wild_tok && ix->d_emap.p && ix->log2_emap) {
      L.emap = reinterpret_cast<const uint2*>(ix->d_emap.p); L.log2_emap = ix->log2_emap; L.emap_comp = 0;
    }
  }
  if (L.wild && !L.wild_tok) {
    // A one-slot neighbour has cosine (n - 1 + c) / n with c the cosine of the two differing
    // vectors: within the threshold iff c > 1 - n * thr.  At n = 8 (c > 0.2) nearly every such
    // window ends in k_lsh_sift; at n = 10 (c > 0) half of them are real neighbours and stay
    // pending, the other half still ends there.
    if (ix->sw.lsh_wmap && ix->d_wmap.p) {
      L.wmap = reinterpret_cast<const uint2*>(ix->d_wmap.p);
      L.log2_wmap = ix->log2_wmap;
    }
  }
  fs_status* st = ix->cur->d_status.p;
  const NSrc nc{&st->n_cands, 1, ccap, 0};
  // per-n-gram records of this string table (k_lsh_gramtab, fs_corpus_update_end)
  const unsigned long long* tab_best = nullptr;
  const uint32_t* tab_cnt = nullptr;
  if (c->gramtab_ready && !c->has_str && !c->has_oov) {
    tab_best = c->d_gramtab_best.p;
    tab_cnt = c->d_gramtab_cnt.p;
  }
  FS_TRY(ix->cur->w_pend.reserve(ccap));
  auto sift = L.n <= 8 ? (L.wmap ? k_lsh_sift<8, true, 0> : k_lsh_sift<8, false, 0>)
                       : (L.wmap ? k_lsh_sift<FS_MAX_WINDOW, true, 0> : k_lsh_sift<FS_MAX_WINDOW, false, 0>);
  switch (L.n) {            // the common window sizes with their size at compile time
    case 6: sift = L.wmap ? k_lsh_sift<8, true, 6> : k_lsh_sift<8, false, 6>; break;
    case 8: sift = L.wmap ? k_lsh_sift<8, true, 8> : k_lsh_sift<8, false, 8>; break;
    case 10: sift = L.wmap ? k_lsh_sift<FS_MAX_WINDOW, true, 10> : k_lsh_sift<FS_MAX_WINDOW, false, 10>; break;
    default: break;
  }
  // one resident set of workgroups each (both kernels loop over their work and are bound by the
  // latency of dependent loads: a second, partial round of workgroups costs a whole round's time)
  auto resident = [&](const void* kern) {
    static std::mutex mu;
    static std::vector<std::pair<const void*, int>> seen;      // workgroups per CU, asked once per kernel
    int per_cu = 0;
    {
      std::lock_guard<std::mutex> lock(mu);
      for (const auto& e : seen)
        if (e.first == kern) per_cu = e.second;
      if (!per_cu) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, 0) != hipSuccess || per_cu < 1)
          per_cu = -1;
        seen.push_back({kern, per_cu});
      }
    }
    return per_cu < 1 ? (uint32_t)kNB : std::min<uint32_t>(kNB, ix->num_cu * (uint32_t)per_cu);
  };
  static const bool full_grid = getenv("FS_LSH_FULL_GRID") && atoi(getenv("FS_LSH_FULL_GRID")) != 0;
  if (near) {
    auto sift2 = L.n <= 8 ? (L.wmap ? k_lsh_sift2<8, true> : k_lsh_sift2<8, false>)
                          : (L.wmap ? k_lsh_sift2<FS_MAX_WINDOW, true> : k_lsh_sift2<FS_MAX_WINDOW, false>);
    const uint32_t blocks = full_grid ? kNB : resident(reinterpret_cast<const void*>(sift2));
    hipLaunchKernelGGL(sift2, dim3(blocks), dim3(256), 0, s, c->dev(), L, ix->gram_dev(),
                       near->slist, near->caps, near->scount, ix->cur->w_bsum.p,
                       ix->cur->w_cpos.p, ccap, ix->cur->w_cg.p, ix->cur->w_cw.p, ix->cur->w_cbest.p,
                       ix->cur->w_bsum.p + kNB, tab_best, tab_cnt, ix->cur->w_pend.p, st);
    if (ix->prof.on) fs_prof_mark(ix, s, "k_lsh_sift2");
  } else {
  const uint32_t sift_blocks = full_grid ? kNB : resident(reinterpret_cast<const void*>(sift));
  hipLaunchKernelGGL(sift, dim3(sift_blocks), dim3(256), 0, s, c->dev(), L, ix->gram_dev(),
                     ix->cur->w_cpos.p, nc, ix->cur->w_cg.p, ix->cur->w_cw.p,
                     ix->cur->w_cbest.p, ix->cur->w_bsum.p + kNB, tab_best, tab_cnt, ix->cur->w_pend.p,
                     &st->lsh_pending);
    if (ix->prof.on) fs_prof_mark(ix, s, "k_lsh_sift");
  }
  // the kept matches' Levenshtein distances a lane per match (k_lsh_lev) where the script
  // windows' bit planes and the string table's records exist; else a wave per match inside
  // k_lsh_verify
  // (a launch of its own costs ~20 us whatever it finds to do -- the chain of loads in front of a
  // pair and the pair itself, 4 us of one lane's time: worth it from some thousands of pending
  // windows on, which the lane's last search tells; FS_LSH_LEV_LANE=2: always)
  const bool defer = ix->sw.lsh_lev_lane && ix->sw.str_fast && ix->strfast_ok && c->strrec_ready &&
                     L.nn <= kLevMaxN && (ix->cur->pend_hint >= (uint32_t)ix->sw.lsh_defer_min || ix->sw.lsh_lev_lane == 2);
  if (defer) {
    FS_TRY(ix->cur->w_mcnt.reserve(ccap));
    FS_TRY(ix->cur->w_mtop_s.reserve((size_t)ccap * L.nn));
    FS_TRY(ix->cur->w_mtop_d.reserve((size_t)ccap * L.nn));
  }
  // the pending windows eight at a time per wave (k_lsh_batch) where the kept matches' Levenshtein
  // distances are k_lsh_lev's anyway; a wave per window (k_lsh_verify) otherwise.  Where the script
  // n-grams one slot away can be enumerated (L.emap), k_lsh_pkeys + k_lsh_enum take the windows
  // first and k_lsh_batch only what they leave.
  typedef void (*BatchFn)(CorpusDev, LshDev, GramIndexDev, const uint32_t*, uint32_t, uint32_t*, uint32_t*, uint32_t*,
                          fs_status*, const uint32_t*, uint32_t*, uint32_t*, double*, const uint32_t*, const uint32_t*);
  typedef void (*PkeysFn)(CorpusDev, LshDev, const uint32_t*, uint32_t, const uint32_t*, const fs_status*, uint32_t*, uint32_t*);
  typedef void (*EnumFn)(CorpusDev, LshDev, GramIndexDev, const uint32_t*, uint32_t, uint32_t*, uint32_t*, uint32_t*,
                         fs_status*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t*, uint32_t*, double*,
                         uint32_t*, uint32_t*);
  BatchFn batch = nullptr;
  PkeysFn pkeys = nullptr;
  EnumFn enumk = nullptr;
  if (defer && ix->sw.lsh_batch && L.H <= kBatchH && L.nn <= 48 && !L.serial_neighbours && !(L.diag & 0xFFFF))
    switch (L.n) {
#define FS_B(N) case N: batch = k_lsh_batch<N>; pkeys = k_lsh_pkeys<N>; enumk = k_lsh_enum<N>; break
      FS_B(6); FS_B(7); FS_B(8); FS_B(9); FS_B(10); FS_B(12);
#undef FS_B
      default: break;
    }
  if (batch) {
    const bool enumerate = L.emap && L.skeys && L.spos && L.gtab && L.nn <= kEnumNN;
    const uint32_t* left = nullptr;
    const uint32_t* n_left = nullptr;
    if (enumerate) {
      FS_TRY(ix->cur->w_pkeys.reserve((size_t)ccap * kBatchH));
      FS_TRY(ix->cur->w_pwork.reserve(ccap));
      FS_TRY(ix->cur->w_left.reserve(ccap));
      const uint32_t kb = full_grid ? kNB : resident(reinterpret_cast<const void*>(pkeys));
      hipLaunchKernelGGL(pkeys, dim3(kb), dim3(256), 0, s, c->dev(), L, ix->cur->w_cpos.p, ccap, ix->cur->w_pend.p,
                         st, ix->cur->w_pkeys.p, ix->cur->w_pwork.p);
      if (ix->prof.on) fs_prof_mark(ix, s, "k_lsh_pkeys");
      const uint32_t eb = full_grid ? kNB : resident(reinterpret_cast<const void*>(enumk));
      hipLaunchKernelGGL(enumk, dim3(eb), dim3(256), 0, s, c->dev(), L, ix->gram_dev(), ix->cur->w_cpos.p, ccap,
                         ix->cur->w_cg.p, ix->cur->w_cw.p, ix->cur->w_bsum.p + kNB, st, ix->cur->w_pend.p,
                         ix->cur->w_pkeys.p, ix->cur->w_pwork.p, ix->cur->w_mcnt.p, ix->cur->w_mtop_s.p,
                         ix->cur->w_mtop_d.p, ix->cur->w_left.p, &st->n_hits);
      if (ix->prof.on) fs_prof_mark(ix, s, "k_lsh_enum");
      left = ix->cur->w_left.p;
      n_left = &st->n_hits;
    }
    const uint32_t blocks = full_grid ? kNB : resident(reinterpret_cast<const void*>(batch));
    hipLaunchKernelGGL(batch, dim3(blocks), dim3(256), 0, s, c->dev(), L, ix->gram_dev(),
                       ix->cur->w_cpos.p, ccap, ix->cur->w_cg.p, ix->cur->w_cw.p, ix->cur->w_bsum.p + kNB, st,
                       ix->cur->w_pend.p, ix->cur->w_mcnt.p, ix->cur->w_mtop_s.p, ix->cur->w_mtop_d.p, left, n_left);
    if (ix->prof.on) fs_prof_mark(ix, s, "k_lsh_batch");
  } else {
  auto verify = defer ? k_lsh_verify<true> : k_lsh_verify<false>;
  const uint32_t verify_blocks = full_grid ? kNB : resident(reinterpret_cast<const void*>(verify));
  hipLaunchKernelGGL(verify, dim3(verify_blocks), dim3(256), 0, s, c->dev(), L, ix->gram_dev(),
                     ix->cur->w_cpos.p, nc, ix->cur->w_cg.p, ix->cur->w_cw.p, ix->cur->w_cbest.p,
                     ix->cur->w_bsum.p + kNB, st, ix->cur->w_pend.p, ix->cur->w_mcnt.p,
                     ix->cur->w_mtop_s.p, ix->cur->w_mtop_d.p);
  if (ix->prof.on) fs_prof_mark(ix, s, "k_lsh_verify");
  }
  if (defer) {
    const StrFast F{ix->d_pat.p, ix->d_clsmap.p, ix->n_cls, ix->str_punct, c->d_strrec.p};
    const uint32_t lev_blocks = full_grid ? kNB : resident(reinterpret_cast<const void*>(k_lsh_lev));
    hipLaunchKernelGGL(k_lsh_lev, dim3(lev_blocks), dim3(256), 0, s, c->dev(), L, ix->gram_dev(), F,
                       ix->cur->w_cpos.p, ccap, ix->cur->w_pend.p, ix->cur->w_mcnt.p,
                       ix->cur->w_mtop_s.p, ix->cur->w_mtop_d.p, ix->cur->w_cg.p, ix->cur->w_cbest.p, st);
    if (ix->prof.on) fs_prof_mark(ix, s, "k_lsh_lev");
  }
  FS_HIP(hipGetLastError());
  return FS_OK;
}
