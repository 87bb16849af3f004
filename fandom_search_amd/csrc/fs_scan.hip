// fs_scan.hip -- the kernels that read the fan token ids (the HBM-bound hot loop).
//
// Replaces, for corpora whose vector table admits the exact-n-gram proof
// (DESIGN.md), the per-window engine.neighbours() call of the reference
// (/root/reference/search.py:176-178): every fan window's n vector ids are
// tested against the script's n-gram set.
//
// k_scan_rows (n = 2..8; the STR variant for batches whose fan tokens carry string ids of
// their own) is the whole search in one kernel, token ids in, output records out; the comment in front of it and
// fs_ranges.h describe it.  Per wave and 512-token sub-tile its scan does
//   * two coalesced global_load_dwordx4 per lane: tokens [8L, 8L+8), requested a pair of
//     sub-tiles ahead
//   * the n-1 halo tokens from lane L+1 by ds_bpermute (no second global read)
//   * per token one ds_read_b32 of the LDS-resident filter of script K-grams; a window is
//     a candidate when its n-K+1 K-grams are all script K-grams (n <= 3: the three-bit
//     Bloom test of the whole n-gram); a lane's eight answers are one byte
// and candidates are verified, turned into records and put into place by the same wave.
//
// ONE chained fallback remains for window sizes k_scan_rows does not take, scripts whose
// alphabet is too large for the per-hit Levenshtein form, and a given-up in-launch wait: k_scan8 (n = 2..8: same loop, Bloom test, candidate
// records or byte bitmap out) or k_scan_simple (any n); counting, expansion and verification
// are then left to fs_post.hip.  k_scan_near is the integer prefilter of the LSH pipeline
// (script 3-grams, at most one differing slot).
// No global atomics on the data path; a record's bytes and place are functions of token
// positions, so the result is deterministic.  Algorithmic HBM traffic: 4 B read per
// token + 32 B per record written.
#include "fs_internal.h"
#include "fs_device.h"
#include "fs_ranges.h"

#include <hip/hip_ext.h>
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <vector>

namespace {

constexpr int kWave = 64;
constexpr int kTokPerLane = 4;
constexpr int kSubTile = kWave * kTokPerLane;   // 256 tokens per bitmap word

// The Bloom filter into LDS: four 16-byte pieces per thread requested together (64 KB,
// 1024 threads: one batch instead of four dependent round trips in front of the scan).
__device__ __forceinline__ void copy_filter_to_lds(const uint32_t* __restrict__ filter,
                                                   uint32_t* s_filter, int log2_words) {
  const uint32_t vecs = (1u << log2_words) / 4;
  const uint4* src = reinterpret_cast<const uint4*>(filter);
  uint4* dst = reinterpret_cast<uint4*>(s_filter);
  const uint32_t nt = blockDim.x;
  uint32_t i = threadIdx.x;
  for (; i + 3 * nt < vecs; i += 4 * nt) {
    const uint4 q0 = src[i], q1 = src[i + nt], q2 = src[i + 2 * nt], q3 = src[i + 3 * nt];
    dst[i] = q0; dst[i + nt] = q1; dst[i + 2 * nt] = q2; dst[i + 3 * nt] = q3;
  }
  for (; i < vecs; i += nt) dst[i] = src[i];
}

// b[j]: wave mask of "window 4*lane + j of sub-tile `word` is filter-positive"
__device__ __forceinline__ void store_ballots(const uint64_t* b, int lane, uint32_t word,
                                              uint32_t n_bm_words, uint64_t* __restrict__ qbm,
                                              uint32_t* __restrict__ qcnt) {
  if (word < n_bm_words && lane < 4) {
    const uint64_t mine = lane == 0 ? b[0] : lane == 1 ? b[1] : lane == 2 ? b[2] : b[3];
    qbm[(size_t)word * 4 + lane] = mine;
    if (lane == 0)
      qcnt[word] = __popcll(b[0]) + __popcll(b[1]) + __popcll(b[2]) + __popcll(b[3]);
  }
}

// ---- integer prefilter of the LSH pipeline -------------------------------------------
// Vector tables whose exact-n-gram proof fails by one slot only (n = 8, 10 on
// the synthetic table): a fan window can have a neighbour within the threshold only if it
// equals a script window in at least n-1 slots (lsh_m_min, from the table's c_max).  Such a
// window keeps every 3-gram that does not contain the odd slot, so among its n-2 3-gram
// tests (one bit per script 3-gram, as in window_flags_sub) all failures lie within three
// consecutive positions.  Sound (no window with a neighbour is lost), integer only, at
// scan rate; the LSH keys, buckets and distances are then computed for the flagged
// windows alone (k_lsh_verify), instead of 5 KB of projection rows for every window.
// Same bitmap as k_scan / k_lsh_scan: four ballot words + one count per 256 tokens.
// (K = 3 also at n = 6, where the four tests pass any window whose first or last 3-gram is a
// script 3-gram: 29 % of the windows of the Zipf-distributed benchmark text over component
// ids.  2-grams measured worse -- 48 %: half of all fan 2-grams occur in the script.)
constexpr int fs_near_k(int n) { return n <= 5 ? 2 : 3; }
template <int N, bool TAIL>
__device__ __forceinline__ void window_ballots_near(const uint32_t* m, const uint32_t* s_filter,
                                                    int word_shift, uint32_t p0, uint32_t n_tok,
                                                    uint64_t* b, const uint32_t* s_keys = nullptr) {
  constexpr int K = fs_near_k(N), T = N - K + 1;  // K-gram tests per window
  constexpr int NB = kTokPerLane + N - K;       // 3-gram positions of the lane
  uint32_t x = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) x ^= fs_rotl(m[k], fs_rot_of(K - 1 - k));
  uint32_t bits = 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    if (j) x = (uint32_t)__builtin_amdgcn_bitop3_b32(fs_rotl(x, 7), fs_rotl(m[j - 1], fs_rot_of(K)),
                                                      m[j - 1 + K], 0x96);
    const uint32_t word = s_filter[x >> word_shift];
    bits = __builtin_amdgcn_alignbit(word >> (x & 31), bits, 1);
  }
  bits >>= 32 - NB;                             // 3-gram j at bit j
  uint32_t flags = 0;
#pragma unroll
  for (int j = 0; j < kTokPerLane; ++j) {
    const uint32_t z = ~(bits >> j) & ((1u << T) - 1);       // failed tests of window j
    // none, or all within three consecutive positions
    bool hit = z == 0 || (31 - __clz((int)z)) - (__ffs((int)z) - 1) < K;
    if constexpr (N == 6) {
      // n = 6: "only the last 3-gram is a script 3-gram" (or only the first) passes the rule
      // above and is what most windows that pass look like.  It leaves ONE slot that may differ
      // (2, or 3), so the window's wildcard key for that slot must be a script window's: a
      // second filter in LDS, over those two keys of every script window (fs_hash.h).
      if (s_keys) {
        const bool only_last = z == 0x7u, only_first = z == 0xEu;
        uint32_t fold = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) fold ^= fs_rotl(m[j + k], fs_rot_of(5 - k));
        const uint32_t term = only_last ? fs_rotl(m[j + 2], fs_rot_of(3)) : fs_rotl(m[j + 3], fs_rot_of(2));
        const uint32_t h = fs_wild_key(fold, term, only_last ? 2 : 3);
        const uint32_t word = s_keys[fs_bloom_word(h, FS_NEAR6_LOG2_WORDS)];
        if (only_last | only_first) hit = fs_bloom_test(word, h) != 0;
      }
    }
    flags |= (hit ? 1u : 0u) << j;
  }
  if (TAIL) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if ((uint64_t)p0 + j + N > n_tok) flags &= ~(1u << j);
  }
  b[0] = __ballot(flags & 1u); b[1] = __ballot(flags & 2u);
  b[2] = __ballot(flags & 4u); b[3] = __ballot(flags & 8u);
}

template <int N, bool NT>
__global__ __launch_bounds__(1024) void k_scan_near(const uint32_t* __restrict__ tok, uint32_t n_tok,
                                                    const uint32_t* __restrict__ filter,
                                                    int log2_words, uint64_t* __restrict__ qbm,
                                                    uint32_t* __restrict__ qcnt,
                                                    uint32_t n_bm_words, uint32_t n_tiles,
                                                    const uint32_t* __restrict__ keys6) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_filter[];
  copy_filter_to_lds(filter, s_filter, log2_words);
  // (n = 6 over component ids: the filter of the middle slots' wildcard keys behind it)
  const uint32_t* s_keys = nullptr;
  if (N == 6 && keys6) {
    copy_filter_to_lds(keys6, s_filter + (1u << log2_words), FS_NEAR6_LOG2_WORDS);
    s_keys = s_filter + (1u << log2_words);
  }
  __syncthreads();
  constexpr int U = 2;
  constexpr int HALO = N - 1;
  constexpr int NV = (HALO + 3) / 4;            // neighbour vectors needed
  const int word_shift = 32 - log2_words;
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
    const uint32_t base = tile * (uint32_t)(kSubTile * U);
    uint32_t a[U][4 + 4 * NV];
    uint4 v[U + 1];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint4* src = reinterpret_cast<const uint4*>(tok + base + u * kSubTile + 4 * lane);
      if constexpr (NT) {
        v[u].x = __builtin_nontemporal_load(&src->x); v[u].y = __builtin_nontemporal_load(&src->y);
        v[u].z = __builtin_nontemporal_load(&src->z); v[u].w = __builtin_nontemporal_load(&src->w);
      } else {
        v[u] = *src;
      }
    }
    // first vectors of the next tile, in lanes 0..3 (the buffer is padded)
    v[U] = *reinterpret_cast<const uint4*>(tok + base + U * kSubTile + 4 * (lane & 3));
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a[u][0] = v[u].x; a[u][1] = v[u].y; a[u][2] = v[u].z; a[u][3] = v[u].w;
#pragma unroll
      for (int d = 1; d <= NV; ++d) {
        const bool wrap = lane + d > 63;        // past lane 63: a vector of the next sub-tile
        const int srcl = (lane + d) & 63;
        const int need = (HALO - 4 * (d - 1)) < 4 ? (HALO - 4 * (d - 1)) : 4;
        // lanes 0 .. d-1 publish the next sub-tile's vector instead of their own
        const bool pub = lane < d;
        (void)wrap;
        a[u][4 * d + 0] = __shfl(pub ? v[u + 1].x : v[u].x, srcl);
        if (need > 1) a[u][4 * d + 1] = __shfl(pub ? v[u + 1].y : v[u].y, srcl);
        if (need > 2) a[u][4 * d + 2] = __shfl(pub ? v[u + 1].z : v[u].z, srcl);
        if (need > 3) a[u][4 * d + 3] = __shfl(pub ? v[u + 1].w : v[u].w, srcl);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int i = 0; i < 4 + HALO; ++i) a[u][i] = fs_premix(a[u][i]);
      const uint32_t p0 = base + u * kSubTile + 4 * lane;
      uint64_t b[4];
      const bool tail = base + (uint32_t)(kSubTile * U) + HALO > n_tok;   // wave-uniform
      if (tail) window_ballots_near<N, true>(a[u], s_filter, word_shift, p0, n_tok, b, s_keys);
      else window_ballots_near<N, false>(a[u], s_filter, word_shift, p0, n_tok, b, s_keys);
      store_ballots(b, lane, tile * U + u, n_bm_words, qbm, qcnt);
    }
  }
}

// ---- eight tokens per lane ---------------------------------------------------
// The same scan with lane L owning tokens [8L, 8L+8) of a 512-token sub-tile: the
// n-1 halo tokens are shuffled in once per 8 windows instead of once per 4, 13
// premixes serve 8 windows and the full fold is paid once per 8.  A lane's eight
// answers are one byte, so the sub-tile's bitmap is written as 64 consecutive
// bytes in natural order (bit p of the 512-bit string <-> window 512 i + p): no
// ballots, no per-lane word selection.  For n <= 9 (the halo fits in the next
// lane's eight tokens).
//
// VALU work per window (the kernel is VALU-issue bound): premix 1; slide
//   x' = rotl(x, 7) ^ rotl(m_out, 7n) ^ m_in            alignbit, alignbit, xor3
// Bloom test: word address 1 (SDWA word select, 64 KB filter) or 2, three shifts of
// the filter word (one through an SDWA byte select, one needs h >> 13 first), and3,
// and one alignbit that shifts the answer into the lane's flag byte.
__device__ __forceinline__ uint32_t fs_shr_by_byte1(uint32_t word, uint32_t h) {
  uint32_t r;     // word >> (bits 8..12 of h)
  asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"
      : "=v"(r) : "v"(h), "v"(word));
  return r;
}
__device__ __forceinline__ uint32_t fs_word_offset14(uint32_t h, uint32_t mask_fffc) {
  uint32_t r;     // byte offset of filter word h >> 18:  (h >> 16) & 0xFFFC
  asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"
      : "=v"(r) : "v"(h), "v"(mask_fffc));
  return r;
}

template <int N, bool TAIL, bool LW14>
__device__ __forceinline__ uint32_t window_flags8(const uint32_t* m, const uint32_t* s_filter,
                                                  int word_shift, uint32_t mask_fffc,
                                                  uint32_t p0, uint32_t n_tok) {
  uint32_t x = 0;
#pragma unroll
  for (int k = 0; k < N; ++k) x ^= fs_rotl(m[k], fs_rot_of(N - 1 - k));
  uint32_t flags = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j) x = (uint32_t)__builtin_amdgcn_bitop3_b32(fs_rotl(x, 7), fs_rotl(m[j - 1], fs_rot_of(N)),
                                                      m[j - 1 + N], 0x96);   // three-way xor
    uint32_t word;
    if constexpr (LW14)
      word = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(s_filter) +
                                                fs_word_offset14(x, mask_fffc));
    else
      word = s_filter[x >> word_shift];
    const uint32_t t = (word >> (x & 31)) & fs_shr_by_byte1(word, x) & (word >> ((x >> 13) & 31));
    flags = __builtin_amdgcn_alignbit(t, flags, 1);       // bit 0 of t enters at bit 31
  }
  flags >>= 24;                                           // window j at bit j
  if (TAIL) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if ((uint64_t)p0 + j + N > n_tok) flags &= ~(1u << j);
  }
  return flags;
}

// The same for k_scan_rows' sub-shingle filter (fs_hash.h): the lane tests the K-grams at
// its 8 + N - K positions (one bit each), window w is a candidate when the K-grams
// w .. w + N - K are all script K-grams.  m[0 .. 8 + N - 1): premixed ids.  VALU per
// position: slide 1 (shift-add: the polynomial hash), word address 1 (SDWA), shift by the
// hash's byte 1 (SDWA) 1, funnel shift 1; plus 2 (N - K) / 8 per window for the runs.
// mask_words: ((1 << log2_words) - 1) << 2, in a register.
__device__ __forceinline__ uint32_t fs_sub_word_offset(uint32_t h, uint32_t mask_words) {
  uint32_t r;     // byte offset of filter word (h >> 18) & (words - 1):  (h >> 16) & mask_words
  asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"
      : "=v"(r) : "v"(h), "v"(mask_words));
  return r;
}

template <int N, int K, bool TAIL>
__device__ __forceinline__ uint32_t window_flags_sub(const uint32_t* m, const uint32_t* s_filter,
                                                     uint32_t mask_words, uint32_t p0, uint32_t n_tok) {
  constexpr int NB = 8 + N - K;                 // K-gram positions of the lane
  constexpr int S = fs_sub_shift(K);
  // three passes, so that the filter words are requested together and waited for once:
  // hashes and word addresses, the LDS reads, the bits
  uint32_t x[NB], w[NB];
  x[0] = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) x[0] = (x[0] << S) + m[k];
#pragma unroll
  for (int j = 1; j < NB; ++j) x[j] = (x[j - 1] << S) + m[j - 1 + K];   // the id K places back has left the 32 bits
  // (the filter sits at LDS address 0 -- k_scan_rows has no static LDS, checked at launch --
  // so a word's byte offset is its address: no add of a base the compiler cannot fold)
  typedef const __attribute__((address_space(3))) uint32_t lds_word;
  (void)s_filter;
#pragma unroll
  for (int j = 0; j < NB; ++j) w[j] = fs_sub_word_offset(x[j], mask_words);
#pragma unroll
  for (int j = 0; j < NB; ++j) w[j] = *reinterpret_cast<lds_word*>((uintptr_t)w[j]);
  asm volatile("" : "+v"(w[0]), "+v"(w[NB - 1]));      // every read is out before the first is used
  // (the shifts apart from the funnel chain: an SDWA result needs a wait state before its
  // next use, which the other shifts fill)
#pragma unroll
  for (int j = 0; j < NB; ++j) w[j] = fs_shr_by_byte1(w[j], x[j]);
  uint32_t bits = 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) bits = __builtin_amdgcn_alignbit(w[j], bits, 1);   // bit 0 enters at bit 31
  bits >>= 32 - NB;                             // K-gram j at bit j
  uint32_t flags = bits;
#pragma unroll
  for (int d = 1; d <= N - K; ++d) flags &= bits >> d;
  flags &= 0xFFu;
  if (TAIL) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if ((uint64_t)p0 + j + N > n_tok) flags &= ~(1u << j);
  }
  return flags;
}

// Work assignment: the kernels that follow cut the sub-tiles into kNB chunks and need the
// candidate count of every chunk.  Workgroup b takes the kChunksPerBlock consecutive
// chunks [4b, 4b+4); inside a chunk each of four waves scans a contiguous quarter of the
// sub-tiles (a "wave range", 10 KB of ids on C2).  The waves add their counts per chunk in
// LDS and the workgroup writes the four sums: no counting kernel, no global atomics.
// Workgroup 0 also clears the search's status block.
//
// Direct path (recs != nullptr): no bitmap leaves the kernel.  A lane whose eight windows
// hold candidates appends one 8-byte record {(position / 8) << 8 | flag byte, rank of its
// first candidate inside the wave range} to the wave range's list; ranks come from a wave
// prefix sum of the flag popcounts (six DPP adds, in place of the wave sum) and the record
// slot from a ballot.  k_verify_direct reads the lists: block = chunk, wave = wave range.
constexpr int kScanBlocks = 512;
constexpr int kChunksPerBlock = fsdev::kNB / kScanBlocks;
static_assert(kChunksPerBlock * 4 == 16, "four waves per chunk, sixteen waves per workgroup");

template <int N, bool NT, bool LW14>
__global__ __launch_bounds__(1024) void k_scan8(const uint32_t* __restrict__ tok, uint32_t n_tok,
                                                const uint32_t* __restrict__ filter,
                                                int log2_words, uint64_t* __restrict__ qbm,
                                                uint32_t* __restrict__ qcnt, uint32_t n_sub,
                                                uint32_t chunk, uint32_t* __restrict__ bsum,
                                                fs_status* __restrict__ zero,
                                                uint2* __restrict__ recs, uint2* __restrict__ info,
                                                uint32_t capw) {
  static_assert(N <= 9, "halo must fit in the next lane's eight tokens");
  extern __shared__ __attribute__((aligned(16))) uint32_t s_filter[];
  __shared__ uint32_t s_csum[kChunksPerBlock];
  if (zero && blockIdx.x == 0 && threadIdx.x == 0) {
    zero->n_cands = 0; zero->n_hits = 0; zero->n_matches = 0; zero->n_rows = 0;
    zero->max_recs = 0; zero->lev_overflow = 0; zero->bad_string = 0; zero->max_rows = 0;
    zero->lsh_pending = 0;
  }
  const int lane = threadIdx.x & 63;
  const uint32_t wave = threadIdx.x >> 6;
  const uint32_t kc = wave >> 2, q = wave & 3;                 // chunk of the workgroup, quarter
  const uint32_t range_id = (blockIdx.x * kChunksPerBlock + kc) * 4 + q;
  const uint32_t per = (chunk + 3) >> 2;                       // sub-tiles per wave range
  const uint64_t chunk_first = ((uint64_t)blockIdx.x * kChunksPerBlock + kc) * chunk;
  uint32_t j0 = q * per, j1 = j0 + per;
  if (j1 > chunk) j1 = chunk;
  if (j0 > j1) j0 = j1;
  if (chunk_first + j1 > n_sub) j1 = chunk_first + j0 < n_sub ? (uint32_t)(n_sub - chunk_first) : j0;
  if ((uint64_t)blockIdx.x * kChunksPerBlock * chunk >= n_sub) {   // nothing to scan: empty chunks
    if (bsum && threadIdx.x < kChunksPerBlock) bsum[blockIdx.x * kChunksPerBlock + threadIdx.x] = 0;
    if (info && lane == 0) info[range_id] = make_uint2(0, 0);
    return;
  }
  if (threadIdx.x < kChunksPerBlock) s_csum[threadIdx.x] = 0;
  {
    copy_filter_to_lds(filter, s_filter, log2_words);
  }
  __syncthreads();
  constexpr int HALO = N - 1;
  constexpr int SUB = 512;
  const int word_shift = 32 - log2_words;
  const int src = (lane + 1) & 63;
  uint32_t mask_fffc = 0xFFFCu;
  asm volatile("" : "+v"(mask_fffc));          // keep the SDWA operand in a register
  uint8_t* bm_bytes = reinterpret_cast<uint8_t*>(qbm);
  uint2* list = recs ? recs + (size_t)range_id * capw : nullptr;
  uint32_t cand_run = 0, rec_run = 0;          // candidates / records of this wave range so far

  for (uint32_t j = j0; j < j1; ++j) {
    const uint32_t sub = (uint32_t)chunk_first + j;
    const uint32_t base = sub * (uint32_t)SUB;
    uint4 v[2][2];
    {
      const uint4* p = reinterpret_cast<const uint4*>(tok + base + 8 * lane);
      if constexpr (NT) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          v[0][h].x = __builtin_nontemporal_load(&p[h].x); v[0][h].y = __builtin_nontemporal_load(&p[h].y);
          v[0][h].z = __builtin_nontemporal_load(&p[h].z); v[0][h].w = __builtin_nontemporal_load(&p[h].w);
        }
      } else {
        v[0][0] = p[0]; v[0][1] = p[1];
      }
    }
    {  // first eight tokens of the next sub-tile, needed by lane 63 only (the buffer is padded)
      const uint4* p = reinterpret_cast<const uint4*>(tok + base + SUB);
      v[1][0] = p[0]; v[1][1] = p[1];
    }
    uint32_t a[16];
    a[0] = v[0][0].x; a[1] = v[0][0].y; a[2] = v[0][0].z; a[3] = v[0][0].w;
    a[4] = v[0][1].x; a[5] = v[0][1].y; a[6] = v[0][1].z; a[7] = v[0][1].w;
    // halo: first HALO tokens of lane L+1; lane 0 publishes the next sub-tile's
    const bool wrap = lane == 0;
    const uint32_t n0[8] = {wrap ? v[1][0].x : v[0][0].x, wrap ? v[1][0].y : v[0][0].y,
                            wrap ? v[1][0].z : v[0][0].z, wrap ? v[1][0].w : v[0][0].w,
                            wrap ? v[1][1].x : v[0][1].x, wrap ? v[1][1].y : v[0][1].y,
                            wrap ? v[1][1].z : v[0][1].z, wrap ? v[1][1].w : v[0][1].w};
#pragma unroll
    for (int h = 0; h < HALO; ++h) a[8 + h] = __shfl(n0[h], src);
#pragma unroll
    for (int k = 0; k < 8 + HALO; ++k) a[k] = fs_premix(a[k]);
    const uint32_t p0 = base + 8 * lane;
    uint32_t flags;
    if (base + (uint32_t)SUB + HALO > n_tok)
      flags = window_flags8<N, true, LW14>(a, s_filter, word_shift, mask_fffc, p0, n_tok);
    else
      flags = window_flags8<N, false, LW14>(a, s_filter, word_shift, mask_fffc, p0, n_tok);
    if (recs) {
      const uint32_t c = __popc(flags);
      const uint32_t inc = fsdev::wave_incl_scan_dpp(c);
      const uint64_t has = __ballot(flags != 0);
      const uint32_t slot = rec_run + __builtin_amdgcn_mbcnt_hi((uint32_t)(has >> 32),
                                          __builtin_amdgcn_mbcnt_lo((uint32_t)has, 0));
      if (flags != 0 && slot < capw) list[slot] = make_uint2(((p0 >> 3) << 8) | flags, cand_run + inc - c);
      rec_run += (uint32_t)__popcll(has);
      cand_run += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    } else {
      bm_bytes[(size_t)sub * 64 + lane] = (uint8_t)flags;
      const uint32_t cnt = fsdev::wave_sum_lane63(__popc(flags));
      if (lane == 63) qcnt[sub] = cnt;
      cand_run += (uint32_t)__builtin_amdgcn_readlane((int)cnt, 63);
    }
  }
  if (lane == 0) {
    if (info) info[range_id] = make_uint2(rec_run, cand_run);
    if (cand_run) atomicAdd(&s_csum[kc], cand_run);              // LDS
  }
  if (bsum) {
    __syncthreads();
    if (threadIdx.x < kChunksPerBlock) bsum[blockIdx.x * kChunksPerBlock + threadIdx.x] = s_csum[threadIdx.x];
  }
}

// ---- the integer prefilter, eight tokens per lane (n >= 7) ---------------------------
// k_scan_near's rule in k_scan8's shape: lane L owns tokens [8L, 8L+8) of a 512-token sub-tile,
// the n - 1 tokens behind them come from the next lane (and the one after it, n > 9) through
// two DPP wave shifts instead of shuffles through the LDS crossbar, the 3-gram hash is the
// polynomial one of fs_hash.h (one instruction per position), and the rule "every failed
// test within three consecutive positions" is worked out for the lane's eight windows at once:
// with R = T - 3 tests that must hold, window j passes iff for some a in 0..R its first a tests
// and its last R - a tests hold -- runs of set bits, R - 1 shift-ands, and one and-or per a.
// Bitmap and counts as k_scan8 writes them (64 bytes per sub-tile, chunk sums for k_expand).
constexpr int kNearK = 3;
template <int N, bool TAIL>
__device__ __forceinline__ uint32_t window_flags_near8(const uint32_t* m, uint32_t mask_words,
                                                       uint32_t p0, uint32_t n_tok) {
  constexpr int K = kNearK, S = fs_sub_shift(K);
  constexpr int NB = 8 + N - K;                 // 3-gram positions of the lane
  constexpr int T = N - K + 1, R = T - K;       // tests per window, tests that must hold
  static_assert(NB <= 32 && R >= 1, "window size");
  uint32_t x[NB], w[NB];
  x[0] = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) x[0] = (x[0] << S) + m[k];
#pragma unroll
  for (int j = 1; j < NB; ++j) x[j] = (x[j - 1] << S) + m[j - 1 + K];
  // (the filter sits at LDS address 0: the kernel has no static LDS)
  typedef const __attribute__((address_space(3))) uint32_t lds_word;
#pragma unroll
  for (int j = 0; j < NB; ++j) w[j] = fs_sub_word_offset(x[j], mask_words);
#pragma unroll
  for (int j = 0; j < NB; ++j) w[j] = *reinterpret_cast<lds_word*>((uintptr_t)w[j]);
  asm volatile("" : "+v"(w[0]), "+v"(w[NB - 1]));
#pragma unroll
  for (int j = 0; j < NB; ++j) w[j] = fs_shr_by_byte1(w[j], x[j]);
  uint32_t bits = 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) bits = __builtin_amdgcn_alignbit(w[j], bits, 1);
  bits >>= 32 - NB;                             // 3-gram j at bit j
  uint32_t run[R + 1];                          // run[r]: bit i = tests i .. i + r - 1 hold
  run[0] = 0xFFFFFFFFu;
  run[1] = bits;
#pragma unroll
  for (int r = 2; r <= R; ++r) run[r] = run[r - 1] & (bits >> (r - 1));
  uint32_t flags = 0;
#pragma unroll
  for (int a = 0; a <= R; ++a) flags |= run[a] & (run[R - a] >> (a + K));
  flags &= 0xFFu;
  if (TAIL) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if ((uint64_t)p0 + j + N > n_tok) flags &= ~(1u << j);
  }
  return flags;
}

template <int N>
__global__ __launch_bounds__(1024) void k_scan_near8(const uint32_t* __restrict__ tok, uint32_t n_tok,
                                                     const uint32_t* __restrict__ filter,
                                                     int log2_words, uint64_t* __restrict__ qbm,
                                                     uint32_t* __restrict__ qcnt, uint32_t n_sub,
                                                     uint32_t chunk, uint32_t* __restrict__ bsum,
                                                     fs_status* __restrict__ zero) {
  static_assert(N >= 6 && N <= 12, "twelve halo tokens are loaded (nv[12])");
  extern __shared__ __attribute__((aligned(16))) uint32_t s_filter[];
  uint32_t* s_csum = s_filter + (1u << log2_words);           // [kChunksPerBlock]
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    zero->n_cands = 0; zero->n_hits = 0; zero->n_matches = 0; zero->n_rows = 0;
    zero->max_recs = 0; zero->lev_overflow = 0; zero->bad_string = 0; zero->max_rows = 0;
    zero->lsh_pending = 0;
  }
  const int lane = threadIdx.x & 63;
  const uint32_t wave = threadIdx.x >> 6;
  const uint32_t kc = wave >> 2, q = wave & 3;                 // chunk of the workgroup, quarter
  const uint32_t per = (chunk + 3) >> 2;                       // sub-tiles per wave range
  const uint64_t chunk_first = ((uint64_t)blockIdx.x * kChunksPerBlock + kc) * chunk;
  uint32_t j0 = q * per, j1 = j0 + per;
  if (j1 > chunk) j1 = chunk;
  if (j0 > j1) j0 = j1;
  if (chunk_first + j1 > n_sub) j1 = chunk_first + j0 < n_sub ? (uint32_t)(n_sub - chunk_first) : j0;
  if ((uint64_t)blockIdx.x * kChunksPerBlock * chunk >= n_sub) {   // nothing to scan: empty chunks
    if (threadIdx.x < kChunksPerBlock) bsum[blockIdx.x * kChunksPerBlock + threadIdx.x] = 0;
    return;
  }
  if (threadIdx.x < kChunksPerBlock) s_csum[threadIdx.x] = 0;
  copy_filter_to_lds(filter, s_filter, log2_words);
  __syncthreads();
  constexpr int HALO = N - 1;
  constexpr int H1 = HALO < 8 ? HALO : 8, H2 = HALO - H1;     // from the next lane, from the one after
  uint32_t mask_words = ((1u << log2_words) - 1u) << 2;
  asm volatile("" : "+v"(mask_words));          // keep the SDWA operand in a register
  uint8_t* bm_bytes = reinterpret_cast<uint8_t*>(qbm);
  uint32_t cand_run = 0;
  for (uint32_t j = j0; j < j1; ++j) {
    const uint32_t sub = (uint32_t)chunk_first + j;
    const uint32_t base = sub * 512u;
    const uint4* p = reinterpret_cast<const uint4*>(tok + base + 8 * lane);
    const uint4 a0 = p[0], a1 = p[1];
    // the first tokens of the next sub-tile, for the last two lanes (the buffer is padded)
    const uint4* nx = reinterpret_cast<const uint4*>(tok + base + 512);
    const uint4 n0 = nx[0], n1 = nx[1], n2 = nx[2];
    uint32_t m[8 + HALO];
    m[0] = fs_premix(a0.x); m[1] = fs_premix(a0.y); m[2] = fs_premix(a0.z); m[3] = fs_premix(a0.w);
    m[4] = fs_premix(a1.x); m[5] = fs_premix(a1.y); m[6] = fs_premix(a1.z); m[7] = fs_premix(a1.w);
    const uint32_t nv[12] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w};
    // wave_shl:1 -- lane L takes lane L + 1's value, lane 63 keeps `old`
#pragma unroll
    for (int h = 0; h < H1; ++h)
      m[8 + h] = (uint32_t)__builtin_amdgcn_update_dpp((int)fs_premix(nv[h]), (int)m[h], 0x130, 0xF, 0xF, false);
#pragma unroll
    for (int h = 0; h < H2; ++h)
      m[16 + h] = (uint32_t)__builtin_amdgcn_update_dpp((int)fs_premix(nv[8 + h]), (int)m[8 + h], 0x130, 0xF, 0xF, false);
    const uint32_t p0 = base + 8 * lane;
    uint32_t flags;
    if (base + 512u + HALO > n_tok) flags = window_flags_near8<N, true>(m, mask_words, p0, n_tok);
    else flags = window_flags_near8<N, false>(m, mask_words, p0, n_tok);
    bm_bytes[(size_t)sub * 64 + lane] = (uint8_t)flags;
    const uint32_t cnt = fsdev::wave_sum_lane63(__popc(flags));
    if (lane == 63) qcnt[sub] = cnt;
    cand_run += (uint32_t)__builtin_amdgcn_readlane((int)cnt, 63);
  }
  if (lane == 0 && cand_run) atomicAdd(&s_csum[kc], cand_run);    // LDS
  __syncthreads();
  if (threadIdx.x < kChunksPerBlock) bsum[blockIdx.x * kChunksPerBlock + threadIdx.x] = s_csum[threadIdx.x];
}

// ---- prefilter scan and wildcard-key filter in one kernel (round 5) ---------------------
// k_near_sift: k_scan_near8's loop with the first stage of k_lsh_sift behind it in the same
// wave.  Round 4 wrote a bitmap of the windows that pass the 3-gram rule (1.18 M of the 20 M of
// a C2 batch at n = 8), k_expand turned it into a list of positions, and k_lsh_sift read every
// one of them again -- position, ids, three filter blocks -- in two passes of its resident
// workgroups, to keep an eighth; k_hitrows and k_rows then walked the same 1.18 M entries.
// Here a lane whose eight windows hold candidates appends {(position - range start) / 8, flag
// byte} to its wave's queue in LDS (k_scan_rows' record), and whenever 64 candidates are queued
// the wave takes one each: the n ids (the stream it has just read: L2), the grouped wildcard
// filter's three 16-byte blocks, requested together, the n key tests.  The survivors go to the
// wave range's own list in memory, in position order; k_lsh_sift2 (fs_lsh.hip) numbers them
// across the ranges and takes the deeper steps.  No bitmap, no k_expand, no global atomics, and
// everything behind this kernel runs over an eighth of the entries.
constexpr uint32_t kSiftQueue = 128;            // queued records per wave: < 64 left over + <= 64 of a sub-tile
struct alignas(16) SiftLds {                    // per wave
  uint32_t rec[kSiftQueue];                     // {(position - range start) / 8 << 8 | flag byte}
  uint32_t cand[64];                            // window positions of the round being made up
};

// Does the window at `ids` (vector ids, or component ids on tables with near-synonyms) have one
// of its N one-slot-wildcard keys in the grouped filter `wb` (fs_hash.h)?  k_lsh_sift's stage 1
// for one candidate per lane, the window size at compile time.
template <int N>
__device__ __forceinline__ bool wild_stage1(const uint32_t* __restrict__ ids, const uint4* __restrict__ wb,
                                            int log2_wild) {
  constexpr int Q = (N + 3) / 4;
  uint32_t kf[4 * Q];
  // (the window start is only 4-byte aligned; the buffers are padded: fs_device.h, load_ids)
  const uint4* src = reinterpret_cast<const uint4*>(ids);
#pragma unroll
  for (int q4 = 0; q4 < Q; ++q4) {
    const uint4 t = src[q4];
    kf[4 * q4] = t.x; kf[4 * q4 + 1] = t.y; kf[4 * q4 + 2] = t.z; kf[4 * q4 + 3] = t.w;
  }
  uint32_t fold = 0, g0 = 0, g1 = 0, g2 = 0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const uint32_t t = fs_rotl(fs_premix(kf[k]), fs_rot_of(N - 1 - k));
    const int X = fs_wild_group(k, N);
    fold ^= t;
    g0 ^= X == 0 ? t : 0u; g1 ^= X == 1 ? t : 0u; g2 ^= X == 2 ? t : 0u;
  }
  const uint4 blk0 = wb[fs_wild_block(fold ^ g0, 0, log2_wild)];
  const uint4 blk1 = wb[fs_wild_block(fold ^ g1, 1, log2_wild)];
  const uint4 blk2 = wb[fs_wild_block(fold ^ g2, 2, log2_wild)];
  uint32_t any = 0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const uint32_t t = fs_rotl(fs_premix(kf[k]), fs_rot_of(N - 1 - k));
    const uint32_t h = fs_wild_fkey(fold, t, k);
    const int X = fs_wild_group(k, N);
    const uint4 q = X == 0 ? blk0 : X == 1 ? blk1 : blk2;
    any |= fsdev::shr_by_byte<0>(q.x, h) & fsdev::shr_by_byte<1>(q.y, h) & fsdev::shr_by_byte<2>(q.z, h) &
           fsdev::shr_by_byte<3>(q.w, h);
  }
  return (any & 1u) != 0;
}

template <int N>
__global__ __launch_bounds__(1024) void k_near_sift(const uint32_t* __restrict__ tok, uint32_t n_tok,
                                                    const uint32_t* __restrict__ filter, int log2_words,
                                                    const uint32_t* __restrict__ wild, int log2_wild,
                                                    uint32_t n_sub, uint32_t chunk,
                                                    uint32_t* __restrict__ slist, uint32_t caps,
                                                    uint32_t* __restrict__ scount,
                                                    uint32_t* __restrict__ bsum,
                                                    fs_status* __restrict__ zero) {
  static_assert(N >= 6 && N <= 12, "twelve halo tokens are loaded (nv[12])");
  extern __shared__ __attribute__((aligned(16))) uint32_t s_filter[];
  uint32_t* s_csum = s_filter + (1u << log2_words);           // [kChunksPerBlock]
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    zero->n_cands = 0; zero->n_hits = 0; zero->n_matches = 0; zero->n_rows = 0;
    zero->max_recs = 0; zero->lev_overflow = 0; zero->bad_string = 0; zero->max_rows = 0;
    zero->lsh_pending = 0;
  }
  const int lane = threadIdx.x & 63;
  const uint32_t wave = threadIdx.x >> 6;
  SiftLds& Q = reinterpret_cast<SiftLds*>(s_csum + 4)[wave];
  const uint32_t kc = wave >> 2, q = wave & 3;                 // chunk of the workgroup, quarter
  const uint32_t range_id = (blockIdx.x * kChunksPerBlock + kc) * 4 + q;
  const uint32_t per = (chunk + 3) >> 2;                       // sub-tiles per wave range
  const uint64_t chunk_first = ((uint64_t)blockIdx.x * kChunksPerBlock + kc) * chunk;
  uint32_t j0 = q * per, j1 = j0 + per;
  if (j1 > chunk) j1 = chunk;
  if (j0 > j1) j0 = j1;
  if (chunk_first + j1 > n_sub) j1 = chunk_first + j0 < n_sub ? (uint32_t)(n_sub - chunk_first) : j0;
  if ((uint64_t)blockIdx.x * kChunksPerBlock * chunk >= n_sub) {   // nothing to scan: empty chunks
    if (threadIdx.x < kChunksPerBlock) bsum[blockIdx.x * kChunksPerBlock + threadIdx.x] = 0;
    if (lane == 0) scount[range_id] = 0;
    return;
  }
  if (threadIdx.x < kChunksPerBlock) s_csum[threadIdx.x] = 0;
  copy_filter_to_lds(filter, s_filter, log2_words);
  __syncthreads();
  constexpr int HALO = N - 1;
  constexpr int H1 = HALO < 8 ? HALO : 8, H2 = HALO - H1;     // from the next lane, from the one after
  uint32_t mask_words = ((1u << log2_words) - 1u) << 2;
  asm volatile("" : "+v"(mask_words));          // keep the SDWA operand in a register
  const uint4* wb = reinterpret_cast<const uint4*>(wild);
  const uint32_t range_base = ((uint32_t)chunk_first + j0) * 512u;
  uint32_t* my_list = slist + (size_t)range_id * caps;
  uint32_t qn = 0, qc = 0, out_n = 0;           // queued records, queued candidates, survivors so far (wave-uniform)

  // one round: the longest run of queued records that holds 64 candidates or fewer, a lane per candidate
  auto round = [&]() {
    const uint32_t r = (uint32_t)lane < qn ? Q.rec[lane] : 0u;
    const uint32_t cnt = __popc(r & 0xFFu);
    const uint32_t inc = fsdev::wave_incl_scan_dpp(cnt);
    const bool fits = (uint32_t)lane < qn && inc <= 64u;
    const uint32_t nfit = (uint32_t)__popcll(__ballot(fits));            // >= 1: a record holds at most eight
    const uint32_t ncand = (uint32_t)__builtin_amdgcn_readlane((int)inc, (int)nfit - 1);
    if (fits) {
      uint32_t f = r & 0xFFu, k = inc - cnt;
      const uint32_t rel = (r >> 8) << 3;
      while (f) {
        Q.cand[k++] = rel + (uint32_t)(__ffs((int)f) - 1);
        f &= f - 1;
      }
    }
    // the records behind them move to the front of the queue
    const uint32_t i0 = (uint32_t)lane + nfit, i1 = i0 + 64u;
    const uint32_t r0 = i0 < qn ? Q.rec[i0] : 0u, r1 = i1 < qn ? Q.rec[i1] : 0u;
    __builtin_amdgcn_wave_barrier();
    Q.rec[lane] = r0;
    Q.rec[lane + 64] = r1;
    qn -= nfit; qc -= ncand;
    __builtin_amdgcn_wave_barrier();
    const bool live = (uint32_t)lane < ncand;
    const uint32_t p = range_base + (live ? Q.cand[lane] : 0u);
    bool pass = live;
    if (wb) pass = wild_stage1<N>(tok + p, wb, log2_wild) && live;
    const uint64_t sb = __ballot(pass);
    if (pass) {
      const uint32_t slot = out_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(sb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sb, 0u));
      if (slot < caps) my_list[slot] = p;
    }
    out_n += (uint32_t)__popcll(sb);
    __builtin_amdgcn_wave_barrier();
  };

  for (uint32_t j = j0; j < j1; ++j) {
    const uint32_t sub = (uint32_t)chunk_first + j;
    const uint32_t base = sub * 512u;
    const uint4* p = reinterpret_cast<const uint4*>(tok + base + 8 * lane);
    const uint4 a0 = p[0], a1 = p[1];
    // the first tokens of the next sub-tile, for the last two lanes (the buffer is padded)
    const uint4* nx = reinterpret_cast<const uint4*>(tok + base + 512);
    const uint4 n0 = nx[0], n1 = nx[1], n2 = nx[2];
    uint32_t m[8 + HALO];
    m[0] = fs_premix(a0.x); m[1] = fs_premix(a0.y); m[2] = fs_premix(a0.z); m[3] = fs_premix(a0.w);
    m[4] = fs_premix(a1.x); m[5] = fs_premix(a1.y); m[6] = fs_premix(a1.z); m[7] = fs_premix(a1.w);
    const uint32_t nv[12] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w};
    // wave_shl:1 -- lane L takes lane L + 1's value, lane 63 keeps `old`
#pragma unroll
    for (int h = 0; h < H1; ++h)
      m[8 + h] = (uint32_t)__builtin_amdgcn_update_dpp((int)fs_premix(nv[h]), (int)m[h], 0x130, 0xF, 0xF, false);
#pragma unroll
    for (int h = 0; h < H2; ++h)
      m[16 + h] = (uint32_t)__builtin_amdgcn_update_dpp((int)fs_premix(nv[8 + h]), (int)m[8 + h], 0x130, 0xF, 0xF, false);
    const uint32_t p0 = base + 8 * lane;
    uint32_t flags;
    if (base + 512u + HALO > n_tok) flags = window_flags_near8<N, true>(m, mask_words, p0, n_tok);
    else flags = window_flags_near8<N, false>(m, mask_words, p0, n_tok);
    const uint64_t has = __ballot(flags != 0);
    if (flags != 0) {
      const uint32_t slot = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(has >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)has, 0u));
      Q.rec[slot] = (((p0 - range_base) >> 3) << 8) | flags;
    }
    qn += (uint32_t)__popcll(has);
    qc += (uint32_t)__builtin_amdgcn_readlane((int)fsdev::wave_sum_lane63(__popc(flags)), 63);
    __builtin_amdgcn_wave_barrier();
    while (qc >= 64u) round();
  }
  while (qn) round();
  if (lane == 0) {
    scount[range_id] = out_n;                   // (beyond caps: k_lsh_sift2 reports it, the search is repeated)
    const uint32_t kept = out_n < caps ? out_n : caps;
    if (kept) atomicAdd(&s_csum[kc], kept);     // LDS
  }
  __syncthreads();
  if (threadIdx.x < kChunksPerBlock) bsum[blockIdx.x * kChunksPerBlock + threadIdx.x] = s_csum[threadIdx.x];
}

// ---- scan and records in one kernel -------------------------------------------------
// k_scan_rows: the scan loop of k_scan8 with the post-scan work of fs_ranges.hip inside
// the same wave.  Tokens in, output records (staged per wave range) out; nothing else
// touches global memory: no candidate records, no per-candidate arrays.
//
// One workgroup of `blockDim.x / 64` waves per CU (the 64 KB filter plus 3.5 KB of queue
// and hit state per wave).  Wave r scans the contiguous sub-tiles [s0, s1) of range r.
// A lane whose eight windows hold candidates appends one 4-byte record
// {(position - range start) / 8 << 8 | flag byte} to the wave's queue in LDS.  Once
// kRecFlush records are queued (about one full round of candidates), and at the end of
// the range, the wave runs the rounds of fs_ranges.h over the queue: verification,
// hits, records of the words up to the scan front.  The ids of the next sub-tile are
// requested before that, so their latency and the rounds' dependent loads overlap;
// the other waves of the SIMD keep scanning meanwhile (flushes fall at data-dependent
// times, so the waves drift apart instead of queueing at the end of the kernel).
#ifndef FS_REC_QUEUE
#define FS_REC_QUEUE 192
#endif
constexpr uint32_t kRecQueue = FS_REC_QUEUE;   // queued records per wave (a sub-tile adds at most 64)
constexpr uint32_t kRecFlush = 34;    // flush threshold: about 1.35 candidates per record, so
                                      // that queue + halo mostly fit one round

struct alignas(16) FusedLds {         // per wave
  fsdev::RangeLds R;
  uint32_t rec[kRecQueue];            // {(position - range start) / 8 << 8 | flag byte}
  uint32_t cand[64];                  // window positions of the round being made up
};

// (the instance that leaves the hand-off to k_compact runs beside other searches' kernels:
// five waves per SIMD -- 96 registers -- leave their waves a slot where four of 121 do not)
template <int N, int K, bool NT, bool LW14, bool STR = false, bool INL = true>
__global__ __launch_bounds__(1024, INL ? 4 : 5) void k_scan_rows(CorpusDev c, GramIndexDev g,
                                                    uint32_t n_sub, fsdev::RangeOut out,
                                                    fsdev::RowSync sy, fsdev::RowFinal fin,
                                                    uint32_t disp_lds, uint32_t diag,
                                                    unsigned long long* __restrict__ dbg,
                                                    StrFast strf, uint4 shares) {
  using namespace fsdev;
  static_assert(N >= 2 && N <= 8, "halo must fit in the next lane's eight tokens");
  static_assert(kRecQueue >= 192, "a pair of sub-tiles adds up to 128 records to a queue of up to 63");
  // FS_DIAG & 2: twenty words per wave range {entry, filter in LDS, scan done, rounds done,
  // finished, rounds, flushes, records, six phase sums of the rounds (RoundClock), three stamps
  // of the hand-off (workgroup together, counts known to the wave, to the workgroup), -, -, -} in
  // ticks of the 100 MHz constant clock (tools/scan_timeline.py)
  unsigned long long t_entry = 0, t_ready = 0, t_scan = 0, t_rounds = 0;
  FinStamps t_fin;
  t_fin.on = dbg != nullptr; t_fin.t0 = 0; t_fin.t1 = 0; t_fin.t2 = 0;
  uint32_t n_rounds = 0, n_flushes = 0;
  if (dbg) t_entry = __builtin_amdgcn_s_memrealtime();
  // all of the kernel's LDS is dynamic, the filter first: its word offsets are then LDS
  // addresses (a static array in front of it costs an address add per filter read)
  extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
  uint32_t* s_filter = s_dyn;
  // displacement seeds of the exact table as bytes behind the filter (disp_lds bytes, a
  // multiple of 16; 0: too many, read from memory)
  const int lw = K ? g.log2_swords : g.log2_words;
  const uint32_t disp_off = disp_lds ? 4u << lw : FS_NO_LDS;
  FusedLds* s_wave = reinterpret_cast<FusedLds*>(s_dyn + (1u << lw) + disp_lds / 4);
  const int lane = threadIdx.x & 63;
  // (the wave index as a scalar: what follows from it -- range, sub-tile numbers, the halo address --
  // then lives in scalar registers and is loaded by scalar loads)
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), n_waves = blockDim.x >> 6;
  FusedLds& W = s_wave[wave];
  RangeLds& S = W.R;
  uint32_t* s_cnt = reinterpret_cast<uint32_t*>(s_wave + n_waves);   // 6 * n_waves + 2 words (finish_rows)
  // shared rounds (fs_ranges.h): the slices the waves post for each other, behind s_cnt
  CoopLds& C = *reinterpret_cast<CoopLds*>(s_cnt + ((6 * n_waves + 2 + 3) & ~3u));
  // ... when the launch puts the records into place itself (FS_DIAG & 256: every wave does all
  // the rounds of its range by itself, as before)
  // (INL: the launch puts the records into place itself; the other instance leaves that to
  // k_compact and is compiled without the registers and the code of what follows from it)
  const bool coop = INL && !STR && !sy.rinfo && out.xstage && !(diag & 256) && n_waves <= kCoopWaves;
  const uint32_t range_id = blockIdx.x * n_waves + wave;
  // Sub-tiles dealt out evenly over the workgroups (the first n_sub % gridDim.x take one
  // more), and inside a workgroup of sixteen waves by how fast its SIMD serves each wave: the
  // four waves of a SIMD (slots w, w + 4, w + 8, w + 12) are served oldest first, and over
  // equal shares the youngest takes 60 % longer than the oldest (in-kernel stamps,
  // tools/scan_timeline.py: 10.3 / 11.0 / 12.2 / 16.5 us), so the shares are 29.8 / 26.8 /
  // 24.5 / 18.9 % of a quarter each.  What comes out does not depend on the split.
  uint32_t s0, s1;
  {
    const uint32_t per = n_sub / gridDim.x, rem = n_sub % gridDim.x;
    const uint32_t b0 = blockIdx.x * per + (blockIdx.x < rem ? blockIdx.x : rem);
    const uint32_t len = per + (blockIdx.x < rem ? 1u : 0u);
    auto cut = [&](uint32_t w) -> uint32_t {             // first sub-tile of wave w (w <= n_waves)
      if (n_waves != 16 || (diag & 16)) return (uint32_t)(((uint64_t)len * w) / n_waves);
      const uint32_t g = w >> 2, r = w & 3;
      const uint32_t share[5] = {0, shares.x, shares.y, shares.z, 1024};   // cumulative, in 1/1024 of a quarter
      const uint32_t at = 4 * share[g] + r * (share[g + 1] - share[g]);
      return (uint32_t)(((uint64_t)len * at) >> 12);
    };
    s0 = b0 + cut(wave);
    s1 = b0 + cut(wave + 1);
  }

  constexpr uint32_t HALO = N - 1;
  constexpr uint32_t SUB = 512;
  const int word_shift = 32 - lw;
  uint32_t mask_fffc = 0xFFFCu;
  asm volatile("" : "+v"(mask_fffc));          // keep the SDWA operand in a register
  uint32_t mask_words = ((1u << lw) - 1u) << 2;
  asm volatile("" : "+v"(mask_words));
  const uint32_t* __restrict__ tok = c.tok;
  const uint32_t n_tok = c.n_tok;

  // Sub-tiles go in pairs: per lane its eight ids of sub-tile A (a*), of sub-tile B (b*)
  // and the first eight ids behind B (h*: lane 63's halo in B; A's halo is lane 0's b*).
  typedef uint32_t v4u __attribute__((ext_vector_type(4)));
  struct Pair { v4u a0, a1, b0, b1, h0, h1; };
  auto request = [&](uint32_t sub) {
    Pair r;
    const v4u* p = reinterpret_cast<const v4u*>(tok + sub * SUB + 8 * lane);
    if constexpr (NT) {
      r.a0 = __builtin_nontemporal_load(&p[0]);
      r.a1 = __builtin_nontemporal_load(&p[1]);
      r.b0 = __builtin_nontemporal_load(&p[SUB / 4]);
      r.b1 = __builtin_nontemporal_load(&p[SUB / 4 + 1]);
    } else {
      r.a0 = p[0]; r.a1 = p[1]; r.b0 = p[SUB / 4]; r.b1 = p[SUB / 4 + 1];
    }
    const v4u* hp = reinterpret_cast<const v4u*>(tok + sub * SUB + 2 * SUB);   // the buffer is padded
    r.h0 = hp[0]; r.h1 = hp[1];
    return r;
  };
  // the point where a pair's ids are needed: whole 16-byte registers, so the compiler
  // does not copy single ids out of a request early (it waits for the data wherever it
  // puts such a copy)
  auto arrive = [&](Pair v) {
    asm volatile("" : "+v"(v.a0), "+v"(v.a1), "+v"(v.b0), "+v"(v.b1), "+v"(v.h0), "+v"(v.h1));
    return v;
  };
  // Prologue: seeds and filter into LDS through registers, four 16-byte pieces per thread
  // in flight.  (Round 4 measured the LDS-DMA form -- global_load_lds_dwordx4 pieces behind
  // the range's first id request: the filter is in LDS at 4.35 us instead of 2.7 us, one
  // loader stream per wave lands 1 KB per ~0.65 us, and the kernel is 0.5 us slower.)
  for (uint32_t e = threadIdx.x; e < disp_lds / 16; e += blockDim.x)     // (a multiple of 16 bytes)
    reinterpret_cast<uint4*>(s_dyn + (1u << lw))[e] = reinterpret_cast<const uint4*>(g.disp8)[e];
  copy_filter_to_lds(K ? g.sfilter : g.filter, s_filter, lw);
  for (uint32_t e = threadIdx.x; e < kCoopSlots + kCoopWaves; e += blockDim.x) C.ring[e] = 0;
  if (threadIdx.x < n_waves) s_cnt[threadIdx.x] = 0;        // records per wave: LDS atomics add to them
  if (threadIdx.x == 0) { C.head = 0; C.tail = 0; C.posted = 0; C.pool = 0; }
  // Instruction priority: a wave that scans goes in front of the waves that work their queues
  // off (the ids come from HBM, the rounds wait on L2 and LDS: with the scanners served first
  // the last of them is through 4 us earlier, 19 against 23 us, and the kernel 1.3-1.7 us
  // shorter).  Only where the launch has the GPU to itself: beside other searches' kernels it
  // measured 0.8 us slower per step.  FS_DIAG & 128: off.
  const bool prio = INL && !sy.rinfo && !(diag & 128);
  if (prio) __builtin_amdgcn_s_setprio(2);
  __syncthreads();
  if (dbg) t_ready = __builtin_amdgcn_s_memrealtime();
  RangeState R;
  R.E = 0; R.hc = 0; R.rows_run = 0; R.hits_run = 0; R.match_acc = 0;
  R.k0 = make_uint4(0, 0, 0, 0); R.k1 = make_uint4(0, 0, 0, 0);
  // the launch puts the records into place itself: a range's first 128 stay in registers
  // (FS_DIAG & 32: all of them through the staging area)
  const bool keep_regs = INL && !STR && !sy.rinfo && !(diag & 32);   // (STR: the per-hit Levenshtein needs the registers)
  RoundClock clk;
  clk.on = !STR && dbg != nullptr && (diag & 4);      // FS_DIAG & 4: the phase sums too (they drain the loads at the phase ends)
  clk.t0 = clk.t1 = clk.t2 = clk.t3 = clk.t4 = clk.t5 = 0; clk.last = 0;
  uint32_t cacc = 0;                           // per lane: candidates seen
  uint32_t my_slices = 0;                      // slices posted by this wave
  // a round over the records [r0, r1) of a queue (all of them fit it), behind `hn` windows in
  // front of `hbase + hn` that are verified whatever the filter said; the lanes of the
  // records write the window positions into cand[]
  auto make_round = [&](const uint32_t* qrec, uint32_t qa, uint32_t r0, uint32_t r1, uint32_t hn, uint32_t hbase) {
    const uint32_t rec = r0 + (uint32_t)lane < r1 ? qrec[r0 + lane] : 0u;
    uint32_t fb = rec & 0xFFu;
    const uint32_t cn = __popc(fb);
    const uint32_t inc = wave_incl_scan_dpp(cn);
    if (cn) {
      uint32_t at = hn + inc - cn;
      const uint32_t pb = qa + ((rec >> 8) << 3) - 1u;
      do {
        W.cand[at++] = pb + (uint32_t)__ffs(fb);
        fb &= fb - 1;
      } while (fb);
    }
    if ((uint32_t)lane < hn) W.cand[lane] = hbase + (uint32_t)lane;
    const uint32_t total = hn + (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    wave_sync();
    return (uint32_t)lane < total ? W.cand[lane] : FS_NONE;
  };
  if (s1 > s0) {
    const uint32_t a = s0 * SUB, bnd = s1 * SUB;
    R.E = a;
    uint32_t rec_cnt = 0, halo_n = a ? HALO : 0;
    // The next pair is requested as soon as this one has arrived, a whole pair of work
    // (and its flushes) ahead; every step issues a request, the last one for the range's
    // last sub-tiles again (a request under a condition would only complicate the waits).
    // premixed ids of a lane's eight tokens (pinned: left to itself the compiler folds the
    // premix into the first K-gram's shifts as three quarter-rate 32-bit multiplies)
    struct M8 { uint32_t v[8]; };
    auto premix8 = [&](const v4u& lo, const v4u& hi) {
      M8 m;
      m.v[0] = fs_premix(lo.x); m.v[1] = fs_premix(lo.y); m.v[2] = fs_premix(lo.z); m.v[3] = fs_premix(lo.w);
      m.v[4] = fs_premix(hi.x); m.v[5] = fs_premix(hi.y); m.v[6] = fs_premix(hi.z); m.v[7] = fs_premix(hi.w);
      asm volatile("" : "+v"(m.v[0]), "+v"(m.v[1]), "+v"(m.v[2]), "+v"(m.v[3]),
                        "+v"(m.v[4]), "+v"(m.v[5]), "+v"(m.v[6]), "+v"(m.v[7]));
      return m;
    };
    // one sub-tile: m = the lane's own premixed ids, halo[h] = premixed id h of the lane behind
    auto scan = [&](const M8& m, const uint32_t* halo, uint32_t j) {
      const uint32_t base = j * SUB;
      uint32_t aa[16];
#pragma unroll
      for (int k = 0; k < 8; ++k) aa[k] = m.v[k];
#pragma unroll
      for (int h = 0; h < (int)HALO; ++h) aa[8 + h] = halo[h];
      const uint32_t p0 = base + 8 * lane;
      uint32_t flags;
      if constexpr (K != 0) {
        if (base + SUB + HALO > n_tok)
          flags = window_flags_sub<N, K, true>(aa, s_filter, mask_words, p0, n_tok);
        else
          flags = window_flags_sub<N, K, false>(aa, s_filter, mask_words, p0, n_tok);
      } else {
        if (base + SUB + HALO > n_tok)
          flags = window_flags8<N, true, LW14>(aa, s_filter, word_shift, mask_fffc, p0, n_tok);
        else
          flags = window_flags8<N, false, LW14>(aa, s_filter, word_shift, mask_fffc, p0, n_tok);
      }
      const uint64_t has = __ballot(flags != 0);
      const uint32_t slot = rec_cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(has >> 32),
                                          __builtin_amdgcn_mbcnt_lo((uint32_t)has, 0));
      if (flags != 0) W.rec[slot] = ((((base - a) >> 3) + (uint32_t)lane) << 8) | flags;
      rec_cnt += (uint32_t)__popcll(has);
      cacc += __popc(flags);
    };
    // records queued so far -> rounds of candidates; F_end = the scan front.  A round takes
    // the records whose candidates fit the hit slots left by the carried hits (a record has
    // at most eight, so the first always fits): lane t looks at record rt + t, a prefix sum of
    // the candidate counts says which records fit and where their candidates go, and the
    // lanes of those records write the window positions into cand[] -- three LDS round
    // trips (records, positions, positions back), no search.
    auto flush = [&](uint32_t F_end, bool final) {
      if (prio) __builtin_amdgcn_s_setprio(0);
      if (clk.on) clk.last = (uint32_t)__builtin_amdgcn_s_memrealtime();
      wave_sync();
      uint32_t rt = 0;                             // records taken so far
      uint32_t rec_end = rec_cnt;                  // ... of those this wave works off itself
      do {
        const uint32_t take = R.hc ? 64u - R.hc : 63u;
        const uint32_t hn = halo_n;
        // the records that fit this round: a prefix sum of their candidate counts
        uint32_t m;
        {
          const uint32_t rec = rt + (uint32_t)lane < rec_end ? W.rec[rt + lane] : 0u;
          const uint32_t cn = __popc(rec & 0xFFu);
          const uint32_t inc = wave_incl_scan_dpp(cn);
          m = (uint32_t)__popcll(__ballot(cn != 0 && hn + inc <= take));      // (a prefix of the lanes)
        }
        // the first position whose hit status is not known after this round: the first
        // candidate left behind, or the scan front
        uint32_t F = F_end;
        if (rt + m < rec_end) {
          const uint32_t r2 = W.rec[rt + m];       // (the same word for every lane)
          const uint32_t nx = a + ((r2 >> 8) << 3) + (uint32_t)__ffs(r2 & 0xFFu) - 1u;
          if (nx < F) F = nx;
          // End of the range, more candidates than this round takes: what is left is cut
          // into slices of one round each and posted for the waves that are through with their
          // own ranges (fs_ranges.h, shared rounds) -- all of it or, with more than
          // kCoopPerWave slices, none of it (this wave then goes on by itself)
          if (final && coop && my_slices == 0) {
            const uint32_t slot0 = wave * kCoopPerWave;
            uint32_t q = rt + m, ns = 0;
            bool ok = true;
            while (q < rec_end) {
              if (ns == kCoopPerWave) { ok = false; break; }
              const uint32_t rq = q + (uint32_t)lane < rec_end ? W.rec[q + lane] : 0u;
              const uint32_t cq = __popc(rq & 0xFFu);
              const uint32_t iq = wave_incl_scan_dpp(cq);
              const uint32_t mm = (uint32_t)__popcll(__ballot(cq != 0 && HALO + iq <= 63u));
              const uint32_t r1st = (uint32_t)__builtin_amdgcn_readfirstlane((int)rq);
              const uint32_t P = a + ((r1st >> 8) << 3) + (uint32_t)__ffs(r1st & 0xFFu) - 1u;
              if (lane == 0) {
                CoopSlice& d = C.slice[slot0 + ns];
                d.r0 = q; d.r1 = q + mm; d.P = P; d.F = F_end; d.a = a;
                d.base = 0; d.rows = 0; d.hits = 0; d.pairs = 0;
                if (ns) C.slice[slot0 + ns - 1].F = P;
              }
              q += mm;
              ++ns;
            }
            if (ok) {
              if (lane == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                const uint32_t t0 = __hip_atomic_fetch_add(&C.tail, ns, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                for (uint32_t k = 0; k < ns; ++k)
                  __hip_atomic_store(&C.ring[t0 + k], slot0 + k + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
              my_slices = ns;
              rec_end = rt + m;                    // this round is the wave's last of the range
            }
          }
        }
        const uint32_t p = make_round(W.rec, a, rt, rt + m, hn, a - hn);
        if (!(diag & 1))
          range_round<N, STR>(c, g, disp_off, S, p, F, a, range_id, out, R,
                              &strf, reinterpret_cast<uint32_t*>(fin.host_st + 1), keep_regs, clk);
        rt += m;
        halo_n = 0;
        ++n_rounds;
      } while (rt < rec_end);
      rec_cnt = 0;
      ++n_flushes;
      if (prio && !final) __builtin_amdgcn_s_setprio(2);
    };
    // (shared rounds: what a range queues is best kept for its end, where the waves share the
    // rounds; a wave that has to empty its queue in mid-range does so by itself)
    const uint32_t flush_req = (diag >> 12) ? (diag >> 12) : coop ? kRecQueue - 128 : kRecFlush;
    const uint32_t flush_at = flush_req < kRecQueue - 128 ? flush_req : kRecQueue - 128;
    Pair nx = request(s0);
    for (uint32_t j = s0; j < s1; j += 2) {
      const Pair v = arrive(nx);
      nx = request(j + 2 < s1 ? j + 2 : j);
      const M8 ma = premix8(v.a0, v.a1), mb = premix8(v.b0, v.b1);
      uint32_t halo[8];
      // A's halo: the first ids of the lane behind, lane 63: lane 0's first ids of B -- one
      // DPP rotation of the wave per id (no LDS), lane 0 offering its B ids
#pragma unroll
      for (int h = 0; h < (int)HALO; ++h)
        halo[h] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(lane == 0 ? mb.v[h] : ma.v[h]), 0x134, 0xF, 0xF, false);   // wave_rol:1
      scan(ma, halo, j);
      const bool odd_end = j + 1 == s1;            // the range ends behind A
      if (!odd_end) {
        // B's halo: lane 63 takes the ids behind B (the same in every lane): a DPP shift of
        // the wave, the lane without a source keeps `old`
        auto hv = [&](int h) {
          return h == 0 ? v.h0.x : h == 1 ? v.h0.y : h == 2 ? v.h0.z : h == 3 ? v.h0.w
               : h == 4 ? v.h1.x : h == 5 ? v.h1.y : h == 6 ? v.h1.z : v.h1.w;
        };
#pragma unroll
        for (int h = 0; h < (int)HALO; ++h)
          halo[h] = (uint32_t)__builtin_amdgcn_update_dpp((int)fs_premix(hv(h)), (int)mb.v[h], 0x130, 0xF, 0xF, false);   // wave_shl:1
        scan(mb, halo, j + 1);
      }
      // ONE place where the queue is worked off (the rounds are a lot of code: a second copy
      // of them competes for the instruction cache): behind the pair.  The queue holds
      // kRecQueue = 192 records, a pair adds at most 128, so a queue below 64 at the start of
      // a pair cannot overflow.
      const bool end = odd_end || j + 2 == s1;
      if (dbg && end) t_scan = __builtin_amdgcn_s_memrealtime();
      if (end ? (rec_cnt | halo_n | R.hc) != 0 : rec_cnt >= flush_at) flush(end ? bnd : j * SUB + 2 * SUB, end);
    }
  }
  // Shared rounds: this wave is through with its range; it takes slices in posting order
  // (a ticket each) until every wave is through and every posted slice has been taken.
  if (prio) __builtin_amdgcn_s_setprio(0);
  if (coop) {
    if (lane == 0) __hip_atomic_fetch_add(&C.posted, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    const RangeOut xout{out.xstage + (size_t)blockIdx.x * out.xpool * pool_rec_bytes(out.wire), out.xpool, out.wire,
                        nullptr, 0};
    for (;;) {
      uint32_t t = 0;
      if (lane == 0) t = __hip_atomic_fetch_add(&C.head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
      uint32_t slot = 0;
      for (;;) {
        slot = __hip_atomic_load(&C.ring[t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (slot) break;
        // (every wave through: the count of posted slices is final, and a ticket below it
        // will be served)
        if (__hip_atomic_load(&C.posted, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == n_waves &&
            t >= __hip_atomic_load(&C.tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
        __builtin_amdgcn_s_sleep(8);     // (a wave that waits for a ticket leaves the issue slots to the others)
      }
      slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
      if (!slot) break;
      CoopSlice& d = C.slice[slot - 1];
      const uint32_t d_r0 = d.r0, d_r1 = d.r1, d_P = d.P, d_F = d.F, d_a = d.a;
      RangeState Rs;
      Rs.E = d_P; Rs.hc = 0; Rs.rows_run = 0; Rs.hits_run = 0; Rs.match_acc = 0; Rs.pool_base = 0;
      Rs.k0 = make_uint4(0, 0, 0, 0); Rs.k1 = make_uint4(0, 0, 0, 0);
      wave_sync();
      const uint32_t p = make_round(s_wave[(slot - 1) / kCoopPerWave].rec, d_a, d_r0, d_r1, HALO, d_P - HALO);
      if (!(diag & 1))
        range_round<N, STR>(c, g, disp_off, S, p, d_F, d_P, 0u, xout, Rs,
                            &strf, reinterpret_cast<uint32_t*>(fin.host_st + 1), false, clk, &C.pool);
      // its records count for the wave that posted it, its hits and pairs for this one
      R.hits_run += Rs.hits_run;
      R.match_acc += Rs.match_acc;
      if (lane == 0) {
        d.base = Rs.pool_base; d.rows = Rs.rows_run - Rs.pool_base;
        __hip_atomic_fetch_add(&s_cnt[(slot - 1) / kCoopPerWave], Rs.rows_run - Rs.pool_base, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      ++n_rounds;
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    R.match_acc += (uint32_t)__shfl_xor((int)R.match_acc, d);
    cacc += (uint32_t)__shfl_xor((int)cacc, d);
  }
  if (dbg) t_rounds = __builtin_amdgcn_s_memrealtime();
  finish_rows(sy, fin, out, g.selfdist, range_id, R.rows_run, R.hits_run, R.match_acc, cacc, s_cnt,
              keep_regs ? &R : nullptr, coop ? &C : nullptr, my_slices,
              reinterpret_cast<uint32_t*>(fin.host_st + 1) + 1, C.stat, t_fin);
  if (dbg && lane == 0) {
    unsigned long long* d = dbg + 20 * (size_t)range_id;
    d[0] = t_entry; d[1] = t_ready; d[2] = t_scan; d[3] = t_rounds;
    d[4] = __builtin_amdgcn_s_memrealtime(); d[5] = n_rounds; d[6] = n_flushes; d[7] = R.rows_run;
    d[8] = clk.t0; d[9] = clk.t1; d[10] = clk.t2; d[11] = clk.t3; d[12] = clk.t4; d[13] = clk.t5;
    d[14] = t_fin.t0; d[15] = t_fin.t1; d[16] = t_fin.t2; d[17] = 0; d[18] = 0; d[19] = 0;
  }
}

// Simple variant for any n <= FS_MAX_WINDOW: every lane reads its ids straight
// from global memory (L1-served).  Slower; kept as the cross-check of k_scan and
// as the fallback for window sizes without a specialisation.
__global__ __launch_bounds__(1024) void k_scan_simple(const uint32_t* __restrict__ tok,
                                                      uint32_t n_tok,
                                                      const uint32_t* __restrict__ filter,
                                                      int log2_words, int n,
                                                      uint64_t* __restrict__ qbm,
                                                      uint32_t* __restrict__ qcnt,
                                                      uint32_t n_bm_words) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_filter[];
  for (uint32_t i = threadIdx.x; i < (1u << log2_words); i += blockDim.x) s_filter[i] = filter[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t word = wave; word < n_bm_words; word += n_waves) {
    const uint32_t p0 = word * kSubTile + 4 * lane;
    uint64_t b[4];
    for (int j = 0; j < 4; ++j) {
      const uint64_t p = (uint64_t)p0 + j;
      bool hit = false;
      if (p + n <= n_tok) {
        const uint32_t h = fs_gram_hash(tok + p, n);
        const uint32_t w = s_filter[fs_bloom_word(h, log2_words)];
        const uint32_t m = fs_bloom_mask(h);
        hit = (w & m) == m;
      }
      b[j] = __ballot(hit);
    }
    store_ballots(b, lane, word, n_bm_words, qbm, qcnt);
  }
}

// hipFuncAttributeMaxDynamicSharedMemorySize is set once per kernel and size (per
// device: the attribute belongs to the loaded code object), not on every launch
static constexpr size_t kCuLdsBytes = 160 * 1024;  // gfx950: LDS per CU = the most one workgroup may ask for
static int ensure_dynamic_lds(const void* kern, int device, size_t lds) {
  struct Seen { const void* k; int dev; size_t lds; };
  static std::vector<Seen> seen;                  // a handful of entries
  static std::mutex mu;                           // handles of different devices may be driven from different threads
  std::lock_guard<std::mutex> lock(mu);
  for (const Seen& e : seen)
    if (e.k == kern && e.dev == device && e.lds >= lds) return FS_OK;
  FS_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  seen.push_back({kern, device, lds});
  return FS_OK;
}

template <int N, bool NT>
int launch_tpl8_k(const fs_index* ix, const CorpusDev& c, uint64_t* qbm, uint32_t* qcnt,
                  uint32_t n_bm_words, hipStream_t s, hipEvent_t e0, hipEvent_t e1,
                  fs_scan_extra* ex) {
  uint32_t* bsum = ex ? ex->bsum : nullptr;
  if (n_bm_words == 0 && !bsum) return FS_OK;
  const size_t lds_filter = (size_t)4 << ix->log2_words;
  // the chained kernels' chunking (chunk_of_block): kNB chunks of `chunk` sub-tiles
  const uint32_t chunk = std::max<uint32_t>(1, (n_bm_words + fsdev::kNB - 1) / fsdev::kNB);
  const bool direct = ex && ex->bsum && ex->recs && ex->info && ex->capw && fs_scan_direct_ok(ix, c.n_tok);
  auto kern = ix->log2_words == 14 ? k_scan8<N, NT, true> : k_scan8<N, NT, false>;
  const size_t lds = lds_filter;
  FS_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), ix->device, lds));
  hipExtLaunchKernelGGL(kern, dim3(kScanBlocks), dim3(1024), (uint32_t)lds, s, e0, e1, 0u, c.tok,
                        c.n_tok, (const uint32_t*)ix->d_filter.p, ix->log2_words, qbm, qcnt,
                        n_bm_words, chunk, bsum, ex ? ex->zero : nullptr,
                        direct ? ex->recs : nullptr, direct ? ex->info : nullptr,
                        direct ? ex->capw : 0u);
  FS_HIP(hipGetLastError());
  if (ex) { ex->counted = bsum != nullptr; ex->direct = direct; }
  return FS_OK;
}

template <int N>
int launch_tpl8(const fs_index* ix, const CorpusDev& c, uint64_t* qbm, uint32_t* qcnt,
                uint32_t n_bm_words, hipStream_t s, hipEvent_t e0, hipEvent_t e1,
                fs_scan_extra* ex) {
  const bool big = (uint64_t)c.n_tok * 4 > (256ull << 20);
  const char e = ix->sw.scan_flags;
  const bool nt = e == 'n' || (!e && big);
  return nt ? launch_tpl8_k<N, true>(ix, c, qbm, qcnt, n_bm_words, s, e0, e1, ex)
            : launch_tpl8_k<N, false>(ix, c, qbm, qcnt, n_bm_words, s, e0, e1, ex);
}

}  // namespace

// tokens per lane (bitmap layout) of the chained kernels' scan: eight with k_scan8
// (n = 2..8), four with k_scan_simple (any n; FS_SCAN_TPL=4 or FS_SCAN_VARIANT=simple ask for it)
int fs_scan_tpl(const fs_index* ix, uint64_t n_tok) {
  (void)n_tok;
  if (ix->sw.scan_simple || ix->sw.scan_tpl == 4) return 4;
  const int n = ix->cfg.window_size;
  return n >= 2 && n <= 8 ? 8 : 4;
}

// tokens of zero padding the corpus buffer carries behind n_tok so that the
// tile loads of the last wave never leave the allocation
uint32_t fs_scan_pad_tokens() { return 512 * 8 + 64; }

// The direct path packs (position / 8) << 8 | flags into 32 bits: positions below 2^26,
// which is also the limit of the eight-tokens-per-lane kernel unless it is forced.
bool fs_scan_direct_ok(const fs_index* ix, uint64_t n_tok) {
  if (!ix->sw.scan_direct) return false;
  return fs_scan_tpl(ix, n_tok) == 8 && n_tok + 1024 < (1ull << 26);
}

// extra (optional, see fs_scan_extra): honoured by the eight-tokens-per-lane kernel
// only; the other kernels leave counting and expansion to k_reduce / k_expand.
int fs_launch_scan(const fs_index* ix, const CorpusDev& c, uint64_t* qbm, uint32_t* qcnt,
                   uint32_t n_bm_words, hipStream_t s, hipEvent_t e0, hipEvent_t e1,
                   fs_scan_extra* extra) {
  if (extra) { extra->counted = false; extra->direct = false; }
  const int n = ix->cfg.window_size;
  if (fs_scan_tpl(ix, c.n_tok) == 8) {
    switch (n) {
      case 2: return launch_tpl8<2>(ix, c, qbm, qcnt, n_bm_words, s, e0, e1, extra);
      case 3: return launch_tpl8<3>(ix, c, qbm, qcnt, n_bm_words, s, e0, e1, extra);
      case 4: return launch_tpl8<4>(ix, c, qbm, qcnt, n_bm_words, s, e0, e1, extra);
      case 5: return launch_tpl8<5>(ix, c, qbm, qcnt, n_bm_words, s, e0, e1, extra);
      case 6: return launch_tpl8<6>(ix, c, qbm, qcnt, n_bm_words, s, e0, e1, extra);
      case 7: return launch_tpl8<7>(ix, c, qbm, qcnt, n_bm_words, s, e0, e1, extra);
      case 8: return launch_tpl8<8>(ix, c, qbm, qcnt, n_bm_words, s, e0, e1, extra);
      default: break;
    }
  }
  if (n_bm_words == 0) return FS_OK;
  const size_t lds = (size_t)4 << ix->log2_words;
  FS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_simple),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  uint32_t blocks = (n_bm_words + 15) / 16;
  const uint32_t max_blocks = ix->num_cu * (lds <= 64 * 1024 ? 2 : 1);
  if (blocks > max_blocks) blocks = max_blocks;
  hipExtLaunchKernelGGL(k_scan_simple, dim3(blocks), dim3(1024), (uint32_t)lds, s, e0, e1, 0u,
                        c.tok, c.n_tok, (const uint32_t*)ix->d_filter.p, ix->log2_words, n, qbm,
                        qcnt, n_bm_words);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

// ---- k_scan_rows (scan + records) -----------------------------------------------------
// bytes of LDS for the displacement seeds (one per bucket, rounded up to 16), 0 = the
// kernel reads them from memory
static uint32_t fs_scan_rows_disp_lds(const fs_index* ix) {
  const uint32_t bytes = ((1u << ix->log2_buckets) + 15u) & ~15u;
  return bytes <= 16 * 1024 && ix->sw.rows_disp_lds ? bytes : 0;
}

// Shares of a SIMD's four waves in their quarter of the workgroup's sub-tiles, oldest slot
// first, cumulative in 1/1024 (FS_ROWS_SHARES=a,b,c,d).  The waves of a SIMD are served
// oldest first; the shares make them finish their scans together.
uint4 fs_scan_rows_shares(const fs_index* ix, bool coop);
uint4 fs_scan_rows_shares(const fs_index* ix, bool coop) {
  const int* sh = ix->sw.rows_shares;
  if (sh[0] > 0 && sh[0] + sh[1] + sh[2] + sh[3] == 1024)
    return make_uint4((uint32_t)sh[0], (uint32_t)(sh[0] + sh[1]), (uint32_t)(sh[0] + sh[1] + sh[2]), 1024u);
  (void)coop;
  return make_uint4(305u, 579u, 830u, 1024u);
}

namespace {

// log2 words of the filter k_scan_rows holds in LDS: the sub-shingle filter's where it
// applies (fs_hash.h), else the Bloom filter's
static int rows_filter_log2(const fs_index* ix) {
  return fs_sub_k((int)ix->cfg.window_size) && ix->sw.scan_sub && ix->d_sfilter.p ? ix->log2_swords
                                                                                   : ix->log2_words;
}

template <int N>
int launch_scan_rows(fs_index* ix, fs_corpus* c, uint32_t n_sub, uint32_t waves, uint32_t blocks,
                     const fsdev::RangeOut& out, const fsdev::RowSync& sy,
                     const fsdev::RowFinal& fin, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
  unsigned long long* dbg = nullptr;
  if (ix->sw.diag & 2) {               // timeline stamps, read back by fs_debug_stamps
    FS_TRY(ix->cur->w_dbg.reserve((size_t)blocks * waves * 20));
    dbg = ix->cur->w_dbg.p;
    ix->cur->dbg_words = (size_t)blocks * waves * 20;
  }
  const uint32_t disp_lds = fs_scan_rows_disp_lds(ix);
  const int lw = rows_filter_log2(ix);
  const size_t lds = ((size_t)4 << lw) + disp_lds + waves * sizeof(FusedLds) +
                     ((6 * waves + 2 + 3) & ~3u) * sizeof(uint32_t) + (sy.rinfo ? 0 : sizeof(fsdev::CoopLds));
  // (non-temporal id loads only on request, FS_SCAN_FLAGS=n: they measured slower at every
  // batch size, 0.50 against 0.61 of peak on a 1 GB batch)
  const bool nt = ix->sw.scan_flags == 'n';
  const bool lw14 = lw == 14;
  constexpr int K = fs_sub_k(N);
  // (the instances with non-temporal id loads, FS_SCAN_FLAGS=n, are gone: they measured slower at
  // every batch size, 0.50 against 0.61 of peak on a 1 GB batch)
  const bool inl = sy.rinfo == nullptr;
  const bool sub = K != 0 && ix->sw.scan_sub && ix->d_sfilter.p;
  auto pick = [&](auto kn, auto kk) {           // kn: no sub-shingle filter, kk: with it
    return sub ? kk : kn;
  };
  (void)nt;
  auto kern = k_scan_rows<N, 0, false, false, false, true>;
  if (!c->has_str) {
    if (inl) kern = lw14 ? pick(k_scan_rows<N, 0, false, true, false, true>, k_scan_rows<N, K, false, true, false, true>)
                         : pick(k_scan_rows<N, 0, false, false, false, true>, k_scan_rows<N, K, false, false, false, true>);
    else kern = lw14 ? pick(k_scan_rows<N, 0, false, true, false, false>, k_scan_rows<N, K, false, true, false, false>)
                     : pick(k_scan_rows<N, 0, false, false, false, false>, k_scan_rows<N, K, false, false, false, false>);
  }
  StrFast strf{nullptr, nullptr, 0, 0, nullptr};
  if (c->has_str) {                    // (fs_scan_rows_shape has checked that the path applies)
    strf = StrFast{ix->d_pat.p, ix->d_clsmap.p, ix->n_cls, ix->str_punct, c->d_strrec.p};
    kern = lw14 ? pick(k_scan_rows<N, 0, false, true, true, true>, k_scan_rows<N, K, false, true, true, true>)
                : pick(k_scan_rows<N, 0, false, false, true, true>, k_scan_rows<N, K, false, false, true, true>);
  }
  FS_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), ix->device, lds));
  {
    // the kernel addresses its filter from LDS address 0: it must not own static LDS
    static std::vector<const void*> checked;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (std::find(checked.begin(), checked.end(), (const void*)kern) == checked.end()) {
      hipFuncAttributes fa;
      FS_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern)));
      if (fa.sharedSizeBytes != 0) { fs_set_error("k_scan_rows owns %zu bytes of static LDS", fa.sharedSizeBytes); return FS_E_DEVICE; }
      checked.push_back((const void*)kern);
    }
  }
  hipExtLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), (uint32_t)lds, s, e0, e1,
                        0u, c->dev(), ix->gram_dev(), n_sub, out,
                        sy, fin, disp_lds, (uint32_t)ix->sw.diag, dbg, strf, fs_scan_rows_shares(ix, sy.rinfo == nullptr));
  FS_HIP(hipGetLastError());
  return FS_OK;
}

}  // namespace

// Shape of a k_scan_rows launch: waves per workgroup (0: the kernel does not apply) and
// workgroups.  One workgroup of sixteen wave ranges per CU when searches run one at a
// time; an index with several lanes overlaps searches on the GPU, and two workgroups of
// eight per CU (when filter + seeds + per-wave state fit twice into the 160 KB of LDS)
// let one search's scan fill the stalls of another's records and hand-off.
uint32_t fs_scan_rows_shape(const fs_index* ix, const fs_corpus* c, uint32_t* blocks) {
  const uint32_t n = ix->cfg.window_size;
  *blocks = 0;
  if (!ix->sw.scan_rows || n < 2 || n > 8 || !c->d_ctab.p || !c->ctab_ready || !ix->ctab_ok) return 0;
  // (string ids of their own: the per-hit Levenshtein form, when the character classes exist)
  if (c->has_str && !(c->ctab_str && c->strrec_ready && ix->strfast_ok && ix->sw.str_fused && ix->sw.str_fast)) return 0;
  if (!c->has_str && c->ctab_str) return 0;
  // a switch that asks for one of the other scan kernels or paths
  const fs_switches& sw = ix->sw;
  if (sw.scan_simple || sw.scan_tpl == 4 || !sw.scan_direct || sw.scan_capw) return 0;
  // (1 KB: s_cnt; the slices of the shared rounds)
  const size_t fixed = ((size_t)4 << rows_filter_log2(ix)) + fs_scan_rows_disp_lds(ix) + 1024 + sizeof(fsdev::CoopLds);
  const size_t cu_lds = kCuLdsBytes;
  // a hit's LDS record holds its window position relative to the wave range in
  // fsdev::kHitPosBits bits: the longest range of the shape must fit (a wave of sixteen
  // takes at most 305/4096 of its workgroup's sub-tiles, else an equal share)
  auto fits = [&](uint32_t waves, uint32_t nblocks) {
    const uint64_t n_sub = (c->n_tok + 511) / 512;
    const uint64_t len = n_sub / nblocks + 1;
    const uint64_t longest = (waves == 16 && !(sw.diag & 16) ? (len * 512 + 4095) / 4096 : (len + waves - 1) / waves) + 1;   // (a share of at most a half of its quarter)
    return longest * 512 + 16 < (1ull << fsdev::kHitPosBits);
  };
  if (sw.rows_waves) {
    *blocks = (uint32_t)ix->num_cu * (uint32_t)std::max(1, sw.rows_blocks_per_cu);
    if (!fits((uint32_t)sw.rows_waves, *blocks)) { *blocks = 0; return 0; }
    return (uint32_t)sw.rows_waves;
  }
  if (sw.rows_blocks_per_cu != 1 && ix->n_lanes > 1 && 2 * (fixed + 8 * sizeof(FusedLds)) <= cu_lds &&
      fits(8, 2u * (uint32_t)ix->num_cu)) {
    *blocks = 2u * (uint32_t)ix->num_cu;
    return 8;
  }
  for (uint32_t w : {16u, 8u, 4u})
    if (fixed + w * sizeof(FusedLds) <= cu_lds && fits(w, (uint32_t)ix->num_cu)) { *blocks = (uint32_t)ix->num_cu; return w; }
  return 0;
}

// tokens -> output records: one launch of fs_scan_rows_blocks() workgroups of `waves` wave ranges
int fs_launch_scan_rows(fs_index* ix, fs_corpus* c, uint32_t waves, uint32_t blocks, uint32_t rcap, fs_row* d_rows,
                        int wire, uint32_t caprow, fs_status* host_st, hipStream_t s,
                        hipEvent_t e0, hipEvent_t e1, uint64_t* count_out, hipEvent_t done,
                        bool* done_attached) {
  fs_index::Lane& ln = *ix->cur;
  const uint32_t n_ranges = blocks * waves;
  const int stage_bytes = wire == 8 ? 8 : 16;          // (fs_row records are staged as 16-byte wire records)
  const uint32_t n_sub = (uint32_t)((c->n_tok + 511) / 512);
  FS_TRY(ln.w_stage.reserve((size_t)n_ranges * caprow * stage_bytes));
  fsdev::RowSync sy;
  FS_TRY(fs_row_sync(ix, blocks, c->n_tok, &sy));
  // shared rounds (the launch puts the records into place itself): a pool of records per
  // workgroup for the slices' output, grown when a search reports that it ran out
  uint32_t xpool = 0;
  if (!sy.rinfo && ix->sw.rows_coop) {
    xpool = std::max<uint32_t>(ix->sw.rows_xpool > 0 ? (uint32_t)ix->sw.rows_xpool : 4096u, ln.xpool_hint);
    FS_TRY(ln.w_xstage.reserve((size_t)blocks * xpool * 32));
  }
  ln.xpool = xpool;
  const fsdev::RangeOut out{ln.w_stage.p, caprow, wire, xpool ? ln.w_xstage.p : nullptr, xpool};
  // `done` (the search's completion event) rides on the search's last dispatch when that
  // dispatch has a free stop-event slot: k_compact, or this kernel when it is not timed
  if (done_attached) *done_attached = false;
  if (done && !sy.rinfo && !e1) { e1 = done; if (done_attached) *done_attached = true; }
  const fsdev::RowFinal fin{reinterpret_cast<uint8_t*>(d_rows), rcap, ln.d_status.p, host_st, count_out, true,
                            (ix->sw.diag & 512) != 0};
  switch (ix->cfg.window_size) {
    case 2: FS_TRY(launch_scan_rows<2>(ix, c, n_sub, waves, blocks, out, sy, fin, s, e0, e1)); break;
    case 3: FS_TRY(launch_scan_rows<3>(ix, c, n_sub, waves, blocks, out, sy, fin, s, e0, e1)); break;
    case 4: FS_TRY(launch_scan_rows<4>(ix, c, n_sub, waves, blocks, out, sy, fin, s, e0, e1)); break;
    case 5: FS_TRY(launch_scan_rows<5>(ix, c, n_sub, waves, blocks, out, sy, fin, s, e0, e1)); break;
    case 6: FS_TRY(launch_scan_rows<6>(ix, c, n_sub, waves, blocks, out, sy, fin, s, e0, e1)); break;
    case 7: FS_TRY(launch_scan_rows<7>(ix, c, n_sub, waves, blocks, out, sy, fin, s, e0, e1)); break;
    case 8: FS_TRY(launch_scan_rows<8>(ix, c, n_sub, waves, blocks, out, sy, fin, s, e0, e1)); break;
    default: fs_set_error("k_scan_rows covers n = 2..8"); return FS_E_UNSUPPORTED;
  }
  if (sy.rinfo) {
    if (done_attached) *done_attached = done != nullptr;
    return fs_launch_compact_after_scan_rows(ix, n_ranges, waves, caprow, wire, rcap, d_rows, host_st,
                                             s, count_out, done);
  }
  return FS_OK;
}

// ---- the floor under k_scan_rows (diagnostics) -----------------------------------------------
// k_stream_floor reads a corpus' ids the way k_scan_rows does -- one workgroup of sixteen waves
// per CU, a contiguous run of 512-token sub-tiles per wave, two 16-byte loads per lane and
// sub-tile, a pair of sub-tiles requested ahead -- and does nothing with them (an XOR, one word
// written per wave).  Its duration is what the launch shape itself costs for this many bytes:
// dispatch, ramp-up of 4096 waves, the HBM stream, completion.  fs_stream_floor times it the way
// bench.py times a search alone (events on each dispatch).
__global__ __launch_bounds__(1024) void k_stream_floor(const uint32_t* __restrict__ tok, uint32_t n_sub,
                                                       uint32_t* __restrict__ sink) {
  const uint32_t n_ranges = gridDim.x * 16, r = blockIdx.x * 16 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const uint32_t s0 = (uint32_t)((uint64_t)n_sub * r / n_ranges), s1 = (uint32_t)((uint64_t)n_sub * (r + 1) / n_ranges);
  uint4 acc = make_uint4(0, 0, 0, 0);
  uint32_t s = s0;
  for (; s + 2 <= s1; s += 2) {
    const uint4* p = reinterpret_cast<const uint4*>(tok + (size_t)s * 512 + 8 * lane);
    const uint4 a0 = p[0], a1 = p[1], b0 = p[128], b1 = p[129];
    acc.x ^= a0.x ^ a1.x ^ b0.x ^ b1.x; acc.y ^= a0.y ^ a1.y ^ b0.y ^ b1.y;
    acc.z ^= a0.z ^ a1.z ^ b0.z ^ b1.z; acc.w ^= a0.w ^ a1.w ^ b0.w ^ b1.w;
  }
  for (; s < s1; ++s) {
    const uint4* p = reinterpret_cast<const uint4*>(tok + (size_t)s * 512 + 8 * lane);
    const uint4 a0 = p[0], a1 = p[1];
    acc.x ^= a0.x ^ a1.x; acc.y ^= a0.y ^ a1.y; acc.z ^= a0.z ^ a1.z; acc.w ^= a0.w ^ a1.w;
  }
  uint32_t x = acc.x ^ acc.y ^ acc.z ^ acc.w;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) x ^= (uint32_t)__shfl_xor((int)x, d);
  if (lane == 0) sink[r] = x;
}

int fs_launch_stream_floor(fs_index* ix, fs_corpus* c, uint32_t reps, double* avg_ms) {
  const uint32_t n_sub = (uint32_t)(c->n_tok / 512);             // whole sub-tiles (the rest is noise)
  const uint32_t blocks = (uint32_t)ix->num_cu;
  FS_TRY(ix->cur->w_bsum.reserve(blocks * 16));
  hipStream_t s = ix->stream;
  hipEvent_t e0, e1;
  FS_HIP(hipEventCreate(&e0));
  FS_HIP(hipEventCreate(&e1));
  double sum = 0;
  for (uint32_t r = 0; r < reps + 2; ++r) {
    hipExtLaunchKernelGGL(k_stream_floor, dim3(blocks), dim3(1024), 0, s, e0, e1, 0u, c->dev().tok, n_sub, ix->cur->w_bsum.p);
    FS_HIP(hipGetLastError());
    FS_HIP(hipEventSynchronize(e1));
    float ms = 0;
    FS_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 2) sum += ms;                                        // (two launches to warm up)
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *avg_ms = sum / reps;
  return FS_OK;
}

// ---- k_scan_near (integer prefilter of the LSH pipeline) --------------------------------
// 0: none; 1: at most one slot of a neighbour may differ in its vector id (k_scan_near over
// the vector ids); 2: tables with near-synonyms, the same over component ids (fs_lsh.hip)
int fs_lsh_prefilter_mode(const fs_index* ix, const fs_corpus* c) {
  const int n = (int)ix->cfg.window_size;
  if (!ix->sw.lsh_prefilter || c->has_oov || ix->script_oov) return 0;
  if (n - ix->lsh_m_min == 1 && ix->d_sfilter3.p)
    return n == 7 || n == 8 || n == 9 || n == 10 || n == 12 ? 1 : 0;
  if (ix->syn_ok && c->ctok_ready && ix->sw.lsh_syn && ix->d_sfilter3c.p)
    return n == 6 || n == 7 || n == 8 || n == 9 || n == 10 || n == 12 ? 2 : 0;
  return 0;
}
bool fs_lsh_prefilter_ok(const fs_index* ix, const fs_corpus* c) { return fs_lsh_prefilter_mode(ix, c) != 0; }
int fs_scan_near_k(int n) { return fs_near_k(n); }
// n >= 7: the eight-tokens-per-lane form (k_scan_near8: polynomial 3-gram hash, at most 2^14
// filter words, k_scan8's bitmap); n = 6 keeps k_scan_near with its second filter
// (decided when the filters are built, fs_lsh_build: the hash has to match)
bool fs_scan_near8_wanted(const fs_index* ix) {
  const int n = (int)ix->cfg.window_size;
  // (n = 6 in this form only for k_near_sift: by itself k_scan_near<6> with its second filter is
  // the better prefilter there)
  return ix->sw.scan_near8 && (n >= 7 || (n == 6 && ix->sw.near_fused)) && (n <= 10 || n == 12);
}
// k_near_sift (prefilter + wildcard filter in one kernel, lists per wave range) takes the search
bool fs_near_fused(const fs_index* ix, const fs_corpus* c) {
  return ix->sw.near_fused && fs_scan_near8(ix) && fs_lsh_prefilter_ok(ix, c);
}
uint32_t fs_near_ranges() { return (uint32_t)fsdev::kNB * 4u; }
bool fs_scan_near8(const fs_index* ix) { return ix->near8; }
int fs_scan_near_log2(const fs_index* ix) {
  return fs_scan_near8(ix) ? std::min(ix->log2_words, FS_SUB_MAX_LOG2_WORDS) : ix->log2_words;
}
// bit of the 3-gram t[0..3) in the filter k_scan_near / k_scan_near8 hold in LDS
void fs_scan_near_bit(const fs_index* ix, const uint32_t* t, uint32_t* word, uint32_t* bit) {
  const int K = fs_near_k((int)ix->cfg.window_size);
  if (fs_scan_near8(ix)) {
    const uint32_t h = fs_sub_hash(t, K);
    *word = fs_sub_word(h, fs_scan_near_log2(ix));
    *bit = fs_sub_bit(h);
  } else {
    const uint32_t h = fs_gram_hash(t, K);
    *word = fs_bloom_word(h, ix->log2_words);
    *bit = h & 31;
  }
}

namespace {
template <int N>
int launch_scan_near(const fs_index* ix, const CorpusDev& c, const uint32_t* ids, const uint32_t* filter,
                     const uint32_t* keys6, uint64_t* qbm, uint32_t* qcnt,
                     uint32_t n_bm_words, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
  const uint32_t tile_tok = kSubTile * 2;
  const uint32_t n_tiles = (uint32_t)(((uint64_t)c.n_tok + tile_tok - 1) / tile_tok);
  if (n_tiles == 0) return FS_OK;
  const size_t lds = ((size_t)4 << ix->log2_words) + (N == 6 && keys6 ? (size_t)4 << FS_NEAR6_LOG2_WORDS : 0);
  const uint32_t max_blocks = ix->num_cu * (lds <= 64 * 1024 ? 2 : 1);
  const uint32_t blocks = std::min<uint32_t>((n_tiles + 15) / 16, max_blocks);
  const bool nt = (uint64_t)c.n_tok * 4 > (256ull << 20);
  auto kern = nt ? k_scan_near<N, true> : k_scan_near<N, false>;
  FS_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), ix->device, lds));
  hipExtLaunchKernelGGL(kern, dim3(blocks), dim3(1024), (uint32_t)lds, s, e0, e1, 0u, ids, c.n_tok,
                        filter, ix->log2_words, qbm, qcnt, n_bm_words, n_tiles, N == 6 ? keys6 : (const uint32_t*)nullptr);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

template <int N>
int launch_scan_near8(const fs_index* ix, const CorpusDev& c, const uint32_t* ids, const uint32_t* filter,
                      uint64_t* qbm, uint32_t* qcnt, uint32_t n_sub, hipStream_t s, hipEvent_t e0,
                      hipEvent_t e1, fs_scan_extra* ex) {
  const int lw = fs_scan_near_log2(ix);
  const size_t lds = ((size_t)4 << lw) + 16 * sizeof(uint32_t);
  // the chained kernels' chunking (chunk_of_block): kNB chunks of `chunk` sub-tiles
  const uint32_t chunk = std::max<uint32_t>(1, (n_sub + fsdev::kNB - 1) / fsdev::kNB);
  auto kern = k_scan_near8<N>;
  FS_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), ix->device, lds));
  hipExtLaunchKernelGGL(kern, dim3(kScanBlocks), dim3(1024), (uint32_t)lds, s, e0, e1, 0u, ids, c.n_tok,
                        filter, lw, qbm, qcnt, n_sub, chunk, ex->bsum, ex->zero);
  FS_HIP(hipGetLastError());
  ex->counted = true;
  return FS_OK;
}
template <int N>
int launch_near_sift(const fs_index* ix, const CorpusDev& c, const uint32_t* ids, const uint32_t* filter,
                     const uint32_t* wild, int log2_wild, uint32_t* slist, uint32_t caps, uint32_t* scount,
                     uint32_t* bsum, fs_status* zero, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
  const int lw = fs_scan_near_log2(ix);
  const uint32_t n_sub = (uint32_t)(((uint64_t)c.n_tok + 511) / 512);
  const size_t lds = ((size_t)4 << lw) + 16 + 16 * sizeof(SiftLds);
  const uint32_t chunk = std::max<uint32_t>(1, (n_sub + fsdev::kNB - 1) / fsdev::kNB);
  auto kern = k_near_sift<N>;
  FS_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), ix->device, lds));
  hipExtLaunchKernelGGL(kern, dim3(kScanBlocks), dim3(1024), (uint32_t)lds, s, e0, e1, 0u, ids, c.n_tok,
                        filter, lw, wild, log2_wild, n_sub, chunk, slist, caps, scount, bsum, zero);
  FS_HIP(hipGetLastError());
  return FS_OK;
}
}  // namespace

int fs_launch_near_sift(const fs_index* ix, const fs_corpus* fc, uint32_t* slist, uint32_t caps, uint32_t* scount,
                        uint32_t* bsum, fs_status* zero, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
  const CorpusDev c = fc->dev();
  const bool comp = fs_lsh_prefilter_mode(ix, fc) == 2;
  const uint32_t* ids = comp ? (const uint32_t*)fc->d_ctok.p : c.tok;
  const uint32_t* filter = comp ? (const uint32_t*)ix->d_sfilter3c.p : (const uint32_t*)ix->d_sfilter3.p;
  const uint32_t *wild = nullptr, *wild_tok = nullptr;
  int log2_wild = 0;
  fs_lsh_wild_of(ix, fc, &wild, &log2_wild, &wild_tok);
  if (wild && (wild_tok ? wild_tok != ids : comp)) { fs_set_error("k_near_sift: the keys' ids are not the scanned ids"); return FS_E_DEVICE; }
  switch (ix->cfg.window_size) {
#define FS_NS(N) case N: return launch_near_sift<N>(ix, c, ids, filter, wild, log2_wild, slist, caps, scount, bsum, zero, s, e0, e1)
    FS_NS(6); FS_NS(7); FS_NS(8); FS_NS(9); FS_NS(10); FS_NS(12);
#undef FS_NS
    default: fs_set_error("k_near_sift covers n = 6..10, 12"); return FS_E_UNSUPPORTED;
  }
}

// ex: chunk sums and status block for the eight-tokens-per-lane form (ex->counted on return:
// k_expand needs no k_reduce in front)
int fs_launch_scan_near(const fs_index* ix, const fs_corpus* fc, uint64_t* qbm, uint32_t* qcnt,
                        uint32_t n_bm_words, hipStream_t s, hipEvent_t e0, hipEvent_t e1,
                        fs_scan_extra* ex) {
  const CorpusDev c = fc->dev();
  const bool comp = fs_lsh_prefilter_mode(ix, fc) == 2;
  const uint32_t* ids = comp ? (const uint32_t*)fc->d_ctok.p : c.tok;
  const uint32_t* filter = comp ? (const uint32_t*)ix->d_sfilter3c.p : (const uint32_t*)ix->d_sfilter3.p;
  const uint32_t* keys6 = comp && ix->sw.lsh_keys6 ? (const uint32_t*)ix->d_keys6c.p : (const uint32_t*)nullptr;
  // the second filter of k_scan_near<6> is optional (a superset test in front of a superset
  // test): with a 3-gram filter of 2^15 words the two together exceed the CU's LDS
  if (((size_t)4 << ix->log2_words) + ((size_t)4 << FS_NEAR6_LOG2_WORDS) > kCuLdsBytes) keys6 = nullptr;
  ex->counted = false;
  if (fs_scan_near8(ix)) {
    switch (ix->cfg.window_size) {
      case 6: return launch_scan_near8<6>(ix, c, ids, filter, qbm, qcnt, n_bm_words, s, e0, e1, ex);
      case 7: return launch_scan_near8<7>(ix, c, ids, filter, qbm, qcnt, n_bm_words, s, e0, e1, ex);
      case 8: return launch_scan_near8<8>(ix, c, ids, filter, qbm, qcnt, n_bm_words, s, e0, e1, ex);
      case 9: return launch_scan_near8<9>(ix, c, ids, filter, qbm, qcnt, n_bm_words, s, e0, e1, ex);
      case 10: return launch_scan_near8<10>(ix, c, ids, filter, qbm, qcnt, n_bm_words, s, e0, e1, ex);
      case 12: return launch_scan_near8<12>(ix, c, ids, filter, qbm, qcnt, n_bm_words, s, e0, e1, ex);
      default: break;
    }
  }
  switch (ix->cfg.window_size) {
    case 6: return launch_scan_near<6>(ix, c, ids, filter, keys6, qbm, qcnt, n_bm_words, s, e0, e1);
    case 7: return launch_scan_near<7>(ix, c, ids, filter, keys6, qbm, qcnt, n_bm_words, s, e0, e1);
    case 8: return launch_scan_near<8>(ix, c, ids, filter, keys6, qbm, qcnt, n_bm_words, s, e0, e1);
    case 9: return launch_scan_near<9>(ix, c, ids, filter, keys6, qbm, qcnt, n_bm_words, s, e0, e1);
    case 10: return launch_scan_near<10>(ix, c, ids, filter, keys6, qbm, qcnt, n_bm_words, s, e0, e1);
    case 12: return launch_scan_near<12>(ix, c, ids, filter, keys6, qbm, qcnt, n_bm_words, s, e0, e1);
    default: fs_set_error("k_scan_near covers n = 6..10, 12"); return FS_E_UNSUPPORTED;
  }
}
