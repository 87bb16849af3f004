// fs_ranges.hip -- everything between the scan's candidate records and the output
// records of the exact pipeline, one wave per wave range, in ONE kernel
// (k_ranges) followed by a copy into place (k_compact).  Replaces the chain
// k_verify_direct -> k_hitrows -> k_rows of fs_post.hip for corpora whose string
// ids equal their vector ids (no per-candidate Levenshtein), and with it the
// per-candidate arrays cpos / cg / cw / hv / hcomb and three grid-wide prefix sums.
//
// Reference semantics (file:line in /root/reference), as in fs_post.hip:
//   search.py:182-184  kept candidates = script windows with the fan window's ids
//   search.py:192-218  n word records per match
//   search.py:224-226  per fan word the FIRST record of minimal dist*lev over the
//                      windows covering it, output ascending by word index
//
// A wave range is a contiguous run of 512-token sub-tiles scanned by one wave of
// k_scan8 (fs_scan.hip); its candidate records {(position / 8) << 8 | flag byte, rank}
// sit in one list.  The record of fan word x depends only on the hits among windows
// x-n+1 .. x, so a range's records follow from its own candidates plus the n-1
// windows in front of it (the "halo", verified unconditionally): no data crosses
// ranges.  Per round of at most 64-(n-1) candidates the wave
//   1. spreads the records' flag bits over its lanes (one candidate per lane, position
//      order), in LDS
//   2. verifies every candidate: ids -> hash -> displacement -> table entry, compared id
//      for id, and the work boundary (three levels of loads, all lanes in flight
//      together); the per-corpus table `sbest` (indexed by table slot, so fetched
//      beside the entry) carries the best rank's record of the matched n-gram
//   3. compacts the hits behind the <= n-1 hits carried over from the last round
//   4. emits the records of the words in [E, F): E = words done so far, F = the
//      first position whose hit status is not known yet (the next round's first
//      candidate, or the end of the range).  Hit j is the first to cover the words
//      [max(p_j, p_{j-1} + n), p_j + n); one lane per word walks the <= n-1 later
//      hits that also cover it for the first minimum of the combined distance
//   5. keeps the hits that may still cover words >= F.
// Rows go to a staging area of `caprow` records per range (position order inside a
// range = output order); k_compact turns the per-range counts into offsets (chunk
// sums written by k_ranges, one block prefix) and copies the records into place,
// publishes the totals and the status block.  Nothing here depends on dispatch
// order: no atomics on the data path, no inter-workgroup hand-off inside a launch.
#include "fs_ranges.h"

#include <algorithm>

namespace {

using namespace fsdev;

constexpr int kRangeWaves = 4;              // ranges (waves) per block of k_ranges
constexpr int kRanges = FS_CHUNKS * 4;      // wave ranges of a k_scan8 launch

template <int N>
__global__ __launch_bounds__(kRangeWaves * 64) void k_ranges(
    CorpusDev c, GramIndexDev g, const fs_best* __restrict__ sbest,
    const uint2* __restrict__ recs, const uint2* __restrict__ info, uint32_t capw,
    uint32_t n_sub, uint32_t chunk, RangeOut out, uint4* __restrict__ rinfo,
    uint4* __restrict__ csum, uint32_t* __restrict__ cmax, fs_status* st) {
  static_assert(N >= 2 && N <= 8, "the eight-tokens-per-lane scan covers n <= 8");
  constexpr uint32_t HALO = N - 1, RS = 64 - HALO;
  __shared__ RangeLds s_all[kRangeWaves];
  __shared__ uint32_t s_sum[kRangeWaves][3];
  const int lane = threadIdx.x & 63;
  const uint32_t wv = threadIdx.x >> 6;
  RangeLds& S = s_all[wv];
  const uint32_t range_id = blockIdx.x * kRangeWaves + wv;
  // the scan's geometry (k_scan8): chunk = range_id / 4, quarter = range_id % 4
  const uint32_t q = range_id & 3;
  const uint32_t per = (chunk + 3) >> 2;
  const uint64_t chunk_first = (uint64_t)(range_id >> 2) * chunk;
  uint32_t j0 = q * per, j1 = j0 + per;
  if (j1 > chunk) j1 = chunk;
  if (j0 > j1) j0 = j1;
  if (chunk_first + j1 > n_sub) j1 = chunk_first + j0 < n_sub ? (uint32_t)(n_sub - chunk_first) : j0;

  RangeState R;
  R.E = 0; R.hc = 0; R.rows_run = 0; R.hits_run = 0; R.match_acc = 0;
  if (j1 > j0) {
    const uint32_t a = (uint32_t)(chunk_first + j0) * 512u;        // tokens [a, bnd)
    const uint32_t bnd = (uint32_t)(chunk_first + j1) * 512u;
    const uint2 inf = info[range_id];
    uint32_t nrec = inf.x;
    const uint32_t ncand = inf.y;
    if (nrec > capw) {                              // the host repeats the search with longer lists
      if (lane == 0) atomicMax(&st->max_recs, nrec);
      nrec = capw;
    }
    const uint2* list = recs + (size_t)range_id * capw;
    const uint32_t halo_n = a ? HALO : 0;
    const uint32_t total = halo_n + ncand;
    R.E = a;
    for (uint32_t r0 = 0; r0 < total; r0 += RS) {
      const uint32_t m = total - r0 < RS ? total - r0 : RS;
      // 1. candidates r0 .. r0 + RS (the last one only as the next round's first)
      S.cand[lane] = FS_NONE;
      wave_sync();
      if ((uint32_t)lane < halo_n && (uint32_t)lane >= r0 && (uint32_t)lane - r0 <= RS)
        S.cand[lane - r0] = a - halo_n + lane;
      for (uint32_t t = lane; t < nrec; t += 64) {
        const uint2 rec = list[t];
        uint32_t flags = rec.x & 0xFFu;
        const uint32_t p0 = (rec.x >> 8) << 3;
        uint32_t li = halo_n + rec.y;
        while (flags) {
          const int bb = __ffs(flags) - 1;
          flags &= flags - 1;
          if (li >= r0 && li - r0 <= RS) S.cand[li - r0] = p0 + (uint32_t)bb;
          ++li;
        }
      }
      wave_sync();
      uint32_t F = bnd;
      if (r0 + RS < total) {
        const uint32_t nx = S.cand[RS];
        if (nx < F) F = nx;
      }
      range_round<N>(c, g, sbest, S, m, F, a, range_id, out, R);
    }
    if (R.rows_run > out.caprow && lane == 0) atomicMax(&st->max_rows, R.rows_run);
  }
  // {records, hits, (window, script window) pairs, -} per range and per block, and the
  // block's largest record count of a range
  uint32_t match = R.match_acc;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) match += (uint32_t)__shfl_xor((int)match, d);
  if (lane == 0) {
    rinfo[range_id] = make_uint4(R.rows_run, R.hits_run, match, 0);
    s_sum[wv][0] = R.rows_run; s_sum[wv][1] = R.hits_run; s_sum[wv][2] = match;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint4 t = make_uint4(0, 0, 0, 0);
    uint32_t mx = 0;
    for (int i = 0; i < kRangeWaves; ++i) {
      t.x += s_sum[i][0]; t.y += s_sum[i][1]; t.z += s_sum[i][2];
      mx = s_sum[i][0] > mx ? s_sum[i][0] : mx;
    }
    csum[blockIdx.x] = t;
    cmax[blockIdx.x] = mx;
  }
}

// Staged records -> their place in the output.  Block b takes the kCompactRanges wave
// ranges [16 b, 16 b + 16): its first record index is the sum of the block sums
// (`csum`, one per `csum_per` ranges) in front of it.  Every thread first requests all
// its pieces (16 bytes, or 8 for the 8-byte wire records), then stores them.  The extra
// block publishes totals and status.
constexpr int kCompactRanges = 16;

template <class V>
__device__ __forceinline__ void compact_copy(const uint8_t* __restrict__ stage,
                                             uint8_t* __restrict__ rows, uint32_t caprow,
                                             uint32_t rec_bytes, uint32_t range0, uint32_t base,
                                             const uint32_t* s_off, const uint32_t* s_poff) {
  const uint32_t P = s_poff[kCompactRanges];
  // piece i of the block: its place in the staging area and in the output
  auto locate = [&](uint32_t i, const V** src, V** dst) {
    uint32_t r = 0;                                        // last range with s_poff[r] <= i
#pragma unroll
    for (int step = kCompactRanges / 2; step > 0; step >>= 1)
      if (s_poff[r + step] <= i) r += step;
    const uint32_t k = i - s_poff[r];
    *src = reinterpret_cast<const V*>(stage + (size_t)(range0 + r) * caprow * rec_bytes) + k;
    *dst = reinterpret_cast<V*>(rows + (size_t)(base + s_off[r]) * rec_bytes) + k;
  };
  // four pieces per thread requested together (named registers: an array here is moved
  // to LDS by the compiler, with a wait behind every load)
  uint32_t i = threadIdx.x;
  for (; i + 3 * kThreads < P; i += 4 * kThreads) {
    const V *s0, *s1, *s2, *s3;
    V *d0, *d1, *d2, *d3;
    locate(i, &s0, &d0); locate(i + kThreads, &s1, &d1);
    locate(i + 2 * kThreads, &s2, &d2); locate(i + 3 * kThreads, &s3, &d3);
    const V v0 = *s0, v1 = *s1, v2 = *s2, v3 = *s3;
    *d0 = v0; *d1 = v1; *d2 = v2; *d3 = v3;
  }
  for (; i < P; i += kThreads) {
    const V* s0;
    V* d0;
    locate(i, &s0, &d0);
    *d0 = *s0;
  }
}

__global__ __launch_bounds__(kThreads) void k_compact(
    const uint4* __restrict__ rinfo, const uint4* __restrict__ csum,
    const uint32_t* __restrict__ cmax, uint32_t csum_per, uint32_t n_ranges,
    const uint32_t* __restrict__ cand_sums, uint32_t n_cand_sums, bool fresh,
    const uint8_t* __restrict__ stage, uint32_t caprow, int rec_bytes, uint32_t rcap,
    uint8_t* __restrict__ rows, fs_status* st, fs_status* host_st, uint64_t* count_out) {
  __shared__ uint32_t s_w[4];
  __shared__ uint32_t s_w2[4];
  __shared__ uint32_t s_n[kCompactRanges], s_off[kCompactRanges + 1], s_poff[kCompactRanges + 1];
  const uint32_t n_blocks = n_ranges / kCompactRanges;
  if (blockIdx.x == n_blocks) {
    // the extra block: totals over all block sums, largest range, status
    const uint32_t n_csum = n_ranges / csum_per;
    uint32_t rws = 0, hits = 0, mt = 0, cands = 0, mx = 0;
    // eight loads requested together, in named registers (an array indexed in an
    // unrolled loop is moved to LDS by the compiler, with a wait behind every load)
    auto ld4 = [&](uint32_t i) { return i < n_csum ? csum[i] : make_uint4(0, 0, 0, 0); };
    auto ldm = [&](uint32_t i) { return i < n_csum ? cmax[i] : 0u; };
    auto ldc = [&](uint32_t i) { return i < n_cand_sums ? cand_sums[i] : 0u; };
    for (uint32_t i0 = threadIdx.x; i0 < n_csum; i0 += kThreads * 4) {
      const uint4 a = ld4(i0), b = ld4(i0 + kThreads), c = ld4(i0 + 2 * kThreads), d = ld4(i0 + 3 * kThreads);
      const uint32_t ma = ldm(i0), mb = ldm(i0 + kThreads), mc = ldm(i0 + 2 * kThreads), md = ldm(i0 + 3 * kThreads);
      rws += a.x + b.x + c.x + d.x; hits += a.y + b.y + c.y + d.y;
      mt += a.z + b.z + c.z + d.z; cands += a.w + b.w + c.w + d.w;
      const uint32_t m1 = ma > mb ? ma : mb, m2 = mc > md ? mc : md;
      mx = mx > m1 ? mx : m1; mx = mx > m2 ? mx : m2;
    }
    if (cand_sums) {
      cands = 0;
      for (uint32_t i0 = threadIdx.x; i0 < n_cand_sums; i0 += kThreads * 8)
        cands += ldc(i0) + ldc(i0 + kThreads) + ldc(i0 + 2 * kThreads) + ldc(i0 + 3 * kThreads) +
                 ldc(i0 + 4 * kThreads) + ldc(i0 + 5 * kThreads) + ldc(i0 + 6 * kThreads) + ldc(i0 + 7 * kThreads);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      const uint32_t o = (uint32_t)__shfl_xor((int)mx, d);
      mx = o > mx ? o : mx;
    }
    if ((threadIdx.x & 63) == 0) s_n[threadIdx.x >> 6] = mx;
    uint32_t tr, th, tm, tc;
    block_excl_scan(rws, s_w, &tr);
    block_excl_scan(hits, s_w2, &th);
    block_excl_scan(mt, s_w, &tm);
    block_excl_scan(cands, s_w2, &tc);
    if (threadIdx.x == 0) {
      fs_status out;
      if (fresh) {                    // k_scan_rows: nothing before this kernel wrote the block
        out.max_recs = 0; out.lev_overflow = 0; out.bad_string = 0;
      } else {
        out = *st;                    // flags and maxima written by the kernels before
      }
      mx = 0;
      for (int i = 0; i < kThreads / 64; ++i) mx = s_n[i] > mx ? s_n[i] : mx;
      out.n_cands = tc; out.n_hits = th; out.n_matches = tm; out.n_rows = tr;
      out.max_rows = mx > caprow ? mx : 0;
      *st = out;
      *host_st = out;
      if (count_out) *count_out = tr;       // FS_ROWS_HEADER
    }
    return;
  }
  const uint32_t nb_before = blockIdx.x * (kCompactRanges / csum_per);   // block sums in front
  if (threadIdx.x < kCompactRanges)
    s_n[threadIdx.x] = rinfo[blockIdx.x * kCompactRanges + threadIdx.x].x;
  uint32_t pre = 0;
  {
    auto ldx = [&](uint32_t i) { return i < nb_before ? csum[i].x : 0u; };
    for (uint32_t i0 = threadIdx.x; i0 < nb_before; i0 += kThreads * 8)
      pre += ldx(i0) + ldx(i0 + kThreads) + ldx(i0 + 2 * kThreads) + ldx(i0 + 3 * kThreads) +
             ldx(i0 + 4 * kThreads) + ldx(i0 + 5 * kThreads) + ldx(i0 + 6 * kThreads) + ldx(i0 + 7 * kThreads);
  }
  uint32_t base;
  block_excl_scan(pre, s_w, &base);
  const uint32_t piece = rec_bytes == 8 ? 8 : 16;
  if (threadIdx.x == 0) {
    uint32_t acc = 0, pacc = 0;
    for (int r = 0; r < kCompactRanges; ++r) {
      s_off[r] = acc; s_poff[r] = pacc;
      uint32_t n = s_n[r] < caprow ? s_n[r] : caprow;
      if (base + acc >= rcap) n = 0;                       // beyond the caller's buffer
      else if (base + acc + n > rcap) n = rcap - (base + acc);
      acc += n;
      pacc += n * (rec_bytes / piece);
    }
    s_off[kCompactRanges] = acc; s_poff[kCompactRanges] = pacc;
  }
  __syncthreads();
  if (piece == 16)
    compact_copy<uint4>(stage, rows, caprow, rec_bytes, blockIdx.x * kCompactRanges, base, s_off, s_poff);
  else
    compact_copy<uint2>(stage, rows, caprow, rec_bytes, blockIdx.x * kCompactRanges, base, s_off, s_poff);
}

// per-corpus copy of the best record of every script n-gram, indexed by table slot
__global__ void k_sbest(GramIndexDev g, const fs_best* __restrict__ gbest, fs_best* __restrict__ sbest) {
  const uint32_t slots = 1u << g.log2_slots;
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < slots; s += gridDim.x * blockDim.x) {
    const uint32_t e0 = g.table[(size_t)s * g.tstride];
    fs_best b;
    b.s = 0; b.lev = 0; b.dist = 0.0; b.comb = 0.0; b.pad = 0.0;
    if (e0) b = gbest[e0 - 1];
    sbest[s] = b;
  }
}

}  // namespace

int fs_launch_sbest(fs_index* ix, fs_corpus* c, hipStream_t s) {
  const size_t slots = (size_t)1 << ix->log2_slots;
  FS_TRY(c->d_sbest.reserve(slots));
  const uint32_t blocks = (uint32_t)std::min<size_t>((slots + 255) / 256, 1024);
  hipLaunchKernelGGL(k_sbest, dim3(blocks), dim3(256), 0, s, ix->gram_dev(), c->d_gbest.p,
                     c->d_sbest.p);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

bool fs_ranges_ok(const fs_index* ix, const fs_corpus* c) {
  const uint32_t n = ix->cfg.window_size;
  return ix->sw.post_ranges && n >= 2 && n <= 8 && !c->has_str && c->d_sbest.p != nullptr;
}

// staged records of `n_ranges` wave ranges -> the caller's buffer, totals, status
int fs_launch_compact(fs_index* ix, uint32_t n_ranges, uint32_t csum_per, uint32_t caprow,
                      int rec_bytes, uint32_t rcap, fs_row* d_rows, fs_status* host_st,
                      hipStream_t s, uint64_t* count_out, bool fresh, const uint32_t* cand_sums,
                      uint32_t n_cand_sums) {
  fs_index::Lane& ln = *ix->cur;
  if (n_ranges % kCompactRanges || kCompactRanges % csum_per) {
    fs_set_error("k_compact: %u ranges, %u per block sum", n_ranges, csum_per);
    return FS_E_INVALID;
  }
  hipLaunchKernelGGL(k_compact, dim3(n_ranges / kCompactRanges + 1), dim3(kThreads), 0, s,
                     ln.w_rinfo.p, ln.w_csum.p,
                     reinterpret_cast<const uint32_t*>(ln.w_csum.p + n_ranges / csum_per), csum_per,
                     n_ranges, cand_sums, n_cand_sums, fresh, ln.w_stage.p, caprow, rec_bytes, rcap,
                     reinterpret_cast<uint8_t*>(d_rows), ln.d_status.p, host_st, count_out);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

// scan records -> output records: k_ranges + k_compact (see the head of this file)
int fs_launch_ranges(fs_index* ix, fs_corpus* c, uint32_t n_sub, uint32_t rcap, fs_row* d_rows,
                     int wire, uint32_t caprow, fs_status* host_st, hipStream_t s,
                     const fs_scan_extra& scan, uint64_t* count_out) {
  fs_index::Lane& ln = *ix->cur;
  const GramIndexDev g = ix->gram_dev();
  const CorpusDev cd = c->dev();
  const int rec_bytes = wire ? wire : 32;
  const uint32_t chunk = std::max<uint32_t>(1, (n_sub + FS_CHUNKS - 1) / FS_CHUNKS);
  FS_TRY(ln.w_stage.reserve((size_t)kRanges * caprow * rec_bytes));
  FS_TRY(ln.w_rinfo.reserve(kRanges));
  FS_TRY(ln.w_csum.reserve(2 * kRanges / kRangeWaves));
  const dim3 grid(kRanges / kRangeWaves), block(kRangeWaves * 64);
  const RangeOut out{ln.w_stage.p, caprow, wire};
#define FS_RANGES(NN)                                                                           \
  hipLaunchKernelGGL((k_ranges<NN>), grid, block, 0, s, cd, g, c->d_sbest.p, scan.recs,         \
                     scan.info, scan.capw, n_sub, chunk, out, ln.w_rinfo.p, ln.w_csum.p,        \
                     reinterpret_cast<uint32_t*>(ln.w_csum.p + kRanges / kRangeWaves),         \
                     ln.d_status.p)
  switch (ix->cfg.window_size) {
    case 2: FS_RANGES(2); break;
    case 3: FS_RANGES(3); break;
    case 4: FS_RANGES(4); break;
    case 5: FS_RANGES(5); break;
    case 6: FS_RANGES(6); break;
    case 7: FS_RANGES(7); break;
    case 8: FS_RANGES(8); break;
    default: fs_set_error("k_ranges covers n = 2..8"); return FS_E_UNSUPPORTED;
  }
#undef FS_RANGES
  FS_HIP(hipGetLastError());
  return fs_launch_compact(ix, kRanges, kRangeWaves, caprow, rec_bytes, rcap, d_rows, host_st, s,
                           count_out, /*fresh=*/false, scan.bsum, FS_CHUNKS);
}
