// fs_ranges.hip -- everything between the scan's candidate records and the output
// records of the exact pipeline, one wave per wave range, in ONE kernel
// (k_ranges).  Replaces the chain
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
// range = output order).  At the end of the launch every workgroup publishes its
// record count, sums the counts of the workgroups in front of it and copies its staged
// records into place (finish_rows, fs_ranges.h: workgroups take their ranges in ticket
// order, so nothing depends on dispatch order or residency); the last one publishes
// totals and status.  The records themselves never depend on timing: their values and
// their order are functions of the token positions alone.
#include "fs_ranges.h"

#include <algorithm>

namespace {

using namespace fsdev;

constexpr int kRangeWaves = 16;             // ranges (waves) per workgroup of k_ranges
constexpr int kRanges = FS_CHUNKS * 4;      // wave ranges of a k_scan8 launch

template <int N>
__global__ __launch_bounds__(kRangeWaves * 64) void k_ranges(
    CorpusDev c, GramIndexDev g, const fs_best* __restrict__ sbest,
    const uint2* __restrict__ recs, const uint2* __restrict__ info, uint32_t capw,
    uint32_t n_sub, uint32_t chunk, RangeOut out, RowSync sy, RowFinal fin) {
  static_assert(N >= 2 && N <= 8, "the eight-tokens-per-lane scan covers n <= 8");
  constexpr uint32_t HALO = N - 1, RS = 64 - HALO;
  __shared__ RangeLds s_all[kRangeWaves];
  __shared__ uint32_t s_cnt[5 * kRangeWaves + 2];
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
  uint32_t ncand = 0;
  if (j1 > j0) {
    const uint32_t a = (uint32_t)(chunk_first + j0) * 512u;        // tokens [a, bnd)
    const uint32_t bnd = (uint32_t)(chunk_first + j1) * 512u;
    // counts and the first 64 records of the list are requested together
    const uint2* list = recs + (size_t)range_id * capw;
    const uint2 inf = info[range_id];
    uint2 rec0 = make_uint2(0, 0);
    if ((uint32_t)lane < capw) rec0 = list[lane];
    uint32_t nrec = inf.x;
    ncand = inf.y;
    if (nrec > capw) {                              // the host repeats the search with longer lists
      if (lane == 0) atomicMax(&fin.st->max_recs, nrec);
      nrec = capw;
    }
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
        const uint2 rec = t < 64 ? rec0 : list[t];
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
  }
  uint32_t match = R.match_acc;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) match += (uint32_t)__shfl_xor((int)match, d);
  finish_rows(sy, fin, out, range_id, R.rows_run, R.hits_run, match, ncand, s_cnt);
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

// The lane's hand-off words for one launch of `n_blocks` workgroups that end in
// finish_rows: granules and statistics per workgroup, next epoch.
int fs_row_sync(fs_index* ix, uint32_t n_blocks, fsdev::RowSync* sy) {
  fs_index::Lane& ln = *ix->cur;
  if (n_blocks > FS_SYNC_BLOCKS) { fs_set_error("%u workgroups in a records kernel", n_blocks); return FS_E_INVALID; }
  if (!ln.w_gran.p) {
    FS_TRY(ln.w_gran.reserve(FS_SYNC_BLOCKS));
    FS_TRY(ln.w_bstat.reserve(FS_SYNC_BLOCKS));
    FS_HIP(hipMemsetAsync(ln.w_gran.p, 0, FS_SYNC_BLOCKS * sizeof(unsigned long long), ln.stream));
    ln.sync_epoch = 0;
  }
  if (++ln.sync_epoch == 0) ln.sync_epoch = 1;     // 0 is the tag of a granule never written
  sy->gran = ln.w_gran.p;
  sy->bstat = ln.w_bstat.p;
  sy->epoch = ln.sync_epoch;
  sy->n_blocks = n_blocks;
  sy->spin_limit = ix->sw.wait_spins >= 0 ? (uint32_t)ix->sw.wait_spins : (1u << 22);
  return FS_OK;
}

// scan records -> output records in place: k_ranges (see the head of this file)
int fs_launch_ranges(fs_index* ix, fs_corpus* c, uint32_t n_sub, uint32_t rcap, fs_row* d_rows,
                     int wire, uint32_t caprow, fs_status* host_st, hipStream_t s,
                     const fs_scan_extra& scan, uint64_t* count_out) {
  fs_index::Lane& ln = *ix->cur;
  const GramIndexDev g = ix->gram_dev();
  const CorpusDev cd = c->dev();
  const int rec_bytes = wire ? wire : 32;
  const uint32_t chunk = std::max<uint32_t>(1, (n_sub + FS_CHUNKS - 1) / FS_CHUNKS);
  FS_TRY(ln.w_stage.reserve((size_t)kRanges * caprow * rec_bytes));
  const dim3 grid(kRanges / kRangeWaves), block(kRangeWaves * 64);
  const RangeOut out{ln.w_stage.p, caprow, wire};
  RowSync sy;
  FS_TRY(fs_row_sync(ix, grid.x, &sy));
  const RowFinal fin{reinterpret_cast<uint8_t*>(d_rows), rcap, ln.d_status.p, host_st, count_out, false};
#define FS_RANGES(NN)                                                                           \
  hipLaunchKernelGGL((k_ranges<NN>), grid, block, 0, s, cd, g, c->d_sbest.p, scan.recs,         \
                     scan.info, scan.capw, n_sub, chunk, out, sy, fin)
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
  return FS_OK;
}
