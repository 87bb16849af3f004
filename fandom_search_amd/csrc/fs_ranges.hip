// fs_ranges.hip -- host side and helper kernels of the records path of k_scan_rows
// (fs_scan.hip; the per-round device code is fs_ranges.h):
//   k_ctab      per batch: the table k_scan_rows verifies against (ids + best record per entry)
//   k_compact   staged records -> their place in the output, totals and status, for
//               indexes that overlap searches on several lanes (with one lane the
//               workgroups of k_scan_rows do that themselves, finish_rows)
//   fs_row_sync the lane's hand-off words of one launch
#include "fs_ranges.h"

#include <hip/hip_ext.h>

#include <algorithm>

namespace {

using namespace fsdev;

// Staged records -> their place in the output.  Block b takes the kCompactRanges wave
// ranges [16 b, 16 b + 16): its first record index is the sum of the block sums
// (`csum`, one per `csum_per` ranges) in front of it.  Every thread first requests all
// its pieces (16 bytes, or 8 for the 8-byte wire records), then stores them.  The extra
// block publishes totals and status.
constexpr int kCompactRanges = 16;

// record i of the block: its place in the staging area and in the output
__device__ __forceinline__ void compact_copy(const uint8_t* __restrict__ stage,
                                             uint8_t* __restrict__ rows, uint32_t caprow,
                                             int wire, const double* __restrict__ selfdist,
                                             uint32_t range0, uint32_t base,
                                             const uint32_t* s_off) {
  const uint32_t P = s_off[kCompactRanges];
  auto locate = [&](uint32_t i, size_t* src, size_t* dst) {
    uint32_t r = 0;                                        // last range with s_off[r] <= i
#pragma unroll
    for (int step = kCompactRanges / 2; step > 0; step >>= 1)
      if (s_off[r + step] <= i) r += step;
    *src = (size_t)(range0 + r) * caprow + (i - s_off[r]);
    *dst = (size_t)base + i;
  };
  // four records per thread requested together (named registers: an array here is moved
  // to LDS by the compiler, with a wait behind every load)
  uint32_t i = threadIdx.x;
  for (; i + 3 * kThreads < P; i += 4 * kThreads) {
    size_t s0, s1, s2, s3, d0, d1, d2, d3;
    locate(i, &s0, &d0); locate(i + kThreads, &s1, &d1);
    locate(i + 2 * kThreads, &s2, &d2); locate(i + 3 * kThreads, &s3, &d3);
    const StagedRec v0 = fetch_staged(stage, wire, selfdist, s0, true), v1 = fetch_staged(stage, wire, selfdist, s1, true),
                    v2 = fetch_staged(stage, wire, selfdist, s2, true), v3 = fetch_staged(stage, wire, selfdist, s3, true);
    store_staged(rows, wire, v0, d0); store_staged(rows, wire, v1, d1);
    store_staged(rows, wire, v2, d2); store_staged(rows, wire, v3, d3);
  }
  for (; i < P; i += kThreads) {
    size_t s0, d0;
    locate(i, &s0, &d0);
    store_staged(rows, wire, fetch_staged(stage, wire, selfdist, s0, true), d0);
  }
}

__global__ __launch_bounds__(kThreads) void k_compact(
    const uint4* __restrict__ rinfo, const uint4* __restrict__ csum,
    const uint32_t* __restrict__ cmax, uint32_t csum_per, uint32_t n_ranges,
    const uint32_t* __restrict__ cand_sums, uint32_t n_cand_sums, bool fresh,
    const uint8_t* __restrict__ stage, uint32_t caprow, int wire, const double* __restrict__ selfdist,
    uint32_t rcap, uint8_t* __restrict__ rows, fs_status* st, fs_status* host_st, uint64_t* count_out) {
  __shared__ uint32_t s_w[4];
  __shared__ uint32_t s_w2[4];
  __shared__ uint32_t s_n[kCompactRanges], s_off[kCompactRanges + 1];
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
        out.max_recs = 0; out.lev_overflow = 0; out.bad_string = 0; out.lsh_pending = 0;
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
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (int r = 0; r < kCompactRanges; ++r) {
      s_off[r] = acc;
      uint32_t n = s_n[r] < caprow ? s_n[r] : caprow;
      if (base + acc >= rcap) n = 0;                       // beyond the caller's buffer
      else if (base + acc + n > rcap) n = rcap - (base + acc);
      acc += n;
    }
    s_off[kCompactRanges] = acc;
  }
  __syncthreads();
  compact_copy(stage, rows, caprow, wire, selfdist, blockIdx.x * kCompactRanges, base, s_off);
}

// per-corpus copy of the best record of every script n-gram, indexed by table slot
// `levtab` (batches with string ids of their own: the table may lack the string of a vector
// id): an n-gram with a rank whose distance is not known gets lev = FS_NONE, which sends its
// hits to the per-hit computation
__global__ void k_ctab(const uint4* __restrict__ proto, uint32_t slots,
                       const fs_best* __restrict__ gbest, uint4* __restrict__ ctab,
                       const uint32_t* __restrict__ levtab, const uint32_t* __restrict__ gcnt, int nn) {
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < slots; s += gridDim.x * blockDim.x) {
    const uint4 q0 = proto[4 * (size_t)s], q1 = proto[4 * (size_t)s + 1];
    uint4 q2 = proto[4 * (size_t)s + 2], q3 = make_uint4(0, 0, 0, 0);
    if (q0.x) {
      const fs_best b = gbest[q0.x - 1];
      const uint64_t db = (uint64_t)__double_as_longlong(b.dist), cb = (uint64_t)__double_as_longlong(b.comb);
      q2.z = b.s; q2.w = b.lev;
      if (levtab)
        for (uint32_t r = 0; r < gcnt[q0.x - 1]; ++r)
          if (levtab[(size_t)(q0.x - 1) * nn + r] == FS_NONE) q2.w = FS_NONE;
      q3 = make_uint4((uint32_t)db, (uint32_t)(db >> 32), (uint32_t)cb, (uint32_t)(cb >> 32));
    }
    uint4* e = ctab + 4 * (size_t)s;
    e[0] = q0; e[1] = q1; e[2] = q2; e[3] = q3;
  }
}

}  // namespace

// the batch table of k_scan_rows: ids from the index, best records from d_gbest
int fs_launch_ctab(fs_index* ix, fs_corpus* c, hipStream_t s) {
  if (!ix->ctab_ok) return FS_OK;
  const size_t slots = (size_t)1 << ix->log2_slots;
  FS_TRY(c->d_ctab.reserve(slots * FS_CTAB_WORDS));
  const uint32_t blocks = (uint32_t)std::min<size_t>((slots + 255) / 256, 1024);
  hipLaunchKernelGGL(k_ctab, dim3(blocks), dim3(256), 0, s,
                     reinterpret_cast<const uint4*>(ix->d_cproto.p), (uint32_t)slots, c->d_gbest.p,
                     reinterpret_cast<uint4*>(c->d_ctab.p),
                     c->has_str ? (const uint32_t*)c->d_levtab.p : (const uint32_t*)nullptr,
                     (const uint32_t*)ix->d_gcnt.p, (int)ix->cfg.nearest_n);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

// staged records of `n_ranges` wave ranges -> the caller's buffer, totals, status
// (the finish of a search on an index with several lanes)
static int launch_compact(fs_index* ix, uint32_t n_ranges, uint32_t csum_per, uint32_t caprow,
                          int wire, uint32_t rcap, fs_row* d_rows, fs_status* host_st,
                          hipStream_t s, uint64_t* count_out, bool fresh, hipEvent_t done = nullptr) {
  fs_index::Lane& ln = *ix->cur;
  if (n_ranges % kCompactRanges || kCompactRanges % csum_per) {
    fs_set_error("k_compact: %u ranges, %u per block sum", n_ranges, csum_per);
    return FS_E_INVALID;
  }
  // (`done`: the search's completion event rides on this dispatch instead of a marker of its own)
  hipExtLaunchKernelGGL(k_compact, dim3(n_ranges / kCompactRanges + 1), dim3(kThreads), 0, s,
                     nullptr, done, 0u, ln.w_rinfo.p, ln.w_csum.p,
                     reinterpret_cast<const uint32_t*>(ln.w_csum.p + n_ranges / csum_per), csum_per,
                     n_ranges, (const uint32_t*)nullptr, 0u, fresh, ln.w_stage.p, caprow, wire,
                     (const double*)ix->d_selfdist.p, rcap, reinterpret_cast<uint8_t*>(d_rows),
                     ln.d_status.p, host_st, count_out);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

// The lane's hand-off words for one launch of `n_blocks` workgroups that end in
// finish_rows: granules and statistics per workgroup, next epoch.
int fs_row_sync(fs_index* ix, uint32_t n_blocks, uint64_t n_tok, fsdev::RowSync* sy) {
  fs_index::Lane& ln = *ix->cur;
  if (n_blocks > FS_SYNC_BLOCKS) { fs_set_error("%u workgroups in a records kernel", n_blocks); return FS_E_INVALID; }
  if (!ln.w_gran.p) {
    FS_TRY(ln.w_gran.reserve(5 * FS_SYNC_BLOCKS));
    FS_HIP(hipMemsetAsync(ln.w_gran.p, 0, 5 * FS_SYNC_BLOCKS * sizeof(unsigned long long), ln.stream));
    ln.sync_epoch = 0;
  }
  if (++ln.sync_epoch == 0) ln.sync_epoch = 1;     // 0 is the tag of a granule never written
  sy->gran = ln.w_gran.p;
  sy->sgran = ln.w_gran.p + FS_SYNC_BLOCKS;
  sy->epoch = ln.sync_epoch;
  sy->n_blocks = n_blocks;
  sy->spin_limit = ix->sw.wait_spins >= 0 ? (uint32_t)ix->sw.wait_spins : 0xffffffffu;
  // time limit of the in-launch hand-off: 500 us + eight times what the scan of this corpus
  // takes at 1 TB/s (a C2 batch: 80 us -> 1.1 ms; a 2 GB corpus: 16.5 ms).  A launch whose
  // workgroups are all resident hands off within microseconds of its last scanner; what this
  // guards against is a workgroup that is not running (co-residency lost), and then the
  // search is repeated through the chained kernels (fs_stats.handoff_fallbacks)
  {
    const double scan_us = (double)n_tok * 4.0 / 1.0e6;           // bytes / (1 TB/s) in us
    const double lim_us = 500.0 + 8.0 * scan_us;
    sy->wait_ticks = (uint32_t)std::min<double>(4.0e9, lim_us * 100.0);
  }
  sy->rinfo = nullptr; sy->csum = nullptr; sy->cmax = nullptr;
  // several lanes (searches overlapped on the GPU): a workgroup waiting inside the launch
  // would hold its CU; the counts go to memory instead and k_compact finishes
  const bool compact = ix->sw.rows_finish == 2 || (ix->sw.rows_finish == 0 && ix->n_lanes > 1);
  if (compact) {
    FS_TRY(ln.w_rinfo.reserve((size_t)n_blocks * 16));
    FS_TRY(ln.w_csum.reserve(2 * (size_t)n_blocks));
    sy->rinfo = ln.w_rinfo.p;
    sy->csum = ln.w_csum.p;
    sy->cmax = reinterpret_cast<uint32_t*>(ln.w_csum.p + n_blocks);
  }
  return FS_OK;
}

int fs_launch_compact_after_scan_rows(fs_index* ix, uint32_t n_ranges, uint32_t waves, uint32_t caprow,
                                      int wire, uint32_t rcap, fs_row* d_rows,
                                      fs_status* host_st, hipStream_t s, uint64_t* count_out,
                                      hipEvent_t done) {
  return launch_compact(ix, n_ranges, waves, caprow, wire, rcap, d_rows, host_st, s, count_out, true,
                        done);
}
