// fs_api.hip -- C ABI of libfandomsearch_hip.so (include/fandom_search.h): host
// side index build and the orchestration of the device pipeline.
#include "fs_internal.h"
#include <sched.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <chrono>
#include <thread>
#include <cmath>
#include <new>
#include <numeric>

static thread_local char g_err[512];

void fs_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

extern "C" const char* fs_last_error(void) { return g_err; }
extern "C" int fs_version(void) { return FS_ABI_VERSION; }

extern "C" const char* fs_strerror(int code) {
  switch (code) {
    case FS_OK: return "ok";
    case FS_E_INVALID: return "invalid argument";
    case FS_E_NOMEM: return "out of memory";
    case FS_E_DEVICE: return "HIP runtime error";
    case FS_E_CAPACITY: return "row buffer too small";
    case FS_E_UNSUPPORTED: return "unsupported parameter";
    case FS_E_UNPROVEN: return "exact n-gram prefilter not proven for this vector table";
    default: return "unknown error";
  }
}

GramIndexDev fs_index::gram_dev() const {
  GramIndexDev g;
  g.stok = d_stok.p; g.filter = d_filter.p; g.table = d_table.p; g.gpos = d_gpos.p;
  g.sfilter = d_sfilter.p;
  g.gcnt = d_gcnt.p; g.selfdist = d_selfdist.p; g.schars = d_schars.p; g.soff = d_soff.p;
  g.log2_words = log2_words; g.log2_swords = log2_swords; g.log2_slots = log2_slots;
  g.tstride = (int)((2 + cfg.window_size + 3) & ~3u);
  g.disp = d_disp.p; g.log2_buckets = log2_buckets;
  g.disp8 = reinterpret_cast<const uint8_t*>(d_disp8.p);
  g.n = (int)cfg.window_size; g.nn = (int)cfg.nearest_n; g.n_grams = n_grams;
  return g;
}

CorpusDev fs_corpus::dev() const {
  CorpusDev c;
  c.tok = d_tok.p; c.str = has_str ? d_str.p : nullptr; c.work_off = d_work_off.p;
  c.blk_work = reinterpret_cast<const uint2*>(d_blk_work.p);
  c.blk4 = reinterpret_cast<const uint4*>(d_blk4.p);
  c.ctab = reinterpret_cast<const uint4*>(d_ctab.p);
  c.chars = d_chars.p; c.coff = d_coff.p;
  c.n_tok = (uint32_t)n_tok; c.n_works = (uint32_t)n_works; c.n_str = (uint32_t)n_str;
  return c;
}

fs_index::~fs_index() {
  if (ev_scan0) (void)hipEventDestroy(ev_scan0);
  if (ev_scan1) (void)hipEventDestroy(ev_scan1);
  if (h_status) (void)hipHostFree(h_status);
  if (h_stage) (void)hipHostFree(h_stage);
  if (host_pool) fs_host_pool_free(host_pool);
  for (hipEvent_t e : prof.ev) (void)hipEventDestroy(e);
  for (int l = 0; l < FS_LANES; ++l)
    if (lanes[l].stream) (void)hipStreamDestroy(lanes[l].stream);
  for (int i = 0; i < FS_SEARCH_SLOTS; ++i) {
    Slot& sl = slots[i];
    if (sl.ev_begin) (void)hipEventDestroy(sl.ev_begin);
    if (sl.ev_scan0) (void)hipEventDestroy(sl.ev_scan0);
    if (sl.ev_scan1) (void)hipEventDestroy(sl.ev_scan1);
    if (sl.ev_end) (void)hipEventDestroy(sl.ev_end);
    if (sl.h_status) (void)hipHostFree(sl.h_status);
  }
}   // `stream` is lanes[0].stream


// FS_* diagnostic switches -> fs_switches (fs_internal.h)
void fs_read_switches(fs_switches* sw) {
  *sw = fs_switches();
  auto num = [](const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; };
  if (const char* e = getenv("FS_SCAN_FLAGS")) sw->scan_flags = e[0] ? e[0] : '-';
  if (const char* e = getenv("FS_SCAN_VARIANT")) sw->scan_simple = e[0] == 's';
  sw->scan_tpl = num("FS_SCAN_TPL");
  if (const char* e = getenv("FS_SCAN_DIRECT")) sw->scan_direct = e[0] != '0';
  sw->scan_capw = num("FS_SCAN_CAPW");
  if (const char* e = getenv("FS_SCAN_ROWS")) sw->scan_rows = e[0] != '0';
  if (const char* e = getenv("FS_SCAN_SUB")) sw->scan_sub = e[0] != '0';
  sw->ranges_caprow = num("FS_RANGES_CAPROW");
  sw->diag = num("FS_DIAG");
  if (const char* e = getenv("FS_LSH_GRAMTAB")) sw->lsh_gramtab = atoi(e) != 0;
  if (const char* e = getenv("FS_LSH_SYN")) sw->lsh_syn = atoi(e) != 0;
  if (getenv("FS_LSH_SHARE")) sw->lsh_share = num("FS_LSH_SHARE");
  if (const char* e = getenv("FS_SHARE_GAMMA")) sw->share_gamma = atof(e);
  if (const char* e = getenv("FS_LSH_KEYS6")) sw->lsh_keys6 = atoi(e) != 0;
  if (const char* e = getenv("FS_LSH_WMAP")) sw->lsh_wmap = atoi(e) != 0;
  if (const char* e = getenv("FS_WAIT_SPINS")) sw->wait_spins = atoi(e);
  sw->rows_finish = num("FS_ROWS_FINISH");
  sw->rows_xpool = num("FS_ROWS_XPOOL");
  if (const char* e = getenv("FS_ROWS_SHARES")) {
    int v[4] = {0, 0, 0, 0};
    if (sscanf(e, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) == 4 && v[0] > 0 && v[1] > 0 && v[2] > 0 && v[3] > 0 &&
        v[0] <= 512 && v[1] <= 512 && v[2] <= 512 && v[3] <= 512 && v[0] + v[1] + v[2] + v[3] == 1024)
      for (int i = 0; i < 4; ++i) sw->rows_shares[i] = v[i];
  }
  if (const char* e = getenv("FS_ROWS_COOP")) sw->rows_coop = e[0] != '0';
  if (const char* e = getenv("FS_LSH_F32_SLACK")) sw->lsh_f32_slack = atof(e);
  if (const char* e = getenv("FS_LSH_F32")) sw->lsh_f32 = e[0] != '0';
  sw->lsh_diag = num("FS_LSH_DIAG");
  sw->lsh_lev_lane = getenv("FS_LSH_LEV_LANE") ? num("FS_LSH_LEV_LANE") : 1;
  sw->scan_near8 = !getenv("FS_SCAN_NEAR8") || num("FS_SCAN_NEAR8") != 0;
  sw->near_fused = !getenv("FS_NEAR_FUSED") || num("FS_NEAR_FUSED") != 0;
  sw->lsh_batch = !getenv("FS_LSH_BATCH") || num("FS_LSH_BATCH") != 0;
  sw->lsh_emap = !getenv("FS_LSH_EMAP") || num("FS_LSH_EMAP") != 0;
  if (getenv("FS_LSH_DEFER_MIN")) sw->lsh_defer_min = num("FS_LSH_DEFER_MIN");
  sw->end_query = !getenv("FS_END_QUERY") || num("FS_END_QUERY") != 0;
  sw->lsh_no_gtab = getenv("FS_LSH_NO_GTAB") != nullptr;
  sw->lsh_serial = getenv("FS_LSH_SERIAL") != nullptr;
  if (const char* e = getenv("FS_LSH_PREFILTER")) sw->lsh_prefilter = e[0] != '0';
  if (const char* e = getenv("FS_LSH_WILD")) sw->lsh_wild = e[0] != '0';
  if (const char* e = getenv("FS_LSH_SELFLEV")) sw->lsh_selflev = e[0] != '0';
  if (const char* e = getenv("FS_ROWS_DISP_LDS")) sw->rows_disp_lds = e[0] != '0';
  if (const char* e = getenv("FS_STR_LEVTAB")) sw->str_levtab = e[0] != '0';
  if (const char* e = getenv("FS_STR_FAST")) sw->str_fast = e[0] != '0';
  if (const char* e = getenv("FS_STR_FUSED")) sw->str_fused = e[0] != '0';
  sw->rows_waves = num("FS_ROWS_WAVES");
  sw->rows_blocks_per_cu = num("FS_ROWS_BLOCKS_PER_CU");
}

static int ceil_log2(uint64_t x) {
  int l = 0;
  while ((1ull << l) < x) ++l;
  return l;
}

// Group the script's windows by their n vector ids; per distinct n-gram keep the
// first `nn` positions in ascending order (what a stable NearestFilter keeps when
// every candidate has the same distance).  Then build the Bloom filter and the
// exact open-addressing table the kernels use.
static int build_gram_index(fs_index* ix, const uint32_t* stok) {
  const uint32_t n = ix->cfg.window_size, nn = ix->cfg.nearest_n;
  const uint64_t W = ix->n_windows;
  std::vector<uint32_t> order(W);
  std::iota(order.begin(), order.end(), 0u);
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    const int c = memcmp(stok + a, stok + b, n * sizeof(uint32_t));
    return c != 0 ? c < 0 : a < b;
  });
  // gkept: entries of the reference's NearestFilter list for a window equal to
  // this n-gram (statistics only): every occurrence once, or once per table when
  // the UniqueFilter is off, capped at N
  std::vector<uint32_t> gpos, gcnt, gkept;
  const uint64_t per_occ = ix->cfg.unique_filter ? 1 : ix->cfg.number_of_hashes;
  for (uint64_t i = 0; i < W;) {
    uint64_t j = i;
    while (j < W && memcmp(stok + order[i], stok + order[j], n * sizeof(uint32_t)) == 0) ++j;
    const uint32_t cnt = (uint32_t)std::min<uint64_t>(j - i, nn);
    gcnt.push_back(cnt);
    gkept.push_back((uint32_t)std::min<uint64_t>((j - i) * per_occ, nn));
    for (uint32_t r = 0; r < nn; ++r) gpos.push_back(r < cnt ? order[i + r] : 0u);
    i = j;
  }
  const uint32_t G = (uint32_t)gcnt.size();
  ix->n_grams = G;

  // filter: about one 32-bit word per n-gram, 4 KiB .. 128 KiB (it lives in LDS)
  int lw = ceil_log2(std::max<uint64_t>(1, (uint64_t)G * 3 / 4));
  if (const char* e = getenv("FS_FILTER_LOG2_WORDS")) lw = atoi(e);
  lw = std::min(15, std::max(10, lw));
  ix->log2_words = lw;
  ix->log2_slots = std::max(4, ceil_log2((uint64_t)G * 2 + 1));
  if (ix->log2_slots > 31) { fs_set_error("script too large"); return FS_E_UNSUPPORTED; }
  const size_t ts = (2 + n + 3) & ~(size_t)3;       // words per table entry
  std::vector<uint32_t> filter(1u << lw, 0u), table(ts << ix->log2_slots, 0u);
  const uint32_t slot_mask = (1u << ix->log2_slots) - 1;
  // hash-and-displace (fs_hash.h): about four n-grams per bucket, largest buckets first
  ix->log2_buckets = std::max(2, ceil_log2(std::max<uint64_t>(1, (uint64_t)G / 4)));
  const uint32_t n_buckets = 1u << ix->log2_buckets;
  std::vector<uint32_t> disp(n_buckets, 0u), ghash(G);
  std::vector<std::vector<uint32_t>> members(n_buckets);
  for (uint32_t g = 0; g < G; ++g) {
    ghash[g] = fs_gram_hash(stok + gpos[(size_t)g * nn], (int)n);
    filter[fs_bloom_word(ghash[g], lw)] |= fs_bloom_mask(ghash[g]);
    members[fs_table_bucket(ghash[g], ix->log2_buckets)].push_back(g);
  }
  auto put = [&](uint32_t g, uint32_t slot) {
    uint32_t* e = &table[ts * slot];                // {gram + 1, kept occurrences, ids[n], pad}
    e[0] = g + 1;
    e[1] = gkept[g];
    memcpy(e + 2, stok + gpos[(size_t)g * nn], n * sizeof(uint32_t));
  };
  std::vector<uint32_t> border(n_buckets);
  for (uint32_t b = 0; b < n_buckets; ++b) border[b] = b;
  std::stable_sort(border.begin(), border.end(), [&](uint32_t a, uint32_t b) {
    return members[a].size() > members[b].size();
  });
  std::vector<uint32_t> overflow;                   // buckets left for linear probing
  std::vector<uint32_t> trial;
  for (uint32_t b : border) {
    const std::vector<uint32_t>& mem = members[b];
    if (mem.empty()) continue;
    bool placed = false;
    for (uint32_t d = 0; d < 4096 && !placed; ++d) {
      trial.clear();
      bool ok = true;
      for (uint32_t g : mem) {
        const uint32_t slot = fs_table_slot_d(ghash[g], d, ix->log2_slots);
        if (table[ts * slot] || std::find(trial.begin(), trial.end(), slot) != trial.end()) { ok = false; break; }
        trial.push_back(slot);
      }
      if (!ok) continue;
      for (size_t i = 0; i < mem.size(); ++i) put(mem[i], trial[i]);
      disp[b] = d;
      placed = true;
    }
    if (!placed) overflow.push_back(b);
  }
  for (uint32_t b : overflow) {                     // after every separable bucket has its slots
    disp[b] = FS_DISP_OVERFLOW;
    for (uint32_t g : members[b]) {
      uint32_t slot = fs_table_slot_d(ghash[g], 0, ix->log2_slots);
      while (table[ts * slot]) slot = (slot + 1) & slot_mask;
      put(g, slot);
    }
  }
  FS_TRY(ix->d_disp.upload(disp.data(), disp.size(), ix->stream));
  FS_TRY(ix->d_filter.upload(filter.data(), filter.size(), ix->stream));
  // batch table of k_scan_rows (fs_hash.h): same placement, 64-byte entries whose best
  // records k_ctab fills in per batch; seeds as bytes
  ix->ctab_ok = false;
  if (n <= FS_CTAB_MAX_N) {
    std::vector<uint32_t> proto((size_t)FS_CTAB_WORDS << ix->log2_slots, 0u);
    for (size_t sl = 0; sl <= slot_mask; ++sl) {
      const uint32_t* e = &table[ts * sl];
      if (e[0]) memcpy(&proto[sl * FS_CTAB_WORDS], e, (2 + n) * sizeof(uint32_t));   // {gram + 1, kept, ids}
    }
    std::vector<uint32_t> disp8((n_buckets + 3) / 4 + 4, 0u);
    for (uint32_t b = 0; b < n_buckets; ++b) {
      const uint32_t d = disp[b] < FS_DISP8_WIDE ? disp[b] : FS_DISP8_WIDE;
      disp8[b >> 2] |= d << (8 * (b & 3));
    }
    FS_TRY(ix->d_cproto.upload(proto.data(), proto.size(), ix->stream));
    FS_TRY(ix->d_disp8.upload(disp8.data(), disp8.size(), ix->stream));
    FS_HIP(hipStreamSynchronize(ix->stream));
    ix->ctab_ok = true;
  }
  // sub-shingle filter (fs_hash.h): one bit per script K-gram.  A window passes by
  // chance with probability (bit density)^(n-K+1); sized for about 2e-3 (false candidates
  // a few per cent of the true ones on the synthetic workloads) -- a small filter is what
  // lets two workgroups share a CU's LDS
  if (const int K = fs_sub_k((int)n)) {
    const double dens = std::pow(2e-3, 1.0 / (double)(n - K + 1));
    int ls = ceil_log2((uint64_t)((double)std::max<uint64_t>(1, ix->n_script) / (32.0 * dens)) + 1);
    if (const char* e = getenv("FS_SFILTER_LOG2_WORDS")) ls = atoi(e);
    ls = std::min(FS_SUB_MAX_LOG2_WORDS, std::max(10, ls));
    ix->log2_swords = ls;
    std::vector<uint32_t> sub(1u << ls, 0u);
    for (uint64_t i = 0; i + K <= ix->n_script; ++i) {
      const uint32_t h = fs_sub_hash(stok + i, K);
      sub[fs_sub_word(h, ls)] |= 1u << fs_sub_bit(h);
    }
    FS_TRY(ix->d_sfilter.upload(sub.data(), sub.size(), ix->stream));
    FS_HIP(hipStreamSynchronize(ix->stream));
  }
  FS_TRY(ix->d_table.upload(table.data(), table.size(), ix->stream));
  FS_TRY(ix->d_gpos.upload(gpos.data(), gpos.size(), ix->stream));
  FS_TRY(ix->d_gcnt.upload(gcnt.data(), gcnt.size(), ix->stream));
  FS_HIP(hipStreamSynchronize(ix->stream));   // the vectors above die with this scope
  ix->info.n_grams = G;
  ix->info.filter_bytes = (uint64_t)4 << lw;
  return FS_OK;
}

// Soundness of the exact n-gram prefilter (DESIGN.md "proof"): a record needs
// cos(F, S) > 1 - threshold.  If F and S differ in at least one slot,
//   cos(F,S) <= ((n-1) a_max^2 + c a_min^2) / ((n-1) a_max^2 + a_min^2)
// with a_min/a_max the extreme vector norms of the table and c the largest
// cosine between a script vector and any other table vector.  When that bound
// is below 1 - threshold only id-identical windows can produce records.
static int prove_exact(fs_index* ix, const uint32_t* stok) {
  fs_index_info& inf = ix->info;
  inf.proof_ok = 0; inf.c_max = 1.0; inf.cos_bound = 1.0; inf.norm_min = 0; inf.norm_max = 0;
  const uint64_t V = ix->n_vec;
  const int D = (int)ix->cfg.emb_dim;
  if (V == 0 || ix->n_windows == 0) {   // nothing can match; the scan finds nothing
    inf.proof_ok = 1; inf.c_max = 0; inf.cos_bound = 0;
    return FS_OK;
  }
  std::vector<uint32_t> rows_u;
  bool script_oov = false;
  {
    std::vector<uint8_t> seen(V, 0);
    for (uint64_t i = 0; i < ix->n_script; ++i) {
      const uint32_t id = stok[i];
      if (id & FS_OOV_FLAG) { script_oov = true; ix->script_oov = true; continue; }   // 3-hot vectors: no proof
      if (!seen[id]) { seen[id] = 1; rows_u.push_back(id); }
    }
  }
  std::vector<double> q(V);
  FS_HIP(hipMemcpyAsync(q.data(), ix->d_q.p, V * sizeof(double), hipMemcpyDeviceToHost, ix->stream));
  DBuf<float> embT;
  DBuf<uint32_t> d_rows_u;
  DBuf<int> d_bits;
  FS_TRY(embT.reserve((size_t)V * D));
  FS_TRY(d_rows_u.upload(rows_u.data(), rows_u.size(), ix->stream));
  FS_TRY(d_bits.reserve(1));
  FS_HIP(hipMemsetAsync(d_bits.p, 0, sizeof(int), ix->stream));
  FS_TRY(fs_launch_cmax(ix->d_emb.p, V, D, d_rows_u.p, (uint32_t)rows_u.size(), ix->d_q.p, embT.p,
                        d_bits.p, ix->stream));
  int bits = 0;
  FS_HIP(hipMemcpyAsync(&bits, d_bits.p, sizeof(int), hipMemcpyDeviceToHost, ix->stream));
  FS_HIP(hipStreamSynchronize(ix->stream));
  float cmaxf;
  memcpy(&cmaxf, &bits, sizeof cmaxf);
  double qmin = q[0], qmax = q[0];
  for (uint64_t v = 1; v < V; ++v) { qmin = std::min(qmin, q[v]); qmax = std::max(qmax, q[v]); }
  inf.norm_min = sqrt(qmin); inf.norm_max = sqrt(qmax);
  inf.c_max = cmaxf;
  ix->lsh_cmax = std::min(1.0, (double)cmaxf + 1e-4);
  if (!(qmin > 0.0) || script_oov) return FS_OK;
  const double c = std::min(1.0, (double)cmaxf + 1e-4);     // float32 accumulation slack
  const double n1 = (double)ix->cfg.window_size - 1.0;
  inf.cos_bound = (n1 * qmax + c * qmin) / (n1 * qmax + qmin);
  inf.proof_ok = inf.cos_bound < 1.0 - ix->cfg.distance_threshold - 1e-6 ? 1u : 0u;
  if (V > FS_MAX_EXACT_ID) inf.proof_ok = 0;   // the scan's 24-bit premix needs ids < 2^24
  return FS_OK;
}

// Script text for the lane-per-pair Levenshtein (k_strbest, batches with string ids): the
// distinct code points of the script become classes 1 .. K (0 = any other code point, which
// equals no script character), and every script window's text "w0 w1 .. wn-1" (search.py:189)
// of at most 64 code points is stored as 7 bit planes of its classes (bit j of plane b = bit b
// of the class of character j; positions behind the text hold class 127, which no fan
// character has).  More than 125 distinct code points: the path is not used.
static int fs_build_str_patterns(fs_index* ix, const uint32_t* script_chars, const uint64_t* script_off,
                                 uint64_t n_script) {
  ix->strfast_ok = false;
  const int n = (int)ix->cfg.window_size;
  if (!ix->n_windows || !script_chars) return FS_OK;
  const uint64_t n_chars = script_off[n_script];
  std::vector<uint32_t> cps(script_chars, script_chars + n_chars);
  cps.push_back((uint32_t)' ');
  std::sort(cps.begin(), cps.end());
  cps.erase(std::unique(cps.begin(), cps.end()), cps.end());
  if (cps.size() > 125) return FS_OK;
  auto cls = [&](uint32_t cp) -> uint32_t {
    const auto it = std::lower_bound(cps.begin(), cps.end(), cp);
    return it != cps.end() && *it == cp ? (uint32_t)(it - cps.begin()) + 1u : 0u;
  };
  std::vector<uint8_t> ccls(n_chars);
  for (uint64_t i = 0; i < n_chars; ++i) ccls[i] = (uint8_t)cls(script_chars[i]);
  const uint8_t space = (uint8_t)cls(' ');
  const uint64_t W = ix->n_windows;
  std::vector<unsigned long long> pat(8 * W, 0ull);
  for (uint64_t s = 0; s < W; ++s) {
    unsigned long long* P = pat.data() + 8 * s;
    const uint64_t la = script_off[s + n] - script_off[s] + (uint64_t)(n - 1);
    P[7] = la;
    if (la > 64 || la == 0) { P[7] = 0xFFFFFFFFull; continue; }
    uint8_t text[64];
    uint32_t at = 0;
    for (int k = 0; k < n; ++k) {
      if (k) text[at++] = space;
      for (uint64_t i = script_off[s + k]; i < script_off[s + k + 1]; ++i) text[at++] = ccls[i];
    }
    for (uint32_t j = 0; j < 64; ++j) {
      const uint32_t cj = j < la ? text[j] : 127u;
      for (int b = 0; b < 7; ++b)
        if ((cj >> b) & 1u) P[b] |= 1ull << j;
    }
  }
  FS_TRY(ix->d_clsmap.upload(cps.data(), cps.size(), ix->stream));
  FS_TRY(ix->d_pat.upload(pat.data(), pat.size(), ix->stream));
  FS_HIP(hipStreamSynchronize(ix->stream));
  ix->n_cls = (uint32_t)cps.size();
  ix->str_punct = cls('[') | (cls(',') << 8) | (cls(' ') << 16) | (cls(']') << 24);
  ix->strfast_ok = true;
  return FS_OK;
}

extern "C" int fs_index_create(const fs_config* cfg, const uint32_t* script_vec,
                               const uint32_t* script_chars, const uint64_t* script_off,
                               uint64_t n_script, const float* emb, uint64_t n_vec,
                               const double* normals, fs_index** out) {
  if (!cfg || !out || cfg->struct_size != sizeof(fs_config)) {
    fs_set_error("fs_config missing or of a different ABI size");
    return FS_E_INVALID;
  }
  *out = nullptr;
  if ((n_script && (!script_vec || !script_off)) || (n_vec && !emb)) {
    fs_set_error("null input buffer");
    return FS_E_INVALID;
  }
  if (cfg->window_size < 1 || cfg->window_size > FS_MAX_WINDOW || cfg->nearest_n < 1 ||
      cfg->nearest_n > 64 || cfg->emb_dim < 1 || cfg->emb_dim > 1024 ||
      cfg->hash_dimensions < 1 || cfg->hash_dimensions > 24 || cfg->number_of_hashes < 1 ||
      cfg->number_of_hashes > 64) {
    fs_set_error("window_size 1..%d, nearest_n 1..64, emb_dim 1..1024, hash_dimensions 1..24, "
                 "number_of_hashes 1..64", FS_MAX_WINDOW);
    return FS_E_UNSUPPORTED;
  }
  if (n_script >= (1ull << 31) || n_vec >= (1ull << 31) ||
      (uint64_t)cfg->emb_dim * cfg->emb_dim * cfg->emb_dim >= (1ull << 31)) {
    fs_set_error("script or vector table too large");
    return FS_E_UNSUPPORTED;
  }
  for (uint64_t i = 0; i < n_script; ++i) {
    const uint32_t id = script_vec[i];
    if (!(id & FS_OOV_FLAG) && id >= n_vec) {
      fs_set_error("script_vec[%llu] = %u is outside the vector table", (unsigned long long)i, id);
      return FS_E_INVALID;
    }
    if (script_off[i + 1] < script_off[i]) { fs_set_error("script_off not monotone"); return FS_E_INVALID; }
  }
  int ndev = 0;
  FS_HIP(hipGetDeviceCount(&ndev));
  if (cfg->device < 0 || cfg->device >= ndev) {
    fs_set_error("device %d of %d", cfg->device, ndev);
    return FS_E_INVALID;
  }
  FS_ENTER(cfg->device);
  fs_index* ix = new (std::nothrow) fs_index();
  if (!ix) return FS_E_NOMEM;
  struct Guard { fs_index* p; ~Guard() { delete p; } } guard{ix};
  ix->cfg = *cfg;
  fs_read_switches(&ix->sw);
  ix->device = cfg->device;
  ix->n_script = n_script; ix->n_vec = n_vec;
  ix->n_windows = n_script >= cfg->window_size ? n_script - cfg->window_size + 1 : 0;
  hipDeviceProp_t prop;
  FS_HIP(hipGetDeviceProperties(&prop, cfg->device));
  ix->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  for (int l = 0; l < FS_LANES; ++l)
    FS_HIP(hipStreamCreateWithFlags(&ix->lanes[l].stream, hipStreamNonBlocking));
  ix->stream = ix->lanes[0].stream;
  if (const char* e = getenv("FS_LANES")) ix->n_lanes = std::min(FS_LANES, std::max(1, atoi(e)));
  FS_HIP(hipEventCreate(&ix->ev_scan0));      // fs_scan_benchmark
  FS_HIP(hipEventCreate(&ix->ev_scan1));
  FS_HIP(hipHostMalloc((void**)&ix->h_status, sizeof(fs_status), hipHostMallocDefault));
  for (int i = 0; i < FS_SEARCH_SLOTS; ++i) {
    fs_index::Slot& sl = ix->slots[i];
    FS_HIP(hipEventCreate(&sl.ev_begin));
    FS_HIP(hipEventCreate(&sl.ev_scan0));
    FS_HIP(hipEventCreate(&sl.ev_scan1));
    FS_HIP(hipEventCreate(&sl.ev_end));
    // the status block and, behind it, the "a wait gave up" word of finish_rows
    FS_HIP(hipHostMalloc((void**)&sl.h_status, 2 * sizeof(fs_status), hipHostMallocDefault));
    memset(sl.h_status, 0, 2 * sizeof(fs_status));
  }
  for (int l = 0; l < FS_LANES; ++l) {
    FS_TRY(ix->lanes[l].d_status.reserve(1));
    FS_TRY(ix->lanes[l].w_bsum.reserve(4096));
    FS_TRY(ix->lanes[l].w_bsum64.reserve(2048));
  }

  // FS_BUILD_TIMES=1: seconds per stage of the index build on stderr
  const bool times = getenv("FS_BUILD_TIMES") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto tick = [&](const char* what) {
    if (!times) return;
    (void)hipStreamSynchronize(ix->stream);
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "fs_index_create: %-28s %8.3f ms\n", what,
            std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  tick("streams, lanes, workspaces");
  // script ids padded by n so that device code may read a full window anywhere
  std::vector<uint32_t> stok(n_script + FS_MAX_WINDOW + 1, 0u);
  if (n_script) memcpy(stok.data(), script_vec, n_script * sizeof(uint32_t));
  FS_TRY(ix->d_stok.upload(stok.data(), stok.size(), ix->stream));
  const uint64_t n_chars = n_script ? script_off[n_script] : 0;
  FS_TRY(ix->d_schars.upload(script_chars, n_chars, ix->stream));
  {
    std::vector<uint64_t> soff(n_script + 1, 0);
    if (n_script) memcpy(soff.data(), script_off, (n_script + 1) * sizeof(uint64_t));
    FS_TRY(ix->d_soff.upload(soff.data(), soff.size(), ix->stream));
    FS_HIP(hipStreamSynchronize(ix->stream));
  }
  FS_TRY(fs_build_str_patterns(ix, script_chars, script_off, n_script));
  FS_TRY(ix->d_emb.upload(emb, (size_t)n_vec * cfg->emb_dim, ix->stream));
  if (normals) {
    const size_t nn = (size_t)cfg->number_of_hashes * cfg->hash_dimensions * cfg->emb_dim *
                      cfg->window_size;
    FS_TRY(ix->d_normals.upload(normals, nn, ix->stream));
  }
  FS_TRY(ix->d_q.reserve(n_vec));
  FS_TRY(ix->d_selfdist.reserve(ix->n_windows));
  FS_TRY(fs_launch_rownorms(ix->d_emb.p, n_vec, (int)cfg->emb_dim, ix->d_q.p, ix->stream));
  FS_TRY(fs_launch_selfdist(ix->d_stok.p, ix->n_windows, (int)cfg->window_size, (int)cfg->emb_dim,
                            n_vec, ix->d_q.p, ix->d_selfdist.p, ix->stream));
  ix->h_selfdist.resize(ix->n_windows);
  if (ix->n_windows)
    FS_HIP(hipMemcpyAsync(ix->h_selfdist.data(), ix->d_selfdist.p, ix->n_windows * sizeof(double),
                          hipMemcpyDeviceToHost, ix->stream));
  tick("uploads, norms, self distances");
  FS_TRY(build_gram_index(ix, stok.data()));
  tick("n-gram index");
  FS_TRY(prove_exact(ix, stok.data()));
  FS_HIP(hipStreamSynchronize(ix->stream));
  tick("proof (c_max)");

  ix->info.n_script = n_script;
  ix->info.n_windows = ix->n_windows;
  const bool exact = ix->info.proof_ok && cfg->mode != FS_MODE_GENERAL;
  if (cfg->mode == FS_MODE_EXACT && !ix->info.proof_ok) {
    fs_set_error("exact mode requested but cos bound %.6f >= 1 - threshold (c_max %.6f)",
                 ix->info.cos_bound, ix->info.c_max);
    return FS_E_UNPROVEN;
  }
  ix->info.path = exact ? FS_MODE_EXACT : FS_MODE_GENERAL;
  if (!exact) FS_TRY(fs_lsh_build(ix));
  tick("LSH structures");
  guard.p = nullptr;
  *out = ix;
  return FS_OK;
}

extern "C" int fs_index_info_get(const fs_index* ix, fs_index_info* info) {
  if (!ix || !info) return FS_E_INVALID;
  *info = ix->info;
  return FS_OK;
}

extern "C" int fs_index_component_sizes(const fs_index* ix, uint32_t* sizes, uint64_t cap, uint64_t* n,
                                        uint32_t* in_use) {
  if (!ix || !n) return FS_E_INVALID;
  *n = ix->comp_sizes.size();
  if (in_use) *in_use = (ix->syn_ok || ix->share_flags) ? 1u : 0u;
  if (sizes)
    for (uint64_t i = 0; i < cap && i < ix->comp_sizes.size(); ++i) sizes[i] = ix->comp_sizes[i];
  return FS_OK;
}

extern "C" int fs_index_share_info(const fs_index* ix, uint32_t* flags, uint32_t* components, uint32_t* largest,
                                   double* gamma) {
  if (!ix) return FS_E_INVALID;
  if (flags) *flags = (uint32_t)ix->share_flags;
  if (components) *components = ix->share_comps;
  if (largest) *largest = ix->share_largest;
  if (gamma) *gamma = ix->share_gamma;
  return FS_OK;
}

extern "C" int fs_index_share_counts(fs_index* ix, uint64_t* out8) {
  if (!ix || !out8) return FS_E_INVALID;
  for (int i = 0; i < 8; ++i) out8[i] = 0;
  if (ix->d_share_cnt.n < 16) return FS_OK;
  FS_ENTER(ix->device);
  FS_HIP(hipDeviceSynchronize());
  FS_HIP(hipMemcpy(out8, ix->d_share_cnt.p, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  FS_HIP(hipMemset(ix->d_share_cnt.p, 0, 8 * sizeof(uint64_t)));
  return FS_OK;
}

extern "C" const char* fs_search_kernel_name(fs_index* ix, fs_corpus* c) {
  static thread_local char name[64];
  if (!ix || !c || c->ix != ix) return "";
  const int n = (int)ix->cfg.window_size;
  const bool exact = ix->info.path == FS_MODE_EXACT && !c->has_oov;
  if (!exact) {
    snprintf(name, sizeof name, !fs_lsh_prefilter_ok(ix, c) ? ((ix->share_flags & 32) ? "k_share_scan<%d>" : "k_lsh_scan") :
                                fs_near_fused(ix, c) ? "k_near_sift<%d>" :
                                fs_scan_near8(ix) ? "k_scan_near8<%d>" : "k_scan_near<%d>", n);
  } else if (uint32_t blocks = 0; fs_scan_rows_shape(ix, c, &blocks)) {
    const int k = ix->sw.scan_sub && ix->d_sfilter.p ? fs_sub_k(n) : 0;
    snprintf(name, sizeof name, "k_scan_rows<%d,%d>", n, k);
  } else if (fs_scan_tpl(ix, c->n_tok) == 8) {
    snprintf(name, sizeof name, "k_scan8<%d>", n);
  } else {
    snprintf(name, sizeof name, "k_scan<%d>", n);
  }
  return name;
}

// Diagnostics: the FS_* switches are read at fs_index_create; a test or sweep that
// changes them on a live index calls this (never needed on the search path).
extern "C" int fs_index_reload_switches(fs_index* ix) {
  if (!ix) return FS_E_INVALID;
  fs_read_switches(&ix->sw);
  return FS_OK;
}

extern "C" void fs_index_destroy(fs_index* ix) {
  if (!ix) return;
  (void)hipSetDevice(ix->device);
  for (int l = 0; l < FS_LANES; ++l)
    if (ix->lanes[l].stream) (void)hipStreamSynchronize(ix->lanes[l].stream);
  // corpora that outlive the index keep their own device buffers and are detached:
  // they can still be destroyed, every other call on them is refused
  for (fs_corpus* c : ix->corpora) c->ix = nullptr;
  delete ix;
}

fs_corpus::~fs_corpus() {
  if (ev_ready) (void)hipEventDestroy(ev_ready);
  if (copy_stream) (void)hipStreamDestroy(copy_stream);
  if (h_check) (void)hipHostFree(h_check);
}

// Enqueue the upload of a batch of works on the corpus's copy stream: ids,
// optional string ids, work offsets, the block -> work table and a device-side
// validation pass.  Returns as soon as everything is queued; the host buffers
// must stay untouched until fs_corpus_update_end.
extern "C" int fs_corpus_update_begin(fs_corpus* c, const uint32_t* tok_vec,
                                      const uint32_t* tok_str, const uint64_t* work_off,
                                      uint64_t n_works) {
  if (!c || !work_off) { fs_set_error("null argument"); return FS_E_INVALID; }
  fs_index* ix = c->ix;
  if (!ix) { fs_set_error("the corpus's index has been destroyed"); return FS_E_INVALID; }
  if (work_off[0] != 0) { fs_set_error("work_off[0] must be 0"); return FS_E_INVALID; }
  const uint64_t T = work_off[n_works];
  if (T && !tok_vec) { fs_set_error("null token buffer"); return FS_E_INVALID; }
  if (T >= (1ull << 32) - 65536 || n_works >= (1ull << 32) - 1) {
    fs_set_error("one corpus batch holds fewer than 2^32 tokens; split the batch");
    return FS_E_UNSUPPORTED;
  }
  const uint32_t n = ix->cfg.window_size;
  uint64_t windows = 0;
  for (uint64_t w = 0; w < n_works; ++w) {
    if (work_off[w + 1] < work_off[w]) { fs_set_error("work_off not monotone"); return FS_E_INVALID; }
    const uint64_t len = work_off[w + 1] - work_off[w];
    if (len >= n) windows += len - n + 1;
  }
  FS_ENTER(ix->device);
  // a search of this corpus still in flight would race with the copies queued below (and a
  // repeat of it would run with the old batch's geometry)
  for (int i = 0; i < FS_SEARCH_SLOTS; ++i)
    if (ix->slots[i].busy && ix->slots[i].c == c) {
      fs_set_error("a search of this corpus is still in flight: finish it (fs_search_corpus_end) "
                   "before the corpus is updated");
      return FS_E_INVALID;
    }
  if (c->pending) FS_HIP(hipEventSynchronize(c->ev_ready));
  hipStream_t cs = c->copy_stream;
  c->n_tok = T; c->n_works = n_works; c->windows = windows;
  c->has_str = tok_str != nullptr;
  const size_t pad = fs_scan_pad_tokens();
  FS_TRY(c->d_tok.reserve(T + pad));
  FS_HIP(hipMemsetAsync(c->d_tok.p + T, 0, pad * sizeof(uint32_t), cs));
  if (T) FS_HIP(hipMemcpyAsync(c->d_tok.p, tok_vec, T * sizeof(uint32_t), hipMemcpyHostToDevice, cs));
  if (tok_str) {
    FS_TRY(c->d_str.reserve(T + FS_MAX_WINDOW + 1));
    FS_HIP(hipMemsetAsync(c->d_str.p + T, 0, (FS_MAX_WINDOW + 1) * sizeof(uint32_t), cs));
    if (T) FS_HIP(hipMemcpyAsync(c->d_str.p, tok_str, T * sizeof(uint32_t), hipMemcpyHostToDevice, cs));
  }
  FS_TRY(c->d_work_off.upload(work_off, n_works + 1, cs));
  c->h_work_off.assign(work_off, work_off + n_works + 1);
  const uint32_t n_blocks = (uint32_t)((T + 255) / 256);
  FS_TRY(c->d_blk_work.reserve(2 * (size_t)n_blocks));
  FS_TRY(c->d_blk4.reserve(4 * (size_t)n_blocks));
  FS_TRY(fs_launch_blk_work(c->d_work_off.p, (uint32_t)n_works, n_blocks,
                            reinterpret_cast<uint2*>(c->d_blk_work.p),
                            reinterpret_cast<uint4*>(c->d_blk4.p), cs));
  FS_TRY(c->d_check.reserve(4));
  FS_HIP(hipMemsetAsync(c->d_check.p, 0, 4 * sizeof(uint32_t), cs));
  FS_TRY(fs_launch_corpus_check(c->d_tok.p, tok_str ? c->d_str.p : nullptr, (uint32_t)T,
                                c->d_check.p, cs));
  FS_HIP(hipMemcpyAsync(c->h_check, c->d_check.p, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, cs));
  FS_HIP(hipEventRecord(c->ev_ready, cs));
  c->pending = true;
  return FS_OK;
}

// Wait for the upload, read the validation result, prepare what the search of
// this batch needs (LSH structures for OOV ids, the per-n-gram Levenshtein table).
extern "C" int fs_corpus_update_end(fs_corpus* c) {
  if (!c) return FS_E_INVALID;
  if (!c->pending) return FS_OK;
  fs_index* ix = c->ix;
  if (!ix) { fs_set_error("the corpus's index has been destroyed"); return FS_E_INVALID; }
  FS_ENTER(ix->device);
  FS_HIP(hipEventSynchronize(c->ev_ready));
  c->pending = false;
  const uint32_t max_row_plus1 = c->h_check[0], any_oov = c->h_check[1], max_str_plus1 = c->h_check[2];
  if (max_row_plus1 > ix->n_vec) {
    fs_set_error("a vector id (%u) is outside the vector table (%llu rows)", max_row_plus1 - 1,
                 (unsigned long long)ix->n_vec);
    return FS_E_INVALID;
  }
  if (max_str_plus1 > c->n_str) {
    fs_set_error("a string id (%u) is outside the string table (%llu strings)", max_str_plus1 - 1,
                 (unsigned long long)c->n_str);
    return FS_E_INVALID;
  }
  if (any_oov && !c->has_str) {
    fs_set_error("out-of-vocabulary vector ids need string ids (tok_str)");
    return FS_E_INVALID;
  }
  c->has_oov = any_oov != 0;
  // out-of-vocabulary vectors are outside the exact n-gram proof: such a batch
  // goes through the LSH pipeline (built now if the index did not need it before)
  if (c->has_oov && ix->info.path == FS_MODE_EXACT) FS_TRY(fs_lsh_build(ix));
  if (!c->has_oov && ix->info.path == FS_MODE_EXACT && !c->levtab_ready) {
    // Levenshtein per (n-gram, rank) against the strings whose ids are the n-gram's vector
    // ids, once per string table: every hit's distance when string id == vector id, and
    // with string ids of their own the distance of the hits whose tokens all carry string
    // id == vector id (entries FS_NONE where such a string does not exist)
    ix->cur = &ix->lanes[0];
    FS_HIP(hipMemsetAsync(ix->cur->d_status.p, 0, sizeof(fs_status), ix->stream));
    FS_TRY(fs_launch_levtab(ix, c, ix->stream));
    FS_HIP(hipMemcpyAsync(ix->h_status, ix->cur->d_status.p, sizeof(fs_status), hipMemcpyDeviceToHost, ix->stream));
    FS_HIP(hipStreamSynchronize(ix->stream));
    if (ix->h_status->bad_string) { fs_set_error("script vector id without a string"); return FS_E_INVALID; }
    if (ix->h_status->lev_overflow) {
      fs_set_error("an n-gram text exceeds %d code points", FS_LEV_MAX);
      return FS_E_UNSUPPORTED;
    }
    c->levtab_ready = true;
  }
  // (the LSH pipeline's k_lsh_lev reads them too, with or without string ids of the batch's own)
  const bool lsh_path = ix->info.path != FS_MODE_EXACT || c->has_oov;
  if (((c->has_str && !c->has_oov && ix->info.path == FS_MODE_EXACT) || (lsh_path && ix->sw.lsh_lev_lane)) &&
      ix->strfast_ok && ix->sw.str_fast && !c->strrec_ready) {
    // the string table as character classes of the script's alphabet, once per string table
    FS_TRY(fs_launch_strrec(ix, c, ix->stream));
    FS_HIP(hipStreamSynchronize(ix->stream));
    c->strrec_ready = true;
  }
  // the batch table of k_scan_rows (ids + this string table's best records), on its own
  // flag: a corpus that is reused (fs_corpus_update_begin) may see its first batch without
  // string ids only after one with them
  // (a batch with string ids of its own gets the table too, for k_scan_rows' per-hit
  // Levenshtein form: its entries mark the n-grams whose table distance is not known; the
  // flavour is rebuilt when a reused corpus changes sides)
  const bool ctab_wanted = !c->has_str || (c->strrec_ready && ix->sw.str_fused);
  if (!c->has_oov && ctab_wanted && ix->info.path == FS_MODE_EXACT && c->levtab_ready &&
      (!c->ctab_ready || c->ctab_str != c->has_str)) {
    ix->cur = &ix->lanes[0];
    FS_TRY(fs_launch_ctab(ix, c, ix->stream));
    FS_HIP(hipStreamSynchronize(ix->stream));
    c->ctab_ready = true;
    c->ctab_str = c->has_str;
  }
  if (!c->has_oov && ix->info.path != FS_MODE_EXACT && ix->syn_ok) {
    // tables with near-synonyms: the component id of every token, for the integer prefilters
    FS_TRY(fs_launch_comp_map(ix, c, ix->stream));
    FS_HIP(hipStreamSynchronize(ix->stream));
    c->ctok_ready = true;
  } else {
    c->ctok_ready = false;
  }
  if (!c->has_str && !c->has_oov && ix->info.path != FS_MODE_EXACT && !c->selflev_ready && ix->sw.lsh_selflev) {
    // LSH pipeline, string id == vector id: the Levenshtein distance of a match with the
    // same id in every slot, once per string table
    FS_TRY(fs_launch_selflev(ix, c, ix->stream));
    FS_HIP(hipStreamSynchronize(ix->stream));
    c->selflev_ready = true;
  }
  if (!c->has_str && !c->has_oov && ix->info.path != FS_MODE_EXACT && !ix->script_oov && !c->gramtab_ready &&
      ix->sw.lsh_gramtab && ix->n_grams) {
    // ... and what a window with the ids of a script n-gram gets (keys, buckets, distances,
    // Levenshtein, first minimum): once per n-gram and string table instead of once per such
    // window of every batch.  A string the table lacks or an over-long text leaves the table
    // unused: the search then reports the error if and when such a window occurs.
    ix->cur = &ix->lanes[0];
    FS_HIP(hipMemsetAsync(ix->cur->d_status.p, 0, sizeof(fs_status), ix->stream));
    FS_TRY(fs_launch_lsh_gramtab(ix, c, ix->stream));
    FS_HIP(hipMemcpyAsync(ix->h_status, ix->cur->d_status.p, sizeof(fs_status), hipMemcpyDeviceToHost, ix->stream));
    FS_HIP(hipStreamSynchronize(ix->stream));
    c->gramtab_ready = !ix->h_status->bad_string && !ix->h_status->lev_overflow;
  }
  return FS_OK;
}

extern "C" int fs_corpus_create(fs_index* ix, const uint32_t* tok_vec, const uint32_t* tok_str,
                                const uint64_t* work_off, uint64_t n_works,
                                const uint32_t* str_chars, const uint64_t* str_off, uint64_t n_str,
                                fs_corpus** out) {
  if (!ix || !out || !work_off || (n_str && (!str_chars || !str_off))) {
    fs_set_error("null argument");
    return FS_E_INVALID;
  }
  *out = nullptr;
  if (n_str >= (1ull << 32)) { fs_set_error("string table too large"); return FS_E_UNSUPPORTED; }
  FS_ENTER(ix->device);
  fs_corpus* c = new (std::nothrow) fs_corpus();
  if (!c) return FS_E_NOMEM;
  struct Guard { fs_corpus* p; ~Guard() { delete p; } } guard{c};
  c->ix = ix; c->n_str = n_str;
  FS_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  FS_HIP(hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
  FS_HIP(hipHostMalloc((void**)&c->h_check, 4 * sizeof(uint32_t), hipHostMallocDefault));
  FS_TRY(c->d_chars.upload(str_chars, n_str ? str_off[n_str] : 0, c->copy_stream));
  {
    std::vector<uint64_t> coff(n_str + 1, 0);
    if (n_str) memcpy(coff.data(), str_off, (n_str + 1) * sizeof(uint64_t));
    FS_TRY(c->d_coff.upload(coff.data(), coff.size(), c->copy_stream));
    FS_HIP(hipStreamSynchronize(c->copy_stream));
  }
  FS_TRY(fs_corpus_update_begin(c, tok_vec, tok_str, work_off, n_works));
  FS_TRY(fs_corpus_update_end(c));
  guard.p = nullptr;
  ix->corpora.push_back(c);
  *out = c;
  return FS_OK;
}

// pinned host memory for staging streamed batches
extern "C" int fs_host_alloc(uint64_t bytes, void** out) {
  if (!out) return FS_E_INVALID;
  *out = nullptr;
  hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
  if (e != hipSuccess) { fs_set_error("hipHostMalloc(%llu) -> %s", (unsigned long long)bytes, hipGetErrorString(e)); return FS_E_NOMEM; }
  return FS_OK;
}

extern "C" void fs_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

extern "C" void fs_corpus_destroy(fs_corpus* c) {
  if (!c) return;
  if (c->ix) {                         // nullptr: the index went first (fs_index_destroy)
    fs_index* ix = c->ix;
    (void)hipSetDevice(ix->device);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    for (int l = 0; l < FS_LANES; ++l) (void)hipStreamSynchronize(ix->lanes[l].stream);
    for (size_t i = 0; i < ix->corpora.size(); ++i)
      if (ix->corpora[i] == c) { ix->corpora.erase(ix->corpora.begin() + (long)i); break; }
    for (int i = 0; i < FS_SEARCH_SLOTS; ++i)
      if (ix->slots[i].c == c) ix->slots[i].c = nullptr;
  } else if (c->copy_stream) {
    (void)hipStreamSynchronize(c->copy_stream);
  }
  delete c;
}

// ---- search: enqueue / finish ---------------------------------------------------
// A search is queued into one of FS_SEARCH_SLOTS slots and finished later; the
// synchronous entry point is begin + end.  Slot i runs on lane i % n_lanes (a lane =
// a stream with its own workspaces and device status block).  One lane by default:
// searches run in queue order.  With FS_LANES=2..4 consecutive searches overlap on
// the GPU: the dependent-load chain after the scan leaves most of the machine idle,
// and the next search's scan fills it (DESIGN.md section 6 has the measured
// trade-off).  Each search's status goes to its slot's pinned block, written by the
// chain's last kernel.

static int search_enqueue(fs_index* ix, fs_index::Slot& sl) {
  fs_index::Lane& ln = ix->lanes[sl.lane];
  ix->cur = &ln;                       // the launchers enqueue into this lane
  hipStream_t s = ln.stream;
  fs_corpus* c = sl.c;
  const uint32_t nn = ix->cfg.nearest_n;
  const uint32_t n_bm = sl.n_bm;
  if (sl.fused_waves) {                // k_scan_rows: nothing between scan and records
  } else if (sl.capw) {                // direct path: records instead of a bitmap
    FS_TRY(ln.w_recs.reserve((size_t)FS_CHUNKS * 4 * sl.capw));
    FS_TRY(ln.w_info.reserve((size_t)FS_CHUNKS * 4));
  } else {
    FS_TRY(ln.w_qbm.reserve((size_t)n_bm * sl.tpl));
    FS_TRY(ln.w_qcnt.reserve(n_bm));
  }
  if (!sl.caprow) {                    // the per-candidate arrays of the chained kernels
    FS_TRY(ln.w_cpos.reserve(sl.ccap));
    FS_TRY(ln.w_cg.reserve(sl.ccap));
    FS_TRY(ln.w_cw.reserve(sl.ccap));
    FS_TRY(ln.w_hv.reserve(sl.ccap));
    FS_TRY(ln.w_hcomb.reserve(sl.ccap));
    FS_TRY(ln.w_mlev.reserve(sl.exact && c->has_str ? sl.ccap * nn : 1));
    FS_TRY(ln.w_cbest.reserve(!sl.exact || c->has_str ? sl.ccap : 1));
  }
  fs_row* d_rows = sl.rows;
  uint64_t* count_out = nullptr;
  if (sl.header) {                     // 32-byte header in front of the records
    count_out = reinterpret_cast<uint64_t*>(sl.rows);
    d_rows = reinterpret_cast<fs_row*>(reinterpret_cast<char*>(sl.rows) + 32);
  }
  // (host rows of the exact pipeline leave the device as 8-byte records: a quarter of the
  // bytes over PCIe; w_rows is sized in fs_row either way)
  static const bool zero_copy = !getenv("FS_HOST_ZEROCOPY") || atoi(getenv("FS_HOST_ZEROCOPY")) != 0;
  sl.host_direct = false;
  if (sl.mode == FS_ROWS_HOST && sl.host_wire8 && zero_copy) {
    // the search's last kernel stores the 8-byte records straight into pinned host memory
    // (2.4 MB per C2 batch over PCIe as the workgroups finish): no copy to queue and wait
    // for behind the search
    const size_t bytes = (size_t)sl.rcap * 8 + 4096;
    if (ix->h_stage_bytes < bytes) {
      if (ix->h_stage) (void)hipHostFree(ix->h_stage);
      ix->h_stage = nullptr; ix->h_stage_bytes = 0; ix->d_stage = nullptr;
      FS_HIP(hipHostMalloc(&ix->h_stage, bytes, hipHostMallocDefault));
      ix->h_stage_bytes = bytes;
    }
    if (!ix->d_stage) FS_HIP(hipHostGetDevicePointer(&ix->d_stage, ix->h_stage, 0));
    d_rows = reinterpret_cast<fs_row*>(ix->d_stage);
    sl.host_direct = true;
  } else if (sl.mode == FS_ROWS_HOST) {
    FS_TRY(ln.w_rows.reserve(sl.host_wire8 ? (sl.rcap + 3) / 4 : sl.rcap));
    d_rows = ln.w_rows.p;
  }
  const int wire = sl.mode == FS_ROWS_DEVICE_PACKED ? 16 : (sl.mode == FS_ROWS_DEVICE_PACKED8 || sl.host_wire8) ? 8 : 0;

  // the status block is cleared by the chain's first kernel (k_reduce) and its final
  // state is written to sl.h_status by the last one (k_rows)
  // timing events only on timed searches: every event record is a barrier packet
  // in the queue (about 1.5 us of bubble each)
  sl.timed = ix->scan_timing_period <= 1 || (ix->searches++ % ix->scan_timing_period) == 0;
  // the whole-search time (fs_stats.total_ms) only for the synchronous call: one more
  // event record, about 4 us of host time per search
  const bool whole = sl.timed && ix->scan_timing_period <= 1 && ix->sync_call;
  if (whole) FS_HIP(hipEventRecord(sl.ev_begin, s));
  sl.whole_timed = whole;
  if (ix->prof.on) fs_prof_mark(ix, s, "<begin>");
  hipEvent_t e0 = sl.timed ? sl.ev_scan0 : nullptr, e1 = sl.timed ? sl.ev_scan1 : nullptr;
  const uint32_t ccap32 = (uint32_t)std::min<uint64_t>(sl.ccap, 0xFFFFFFFFull);
  const uint32_t rcap32 = (uint32_t)std::min<uint64_t>(sl.rcap, 0xFFFFFFFFull);
  bool end_attached = false;           // sl.ev_end rides on the last dispatch (no marker of its own)
  if (sl.exact && sl.fused_waves) {
    // (the whole-search timing of the synchronous call keeps its own end marker)
    FS_TRY(fs_launch_scan_rows(ix, c, sl.fused_waves, sl.fused_blocks, rcap32, d_rows, wire, sl.caprow, sl.h_status, s,
                               e0, e1, count_out, whole || ix->prof.on ? nullptr : sl.ev_end, &end_attached));
    if (ix->prof.on) fs_prof_mark(ix, s, ix->n_lanes > 1 ? "k_scan_rows+k_compact" : fs_search_kernel_name(ix, c));
  } else if (sl.exact) {
    fs_scan_extra ex;
    ex.bsum = ln.w_bsum.p; ex.zero = ln.d_status.p;
    if (sl.capw) { ex.recs = ln.w_recs.p; ex.info = ln.w_info.p; ex.capw = sl.capw; }
    FS_TRY(fs_launch_scan(ix, c->dev(), ln.w_qbm.p, ln.w_qcnt.p, n_bm, s, e0, e1, &ex));
    if (ix->prof.on) fs_prof_mark(ix, s, fs_search_kernel_name(ix, c));
    FS_TRY(fs_launch_post(ix, c, n_bm, sl.tpl, ccap32, rcap32, d_rows, wire, sl.h_status, s, ex, count_out));
    if (ix->prof.on) fs_prof_mark(ix, s, "verify .. k_rows");
  } else {
    // tables whose proof fails by one slot only: the integer prefilter flags the windows
    // that can have a neighbour at all, the LSH work runs on those
    fs_scan_extra ex;
    ex.bsum = ln.w_bsum.p; ex.zero = ln.d_status.p;
    if (sl.caps) {
      // the prefilter and the wildcard filter in one kernel, the survivors in lists per wave
      // range: no bitmap, no k_expand, an eighth of the entries for everything behind
      FS_TRY(ln.w_slist.reserve((size_t)fs_near_ranges() * sl.caps));
      FS_TRY(ln.w_scount.reserve(fs_near_ranges()));
      FS_TRY(fs_launch_near_sift(ix, c, ln.w_slist.p, sl.caps, ln.w_scount.p, ln.w_bsum.p, ln.d_status.p, s, e0, e1));
      if (ix->prof.on) fs_prof_mark(ix, s, fs_search_kernel_name(ix, c));
      const fs_near_lists near{ln.w_slist.p, ln.w_scount.p, sl.caps};
      FS_TRY(fs_launch_lsh_verify(ix, c, ccap32, s, &near));
    } else {
      if (fs_lsh_prefilter_ok(ix, c))
        FS_TRY(fs_launch_scan_near(ix, c, ln.w_qbm.p, ln.w_qcnt.p, n_bm, s, e0, e1, &ex));
      else
        FS_TRY(fs_launch_lsh_scan(ix, c->dev(), ln.w_qbm.p, ln.w_qcnt.p, n_bm, s, e0, e1));
      if (ix->prof.on) fs_prof_mark(ix, s, fs_search_kernel_name(ix, c));
      FS_TRY(fs_launch_expand(ix, c, n_bm, ccap32, sl.tpl, s, ex.counted));
      if (ix->prof.on) fs_prof_mark(ix, s, ex.counted ? "k_expand" : "k_reduce+k_expand");
      FS_TRY(fs_launch_lsh_verify(ix, c, ccap32, s));
    }
    FS_TRY(fs_launch_rows(ix, c, ln.w_cbest.p, 1, ccap32, rcap32, d_rows, 0, sl.h_status, s, count_out));
    if (ix->prof.on) fs_prof_mark(ix, s, "k_hitrows+k_rows");
  }
  ++sl.launches;
  sl.lane_seq = ++ln.enqueued;
  if (!end_attached) FS_HIP(hipEventRecord(sl.ev_end, s));
  ix->cur = &ix->lanes[0];
  return FS_OK;
}

// ---- per-kernel times of one search (diagnostics) ----------------------------------------
int fs_prof_mark(fs_index* ix, hipStream_t s, const char* name) {
  fs_index::Prof& pf = ix->prof;
  if (pf.used == pf.ev.size()) {
    hipEvent_t e;
    FS_HIP(hipEventCreate(&e));
    pf.ev.push_back(e);
  }
  FS_HIP(hipEventRecord(pf.ev[pf.used++], s));
  pf.names.push_back(name);
  return FS_OK;
}

extern "C" int fs_search_corpus(fs_index* ix, fs_corpus* c, fs_row* rows, uint64_t cap, int rows_on_device,
                                uint64_t* n_rows, fs_stats* st);

extern "C" int fs_search_profile(fs_index* ix, fs_corpus* c, fs_row* rows, uint64_t cap, int rows_on_device,
                                 char* names, uint64_t names_cap, double* ms, uint32_t ms_cap, uint32_t* n_out) {
  if (!ix || !c || !names || !ms || !n_out) { fs_set_error("null argument"); return FS_E_INVALID; }
  *n_out = 0;
  if (names_cap) names[0] = 0;
  for (int i = 0; i < FS_SEARCH_SLOTS; ++i)
    if (ix->slots[i].busy) { fs_set_error("fs_search_profile needs the index to itself"); return FS_E_INVALID; }
  FS_ENTER(ix->device);
  fs_index::Prof& pf = ix->prof;
  uint64_t n_rows = 0;
  // (a search that grows a workspace runs its kernels again: the marks of the last run count)
  pf.on = true; pf.used = 0; pf.names.clear();
  const int rc = fs_search_corpus(ix, c, rows, cap, rows_on_device, &n_rows, nullptr);
  pf.on = false;
  if (rc != FS_OK) return rc;
  // marks of the last enqueue: from the last "begin" mark on
  size_t first = 0;
  for (size_t i = 0; i < pf.used; ++i)
    if (pf.names[i][0] == '<') first = i;
  uint32_t n = 0;
  size_t at = 0;
  for (size_t i = first + 1; i < pf.used; ++i) {
    float t = 0;
    FS_HIP(hipEventElapsedTime(&t, pf.ev[i - 1], pf.ev[i]));
    if (n < ms_cap) ms[n] = t;
    const size_t len = strlen(pf.names[i]);
    if (at + len + 2 <= names_cap) {
      memcpy(names + at, pf.names[i], len);
      at += len;
      names[at++] = '\n';
      names[at] = 0;
    }
    ++n;
  }
  *n_out = n;
  return FS_OK;
}

extern "C" int fs_search_corpus_begin(fs_index* ix, fs_corpus* c, fs_row* rows, uint64_t cap,
                                      int rows_mode, uint32_t* ticket) {
  if (!ix || !c || c->ix != ix || !ticket || (cap && !rows)) {
    fs_set_error("null or mismatched handle");
    return FS_E_INVALID;
  }
  const bool header = (rows_mode & FS_ROWS_HEADER) != 0;
  rows_mode &= ~FS_ROWS_HEADER;
  if (header && rows_mode == FS_ROWS_HOST) {
    fs_set_error("FS_ROWS_HEADER goes with the device row modes");
    return FS_E_INVALID;
  }
  if (rows_mode != FS_ROWS_HOST && rows_mode != FS_ROWS_DEVICE &&
      rows_mode != FS_ROWS_DEVICE_PACKED && rows_mode != FS_ROWS_DEVICE_PACKED8) {
    fs_set_error("rows_mode must be FS_ROWS_HOST, FS_ROWS_DEVICE, FS_ROWS_DEVICE_PACKED or "
                 "FS_ROWS_DEVICE_PACKED8");
    return FS_E_INVALID;
  }
  static const bool trace_begin = getenv("FS_TRACE_BEGIN") != nullptr;
  const auto tb0 = std::chrono::steady_clock::now();
  FS_ENTER(ix->device);
  FS_TRY(fs_corpus_update_end(c));          // no-op unless an upload is in flight
  const uint32_t id = ix->next_slot % FS_SEARCH_SLOTS;
  fs_index::Slot& sl = ix->slots[id];
  if (sl.busy) {
    fs_set_error("%d searches already in flight on this index", FS_SEARCH_SLOTS);
    return FS_E_INVALID;
  }
  if (rows_mode == FS_ROWS_HOST)
    for (int i = 0; i < FS_SEARCH_SLOTS; ++i)
      if (ix->slots[i].busy) {
        fs_set_error("host rows need the index to itself: finish the searches in flight first");
        return FS_E_INVALID;
      }
  const uint64_t T = c->n_tok;
  sl.c = c; sl.rows = rows; sl.cap = cap; sl.mode = rows_mode; sl.launches = 0; sl.fallbacks = 0;
  sl.header = header;
  sl.lane = (int)(id % (uint32_t)ix->n_lanes);
  const fs_index::Lane& ln = ix->lanes[sl.lane];
  sl.exact = ix->info.path == FS_MODE_EXACT && !c->has_oov;
  static const bool host8 = !getenv("FS_HOST_WIRE8") || atoi(getenv("FS_HOST_WIRE8")) != 0;
  sl.host_wire8 = rows_mode == FS_ROWS_HOST && sl.exact && ix->n_script < (1ull << 18) && host8;
  // (bitmap layout of the chained kernels: the LSH pipeline's scans write four tokens per lane,
  // k_scan_near8 eight)
  sl.tpl = sl.exact ? fs_scan_tpl(ix, T) : (fs_scan_near8(ix) && fs_lsh_prefilter_ok(ix, c)) ? 8 : 4;
  sl.n_bm = (uint32_t)((T + 64 * sl.tpl - 1) / (64 * sl.tpl));
  if (rows_mode == FS_ROWS_DEVICE_PACKED8 && (!sl.exact || ix->n_script >= (1ull << 18))) {
    fs_set_error("8-byte rows exist for the exact n-gram pipeline and scripts below 2^18 tokens");
    return FS_E_UNSUPPORTED;
  }
  if (rows_mode == FS_ROWS_DEVICE_PACKED && !sl.exact) {
    fs_set_error("packed rows exist for the exact n-gram pipeline only (there the distance is a "
                 "function of the matched script window)");
    return FS_E_UNSUPPORTED;
  }
  // direct path (the scan writes candidate records per wave range): 64 records per
  // range to start with (C2 needs about 20), more once a search has asked for it
  sl.capw = 0;
  if (sl.exact && fs_scan_direct_ok(ix, T)) {
    sl.capw = std::max<uint32_t>(64, ln.capw_hint);
    if (ix->sw.scan_capw > 0) sl.capw = (uint32_t)ix->sw.scan_capw;   // tests: force the growth path
  }
  // k_scan_rows (tokens -> records in one kernel): staged records per wave range, more once
  // a search has asked for it
  sl.caprow = 0;
  sl.fused_waves = sl.exact ? fs_scan_rows_shape(ix, c, &sl.fused_blocks) : 0;
  if (sl.fused_waves) {
    const uint32_t ranges = sl.fused_blocks * sl.fused_waves;
    // staged records per wave range: a sixteenth of its tokens (C2: 305, 72 used on average)
    const uint32_t dflt = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(128, T / ranges / 16), 1u << 20);
    sl.caprow = std::max<uint32_t>(dflt, ln.caprow_hint);
    if (ix->sw.ranges_caprow > 0) sl.caprow = (uint32_t)ix->sw.ranges_caprow;
  }
  if (sl.fused_waves) { sl.capw = 0; sl.tpl = 8; sl.n_bm = (uint32_t)((T + 511) / 512); }
  // k_near_sift: list entries per wave range -- an eighth of its tokens to start with (a C2
  // batch uses 17 of 305 on average at n = 8, 112 over component ids at n = 6), more once a
  // search has asked for it
  sl.caps = 0;
  if (!sl.exact && fs_near_fused(ix, c)) {
    sl.caps = std::max<uint32_t>((uint32_t)std::min<uint64_t>(std::max<uint64_t>(64, T / fs_near_ranges() / 8), 1u << 20),
                                 ln.caps_hint);
    if (ix->sw.scan_capw > 0) sl.caps = (uint32_t)ix->sw.scan_capw;   // tests: force the growth path
  }
  // capacities: grown from the device totals when a stage overflows
  sl.ccap = std::max<uint64_t>(std::max<uint64_t>(4096, T / 16), ln.w_cpos.n);
  sl.rcap = rows_mode != FS_ROWS_HOST
                ? cap
                : std::max<uint64_t>(std::max<uint64_t>(4096, T / 16), ln.w_rows.n);
  const auto tb1 = std::chrono::steady_clock::now();
  FS_TRY(search_enqueue(ix, sl));
  if (trace_begin) {
    const auto tb2 = std::chrono::steady_clock::now();
    fprintf(stderr, "fs_search_corpus_begin: set-up %.1f us, enqueue %.1f us\n",
            std::chrono::duration<double, std::micro>(tb1 - tb0).count(),
            std::chrono::duration<double, std::micro>(tb2 - tb1).count());
  }
  sl.busy = true;
  *ticket = id;
  ++ix->next_slot;
  return FS_OK;
}

// 8-byte wire records {token position, orig_ix | k << 18 | lev << 22} (ascending positions) ->
// A few threads that stay with the index (host rows are expanded once per search: starting
// eight threads per call cost more than their work).  A worker that has finished keeps
// polling for about 0.2 ms before it sleeps, so back-to-back searches do not pay a wake-up.
struct fs_host_pool {
  std::vector<std::thread> threads;
  std::mutex mu;
  std::condition_variable cv;
  std::function<void(unsigned)> job;      // called with the worker's number 1 .. n
  std::atomic<uint64_t> generation{0};
  std::atomic<unsigned> pending{0};
  bool stop = false;
  void worker(unsigned id) {
    uint64_t seen = 0;
    for (;;) {
      // poll, then sleep
      bool got = false;
      const auto t0 = std::chrono::steady_clock::now();
      while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(200)) {
        if (generation.load(std::memory_order_acquire) != seen) { got = true; break; }
        __builtin_ia32_pause();
      }
      if (!got) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || generation.load(std::memory_order_acquire) != seen; });
        if (stop) return;
      }
      seen = generation.load(std::memory_order_acquire);
      job(id);
      pending.fetch_sub(1, std::memory_order_acq_rel);
    }
  }
  void start(unsigned n) {
    for (unsigned i = 1; i <= n; ++i) threads.emplace_back([this, i] { worker(i); });
  }
  void run(const std::function<void(unsigned)>& f) {      // f(0) here, f(1..n) on the workers
    job = f;
    pending.store((unsigned)threads.size(), std::memory_order_release);
    {
      std::lock_guard<std::mutex> lk(mu);
      generation.fetch_add(1, std::memory_order_acq_rel);
    }
    cv.notify_all();
    f(0);
    while (pending.load(std::memory_order_acquire)) __builtin_ia32_pause();
  }
  ~fs_host_pool() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv.notify_all();
    for (std::thread& t : threads) t.join();
  }
};

void fs_host_pool_free(fs_host_pool* p) { delete p; }

// fs_row on the host, as fs_rows_unpack8 does on the device: the work of a record by a walk
// along the work offsets (a new bisection where a record lies in front of the one before it), dist = the matched script window's distance to itself, comb =
// dist * lev (one IEEE multiplication, as __dmul_rn).  A few threads, each over a slice.
static void fs_expand_rows8_host(fs_index* ix, const uint32_t* rec, uint64_t n, const uint64_t* work_off,
                                 uint64_t n_works, const double* selfdist, fs_row* rows) {
  auto slice = [=](uint64_t lo, uint64_t hi) {
    if (lo >= hi) return;
    // the work of the slice's first record: last w with work_off[w] <= position
    uint64_t a = 0, b = n_works;
    const uint64_t x0 = rec[2 * lo];
    while (b - a > 1) {
      const uint64_t mid = a + ((b - a) >> 1);
      if (work_off[mid] <= x0) a = mid; else b = mid;
    }
    uint64_t w = a;
    for (uint64_t i = lo; i < hi; ++i) {
      const uint32_t x = rec[2 * i], y = rec[2 * i + 1];
      if (x < work_off[w]) {
        // a record behind its successor (the search's own records ascend; a caller's need
        // not): look its work up from scratch, as k_unpack8 does for every record
        uint64_t l = 0, r = w;
        while (r - l > 1) {
          const uint64_t mid = l + ((r - l) >> 1);
          if (work_off[mid] <= x) l = mid; else r = mid;
        }
        w = l;
      }
      while (w + 1 < n_works && work_off[w + 1] <= x) ++w;
      const uint32_t orig = y & 0x3FFFFu, k = (y >> 18) & 0xFu, lev = y >> 22;
      fs_row r;
      r.work = (uint32_t)w; r.fan_ix = (uint32_t)(x - work_off[w]); r.orig_ix = orig; r.lev = lev;
      r.dist = selfdist[orig - k];
      r.comb = r.dist * (double)lev;
      rows[i] = r;
    }
  };
  if (n < 32768) { slice(0, n); return; }
  if (!ix->host_pool) {
    // threads for the expansion: this process's share of the cores it may run on (its
    // affinity mask, divided by the ranks of the node: every rank has such a pool, next to
    // its tokeniser processes), at most 12
    unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) hw = std::max(1, CPU_COUNT(&set));
    if (const char* e = getenv("LOCAL_WORLD_SIZE")) hw = std::max(1u, hw / (unsigned)std::max(1, atoi(e)));
    else if (const char* e2 = getenv("WORLD_SIZE")) hw = std::max(1u, hw / (unsigned)std::max(1, atoi(e2)));
    ix->host_pool = new fs_host_pool;
    ix->host_pool->start(std::min(12u, hw) - 1);
  }
  const uint64_t nt = ix->host_pool->threads.size() + 1;
  ix->host_pool->run([&](unsigned t) { slice(n * t / nt, n * (t + 1) / nt); });
}

extern "C" int fs_search_corpus_end(fs_index* ix, uint32_t ticket, uint64_t* n_rows, fs_stats* st) {
  if (!ix || ticket >= FS_SEARCH_SLOTS || !n_rows || !ix->slots[ticket].busy) {
    fs_set_error("no such search in flight");
    return FS_E_INVALID;
  }
  FS_ENTER(ix->device);
  fs_index::Slot& sl = ix->slots[ticket];
  sl.busy = false;
  if (!sl.c) { fs_set_error("the corpus of this search has been destroyed"); return FS_E_INVALID; }
  float scan_ms = 0, total_ms = 0;
  for (int attempt = 0;; ++attempt) {
    if (sl.mode == FS_ROWS_HOST) {
      // the caller waits for the rows right here: poll for a while (a blocking wait wakes up
      // some 20 us after the event), then block
      const auto t0 = std::chrono::steady_clock::now();
      while (hipEventQuery(sl.ev_end) == hipErrorNotReady &&
             std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(2))
        __builtin_ia32_pause();
    }
    FS_HIP(hipEventSynchronize(sl.ev_end));
    // nothing else is queued on the lane: one query lets the runtime see that the stream has
    // run dry.  A device-wide synchronize (the caller's, at the end of a run) otherwise sends a
    // marker down every stream that has launched anything and waits for each in turn -- measured
    // 43-53 us for four idle lanes, 7 us after the queries; the query is < 1 us of host time.
    if (ix->sw.end_query && ix->lanes[sl.lane].enqueued == sl.lane_seq)
      (void)hipStreamQuery(ix->lanes[sl.lane].stream);
    scan_ms = 0;
    if (sl.n_bm && sl.timed) FS_HIP(hipEventElapsedTime(&scan_ms, sl.ev_scan0, sl.ev_scan1));
    total_ms = 0;
    if (sl.whole_timed) FS_HIP(hipEventElapsedTime(&total_ms, sl.ev_begin, sl.ev_end));
    const fs_status& hs = *sl.h_status;
    if (!sl.exact && ((ix->sw.lsh_diag & 0x780000) || hs.max_rows >= 0x70000000u)) fprintf(stderr, "lsh diag: %u (0x%x) bucket members / code for %u pending windows\n", hs.max_rows, hs.max_rows, hs.lsh_pending);
    if (hs.bad_string) { fs_set_error("fan string id outside the string table"); return FS_E_INVALID; }
    if (hs.lev_overflow) {
      fs_set_error("an n-gram text exceeds %d code points", FS_LEV_MAX);
      return FS_E_UNSUPPORTED;
    }
    bool again = false;
    uint32_t* gave_up = reinterpret_cast<uint32_t*>(sl.h_status + 1);
    if (*gave_up) {
      // finish_rows: a workgroup's wait for the workgroups in front of it ran out.  Run the
      // search again through the chained kernels (no hand-off inside a launch).
      *gave_up = 0;
      gave_up[1] = 0;
      if (!sl.caprow) { fs_set_error("in-launch wait flagged on a search without one"); return FS_E_DEVICE; }
      const uint64_t T = sl.c->n_tok;
      sl.caprow = 0; sl.fused_waves = 0;
      sl.tpl = fs_scan_tpl(ix, T);
      sl.n_bm = (uint32_t)((T + 64 * sl.tpl - 1) / (64 * sl.tpl));
      sl.capw = fs_scan_direct_ok(ix, T) ? std::max<uint32_t>(64, ix->lanes[sl.lane].capw_hint) : 0;
      sl.ccap = std::max<uint64_t>(std::max<uint64_t>(4096, T / 16), ix->lanes[sl.lane].w_cpos.n);
      ix->wait_fallbacks++;
      sl.fallbacks++;
      FS_TRY(search_enqueue(ix, sl));
      continue;
    }
    // shared rounds: a workgroup's pool for the slices' records was too small (the word behind
    // the give-up word holds what one of them needed)
    if (gave_up[1]) {
      fs_index::Lane& pl = ix->lanes[sl.lane];
      pl.xpool_hint = std::max<uint32_t>(2 * pl.xpool, gave_up[1] + gave_up[1] / 4 + 64);
      gave_up[1] = 0;
      again = true;
    }
    if (!sl.caprow && hs.n_cands > sl.ccap) { sl.ccap = (uint64_t)hs.n_cands + hs.n_cands / 8; again = true; }
    if (sl.caprow && hs.max_rows > sl.caprow) {
      sl.caprow = (hs.max_rows + hs.max_rows / 4 + 7) & ~7u;
      ix->lanes[sl.lane].caprow_hint = sl.caprow;
      again = true;
    }
    if (sl.caps && hs.max_recs > sl.caps) {
      sl.caps = (hs.max_recs + hs.max_recs / 4 + 7) & ~7u;
      ix->lanes[sl.lane].caps_hint = sl.caps;
      again = true;
    }
    if (sl.capw && hs.max_recs > sl.capw) {
      sl.capw = (hs.max_recs + hs.max_recs / 4 + 7) & ~7u;
      ix->lanes[sl.lane].capw_hint = sl.capw;
      again = true;
    }
    else if (sl.mode == FS_ROWS_HOST && hs.n_rows > sl.rcap && hs.n_rows <= sl.cap) {
      sl.rcap = hs.n_rows; again = true;
    }
    if (!again) break;
    if (attempt == 7) { fs_set_error("workspace growth did not converge"); return FS_E_DEVICE; }
    // a workspace was too small: grow it (the reallocation waits for everything
    // queued so far) and run this search again
    FS_TRY(search_enqueue(ix, sl));
  }
  const fs_status hs = *sl.h_status;
  *n_rows = hs.n_rows;
  if (st) {
    memset(st, 0, sizeof *st);
    st->windows_processed = sl.c->windows;
    st->candidates = hs.n_cands;
    st->matches = hs.n_matches;
    st->rows = hs.n_rows;
    st->scan_ms = scan_ms;
    st->total_ms = total_ms;
    st->path = sl.exact ? FS_MODE_EXACT : FS_MODE_GENERAL;
    st->scan_launches = sl.launches;
    st->lsh_pending = sl.exact ? 0 : hs.lsh_pending;
    st->handoff_fallbacks = sl.fallbacks;
  }
  if (!sl.exact) ix->lanes[sl.lane].pend_hint = hs.lsh_pending;
  if (hs.n_rows > sl.cap) return FS_E_CAPACITY;
  if (sl.mode == FS_ROWS_HOST && hs.n_rows) {
    fs_index::Lane& ln = ix->lanes[sl.lane];
    if (sl.host_wire8) {
      // 8-byte records into pinned memory (a quarter of fs_row over PCIe, and no pageable
      // landing buffer in the copy's way), then fs_row on the host cores
      const size_t bytes = (size_t)hs.n_rows * 8;
      if (!sl.host_direct) {
        if (ix->h_stage_bytes < bytes) {
          if (ix->h_stage) (void)hipHostFree(ix->h_stage);
          ix->h_stage = nullptr; ix->h_stage_bytes = 0; ix->d_stage = nullptr;
          FS_HIP(hipHostMalloc(&ix->h_stage, bytes + bytes / 4 + 4096, hipHostMallocDefault));
          ix->h_stage_bytes = bytes + bytes / 4 + 4096;
        }
        FS_HIP(hipMemcpyAsync(ix->h_stage, ln.w_rows.p, bytes, hipMemcpyDeviceToHost, ln.stream));
        FS_HIP(hipStreamSynchronize(ln.stream));
      }
      fs_expand_rows8_host(ix, reinterpret_cast<const uint32_t*>(ix->h_stage), hs.n_rows, sl.c->h_work_off.data(),
                           sl.c->n_works, ix->h_selfdist.data(), sl.rows);
    } else {
      FS_HIP(hipMemcpyAsync(sl.rows, ln.w_rows.p, (size_t)hs.n_rows * sizeof(fs_row),
                            hipMemcpyDeviceToHost, ln.stream));
      FS_HIP(hipStreamSynchronize(ln.stream));
    }
  }
  return FS_OK;
}

extern "C" int fs_search_corpus(fs_index* ix, fs_corpus* c, fs_row* rows, uint64_t cap,
                                int rows_on_device, uint64_t* n_rows, fs_stats* st) {
  if (!n_rows) { fs_set_error("null argument"); return FS_E_INVALID; }
  uint32_t ticket = 0;
  if (!ix) { fs_set_error("null argument"); return FS_E_INVALID; }
  ix->sync_call = true;
  const int rc = fs_search_corpus_begin(ix, c, rows, cap, rows_on_device, &ticket);
  ix->sync_call = false;
  if (rc != FS_OK) return rc;
  return fs_search_corpus_end(ix, ticket, n_rows, st);
}

extern "C" int fs_search(fs_index* ix, const uint32_t* tok_vec, const uint32_t* tok_str,
                         const uint64_t* work_off, uint64_t n_works, const uint32_t* str_chars,
                         const uint64_t* str_off, uint64_t n_str, fs_row* rows, uint64_t cap,
                         uint64_t* n_rows, fs_stats* st) {
  fs_corpus* c = nullptr;
  FS_TRY(fs_corpus_create(ix, tok_vec, tok_str, work_off, n_works, str_chars, str_off, n_str, &c));
  const int rc = fs_search_corpus(ix, c, rows, cap, 0, n_rows, st);
  fs_corpus_destroy(c);
  return rc;
}

// Diagnostics (FS_DIAG & 2): the timeline stamps the last k_scan_rows launch of lane
// `lane` left behind, eight words per wave range (fs_scan.hip).
extern "C" int fs_debug_stamps(fs_index* ix, uint32_t lane, uint64_t* out, uint64_t cap, uint64_t* n) {
  if (!ix || !n || lane >= FS_LANES) return FS_E_INVALID;
  FS_ENTER(ix->device);
  fs_index::Lane& ln = ix->lanes[lane];
  FS_HIP(hipStreamSynchronize(ln.stream));
  *n = ln.dbg_words;
  if (!out || cap < ln.dbg_words || !ln.dbg_words) return ln.dbg_words && out ? FS_E_CAPACITY : FS_OK;
  FS_HIP(hipMemcpy(out, ln.w_dbg.p, ln.dbg_words * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return FS_OK;
}

extern "C" int fs_scan_benchmark(fs_index* ix, fs_corpus* c, uint32_t reps, double* avg_ms) {
  if (!ix || !c || c->ix != ix || !avg_ms || reps == 0) return FS_E_INVALID;
  FS_ENTER(ix->device);
  hipStream_t s = ix->stream;
  const int tpl = fs_scan_tpl(ix, c->n_tok);
  const uint32_t n_bm = (uint32_t)((c->n_tok + 64 * tpl - 1) / (64 * tpl));
  FS_TRY(ix->cur->w_qbm.reserve((size_t)n_bm * tpl));
  FS_TRY(ix->cur->w_qcnt.reserve(n_bm));
  FS_TRY(fs_launch_scan(ix, c->dev(), ix->cur->w_qbm.p, ix->cur->w_qcnt.p, n_bm, s));   // warm
  FS_HIP(hipEventRecord(ix->ev_scan0, s));
  for (uint32_t r = 0; r < reps; ++r)
    FS_TRY(fs_launch_scan(ix, c->dev(), ix->cur->w_qbm.p, ix->cur->w_qcnt.p, n_bm, s));
  FS_HIP(hipEventRecord(ix->ev_scan1, s));
  FS_HIP(hipStreamSynchronize(s));
  float ms = 0;
  FS_HIP(hipEventElapsedTime(&ms, ix->ev_scan0, ix->ev_scan1));
  *avg_ms = (double)ms / reps;
  return FS_OK;
}

extern "C" int fs_stream_floor(fs_index* ix, fs_corpus* c, uint32_t reps, double* avg_ms) {
  if (!ix || !c || c->ix != ix || !avg_ms || reps == 0) return FS_E_INVALID;
  FS_ENTER(ix->device);
  return fs_launch_stream_floor(ix, c, reps, avg_ms);
}

extern "C" int fs_rows_unpack8(fs_index* ix, const void* packed, uint64_t n,
                               const uint64_t* work_off, uint64_t n_works, fs_row* rows) {
  if (!ix || (n && (!packed || !rows || !work_off || !n_works))) { fs_set_error("null argument"); return FS_E_INVALID; }
  if (ix->info.path != FS_MODE_EXACT || ix->n_script >= (1ull << 18)) {
    fs_set_error("8-byte rows exist for the exact n-gram pipeline and scripts below 2^18 tokens");
    return FS_E_UNSUPPORTED;
  }
  FS_ENTER(ix->device);
  FS_TRY(fs_launch_unpack8(ix, packed, n, work_off, n_works, rows, ix->stream));
  FS_HIP(hipStreamSynchronize(ix->stream));
  return FS_OK;
}

extern "C" int fs_rows_unpack(fs_index* ix, const void* packed, uint64_t n, fs_row* rows) {
  if (!ix || (n && (!packed || !rows))) return FS_E_INVALID;
  FS_ENTER(ix->device);
  FS_TRY(fs_launch_unpack(ix, packed, n, rows, ix->stream));
  FS_HIP(hipStreamSynchronize(ix->stream));
  return FS_OK;
}

extern "C" int fs_reuse_histogram(int device, const uint32_t* orig_ix, const double* comb,
                                  uint64_t n_rows, uint64_t n_script, const double* thresholds,
                                  uint32_t n_thr, uint32_t* counts) {
  if ((n_rows && (!orig_ix || !comb)) || !thresholds || !n_thr || n_thr > 64 ||
      (n_script && !counts)) {
    fs_set_error("null argument or n_thr outside 1..64");
    return FS_E_INVALID;
  }
  for (uint32_t t = 1; t < n_thr; ++t)
    if (!(thresholds[t - 1] <= thresholds[t])) { fs_set_error("thresholds must ascend"); return FS_E_INVALID; }
  FS_ENTER(device);
  DBuf<uint32_t> d_orig, d_counts;
  DBuf<double> d_comb, d_thr;
  FS_TRY(d_orig.upload(orig_ix, n_rows, nullptr));
  FS_TRY(d_comb.upload(comb, n_rows, nullptr));
  FS_TRY(d_thr.upload(thresholds, n_thr, nullptr));
  FS_TRY(d_counts.reserve(n_script * (n_thr + 1)));
  FS_TRY(fs_launch_histogram(d_orig.p, d_comb.p, nullptr, n_rows, n_script, d_thr.p, n_thr,
                             d_counts.p, nullptr));
  if (n_script)
    FS_HIP(hipMemcpy(counts, d_counts.p, n_script * (n_thr + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
  FS_HIP(hipDeviceSynchronize());
  return FS_OK;
}

extern "C" int fs_reuse_histogram_rows(fs_index* ix, const fs_row* d_rows, uint64_t n_rows,
                                       const double* thresholds, uint32_t n_thr, uint32_t* d_counts) {
  if (!ix || (n_rows && !d_rows) || !thresholds || !n_thr || n_thr > 64 || !d_counts) return FS_E_INVALID;
  for (uint32_t t = 1; t < n_thr; ++t)
    if (!(thresholds[t - 1] <= thresholds[t])) { fs_set_error("thresholds must ascend"); return FS_E_INVALID; }
  FS_ENTER(ix->device);
  DBuf<double> d_thr;
  FS_TRY(d_thr.upload(thresholds, n_thr, ix->stream));
  FS_TRY(fs_launch_histogram(nullptr, nullptr, d_rows, n_rows, ix->n_script, d_thr.p, n_thr, d_counts,
                             ix->stream));
  FS_HIP(hipStreamSynchronize(ix->stream));
  return FS_OK;
}

extern "C" int fs_index_set_scan_timing(fs_index* ix, uint32_t period) {
  if (!ix) return FS_E_INVALID;
  ix->scan_timing_period = period ? period : 1;
  return FS_OK;
}
