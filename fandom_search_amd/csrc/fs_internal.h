// fs_internal.h -- shared declarations of libfandomsearch_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>

#include "../../include/fandom_search.h"
#include "fs_hash.h"

#define FS_MAX_WINDOW 16          // n <= 16: at most 4 neighbour vectors of halo
#define FS_NONE 0xFFFFFFFFu
#define FS_LANES 4                // streams (with workspaces) searches are spread over; see fs_index::Lane
#define FS_LANES_DEFAULT 1
#define FS_CHUNKS 2048             // = fsdev::kNB: chunks of the chained kernels (and partial sums)
#define FS_SEARCH_SLOTS 4         // searches that may be in flight on one index
#define FS_LEV_MAX 512            // code points per side handled by lev_device
#define FS_WAIT_GAVE_UP 0x80000000u   // fs_status.lev_overflow: an in-kernel wait gave up (finish_rows)
#define FS_SYNC_BLOCKS 1024       // workgroups a records kernel may have (granules per lane)

void fs_set_error(const char* fmt, ...);

#define FS_HIP(call)                                                        \
  do {                                                                      \
    hipError_t e_ = (call);                                                 \
    if (e_ != hipSuccess) {                                                 \
      fs_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call,             \
                   hipGetErrorString(e_));                                  \
      return FS_E_DEVICE;                                                   \
    }                                                                       \
  } while (0)

// At an entry point: select the device and drop an error another user of the same HIP
// runtime (PyTorch in this process) may have left behind -- hipGetLastError() is how
// launches are checked here, and it reports the thread's last error whoever caused it.
#define FS_ENTER(dev)                 \
  do {                                \
    FS_HIP(hipSetDevice(dev));        \
    (void)hipGetLastError();          \
  } while (0)

#define FS_TRY(call)                \
  do {                              \
    int rc_ = (call);               \
    if (rc_ != FS_OK) return rc_;   \
  } while (0)

// device buffer that only ever grows
template <class T>
struct DBuf {
  T* p = nullptr;
  size_t n = 0;
  int reserve(size_t count) {
    if (count <= n && p) return FS_OK;
    if (p) (void)hipFree(p);
    p = nullptr; n = 0;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
    if (e != hipSuccess) {
      fs_set_error("hipMalloc(%zu bytes) -> %s", count * sizeof(T), hipGetErrorString(e));
      p = nullptr;
      return FS_E_NOMEM;
    }
    n = count;
    return FS_OK;
  }
  int upload(const T* host, size_t count, hipStream_t s) {
    FS_TRY(reserve(count));
    if (count) FS_HIP(hipMemcpyAsync(p, host, count * sizeof(T), hipMemcpyHostToDevice, s));
    return FS_OK;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
  ~DBuf() { release(); }
  DBuf() = default;
  DBuf(const DBuf&) = delete;
  DBuf& operator=(const DBuf&) = delete;
};

// Device-side totals and overflow flags of one search; read back in one copy.
struct fs_status {
  uint32_t n_cands;      // filter-positive windows flagged by the scan
  uint32_t n_hits;       // verified (window, n-gram) hits inside a work
  uint32_t n_matches;    // (window, script window) pairs
  uint32_t n_rows;       // output records
  uint32_t max_recs;     // direct path: largest record count of a wave range beyond its capacity
  uint32_t lev_overflow; // a Levenshtein operand exceeded FS_LEV_MAX
  uint32_t bad_string;   // string id outside the string table
  uint32_t max_rows;     // k_scan_rows: largest record count of a wave range beyond its staging capacity
  uint32_t lsh_pending;  // general pipeline: candidate windows k_lsh_sift left to k_lsh_verify (a wave each)
};

// what one hit offers to the fan words of its window (first-minimum rank)
struct alignas(16) fs_best {
  uint32_t s;      // script position of the chosen rank
  uint32_t lev;
  double dist;
  double comb;
  double pad;
};

// Script-side exact n-gram index (device pointers + geometry), passed by value
// to kernels.
struct GramIndexDev {
  const uint32_t* stok;      // [n_script] script vector ids
  const uint32_t* filter;    // [1 << log2_words] blocked Bloom filter
  const uint32_t* sfilter;   // [1 << log2_words] one bit per script K-gram (fs_hash.h), or nullptr
  const uint32_t* table;     // [1 << log2_slots][tstride] {gram id + 1 (0 = empty), kept occurrences, the n vector ids, pad}
  const uint32_t* gpos;      // [n_grams][nn] first <= nn script positions, ascending
  const uint32_t* gcnt;      // [n_grams] min(occurrences, nn)
  const double*   selfdist;  // [n_windows] canonical distance of a window to itself
  const uint32_t* schars;    // script word text
  const uint64_t* soff;      // [n_script + 1]
  int log2_words;
  int log2_swords;           // of sfilter
  int log2_slots;
  int tstride;               // words per table entry: 2 + n rounded up to a multiple of 4
  const uint32_t* disp;      // [1 << log2_buckets] displacement seed per bucket (fs_hash.h)
  int log2_buckets;
  const uint8_t* disp8;      // the seeds as bytes (FS_DISP8_WIDE: see disp), or nullptr
  int n;                     // window size
  int nn;                    // NearestFilter N
  uint32_t n_grams;
};

struct CorpusDev {
  const uint32_t* tok;       // [n_tok + pad] vector ids
  const uint32_t* str;       // [n_tok] string ids or nullptr (== vector ids)
  const uint64_t* work_off;  // [n_works + 1]
  const uint2* blk_work;     // [ceil(n_tok / 256)] {work of token 256*i, end of that work}
  const uint4* blk4;         // same blocks: {work w, start of w, end of w, end of w + 1}
  const uint4* ctab;         // [1 << log2_slots][4] batch table of k_scan_rows (fs_hash.h), or nullptr
  const uint32_t* chars;     // fan-side string table
  const uint64_t* coff;      // [n_str + 1]
  uint32_t n_tok;
  uint32_t n_works;
  uint32_t n_str;
};

// What the lane-per-pair Levenshtein needs (fs_post.hip, k_strbest)
struct StrFast {
  const unsigned long long* pat;   // fs_index::d_pat
  const uint32_t* clsmap;
  uint32_t n_cls;
  uint32_t punct;
  const uint4* strrec;             // fs_corpus::d_strrec
};

// LSH pipeline: per script window, everything the first slot of a candidate's
// distance needs, in one 32-byte record
struct alignas(32) fs_swin {
  double ss;       // sum of q over the window's slots
  double rss;      // sqrt(ss), correctly rounded
  double qu0;      // q of the first slot's vector
  uint32_t u0;     // first slot's vector id
  int32_t r0;      // its row in the pair table, -1: none
};

// what the canonical distance needs of one script token: a 16-byte read per slot
struct alignas(16) fs_spos {
  double q;        // q of the token's vector
  int32_t row;     // its row in the pair table, -1: none
  uint32_t id;     // its vector id
};

struct fs_corpus;

// Diagnostic switches (FS_* environment variables), read once at fs_index_create and
// again only on fs_index_reload_switches: nothing on the per-search path calls getenv.
struct fs_switches {
  char scan_flags = 0;            // FS_SCAN_FLAGS: 'n' non-temporal id loads, other: default policy
  bool scan_simple = false;       // FS_SCAN_VARIANT=simple
  int scan_tpl = 0;               // FS_SCAN_TPL=4: the chained kernels scan with k_scan_simple
  bool scan_direct = true;        // FS_SCAN_DIRECT=0: bitmap + k_expand instead of candidate records
  int scan_capw = 0;              // FS_SCAN_CAPW: records per wave range to start with (tests)
  bool scan_rows = true;          // FS_SCAN_ROWS=0: separate scan and post-scan kernels
  bool scan_sub = true;           // FS_SCAN_SUB=0: k_scan_rows tests the full n-gram (Bloom) instead of K-gram runs
  int diag = 0;                   // FS_DIAG bits: 1 k_scan_rows without its rounds (results invalid), 2 in-kernel
                                  // timeline stamps (fs_debug_stamps), 16 equal shares per wave, bits 8..: flush threshold
  int wait_spins = -1;            // FS_WAIT_SPINS: polls before finish_rows gives up (tests: 0)
  double lsh_f32_slack = 1.0;     // FS_LSH_F32_SLACK: factor on the float32 key bound (tests force the fallback)
  bool lsh_f32 = true;            // FS_LSH_F32=0: float64 keys only
  int lsh_diag = 0;               // FS_LSH_DIAG
  int lsh_lev_lane = 1;           // FS_LSH_LEV_LANE=0: the kept matches' Levenshtein distances a wave per match inside k_lsh_verify; 2: a lane per match (k_lsh_lev) however few windows are pending
  bool scan_near8 = true;         // FS_SCAN_NEAR8=0: k_scan_near (four tokens per lane) also for n >= 7; read when the index is built
  bool end_query = true;          // FS_END_QUERY=0: fs_search_corpus_end does not poll its lane's stream
  bool lsh_no_gtab = false;       // FS_LSH_NO_GTAB
  bool lsh_serial = false;        // FS_LSH_SERIAL: neighbour lists on one lane (cross-check of the wave form)
  bool lsh_prefilter = true;      // FS_LSH_PREFILTER=0: always the full key scan
  bool rows_disp_lds = true;      // FS_ROWS_DISP_LDS=0: k_scan_rows reads the displacement seeds from memory (as with > 16 K buckets)
  bool str_fused = true;          // FS_STR_FUSED=0: batches with string ids take the chained kernels instead of k_scan_rows with per-hit Levenshtein
  bool str_fast = true;           // FS_STR_FAST=0: batches with string ids take k_matchlev + k_cbest (a wave per pair) instead of k_strbest
  bool str_levtab = true;         // FS_STR_LEVTAB=0: batches with string ids compute every Levenshtein distance per match
  bool lsh_selflev = true;        // FS_LSH_SELFLEV=0: every Levenshtein distance of the LSH pipeline computed per match
  bool lsh_wild = true;           // FS_LSH_WILD=0: no wildcard-key filter in front of k_lsh_verify
  bool lsh_wmap = true;           // FS_LSH_WMAP=0: windows one slot away from a script n-gram always take the full LSH path
  bool lsh_keys6 = true;          // FS_LSH_KEYS6=0: n = 6 over component ids without the middle-slot key filter in k_scan_near
  bool lsh_syn = true;            // FS_LSH_SYN=0: no component-id prefilter for tables with near-synonyms
  int lsh_share = 35;             // FS_LSH_SHARE: the share rule (fs_lsh.hip) on tables no integer prefilter applies to; bit 5 k_share_scan -- the script windows behind a window's keys instead of the key scan --, else bit 0 a gate kernel in front of k_lsh_scan (k_share_gate) and bit 1 the pairs' test inside it; bit 2 the gate asks for every heavy subset (the script's filter holds its heavy subsets only; n <= 6, without bit 5), bit 3 out-of-vocabulary fan tokens count as possibly near; 0: off.  Read when the index is built
  double share_gamma = 0.7;       // FS_SHARE_GAMMA: cosine above which two vectors are near in the share rule
  bool lsh_emap = true;           // FS_LSH_EMAP=0: k_lsh_batch walks the buckets of every pending window instead of enumerating the script n-grams one slot away
  bool lsh_batch = true;          // FS_LSH_BATCH=0: the pending windows a wave each (k_lsh_verify) instead of eight per wave level by level (k_lsh_batch)
  int lsh_defer_min = 8192;       // FS_LSH_DEFER_MIN: pending windows of the lane's last search from which on the kept matches' Levenshtein distances go to k_lsh_lev (and the windows to k_lsh_batch)
  bool near_fused = true;         // FS_NEAR_FUSED=0: k_scan_near8 + k_expand + k_lsh_sift (round 4's chain) instead of k_near_sift + k_lsh_sift2; read when the index is built (n = 6: which prefilter kernel the 3-gram filter is laid out for)
  bool lsh_gramtab = true;        // FS_LSH_GRAMTAB=0: no per-n-gram records (k_lsh_gramtab): every window with a script n-gram's ids walks the buckets
  int rows_waves = 0;             // FS_ROWS_WAVES: waves per workgroup of k_scan_rows (experiments)
  int rows_blocks_per_cu = 0;     // FS_ROWS_BLOCKS_PER_CU: workgroups of k_scan_rows per CU (experiments)
  int rows_finish = 0;            // FS_ROWS_FINISH: 1 inside the launch, 2 k_compact, 0: by number of lanes
  int rows_shares[4] = {0, 0, 0, 0};   // FS_ROWS_SHARES=a,b,c,d: scan shares of a SIMD's waves by slot age, sum 1024
  bool rows_coop = true;          // FS_ROWS_COOP=0: no shared rounds (every wave works off its own range's candidates)
  int rows_xpool = 0;             // FS_ROWS_XPOOL: records in a workgroup's pool for the shared rounds (tests: force the growth)
  int ranges_caprow = 0;          // FS_RANGES_CAPROW: staged records per wave range of k_scan_rows to start with (tests)
};
void fs_read_switches(fs_switches* sw);

struct fs_index {
  fs_config cfg;
  fs_switches sw;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_scan0 = nullptr, ev_scan1 = nullptr;   // fs_scan_benchmark
  fs_index_info info{};
  uint64_t n_script = 0, n_windows = 0, n_vec = 0;
  uint32_t n_grams = 0;
  int log2_words = 0, log2_slots = 0, log2_buckets = 0, log2_swords = 0;
  bool ctab_ok = false;                // d_cproto / d_disp8 built (window sizes k_scan_rows takes)
  int num_cu = 256;

  DBuf<uint32_t> d_stok, d_filter, d_sfilter, d_table, d_disp, d_gpos, d_gcnt, d_schars, d_cproto, d_disp8;
  DBuf<uint64_t> d_soff;
  // lane-per-pair Levenshtein of batches with string ids (k_strbest): script characters as classes
  DBuf<uint32_t> d_clsmap;             // sorted distinct code points of the script text (and ' '): class = index + 1
  DBuf<unsigned long long> d_pat;      // [n_windows][8]: 7 bit planes of the window text's classes, its length (> 64: none)
  uint32_t n_cls = 0, str_punct = 0;   // classes of '[', ',', ' ', ']' (a byte each)
  bool strfast_ok = false;
  DBuf<double> d_q, d_selfdist;
  std::vector<double> h_selfdist;   // host copy (host-row searches expand 8-byte records on the host)
  void* h_stage = nullptr;          // pinned landing buffer of those records
  size_t h_stage_bytes = 0;
  void* d_stage = nullptr;          // the same buffer as the device sees it (records stored there by the search itself)
  struct fs_host_pool* host_pool = nullptr;   // threads that expand host rows (fs_api.hip)
  DBuf<float> d_emb;
  DBuf<double> d_normals;

  // general (LSH) pipeline, built on demand (fs_lsh_build)
  DBuf<double> d_nt, d_atab, d_ss, d_gtab;
  DBuf<fs_swin> d_sw;            // per script window: what a bucket candidate's first slot needs
  DBuf<int32_t> d_sidx;
  DBuf<fs_spos> d_spos;      // per script token (k_spos)
  DBuf<uint32_t> d_skeys;    // [W][H] LSH keys of the script windows
  DBuf<uint32_t> d_emap, d_emapc;   // wildcard keys -> distinct script n-gram, over vector ids / component ids (build_emap)
  int log2_emap = 0, log2_emapc = 0;
  DBuf<float> d_atab32, d_amax, d_nt32, d_ntmax;
  DBuf<uint32_t> d_boff, d_bids;
  bool lsh_ready = false;
  double lsh_cmax = 1.0;     // sound bound on the cosine of two distinct table vectors
  bool script_oov = false;   // the script holds out-of-vocabulary (3-hot) vectors
  int lsh_m_min = 0;         // fewer id-identical slots than this cannot reach the threshold
  DBuf<uint32_t> d_sfilter3; // one bit per script 3-gram: the <= 1 mismatch prefilter (fs_scan.hip)
  bool near8 = false;        // ... in the form k_scan_near8 reads (fs_scan_near_bit)
  DBuf<uint32_t> d_wild;     // one-slot-wildcard keys of the script windows (fs_hash.h; k_lsh_verify)
  int log2_wild = 0;
  DBuf<uint32_t> d_wmap;     // the same keys as an exact map {key, first window of the n-gram + 1}
  int log2_wmap = 0;
  // Tables with near-synonyms (the proof fails by more than one slot): the same two filters
  // over *component ids* -- connected components of "near" pairs of table vectors, at most one
  // slot of a neighbour within the threshold joins two components (fs_lsh.hip)
  DBuf<uint32_t> d_comp;     // [V] component id of a table vector
  DBuf<uint32_t> d_compa;    // [V] component id of a table vector under the angular relation of the share rule
  DBuf<uint64_t> d_ssig;     // [W] the script windows' component signatures (fs_share_sig)
  DBuf<uint32_t> d_sharef;   // the script windows' subset keys (fs_hash.h), a blocked Bloom filter
  int log2_sharef = 0;
  DBuf<uint32_t> d_smap;     // ... as an exact map (k_share_scan): 2^log2_smap buckets of four {key, list}
  DBuf<uint32_t> d_slists;   // a key's script windows behind their number, four words each: {window, signature word, 0} (the map names the first of them)
  int log2_smap = 0;
  DBuf<uint32_t> d_oovmap;   // the script's out-of-vocabulary vectors for share_comp: 2^log2_oovmap {key, component + 1}
  int log2_oovmap = 0;
  DBuf<uint32_t> d_share_cnt; // diagnostics (FS_SHARE_COUNT=1): eight 64-bit counters of k_share_scan
  int share_flags = 0;       // 0: the share rule is not in use; else sw.lsh_share's bits (bit 3 also set when the table does not prove out-of-vocabulary tokens far)
  double share_gamma = 0.0;
  uint32_t share_comps = 0, share_largest = 0;
  DBuf<uint32_t> d_sfilter3c, d_wildc, d_keys6c;   // (d_keys6c: n = 6, the wildcard keys of slots 2 and 3)
  int log2_wildc = 0;
  bool syn_ok = false;
  uint32_t n_comp = 0, comp_largest = 0;
  std::vector<uint32_t> comp_sizes;   // members per component of the near-pair graph (fs_index_component_sizes)

  // Lanes: a stream with its own workspaces (grown on demand) and status block.
  // Searches are spread over n_lanes of them (FS_LANES in the environment, default
  // one); with more than one, the latency-bound verify / rows chain of a search runs
  // beside the scan of the next.  `cur` is the lane the launchers enqueue into
  // (host-side, set by search_enqueue); `stream` is lane 0's stream, used by
  // everything that is not a search.
  struct Lane {
    hipStream_t stream = nullptr;
    DBuf<uint64_t> w_qbm, w_bsum64, w_hv, w_gate;   // (w_gate: the share rule's gate bits, k_share_gate)
    DBuf<uint32_t> w_qcnt, w_cpos, w_cg, w_cw, w_mlev, w_bsum, w_pend;
    DBuf<uint32_t> w_mcnt, w_mtop_s;   // k_lsh_verify -> k_lsh_lev: kept matches per pending window
    DBuf<uint32_t> w_pkeys, w_pwork, w_left;   // k_lsh_pkeys -> k_lsh_enum: keys and work per pending window; what is left to k_lsh_batch
    DBuf<double> w_mtop_d;
    DBuf<uint2> w_recs, w_info;    // direct path: candidate records and counts per wave range
    DBuf<uint32_t> w_slist, w_scount;   // k_near_sift: survivors of the wildcard filter per wave range (caps each), their counts
    uint32_t caps_hint = 0;        //   entries per wave range the last searches needed
    uint32_t capw_hint = 0;        // records per wave range that the last searches needed
    DBuf<uint8_t> w_stage;         // k_scan_rows: staged records, caprow per wave range
    uint32_t caprow_hint = 0;      //   staged records per wave range the last searches needed
    DBuf<uint8_t> w_xstage;        // k_scan_rows, shared rounds: a pool of records per workgroup
    uint32_t xpool = 0, xpool_hint = 0;   //   its size in the last launch / what the last searches needed
    DBuf<unsigned long long> w_gran;   // finish_rows: {epoch, records} per workgroup, then four statistics granules each
    DBuf<uint4> w_rinfo, w_csum;   // k_compact: {records, hits, pairs, candidates} per range / per workgroup
    uint32_t sync_epoch = 0;
    uint32_t pend_hint = 0xFFFFFFFFu;   // LSH pipeline: windows the lane's last search left to k_lsh_verify (none yet: many)
    uint64_t enqueued = 0;         // searches put on this lane so far
    DBuf<fs_best> w_cbest;
    DBuf<double> w_hcomb;          // per candidate: combined distance of its best rank (+inf: no hit)
    DBuf<fs_row> w_rows;
    DBuf<fs_status> d_status;
    DBuf<unsigned long long> w_dbg;    // FS_DIAG & 2: timeline stamps of the last k_scan_rows launch
    size_t dbg_words = 0;
  };
  Lane lanes[FS_LANES];
  Lane* cur = &lanes[0];
  int n_lanes = FS_LANES_DEFAULT;
  fs_status* h_status = nullptr;   // pinned

  // searches in flight (fs_search_corpus_begin / _end)
  struct Slot {
    hipEvent_t ev_begin = nullptr, ev_scan0 = nullptr, ev_scan1 = nullptr, ev_end = nullptr;
    fs_status* h_status = nullptr;    // pinned
    bool busy = false;
    uint64_t lane_seq = 0;            // Lane::enqueued when this search went in
    fs_corpus* c = nullptr;
    fs_row* rows = nullptr;
    uint64_t cap = 0, ccap = 0, rcap = 0;
    int mode = 0;
    bool header = false;              // FS_ROWS_HEADER: rows = 32-byte header + records
    bool host_wire8 = false;          // FS_ROWS_HOST: 8-byte records cross PCIe, fs_row made on the host
    bool host_direct = false;         //   ... stored into pinned host memory by the search's last kernel
    bool exact = false;
    uint32_t n_bm = 0, launches = 0, fallbacks = 0;
    bool timed = false;               // this search's scan carries timing events
    bool whole_timed = false;         // ... and ev_begin / ev_end bracket the whole search
    int tpl = 4;                      // tokens per lane of the bitmap layout
    int lane = 0;                     // the lane (stream + workspaces) it was queued on
    uint32_t capw = 0;                // direct path: record capacity per wave range (0: bitmap path)
    uint32_t caps = 0;                // k_near_sift: list entries per wave range (0: the chained prefilter)
    uint32_t caprow = 0;              // k_scan_rows: staged records per wave range
    uint32_t fused_waves = 0;         // k_scan_rows: waves per workgroup (0: separate kernels)
    uint32_t fused_blocks = 0;        //              workgroups
  };
  Slot slots[FS_SEARCH_SLOTS];
  bool sync_call = false;              // inside fs_search_corpus (begin + end in one call)
  uint32_t next_slot = 0;
  uint32_t scan_timing_period = 1;    // attach timing events to every k-th scan
  uint64_t searches = 0;
  uint64_t wait_fallbacks = 0;        // searches repeated through the chained kernels (finish_rows gave up)
  // fs_search_profile: an event behind every kernel of one search (diagnostics; bench.py's
  // per-kernel shares come from here)
  struct Prof { bool on = false; std::vector<hipEvent_t> ev; std::vector<const char*> names; size_t used = 0; } prof;

  // corpora created on this index and still alive: fs_index_destroy detaches them, so that
  // a corpus destroyed after its index does not touch freed memory
  std::vector<fs_corpus*> corpora;

  GramIndexDev gram_dev() const;
  ~fs_index();
};

struct fs_corpus {
  fs_index* ix = nullptr;
  uint64_t n_tok = 0, n_works = 0, n_str = 0;
  uint64_t windows = 0;      // sum over works of max(0, len - n + 1)
  std::vector<uint64_t> h_work_off;   // host copy of the work offsets (host-row searches)
  bool has_oov = false;
  bool has_str = false;
  DBuf<uint32_t> d_tok, d_str, d_chars, d_levtab, d_blk_work, d_blk4, d_check;
  hipStream_t copy_stream = nullptr;   // uploads run here, beside the search stream
  hipEvent_t ev_ready = nullptr;
  uint32_t* h_check = nullptr;         // pinned: {max table id + 1, any OOV, max string id + 1}
  bool pending = false;                // an upload is queued and not yet waited for
  DBuf<uint64_t> d_work_off, d_coff;
  DBuf<fs_best> d_gbest;
  DBuf<uint4> d_strrec;                // [n_str] {length, classes of the first 15 code points} (k_strrec)
  bool strrec_ready = false;
  DBuf<uint32_t> d_ctab;               // batch table of k_scan_rows: ids + this batch's best records (k_ctab)
  bool levtab_ready = false;
  bool ctab_ready = false;
  bool ctab_str = false;               // ... built for a batch with string ids of its own (entries mark unknown table distances)
  DBuf<unsigned long long> d_gramtab_best;   // LSH pipeline: per script n-gram, what a window with its ids and
  DBuf<uint32_t> d_gramtab_cnt;        //   their strings gets (fs_best; kept matches + 1), k_lsh_gramtab
  bool gramtab_ready = false;
  DBuf<uint32_t> d_ctok;               // LSH pipeline, tables with near-synonyms: component id per token (+ the scan's pad)
  bool ctok_ready = false;
  DBuf<uint32_t> d_selflev;            // LSH pipeline, string id == vector id: Levenshtein of script window w
  bool selflev_ready = false;          // against the strings of its own ids (k_selflev), FS_NONE: not known
  CorpusDev dev() const;
  ~fs_corpus();
};

// ---- kernel launchers (fs_scan.hip / fs_post.hip / fs_build.hip) ----------
// What a scan launch may produce beyond the bitmap (eight-tokens-per-lane kernel only):
//   bsum/zero  the FS_CHUNKS chunk sums the next kernel needs and a cleared status block
//              -> counted: no counting kernel
//   recs/info  the direct path.  A chunk is scanned by four waves, each over a contiguous
//              quarter of its sub-tiles (a "wave range").  A lane that finds candidates
//              among its eight windows appends ONE record {position/8 << 8 | flag byte,
//              rank of its first candidate inside the wave range} to the wave range's
//              list (capacity capw records); info[range] = {records, candidates}.
//              -> direct: no bitmap in global memory, no expand kernel; block b of
//              k_verify_direct takes chunk b, wave q its q-th wave range.
struct fs_scan_extra {
  uint32_t* bsum = nullptr;
  fs_status* zero = nullptr;
  uint2* recs = nullptr;
  uint2* info = nullptr;
  uint32_t capw = 0;
  bool counted = false, direct = false;    // out
};
int fs_launch_scan(const fs_index* ix, const CorpusDev& c, uint64_t* qbm, uint32_t* qcnt,
                   uint32_t n_bm_words, hipStream_t s, hipEvent_t e0 = nullptr,
                   hipEvent_t e1 = nullptr, fs_scan_extra* extra = nullptr);
bool fs_scan_direct_ok(const fs_index* ix, uint64_t n_tok);   // the direct path applies
uint32_t fs_scan_pad_tokens();
int fs_scan_tpl(const fs_index* ix, uint64_t n_tok);   // tokens per lane (bitmap layout)

int fs_launch_post(fs_index* ix, fs_corpus* c, uint32_t n_sub, int tpl, uint32_t ccap,
                   uint32_t rcap, fs_row* d_rows, int wire, fs_status* host_st, hipStream_t s,
                   const fs_scan_extra& scan, uint64_t* count_out = nullptr);
int fs_launch_unpack(fs_index* ix, const void* packed, uint64_t n, fs_row* rows, hipStream_t s);
int fs_launch_expand(fs_index* ix, fs_corpus* c, uint32_t n_sub, uint32_t ccap, int tpl,
                     hipStream_t s, bool counted = false);
int fs_launch_rows(fs_index* ix, fs_corpus* c, const fs_best* best_tab, int best_per_cand,
                   uint32_t ccap, uint32_t rcap, fs_row* d_rows, int wire, fs_status* host_st,
                   hipStream_t s, uint64_t* count_out = nullptr);
int fs_launch_unpack8(fs_index* ix, const void* packed, uint64_t n, const uint64_t* work_off,
                      uint64_t n_works, fs_row* rows, hipStream_t s);
int fs_lsh_build(fs_index* ix);
int fs_launch_lsh_scan(fs_index* ix, const CorpusDev& c, uint64_t* qbm, uint32_t* qcnt,
                       uint32_t n_sub, hipStream_t s, hipEvent_t e0 = nullptr,
                       hipEvent_t e1 = nullptr);
int fs_launch_selflev(fs_index* ix, fs_corpus* c, hipStream_t s);
int fs_launch_lsh_gramtab(fs_index* ix, fs_corpus* c, hipStream_t s);
int fs_launch_near_pairs(const float* emb, uint64_t n_vec, int D, const uint32_t* rows_u, uint32_t n_u,
                         const double* q, float* embT_scratch, float coef, float gamma, uint2* pairs, uint32_t cap,
                         uint32_t* count, hipStream_t s);      // gamma > -1.5: the angular relation cos > gamma
int fs_launch_coordmax(const float* emb, int D, const uint32_t* rows_u, uint32_t n_u, const double* q,
                       int* d_out_bits, hipStream_t s);
int fs_launch_comp_map(fs_index* ix, fs_corpus* c, hipStream_t s);      // component ids of a batch's tokens
int fs_lsh_prefilter_mode(const fs_index* ix, const fs_corpus* c);
int fs_scan_near_k(int n);                                               // K of k_scan_near's K-gram tests      // 0 none, 1 vector ids, 2 component ids
struct fs_near_lists { const uint32_t* slist; const uint32_t* scount; uint32_t caps; };
int fs_launch_lsh_verify(fs_index* ix, fs_corpus* c, uint32_t ccap, hipStream_t s, const fs_near_lists* near = nullptr);
int fs_prof_mark(fs_index* ix, hipStream_t s, const char* name);      // fs_api.hip
void fs_lsh_wild_of(const fs_index* ix, const fs_corpus* c, const uint32_t** wild, int* log2_wild,
                    const uint32_t** wild_tok);                        // fs_lsh.hip
int fs_launch_stream_floor(fs_index* ix, fs_corpus* c, uint32_t reps, double* avg_ms);   // fs_scan.hip
bool fs_near_fused(const fs_index* ix, const fs_corpus* c);            // fs_scan.hip: k_near_sift takes this search's prefilter
uint32_t fs_near_ranges();                                              // wave ranges of k_near_sift
int fs_launch_near_sift(const fs_index* ix, const fs_corpus* c, uint32_t* slist, uint32_t caps, uint32_t* scount,
                        uint32_t* bsum, fs_status* zero, hipStream_t s, hipEvent_t e0, hipEvent_t e1);
// fs_scan.hip: the integer prefilter of the LSH pipeline ("all but one slot identical")
bool fs_lsh_prefilter_ok(const fs_index* ix, const fs_corpus* c);
bool fs_scan_near8_wanted(const fs_index* ix);
bool fs_scan_near8(const fs_index* ix);
int fs_scan_near_log2(const fs_index* ix);
void fs_scan_near_bit(const fs_index* ix, const uint32_t* t, uint32_t* word, uint32_t* bit);
int fs_launch_scan_near(const fs_index* ix, const fs_corpus* c, uint64_t* qbm, uint32_t* qcnt,
                        uint32_t n_bm_words, hipStream_t s, hipEvent_t e0, hipEvent_t e1,
                        fs_scan_extra* ex);
int fs_launch_corpus_check(const uint32_t* tok, const uint32_t* str, uint32_t n_tok,
                           uint32_t* check, hipStream_t s);
int fs_launch_histogram(const uint32_t* d_orig, const double* d_comb, const fs_row* d_rows,
                        uint64_t n_rows, uint64_t n_script, const double* d_thr, uint32_t n_thr,
                        uint32_t* d_counts, hipStream_t s);
int fs_launch_blk_work(const uint64_t* work_off, uint32_t n_works, uint32_t n_blocks,
                       uint2* blk_work, uint4* blk4, hipStream_t s);

void fs_host_pool_free(struct fs_host_pool* p);
int fs_launch_levtab(fs_index* ix, fs_corpus* c, hipStream_t s);
int fs_launch_strrec(fs_index* ix, fs_corpus* c, hipStream_t s);

// fs_ranges.hip: helpers of the records path of k_scan_rows
int fs_launch_ctab(fs_index* ix, fs_corpus* c, hipStream_t s);
// fs_scan.hip: scan + records in one kernel (k_scan_rows)
namespace fsdev { struct RowSync; }
int fs_row_sync(fs_index* ix, uint32_t n_blocks, uint64_t n_tok, fsdev::RowSync* sy);   // fs_ranges.hip
int fs_launch_compact_after_scan_rows(fs_index* ix, uint32_t n_ranges, uint32_t waves, uint32_t caprow,
                                      int wire, uint32_t rcap, fs_row* d_rows,
                                      fs_status* host_st, hipStream_t s, uint64_t* count_out,
                                      hipEvent_t done);
uint32_t fs_scan_rows_shape(const fs_index* ix, const fs_corpus* c, uint32_t* blocks);   // waves per workgroup, 0: does not apply
int fs_launch_scan_rows(fs_index* ix, fs_corpus* c, uint32_t waves, uint32_t blocks, uint32_t rcap, fs_row* d_rows,
                        int wire, uint32_t caprow, fs_status* host_st, hipStream_t s,
                        hipEvent_t e0, hipEvent_t e1, uint64_t* count_out, hipEvent_t done = nullptr,
                        bool* done_attached = nullptr);

int fs_launch_rownorms(const float* emb, uint64_t n_vec, int D, double* q, hipStream_t s);
int fs_launch_selfdist(const uint32_t* stok, uint64_t n_windows, int n, int D,
                       uint64_t n_vec, const double* q, double* selfdist, hipStream_t s);
int fs_launch_cmax(const float* emb, uint64_t n_vec, int D, const uint32_t* rows_u,
                   uint32_t n_u, const double* q, float* embT_scratch, int* d_out_bits,
                   hipStream_t s);
