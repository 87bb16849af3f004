/* fandom_search.h -- C ABI of the MI355X 6-gram text-reuse search library
 * (libfandomsearch_hip.so).
 *
 * The reference (senderle/fandom-search) is pure Python and has no FFI; its
 * seams for this path are Python call sites.  Each entry point below names the
 * reference interface it stands behind (file:line in the reference tree):
 *
 *   fs_index_create      AnnIndexSearch.__init__ + build_lsh_engine
 *                        (search.py:131-154, 86-124): script tokens -> LSH index
 *   fs_corpus_create     the per-work read + tokenise + mk_vectors step of
 *                        AnnIndexSearch.search (search.py:164-173), batched:
 *                        packed vector-id / string-id buffers for many works
 *   fs_search_corpus     AnnIndexSearch.search (search.py:163-226) over every
 *                        work of a corpus, i.e. one pool.map batch
 *                        (search.py:381-386) in a single call
 *   fs_search            convenience: fs_corpus_create + fs_search_corpus +
 *                        fs_corpus_destroy on host buffers
 *   fs_stats             AnnIndexSearch.windows_processed / reset_stats
 *                        (search.py:156-161) plus kernel timings
 *
 * Conventions: every function returns FS_OK (0) or a negative FS_E_* code and
 * never throws across the boundary; the caller owns every buffer it passes,
 * the library owns only the opaque handles; a handle is bound to one device
 * and is not thread-safe (use one handle per device / per rank).  Rows come
 * back sorted by (work, fan_ix) -- the order of `sorted(values)` at
 * search.py:226 followed by the in-order concatenation of search.py:386.
 *
 * Data model
 *   vector id   uint32.  id < n_vec: row of the embedding matrix `emb`.
 *               id & FS_OOV_FLAG: out-of-vocabulary 3-hot vector of
 *               search.py:79-83, low 31 bits = (a*D + b)*D + c, a <= b <= c.
 *               Two tokens have equal vector ids iff they have equal vectors.
 *   string      UTF-32 code points in a string table: chars[off[i]..off[i+1])
 *   work        tokens [work_off[w], work_off[w+1]) of the packed buffers
 */
#ifndef FANDOM_SEARCH_H
#define FANDOM_SEARCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FS_ABI_VERSION 1
#define FS_OOV_FLAG 0x80000000u

enum {
  FS_OK = 0,
  FS_E_INVALID = -1,      /* bad argument / inconsistent sizes                */
  FS_E_NOMEM = -2,        /* host or device allocation failed                 */
  FS_E_DEVICE = -3,       /* HIP runtime error (fs_last_error has the text)   */
  FS_E_CAPACITY = -4,     /* rows buffer too small; *n_rows = rows required   */
  FS_E_UNSUPPORTED = -5,  /* parameter outside what the kernels are built for */
  FS_E_UNPROVEN = -6      /* mode FS_MODE_EXACT but the prefilter proof fails */
};

/* which device pipeline produced (or will produce) the rows */
enum {
  FS_MODE_AUTO = 0,     /* exact n-gram scan when the index proves it sound,
                           otherwise the general LSH pipeline                 */
  FS_MODE_GENERAL = 1,  /* always the LSH pipeline (A6/A12 of SURVEY 8(a))    */
  FS_MODE_EXACT = 2     /* exact scan or FS_E_UNPROVEN                        */
};

typedef struct fs_config {
  uint32_t struct_size;        /* sizeof(fs_config), for ABI growth           */
  uint32_t window_size;        /* n            search.py:337  default 6       */
  uint32_t number_of_hashes;   /* H            search.py:338  default 15      */
  uint32_t hash_dimensions;    /* B <= 24      search.py:339  default 14      */
  uint32_t emb_dim;            /* D            300 for en_core_web_md         */
  uint32_t nearest_n;          /* NearPy NearestFilter(N) default 10          */
  uint32_t unique_filter;      /* NearPy UniqueFilter on fetch: 0 = NearPy 1.0.0 for the reference's call (host default), 1 = NearPy 0.2.x */
  uint32_t mode;               /* FS_MODE_*                                   */
  int32_t  device;             /* HIP device ordinal                          */
  uint32_t reserved;
  double   distance_threshold; /*              search.py:340  default 0.1     */
} fs_config;

/* One output record = the numeric half of a row of new_record_structure
 * (search.py:20-37); the host joins file name, words, orth ids, character and
 * scene by (work, fan_ix, orig_ix). */
typedef struct fs_row {
  uint32_t work;     /* index into work_off                                   */
  uint32_t fan_ix;   /* FAN_WORK_WORD_INDEX                                   */
  uint32_t orig_ix;  /* ORIGINAL_SCRIPT_WORD_INDEX                            */
  uint32_t lev;      /* BEST_LEVENSHTEIN_DISTANCE                             */
  double   dist;     /* BEST_MATCH_DISTANCE                                   */
  double   comb;     /* BEST_COMBINED_DISTANCE = dist * lev                   */
} fs_row;            /* 32 bytes                                              */

typedef struct fs_stats {
  uint64_t windows_processed;  /* search.py:177 counter, summed over works     */
  uint64_t candidates;         /* scan: filter positives; LSH: bucket entries */
  uint64_t matches;            /* (window, script window) pairs kept          */
  uint64_t rows;               /* records after per-word dedupe               */
  double   scan_ms;            /* dominant kernel, HIP-event time; 0 if untimed */
  double   total_ms;           /* all device work of the call, HIP events; fs_search_corpus
                                  and fs_search only (0 for _begin/_end pairs and on
                                  searches without timing, fs_index_set_scan_timing) */
  uint32_t path;               /* FS_MODE_GENERAL or FS_MODE_EXACT            */
  uint32_t scan_launches;      /* launches of the dominant kernel in the call */
  uint32_t lsh_pending;        /* general pipeline: candidate windows that took the full LSH
                                  path (keys, buckets, distances), a wave each; the others
                                  ended in the lane-per-candidate steps                  */
  uint32_t handoff_fallbacks;  /* times this call ran its search again through the chained
                                  kernels because the in-launch hand-off of k_scan_rows gave
                                  up (a workgroup waited longer than a few scan times for
                                  the ones in front: co-residency lost); 0 in normal running */
} fs_stats;

typedef struct fs_index_info {
  uint32_t path;            /* pipeline fs_search_corpus will run             */
  uint32_t proof_ok;        /* exact-n-gram prefilter proven sound            */
  double   c_max;           /* max cosine between distinct vectors            */
  double   cos_bound;       /* upper bound on cos(F,S) with >= 1 mismatch     */
  double   norm_min, norm_max;
  uint64_t n_script;        /* script tokens                                  */
  uint64_t n_windows;       /* script windows                                 */
  uint64_t n_grams;         /* distinct script n-grams (by vector id)         */
  uint64_t filter_bytes;    /* size of the LDS-resident n-gram filter         */
} fs_index_info;

typedef struct fs_index fs_index;
typedef struct fs_corpus fs_corpus;

int fs_version(void);
const char* fs_strerror(int code);
/* text of the last FS_E_DEVICE / FS_E_INVALID on this thread */
const char* fs_last_error(void);

/* Build the script index on cfg->device.
 *   script_vec[n_script]        vector id of each (lower-cased) script word
 *   script_chars/script_off     text of each script word: word i is
 *                               script_chars[script_off[i]..script_off[i+1])
 *   emb[n_vec][D] float32       vector table (spaCy vectors.data)
 *   normals[H][B][D*n] float64  LSH hyperplanes (NearPy draws them unseeded,
 *                               search.py:114-115; here they are an input)   */
int fs_index_create(const fs_config* cfg,
                    const uint32_t* script_vec,
                    const uint32_t* script_chars, const uint64_t* script_off,
                    uint64_t n_script,
                    const float* emb, uint64_t n_vec,
                    const double* normals,
                    fs_index** out);
int fs_index_info_get(const fs_index* ix, fs_index_info* info);
void fs_index_destroy(fs_index* ix);

/* Upload a batch of works (host buffers) to the index's device.
 *   tok_vec[n_tok]   vector id per fan token (what the kernels scan)
 *   tok_str[n_tok]   string id per fan token, or NULL when string id ==
 *                    vector id (synthetic corpora)
 *   str_chars/str_off/n_str   string table of the fan side
 *   work_off[n_works+1]       token offsets, work_off[0] == 0               */
int fs_corpus_create(fs_index* ix,
                     const uint32_t* tok_vec, const uint32_t* tok_str,
                     const uint64_t* work_off, uint64_t n_works,
                     const uint32_t* str_chars, const uint64_t* str_off,
                     uint64_t n_str,
                     fs_corpus** out);
/* A corpus may be destroyed before or after its index: fs_index_destroy detaches
 * the corpora that are still alive (a detached corpus can only be destroyed). */
void fs_corpus_destroy(fs_corpus* c);

/* Streaming (BASELINE configs[4]: corpora larger than one batch, streamed from
 * pinned host memory).  fs_corpus_update_begin replaces the works of `c` with a
 * new batch: the copies, the block->work table and a device-side validation of
 * the ids are queued on the corpus's own copy stream and the call returns at
 * once, so the upload overlaps a search that is running on another corpus of
 * the same index.  The host buffers must stay untouched until
 * fs_corpus_update_end (or the next fs_search_corpus on `c`) has returned.
 * The string table given at fs_corpus_create is kept.  Use fs_host_alloc for
 * the staging buffers: copies from pageable memory do not overlap. */
int fs_corpus_update_begin(fs_corpus* c, const uint32_t* tok_vec, const uint32_t* tok_str,
                           const uint64_t* work_off, uint64_t n_works);
int fs_corpus_update_end(fs_corpus* c);
int fs_host_alloc(uint64_t bytes, void** out);
void fs_host_free(void* p);

/* where fs_search_corpus leaves the records */
enum {
  FS_ROWS_HOST = 0,           /* `rows` is a host buffer of fs_row                        */
  FS_ROWS_DEVICE = 1,         /* `rows` is a device buffer of fs_row                      */
  FS_ROWS_DEVICE_PACKED = 2,  /* `rows` is a device buffer of 16-byte wire records
                                 {work, fan_ix, orig_ix, lev | k << 16} (exact n-gram
                                 pipeline only, else FS_E_UNSUPPORTED): half the bytes
                                 for the gather; fs_rows_unpack restores fs_row          */
  FS_ROWS_DEVICE_PACKED8 = 3, /* 8-byte wire records {token position in the batch,
                                 orig_ix | k << 18 | lev << 22}: a quarter of the bytes.
                                 Exact pipeline and scripts below 2^18 tokens only
                                 (else FS_E_UNSUPPORTED); fs_rows_unpack8 restores fs_row
                                 given the batch's work offsets                          */
  FS_ROWS_HEADER = 0x100      /* flag, with a device mode: `rows` points to a 32-byte
                                 header followed by the `cap` records; the search writes
                                 the record count (uint64) into the header's first eight
                                 bytes, so that count and records can travel in one
                                 collective without a host-side step in between          */
};

/* Search every work of `c`.  `rows` is a host buffer of `cap` records, or,
 * when rows_on_device != 0, a 16-byte aligned device pointer on the index's
 * device (for a collective gather without a host round trip).  On FS_E_CAPACITY *n_rows is
 * the number required.  `st` may be NULL. */
int fs_search_corpus(fs_index* ix, fs_corpus* c,
                     fs_row* rows, uint64_t cap, int rows_on_device,
                     uint64_t* n_rows, fs_stats* st);

/* The same in two halves: _begin queues the search and returns a ticket, _end waits
 * for it and delivers what fs_search_corpus delivers.  Up to four searches may be in
 * flight per index, so the host can queue the next batch while the GPU still works
 * on the previous ones.  With FS_LANES=2..4 in the environment consecutive searches
 * go to alternating streams of the index and overlap on the GPU (the verify / rows
 * chain of one beside the scan of the next; searches are independent, every one
 * writes only its own rows buffer); by default they run in order.  Device row modes
 * only while another search is in flight; the rows buffers of searches in flight
 * must be distinct; a corpus must not be updated while a search on it is in
 * flight. */
int fs_search_corpus_begin(fs_index* ix, fs_corpus* c, fs_row* rows, uint64_t cap,
                           int rows_mode, uint32_t* ticket);
int fs_search_corpus_end(fs_index* ix, uint32_t ticket, uint64_t* n_rows, fs_stats* st);

/* Expand n packed wire records (device) into fs_row records (device) on an index
 * built from the same script, vectors and config: dist = the matched script
 * window's self distance, comb = dist * lev. */
int fs_rows_unpack(fs_index* ix, const void* packed, uint64_t n, fs_row* rows);

/* The same for n 8-byte wire records: work_off (device, n_works + 1 offsets of the
 * batch the records come from) turns a token position back into (work, fan_ix). */
int fs_rows_unpack8(fs_index* ix, const void* packed, uint64_t n, const uint64_t* work_off,
                    uint64_t n_works, fs_row* rows);

/* fs_corpus_create + fs_search_corpus + fs_corpus_destroy. */
int fs_search(fs_index* ix,
              const uint32_t* tok_vec, const uint32_t* tok_str,
              const uint64_t* work_off, uint64_t n_works,
              const uint32_t* str_chars, const uint64_t* str_off,
              uint64_t n_str,
              fs_row* rows, uint64_t cap, uint64_t* n_rows, fs_stats* st);

/* `ao3.py format` aggregation (ao3.py:351-363, 407-416): for every script word the
 * number of match records whose BEST_COMBINED_DISTANCE is <= each of `n_thr`
 * ascending thresholds (the reference uses 0, 0.05, ... 0.5), plus, in column
 * n_thr, the number of records of that word at all.
 *   orig_ix[n_rows], comb[n_rows]   host arrays (two columns of the match CSV)
 *   counts[n_script][n_thr + 1]     host, uint32
 * NaN never satisfies <=, as in pandas.  Runs on HIP device `device`. */
int fs_reuse_histogram(int device, const uint32_t* orig_ix, const double* comb, uint64_t n_rows,
                       uint64_t n_script, const double* thresholds, uint32_t n_thr,
                       uint32_t* counts);
/* The same over device-resident fs_row records (e.g. straight after a search or a
 * gather); d_counts is a device buffer of n_script * (n_thr + 1) uint32. */
int fs_reuse_histogram_rows(fs_index* ix, const fs_row* d_rows, uint64_t n_rows,
                            const double* thresholds, uint32_t n_thr, uint32_t* d_counts);

/* Timing events ride on every `period`-th scan launch only (default 1 = every
 * launch); searches in between report scan_ms = 0.  The events cost a few
 * microseconds of stream time per launch, which matters for sub-100 us searches. */
int fs_index_set_scan_timing(fs_index* ix, uint32_t period);

/* Diagnostics: name of the kernel that dominates a search of `c` on `ix` as things stand
 * (pipeline, window size, corpus size, switches), e.g. "k_scan_rows<6,4>": what a profile
 * of the search lists first.  Static storage, valid until the next call on this thread. */
const char* fs_search_kernel_name(fs_index* ix, fs_corpus* c);

/* ---- host text front end (search.py:164-166: read a fan work, tokenise, drop whitespace) ----
 * spaCy's tokenizer splits a text on whitespace and treats every chunk by itself, with a cache
 * chunk -> tokens; fs_textenc is that cache and the splitting, natively and on `threads` host
 * threads: files in, string ids out.  The host teaches it what a chunk's tokens are
 * (fs_textenc_add: the vocabulary's plain words to start with, then every chunk its rule
 * tokenizer has been run on); a chunk it has not been taught comes back as a placeholder
 * 0x80000000 | k with its text (unk_bytes[unk_off[k] .. unk_off[k+1])), never as a guess.
 * status[i]: 0 encoded; 1 left to the host (a text of 100000 bytes or more -- the reference
 * cuts such texts into pieces first, search.py:47-63 -- or malformed UTF-8); < 0: -errno of
 * the read.  The result pointers are the encoder's own buffers, valid until the next call on
 * it.  No GPU is involved. */
typedef struct fs_textenc fs_textenc;
int fs_textenc_create(fs_textenc** out);
void fs_textenc_destroy(fs_textenc* enc);
int fs_textenc_add(fs_textenc* enc, const uint8_t* chunk_bytes, const uint64_t* chunk_off, uint64_t n_chunks,
                   const uint64_t* piece_off, const uint32_t* piece_ids);
int fs_textenc_encode_files(fs_textenc* enc, const char* paths /* n_files strings, each 0-terminated */,
                            uint64_t n_files, uint32_t threads,
                            const uint32_t** tok, uint64_t* n_tok, const uint64_t** work_off /* n_files + 1 */,
                            const int32_t** status, const uint8_t** unk_bytes, const uint64_t** unk_off,
                            uint64_t* n_unk);
/* The same with the pass the search makes over a batch's tokens next (search.py:65-84 looks
 * every token's vector up; here a token's vector id is vec_of_sid[string id], n_sid entries):
 * tok_vec[i] = vector id of token i (0 for a placeholder), *n_oov = tokens whose vector id has
 * FS_OOV_FLAG set, *ids_equal = 1 when every token's vector id equals its string id (a batch
 * without capitalised or out-of-vocabulary words: no string ids need to travel to the GPU).
 * Made by the threads that encode, each for its own files.  FS_E_INVALID when a string id the
 * encoder was taught lies beyond n_sid. */
int fs_textenc_encode_files_vec(fs_textenc* enc, const char* paths, uint64_t n_files, uint32_t threads,
                                const uint32_t* vec_of_sid, uint64_t n_sid,
                                const uint32_t** tok, uint64_t* n_tok, const uint64_t** work_off,
                                const int32_t** status, const uint8_t** unk_bytes, const uint64_t** unk_off,
                                uint64_t* n_unk, const uint32_t** tok_vec, uint64_t* n_oov, int32_t* ids_equal);

/* Diagnostics: tables with near-synonyms -- the sizes of the connected components of the graph
 * of "near" vector pairs that the integer prefilters of the LSH pipeline work over (0 entries:
 * the graph was not built: the exact pipeline, or a proof that fails by one slot only).
 * *in_use = 1 when the prefilters run over component ids, 0 when the components were judged
 * too coarse (one holds an eighth of the table) and searches take the plain LSH pipeline.  On an
 * index that uses the share rule (fs_index_share_info) the components are those of its angular
 * relation, and *in_use = 1. */
int fs_index_component_sizes(const fs_index* ix, uint32_t* sizes, uint64_t cap, uint64_t* n,
                             uint32_t* in_use);

/* Diagnostics: the share rule of the LSH pipeline (tables whose vectors are not unit length, where
 * neither integer prefilter applies).  *flags = 0: not in use; else bit 0 the
 * windows' gate, bit 1 the pairs' test, bit 2 the gate over heavy subsets of both sides, bit 3
 * out-of-vocabulary fan tokens count as possibly near, bit 4 set.  *components / *largest: the
 * components of the relation "cosine > *gamma" over the table. */
int fs_index_share_info(const fs_index* ix, uint32_t* flags, uint32_t* components, uint32_t* largest,
                        double* gamma);

/* Diagnostics: with FS_SHARE_COUNT=1 in the environment when the index is built, k_share_scan counts
 * what passes what; this reads the eight counters and sets them to zero: windows, windows with a
 * key in the filter, (window, key) entries of the map, (window, script window) pairs through the
 * pairs' test, pairs whose distance was computed, windows flagged, windows flagged as they are (no
 * constraint, or no room), 0.  All zero when the counters are off or the rule is not in use. */
int fs_index_share_counts(fs_index* ix, uint64_t* out8);

/* Diagnostics: one synchronous search of `c` (arguments as fs_search_corpus) with a HIP event
 * behind every kernel of it.  names: the kernels' names in launch order, '\n'-separated;
 * ms[i]: time from the previous mark to the one behind kernel i (its duration when nothing else
 * runs on the GPU); *n = number of entries (also when the buffers hold fewer).  What bench.py
 * names as the dominant kernel of a search that is more than one launch comes from here.
 * Not part of the search path. */
int fs_search_profile(fs_index* ix, fs_corpus* c, fs_row* rows, uint64_t cap, int rows_on_device,
                      char* names, uint64_t names_cap, double* ms, uint32_t ms_cap, uint32_t* n);

/* Diagnostics: the FS_* environment switches (kernel variants, forced capacities) are
 * read once at fs_index_create; a test or sweep that changes them on a live index
 * calls this to have them read again.  Not part of the search path. */
int fs_index_reload_switches(fs_index* ix);

/* Diagnostics: `reps` back-to-back launches of the scan kernel alone over `c`,
 * timed with one pair of HIP events; *avg_ms = time per launch.  Used by
 * tools/scan_sweep.py to compare kernel variants without per-launch event
 * overhead.  Not part of the search path. */
int fs_scan_benchmark(fs_index* ix, fs_corpus* c, uint32_t reps, double* avg_ms);

/* Diagnostics: the floor under k_scan_rows.  `reps` launches of a kernel that only READS the ids
 * of `c` in k_scan_rows' launch shape (a workgroup of sixteen waves per CU, contiguous runs of
 * 512-token sub-tiles, 16-byte loads, pairs requested ahead) and keeps nothing; *avg_ms = its
 * dispatch-to-completion time, HIP events on every dispatch, one launch at a time.  What a
 * search's own kernel takes above this is its own work. */
int fs_stream_floor(fs_index* ix, fs_corpus* c, uint32_t reps, double* avg_ms);

/* ---- the batch files (search.py:192-218 the twelve fields of a record, :331-334 write_records) ----
 * fs_row records in, the bytes csv.writer(out).writerows(records) puts into a batch file out:
 * FAN_WORK_FILENAME = name of rows[i].work, FAN_WORK_WORD / _ORTH_ID = text and spaCy key
 * (MurmurHash64A of the UTF-8 bytes, seed 1) of string fan_sid[i] of the strings added so far,
 * the ORIGINAL_SCRIPT_* columns = entry rows[i].orig_ix of the four tables of fs_csvw_set_script
 * (each the text a record shows: the lower-case word, its key in decimal, the character name,
 * the scene number; None = the empty string), distances by Python's repr(float).  `excel`
 * dialect: "\r\n" behind a record, a field quoted -- its quotes doubled -- when it holds ',',
 * '"', CR or LF.  A string table is {bytes, off[n + 1]}.  *out is the writer's own buffer, valid
 * until its next call; writing it to a file is the caller's.  No GPU is involved. */
typedef struct fs_csvw fs_csvw;
int fs_csvw_create(fs_csvw** out);
void fs_csvw_destroy(fs_csvw* w);
int fs_csvw_set_script(fs_csvw* w, uint64_t n_script,
                       const uint8_t* word_bytes, const uint64_t* word_off,
                       const uint8_t* orth_bytes, const uint64_t* orth_off,
                       const uint8_t* char_bytes, const uint64_t* char_off,
                       const uint8_t* scene_bytes, const uint64_t* scene_off);
int fs_csvw_add_strings(fs_csvw* w, const uint8_t* bytes, const uint64_t* off, uint64_t n);   /* ids go on counting */
uint64_t fs_csvw_strings(const fs_csvw* w);                                                  /* strings added so far */
int fs_csvw_format(fs_csvw* w, const fs_row* rows, uint64_t n_rows,
                   const uint8_t* name_bytes, const uint64_t* name_off, uint64_t n_works,
                   const uint32_t* fan_sid, const uint8_t** out, uint64_t* out_len);

/* Diagnostics (FS_DIAG=2 in the environment at fs_index_create): per wave range of the
 * last k_scan_rows launch on stream `lane`, eight uint64 {entry, filter staged, scan done,
 * rounds done, finished (ticks of the 100 MHz constant clock), rounds, flushes, records}.
 * *n = words available; FS_E_CAPACITY when cap is smaller.  tools/scan_timeline.py. */
int fs_debug_stamps(fs_index* ix, uint32_t lane, uint64_t* out, uint64_t cap, uint64_t* n);

#ifdef __cplusplus
}
#endif
#endif /* FANDOM_SEARCH_H */
