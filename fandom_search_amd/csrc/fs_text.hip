// fs_text.hip -- the host text front end of the search command: read fan works, split them
// into whitespace-delimited chunks and turn every chunk into string ids, on host threads.
//
// Replaces, for the bulk of a corpus, what the reference does per work in Python
// (/root/reference/search.py:164-166: read the file, sp_parse_chunks -> spaCy's tokenizer, drop
// the whitespace tokens) and what fandom_search_amd/tokenizer.py restates: spaCy's tokenizer
// first splits on whitespace and then treats every chunk by itself (prefixes, suffixes,
// infixes, special cases), and keeps a cache chunk -> tokens.  That cache is what lives here,
// natively: a table from a chunk's bytes to the string ids of its tokens, filled by the host
// (Python) side -- the vocabulary's plain words to start with, every chunk the rules have been
// run on since.  A chunk the table does not hold is NOT guessed at: it comes back as a
// placeholder with its text, the Python rules (tokenizer.py, the oracle of this file) are run
// on it once, and the table learns it.  So the token stream is the rule tokenizer's by
// construction; this file only does the splitting, the hashing and the file I/O, at memory
// speed on all cores instead of 3 M tokens/s per Python process.
//
// Host code only (no kernel, no HIP call); it lives in the library because the C-ABI is the
// drop-in boundary (include/fandom_search.h: fs_textenc_*).
#include "../../include/fandom_search.h"

#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

namespace {

inline uint64_t hash_bytes(const uint8_t* p, size_t n) {
  // 64-bit FNV-1a over 8-byte words with a multiply-fold finish: only selects a table slot,
  // every hit is compared byte for byte
  uint64_t h = 0xcbf29ce484222325ull ^ (n * 0x9E3779B97F4A7C15ull);
  while (n >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    h = (h ^ w) * 0x100000001b3ull;
    h ^= h >> 29;
    p += 8; n -= 8;
  }
  uint64_t w = 0;
  memcpy(&w, p, n);
  h = (h ^ w) * 0x100000001b3ull;
  h ^= h >> 32;
  h *= 0x9E3779B97F4A7C15ull;
  h ^= h >> 29;
  return h;
}

// The same value for bytes that have eight readable bytes behind them (a file's buffer is padded):
// the tail as one masked load instead of a memcpy of variable length.
inline uint64_t hash_bytes_padded(const uint8_t* p, size_t n) {
  uint64_t h = 0xcbf29ce484222325ull ^ (n * 0x9E3779B97F4A7C15ull);
  while (n >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    h = (h ^ w) * 0x100000001b3ull;
    h ^= h >> 29;
    p += 8; n -= 8;
  }
  uint64_t w;
  memcpy(&w, p, 8);
  w &= n ? ((1ull << (8 * n)) - 1ull) : 0ull;
  h = (h ^ w) * 0x100000001b3ull;
  h ^= h >> 32;
  h *= 0x9E3779B97F4A7C15ull;
  h ^= h >> 29;
  return h;
}

struct Entry {
  uint64_t hash = 0;
  uint64_t off = 0;      // the chunk's bytes in `arena`; a chunk of up to eight bytes: the bytes themselves
  uint32_t len = 0;
  uint32_t n_pieces = 0; // 0: empty slot
  uint64_t first = 0;    // its string ids in `pieces`; a chunk of one token: the id itself
};
// (a word of up to eight bytes that is one token -- nearly every chunk of a text -- is decided by its
// 32-byte entry alone: no second and third cache miss for the arena's bytes and the id)

inline uint64_t first_bytes(const uint8_t* p, uint32_t n) {      // n <= 8; p has eight readable bytes
  uint64_t w;
  memcpy(&w, p, 8);
  return n >= 8 ? w : w & ((1ull << (8 * n)) - 1ull);
}

struct ThreadOut {
  std::vector<uint32_t> tok;
  std::vector<uint64_t> work_len;           // tokens per file of this thread's range
  std::vector<int32_t> status;
  std::vector<uint8_t> unk_bytes;           // distinct unknown chunks of this thread
  std::vector<uint64_t> unk_off{0};
  std::vector<Entry> unk_table;             // open addressing over this thread's unknown chunks (first = index)
  uint64_t unk_count = 0;
};

}  // namespace

struct fs_textenc {
  std::vector<Entry> table;                 // open addressing, power of two
  uint64_t used = 0;
  std::vector<uint8_t> arena;
  std::vector<uint32_t> pieces;
  // results of the last fs_textenc_encode_files
  std::vector<uint32_t> tok;
  std::vector<uint32_t> tok_vec;                // (fs_textenc_encode_files_vec)
  std::vector<uint64_t> work_off;
  std::vector<int32_t> status;
  std::vector<uint8_t> unk_bytes;
  std::vector<uint64_t> unk_off;
  std::string error;
};

namespace {

const Entry* find(const std::vector<Entry>& table, const std::vector<uint8_t>& arena, const uint8_t* p,
                  uint32_t n, uint64_t h) {
  if (table.empty()) return nullptr;
  const uint64_t mask = table.size() - 1;
  for (uint64_t i = h & mask;; i = (i + 1) & mask) {
    const Entry& e = table[i];
    if (!e.n_pieces) return nullptr;
    if (e.hash == h && e.len == n &&
        (n <= 8 ? e.off == first_bytes(p, n) : memcmp(arena.data() + e.off, p, n) == 0)) return &e;
  }
}

void grow(fs_textenc* enc) {
  std::vector<Entry> old;
  old.swap(enc->table);
  enc->table.assign(old.empty() ? (1u << 16) : old.size() * 2, Entry());
  const uint64_t mask = enc->table.size() - 1;
  for (const Entry& e : old)
    if (e.n_pieces) {
      uint64_t i = e.hash & mask;
      while (enc->table[i].n_pieces) i = (i + 1) & mask;
      enc->table[i] = e;
    }
}

// Python's str.split() separators (str.isspace()): length in bytes of the whitespace character
// at p, 0 if the character there is none.  *bad = true on malformed UTF-8.
inline int space_len(const uint8_t* p, const uint8_t* end) {
  const uint8_t c = p[0];
  if (c < 0x80) return (c == 0x20 || (c >= 0x09 && c <= 0x0d) || (c >= 0x1c && c <= 0x1f)) ? 1 : 0;
  if (c == 0xC2 && p + 1 < end) return (p[1] == 0x85 || p[1] == 0xA0) ? 2 : 0;
  if (c == 0xE1 && p + 2 < end) return (p[1] == 0x9A && p[2] == 0x80) ? 3 : 0;               // U+1680
  if (c == 0xE2 && p + 2 < end) {
    if (p[1] == 0x80) return ((p[2] >= 0x80 && p[2] <= 0x8A) || p[2] == 0xA8 || p[2] == 0xA9 || p[2] == 0xAF) ? 3 : 0;
    if (p[1] == 0x81 && p[2] == 0x9F) return 3;                                               // U+205F
    return 0;
  }
  if (c == 0xE3 && p + 2 < end) return (p[1] == 0x80 && p[2] == 0x80) ? 3 : 0;               // U+3000
  return 0;
}

// strict UTF-8 (what Python's codec accepts): length of the sequence at p, 0 if malformed
inline int utf8_len(const uint8_t* p, const uint8_t* end) {
  const uint8_t c = p[0];
  if (c < 0x80) return 1;
  if (c < 0xC2) return 0;
  if (c < 0xE0) return (p + 1 < end && (p[1] & 0xC0) == 0x80) ? 2 : 0;
  if (c < 0xF0) {
    if (p + 2 >= end || (p[1] & 0xC0) != 0x80 || (p[2] & 0xC0) != 0x80) return 0;
    if (c == 0xE0 && p[1] < 0xA0) return 0;
    if (c == 0xED && p[1] > 0x9F) return 0;                    // surrogates
    return 3;
  }
  if (c < 0xF5) {
    if (p + 3 >= end || (p[1] & 0xC0) != 0x80 || (p[2] & 0xC0) != 0x80 || (p[3] & 0xC0) != 0x80) return 0;
    if (c == 0xF0 && p[1] < 0x90) return 0;
    if (c == 0xF4 && p[1] > 0x8F) return 0;
    return 4;
  }
  return 0;
}

uint32_t unknown_index(ThreadOut& o, const uint8_t* p, uint32_t n, uint64_t h) {
  if (o.unk_table.empty()) o.unk_table.assign(1u << 10, Entry());
  if ((o.unk_count + 1) * 2 > o.unk_table.size()) {
    std::vector<Entry> old;
    old.swap(o.unk_table);
    o.unk_table.assign(old.size() * 2, Entry());
    const uint64_t mask = o.unk_table.size() - 1;
    for (const Entry& e : old)
      if (e.n_pieces) {
        uint64_t i = e.hash & mask;
        while (o.unk_table[i].n_pieces) i = (i + 1) & mask;
        o.unk_table[i] = e;
      }
  }
  const uint64_t mask = o.unk_table.size() - 1;
  for (uint64_t i = h & mask;; i = (i + 1) & mask) {
    Entry& e = o.unk_table[i];
    if (!e.n_pieces) {
      e.hash = h; e.len = n; e.n_pieces = 1; e.off = o.unk_bytes.size(); e.first = o.unk_count;
      o.unk_bytes.insert(o.unk_bytes.end(), p, p + n);
      o.unk_off.push_back(o.unk_bytes.size());
      return (uint32_t)o.unk_count++;
    }
    if (e.hash == h && e.len == n && memcmp(o.unk_bytes.data() + e.off, p, n) == 0) return (uint32_t)e.first;
  }
}

// One file: 0 = encoded, 1 = left to the host (100000 characters or more: the reference
// tokenises such a text in pieces cut at a space, search.py:47-63, with rules of its own for
// where no space is found; or malformed UTF-8, which the reference's open() refuses),
// negative = errno of the read.
int32_t encode_file(const fs_textenc* enc, const char* path, std::vector<uint8_t>& buf, ThreadOut& o) {
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return -errno;
  struct stat st;
  if (fstat(fd, &st) != 0) { const int e = errno; close(fd); return -e; }
  const size_t size = (size_t)st.st_size;
  if (size >= 100000) { close(fd); return 1; }           // (bytes >= characters)
  buf.resize(size + 8);
  size_t got = 0;
  while (got < size) {
    const ssize_t r = read(fd, buf.data() + got, size - got);
    if (r < 0) { const int e = errno; close(fd); return -e; }
    if (r == 0) break;
    got += (size_t)r;
  }
  close(fd);
  memset(buf.data() + got, 0, 8);
  const uint8_t* p = buf.data();
  const uint8_t* end = p + got;
  const size_t tok0 = o.tok.size();
  while (p < end) {
    int sl;
    while (p < end && (sl = space_len(p, end)) != 0) p += sl;
    if (p >= end) break;
    const uint8_t* c0 = p;
    bool ascii = true;
    while (p < end) {
      if (*p < 0x80) {
        const uint8_t c = *p;
        if (c == 0x20 || (c >= 0x09 && c <= 0x0d) || (c >= 0x1c && c <= 0x1f)) break;
        ++p;
        continue;
      }
      if (space_len(p, end)) break;
      const int ul = utf8_len(p, end);
      if (!ul) { o.tok.resize(tok0); return 1; }         // malformed: the host's decoder decides
      ascii = false;
      p += ul;
    }
    (void)ascii;
    const uint32_t n = (uint32_t)(p - c0);
    const uint64_t h = hash_bytes_padded(c0, n);
    if (const Entry* e = find(enc->table, enc->arena, c0, n, h)) {
      if (e->n_pieces == 1) {
        o.tok.push_back((uint32_t)e->first);
      } else {
        const uint32_t* ids = enc->pieces.data() + e->first;
        o.tok.insert(o.tok.end(), ids, ids + e->n_pieces);
      }
    } else {
      o.tok.push_back(0x80000000u | unknown_index(o, c0, n, h));
    }
  }
  return 0;
}

}  // namespace

extern "C" int fs_textenc_create(fs_textenc** out) {
  if (!out) return FS_E_INVALID;
  *out = new (std::nothrow) fs_textenc();
  return *out ? FS_OK : FS_E_NOMEM;
}

extern "C" void fs_textenc_destroy(fs_textenc* enc) { delete enc; }

extern "C" int fs_textenc_add(fs_textenc* enc, const uint8_t* chunk_bytes, const uint64_t* chunk_off,
                              uint64_t n_chunks, const uint64_t* piece_off, const uint32_t* piece_ids) {
  if (!enc || !chunk_off || !piece_off || (n_chunks && (!chunk_bytes || !piece_ids))) return FS_E_INVALID;
  for (uint64_t i = 0; i < n_chunks; ++i) {
    const uint8_t* p = chunk_bytes + chunk_off[i];
    const uint64_t n = chunk_off[i + 1] - chunk_off[i];
    const uint64_t np = piece_off[i + 1] - piece_off[i];
    if (!n || !np || n > 0xFFFFFFFFull || np > 0xFFFFFFFFull) return FS_E_INVALID;
    if ((enc->used + 1) * 2 > enc->table.size()) grow(enc);
    const uint64_t h = hash_bytes(p, n);
    uint8_t pad[16] = {0};                 // (find reads eight bytes of a short chunk)
    if (n <= 8) memcpy(pad, p, n);
    const uint8_t* q = n <= 8 ? pad : p;
    if (find(enc->table, enc->arena, q, (uint32_t)n, h)) continue;       // known already: kept
    Entry e;
    e.hash = h; e.len = (uint32_t)n; e.n_pieces = (uint32_t)np;
    if (n <= 8) {
      e.off = first_bytes(q, (uint32_t)n);
    } else {
      e.off = enc->arena.size();
      enc->arena.insert(enc->arena.end(), p, p + n);
    }
    if (np == 1) {
      e.first = piece_ids[piece_off[i]];
    } else {
      e.first = enc->pieces.size();
      enc->pieces.insert(enc->pieces.end(), piece_ids + piece_off[i], piece_ids + piece_off[i + 1]);
    }
    const uint64_t mask = enc->table.size() - 1;
    uint64_t s = h & mask;
    while (enc->table[s].n_pieces) s = (s + 1) & mask;
    enc->table[s] = e;
    ++enc->used;
  }
  return FS_OK;
}

namespace {

// Reusable two-phase helper: the workers meet the caller once between the phases.
struct Phase {
  std::atomic<uint32_t> done{0};
  std::atomic<uint32_t> go{0};
};

int encode_files_impl(fs_textenc* enc, const char* paths, uint64_t n_files, uint32_t threads,
                      const uint32_t* vec_of_sid, uint64_t n_sid,
                      const uint32_t** tok, uint64_t* n_tok, const uint64_t** work_off,
                      const int32_t** status, const uint8_t** unk_bytes,
                      const uint64_t** unk_off, uint64_t* n_unk,
                      const uint32_t** tok_vec, uint64_t* n_oov, int32_t* ids_equal) {
  std::vector<const char*> path(n_files);
  {
    const char* p = paths;
    for (uint64_t i = 0; i < n_files; ++i) { path[i] = p; p += strlen(p) + 1; }
  }
  const uint32_t T = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(threads ? threads : 1, std::max<uint64_t>(1, n_files / 8)));
  std::vector<ThreadOut> outs(T);
  std::vector<uint64_t> tok_at(T + 1, 0), unk_at(T + 1, 0), oov(T, 0), differ(T, 0), bad_sid(T, 0);
  Phase ph;
  // phase 1: a thread's files, tokens into its own vector; phase 2 (once every thread's token
  // count is known): its tokens into their place of the one output array -- unknown chunks
  // renumbered into the merged list -- and, with a vector id table, their vector ids beside them
  auto phase1 = [&](uint32_t t) {
    const uint64_t lo = n_files * t / T, hi = n_files * (t + 1) / T;
    ThreadOut& o = outs[t];
    std::vector<uint8_t> buf;
    o.work_len.reserve(hi - lo);
    o.status.reserve(hi - lo);
    o.tok.reserve(std::min<uint64_t>((hi - lo) * 2048, 1u << 22));
    for (uint64_t i = lo; i < hi; ++i) {
      const size_t before = o.tok.size();
      const int32_t rc = encode_file(enc, path[i], buf, o);
      if (rc != 0) o.tok.resize(before);
      o.status.push_back(rc);
      o.work_len.push_back(o.tok.size() - before);
    }
  };
  auto phase2 = [&](uint32_t t) {
    const ThreadOut& o = outs[t];
    uint32_t* dst = enc->tok.data() + tok_at[t];
    uint32_t* vdst = vec_of_sid ? enc->tok_vec.data() + tok_at[t] : nullptr;
    const uint32_t ub = (uint32_t)unk_at[t];
    uint64_t n_o = 0, n_d = 0, n_b = 0;
    const size_t n = o.tok.size();
    for (size_t i = 0; i < n; ++i) {
      const uint32_t v = o.tok[i];
      if (v & 0x80000000u) {
        dst[i] = 0x80000000u | (ub + (v & 0x7FFFFFFFu));
        if (vdst) vdst[i] = 0;
        continue;
      }
      dst[i] = v;
      if (vdst) {
        uint32_t x = 0;
        if (v < n_sid) x = vec_of_sid[v]; else ++n_b;
        vdst[i] = x;
        n_o += x >> 31;
        n_d += x != v;
      }
    }
    oov[t] = n_o; differ[t] = n_d; bad_sid[t] = n_b;
  };
  auto between = [&]() {
    for (uint32_t t = 0; t < T; ++t) { tok_at[t + 1] = tok_at[t] + outs[t].tok.size(); unk_at[t + 1] = unk_at[t] + outs[t].unk_count; }
    enc->tok.resize(tok_at[T]);
    if (vec_of_sid) enc->tok_vec.resize(tok_at[T]);
  };
  if (T == 1) {
    phase1(0);
    between();
    phase2(0);
  } else {
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < T; ++t)
      th.emplace_back([&, t]() {
        phase1(t);
        ph.done.fetch_add(1, std::memory_order_release);
        while (!ph.go.load(std::memory_order_acquire)) std::this_thread::yield();
        phase2(t);
      });
    phase1(0);
    while (ph.done.load(std::memory_order_acquire) != T - 1) std::this_thread::yield();
    between();
    ph.go.store(1, std::memory_order_release);
    phase2(0);
    for (auto& x : th) x.join();
  }
  const uint64_t total = tok_at[T], unk_total = unk_at[T];
  enc->work_off.assign(n_files + 1, 0);
  enc->status.resize(n_files);
  enc->unk_bytes.clear();
  enc->unk_off.assign(1, 0);
  uint64_t file = 0, n_o = 0, n_d = 0, n_b = 0;
  for (uint32_t t = 0; t < T; ++t) {
    const ThreadOut& o = outs[t];
    for (size_t i = 0; i < o.work_len.size(); ++i, ++file) {
      enc->work_off[file + 1] = enc->work_off[file] + o.work_len[i];
      enc->status[file] = o.status[i];
    }
    const uint64_t b0 = enc->unk_bytes.size();
    enc->unk_bytes.insert(enc->unk_bytes.end(), o.unk_bytes.begin(), o.unk_bytes.end());
    for (size_t i = 1; i < o.unk_off.size(); ++i) enc->unk_off.push_back(b0 + o.unk_off[i]);
    n_o += oov[t]; n_d += differ[t]; n_b += bad_sid[t];
  }
  if (unk_total >= 0x7FFFFFFFull) return FS_E_UNSUPPORTED;
  if (n_b) return FS_E_INVALID;          // a string id the vector id table does not reach
  *tok = enc->tok.data(); *n_tok = total; *work_off = enc->work_off.data(); *status = enc->status.data();
  *unk_bytes = enc->unk_bytes.data(); *unk_off = enc->unk_off.data(); *n_unk = unk_total;
  if (vec_of_sid) { *tok_vec = enc->tok_vec.data(); *n_oov = n_o; *ids_equal = n_d == 0 && unk_total == 0; }
  return FS_OK;
}

}  // namespace

extern "C" int fs_textenc_encode_files(fs_textenc* enc, const char* paths, uint64_t n_files, uint32_t threads,
                                       const uint32_t** tok, uint64_t* n_tok, const uint64_t** work_off,
                                       const int32_t** status, const uint8_t** unk_bytes,
                                       const uint64_t** unk_off, uint64_t* n_unk) {
  if (!enc || (n_files && !paths) || !tok || !n_tok || !work_off || !status || !unk_bytes || !unk_off || !n_unk)
    return FS_E_INVALID;
  return encode_files_impl(enc, paths, n_files, threads, nullptr, 0, tok, n_tok, work_off, status, unk_bytes, unk_off,
                           n_unk, nullptr, nullptr, nullptr);
}

extern "C" int fs_textenc_encode_files_vec(fs_textenc* enc, const char* paths, uint64_t n_files, uint32_t threads,
                                           const uint32_t* vec_of_sid, uint64_t n_sid,
                                           const uint32_t** tok, uint64_t* n_tok, const uint64_t** work_off,
                                           const int32_t** status, const uint8_t** unk_bytes,
                                           const uint64_t** unk_off, uint64_t* n_unk,
                                           const uint32_t** tok_vec, uint64_t* n_oov, int32_t* ids_equal) {
  if (!enc || (n_files && !paths) || !tok || !n_tok || !work_off || !status || !unk_bytes || !unk_off || !n_unk ||
      !vec_of_sid || !tok_vec || !n_oov || !ids_equal)
    return FS_E_INVALID;
  return encode_files_impl(enc, paths, n_files, threads, vec_of_sid, n_sid, tok, n_tok, work_off, status, unk_bytes,
                           unk_off, n_unk, tok_vec, n_oov, ids_equal);
}
