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

struct Entry {
  uint64_t hash = 0;
  uint64_t off = 0;      // the chunk's bytes in `arena`
  uint32_t len = 0;
  uint32_t n_pieces = 0; // 0: empty slot
  uint64_t first = 0;    // its string ids in `pieces`
};

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
    if (e.hash == h && e.len == n && memcmp(arena.data() + e.off, p, n) == 0) return &e;
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
    const uint64_t h = hash_bytes(c0, n);
    if (const Entry* e = find(enc->table, enc->arena, c0, n, h)) {
      const uint32_t* ids = enc->pieces.data() + e->first;
      o.tok.insert(o.tok.end(), ids, ids + e->n_pieces);
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
    if (find(enc->table, enc->arena, p, (uint32_t)n, h)) continue;       // known already: kept
    Entry e;
    e.hash = h; e.len = (uint32_t)n; e.n_pieces = (uint32_t)np;
    e.off = enc->arena.size();
    enc->arena.insert(enc->arena.end(), p, p + n);
    e.first = enc->pieces.size();
    enc->pieces.insert(enc->pieces.end(), piece_ids + piece_off[i], piece_ids + piece_off[i + 1]);
    const uint64_t mask = enc->table.size() - 1;
    uint64_t s = h & mask;
    while (enc->table[s].n_pieces) s = (s + 1) & mask;
    enc->table[s] = e;
    ++enc->used;
  }
  return FS_OK;
}

extern "C" int fs_textenc_encode_files(fs_textenc* enc, const char* paths, uint64_t n_files, uint32_t threads,
                                       const uint32_t** tok, uint64_t* n_tok, const uint64_t** work_off,
                                       const int32_t** status, const uint8_t** unk_bytes,
                                       const uint64_t** unk_off, uint64_t* n_unk) {
  if (!enc || (n_files && !paths) || !tok || !n_tok || !work_off || !status || !unk_bytes || !unk_off || !n_unk)
    return FS_E_INVALID;
  std::vector<const char*> path(n_files);
  {
    const char* p = paths;
    for (uint64_t i = 0; i < n_files; ++i) { path[i] = p; p += strlen(p) + 1; }
  }
  const uint32_t T = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(threads ? threads : 1, std::max<uint64_t>(1, n_files / 8)));
  std::vector<ThreadOut> outs(T);
  auto work = [&](uint32_t t) {
    const uint64_t lo = n_files * t / T, hi = n_files * (t + 1) / T;
    ThreadOut& o = outs[t];
    std::vector<uint8_t> buf;
    o.work_len.reserve(hi - lo);
    o.status.reserve(hi - lo);
    for (uint64_t i = lo; i < hi; ++i) {
      const size_t before = o.tok.size();
      const int32_t rc = encode_file(enc, path[i], buf, o);
      if (rc != 0) o.tok.resize(before);
      o.status.push_back(rc);
      o.work_len.push_back(o.tok.size() - before);
    }
  };
  if (T == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < T; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  // merge: tokens in file order, the threads' unknown chunks renumbered into one list (a chunk
  // two threads met appears twice: the host resolves by text, once each)
  uint64_t total = 0, unk_total = 0;
  for (const ThreadOut& o : outs) { total += o.tok.size(); unk_total += o.unk_count; }
  enc->tok.resize(total);
  enc->work_off.assign(n_files + 1, 0);
  enc->status.resize(n_files);
  enc->unk_bytes.clear();
  enc->unk_off.assign(1, 0);
  uint64_t at = 0, file = 0, unk_base = 0;
  for (ThreadOut& o : outs) {
    for (size_t i = 0; i < o.tok.size(); ++i) {
      const uint32_t v = o.tok[i];
      enc->tok[at + i] = (v & 0x80000000u) ? (0x80000000u | (uint32_t)(unk_base + (v & 0x7FFFFFFFu))) : v;
    }
    at += o.tok.size();
    for (size_t i = 0; i < o.work_len.size(); ++i, ++file) {
      enc->work_off[file + 1] = enc->work_off[file] + o.work_len[i];
      enc->status[file] = o.status[i];
    }
    const uint64_t b0 = enc->unk_bytes.size();
    enc->unk_bytes.insert(enc->unk_bytes.end(), o.unk_bytes.begin(), o.unk_bytes.end());
    for (size_t i = 1; i < o.unk_off.size(); ++i) enc->unk_off.push_back(b0 + o.unk_off[i]);
    unk_base += o.unk_count;
  }
  if (unk_total >= 0x7FFFFFFFull) return FS_E_UNSUPPORTED;
  *tok = enc->tok.data(); *n_tok = total; *work_off = enc->work_off.data(); *status = enc->status.data();
  *unk_bytes = enc->unk_bytes.data(); *unk_off = enc->unk_off.data(); *n_unk = unk_total;
  return FS_OK;
}
