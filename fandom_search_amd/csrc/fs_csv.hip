// fs_csv.hip -- the batch files of the search command, natively: fs_row records in, the
// reference's CSV bytes out.
//
// Replaces, per 500-work batch, what the reference does in Python at
// /root/reference/search.py:192-218 (the twelve fields of a record) and :331-334
// (write_records: csv.writer(out).writerows(records)), and what search.join_records +
// search.write_records restate here: a record's numbers joined with its work's file name, the
// fan word's text and spaCy key, and the matched script word's columns, then written in the
// `excel` dialect -- ',' between fields, "\r\n" behind a record, a field quoted (and its quotes
// doubled) when it holds ',', '"', CR or LF, None as the empty field, integers by str(),
// floats by repr().  100 000 works make three million records; in Python that is the largest
// share of the command's host time (six core-seconds per 100 000 works against four for
// reading and tokenising them, fs_text.hip).
//
// float repr: Python prints the shortest digit string that reads back as the same double,
// in positional notation when the decimal exponent lies in [-4, 16), else as d.ddde-XX;
// std::to_chars gives the same shortest digits (scientific), the layout is redone here.
// tests/test_csvw.py holds the bytes against csv.writer on the golden records, on fields with
// quotes / commas / line breaks, and on a million random doubles against repr().
//
// Host code only (no kernel, no HIP call); it lives in the library because the C-ABI is the
// drop-in boundary (include/fandom_search.h: fs_csvw_*).
#include "../../include/fandom_search.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <charconv>
#include <new>
#include <string>
#include <vector>

namespace {

struct Strings {                         // a table of UTF-8 strings
  std::vector<uint8_t> bytes;
  std::vector<uint64_t> off{0};
  uint64_t size() const { return off.size() - 1; }
  const uint8_t* at(uint64_t i, uint64_t* n) const { *n = off[i + 1] - off[i]; return bytes.data() + off[i]; }
  void assign(const uint8_t* b, const uint64_t* o, uint64_t n) {
    bytes.assign(b, b + (n ? o[n] : 0));
    off.assign(o, o + n + 1);
    if (!n) off.assign(1, 0);
  }
};

// spaCy's StringStore key: MurmurHash64A of the UTF-8 bytes, seed 1 (vocab.murmurhash64a)
uint64_t murmur64a(const uint8_t* p, uint64_t n, uint64_t seed) {
  const uint64_t m = 0xc6a4a7935bd1e995ull;
  uint64_t h = seed ^ (n * m);
  const uint64_t body = n - n % 8;
  for (uint64_t i = 0; i < body; i += 8) {
    uint64_t k;
    memcpy(&k, p + i, 8);
    k *= m; k ^= k >> 47; k *= m;
    h ^= k; h *= m;
  }
  if (n % 8) {
    uint64_t k = 0;
    memcpy(&k, p + body, n % 8);
    h ^= k; h *= m;
  }
  h ^= h >> 47; h *= m; h ^= h >> 47;
  return h;
}

// (all of the appends write through a raw pointer into space the caller has made sure of: a record's
// bytes are bounded by its strings' lengths, fs_csvw_format reserves per record)

// repr(float) of CPython 3 (float_repr_style 'short'): PyOS_double_to_string(x, 'r', 0, Py_DTSF_ADD_DOT_0);
// at most 24 bytes
inline char* put_repr(char* o, double x) {
  if (isnan(x)) { memcpy(o, "nan", 3); return o + 3; }
  if (isinf(x)) { if (x < 0) { memcpy(o, "-inf", 4); return o + 4; } memcpy(o, "inf", 3); return o + 3; }
  if (x == 0.0) { if (signbit(x)) *o++ = '-'; memcpy(o, "0.0", 3); return o + 3; }
  char buf[40];
  const auto r = std::to_chars(buf, buf + sizeof buf, x, std::chars_format::scientific);
  // [-]d[.ddd]e[+-]XX
  const char* p = buf;
  if (*p == '-') { *o++ = '-'; ++p; }
  const char* e = p;
  while (e < r.ptr && *e != 'e') ++e;
  char digits[24];
  int nd = 0;
  for (const char* q = p; q < e; ++q)
    if (*q != '.') digits[nd++] = *q;
  int exp10 = 0;
  {
    const char* q = e + 1;
    const bool neg = *q == '-';
    if (*q == '-' || *q == '+') ++q;
    for (; q < r.ptr; ++q) exp10 = exp10 * 10 + (*q - '0');
    if (neg) exp10 = -exp10;
  }
  const int decpt = exp10 + 1;             // digits d1 d2 ... mean 0.d1d2... * 10^decpt
  if (decpt > -4 && decpt <= 16) {
    if (decpt <= 0) {
      *o++ = '0'; *o++ = '.';
      for (int k = 0; k < -decpt; ++k) *o++ = '0';
      memcpy(o, digits, (size_t)nd); o += nd;
    } else if (decpt >= nd) {
      memcpy(o, digits, (size_t)nd); o += nd;
      for (int k = 0; k < decpt - nd; ++k) *o++ = '0';
      *o++ = '.'; *o++ = '0';
    } else {
      memcpy(o, digits, (size_t)decpt); o += decpt;
      *o++ = '.';
      memcpy(o, digits + decpt, (size_t)(nd - decpt)); o += nd - decpt;
    }
    return o;
  }
  *o++ = digits[0];
  if (nd > 1) { *o++ = '.'; memcpy(o, digits + 1, (size_t)(nd - 1)); o += nd - 1; }
  *o++ = 'e';
  int ex = decpt - 1;
  *o++ = ex < 0 ? '-' : '+';
  if (ex < 0) ex = -ex;
  char eb[8];
  int ne = 0;
  do { eb[ne++] = (char)('0' + ex % 10); ex /= 10; } while (ex);
  if (ne < 2) eb[ne++] = '0';
  while (ne) *o++ = eb[--ne];
  return o;
}

inline char* put_u64(char* o, uint64_t v) {          // at most 20 bytes
  return std::to_chars(o, o + 20, v).ptr;
}

// a string field in QUOTE_MINIMAL: quoted when it holds the delimiter, the quote character or a
// character of the line terminator; quotes inside are doubled.  At most 2 n + 2 bytes.
inline char* put_field(char* o, const uint8_t* p, uint64_t n) {
  bool quote = false;
  for (uint64_t i = 0; i < n; ++i) quote |= p[i] == ',' || p[i] == '"' || p[i] == '\r' || p[i] == '\n';
  if (!quote) { memcpy(o, p, (size_t)n); return o + n; }
  *o++ = '"';
  for (uint64_t i = 0; i < n; ++i) {
    if (p[i] == '"') *o++ = '"';
    *o++ = (char)p[i];
  }
  *o++ = '"';
  return o;
}

}  // namespace

struct fs_csvw {
  Strings word, orth, character, scene;      // the script's columns, as the text a record shows
  Strings fan;                               // the vocabulary's strings, by string id
  std::vector<uint64_t> fan_orth;            // ... and their spaCy keys
  std::string buf;
};

extern "C" int fs_csvw_create(fs_csvw** out) {
  if (!out) return FS_E_INVALID;
  *out = new (std::nothrow) fs_csvw();
  return *out ? FS_OK : FS_E_NOMEM;
}

extern "C" void fs_csvw_destroy(fs_csvw* w) { delete w; }

extern "C" int fs_csvw_set_script(fs_csvw* w, uint64_t n_script,
                                  const uint8_t* word_bytes, const uint64_t* word_off,
                                  const uint8_t* orth_bytes, const uint64_t* orth_off,
                                  const uint8_t* char_bytes, const uint64_t* char_off,
                                  const uint8_t* scene_bytes, const uint64_t* scene_off) {
  if (!w || !word_off || !orth_off || !char_off || !scene_off) return FS_E_INVALID;
  if (n_script && (word_off[0] || orth_off[0] || char_off[0] || scene_off[0])) return FS_E_INVALID;
  w->word.assign(word_bytes, word_off, n_script);
  w->orth.assign(orth_bytes, orth_off, n_script);
  w->character.assign(char_bytes, char_off, n_script);
  w->scene.assign(scene_bytes, scene_off, n_script);
  return FS_OK;
}

extern "C" int fs_csvw_add_strings(fs_csvw* w, const uint8_t* bytes, const uint64_t* off, uint64_t n) {
  if (!w || !off || (n && off[n] && !bytes)) return FS_E_INVALID;
  const uint64_t base = w->fan.bytes.size();
  if (n) w->fan.bytes.insert(w->fan.bytes.end(), bytes + off[0], bytes + off[n]);
  for (uint64_t i = 0; i < n; ++i) {
    if (off[i + 1] < off[i]) return FS_E_INVALID;
    w->fan.off.push_back(base + (off[i + 1] - off[0]));
    w->fan_orth.push_back(murmur64a(bytes + off[i], off[i + 1] - off[i], 1));
  }
  return FS_OK;
}

extern "C" uint64_t fs_csvw_strings(const fs_csvw* w) { return w ? w->fan.size() : 0; }

extern "C" int fs_csvw_format(fs_csvw* w, const fs_row* rows, uint64_t n_rows,
                              const uint8_t* name_bytes, const uint64_t* name_off, uint64_t n_works,
                              const uint32_t* fan_sid, const uint8_t** out, uint64_t* out_len) {
  if (!w || (n_rows && (!rows || !fan_sid || !name_off)) || !out || !out_len) return FS_E_INVALID;
  std::string& b = w->buf;
  const uint64_t n_script = w->word.size(), n_fan = w->fan.size();
  size_t used = 0;
  if (b.size() < (size_t)n_rows * 128 + 4096) b.resize((size_t)n_rows * 128 + 4096);
  for (uint64_t i = 0; i < n_rows; ++i) {
    const fs_row& r = rows[i];
    if (r.work >= n_works || r.orig_ix >= n_script || fan_sid[i] >= n_fan) return FS_E_INVALID;
    uint64_t n_name = name_off[r.work + 1] - name_off[r.work], n_fan_w, n_word, n_orth, n_char, n_scene;
    const uint8_t* p_fan = w->fan.at(fan_sid[i], &n_fan_w);
    const uint8_t* p_word = w->word.at(r.orig_ix, &n_word);
    const uint8_t* p_orth = w->orth.at(r.orig_ix, &n_orth);
    const uint8_t* p_char = w->character.at(r.orig_ix, &n_char);
    const uint8_t* p_scene = w->scene.at(r.orig_ix, &n_scene);
    const size_t need = (size_t)(2 * (n_name + n_fan_w + n_word + n_char) + n_orth + n_scene) + 8 + 4 * 20 + 2 * 24 + 13;
    if (b.size() - used < need) b.resize(std::max(b.size() * 2, used + need));
    char* o = &b[used];
    o = put_field(o, name_bytes + name_off[r.work], n_name); *o++ = ',';
    o = put_u64(o, r.fan_ix); *o++ = ',';
    o = put_field(o, p_fan, n_fan_w); *o++ = ',';
    o = put_u64(o, w->fan_orth[fan_sid[i]]); *o++ = ',';
    o = put_u64(o, r.orig_ix); *o++ = ',';
    o = put_field(o, p_word, n_word); *o++ = ',';
    memcpy(o, p_orth, (size_t)n_orth); o += n_orth; *o++ = ',';
    o = put_field(o, p_char, n_char); *o++ = ',';
    memcpy(o, p_scene, (size_t)n_scene); o += n_scene; *o++ = ',';
    o = put_repr(o, r.dist); *o++ = ',';
    o = put_u64(o, r.lev); *o++ = ',';
    o = put_repr(o, r.comb);
    *o++ = '\r'; *o++ = '\n';
    used = (size_t)(o - b.data());
  }
  *out = reinterpret_cast<const uint8_t*>(b.data());
  *out_len = used;
  return FS_OK;
}
