// fs_hash.h -- n-gram shingle hash shared by the host index builder and the
// device scan / verify kernels.  Everything here is integer and exact; a hash
// only ever selects candidates, every candidate is compared id-for-id before it
// becomes a hit.
//
//   m(t)  = low 32 bits of (t & 0xFFFFFF) * 0x9E3779      one full-rate
//                                                         v_mul_u32_u24 per token
//   X(w)  = XOR_k rotl(m(t[k]), 7*(n-1-k) mod 32)         k = 0..n-1
//
// The XOR-rotate fold slides along the token stream:
//   X(w+1) = rotl(X(w) ^ rotl(m(t[0]), 7*(n-1)), 7) ^ m(t[n])
// so a lane that owns four consecutive windows pays the full fold once and three
// operations for each further window.  m() is injective for ids < 2^24 (odd
// multiplier), the rotation amounts 7j mod 32 are distinct for j < 32, so two
// windows that differ in one token never collide.  Measured false-positive rate
// of the Bloom test on the synthetic corpora equals that of a multiply-finalised
// hash (0.42 % at one filter word per n-gram).
//
// Vector ids on this path are embedding rows (< 2^24); out-of-vocabulary ids
// (bit 31 set) never reach the exact scan.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FS_HD __host__ __device__ __forceinline__
#define FS_HDC __host__ __device__ constexpr
#else
#define FS_HD static inline
#define FS_HDC static constexpr
#endif

#define FS_TOKEN_MUL24 0x9E3779u
#define FS_MAX_EXACT_ID (1u << 24)

FS_HD uint32_t fs_rotl(uint32_t x, int r) {
  r &= 31;
  return (x << r) | (x >> ((32 - r) & 31));
}

FS_HD uint32_t fs_premix(uint32_t tok) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul24(tok, FS_TOKEN_MUL24);
#else
  return (uint32_t)((uint64_t)(tok & 0xFFFFFFu) * FS_TOKEN_MUL24);
#endif
}

// full-rate mixing for filters whose hashes are made many at a time (k_lsh_sift)
FS_HD uint32_t fs_mul24(uint32_t a, uint32_t m) {        // low 32 bits of (a mod 2^24) * m, m < 2^24
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul24(a, m);
#else
  return (uint32_t)((uint64_t)(a & 0xFFFFFFu) * m);
#endif
}
FS_HD uint32_t fs_mix24(uint32_t x) {
  x ^= x >> 16;
  return fs_mul24(x, 0x9E3779u) ^ fs_rotl(fs_mul24(x >> 8, 0x85EBCBu), 16);
}

// rotation applied to the token that is j places before the end of the window
FS_HD int fs_rot_of(int j) { return (7 * j) & 31; }

// hash of the n vector ids t[0..n)
FS_HD uint32_t fs_gram_hash(const uint32_t* t, int n) {
  uint32_t x = 0;
  for (int k = 0; k < n; ++k) x ^= fs_rotl(fs_premix(t[k]), fs_rot_of(n - 1 - k));
  return x;
}

// Blocked Bloom filter: one 32-bit word, three bit positions, all taken from
// disjoint bit fields of the hash (word index: top `log2_words` bits; bit
// positions: bits 0-4, 8-12 and 13-17 -- the second field sits in byte 1 so that
// the scan can use it as a shift amount through an SDWA byte select).
FS_HD uint32_t fs_bloom_word(uint32_t h, int log2_words) { return h >> (32 - log2_words); }
FS_HD uint32_t fs_bloom_mask(uint32_t h) {
  return (1u << (h & 31)) | (1u << ((h >> 8) & 31)) | (1u << ((h >> 13) & 31));
}
// the same test written as three shifts of the filter word (a 32-bit shift uses
// the low five bits of its amount)
FS_HD uint32_t fs_bloom_test(uint32_t word, uint32_t h) {
  return (word >> (h & 31)) & (word >> ((h >> 8) & 31)) & (word >> ((h >> 13) & 31)) & 1u;
}

// Sub-shingle filter of k_scan_rows: one bit per script K-gram (K = fs_sub_k(n) <= n),
// word = top log2_words bits of the K-gram's hash, bit = its low five bits.  A fan window
// of n ids can equal a script window only if all its n-K+1 K-grams are script K-grams,
// so "n-K+1 consecutive set bits" is a sound candidate test (no false negatives) that
// costs one LDS word, one shift and one funnel shift per token instead of the three-bit
// test of the full n-gram; every candidate is still compared id for id afterwards.
// K = 4 for n >= 6 (n-3 tests per window), 3 for n = 4, 5; n <= 3 keeps the Bloom test.
constexpr int fs_sub_k(int n) { return n >= 6 ? 4 : n >= 4 ? 3 : 0; }
// The K-gram hash is a polynomial in 2^S of the premixed ids, modulo 2^32, with K * S >= 32:
//   P(w) = sum_k m(t[k]) << (S * (K - 1 - k))        S = 8 for K = 4, 11 for K = 3
// so the oldest id has left the 32 bits when the hash slides on, and a step costs ONE
// instruction: P(w + 1) = (P(w) << S) + m(t[K])  (v_lshl_add_u32; the XOR-rotate fold above
// needs three).  Filter word = bits [18, 18 + log2_words) of the hash (log2_words <= 14: one
// SDWA instruction turns bits 16.. into the word's byte offset), bit = bits 8..12 (an SDWA
// byte select as shift amount).  A hash only selects candidates; every candidate is compared
// id for id afterwards.
constexpr int fs_sub_shift(int K) { return K >= 4 ? 8 : 11; }
#define FS_SUB_MAX_LOG2_WORDS 14
FS_HD uint32_t fs_sub_hash(const uint32_t* t, int K) {
  uint32_t x = 0;
  for (int k = 0; k < K; ++k) x = (x << fs_sub_shift(K)) + fs_premix(t[k]);
  return x;
}
FS_HD uint32_t fs_sub_word(uint32_t h, int log2_words) { return (h >> 18) & ((1u << log2_words) - 1u); }
FS_HD uint32_t fs_sub_bit(uint32_t h) { return (h >> 8) & 31u; }

// Exact (verification) table, hash-and-displace: 2^log2_buckets buckets, each with a
// displacement seed d; an n-gram with hash h lives in slot fs_table_slot_d(h, d).
// The seed of a bucket is chosen at build time so that all its n-grams fall into
// free slots of their own.  FS_DISP_OVERFLOW marks a bucket that could not be
// separated (different n-grams with the same 32-bit hash): its members were placed
// by linear probing from their slot.
#define FS_DISP_OVERFLOW 0x80000000u
#define FS_DISP_MASK 0x7FFFFFFFu
FS_HD uint32_t fs_table_bucket(uint32_t h, int log2_buckets) {
  return (h * 0x85EBCA6Bu) >> (32 - log2_buckets);
}
FS_HD uint32_t fs_table_slot_d(uint32_t h, uint32_t d, int log2_slots) {
  return ((h ^ (d * 0x9E3779B9u)) * 0xC2B2AE35u) >> (32 - log2_slots);
}

// Batch table of k_scan_rows: the exact table's placement (slot of an n-gram as above)
// with 64-byte entries that hold the n-gram's ids together with its best record for one
// fan batch, so a lookup is one cache line; the displacement seeds travel as bytes
// (FS_DISP8_WIDE: look the seed up in the 32-bit array) and live in LDS during the scan.
#define FS_CTAB_WORDS 16
#define FS_CTAB_MAX_N 8
#define FS_DISP8_WIDE 0xFFu

// One-slot-wildcard keys (LSH pipeline, windows that may differ from a script window in
// at most one slot): key j of a window is a hash of its ids with slot j left out, i.e.
// the XOR-rotate fold of all n ids with slot j's term taken out again, mixed with j.  A
// blocked Bloom filter (one 32-bit word, four bits) holds the n keys of every script
// window; a fan window none of whose n keys is present differs from every script window
// in two or more slots.
FS_HD uint32_t fs_wild_key(uint32_t fold_all, uint32_t term_j, int j) {
  uint32_t h = (fold_all ^ term_j) + 0x9E3779B9u * (uint32_t)(j + 1);
  h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
// The filter is *grouped*: the n slots fall into three groups (slot k: group 3k / n), and the key
// that leaves out a slot of group X lies in the 16-byte block chosen by the ids of the other
// two groups -- which the window shares with the script window whatever its slot of group X
// holds.  A window therefore looks at three blocks (one 16-byte read each) for its n keys; a
// key sets four of the block's 128 bits.
FS_HD int fs_wild_group(int k, int n) { return 3 * k / n; }
FS_HD uint32_t fs_wild_block(uint32_t fold_others, int group, int log2_blocks) {
  const uint32_t h = fs_mix24(fold_others + 0x7F4A7C15u * (uint32_t)(group + 1));
  return h >> (32 - log2_blocks);
}
// The filter's own hash of a key (fs_wild_key is the exact map's): full-rate instructions only --
// k_lsh_sift makes n of them per candidate and is bound by its VALU work -- and the four bits of a
// key one to each dword of the block, at bits 0-4 of the hash's four bytes, so that a test is
// four shifts with an SDWA byte select as the amount and two ands.  (A hash only selects: what
// the filter lets through is decided exactly behind it.)
FS_HD uint32_t fs_wild_fkey(uint32_t fold_all, uint32_t term_j, int j) {
  return fs_mix24((fold_all ^ term_j) + 0x9E3779B9u * (uint32_t)(j + 1));
}
FS_HD uint32_t fs_wild_fbit(uint32_t h, int i) { return (h >> (8 * i)) & 31u; }   // its bit in dword i
// n = 6 over component ids: the keys of the two middle slots (2 and 3) of every script window in a
// filter of their own that k_scan_near holds in LDS (2^14 words, three bits per key as the
// Bloom filter above)
#define FS_NEAR6_LOG2_WORDS 14

// The same keys in an exact map (buckets of four {key, script window + 1}, one entry per distinct
// script n-gram and slot, a full bucket spills into the next): the script windows that equal a
// fan window in all slots but one can be enumerated, not just shown to be possible.
FS_HD uint32_t fs_wmap_slot(uint32_t h, int log2_slots) { return (h * 0x9E3779B1u) >> (32 - log2_slots); }

// Subset keys of the share rule (fs_lsh.hip, "windows on tables that are not unit length"): a key
// names the slots of a subset (`mask`) and the component ids the window holds there -- the fold of
// the subset's terms, mixed with the mask.  A blocked Bloom filter (one 32-bit word, three bits:
// fs_bloom_word / fs_bloom_test) holds the keys of every script window.
FS_HD uint32_t fs_share_term(uint32_t comp, int k) { return fs_rotl(fs_premix(comp), fs_rot_of(k)); }
FS_HD uint32_t fs_share_raw(uint32_t fold, uint32_t mask) { return fold + 0x9E3779B9u * mask; }
FS_HD uint32_t fs_share_finish(uint32_t h) {
  h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
FS_HD uint32_t fs_share_key(uint32_t fold, uint32_t mask) { return fs_share_finish(fs_share_raw(fold, mask)); }
// ... and the few bits of a component id the pairs' test compares: n of them in a 64-bit word
FS_HD int fs_share_sig_bits(int n) { return 64 / n > 10 ? 10 : 64 / n; }
FS_HD uint32_t fs_share_sig(uint32_t comp, int n) { return (fs_mix24(comp) >> 7) & ((1u << fs_share_sig_bits(n)) - 1u); }
// Windows of more than six slots take the share rule block by block (fs_lsh.hip): the slots in two or
// three runs of at most five, the subsets and their keys inside a run.  Run r of a window of n slots
// starts at fs_share_block_start(n, r) (r = fs_share_blocks(n): the window's end).
FS_HDC int fs_share_blocks(int n) { return n <= 6 ? 1 : n <= 10 ? 2 : 3; }
FS_HDC int fs_share_block_start(int n, int r) {
  const int b = fs_share_blocks(n);
  if (r <= 0) return 0;
  if (r >= b) return n;
  // the longer runs first: 7 = 4 + 3, 9 = 5 + 4, 11 = 4 + 4 + 3
  return b == 2 ? (n + 1) / 2 : (r == 1 ? (n + 2) / 3 : (n + 2) / 3 + (n + 1) / 3);
}
