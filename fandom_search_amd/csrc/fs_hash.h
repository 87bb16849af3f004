// fs_hash.h -- n-gram shingle hash shared by the host index builder and the
// device scan / verify kernels.  Everything here is integer and exact; a hash
// only ever selects candidates, every candidate is compared id-for-id before it
// becomes a hit.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FS_HD __host__ __device__ __forceinline__
#else
#define FS_HD static inline
#endif

// per-token premix: one 32-bit multiply per token, shared by the n windows that
// contain the token
#define FS_TOKEN_MUL 0x9E3779B1u

FS_HD uint32_t fs_rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

FS_HD uint32_t fs_premix(uint32_t tok) { return tok * FS_TOKEN_MUL; }

// fold one premixed token into a running window hash
FS_HD uint32_t fs_fold(uint32_t h, uint32_t m) { return fs_rotl(h, 7) + m; }

FS_HD uint32_t fs_finish(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  return h;
}

// hash of the n vector ids t[0..n)
FS_HD uint32_t fs_gram_hash(const uint32_t* t, int n) {
  uint32_t h = fs_premix(t[0]);
  for (int k = 1; k < n; ++k) h = fs_fold(h, fs_premix(t[k]));
  return fs_finish(h);
}

// Blocked Bloom filter: one 32-bit word, three bit positions, all taken from
// disjoint bit fields of the hash (word index: top `log2_words` bits).
FS_HD uint32_t fs_bloom_word(uint32_t h, int log2_words) { return h >> (32 - log2_words); }
FS_HD uint32_t fs_bloom_mask(uint32_t h) {
  return (1u << (h & 31)) | (1u << ((h >> 5) & 31)) | (1u << ((h >> 10) & 31));
}

// slot of the exact (verification) table, 2^log2_slots entries
FS_HD uint32_t fs_table_slot(uint32_t h, int log2_slots) {
  return (h * 0xC2B2AE35u) >> (32 - log2_slots);
}
