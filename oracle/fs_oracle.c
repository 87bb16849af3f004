/* fs_oracle.c -- TEST INFRASTRUCTURE ONLY.  Plain-C CPU restatement of the
 * reference's search path (/root/reference/search.py:65-226 plus the NearPy /
 * python-Levenshtein semantics it calls), in the canonical arithmetic of
 * DESIGN.md.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (libfandomsearch_hip.so)
 * never links or calls it.
 *
 * PARITY UNPINNED: the reference has no tests / fixtures / golden vectors and
 * cannot run in this image (nearpy, spacy, Levenshtein absent).
 * This file is pinned against oracle/search_restated.py (the literal Python
 * restatement, canonical mode: bit-identical rows) and against hand-derived
 * known answers in tests/.
 *
 * It is the reference's algorithm, not the product's: every fan window is
 * hashed with all H random-projection tables, every bucket candidate gets a
 * cosine distance, NearestFilter / threshold / Levenshtein / per-word dedupe
 * follow in the reference's order.  No exact-n-gram shortcut lives here.
 *
 *   mk_vectors                search.py:65-84     vec_of()
 *   window build              search.py:94-95     implicit (ids i..i+n-1)
 *   RandomBinaryProjections   search.py:114-115   window_keys()
 *   Engine.store_vector       search.py:122-123   fo_index_create (CSR buckets)
 *   Engine.neighbours         search.py:178       search_work(): candidates,
 *                                                 UniqueFilter, CosineDistance,
 *                                                 NearestFilter
 *   threshold                 search.py:182-184
 *   Levenshtein               search.py:189-190   lev()
 *   word records + dedupe     search.py:192-226
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off matters: the contract is separate multiply and add.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/fandom_search.h"

typedef struct fo_index {
  fs_config cfg;
  uint32_t n, H, B, D, C;      /* C = H*B projection columns */
  uint64_t n_script, n_win, n_vec;
  float* emb;                  /* [n_vec][D] */
  double* nt;                  /* normals transposed: [n][D][C] */
  uint32_t* svec;              /* [n_script] */
  uint32_t* schars;
  uint64_t* soff;              /* [n_script+1] */
  /* vector slots: table rows 0..n_vec-1, then OOV ids as met */
  uint64_t n_slot, cap_slot;
  uint32_t* oov_id;            /* [n_slot - n_vec] */
  double** vec;                /* [slot] -> D doubles (lazy) */
  double** arow;               /* [slot] -> n*C doubles (lazy) */
  double* q;                   /* [slot] */
  uint8_t* have;               /* [slot] */
  /* LSH tables: CSR per hash */
  uint32_t* boff;              /* [H][2^B + 1] */
  uint32_t* bids;              /* [H][n_win] */
  double* ss;                  /* [n_win] sum of q over the window */
  int threads;
} fo_index;

/* ---- vector slots ------------------------------------------------------- */

static int64_t slot_find(const fo_index* ix, uint32_t id) {
  if (!(id & FS_OOV_FLAG)) return id < ix->n_vec ? (int64_t)id : -1;
  for (uint64_t i = ix->n_vec; i < ix->n_slot; ++i)
    if (ix->oov_id[i - ix->n_vec] == id) return (int64_t)i;
  return -1;
}

static int64_t slot_add(fo_index* ix, uint32_t id) {
  int64_t s = slot_find(ix, id);
  if (s >= 0 || !(id & FS_OOV_FLAG)) return s;
  if (ix->n_slot == ix->cap_slot) {
    uint64_t nc = ix->cap_slot * 2 + 16;
    ix->oov_id = realloc(ix->oov_id, (nc - ix->n_vec) * sizeof(uint32_t));
    ix->vec = realloc(ix->vec, nc * sizeof(double*));
    ix->arow = realloc(ix->arow, nc * sizeof(double*));
    ix->q = realloc(ix->q, nc * sizeof(double));
    ix->have = realloc(ix->have, nc);
    for (uint64_t i = ix->cap_slot; i < nc; ++i) {
      ix->vec[i] = NULL; ix->arow[i] = NULL; ix->have[i] = 0;
    }
    ix->cap_slot = nc;
  }
  ix->oov_id[ix->n_slot - ix->n_vec] = id;
  return (int64_t)ix->n_slot++;
}

/* mk_vectors (search.py:65-84): table row widened to double, or the 3-hot. */
static void vec_of(const fo_index* ix, uint64_t slot, double* out) {
  uint32_t D = ix->D;
  if (slot < ix->n_vec) {
    const float* e = ix->emb + slot * D;
    for (uint32_t d = 0; d < D; ++d) out[d] = (double)e[d];
  } else {
    uint32_t code = ix->oov_id[slot - ix->n_vec] & ~FS_OOV_FLAG;
    for (uint32_t d = 0; d < D; ++d) out[d] = 0.0;
    out[code % D] = 1.0;
    out[(code / D) % D] = 1.0;
    out[code / (D * D)] = 1.0;
  }
}

/* canonical per-token tables:
 *   A[k][c] = seqsum_d normals[c][k*D+d] * vec[d],  q = seqsum_d vec[d]^2 */
static void slot_prepare(fo_index* ix, uint64_t slot) {
  if (ix->have[slot]) return;
  uint32_t D = ix->D, C = ix->C, n = ix->n;
  double* v = malloc(D * sizeof(double));
  double* a = malloc((size_t)n * C * sizeof(double));
  vec_of(ix, slot, v);
  double q = 0.0;
  for (uint32_t d = 0; d < D; ++d) q = q + v[d] * v[d];
  for (uint32_t k = 0; k < n; ++k) {
    double* ak = a + (size_t)k * C;
    for (uint32_t c = 0; c < C; ++c) ak[c] = 0.0;
    const double* ntk = ix->nt + (size_t)k * D * C;
    for (uint32_t d = 0; d < D; ++d) {
      const double* row = ntk + (size_t)d * C;
      double vd = v[d];
      for (uint32_t c = 0; c < C; ++c) ak[c] = ak[c] + row[c] * vd;
    }
  }
  ix->vec[slot] = v; ix->arow[slot] = a; ix->q[slot] = q; ix->have[slot] = 1;
}

/* RandomBinaryProjections.hash_vector for all H tables: key bit j of table h
 * is (projection[h*B+j] > 0.0), first projection = most significant bit. */
static void window_keys(const fo_index* ix, const int64_t* slots,
                        double* p, uint32_t* keys) {
  uint32_t C = ix->C, n = ix->n, B = ix->B;
  const double* a0 = ix->arow[slots[0]];
  for (uint32_t c = 0; c < C; ++c) p[c] = a0[c];
  for (uint32_t k = 1; k < n; ++k) {
    const double* ak = ix->arow[slots[k]] + (size_t)k * C;
    for (uint32_t c = 0; c < C; ++c) p[c] = p[c] + ak[c];
  }
  for (uint32_t h = 0; h < ix->H; ++h) {
    uint32_t key = 0;
    for (uint32_t j = 0; j < B; ++j)
      key = (key << 1) | (p[h * B + j] > 0.0 ? 1u : 0u);
    keys[h] = key;
  }
}

/* ---- index -------------------------------------------------------------- */

void fo_index_destroy(fo_index* ix) {
  if (!ix) return;
  for (uint64_t i = 0; i < ix->cap_slot; ++i) { free(ix->vec[i]); free(ix->arow[i]); }
  free(ix->emb); free(ix->nt); free(ix->svec); free(ix->schars); free(ix->soff);
  free(ix->oov_id); free(ix->vec); free(ix->arow); free(ix->q); free(ix->have);
  free(ix->boff); free(ix->bids); free(ix->ss);
  free(ix);
}

int fo_index_create(const fs_config* cfg, const uint32_t* script_vec,
                    const uint32_t* script_chars, const uint64_t* script_off,
                    uint64_t n_script, const float* emb, uint64_t n_vec,
                    const double* normals, int threads, fo_index** out) {
  if (!cfg || !out || cfg->window_size == 0 || cfg->hash_dimensions > 24 ||
      cfg->hash_dimensions == 0)
    return FS_E_INVALID;
  fo_index* ix = calloc(1, sizeof(fo_index));
  ix->cfg = *cfg;
  ix->n = cfg->window_size; ix->H = cfg->number_of_hashes;
  ix->B = cfg->hash_dimensions; ix->D = cfg->emb_dim; ix->C = ix->H * ix->B;
  ix->n_script = n_script; ix->n_vec = n_vec;
  ix->n_win = n_script >= ix->n ? n_script - ix->n + 1 : 0;
  ix->threads = threads;
  uint32_t n = ix->n, D = ix->D, C = ix->C;

  ix->emb = malloc(n_vec * D * sizeof(float));
  memcpy(ix->emb, emb, n_vec * D * sizeof(float));
  ix->nt = malloc((size_t)n * D * C * sizeof(double));
  for (uint32_t c = 0; c < C; ++c)
    for (uint32_t k = 0; k < n; ++k)
      for (uint32_t d = 0; d < D; ++d)
        ix->nt[((size_t)k * D + d) * C + c] = normals[(size_t)c * n * D + k * D + d];
  ix->svec = malloc((n_script + 1) * sizeof(uint32_t));
  memcpy(ix->svec, script_vec, n_script * sizeof(uint32_t));
  ix->soff = malloc((n_script + 1) * sizeof(uint64_t));
  memcpy(ix->soff, script_off, (n_script + 1) * sizeof(uint64_t));
  ix->schars = malloc((script_off[n_script] + 1) * sizeof(uint32_t));
  memcpy(ix->schars, script_chars, script_off[n_script] * sizeof(uint32_t));

  ix->n_slot = ix->cap_slot = n_vec;
  ix->vec = calloc(n_vec + 1, sizeof(double*));
  ix->arow = calloc(n_vec + 1, sizeof(double*));
  ix->q = calloc(n_vec + 1, sizeof(double));
  ix->have = calloc(n_vec + 1, 1);
  ix->oov_id = NULL;

  int64_t* sslot = malloc((n_script + 1) * sizeof(int64_t));
  for (uint64_t i = 0; i < n_script; ++i) {
    sslot[i] = slot_add(ix, script_vec[i]);
    if (sslot[i] < 0) { free(sslot); fo_index_destroy(ix); return FS_E_INVALID; }
    slot_prepare(ix, (uint64_t)sslot[i]);
  }

  /* Engine.store_vector: every window goes into one bucket per table, in
   * ascending window index (insertion order). */
  uint32_t nb = 1u << ix->B;
  uint64_t W = ix->n_win;
  ix->boff = calloc((size_t)ix->H * (nb + 1), sizeof(uint32_t));
  ix->bids = malloc((size_t)ix->H * (W + 1) * sizeof(uint32_t));
  ix->ss = malloc((W + 1) * sizeof(double));
  uint32_t* keys = malloc((W + 1) * ix->H * sizeof(uint32_t));
  double* p = malloc(C * sizeof(double));
  for (uint64_t w = 0; w < W; ++w) {
    window_keys(ix, sslot + w, p, keys + w * ix->H);
    double ss = 0.0;
    for (uint32_t k = 0; k < n; ++k) ss = ss + ix->q[sslot[w + k]];
    ix->ss[w] = ss;
  }
  for (uint32_t h = 0; h < ix->H; ++h) {
    uint32_t* off = ix->boff + (size_t)h * (nb + 1);
    for (uint64_t w = 0; w < W; ++w) off[keys[w * ix->H + h] + 1]++;
    for (uint32_t b = 0; b < nb; ++b) off[b + 1] += off[b];
    uint32_t* cur = malloc(nb * sizeof(uint32_t));
    memcpy(cur, off, nb * sizeof(uint32_t));
    for (uint64_t w = 0; w < W; ++w)
      ix->bids[(size_t)h * W + cur[keys[w * ix->H + h]]++] = (uint32_t)w;
    free(cur);
  }
  free(p); free(keys); free(sslot);
  *out = ix;
  return FS_OK;
}

/* ---- per-work search ---------------------------------------------------- */

typedef struct { uint32_t* a; uint32_t* b; uint32_t* row; size_t cap; } levbuf;

static void lev_reserve(levbuf* lb, size_t need) {
  if (need <= lb->cap) return;
  lb->cap = need * 2 + 64;
  lb->a = realloc(lb->a, lb->cap * sizeof(uint32_t));
  lb->b = realloc(lb->b, lb->cap * sizeof(uint32_t));
  lb->row = realloc(lb->row, (lb->cap + 1) * sizeof(uint32_t));
}

/* Levenshtein.distance(match_str, fan_context) of search.py:189-190:
 *   match_str   = script words s..s+n-1 joined by one space
 *   fan_context = str(list of tokens) = '[' + ', '.join(texts) + ']'      */
static uint32_t lev(const fo_index* ix, uint64_t s, const uint32_t* fstr,
                    const uint32_t* chars, const uint64_t* off, levbuf* lb) {
  uint32_t n = ix->n;
  size_t la = 0, lbn = 2;
  for (uint32_t k = 0; k < n; ++k) {
    la += ix->soff[s + k + 1] - ix->soff[s + k] + (k ? 1 : 0);
    lbn += off[fstr[k] + 1] - off[fstr[k]] + (k ? 2 : 0);
  }
  lev_reserve(lb, la > lbn ? la : lbn);
  size_t i = 0;
  for (uint32_t k = 0; k < n; ++k) {
    if (k) lb->a[i++] = ' ';
    for (uint64_t c = ix->soff[s + k]; c < ix->soff[s + k + 1]; ++c) lb->a[i++] = ix->schars[c];
  }
  i = 0;
  lb->b[i++] = '[';
  for (uint32_t k = 0; k < n; ++k) {
    if (k) { lb->b[i++] = ','; lb->b[i++] = ' '; }
    for (uint64_t c = off[fstr[k]]; c < off[fstr[k] + 1]; ++c) lb->b[i++] = chars[c];
  }
  lb->b[i++] = ']';
  uint32_t* row = lb->row;
  for (size_t j = 0; j <= lbn; ++j) row[j] = (uint32_t)j;
  for (size_t x = 1; x <= la; ++x) {
    uint32_t diag = row[0];
    row[0] = (uint32_t)x;
    uint32_t ca = lb->a[x - 1];
    for (size_t j = 1; j <= lbn; ++j) {
      uint32_t up = row[j];
      uint32_t best = diag + (ca != lb->b[j - 1]);
      if (up + 1 < best) best = up + 1;
      if (row[j - 1] + 1 < best) best = row[j - 1] + 1;
      diag = up;
      row[j] = best;
    }
  }
  return row[lbn];
}

typedef struct {
  double* p; uint32_t* keys; int64_t* slots;
  uint32_t* cand; size_t cand_cap; uint32_t* seen; uint32_t stamp;
  uint32_t* top_s; double* top_d;
  levbuf lb;
  fs_row* best; uint8_t* best_set; size_t best_cap;
} scratch;

static void search_work(const fo_index* ix, uint32_t work, const uint32_t* tv,
                        const uint32_t* ts, uint64_t T, const uint32_t* chars,
                        const uint64_t* off, scratch* sc, fs_row** out,
                        uint64_t* n_out, uint64_t* n_cand, uint64_t* n_match) {
  uint32_t n = ix->n, H = ix->H, D = ix->D, N = ix->cfg.nearest_n;
  uint64_t W = ix->n_win;
  uint32_t nb = 1u << ix->B;
  *out = NULL; *n_out = 0;
  if (T < n) return;
  if (T > sc->best_cap) {
    sc->best_cap = T;
    sc->best = realloc(sc->best, T * sizeof(fs_row));
    sc->best_set = realloc(sc->best_set, T);
    sc->slots = realloc(sc->slots, T * sizeof(int64_t));
  }
  memset(sc->best_set, 0, T);
  for (uint64_t i = 0; i < T; ++i) sc->slots[i] = slot_find(ix, tv[i]);

  for (uint64_t f = 0; f + n <= T; ++f) {
    window_keys(ix, sc->slots + f, sc->p, sc->keys);
    /* Engine._get_candidates: bucket contents of table 0, 1, ... appended */
    size_t nc = 0;
    if (++sc->stamp == 0) { memset(sc->seen, 0, (W + 1) * sizeof(uint32_t)); sc->stamp = 1; }
    for (uint32_t h = 0; h < H; ++h) {
      const uint32_t* o = ix->boff + (size_t)h * (nb + 1) + sc->keys[h];
      for (uint32_t e = o[0]; e < o[1]; ++e) {
        uint32_t s = ix->bids[(size_t)h * W + e];
        if (ix->cfg.unique_filter) {        /* UniqueFilter: first insertion */
          if (sc->seen[s] == sc->stamp) continue;
          sc->seen[s] = sc->stamp;
        }
        if (nc == sc->cand_cap) {
          sc->cand_cap = sc->cand_cap * 2 + 256;
          sc->cand = realloc(sc->cand, sc->cand_cap * sizeof(uint32_t));
        }
        sc->cand[nc++] = s;
      }
    }
    *n_cand += nc;
    if (!nc) continue;
    /* CosineDistance on unit vectors, canonical form:
     *   1.0 - SF / (sqrt(SS) * sqrt(FF)) */
    double ff = 0.0;
    for (uint32_t k = 0; k < n; ++k) ff = ff + ix->q[sc->slots[f + k]];
    double nf = sqrt(ff);
    uint32_t nt = 0;                       /* NearestFilter: stable top-N */
    for (size_t ci = 0; ci < nc; ++ci) {
      uint32_t s = sc->cand[ci];
      double sf = 0.0;
      for (uint32_t k = 0; k < n; ++k) {
        int64_t fs = sc->slots[f + k];
        int64_t ssl = slot_find(ix, ix->svec[s + k]);
        if (fs == ssl) { sf = sf + ix->q[fs]; continue; }
        const double* u = ix->vec[ssl]; const double* v = ix->vec[fs];
        double g = 0.0;
        for (uint32_t d = 0; d < D; ++d) g = g + u[d] * v[d];
        sf = sf + g;
      }
      double dist = 1.0 - sf / (sqrt(ix->ss[s]) * nf);
      if (dist != dist) continue;          /* NaN: zero vector, unreachable */
      uint32_t pos = nt;
      while (pos > 0 && dist < sc->top_d[pos - 1]) --pos;   /* ties stay behind */
      if (pos >= N) continue;
      uint32_t last = nt < N ? nt : N - 1;
      for (uint32_t m = last; m > pos; --m) { sc->top_d[m] = sc->top_d[m - 1]; sc->top_s[m] = sc->top_s[m - 1]; }
      sc->top_d[pos] = dist; sc->top_s[pos] = s;
      if (nt < N) ++nt;
    }
    for (uint32_t m = 0; m < nt; ++m) {
      double dist = sc->top_d[m];
      if (!(dist < ix->cfg.distance_threshold)) continue;   /* search.py:184 */
      uint32_t s = sc->top_s[m];
      ++*n_match;
      uint32_t fstr[64];
      for (uint32_t k = 0; k < n; ++k) fstr[k] = ts ? ts[f + k] : tv[f + k];
      uint32_t lv = lev(ix, s, fstr, chars, off, &sc->lb);
      double comb = dist * (double)lv;
      for (uint32_t k = 0; k < n; ++k) {    /* search.py:192-218, 224-225 */
        uint64_t fw = f + k;
        if (sc->best_set[fw] && !(comb < sc->best[fw].comb)) continue;
        sc->best_set[fw] = 1;
        fs_row* r = &sc->best[fw];
        r->work = work; r->fan_ix = (uint32_t)fw; r->orig_ix = s + k;
        r->lev = lv; r->dist = dist; r->comb = comb;
      }
    }
  }
  uint64_t cnt = 0;
  for (uint64_t i = 0; i < T; ++i) cnt += sc->best_set[i];
  if (!cnt) return;
  fs_row* rows = malloc(cnt * sizeof(fs_row));
  uint64_t j = 0;
  for (uint64_t i = 0; i < T; ++i) if (sc->best_set[i]) rows[j++] = sc->best[i];
  *out = rows; *n_out = cnt;
}

int fo_search(fo_index* ix, const uint32_t* tok_vec, const uint32_t* tok_str,
              const uint64_t* work_off, uint64_t n_works,
              const uint32_t* str_chars, const uint64_t* str_off, uint64_t n_str,
              fs_row* rows, uint64_t cap, uint64_t* n_rows, fs_stats* st) {
  (void)n_str;
  if (!ix || !tok_vec || !work_off || !n_rows || ix->n > 64) return FS_E_INVALID;
  uint64_t T = work_off[n_works];
  /* tables for every vector id of the batch, before the parallel region */
  for (uint64_t i = 0; i < T; ++i)
    if (slot_add(ix, tok_vec[i]) < 0) return FS_E_INVALID;
  {
    uint8_t* used = calloc(ix->n_slot, 1);
    for (uint64_t i = 0; i < T; ++i) used[slot_find(ix, tok_vec[i])] = 1;
    int64_t ns = (int64_t)ix->n_slot;
#pragma omp parallel for schedule(dynamic, 8) num_threads(ix->threads > 0 ? ix->threads : 1)
    for (int64_t s = 0; s < ns; ++s)
      if (used[s]) slot_prepare(ix, (uint64_t)s);
    free(used);
  }
  fs_row** wrows = calloc(n_works + 1, sizeof(fs_row*));
  uint64_t* wcnt = calloc(n_works + 1, sizeof(uint64_t));
  uint64_t tot_cand = 0, tot_match = 0, tot_win = 0;
#pragma omp parallel num_threads(ix->threads > 0 ? ix->threads : 1) reduction(+:tot_cand,tot_match,tot_win)
  {
    scratch sc; memset(&sc, 0, sizeof sc);
    sc.p = malloc(ix->C * sizeof(double));
    sc.keys = malloc(ix->H * sizeof(uint32_t));
    sc.seen = calloc(ix->n_win + 1, sizeof(uint32_t));
    sc.top_s = malloc((ix->cfg.nearest_n + 1) * sizeof(uint32_t));
    sc.top_d = malloc((ix->cfg.nearest_n + 1) * sizeof(double));
#pragma omp for schedule(dynamic, 4)
    for (int64_t w = 0; w < (int64_t)n_works; ++w) {
      uint64_t b = work_off[w], e = work_off[w + 1];
      uint64_t c = 0, m = 0;
      search_work(ix, (uint32_t)w, tok_vec + b, tok_str ? tok_str + b : NULL,
                  e - b, str_chars, str_off, &sc, &wrows[w], &wcnt[w], &c, &m);
      tot_cand += c; tot_match += m;
      if (e - b >= ix->n) tot_win += e - b - ix->n + 1;
    }
    free(sc.p); free(sc.keys); free(sc.seen); free(sc.top_s); free(sc.top_d);
    free(sc.cand); free(sc.lb.a); free(sc.lb.b); free(sc.lb.row);
    free(sc.best); free(sc.best_set); free(sc.slots);
  }
  uint64_t total = 0;
  for (uint64_t w = 0; w < n_works; ++w) total += wcnt[w];
  int rc = FS_OK;
  if (total > cap || !rows) rc = total ? FS_E_CAPACITY : FS_OK;
  uint64_t j = 0;
  for (uint64_t w = 0; w < n_works; ++w) {
    if (rc == FS_OK && wcnt[w]) { memcpy(rows + j, wrows[w], wcnt[w] * sizeof(fs_row)); j += wcnt[w]; }
    free(wrows[w]);
  }
  free(wrows); free(wcnt);
  *n_rows = total;
  if (st) {
    memset(st, 0, sizeof *st);
    st->windows_processed = tot_win; st->candidates = tot_cand;
    st->matches = tot_match; st->rows = total; st->path = FS_MODE_GENERAL;
  }
  return rc;
}

/* LSH keys of script window w (tests compare them with the device's). */
int fo_script_keys(fo_index* ix, uint64_t w, uint32_t* keys) {
  if (!ix || w >= ix->n_win) return FS_E_INVALID;
  int64_t slots[64];
  double* p = malloc(ix->C * sizeof(double));
  for (uint32_t k = 0; k < ix->n; ++k) slots[k] = slot_find(ix, ix->svec[w + k]);
  window_keys(ix, slots, p, keys);
  free(p);
  return FS_OK;
}
