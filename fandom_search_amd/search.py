"""Host side of the n-gram text-reuse search: the reference's `search.py`
interface on top of libfandomsearch_hip.so.

Same names and argument meaning as /root/reference/search.py so that the CLI
(`ao3.py search`) and callers of `AnnIndexSearch` can switch over:

  new_record_structure   search.py:20-37    12-column output schema
  load_markup_script     search.py:290-329  script markup -> word rows
  validate_markup_script search.py:228-288  markup linter
  AnnIndexSearch         search.py:130-226  .search(filename) -> sorted records
  write_records          search.py:331-334  csv.writer rows
  analyze                search.py:336-399  batch driver, batch + dated CSVs

What differs, and why:
  * the per-window engine.neighbours / Levenshtein / dedupe work runs on the
    GPU for a whole batch of works at once (AnnIndexSearch.search_batch); there
    is no multiprocessing.Pool and no CPU fallback
  * the inputs the reference leaves to chance are explicit: LSH hyperplanes
    (NearPy draws them unseeded, search.py:114-115) come from a seeded
    generator, the out-of-vocabulary hash (salted hash(), search.py:81-83) is
    seeded, and the directory listing (search.py:349) is sorted before the
    reference's seeded shuffle
  * spaCy / en_core_web_md are replaced by fandom_search_amd.vocab (tokenizer,
    string hashes, vector table)
"""

import csv
import datetime
import os
import time
import random
import re
import shutil

import numpy as np

from . import abi, synth, vocab as vocab_mod

new_record_structure = {
    'fields': ['FAN_WORK_FILENAME',
               'FAN_WORK_WORD_INDEX',
               'FAN_WORK_WORD',
               'FAN_WORK_ORTH_ID',
               'ORIGINAL_SCRIPT_WORD_INDEX',
               'ORIGINAL_SCRIPT_WORD',
               'ORIGINAL_SCRIPT_ORTH_ID',
               'ORIGINAL_SCRIPT_CHARACTER',
               'ORIGINAL_SCRIPT_SCENE',
               'BEST_MATCH_DISTANCE',
               'BEST_LEVENSHTEIN_DISTANCE',
               'BEST_COMBINED_DISTANCE',
               ],
    'types': [str, int, str, int, int, str,
              int, str, int, float, int, float
              ]
}

_VOCAB = None
SHUFFLE_SEED = 4815162342        # search.py:354


def get_vocab():
    """The process-wide vocabulary (stands where get_spacy_model stands,
    search.py:40-45).  FANDOM_SEARCH_VECTORS=<file.npz with 'words' and 'vectors'>
    selects the vector table (e.g. an export of en_core_web_md).  The synthetic
    vocabulary of SURVEY.md 8(d) (8192 pseudo-words, random vectors: no semantic
    similarity) must be asked for: FANDOM_SEARCH_SYNTHETIC_VOCAB=1 or `ao3.py search
    --synthetic-vocab`; there is no silent default."""
    global _VOCAB
    if _VOCAB is None:
        path = os.environ.get("FANDOM_SEARCH_VECTORS")
        if path:
            data = np.load(path, allow_pickle=False)
            # 'words' + 'vectors' (+ 'rows': the row of each word when words share rows, as
            # tools/export_spacy_vectors.py writes for spaCy's key2row)
            _VOCAB = vocab_mod.Vocab([str(w) for w in data["words"]], data["vectors"],
                                     rows=data["rows"] if "rows" in data.files else None)
        elif os.environ.get("FANDOM_SEARCH_SYNTHETIC_VOCAB", "") not in ("", "0"):
            _VOCAB = vocab_mod.Vocab(synth.vocab_words(), synth.embedding())
        else:
            raise RuntimeError(
                "no vector table: set FANDOM_SEARCH_VECTORS=<file.npz with 'words' and "
                "'vectors'> (the reference uses spaCy's en_core_web_md), or ask for the "
                "synthetic benchmark vocabulary with --synthetic-vocab / "
                "FANDOM_SEARCH_SYNTHETIC_VOCAB=1 (random vectors: results carry no "
                "semantic similarity)")
    return _VOCAB


def set_vocab(v):
    global _VOCAB
    _VOCAB = v


def default_normals(window_size, number_of_hashes, hash_dimensions, dim):
    return synth.lsh_normals(window_size, number_of_hashes, hash_dimensions,
                             dim)


# ---------------------------------------------------------------------------
# script markup
# ---------------------------------------------------------------------------

_LINE_REX = re.compile('LINE<<(?P<line>[^>]*)>>')
_SCENE_REX = re.compile('SCENE_NUMBER<<(?P<scene>[^>]*)>>')
_CHAR_REX = re.compile('CHARACTER_NAME<<(?P<character>[^>]*)>>')
_EXPECTED_TAGS = frozenset(('LINE', 'DIRECTION', 'SCENE_NUMBER',
                            'SCENE_DESCRIPTION', 'CHARACTER_NAME'))


def load_markup_script(filename):
    """Rows [LOWERCASE, SPACY_ORTH_ID, SCENE, CHARACTER] under a header row.

    Per text line, first match wins in the order scene / character / line
    (search.py:304-321).  A scene label keeps only its digits; the first label
    without digits switches, for the rest of the file, to the running count of
    scene tags (search.py:309-317)."""
    rows = [['LOWERCASE', 'SPACY_ORTH_ID', 'SCENE', 'CHARACTER']]
    scene = None
    scene_tags = 0
    count_scenes = False
    character = None
    with open(filename, encoding='utf-8') as ip:
        for text in ip:
            m = _SCENE_REX.search(text)
            if m:
                scene_tags += 1
                digits = ''.join(ch for ch in m.group('scene') if ch.isdigit())
                try:
                    # (isdigit() also accepts characters int() refuses, e.g. superscripts:
                    # the reference catches the ValueError, search.py:309-314)
                    scene = int(digits)
                except ValueError:
                    count_scenes = True
                    print("Error in Scene markup: {}".format(text))
                if count_scenes:
                    scene = scene_tags
                continue
            m = _CHAR_REX.search(text)
            if m:
                character = m.group('character')
                continue
            m = _LINE_REX.search(text)
            if m:
                for tok in vocab_mod.tokenize(m.group('line')):
                    low = tok.lower()
                    rows.append([low, vocab_mod.hash_string(low), scene,
                                 character])
    return rows


def validate_markup_script(filename, interactive=False):
    """Markup linter (search.py:228-285): unbalanced << or >> delimiters and
    unknown tag labels, each reported with its line number."""
    with open(filename, encoding='utf-8') as ip:
        script = ip.read()

    def line_of(pos):
        return script[:pos + 1].count('\n') + 1

    print('Checking script for markup errors.')
    print()
    errs = False
    checks = (('Unbalanced left tag delimiters:', re.compile('<<[^>]*<<')),
              ('Unbalanced right tag delimiters:', re.compile('>>[^<]*>>')))
    for title, rex in checks:
        found = list(rex.finditer(script))
        if found:
            print(title)
            for m in found:
                print('  On line {}'.format(line_of(m.start())))
                print('    {}'.format(m.group().strip()))
            errs = True
            print()

    tag_rex = re.compile(r'>>\s*([^<]*)\s*<<')
    bad = [m for m in tag_rex.finditer(script)
           if m.group(1).strip() not in _EXPECTED_TAGS]
    if bad:
        print('Unexpected tag labels:')
        for m in bad:
            print('  On line {}'.format(line_of(m.start(1))))
            print('    {}'.format(m.group(1).strip()))
        errs = True
        print()

    if not errs:
        print('No markup errors found.')
        return True
    if interactive:
        print('Errors were found in the script markup. Do you want to '
              'continue? (Default is no.)')
        print()
        r = ''
        while r.lower() not in ('y', 'yes', 'n', 'no'):
            r = input('Enter y for yes or n for no: ')
            if not r.strip():
                r = 'n'
        return r.lower() in ('y', 'yes')
    return False


def validate_cmd(args):
    return validate_markup_script(args.script)


# ---------------------------------------------------------------------------
# search
# ---------------------------------------------------------------------------

def read_work_tokens(filename):
    """Token texts of a fan work (search.py:164-166: read, chunked parse,
    whitespace tokens dropped)."""
    with open(filename, encoding='utf8') as fan_file:
        fan = fan_file.read()
    return [t for ch in vocab_mod.chunk_text(fan)
            for t in vocab_mod.tokenize(ch)]


def _factorize(texts):
    """(codes uint32, uniques list) of a list of strings, first-appearance order."""
    if len(texts) == 0:
        return np.zeros(0, np.uint32), []
    try:
        import pandas as pd
        codes, uniques = pd.factorize(np.asarray(texts, dtype=object), sort=False)
    except ImportError:
        uniques, codes = np.unique(np.asarray(texts, dtype=object), return_inverse=True)
    return np.asarray(codes, dtype=np.uint32), [str(u) for u in uniques]


NEW_STRING = 1 << 31      # tokenize_files: codes from here on index the chunk's list of new strings


def tokenize_files(filenames, vocab=None):
    """What a pool worker does with a chunk of works (the reference's workers do the same at
    search.py:164-166: read, chunked parse, whitespace tokens dropped): token counts per
    work and the chunk's tokens as string ids of the vocabulary the worker inherited when
    it was forked; a string that vocabulary did not hold yet travels back as text, once per
    chunk, and its tokens carry NEW_STRING + its place in that list.  The compact form is
    what makes the pool pay: 4 bytes per token instead of one Python string."""
    lens = np.zeros(len(filenames), dtype=np.int64)
    flat = []
    for i, f in enumerate(filenames):
        toks = read_work_tokens(f)
        lens[i] = len(toks)
        flat += toks
    codes, uniques = _factorize(flat)
    v = vocab if vocab is not None else _VOCAB
    known = v._string_id if v is not None else {}
    new = []
    usid = np.zeros(len(uniques), dtype=np.uint32)
    for k, t in enumerate(uniques):
        sid = known.get(t)
        if sid is None:
            sid = NEW_STRING + len(new)
            new.append(t)
        usid[k] = sid
    return lens, (usid[codes] if len(codes) else np.zeros(0, np.uint32)), new


class TokenPool(object):
    """Worker processes that read and tokenise fan works while the GPU searches and the
    parent writes records (/root/reference/search.py:381-385 runs its whole search in such
    a pool; here only the host text work is left for it).  Must be created BEFORE the
    process touches the GPU: the workers are forked.  `start(filenames)` queues a list of
    works in chunks of 31 (the reference's chunksize), `get(filenames)` returns
    (lens, codes, uniques) of exactly that list, waiting for a queued job or doing the
    work here when nothing was queued."""

    CHUNK = 31

    def __init__(self, processes):
        import multiprocessing
        self.processes = int(processes)
        self.pool = multiprocessing.get_context("fork").Pool(self.processes) if self.processes > 1 else None
        self.pending = {}
        self.native = None                     # textenc.TextEncoder: reading and tokenising on native threads
        self.writes = []                       # [job or None, function, arguments] of queued writes
        self._pids = sorted(p.pid for p in self.pool._pool) if self.pool is not None else None

    def start(self, filenames):
        key = tuple(filenames)
        if self.native is not None:
            self.native.start(key)
            return
        if self.pool is None or not key or key in self.pending:
            return
        chunks = [list(key[i:i + self.CHUNK]) for i in range(0, len(key), self.CHUNK)]
        self.pending[key] = self.pool.map_async(tokenize_files, chunks, chunksize=1)

    def get(self, filenames):
        if self.native is not None:
            lens, sids = self.native.encode_files(list(filenames))
            return [(lens, sids, [])]
        job = self.pending.pop(tuple(filenames), None)
        # (jobs queued for a list that was never asked for -- share() and the split of a batch
        # disagreeing -- are dropped, not kept for the life of the pool: only the next cluster's
        # job is ever ahead of the one asked for)
        while len(self.pending) > 2:
            self.pending.pop(next(iter(self.pending)))
        if job is not None and self._workers_alive():
            return job.get()
        return [tokenize_files(list(filenames))]

    def write_async(self, fn, args):
        """Queue fn(*args) -- a batch file to be written -- on a worker.  The arguments are
        kept until finish_writes() has seen the job complete: a pool that loses a worker is
        given up, and its queued jobs would never be done."""
        job = self.pool.apply_async(fn, args) if self.pool is not None and self._workers_alive() else None
        self.writes.append([job, fn, args])

    def finish_writes(self):
        """Wait for every queued write (raises what a writer raised).  No unbounded wait: a job
        is polled with the workers' liveness checked in between, and what a lost pool left
        undone -- including a file a killed writer left half written -- is redone here."""
        import multiprocessing
        writes, self.writes = self.writes, []
        for job, fn, args in writes:
            while job is not None and not job.ready() and self.pool is not None and self._workers_alive():
                try:
                    job.get(timeout=0.25)
                except multiprocessing.TimeoutError:
                    pass
                except Exception:
                    break                       # (ready now: raised again below)
            if job is not None and job.ready():
                job.get()
            else:
                fn(*args)

    def _workers_alive(self):
        """multiprocessing.Pool re-forks a worker that died (out of memory, a signal) -- from
        a parent that holds the HIP runtime and its threads by then, and with a vocabulary
        newer than the replacement's protocol assumes.  A pool that has lost a worker is shut
        down instead and the text work done in the parent from there on."""
        if self.pool is None:
            return False
        pids = sorted(p.pid for p in self.pool._pool)
        if self._pids is None:
            self._pids = pids
        if pids != self._pids or any(not p.is_alive() for p in self.pool._pool):
            import sys
            print("warning: a tokeniser process was lost; tokenising in the parent from here on",
                  file=sys.stderr)
            self._abandon()
            return False
        return True

    def _abandon(self):
        """Give a pool up that has lost a worker: Pool.terminate() can wait for ever for a
        queue lock the dead worker held, so the remaining workers are killed and the pool's
        own finaliser is cancelled (its helper threads are daemons)."""
        pool, self.pool = self.pool, None
        self.pending.clear()
        if pool is None:
            return
        try:
            pool._terminate.cancel()
            # (its maintenance thread stops forking replacements: it looks at its own state)
            pool._state = "TERMINATE"
            pool._worker_handler._state = "TERMINATE"
            pool._change_notifier.put(None)
        except Exception:
            pass
        import time
        time.sleep(0.05)                       # (the thread has seen its state before the workers go)
        for p in list(pool._pool):
            try:
                p.kill()
            except Exception:
                pass

    def close(self):
        if self.pool is not None:
            if any(not p.is_alive() for p in self.pool._pool):
                self._abandon()
                return
            self.pool.terminate()
            self.pool.join()
            self.pool = None


def usable_cpus():
    """Cores this process may keep busy: its affinity mask, cut down to the CPU quota of its
    control group (cgroup v2 cpu.max, v1 cpu.cfs_quota_us) -- a container on a 256-core host
    with a quota of 16 has 16, whatever the mask says; workers beyond it only take turns."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, period = fh.read().split()[:2]
            if q != "max" and float(period) > 0:
                quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, period = float(fq.read()), float(fp.read())
                if q > 0 and period > 0:
                    quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        cores = min(cores, max(1, int(quota + 0.999)))
    return max(1, cores)


def default_workers(world=1):
    """Batch-file writer (and, without the native text front end, tokeniser) processes per
    rank: FANDOM_SEARCH_WORKERS, else up to 16 of this rank's share of the cores the process
    may keep busy (0 or 1: in the parent) -- half of that share where the native encoder's
    threads (default_text_threads) take the other half."""
    env = os.environ.get("FANDOM_SEARCH_WORKERS")
    if env is not None:
        return max(0, int(env))
    share = usable_cpus() // max(1, world)
    from . import csvw, textenc
    if textenc.enabled():
        if csvw.enabled() and world == 1:
            return 0                        # text and batch files are both native: nothing left to fork for
        share = (share + 1) // 2
    return max(1, min(16, share))


def default_text_threads(world=1):
    """Threads of one native text encoder (a rank runs FANDOM_SEARCH_TEXT_HANDLES of them, default
    two, on consecutive batches): FANDOM_SEARCH_TEXT_THREADS, else what FANDOM_SEARCH_WORKERS
    says, else this rank's share of the cores -- less the half default_workers' forked writers
    take where they run -- divided among the encoders."""
    env = os.environ.get("FANDOM_SEARCH_TEXT_THREADS")
    if env is not None:
        return max(1, int(env))
    env = os.environ.get("FANDOM_SEARCH_WORKERS")
    if env is not None and int(env) > 0:
        return int(env)
    from . import csvw
    share = usable_cpus() // max(1, world)
    if not (csvw.enabled() and world == 1):
        share //= 2                         # (the forked writers take the other half)
    # (two native encoders work on consecutive batches at the same time, textenc.TextEncoder:
    # the threads are per encoder)
    share //= max(1, int(os.environ.get("FANDOM_SEARCH_TEXT_HANDLES", "2")))
    return max(1, min(16, share))


class AnnIndexSearch(object):
    def __init__(self, original_script_filename, window_size,
                 number_of_hashes, hash_dimensions, distance_threshold,
                 vocab=None, normals=None, device=0, mode=abi.FS_MODE_AUTO,
                 unique_filter=None):
        from .engine import ScriptIndex   # needs the HIP library: fail loudly

        orig_csv = load_markup_script(original_script_filename)[1:]
        self.word_index = tuple(range(len(orig_csv)))
        self.word_lowercase = tuple(r[0] for r in orig_csv)
        self.orth_id = tuple(r[1] for r in orig_csv)
        self.scene = tuple(r[2] for r in orig_csv)
        self.character = tuple(r[3] for r in orig_csv)

        self.window_size = window_size
        self.distance_threshold = distance_threshold
        self.vocab = vocab or get_vocab()
        if normals is None:
            normals = default_normals(window_size, number_of_hashes,
                                      hash_dimensions, self.vocab.dim)
        _sids, script_vec = self.vocab.encode(list(self.word_lowercase))
        cfg = abi.make_config(window_size=window_size,
                              number_of_hashes=number_of_hashes,
                              hash_dimensions=hash_dimensions,
                              distance_threshold=distance_threshold,
                              emb_dim=self.vocab.dim, device=device, mode=mode,
                              unique_filter=unique_filter)
        self.engine = ScriptIndex(script_vec, self.word_lowercase,
                                  self.vocab.vectors, normals, cfg=cfg)
        self.last_stats = None
        self.last_oov_rate = None
        self.token_pool = None      # analyze() lends its TokenPool
        self.reset_stats()

    def prefetch(self, filenames):
        """Start reading and tokenising works that a later search_rows / search_shard call
        will ask for (no-op without a pool)."""
        if self.token_pool is not None:
            self.token_pool.start(filenames)

    def _encode_files(self, filenames):
        """(string ids, vector ids, work offsets) of the works: tokenised by the pool (or
        here) against the vocabulary as it was when the pool was forked; strings it has not
        seen get their ids here, in work order (so the ids do not depend on the pool)."""
        v = self.vocab
        # (the workers hold the process-wide vocabulary; an index built on another one
        # tokenises here)
        made = None          # (tokens, vector ids, OOV tokens, ids equal) where the encoder's thread made them
        if self.token_pool is not None and v is _VOCAB:
            parts = self.token_pool.get(filenames)
            if self.token_pool.native is not None:
                made = self.token_pool.native.last_vec
        else:
            from . import textenc
            if textenc.enabled():
                if getattr(self, "_text", None) is None:
                    self._text = textenc.TextEncoder(v)
                lens, sids = self._text.encode_files(list(filenames))
                parts = [(lens, sids, [])]
                made = self._text.last_vec
            else:
                parts = [tokenize_files(list(filenames), v)]
        self._made = None
        if made is not None and len(parts) == 1 and parts[0][1] is made[0] and not parts[0][2]:
            lens = parts[0][0]
            off = np.zeros(len(lens) + 1, dtype=np.uint64)
            off[1:] = np.cumsum(lens, dtype=np.uint64)
            self._made = made[1:]                # (_corpus_of_ids: for exactly this vector id array)
            return made[0], made[1], off
        lens = np.concatenate([p[0] for p in parts]) if parts else np.zeros(0, np.int64)
        off = np.zeros(len(lens) + 1, dtype=np.uint64)
        off[1:] = np.cumsum(lens, dtype=np.uint64)
        strs = []
        for _, sids, new in parts:
            if new:
                # (a worker's vocabulary is the parent's at fork time: what it calls new may
                # have been added here since)
                ids = np.fromiter((v.string_id(t) for t in new), dtype=np.uint32, count=len(new))
                fresh = sids >= NEW_STRING
                sids = sids.copy()
                sids[fresh] = ids[sids[fresh] - NEW_STRING]
            strs.append(sids)
        tok_str = np.concatenate(strs) if strs else np.zeros(0, np.uint32)
        tok_vec = v.vec_ids()[tok_str] if len(tok_str) else np.zeros(0, np.uint32)
        return tok_str, tok_vec, off

    def reset_stats(self):
        self._windows_processed = 0

    @property
    def windows_processed(self):
        return self._windows_processed

    def search(self, filename):
        """Sorted 12-field records of one fan work (search.py:163-226)."""
        return self.search_batch([filename])

    def search_batch(self, filenames):
        """Records of many works from one pass over the batch, in the order
        `[r for r_set in pool.map(...) for r in r_set]` produces
        (search.py:382-386): works in list order, words ascending."""
        rows, words = self.search_rows(filenames)
        return self.records(filenames, rows, words)

    def search_rows(self, filenames):
        """(fs_row array with work indices into `filenames`, fan word text per
        row) -- the form that travels between ranks (fandom_search_amd.dist)."""
        import time
        t0 = time.perf_counter()
        enc = self._encode_files(filenames)
        t1 = time.perf_counter()
        out = self._search_encoded(*enc)
        t = self.__dict__.setdefault("host_times", {"encode": 0.0, "search": 0.0})
        t["encode"] += t1 - t0
        t["search"] += time.perf_counter() - t1
        return out

    def _corpus_of_ids(self, tok_str, tok_vec, off):
        """(string ids, vector ids, work offsets, device corpus) of encoded works."""
        v = self.vocab
        chars, coff = v.string_table()
        made = getattr(self, "_made", None)
        if made is not None and made[0] is tok_vec:
            self.last_oov_rate, same = made[1] / float(len(tok_vec)), made[2]
        else:
            self.last_oov_rate = float(np.count_nonzero(tok_vec & np.uint32(abi.FS_OOV_FLAG))) / len(tok_vec) \
                if len(tok_vec) else 0.0
            same = bool(np.array_equal(tok_str, tok_vec))
        # one device corpus serves batch after batch while the string table stays as it is
        # (fs_corpus_update_begin/_end: the ids are replaced, the tables built once per string
        # table -- Levenshtein per n-gram, the batch table of k_scan_rows -- are kept)
        kept = getattr(self, "_corpus", None)
        if kept is not None and kept[0] == len(v.strings) and kept[1]._h:
            corpus = kept[1]
            corpus.update_begin(tok_vec, off, tok_str=None if same else tok_str)
            corpus.update_end()
        else:
            if kept is not None:
                kept[1].close()
            corpus = self.engine.corpus(tok_vec, off, chars, coff,
                                        tok_str=None if same else tok_str)
            self._corpus = (len(v.strings), corpus)
        return tok_str, tok_vec, off, corpus

    def search_tokens(self, texts):
        """Records of works given as lists of token texts."""
        off = np.zeros(len(texts) + 1, dtype=np.uint64)
        if texts:
            off[1:] = np.cumsum([len(toks) for toks in texts], dtype=np.uint64)
        # one pass over the whole batch: the vocabulary is consulted once per distinct
        # string of the batch, not once per work
        tok_str, tok_vec = self.vocab.encode([t for toks in texts for t in toks])
        return self._search_encoded(tok_str, tok_vec, off)

    def _search_encoded(self, tok_str, tok_vec, off):
        import time
        v = self.vocab
        ht = self.__dict__.setdefault("host_times", {"encode": 0.0, "search": 0.0})
        t0 = time.perf_counter()
        tok_str, tok_vec, off, corpus = self._corpus_of_ids(tok_str, tok_vec, off)
        t1 = time.perf_counter()
        rows, st = self.engine.search(corpus, reuse=True)
        rows = rows.copy()                        # (the engine's buffer is written by the next batch)
        t2 = time.perf_counter()
        self.last_stats = st
        self._windows_processed += int(st.windows_processed)
        pos = off[rows['work']].astype(np.int64) + rows['fan_ix'].astype(np.int64)
        # the fan words of the records: as string ids (what the native batch writer takes, csvw.py)
        # and, unless the caller has said it does not need them (want_words), as text
        self.last_word_sids = tok_str[pos]
        words = list(map(v.strings.__getitem__, self.last_word_sids.tolist())) \
            if getattr(self, "want_words", True) else None
        t3 = time.perf_counter()
        for k, dt in (("corpus to the GPU", t1 - t0), ("fs_search_corpus", t2 - t1), ("fan words of the records", t3 - t2)):
            ht[k] = ht.get(k, 0.0) + dt
        return rows, words

    def search_shard(self, filenames):
        """This rank's share of a batch for fandom_search_amd.dist.search_sharded: the
        records stay in HBM (wire records of the exact pipeline: 8 bytes, 16 for scripts of
        2^18 tokens and more; fs_row from the LSH pipeline) behind a 32-byte header that
        the search itself fills with their number; the fan words are looked up from the
        token positions of a host copy taken beside the gather."""
        import torch
        from . import _lib
        from .dist import HDR, Shard
        tok_str, tok_vec, off, corpus = self._corpus_of_ids(*self._encode_files(filenames))
        exact = self.engine.info["path"] == abi.FS_MODE_EXACT and \
            not bool((tok_vec & np.uint32(abi.FS_OOV_FLAG)).any())
        packed = False
        if exact:
            packed = 8 if len(self.word_lowercase) < abi.PACKED8_MAX_SCRIPT else 16
        rec = packed if packed else 32
        cap = max(1024, len(tok_vec) // 16)
        while True:
            buf = torch.zeros(HDR + cap * rec, dtype=torch.uint8, device="cuda")
            from .engine import torch_ready
            torch_ready()                       # (torch's fill is done before the search writes header and records)
            try:
                n, st = self.engine.search_end(self.engine.search_begin(
                    corpus, buf.data_ptr(), cap, packed=packed, header=True))
                break
            except _lib.FsError as e:
                if e.code != abi.FS_E_CAPACITY:
                    raise
                cap = int(e.required) + 16
        self.last_stats = st
        self._windows_processed += int(st.windows_processed)
        host = buf[HDR:HDR + n * rec].cpu().numpy()
        if rec == 8:
            pos = host.view(np.uint32)[0::2].astype(np.int64)         # token position
        else:
            r = host.view(np.uint32).reshape(n, rec // 4)
            pos = off[r[:, 0]].astype(np.int64) + r[:, 1].astype(np.int64)
        words = list(map(self.vocab.strings.__getitem__, tok_str[pos].tolist()))
        return Shard(buf, rec, n, off, words)

    def records(self, filenames, rows, words):
        """Join the numeric rows with file name, fan word / orth id and the
        script columns into the 12-field records of search.py:203-217."""
        return join_records(filenames, rows, words, self.word_lowercase,
                            self.orth_id, self.character, self.scene)


def join_records(filenames, rows, words, word_lowercase, orth_id, character,
                 scene):
    out = []
    for w, f, o, l, d, c, fw in zip(rows['work'].tolist(), rows['fan_ix'].tolist(),
                                    rows['orig_ix'].tolist(), rows['lev'].tolist(),
                                    rows['dist'].tolist(), rows['comb'].tolist(),
                                    words):
        out.append([filenames[w], f, fw, vocab_mod.hash_string(fw),
                    o, word_lowercase[o], orth_id[o], character[o], scene[o],
                    d, l, c])
    return out


def write_records(records, filename):
    with open(filename, 'w', encoding='utf-8') as out:
        wr = csv.writer(out)
        wr.writerows(records)


_script_columns_cache = {}


def _script_columns(script_filename):
    """(word_lowercase, orth_id, character, scene) of a marked-up script, as
    AnnIndexSearch.__init__ keeps them; cached per process."""
    cols = _script_columns_cache.get(script_filename)
    if cols is None:
        orig_csv = load_markup_script(script_filename)[1:]
        cols = (tuple(r[0] for r in orig_csv), tuple(r[1] for r in orig_csv),
                tuple(r[3] for r in orig_csv), tuple(r[2] for r in orig_csv))
        _script_columns_cache[script_filename] = cols
    return cols


def _write_batch(script_filename, out_name, filenames, row_bytes, words):
    """One batch CSV, in a worker of the token pool: join the numeric rows with the script
    columns and write them (join_records + write_records), while the parent goes on to the
    next cluster.  The rows travel as bytes."""
    rows = np.frombuffer(row_bytes, dtype=abi.ROW_DTYPE)
    word_lowercase, orth_id, character, scene = _script_columns(script_filename)
    write_records(join_records(filenames, rows, words, word_lowercase, orth_id, character, scene),
                  out_name)
    return len(rows)


def listing_order():
    """'sorted' (default) or 'os' (FANDOM_SEARCH_LISTING, `ao3.py search --listing`)."""
    order = os.environ.get("FANDOM_SEARCH_LISTING", "sorted")
    if order not in ("sorted", "os"):
        raise ValueError("FANDOM_SEARCH_LISTING must be 'sorted' or 'os', not %r" % (order,))
    return order


def list_fan_works(fan_work_directory, skip_works=0, num_works=-1, order=None):
    """Work list of analyze (search.py:345-358): directory listing, the reference's
    seeded shuffle, then the -s / -n window.  The reference shuffles os.listdir()'s list as
    it comes (search.py:349,354-355) -- an order that belongs to the file system, not to
    the directory's contents -- so the default here sorts the names first and a run can be
    repeated anywhere; order='os' (FANDOM_SEARCH_LISTING=os, --listing os) takes the
    listing as the reference does: on the same directory of the same file system the -s /
    -n window then selects the very works the reference's run would."""
    subsample_start = 0 if skip_works < 0 else skip_works
    subsample_end = (None if num_works < 0 else num_works + subsample_start)
    fan_works = os.listdir(fan_work_directory)
    if (order or listing_order()) == "sorted":
        fan_works = sorted(fan_works)
    fan_works = [os.path.join(fan_work_directory, f) for f in fan_works]
    random.seed(SHUFFLE_SEED)
    random.shuffle(fan_works)
    return fan_works[subsample_start:subsample_end]


def unused_result_name(filename_base):
    """match-<n>gram-YYYYMMDD.csv, then -YYYYMMDD-1.csv, ... (search.py:390-396)."""
    i = 0
    today_str = '-{:%Y%m%d}.csv'.format(datetime.date.today())
    name_check = filename_base.format(today_str)
    while os.path.exists(name_check):
        i += 1
        today_str = '-{:%Y%m%d}-{}.csv'.format(datetime.date.today(), i)
        name_check = filename_base.format(today_str)
    return name_check


_startup = []


def _startup_lap(what):
    """FANDOM_SEARCH_TIMING: seconds since ao3.py started (FANDOM_SEARCH_T0), at the named point."""
    t0 = os.environ.get("FANDOM_SEARCH_T0")
    if os.environ.get("FANDOM_SEARCH_TIMING") and t0:
        _startup.append((what, time.time() - float(t0)))


def analyze(args,
            window_size=6,
            number_of_hashes=15,
            hash_dimensions=14,
            distance_threshold=0.1,
            chunk_size=500,
            searcher=None):
    """`ao3.py search` (search.py:336-399): per cluster of `chunk_size` works a
    header-less batch CSV, at the end a dated CSV with header, both in the
    current directory.

    Launched under torch.distributed.run (WORLD_SIZE > 1) every cluster is
    split over the ranks, one GPU each, and rank 0 writes the files: the bytes
    are the same for any number of GPUs.  `searcher` (tests) replaces the
    AnnIndexSearch instance; it needs search_rows() and the script columns."""
    from . import dist
    # the tokeniser processes are forked before anything touches the GPU (the process group
    # of a multi-GPU run does); they read and tokenise cluster i + 1 while the GPU searches
    # cluster i and the parent writes its records (the reference's Pool(4) does its whole
    # search in the workers, search.py:381-385)
    pool = None
    _startup_lap("interpreter, imports, arguments")
    if searcher is None:
        n_workers = default_workers(dist.env_world()[2])
        if n_workers <= 1 and dist.env_world()[2] == 1 and os.environ.get("FANDOM_SEARCH_EARLY_HIP", "1") != "0":
            # nobody is forked from here on (text and batch files are native): the HIP runtime's
            # start-up -- a quarter of a second, most of what `script index on the GPU` is -- runs on
            # a thread beside the vector table, the encoder's teaching and the script's parse instead of behind them
            import threading
            from . import _lib
            L = _lib.load()                       # (on this thread: one loader)

            def warm():
                try:
                    import ctypes as C
                    p = C.c_void_p()
                    if L.fs_host_alloc(4096, C.byref(p)) == 0:
                        L.fs_host_free(p)
                except Exception:
                    pass                          # (whatever it is, fs_index_create says it again)
            threading.Thread(target=warm, daemon=True).start()
        get_vocab()               # (host only) the workers inherit the string table ...
        from . import tokenizer   # noqa: F401  ... and the compiled tokenizer rules
        _startup_lap("vector table")
        pool = TokenPool(n_workers)
        _startup_lap("fork the token pool")
        from . import textenc
        if textenc.enabled():
            # reading and tokenising on native threads of this process (fs_textenc_*); the
            # forked workers then only write the batch files
            pool.native = textenc.TextEncoder(get_vocab(), default_text_threads(dist.env_world()[2]))
            _startup_lap("native text encoder")
    try:
        return _analyze(args, window_size, number_of_hashes, hash_dimensions, distance_threshold,
                        chunk_size, searcher, pool)
    finally:
        if pool is not None:
            pool.close()


def _analyze(args, window_size, number_of_hashes, hash_dimensions, distance_threshold,
             chunk_size, searcher, pool):
    from . import dist
    rank, local_rank, world = dist.init_from_env()
    fan_works = list_fan_works(args.fan_works, args.skip_works, args.num_works)
    if world > 1 and listing_order() == "os":
        # (the file system's order is one process's view: every rank works on rank 0's list)
        import torch.distributed as tdist
        box = [fan_works]
        tdist.broadcast_object_list(box, src=0)
        fan_works = box[0]
    window_size = getattr(args, 'window_size', None) or window_size
    device = getattr(args, 'device', 0) or 0
    if world > 1:
        import torch.distributed as tdist
        device = local_rank if tdist.get_backend() == "nccl" else 0   # gloo: rehearsal on GPU 0

    fan_clusters = [fan_works[i:i + chunk_size]
                    for i in range(0, len(fan_works), chunk_size)]

    filename_base = 'match-{}gram{{}}'.format(window_size)
    batch_filename = filename_base.format('-batch-{}.csv')

    n_batches = 0

    def share(cluster):
        """The works of `cluster` this rank reads (all of them without a launcher)."""
        if world == 1:
            return cluster
        b = dist.split_contiguous([os.path.getsize(f) for f in cluster], world)
        return cluster[b[rank]:b[rank + 1]]

    if pool is not None and fan_clusters:
        pool.start(share(fan_clusters[0]))          # beside the index build
    ann_index = searcher or AnnIndexSearch(args.script,
                                           window_size,
                                           number_of_hashes,
                                           hash_dimensions,
                                           distance_threshold,
                                           device=device)

    if pool is not None:
        ann_index.token_pool = pool
    import time
    timing = {} if os.environ.get("FANDOM_SEARCH_TIMING") else None
    t_last = [time.perf_counter()]

    def lap(what):
        if timing is not None:
            now = time.perf_counter()
            timing[what] = timing.get(what, 0.0) + now - t_last[0]
            t_last[0] = now

    # batch files by the native writer (fs_csvw_*: join and csv bytes in one call per batch, on a
    # thread of this process) where the records come from this process' own search
    native_csv = None
    if searcher is None and world == 1:
        from . import csvw
        if csvw.enabled():
            native_csv = csvw.CsvWriter(ann_index.word_lowercase, ann_index.orth_id, ann_index.character,
                                        ann_index.scene, ann_index.vocab.strings)
            ann_index.want_words = False
    lap("index")
    _startup_lap("script index on the GPU (library load, HIP start-up, fs_index_create)")
    failure = None
    t_start = time.time()
    for i, fan_cluster in enumerate(fan_clusters):
        if pool is not None and i + 1 < len(fan_clusters):
            pool.start(share(fan_clusters[i + 1]))
            if pool.native is not None and i + 2 < len(fan_clusters):
                pool.start(share(fan_clusters[i + 2]))      # (the native encoder takes two batches at a time)
        if rank == 0:
            print('Processing cluster {} ({}-{})'.format(i,
                                                         chunk_size * i,
                                                         chunk_size * (i + 1)))
        # the rank that receives, joins and writes this batch: they take turns (every pair of
        # GPUs has its own xGMI link; FANDOM_SEARCH_GATHER_ROOT=0: always rank 0)
        root = i % world if os.environ.get("FANDOM_SEARCH_GATHER_ROOT", "rotate") != "0" else 0
        if world > 1:
            weights = [os.path.getsize(f) for f in fan_cluster]
            rows, words = dist.search_sharded(fan_cluster, weights, ann_index, root=root)
        else:
            rows, words = ann_index.search_rows(fan_cluster)
        lap("tokens + search")
        oov = getattr(ann_index, 'last_oov_rate', None)
        if oov is not None and oov > 0.2:
            import sys
            print('warning: {}{:.0%} of the fan tokens of this batch have no row in the vector '
                  'table (out-of-vocabulary 3-hot vectors, search.py:79-83); check '
                  'FANDOM_SEARCH_VECTORS'.format('rank {}: '.format(rank) if world > 1 else '', oov),
                  file=sys.stderr)
        n_batches = i + 1
        if rank != root or failure is not None:
            continue
        try:
            if native_csv is not None:
                native_csv.write_async(batch_filename.format(i), fan_cluster, rows, ann_index.last_word_sids)
                lap("hand batch to a writer")
                continue
            if pool is not None and pool.pool is not None and searcher is None:
                # the batch file is written by a worker of the token pool (they are forked, and
                # parse the script's columns themselves) while this process searches the next
                # cluster; the reference writes every batch file before the next pool.map
                # (search.py:386-388), the bytes are the same
                pool.write_async(
                    _write_batch, (args.script, batch_filename.format(i), list(fan_cluster),
                                   np.ascontiguousarray(rows, dtype=abi.ROW_DTYPE).tobytes(), words))
                lap("hand batch to a writer")
                continue
            records = join_records(fan_cluster, rows, words,
                                   ann_index.word_lowercase, ann_index.orth_id,
                                   ann_index.character, ann_index.scene)
            lap("join records")
            write_records(records, batch_filename.format(i))
            lap("write batch csv")
        except Exception as e:
            # with other ranks around, a failure of the host work is told to them at the end (they
            # are inside the next batch's collectives by now); alone, it is raised here
            if world == 1:
                raise
            failure = e
    try:
        if native_csv is not None:
            ann_index.want_words = True
            native_csv.finish()                     # (raises what a write raised)
            native_csv.close()
        if pool is not None:
            pool.finish_writes()                    # (raises what a writer raised)
    except Exception as e:                          # told to the other ranks below, then raised
        failure = failure or e
    lap("wait for the writers")
    if world > 1:
        # every rank's batch files are on disk -- or one of them failed, and then every rank
        # raises instead of waiting in a barrier for a rank that is gone; rank 0 also makes
        # sure the files the other ranks wrote are the ones it is about to concatenate
        mine = [i for i in range(n_batches)
                if (i % world if os.environ.get("FANDOM_SEARCH_GATHER_ROOT", "rotate") != "0" else 0) == rank]
        dist.collect_batch_files(batch_filename, n_batches, mine, failure, t_start)
    elif failure is not None:
        raise failure

    if pool is not None:
        ann_index.token_pool = None
    name = None
    if rank == 0:
        # the dated file = header row + the records of every batch file, in order
        # (search.py:386-399 accumulates the same rows and writes them with the same writer)
        name = unused_result_name(filename_base)
        write_records([new_record_structure['fields']], name)
        with open(name, 'ab') as out:
            for i in range(n_batches):
                with open(batch_filename.format(i), 'rb') as part:
                    shutil.copyfileobj(part, out)
    lap("write dated csv")
    if timing is not None:
        import sys
        if _startup:
            print("since process start: " + ", ".join("%s %.3f s" % kv for kv in _startup), file=sys.stderr)
        print("analyze: " + ", ".join("%s %.3f s" % kv for kv in timing.items()), file=sys.stderr)
        ht = getattr(ann_index, "host_times", None)
        if ht:
            print("tokens + search = " + ", ".join("%s %.3f s" % kv for kv in ht.items()), file=sys.stderr)
    if world > 1:
        dist.finalize()
    return name
