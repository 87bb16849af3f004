"""`ao3.py format`: per-script-word reuse counts from a match CSV
(/root/reference/ao3.py:346-428).

The aggregation -- for each script word, how many match records have
BEST_COMBINED_DISTANCE <= t for t = 0, 0.05, ... 0.5 -- runs on the GPU
(fs_reuse_histogram: one atomic per record into its first satisfied threshold,
then a cumulative pass); the join with the script table, the emotion-lexicon
columns and the top-eight character columns is host plumbing on pandas, written
with DataFrame.to_csv like the reference.

The reference takes its lexicon from lextrie's `emolex_en` plugin, which is not
redistributable here: `--lexicon FILE` reads NRC-style lines
"word<TAB>TAG[<TAB>0|1]"; without it the ten emotion columns are 0.
"""

import collections
import ctypes as C

import numpy as np

from . import _lib, abi
from . import search as search_mod

THRESHOLDS = [0.0, 0.05, 0.1, 0.15, 0.2, 0.25, 0.3, 0.35, 0.4, 0.45, 0.5]
THRESHOLD_NAMES = ['Frequency of Reuse (Exact Matches)'] + \
    ['Frequency of Reuse (0-{})'.format(str(t)) for t in THRESHOLDS[1:]]
EMO_TERMS = ['ANGER', 'ANTICIPATION', 'DISGUST', 'FEAR', 'JOY', 'SADNESS',
             'SURPRISE', 'TRUST', 'NEGATIVE', 'POSITIVE']


def reuse_histogram(orig_ix, comb, n_script, thresholds=THRESHOLDS, device=0):
    """counts[n_script][len(thresholds) + 1] (uint32): records with comb <= t per
    script word, last column = all records of the word."""
    orig = abi.as_u32(orig_ix)
    comb = np.ascontiguousarray(comb, dtype=np.float64)
    thr = np.ascontiguousarray(thresholds, dtype=np.float64)
    counts = np.zeros((int(n_script), len(thr) + 1), dtype=np.uint32)
    _lib.check(_lib.load().fs_reuse_histogram(
        device, abi.ptr(orig, C.c_uint32), abi.ptr(comb, C.c_double), len(orig), int(n_script),
        abi.ptr(thr, C.c_double), len(thr), abi.ptr(counts, C.c_uint32)), "fs_reuse_histogram")
    return counts


def load_lexicon(path):
    """word -> set of tags from NRC-style lines."""
    lex = collections.defaultdict(set)
    if path:
        with open(path, encoding='utf-8') as fh:
            for line in fh:
                parts = line.rstrip('\n').split('\t')
                if len(parts) >= 2 and (len(parts) < 3 or parts[2].strip() != '0'):
                    lex[parts[0]].add(parts[1].upper())
    return lex


def format_frame(match_table, script_file, lexicon=None, device=0):
    import pandas as pd
    matches = pd.read_csv(match_table)
    rows = search_mod.load_markup_script(script_file)
    header, body = list(rows[0]), rows[1:]
    lex = lexicon if lexicon is not None else {}

    # rows whose index is not a script word (other script, negative) are dropped by
    # the reference's reindex
    ix = matches.ORIGINAL_SCRIPT_WORD_INDEX.to_numpy()
    ok = (ix >= 0) & (ix < len(body))
    counts = reuse_histogram(ix[ok].astype(np.uint32),
                             matches.BEST_COMBINED_DISTANCE.to_numpy()[ok], len(body),
                             device=device)
    frame = pd.DataFrame(counts[:, :len(THRESHOLDS)].astype(np.int64), columns=THRESHOLD_NAMES)
    frame.index.name = 'ORIGINAL_SCRIPT_WORD_INDEX'

    # per index the max ORIGINAL_SCRIPT_WORD of the CSV (NaN where no record)
    words = matches.groupby('ORIGINAL_SCRIPT_WORD_INDEX').aggregate(
        {'ORIGINAL_SCRIPT_WORD': 'max'})
    frame = frame.join(words)

    table = [list(r) + [int(t in lex.get(r[0], ())) for t in EMO_TERMS] for r in body]
    os_markup = pd.DataFrame(table, columns=header + EMO_TERMS)
    os_markup.index.name = 'ORIGINAL_SCRIPT_WORD_INDEX'
    top_eight = [name for name, _ in collections.Counter(os_markup.CHARACTER).most_common(8)]
    for name in top_eight:
        os_markup = os_markup.assign(**{"CHARACTER_" + name.upper():
                                        1 * (os_markup.CHARACTER == name)})
    return frame.join(os_markup)


def format_data(args):
    lex = load_lexicon(getattr(args, 'lexicon', None))
    frame = format_frame(args.matches, args.script, lex, getattr(args, 'device', 0) or 0)
    frame.to_csv(args.output)
    return args.output
