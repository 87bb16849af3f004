"""TEST INFRASTRUCTURE ONLY -- pandas restatement of the reference's `format`
aggregation (/root/reference/ao3.py:346-428), with the lexicon lookup
(lextrie's emolex plugin, absent here) injected as `get_lex_tags(word) -> set`.

PARITY UNPINNED: the reference ships no fixture for this command either.

Steps, in the reference's order:
  :351-363  one boolean column per threshold: BEST_COMBINED_DISTANCE <= t for
            t = 0 ("Exact Matches") and 0.05 .. 0.5
  :365-385  script rows + ten 0/1 emotion-lexicon columns per script word
  :390-405  one 0/1 column per top-eight character (CHARACTER_<NAME upper>)
  :407-416  per ORIGINAL_SCRIPT_WORD_INDEX the sums of the boolean columns,
            re-indexed to every script word with 0
  :418-426  joined with the per-index max of ORIGINAL_SCRIPT_WORD and with the
            script table; written with DataFrame.to_csv
"""

import collections

import numpy
import pandas as pd

THRESHOLDS = [0.05, 0.1, 0.15, 0.2, 0.25, 0.3, 0.35, 0.4, 0.45, 0.5]
EMO_TERMS = ['ANGER', 'ANTICIPATION', 'DISGUST', 'FEAR', 'JOY', 'SADNESS',
             'SURPRISE', 'TRUST', 'NEGATIVE', 'POSITIVE']


def format_frame(match_table, script_rows, get_lex_tags):
    matches = pd.read_csv(match_table)
    names = ['Frequency of Reuse (Exact Matches)'] + \
        ['Frequency of Reuse (0-{})'.format(str(t)) for t in THRESHOLDS]
    flagged = matches
    for t, name in zip([0] + THRESHOLDS, names):
        flagged = flagged.assign(**{name: matches.BEST_COMBINED_DISTANCE <= t})

    header = list(script_rows[0]) + EMO_TERMS
    body = []
    for r in script_rows[1:]:
        tags = get_lex_tags(r[0])
        body.append(list(r) + [int(t in tags) for t in EMO_TERMS])
    os_markup = pd.DataFrame(body, columns=header)
    os_markup.index.name = 'ORIGINAL_SCRIPT_WORD_INDEX'

    top_eight = [name for name, _ in collections.Counter(os_markup.CHARACTER).most_common(8)]
    named = os_markup
    for name in top_eight:
        named = named.assign(**{"CHARACTER_" + name.upper(): 1 * (os_markup.CHARACTER == name)})

    counts = flagged.groupby('ORIGINAL_SCRIPT_WORD_INDEX').aggregate(
        {name: numpy.sum for name in names})
    counts = counts.reindex(named.index, fill_value=0)
    words = flagged.groupby('ORIGINAL_SCRIPT_WORD_INDEX').aggregate(
        {'ORIGINAL_SCRIPT_WORD': numpy.max})
    return counts.join(words).join(named)


def format_data(match_table, script_rows, get_lex_tags, output):
    format_frame(match_table, script_rows, get_lex_tags).to_csv(output)
