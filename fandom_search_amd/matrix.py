"""`ao3.py matrix`: works x phrases count matrix from a match CSV.

The reference advertises this command (README.md:74,149-160) but its code is
an orphan (/root/reference/_deprecated.py:83-89 subparser, :91-302
StrictNgramDedupe, :305-316 process): never imported, and `process` indexes an
argparse Namespace like a dict, so it cannot run.  This module implements the
evident intent of that code:

  1. group match rows by fan work (first-appearance order)        (:97-101)
  2. per work: runs of consecutive FAN_WORK_WORD_INDEX, inside each run the
     runs of consecutive ORIGINAL_SCRIPT_WORD_INDEX, keeping runs of at least n
     rows -- contiguous verbatim spans                            (:251-277)
  3. count, over all works, every n-gram start inside those spans, keyed by
     script word index                                            (:104-108, :279-282)
  4. per span keep the n-gram whose start is most common (first maximum)
                                                                  (:297-302)
  5. drop it when some start within +-(n-1) script words is more common (first
     maximum over ascending start positions must be the n-gram itself)
                                                                  (:290-295)
  6. matrix: one column per phrase (lower-cased script words joined by spaces)
     ordered by script index, a '(total)' row, one row per work sorted by file
     name; file '<m>-most-common-perfect-matches-no-overlap-<n>-gram-match-
     matrix.csv'                                                  (:124-149, :310)
"""

import collections
import csv


def _runs(rows, key):
    """Stable sort by int(row[key]) and split where the key is not previous+1."""
    rows = sorted(rows, key=lambda r: int(r[key]))
    out, cur, prev = [], [], None
    for r in rows:
        val = int(r[key])
        if cur and val != prev + 1:
            out.append(cur)
            cur = []
        cur.append(r)
        prev = val
    if cur:
        out.append(cur)
    return out


class StrictNgramDedupe(object):
    def __init__(self, data_path, ngram_size):
        self.ngram_size = n = int(ngram_size)
        with open(data_path, encoding='UTF8') as ip:
            self.data = list(csv.DictReader(ip))
        self.work_matches = collections.OrderedDict()
        for r in self.data:
            self.work_matches.setdefault(r['FAN_WORK_FILENAME'], []).append(r)

        spans = [span for rows in self.work_matches.values()
                 for span in self.segment_full(rows)]
        self.starts_counter = collections.Counter(
            int(span[i]['ORIGINAL_SCRIPT_WORD_INDEX'])
            for span in spans for i in range(len(span) - n + 1))
        picked = [self.top_ngram(span) for span in spans]
        self.filtered_matches = [ng for ng in picked if self.no_better_match(ng)]

    def segment_full(self, rows):
        n = self.ngram_size
        return [orig_run
                for fan_run in _runs(rows, 'FAN_WORK_WORD_INDEX')
                for orig_run in _runs(fan_run, 'ORIGINAL_SCRIPT_WORD_INDEX')
                if len(orig_run) >= n]

    def top_ngram(self, span):
        n = self.ngram_size
        count = self.starts_counter
        start = max(range(len(span) - n + 1),
                    key=lambda i: count[int(span[i]['ORIGINAL_SCRIPT_WORD_INDEX'])])
        return span[start:start + n]

    def no_better_match(self, ng):
        n = self.ngram_size
        start = int(ng[0]['ORIGINAL_SCRIPT_WORD_INDEX'])
        best = max(range(start - n + 1, start + n),
                   key=lambda s: self.starts_counter[s])
        return best == start

    def num_ngrams(self):
        return len(set(int(ng[0]['ORIGINAL_SCRIPT_WORD_INDEX'])
                       for ng in self.filtered_matches))

    @staticmethod
    def match_to_phrase(match):
        return ' '.join(m['ORIGINAL_SCRIPT_WORD'].lower() for m in match)

    def matrix_rows(self):
        phrase_ix = {}
        works = set()
        cells = collections.defaultdict(int)
        for m in self.filtered_matches:
            phrase = self.match_to_phrase(m)
            phrase_ix[phrase] = int(m[0]['ORIGINAL_SCRIPT_WORD_INDEX'])
            works.add(m[0]['FAN_WORK_FILENAME'])
            cells[(m[0]['FAN_WORK_FILENAME'], phrase)] += 1
        phrases = sorted(phrase_ix, key=phrase_ix.get)
        works = sorted(works)
        body = [[cells[(fn, ph)] for ph in phrases] for fn in works]
        totals = [sum(col) for col in zip(*body)] if body else []
        return ([['FILENAME'] + phrases, ['(total)'] + totals]
                + [[fn] + r for fn, r in zip(works, body)])

    def write_match_work_count_matrix(self, out_filename):
        with open(out_filename, 'w', encoding='utf-8') as op:
            csv.writer(op).writerows(self.matrix_rows())


def matrix_filename(prefix, ngram_size):
    return ('{}-most-common-perfect-matches-no-overlap-{}-gram-match-matrix.csv'
            .format(prefix, ngram_size))


def process(args):
    """`ao3.py matrix i m [-n N]`."""
    dd = StrictNgramDedupe(args.i, ngram_size=args.n)
    out = matrix_filename(args.m, int(args.n))
    dd.write_match_work_count_matrix(out)
    return out
