#!/usr/bin/env python3
"""Export a spaCy model's vector table to the .npz that FANDOM_SEARCH_VECTORS / --vectors
expects (runs where spaCy and the model are installed; neither is in the build image).

  python tools/export_spacy_vectors.py en_core_web_md vectors_md.npz

The reference looks a token's vector up through `token.has_vector` / `token.vector`
(/root/reference/search.py:65-84): a key of `nlp.vocab.vectors` and the row it maps to.
Written:
  words    every key of the table as text (keys whose string is not in the StringStore are
           skipped: no token can produce them)
  rows     the row of `vectors` for each word (spaCy's key2row; en_core_web_md maps 685k
           keys onto 20k rows, so many words share a row, and some rows are all zeros)
  vectors  vectors.data, float32 (rows, dim)
The loader gives words that share a row the same vector id, which is what makes their
windows identical for the search, as their equal vectors do in the reference.
"""
import sys

import numpy as np


def main():
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    import spacy
    nlp = spacy.load(sys.argv[1])
    vec = nlp.vocab.vectors
    words, rows = [], []
    for key, row in vec.key2row.items():
        try:
            text = nlp.vocab.strings[key]
        except KeyError:
            continue
        words.append(text)
        rows.append(int(row))
    data = np.ascontiguousarray(vec.data, dtype=np.float32)
    np.savez_compressed(sys.argv[2], words=np.array(words, dtype=np.str_), rows=np.array(rows, dtype=np.int64),
                        vectors=data)
    zero = int((np.abs(data).sum(axis=1) == 0).sum())
    print("%d words on %d rows of %d dimensions (%d all-zero rows) -> %s"
          % (len(words), data.shape[0], data.shape[1], zero, sys.argv[2]))


if __name__ == "__main__":
    main()
