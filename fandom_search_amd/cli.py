"""Command line of the search path: the `search`, `format`, `matrix` and
`validate` sub-commands of the reference's ao3.py (/root/reference/ao3.py:509-526
and _deprecated.py:83-89), same positionals, flags and output files.  The
reference's scrape / clean / getmeta / vis sub-commands are outside this package
(out of scope)."""

import argparse
import sys


def build_parser():
    parser = argparse.ArgumentParser(
        description='n-gram text-reuse search of fan works against a script '
                    '(MI355X build of the `ao3.py search` path).')
    subparsers = parser.add_subparsers(help='search, format, matrix or validate')

    validate_parser = subparsers.add_parser('validate', help='validate script markup')
    validate_parser.add_argument('script', action='store',
                                 help='filename for markup version of script')
    validate_parser.set_defaults(func=_validate)

    search_parser = subparsers.add_parser(
        'search', help='compare fanworks with the original script')
    search_parser.add_argument('fan_works', action='store',
                               help='directory of fanwork text files')
    search_parser.add_argument('script', action='store',
                               help='filename for markup version of script')
    search_parser.add_argument('-n', '--num-works', default=-1, type=int,
                               help="number of works to search (for subsampling)")
    search_parser.add_argument('-s', '--skip-works', default=0, type=int,
                               help="number of works to skip (for subsampling)")
    # additions; defaults reproduce the reference's behaviour
    search_parser.add_argument('--window-size', default=None, type=int,
                               help='n-gram size (reference: fixed at 6)')
    search_parser.add_argument('--device', default=0, type=int,
                               help='HIP device ordinal')
    search_parser.add_argument('--vectors', default=None,
                               help="vector table, .npz with 'words' and 'vectors' (what the "
                                    "reference takes from spaCy's en_core_web_md); also "
                                    "FANDOM_SEARCH_VECTORS")
    search_parser.add_argument('--synthetic-vocab', action='store_true',
                               help='use the synthetic benchmark vocabulary (8192 pseudo-words '
                                    'with random vectors: no semantic similarity)')
    search_parser.add_argument('--unique-filter', default=None, type=int, choices=(0, 1),
                               help="NearPy's UniqueFilter on a query's bucket contents: 0 = what "
                                    "NearPy 1.0.0 does for the reference's call (default), 1 = "
                                    "NearPy 0.2.x; also FANDOM_SEARCH_UNIQUE_FILTER")
    search_parser.add_argument('--listing', default=None, choices=('sorted', 'os'),
                               help="order of the directory listing in front of the seeded shuffle: "
                                    "'sorted' (default: a run repeats anywhere) or 'os' (os.listdir() as "
                                    "it comes, what the reference shuffles); also FANDOM_SEARCH_LISTING")
    search_parser.set_defaults(func=_search)

    data_parser = subparsers.add_parser(
        'format', help='takes a script and outputs a csv with reuse counts for each word '
                       'formatted for javascript visualization')
    data_parser.add_argument('matches', action='store', help='filename for search output')
    data_parser.add_argument('script', action='store',
                             help='filename for markup version of script')
    data_parser.add_argument('-o', '--output', action='store', default='js-data.csv',
                             help='filename for csv output file of data formatted for visualization')
    data_parser.add_argument('--lexicon', default=None,
                             help='emotion lexicon, lines "word<TAB>TAG[<TAB>0|1]" '
                                  '(the reference uses lextrie emolex_en)')
    data_parser.add_argument('--device', default=0, type=int, help='HIP device ordinal')
    data_parser.set_defaults(func=_format)

    matrix_parser = subparsers.add_parser(
        'matrix', help='deduplicates and builds matrix for best n-gram matches')
    matrix_parser.add_argument('i', action='store', help='input csv file')
    matrix_parser.add_argument('m', action='store',
                               help='fandom/movie name for output file prefix')
    matrix_parser.add_argument('-n', action='store', default=6, type=int,
                               help='n-gram size, default is 6-grams')
    matrix_parser.set_defaults(func=_matrix)
    return parser


def _validate(args):
    from . import search
    return search.validate_cmd(args)


def _search(args):
    import os
    from . import search
    if getattr(args, 'vectors', None):
        os.environ['FANDOM_SEARCH_VECTORS'] = args.vectors
    if getattr(args, 'synthetic_vocab', False):
        os.environ['FANDOM_SEARCH_SYNTHETIC_VOCAB'] = '1'
    if getattr(args, 'unique_filter', None) is not None:
        os.environ['FANDOM_SEARCH_UNIQUE_FILTER'] = str(args.unique_filter)
    if getattr(args, 'listing', None):
        os.environ['FANDOM_SEARCH_LISTING'] = args.listing
    return search.analyze(args)


def _format(args):
    from . import format as format_mod
    return format_mod.format_data(args)


def _matrix(args):
    from . import matrix
    return matrix.process(args)


def main(argv=None):
    parser = build_parser()
    args = parser.parse_args(argv)
    if hasattr(args, 'func'):
        args.func(args)
    else:
        parser.print_help()
    return 0


if __name__ == '__main__':
    sys.exit(main())
