#!/usr/bin/env python3
"""`python ao3.py {search,matrix,validate} ...` -- the command line the
reference documents for this path, served by fandom_search_amd.cli."""
import sys

from fandom_search_amd.cli import main

if __name__ == '__main__':
    sys.exit(main())
