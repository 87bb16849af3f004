#!/usr/bin/env python3
"""`python ao3.py {search,matrix,validate} ...` -- the command line the
reference documents for this path, served by fandom_search_amd.cli."""
import os
import sys
import time

if os.environ.get('FANDOM_SEARCH_TIMING'):
    os.environ.setdefault('FANDOM_SEARCH_T0', repr(time.time()))   # (before the imports below)

from fandom_search_amd.cli import main

if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'search' and int(os.environ.get('WORLD_SIZE', '1')) == 1:
        # a process of its own that runs one search and no collective never touches torch:
        # the HIP library is then loaded against the system runtime alone, which saves the
        # second that importing torch costs (fandom_search_amd/_lib.py: load)
        from fandom_search_amd import _lib
        _lib.PRELOAD_TORCH = False
    sys.exit(main())
