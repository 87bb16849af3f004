"""MI355X-native n-gram text-reuse search (the `ao3.py search` path of
senderle/fandom-search).  The compute path is libfandomsearch_hip.so
(csrc/, hand-written HIP for gfx950) behind include/fandom_search.h; there is
no CPU fallback."""

__all__ = ["abi", "synth", "vocab", "search", "engine", "matrix", "cli"]
