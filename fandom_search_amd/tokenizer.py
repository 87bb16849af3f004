"""Rule-based English tokenizer in the manner of spaCy 2.x (the reference tokenises with
`spacy.load('en_core_web_md')`, /root/reference/search.py:40-63, 166, 322-323).

spaCy is not available in the build or run image, so this module restates its
tokenizer: the algorithm of `spacy/tokenizer.pyx` and the English data of
`spacy/lang/punctuation.py`, `char_classes.py`, `tokenizer_exceptions.py` and
`lang/en/tokenizer_exceptions.py` (2.1/2.2 era, the reference's README dates it to 2019).
It is a restatement from the published rules, pinned by the examples in spaCy's
documentation and test-suite (tests/test_tokenizer.py), not by running spaCy: parity with
spaCy on arbitrary text is UNPINNED.  The exception tables are the full generated sets
(contraction stems x suffixes x case, times, abbreviations, months, US states, emoticons);
the URL pattern keeps the structure of spaCy's and not every clause of it.

Algorithm (spaCy `Tokenizer.__call__` / `_tokenize`):
  1. split on whitespace; runs of whitespace beyond a single space would be whitespace
     tokens, which the reference drops (`not t.is_space`), so they are not produced
  2. per chunk, repeatedly: stop if the chunk is a special case or matches `token_match`
     (URLs); peel ONE prefix and/or ONE suffix (punctuation, quotes, currency, "'s",
     units behind digits ...), checking for a special case after each peel
  3. the remainder: special case -> its pieces; `token_match` -> whole; else split at
     infixes (hyphens between letters, ellipses, "," and "." between letters ...)
  4. output prefixes + middle + suffixes (in order)

`tokenize(text)` is what `fandom_search_amd.vocab.tokenize` calls by default.  With
spaCy installed, FANDOM_SEARCH_TOKENIZER=spacy uses `spacy.blank("en")` instead (the
same rules, from the source).
"""

import re
import unicodedata

# ---- character classes (char_classes.py) -------------------------------------------


def _class_of(categories, limit=0x2000):
    """Regex character-class body for the code points below `limit` whose Unicode
    category is in `categories` (spaCy enumerates the Latin, Greek, Cyrillic ... blocks by
    hand; same sets for those blocks)."""
    out, start, prev = [], None, None
    for cp in range(limit):
        if unicodedata.category(chr(cp)) in categories:
            if start is None:
                start = cp
            prev = cp
        elif start is not None:
            out.append((start, prev))
            start = None
    if start is not None:
        out.append((start, prev))
    return "".join(re.escape(chr(a)) if a == b else "%s-%s" % (re.escape(chr(a)), re.escape(chr(b)))
                   for a, b in out)


ALPHA_LOWER = _class_of(("Ll",))
ALPHA_UPPER = _class_of(("Lu", "Lt"))
# letters without case (CJK, Hebrew, Arabic ...) count as ALPHA only
ALPHA = _class_of(("Ll", "Lu", "Lt", "Lo", "Lm"), limit=0x3000) + "぀-ヿ一-鿿가-힯"

_PUNCT = r"… …… , : ; \! \? ¿ ؟ ¡ \( \) \[ \] \{ \} < > _ # \* & 。 ？ ！ ， 、 ； ： ～ · । ، ؛ ٪"
_QUOTES = r"\' \" ” “ ` ‘ ´ ’ ‚ , „ » « 「 」 『 』 （ ） 〔 〕 【 】 《 》 〈 〉"
_CURRENCY = r"\$ £ € ¥ ฿ US\$ C\$ A\$ ₽ ﷼ ₴"
_UNITS = ("km km² km³ m m² m³ dm dm² dm³ cm cm² cm³ mm mm² mm³ ha µm nm yd in ft "
          "kg g mg µg t lb oz m/s km/h kmh mph hPa Pa mbar mb MB kb KB gb GB tb "
          "TB T G M K %")
_HYPHENS = "- – — -- --- —— ~"
# symbols and pictographs (LIST_ICONS: So category blocks)
_ICONS = "[¦©®°℀-⅏←-⇿⌀-⏿①-➿⤀-⯿" \
         "㈀-㋿\U0001f000-\U0001faff]"


def _split(s):
    return [x for x in s.strip().split(" ") if x]


LIST_PUNCT = _split(_PUNCT)
LIST_ELLIPSES = [r"\.\.+", "…"]
LIST_QUOTES = _split(_QUOTES)
LIST_CURRENCY = _split(_CURRENCY)
LIST_HYPHENS = _split(_HYPHENS)
LIST_ICONS = [_ICONS]
PUNCT = "|".join(LIST_PUNCT)
CONCAT_QUOTES = "".join(q.replace("\\", "") for q in LIST_QUOTES)
CONCAT_QUOTES_CLASS = re.escape(CONCAT_QUOTES)
CURRENCY = "|".join(LIST_CURRENCY)
UNITS = "|".join(re.escape(u) for u in sorted(_UNITS.split(), key=len, reverse=True))
HYPHENS = "|".join(re.escape(h) for h in sorted(LIST_HYPHENS, key=len, reverse=True))

# ---- punctuation rules (lang/punctuation.py) ---------------------------------------------

_prefixes = (["§", "%", "=", "—", "–", r"\+(?![0-9])"] + LIST_PUNCT + LIST_ELLIPSES + LIST_QUOTES
             + LIST_CURRENCY + LIST_ICONS)

_suffixes = (LIST_PUNCT + LIST_ELLIPSES + LIST_QUOTES + LIST_ICONS
             + ["'s", "'S", "’s", "’S", "—", "–"]
             + [r"(?<=[0-9])\+",
                r"(?<=°[FfCcKk])\.",
                r"(?<=[0-9])(?:%s)" % CURRENCY,
                r"(?<=[0-9])(?:%s)" % UNITS,
                r"(?<=[0-9%s%s(?:%s)])\." % (ALPHA_LOWER, r"%²\-\+", CONCAT_QUOTES_CLASS),
                r"(?<=[%s][%s])\." % (ALPHA_UPPER, ALPHA_UPPER)])

_infixes = (LIST_ELLIPSES + LIST_ICONS
            + [r"(?<=[0-9])[+\-\*^](?=[0-9-])",
               r"(?<=[%s%s])\.(?=[%s%s])" % (ALPHA_LOWER, CONCAT_QUOTES_CLASS, ALPHA_UPPER,
                                               CONCAT_QUOTES_CLASS),
               r"(?<=[%s]),(?=[%s])" % (ALPHA, ALPHA),
               r"(?<=[%s])(?:%s)(?=[%s])" % (ALPHA, HYPHENS, ALPHA),
               r"(?<=[%s0-9])[:<>=/](?=[%s])" % (ALPHA, ALPHA)])


def _fix_suffix_lookbehind(pattern):
    """spaCy writes `(?<=[0-9a-z%²\\-\\+(?:...)])\\.`: inside a character class the group
    syntax is just more characters, i.e. the class also holds '(', '?', ':', ')'.  Kept as
    is (Python's re accepts it); this hook only exists to say so."""
    return pattern


PREFIX_RE = re.compile("|".join("^" + p for p in _prefixes))
SUFFIX_RE = re.compile("|".join(_fix_suffix_lookbehind(s) + "$" for s in _suffixes))
INFIX_RE = re.compile("|".join(_infixes))

# token_match: URLs (tokenizer_exceptions.py URL_PATTERN, simplified to its structure:
# optional scheme and credentials, a host name with a TLD or an IPv4 address, optional
# port, path, query)
URL_RE = re.compile(
    r"^(?:(?:[\w\+\-\.]{2,})://)?(?:\S+(?::\S*)?@)?"
    r"(?:(?:[1-9]\d?|1\d\d|2[01]\d|22[0-3])(?:\.(?:1?\d{1,2}|2[0-4]\d|25[0-5])){2}"
    r"(?:\.(?:[1-9]\d?|1\d\d|2[0-4]\d|25[0-4]))"
    # host and TLD in lower case only, as in spaCy's pattern (no IGNORECASE): "Hello.World" is
    # not a URL and is split at the period by the infix rule
    r"|(?:(?:[a-z0-9\u00a1-\uffff][a-z0-9\u00a1-\uffff_-]{0,62})?[a-z0-9\u00a1-\uffff]\.)+"
    r"(?:[a-z\u00a1-\uffff]{2,63}))"
    r"(?::\d{2,5})?(?:[/?#]\S*)?$")


def token_match(s):
    return URL_RE.match(s) is not None


# ---- special cases (tokenizer_exceptions.py, lang/en/tokenizer_exceptions.py) ------------------

_EMOTICONS = r"""
:) :-) :)) :-)) :))) :-))) (: (-: =) (= ") :] :-] [: [-: :o) (o: :} :-} 8) 8-) (-8 ;) ;-) (; (-;
:( :-( :(( :-(( :((( :-((( ): )-: =( >:( :') :'-) :'( :'-( :/ :-/ =/ =| :| :-| :1 :P :-P :p :-p
:O :-O :o :-o :0 :-0 :() >:o :* :-* :3 :-3 =3 :> :-> :X :-X :x :-x :D :-D ;D ;-D =D xD XD xDD XDD
8D 8-D ^_^ ^__^ ^___^ >.< >.> <.< ._. ;_; -_- -__- v.v V.V v_v V_V o_o o_O O_o O_O 0_o o_0 0_0
o.O O.o O.O o.o 0.0 o.0 0.o @_@ <3 <33 <333 </3 (^_^) (-_-) (._.) (>_<) (*_*) (¬_¬)
ಠ_ಠ ಠ︵ಠ (ಠ_ಠ) ¯\(ツ)/¯ (╯°□°）╯︵┻━┻ ><(((*>
""".split()


def _base_exceptions():
    """spacy/lang/tokenizer_exceptions.py BASE_EXCEPTIONS (the whitespace entries cannot
    occur: chunks come from a whitespace split)."""
    exc = {}
    for orth in ["\\t", "\\n", "\u2014", "'", '\\")', "<space>", "''", "C++"]:
        exc[orth] = [orth]
    for c in "abcdefghijklmnopqrstuvwxyz\xe4\xf6\xfc":
        exc[c + "."] = [c + "."]
    for orth in _EMOTICONS:
        exc[orth] = [orth]
    return exc


def _english_exceptions():
    """spacy/lang/en/tokenizer_exceptions.py (2.x), every generated form: pronoun, wh-word
    and verb stems x 'm / 'll / 'll've / 'd / 'd've / 've / 're / 's / n't / n't've, each
    with and without the apostrophe and in lower and title case; times; the fixed lists of
    contractions, abbreviations, months and US states."""
    exc = {}

    def add(orth, pieces):
        exc[orth] = list(pieces)

    for pron in ["i"]:
        for orth in [pron, pron.title()]:
            add(orth + "'m", [orth, "'m"])
            add(orth + "m", [orth, "m"])
            add(orth + "'ma", [orth, "'m", "a"])
            add(orth + "ma", [orth, "m", "a"])
    for pron in ["i", "you", "he", "she", "it", "we", "they"]:
        for orth in [pron, pron.title()]:
            add(orth + "'ll", [orth, "'ll"])
            add(orth + "ll", [orth, "ll"])
            add(orth + "'ll've", [orth, "'ll", "'ve"])
            add(orth + "llve", [orth, "ll", "ve"])
            add(orth + "'d", [orth, "'d"])
            add(orth + "d", [orth, "d"])
            add(orth + "'d've", [orth, "'d", "'ve"])
            add(orth + "dve", [orth, "d", "ve"])
    for pron in ["i", "you", "we", "they"]:
        for orth in [pron, pron.title()]:
            add(orth + "'ve", [orth, "'ve"])
            add(orth + "ve", [orth, "ve"])
    for pron in ["you", "we", "they"]:
        for orth in [pron, pron.title()]:
            add(orth + "'re", [orth, "'re"])
            add(orth + "re", [orth, "re"])
    for pron in ["he", "she", "it"]:
        for orth in [pron, pron.title()]:
            add(orth + "'s", [orth, "'s"])
            add(orth + "s", [orth, "s"])
    for word in ["who", "what", "when", "where", "why", "how", "there", "that"]:
        for orth in [word, word.title()]:
            add(orth + "'s", [orth, "'s"])
            add(orth + "s", [orth, "s"])
            add(orth + "'ll", [orth, "'ll"])
            add(orth + "ll", [orth, "ll"])
            add(orth + "'ll've", [orth, "'ll", "'ve"])
            add(orth + "llve", [orth, "ll", "ve"])
            add(orth + "'re", [orth, "'re"])
            add(orth + "re", [orth, "re"])
            add(orth + "'ve", [orth, "'ve"])
            add(orth + "ve", [orth, "ve"])
            add(orth + "'d", [orth, "'d"])
            add(orth + "d", [orth, "d"])
            add(orth + "'d've", [orth, "'d", "'ve"])
            add(orth + "dve", [orth, "d", "ve"])
    for verb in ["ca", "could", "do", "does", "did", "had", "may", "might", "must", "need",
                 "ought", "sha", "should", "wo", "would"]:
        for orth in [verb, verb.title()]:
            add(orth + "n't", [orth, "n't"])
            add(orth + "nt", [orth, "nt"])
            add(orth + "n't've", [orth, "n't", "'ve"])
            add(orth + "ntve", [orth, "nt", "ve"])
    for verb in ["could", "might", "must", "should", "would"]:
        for orth in [verb, verb.title()]:
            add(orth + "'ve", [orth, "'ve"])
            add(orth + "ve", [orth, "ve"])
    for verb in ["ai", "are", "is", "was", "were", "have", "has", "dare"]:
        for orth in [verb, verb.title()]:
            add(orth + "n't", [orth, "n't"])
            add(orth + "nt", [orth, "nt"])
    # contractions that end or begin with an apostrophe: one token with and without it
    for stem in ["doin", "goin", "nothin", "nuthin", "ol", "somethin", "lovin", "havin"]:
        for orth in [stem, stem.title()]:
            add(orth, [orth])
            add(orth + "'", [orth + "'"])
    for stem in ["cause", "em", "ll", "nuff"]:
        add(stem, [stem])
        add("'" + stem, ["'" + stem])
    for h in range(1, 13):
        for period in ["a.m.", "am", "p.m.", "pm"]:
            add("%d%s" % (h, period), ["%d" % h, period])
    for orth, pieces in [
            ("y'all", ("y'", "all")), ("yall", ("y", "all")),
            ("how'd'y", ("how", "'d", "'y")), ("How'd'y", ("How", "'d", "'y")),
            ("not've", ("not", "'ve")), ("notve", ("not", "ve")),
            ("Not've", ("Not", "'ve")), ("Notve", ("Not", "ve")),
            ("cannot", ("can", "not")), ("Cannot", ("Can", "not")),
            ("gonna", ("gon", "na")), ("Gonna", ("Gon", "na")),
            ("gotta", ("got", "ta")), ("Gotta", ("Got", "ta")),
            ("let's", ("let", "'s")), ("Let's", ("Let", "'s"))]:
        add(orth, pieces)
    for orth in ["'S", "'s", "\u2018S", "\u2018s", "and/or", "w/o", "'re", "'Cause", "'cause", "'cos",
                 "'Cos", "'coz", "'Coz", "'cuz", "'Cuz", "'bout", "ma'am", "Ma'am", "o'clock",
                 "O'clock",
                 "Mt.", "Ak.", "Ala.", "Apr.", "Ariz.", "Ark.", "Aug.", "Calif.", "Colo.", "Conn.",
                 "Dec.", "Del.", "Feb.", "Fla.", "Ga.", "Ia.", "Id.", "Ill.", "Ind.", "Jan.", "Jul.",
                 "Jun.", "Kan.", "Kans.", "Ky.", "La.", "Mar.", "Mass.", "May.", "Mich.", "Minn.",
                 "Miss.", "N.C.", "N.D.", "N.H.", "N.J.", "N.M.", "N.Y.", "Neb.", "Nebr.", "Nev.",
                 "Nov.", "Oct.", "Okla.", "Ore.", "Pa.", "S.C.", "Sep.", "Sept.", "Tenn.", "Va.",
                 "Wash.", "Wis.",
                 "'d", "a.m.", "Adm.", "Bros.", "co.", "Co.", "Corp.", "D.C.", "Dr.", "e.g.", "E.g.",
                 "E.G.", "Gen.", "Gov.", "i.e.", "I.e.", "I.E.", "Inc.", "Jr.", "Ltd.", "Md.",
                 "Messrs.", "Mo.", "Mont.", "Mr.", "Mrs.", "Ms.", "p.m.", "Ph.D.", "Prof.", "Rep.",
                 "Rev.", "Sen.", "St.", "vs.", "v.s."]:
        add(orth, [orth])
    for excluded in ["Ill", "ill", "Its", "its", "Hell", "hell", "Shell", "shell", "Shed", "shed",
                     "were", "Were", "Well", "well", "Whore", "whore"]:
        exc.pop(excluded, None)
    merged = _base_exceptions()
    merged.update(exc)                 # update_exc(BASE_EXCEPTIONS, _exc)
    return merged


SPECIAL_CASES = _english_exceptions()


# ---- the tokenizer (tokenizer.pyx) ------------------------------------------------------------

def _find_prefix(s):
    m = PREFIX_RE.search(s)
    return m.end() - m.start() if m else 0


def _find_suffix(s):
    m = SUFFIX_RE.search(s)
    return m.end() - m.start() if m else 0


def _tokenize_chunk(string, out):
    prefixes, suffixes = [], []
    last_size = 0
    # _split_affixes
    while string and len(string) != last_size:
        if token_match(string):
            break
        if string in SPECIAL_CASES:
            break
        last_size = len(string)
        pre_len = _find_prefix(string)
        prefix = minus_pre = None
        if pre_len:
            prefix, minus_pre = string[:pre_len], string[pre_len:]
            if minus_pre and minus_pre in SPECIAL_CASES:
                string = minus_pre
                prefixes.append(prefix)
                break
        suf_len = _find_suffix(string)
        suffix = minus_suf = None
        if suf_len:
            suffix, minus_suf = string[-suf_len:], string[:-suf_len]
            if minus_suf and minus_suf in SPECIAL_CASES:
                string = minus_suf
                suffixes.append(suffix)
                break
        if pre_len and suf_len and pre_len + suf_len <= len(string):
            string = string[pre_len:-suf_len]
            prefixes.append(prefix)
            suffixes.append(suffix)
        elif pre_len:
            string = minus_pre
            prefixes.append(prefix)
        elif suf_len:
            string = minus_suf
            suffixes.append(suffix)
        if string and string in SPECIAL_CASES:
            break
    # _attach_tokens
    out.extend(prefixes)
    if string:
        if string in SPECIAL_CASES:
            out.extend(SPECIAL_CASES[string])
        elif token_match(string):
            out.append(string)
        else:
            matches = [m for m in INFIX_RE.finditer(string)]
            if not matches:
                out.append(string)
            else:
                start = 0
                for m in matches:
                    if m.start() == m.end():          # zero-width: split, no token of its own
                        if m.start() != start:
                            out.append(string[start:m.start()])
                            start = m.start()
                        continue
                    if m.start() != start:
                        out.append(string[start:m.start()])
                    if m.start() != m.end():
                        out.append(string[m.start():m.end()])
                    start = m.end()
                if start < len(string):
                    out.append(string[start:])
    out.extend(reversed(suffixes))


# The commonest kind of chunk in running text besides the bare word: a word with punctuation
# around it -- up to two of  " (  in front, ASCII letters, one full stop behind a lower-case
# letter, up to three of  , ! ? ; : " )  behind.  _tokenize_chunk strips those marks one at a
# time, each a token of its own, and looks what is left up among the special cases at every step;
# here that is decided in one go: no piece the stripping can leave on its way (from a mark in
# front or the word's start to the word's end, the full stop or a mark behind) is a special case
# -- "Mr." in "Mr.,", the emoticon in "(o:," -- else the chunk takes the long way.  A sixth of the
# time (1.3 against 7.8 us per chunk); tests/test_tokenizer.py holds the two against each other
# for words of every kind under every combination of marks.
_AFFIX_RE = re.compile(r'(["(]{0,2})([A-Za-z]+)(\.?)([,!?;:")]{0,3})\Z')


def _affix_shortcut(chunk):
    m = _AFFIX_RE.match(chunk)
    if m is None:
        return None
    pre, word, dot, suf = m.groups()
    if dot and not 'a' <= word[-1] <= 'z':
        return None                            # (a full stop behind a capital or a single capital: other rules)
    i = len(pre)
    j = i + len(word)
    n = len(chunk)
    if n == len(word):
        return None                            # the bare word: the caller's own shortcut
    special = SPECIAL_CASES
    for a in range(i + 1):
        for b in range(j, n + 1):
            if (a != i or b != j) and chunk[a:b] in special:
                return None
    return tuple(pre) + tuple(special.get(word, (word,))) + tuple(dot) + tuple(suf)


_CACHE = {}            # chunk -> tokens (spaCy keeps the same kind of cache)
_PLAIN = set()         # chunks that are their own single token
_CACHE_LIMIT = 1 << 20


def tokenize(text):
    """Token texts of `text` in spaCy's English manner, whitespace tokens left out
    (the reference drops them, search.py:166, 323)."""
    chunks = text.split()
    if _PLAIN.issuperset(chunks):              # every chunk known to be its own single token:
        return chunks                          # one pass at C speed
    out = []
    cache = _CACHE
    for chunk in chunks:
        hit = cache.get(chunk)
        if hit is None:
            if chunk.isalpha() and chunk not in SPECIAL_CASES:
                hit = (chunk,)                 # letters only: no affix, infix or URL rule applies
            else:
                hit = _affix_shortcut(chunk)
                if hit is None:
                    pieces = []
                    _tokenize_chunk(chunk, pieces)
                    hit = tuple(pieces)
            if len(cache) < _CACHE_LIMIT:
                cache[chunk] = hit
                if len(hit) == 1 and hit[0] == chunk:
                    _PLAIN.add(chunk)
        out.extend(hit)
    return out
