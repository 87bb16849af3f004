"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the NearPy pieces that
/root/reference/search.py calls.  Never imported by the product path.

PARITY UNPINNED: NearPy is a third-party, un-vendored, unpinned dependency of
the reference (requirements.txt:2; era-appropriate release 1.0.0) and is absent
from /root/reference and from this image, and the reference holds no tests,
fixtures or golden vectors for this path.  What is
restated below is NearPy's published algorithm as the reference's call sites
use it (search.py:112-123 index build, search.py:178 query); it is pinned only
by hand-derived known answers (tests/test_oracle_known_answers.py).

Restated pieces (NearPy 1.0.0 public API):
  RandomBinaryProjections(name, projection_count)   search.py:114-115
      reset(dim): normals = RandomState(rand_seed).randn(projection_count, dim)
      hash_vector(v): ''.join('1' if x > 0.0 else '0' for x in dot(normals, v))
      -> the reference passes no seed, so hyperplanes are an INPUT here.
  MemoryStorage        buckets[hash_name][key] -> list of (vector, data)
  UniqueFilter         dict keyed by data, first-insertion order kept
  NearestFilter(10)    stable sort by distance, first N
  CosineDistance       1.0 - dot(x, y) / (norm(x) * norm(y))
  Engine               search.py:118-123,178; defaults the reference takes:
                       vector_filters=[NearestFilter(10)],
                       fetch_vector_filters=[UniqueFilter()],
                       storage=MemoryStorage()

Arithmetic: every float step goes through an `arith` object.
  LiteralArith    numpy.dot / numpy.linalg.norm on the raveled 300n-vector,
                  exactly the calls NearPy makes.  BLAS summation order is
                  implementation-defined, so this mode is compared with a
                  tolerance (1e-12 absolute on distances).
  CanonicalArith  the fixed evaluation order of DESIGN.md "canonical
                  arithmetic contract" (per-token tables, sequential sums, no
                  FMA); oracle/fs_oracle.c and the HIP kernels follow the same
                  order and are compared bit for bit.
"""

import math
from collections import defaultdict

import numpy


def unitvec(vec):
    """nearpy.utils.unitvec: scale to unit length, zero vector unchanged."""
    vec = numpy.asarray(vec, dtype=float)
    veclen = numpy.linalg.norm(vec)
    if veclen > 0.0:
        return vec / veclen
    return vec


def _seqsum(a, axis=-1):
    """Strict left-to-right float64 sum (numpy.sum is pairwise; cumsum is a
    plain running sum)."""
    a = numpy.asarray(a, dtype=numpy.float64)
    if a.shape[axis] == 0:
        return numpy.zeros(a.shape[:axis] + a.shape[axis + 1:] if axis != -1
                           else a.shape[:-1])
    return numpy.take(numpy.cumsum(a, axis=axis), -1, axis=axis)


class LiteralArith(object):
    """numpy calls as NearPy issues them.  A window is an (n, D) float64
    array; NearPy sees its ravel()."""
    name = "literal"

    def project(self, normals, win):
        # RandomBinaryProjections.hash_vector: numpy.dot(self.normals, v)
        return numpy.dot(normals, win.ravel())

    def stored(self, win):
        # Engine.store_vector: nv = unitvec(v)
        return unitvec(win.ravel())

    def query(self, win):
        # Engine._append_distances: nv = unitvec(v)
        return unitvec(win.ravel())

    def distance(self, x, y):
        # CosineDistance.distance
        with numpy.errstate(invalid="ignore", divide="ignore"):
            return float(1.0 - numpy.dot(x, y)
                         / (numpy.linalg.norm(x) * numpy.linalg.norm(y)))


class CanonicalArith(object):
    """Fixed-order float64 arithmetic (canonical contract):

      A[k](tok)[c] = seqsum_d normals[c][k*D + d] * tok[d]      (mul, then add)
      p[c]         = ((((A[0]+A[1])+A[2])+ ...)+A[n-1])[c]
      q(tok)       = seqsum_d tok[d] * tok[d]
      g(u, v)      = seqsum_d u[d] * v[d]
      SS = seq_k q(s_k), FF = seq_k q(f_k), SF = seq_k g(s_k, f_k)
      distance     = 1.0 - SF / (sqrt(SS) * sqrt(FF))
    """
    name = "canonical"

    def __init__(self):
        self._a = {}
        self._q = {}

    def _a_row(self, normals, k, tok):
        key = (id(normals), k, tok.tobytes())
        row = self._a.get(key)
        if row is None:
            d = tok.shape[0]
            row = _seqsum(normals[:, k * d:(k + 1) * d] * tok[None, :], axis=1)
            self._a[key] = row
        return row

    def project(self, normals, win):
        acc = self._a_row(normals, 0, win[0])
        for k in range(1, win.shape[0]):
            acc = acc + self._a_row(normals, k, win[k])
        return acc

    def _qtok(self, tok):
        key = tok.tobytes()
        val = self._q.get(key)
        if val is None:
            val = float(_seqsum(tok * tok))
            self._q[key] = val
        return val

    def stored(self, win):
        return win

    def query(self, win):
        return win

    def distance(self, s, f):
        ss = 0.0
        ff = 0.0
        sf = 0.0
        for k in range(s.shape[0]):
            ss = ss + self._qtok(s[k])
            ff = ff + self._qtok(f[k])
            if s[k] is f[k] or numpy.array_equal(s[k], f[k]):
                sf = sf + self._qtok(s[k])
            else:
                sf = sf + float(_seqsum(s[k] * f[k]))
        den = numpy.float64(math.sqrt(ss)) * numpy.float64(math.sqrt(ff))
        with numpy.errstate(invalid="ignore", divide="ignore"):
            return float(numpy.float64(1.0) - numpy.float64(sf) / den)


class RandomBinaryProjections(object):
    def __init__(self, hash_name, projection_count, normals, arith):
        self.hash_name = hash_name
        self.projection_count = projection_count
        self.normals = numpy.ascontiguousarray(normals, dtype=numpy.float64)
        assert self.normals.shape[0] == projection_count
        self.arith = arith

    def hash_vector(self, win):
        projection = self.arith.project(self.normals, win)
        return ["".join(["1" if x > 0.0 else "0" for x in projection])]


class MemoryStorage(object):
    def __init__(self):
        self.buckets = {}

    def store_vector(self, hash_name, bucket_key, v, data):
        self.buckets.setdefault(hash_name, defaultdict(list))[bucket_key] \
            .append((v, data))

    def get_bucket(self, hash_name, bucket_key):
        if hash_name not in self.buckets:
            return []
        if bucket_key not in self.buckets[hash_name]:
            return []
        return self.buckets[hash_name][bucket_key]


class UniqueFilter(object):
    def filter_vectors(self, input_list):
        unique_dict = {}
        for v in input_list:
            unique_dict[v[1]] = v
        return list(unique_dict.values())


class NearestFilter(object):
    def __init__(self, n):
        self.n = n

    def filter_vectors(self, input_list):
        return sorted(input_list, key=lambda x: x[2])[:self.n]


class Engine(object):
    def __init__(self, lshashes, arith, unique_filter=False, nearest=10):
        """unique_filter: whether neighbours() sends the bucket contents through UniqueFilter.
        NearPy 1.0.0's neighbours(v, distance=None, fetch_vector_filters=None,
        vector_filters=None) tests the ARGUMENT (`if fetch_vector_filters:`) and never falls
        back to the engine's own [UniqueFilter()], unlike vector_filters and distance; the
        reference passes nothing (search.py:178): OFF.  NearPy 0.2.x applied
        self.fetch_vector_filters: ON gives that behaviour."""
        self.lshashes = lshashes
        self.arith = arith
        self.storage = MemoryStorage()
        self.fetch_vector_filters = [UniqueFilter()] if unique_filter else []
        self.vector_filters = [NearestFilter(nearest)]
        self.candidate_count = 0

    def store_vector(self, win, data):
        nv = self.arith.stored(win)
        for lshash in self.lshashes:
            for bucket_key in lshash.hash_vector(win):
                self.storage.store_vector(lshash.hash_name, bucket_key, nv,
                                          data)

    def neighbours(self, win):
        candidates = []
        for lshash in self.lshashes:
            for bucket_key in lshash.hash_vector(win):
                candidates.extend(
                    self.storage.get_bucket(lshash.hash_name, bucket_key))
        for flt in self.fetch_vector_filters:
            candidates = flt.filter_vectors(candidates)
        self.candidate_count += len(candidates)
        nv = self.arith.query(win)
        candidates = [(x[0], x[1], self.arith.distance(x[0], nv))
                      for x in candidates]
        for flt in self.vector_filters:
            candidates = flt.filter_vectors(candidates)
        return candidates
