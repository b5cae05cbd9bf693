"""The device BVH build (csrc/jtx_build_dev.hip) reproduces std::partition's element order WITHOUT running it: it claims that which
elements libstdc++'s partition swaps is a function of the predicate flags alone -- with L trues, the k-th false standing in [0, L)
(from the left) changes places with the k-th true standing in [L, n) (from the right), everything else stays.  Here: libstdc++'s
algorithm (stl_algo.h __partition for bidirectional iterators, which random-access iterators take as well) run literally against
that closed form, on exhaustive small cases and random large ones.  No GPU."""
import itertools

import numpy as np


def libstdcxx_partition(a, pred):
    """stl_algo.h: __partition(first, last, pred, bidirectional_iterator_tag), element for element"""
    a = list(a)
    first, last = 0, len(a)
    while True:
        while True:
            if first == last:
                return a, first
            if pred(a[first]):
                first += 1
            else:
                break
        last -= 1
        while True:
            if first == last:
                return a, first
            if not pred(a[last]):
                last -= 1
            else:
                break
        a[first], a[last] = a[last], a[first]
        first += 1


def replay(a, flags):
    """k_partition_slots + k_scatter: ranks from one exclusive scan of the flags"""
    a = np.asarray(a); f = np.asarray(flags, bool)
    n, L = len(a), int(f.sum())
    scan = np.concatenate([[0], np.cumsum(f)[:-1]]) if n else np.zeros(0, int)          # trues before i
    at = np.arange(n)
    in_left = at < L
    falses_left = np.flatnonzero(~f & in_left)                                          # k-th false from the left: k = (i - start) - scan[i]
    trues_right = np.flatnonzero(f & ~in_left)                                          # k-th true from the right: k = L - scan[i] - 1
    kf = falses_left - scan[falses_left]
    kt = L - scan[trues_right] - 1
    slotF = np.empty(len(falses_left), int); slotF[kf] = falses_left
    slotT = np.empty(len(trues_right), int); slotT[kt] = trues_right
    assert len(slotF) == len(slotT)                                                     # as many falses on the left as trues on the right
    out = a.copy()
    out[slotT[kf]] = a[falses_left]
    out[slotF[kt]] = a[trues_right]
    return out, L


def check(flags):
    a = np.arange(len(flags))
    want, mid = libstdcxx_partition(a, lambda x: bool(flags[x]))
    got, L = replay(a, flags)
    assert mid == L
    assert list(got) == want, (flags, list(got), want)


def test_exhaustive_small():
    for n in range(0, 11):
        for flags in itertools.product((0, 1), repeat=n):
            check(flags)


def test_random_large():
    rs = np.random.RandomState(3)
    for n in (17, 64, 257, 1000, 4099):
        for p in (0.02, 0.3, 0.5, 0.9, 1.0, 0.0):
            check((rs.rand(n) < p).astype(int))
    check(np.tile([1, 0], 500)); check(np.tile([0, 1], 500)); check(np.r_[np.zeros(300, int), np.ones(300, int)])
