"""ORACLE (test infrastructure only): restatement of the reference's ``BrownianInterval``.

Follows torchsde/_brownian/brownian_interval.py line by line in behaviour -- stored binary tree of intervals
(:129-350), per-node seeds from ``numpy.random.SeedSequence(entropy, spawn_key=(key, depth), pool_size)``
(:332-339), a fresh ``torch.Generator`` + ``torch.randn`` per node draw (:30-32), the insertion-ordered LRU
cache of (W, H) (:114-126), the search that starts from the last queried interval (:271-315, :638-641), the
dependency-tree heuristic (:623-634, :689-712), sub-interval merging (:643-672) and H->U (:102-103) -- but
written iteratively (no trampoline) and without the Levy-area ``A`` branch.

Because it performs the same seed derivations and the same torch CPU draws, it is pinned BIT-FOR-BIT against
the real reference by tests/golden/brownian_seq.npz (tests/test_oracle_brownian_ref.py). It serves as
(a) the cost-faithful CPU baseline in bench.py (``cpu_baseline.kind = "port"``) and (b) the statistical
yardstick for the new counter-RNG generator.
"""
import math

import numpy as np
import torch

_RSQRT3 = 1 / math.sqrt(3)


def _randn(size, dtype, seed):
    """brownian_interval.py:30-32 (CPU)."""
    gen = torch.Generator("cpu").manual_seed(int(seed))
    return torch.randn(size, dtype=dtype, generator=gen)


class _FifoCache(dict):
    """brownian_interval.py:114-126: eviction by insertion order; reads do not refresh."""

    def __init__(self, max_size):
        super().__init__()
        self._max, self._order = max_size, []

    def __setitem__(self, key, value):
        if key in self:
            self._order.remove(key)
        elif len(self) >= self._max:
            del self[self._order.pop(0)]
        super().__setitem__(key, value)
        self._order.append(key)


class _Node:
    __slots__ = ("start", "end", "parent", "is_left", "top", "midway", "spawn_key", "depth", "W_seed", "H_seed",
                 "left", "right")

    def __init__(self, start, end, parent, is_left, top):
        self.start, self.end = top._round(start), top._round(end)
        self.parent, self.is_left, self.top = parent, is_left, top
        self.midway = None

    # -- tree construction (:317-350) -------------------------------------------------------------
    def split_exact(self, midway):
        top = self.top
        self.midway = top._round(midway)
        if self.parent is None:
            self.spawn_key, self.depth = 0, 0
        else:
            self.spawn_key = 2 * self.parent.spawn_key + (0 if self.is_left else 1)
            self.depth = self.parent.depth + 1
        seq = np.random.SeedSequence(entropy=top.entropy, spawn_key=(self.spawn_key, self.depth),
                                     pool_size=top.pool_size)
        self.W_seed, self.H_seed, _, _ = seq.generate_state(4)
        self.left = _Node(self.start, midway, self, True, top)
        self.right = _Node(midway, self.end, self, False, top)

    def split(self, midway):
        if self.top.halfway_tree:
            self.split_exact(0.5 * (self.end + self.start))
            if midway > self.midway:
                self.right.split(midway)
            elif midway < self.midway:
                self.left.split(midway)
        else:
            self.split_exact(midway)

    # -- location (:271-315) -------------------------------------------------------------------------
    def loc(self, ta, tb):
        top = self.top
        out = []
        todo = [(self, top._round(ta), top._round(tb))]
        while todo:
            node, a, b = todo.pop()
            while True:
                if a < node.start or b > node.end:
                    node = node.parent
                elif a == node.start and b == node.end:
                    out.append(node)
                    break
                elif node.midway is None:
                    if a == node.start:
                        node.split(b)
                        node = node.left
                    else:
                        node.split(a)
                        node = node.right
                elif b <= node.midway:
                    node = node.left
                elif a >= node.midway:
                    node = node.right
                else:
                    todo.append((node.right, node.midway, b))
                    b = node.midway
                    node = node.left
        return out

    # -- values (:188-241) ----------------------------------------------------------------------------
    def increment_and_H(self):
        top = self.top
        chain, node = [], self
        W = H = None
        while True:
            if node.parent is None:
                W, H = top.w_h
                break
            hit = top.cache.get(node) if isinstance(top.cache, dict) else None
            if hit is not None:
                W, H = hit
                break
            chain.append(node)
            node = node.parent
        for child in reversed(chain):
            W, H = child._from_parent(W, H)
            top.cache[child] = (W, H)
        return W, H

    def _from_parent(self, W, H):
        p, top = self.parent, self.top
        h_reciprocal = 1 / (p.end - p.start)
        left_diff = p.midway - p.start
        right_diff = p.end - p.midway
        if top.have_H:
            l2, r2 = left_diff ** 2, right_diff ** 2
            l3, r3 = left_diff * l2, right_diff * r2
            v = 0.5 * math.sqrt(left_diff * right_diff / (l3 + r3))
            a = v * l2 * h_reciprocal
            b = v * r2 * h_reciprocal
            c = v * _RSQRT3
            X1 = _randn(top.size, top.dtype, p.W_seed)
            X2 = _randn(top.size, top.dtype, p.H_seed)
            third = 2 * (a * left_diff + b * right_diff) * h_reciprocal
            if self.is_left:
                first = left_diff * h_reciprocal
                second = 6 * first * right_diff * h_reciprocal
                return first * W + second * H + third * X1, first ** 2 * H - a * X1 + c * right_diff * X2
            first = right_diff * h_reciprocal
            second = 6 * first * left_diff * h_reciprocal
            return first * W - second * H - third * X1, first ** 2 * H - b * X1 - c * left_diff * X2
        mean = left_diff * W * h_reciprocal
        var = left_diff * right_diff * h_reciprocal
        left_W = mean + math.sqrt(var) * _randn(top.size, top.dtype, p.W_seed)
        return (left_W, None) if self.is_left else (W - left_W, None)


class BrownianIntervalRef:
    """Reference-algorithm Brownian motion on the CPU: ``bm(ta, tb, return_U=False)``."""

    def __init__(self, t0=0., t1=1., size=None, dtype=torch.float32, entropy=None, dt=None, tol=0., pool_size=8,
                 cache_size=45, halfway_tree=False, levy_area_approximation="none"):
        t0, t1 = float(t0), float(t1)
        self.size, self.dtype = tuple(size), dtype
        self.entropy = np.random.randint(0, 2 ** 31 - 1) if entropy is None else entropy
        self.pool_size, self.cache_size, self.halfway_tree = pool_size, cache_size, halfway_tree
        self.have_H = levy_area_approximation in ("space-time", "davie", "foster")
        self.dt = None if dt is None else float(dt)
        if tol == 0.:
            self._round = lambda x: x
        else:
            ndigits = -int(math.log10(tol))
            self._round = lambda x: round(x, ndigits)
        self.cache = {} if cache_size is None else _FifoCache(cache_size)
        self.root = _Node(t0, t1, None, None, self)
        self.last = self.root
        seq = np.random.SeedSequence(entropy=self.entropy, pool_size=pool_size)
        w_seed, h_seed, _ = seq.generate_state(3)
        W = _randn(self.size, dtype, w_seed) * math.sqrt(t1 - t0)
        H = _randn(self.size, dtype, h_seed) * math.sqrt((t1 - t0) / 12)
        self.w_h = (W, H)
        if not halfway_tree:
            self._average_dt = 0
            self._tree_dt = t1 - t0
            self._num_evaluations = -100
            if self.dt is not None:
                self._create_dependency_tree(self.dt)

    def _create_dependency_tree(self, dt):
        """:689-712."""
        cache_size = 100 if self.cache_size is None else min(self.cache_size, 100)
        self._tree_dt = min(self._tree_dt, dt)
        piece_length = self._tree_dt * cache_size * 0.8
        todo = [self.root]
        # depth-first, left before right, exactly like the reference's recursion
        while todo:
            node = todo.pop()
            if node.end - node.start > piece_length:
                midway = (node.end + node.start) / 2
                node.loc(node.start, midway)
                todo.append(node.right)
                todo.append(node.left)

    def __call__(self, ta, tb, return_U=False):
        ta, tb = float(ta), float(tb)
        ta = min(max(ta, self.root.start), self.root.end)
        tb = min(max(tb, self.root.start), self.root.end)
        if ta > tb:
            raise RuntimeError("ta <= tb required")
        if ta == tb:
            W = torch.zeros(self.size, dtype=self.dtype)
            H = torch.zeros(self.size, dtype=self.dtype) if self.have_H else None
        else:
            if self.dt is None and not self.halfway_tree:   # :623-634
                self._num_evaluations += 1
                if self._num_evaluations > 0:
                    dt = tb - ta
                    self._average_dt = (dt + self._average_dt * (self._num_evaluations - 1)) / self._num_evaluations
                    if self._average_dt < 0.5 * self._tree_dt:
                        self._create_dependency_tree(dt)
            intervals = self.last.loc(ta, tb)
            self.last = intervals[-1]
            W, H = intervals[0].increment_and_H()
            for iv in intervals[1:]:   # :647-672
                Wi, Hi = iv.increment_and_H()
                if self.have_H:
                    term1 = (iv.end - iv.start) * (Hi + 0.5 * W)
                    term2 = (iv.start - ta) * (H - 0.5 * Wi)
                    H = (term1 + term2) / (iv.end - ta)
                W = W + Wi
        if return_U:
            return W, (tb - ta) * (.5 * W + H)
        return W
