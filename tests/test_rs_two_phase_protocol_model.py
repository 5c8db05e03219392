"""CPU model check of the row-split kernel's two-phase cross-rank exchange (csrc/update_rs.hip RSX2_*, round 6).

From 4 ranks on, the data-parallel form of the row-split kernel all-reduces the 15 tagged 16-byte words of an optimiser lane
(9 for layers 2 / 3 -- phase A, words 0..8 -- and 6 for layer 1 -- phase B, words 9..14) as reduce-scatter + all-gather:

  * word g belongs to rank (g // G) mod W with G = 8 // W consecutive words (a unit) per owner;
  * SCATTER: rank r stores word g into slot [source = r][g] of the OWNER's region;
  * REDUCE_OWN: the owner polls the G x W entries of a unit in one batch, adds the W contributions in rank order (its own
    at its position), takes the mean, stores the finished word into slot [source = owner][g] of every OTHER rank's region;
  * GATHER: a rank reads [source = owner(g)][g] of its own region for the words it does not own.

Every word carries the global step as its tag; a poll accepts a slot only with the current tag; slots are double-buffered by the
tag's parity.  This file checks, without a GPU, what the kernel's comments claim about that protocol:

  * every word has exactly one owner, a unit's poll batch is at most 8 loads, and in any rank's region the entries written by
    SCATTER and by REDUCE_OWN are disjoint;
  * under a random interleaving of the ranks' programs with random store latencies (stores to different slots may land in any
    order) every accepted word is the word of THAT step from THAT sender, all ranks end every step with identical values equal to
    the rank-order mean, no store overwrites an entry some rank has yet to read, and the run terminates -- also with ONE parity
    (the kernel's double buffering is slack, as its comment argues).
The GPU tests check the arithmetic of the kernel itself (tests/test_gpu_parity.py, `rowsplit_4_ranks`, `rowsplit_8_ranks`)."""
import random

import numpy as np
import pytest

PHASES = {"A": (0, 9), "B": (9, 6)}          # first word, number of words


def owner(g, W):
    G = 1 if W >= 8 else 8 // W
    return (g // G) % W


@pytest.mark.parametrize("W", [4, 8])
def test_ownership_and_slot_map(W):
    G = 1 if W >= 8 else 8 // W
    words = range(15)
    assert all(0 <= owner(g, W) < W for g in words)
    # a unit's poll batch: G x W entries, the kernel's x_[XG_][XW_]
    assert G * W <= 8
    for name, (w0, nw) in PHASES.items():
        units = sorted({g // G for g in range(w0, w0 + nw)})
        for u in units:                                # the words of a unit that fall into this phase share an owner
            assert len({owner(g, W) for g in range(u * G, u * G + G) if w0 <= g < w0 + nw}) == 1
    # in the region of rank x: SCATTER writes [src != x][g] with owner(g) == x; REDUCE_OWN of rank o != x writes [src = o][g] with
    # owner(g) == o -- disjoint sets, and neither touches [src = x][*]
    for x in range(W):
        scatter = {(s, g) for g in words if owner(g, W) == x for s in range(W) if s != x}
        gather = {(owner(g, W), g) for g in words if owner(g, W) != x}
        assert not (scatter & gather)
        assert all(s != x for s, _ in scatter | gather)
    # exposed batches of the layer-1 phase per rank: at most one unit
    w0, nw = PHASES["B"]
    for r in range(W):
        assert len({g // G for g in range(w0, w0 + nw) if owner(g, W) == r}) <= 1


class Net:
    """Slots of all regions; stores are delivered after a random delay, in any order across different slots."""

    def __init__(self, rng, parities):
        self.slots, self.flight, self.rng, self.parities, self.now = {}, [], rng, parities, 0
        self.reads_due = {}              # slot -> set of (reader, tag) that still have to accept the CURRENT content
        self.violations = []

    def store(self, region, par, src, g, tag, value, readers):
        self.flight.append((self.now + self.rng.randint(1, 40), (region, par % self.parities, src, g), tag, value, readers))

    def tick(self):
        self.now += 1
        due = [f for f in self.flight if f[0] <= self.now]
        for f in due:
            self.flight.remove(f)
            _, key, tag, value, readers = f
            if self.reads_due.get(key):
                self.violations.append(("overwrote an entry a rank had yet to read", key, tag, sorted(self.reads_due[key])))
            self.slots[key] = (tag, value)
            self.reads_due[key] = set(readers)

    def poll(self, reader, region, par, src, g, tag):
        key = (region, par % self.parities, src, g)
        cur = self.slots.get(key)
        if cur is None or cur[0] != tag:
            return None
        self.reads_due[key].discard(reader)
        return cur[1]


def rank_program(r, W, steps, grads, net, out):
    """Generator: one rank's exchange program per step (A scattered early, B scattered, reduce A, reduce B, gather A, gather B)."""
    for s in range(steps):
        tag = s + 1
        mine = grads[s][r].copy()
        for name in ("A", "B"):
            w0, nw = PHASES[name]
            for g in range(w0, w0 + nw):
                o = owner(g, W)
                if o != r:
                    net.store(o, tag, r, g, tag, float(grads[s][r][g]), [o])
            yield
        for name in ("A", "B"):
            w0, nw = PHASES[name]
            for g in range(w0, w0 + nw):
                if owner(g, W) != r:
                    continue
                got = {}
                while len(got) < W - 1:
                    for src in range(W):
                        if src != r and src not in got:
                            v = net.poll(r, r, tag, src, g, tag)
                            if v is not None:
                                got[src] = v
                    if len(got) < W - 1:
                        yield
                acc = np.float32(grads[s][0][g] if r == 0 else got[0])
                for src in range(1, W):
                    acc = np.float32(acc + np.float32(grads[s][r][g] if src == r else got[src]))
                mine[g] = np.float32(acc * np.float32(1.0 / W))
                for x in range(W):
                    if x != r:
                        net.store(x, tag, r, g, tag, float(mine[g]), [x])
            yield
        for name in ("A", "B"):
            w0, nw = PHASES[name]
            need = [g for g in range(w0, w0 + nw) if owner(g, W) != r]
            while need:
                for g in list(need):
                    v = net.poll(r, r, tag, owner(g, W), g, tag)
                    if v is not None:
                        mine[g] = np.float32(v)
                        need.remove(g)
                if need:
                    yield
        out[s][r] = mine
        yield


@pytest.mark.parametrize("W", [4, 8])
@pytest.mark.parametrize("parities", [2, 1])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_interleaving_keeps_replicas_identical(W, parities, seed):
    rng = random.Random(1000 * W + 10 * parities + seed)
    steps = 12
    g = np.random.default_rng(seed)
    grads = g.standard_normal((steps, W, 15)).astype(np.float32)
    net = Net(rng, parities)
    out = [[None] * W for _ in range(steps)]
    progs = [rank_program(r, W, steps, grads, net, out) for r in range(W)]
    alive = list(range(W))
    for _ in range(400000):
        if not alive:
            break
        net.tick()
        r = rng.choice(alive)                          # an arbitrary rank runs until its next wait point
        try:
            next(progs[r])
        except StopIteration:
            alive.remove(r)
    assert not alive, "a rank waited forever"
    assert not net.violations, net.violations[:3]
    for s in range(steps):
        want = grads[s][0].copy()
        for src in range(1, W):
            want = (want + grads[s][src]).astype(np.float32)
        want = (want * np.float32(1.0 / W)).astype(np.float32)
        for r in range(W):
            assert np.array_equal(out[s][r], out[s][0]), (s, r)            # replicas: identical bits
        assert np.array_equal(out[s][0], want), s                          # = the rank-order mean
