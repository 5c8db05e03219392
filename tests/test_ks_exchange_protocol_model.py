"""CPU model check of the inter-workgroup exchange protocol of the feature-split kernels (csrc/update_ks.hip, round 5).

The kernel all-reduces the 64 x 64 partial layer-1 pre-activations over the S workgroups of a network as reduce-scatter +
all-gather of BARE floats: no tags, a slot holds a sentinel until its producer fills it, the consumer puts the sentinel back after
reading, two parities of slots, and ONE `s_waitcnt vmcnt(0)` per step (before the reduce-scatter stores) as the only ordering
instruction.  The source argues in a comment why a slot is always reset before it is refilled; this file checks the argument by
simulation: S workgroups x 4 waves (one lane each: all lanes of a wave run the same program on their own cells) execute the
kernel's memory program under a RANDOM scheduler with RANDOM store latencies -- stores of one thread to DIFFERENT cells may become
visible in any order, exactly what a GPU allows without fences -- and every delivery / read is checked:

  * a data store must never land on a cell that still holds unconsumed data, a reset never on fresh data;
  * every value an owner sums and every broadcast a workgroup consumes must be the value of THAT step;
  * the run must finish (no workgroup waits forever).

The same model with the `vmcnt(0)` removed is run against an adversarial latency assignment (resets slow, data fast) and must
FAIL: the instruction is necessary, not decoration.  None of this needs a GPU; the GPU tests check the arithmetic."""
import random

import pytest

SENT = None            # the sentinel (0xFFFFFFFF in the kernel)


class Mem:
    def __init__(self, rng, latency):
        self.cells = {}                 # address -> value (absent = sentinel)
        self.flight = []                # (due, seq, thread, address, value)
        self.rng, self.latency, self.now, self.seq = rng, latency, 0, 0
        self.violations = []

    def store(self, thread, addr, value):
        self.seq += 1
        self.flight.append((self.now + self.latency(thread, addr, value, self.rng), self.seq, thread, addr, value))

    def pending(self, thread):
        return any(f[2] == thread for f in self.flight)

    def tick(self):
        self.now += 1
        due = sorted(f for f in self.flight if f[0] <= self.now)
        # same-address stores of one thread stay in program order (hardware does that much); anything else: by due time
        for f in due:
            older = [g for g in self.flight if g[2] == f[2] and g[3] == f[3] and g[1] < f[1]]
            if older:
                continue
            self.flight.remove(f)
            _, _, th, addr, val = f
            cur = self.cells.get(addr, SENT)
            if val is SENT:
                if cur is not SENT and cur[0] != "consumed":
                    self.violations.append(("reset landed on fresh data", addr, cur))
                self.cells.pop(addr, None)
            else:
                if cur is not SENT:
                    self.violations.append(("data landed on unconsumed data", addr, cur, val))
                self.cells[addr] = val

    def load(self, addr):
        return self.cells.get(addr, SENT)

    def mark_consumed(self, addr):
        v = self.cells.get(addr)
        if v is not None:
            self.cells[addr] = ("consumed",) + tuple(v)


def owner(u, S):
    return (u * S) >> 4


def thread_program(k, w, S, steps, mem, fence):
    """Generator: one lane of wave w of slice k.  Yields after every memory instruction (the scheduler interleaves threads)."""
    me = (k, w)
    units = [4 * fq + w for fq in range(4)]
    for s in range(steps):
        par = s & 1
        if fence:
            while mem.pending(me):                               # s_waitcnt vmcnt(0): the resets of step s-1 are acknowledged
                yield
        for u in units:                                          # reduce-scatter stores
            o = owner(u, S)
            if o != k:
                mem.store(me, ("rs", par, o, k, u), ("partial", s, k, u))
                yield
        for u in units:                                          # owner phase
            if owner(u, S) != k:
                continue
            srcs = [c for c in range(S) if c != k]
            while True:
                vals = [mem.load(("rs", par, k, c, u)) for c in srcs]
                yield
                if all(v is not SENT and v[0] != "consumed" for v in vals):
                    break
            for c, v in zip(srcs, vals):
                assert v == ("partial", s, c, u), ("owner summed a value of another step", me, s, v)
                mem.mark_consumed(("rs", par, k, c, u))
                mem.store(me, ("rs", par, k, c, u), SENT)        # reset
                yield
            for c in srcs:                                       # broadcast
                mem.store(me, ("ag", par, c, u), ("sum", s, u))
                yield
        need = [u for u in units if owner(u, S) != k]
        while True:                                              # all-gather poll
            vals = [mem.load(("ag", par, k, u)) for u in need]
            yield
            if all(v is not SENT and v[0] != "consumed" for v in vals):
                break
        for u, v in zip(need, vals):
            assert v == ("sum", s, u), ("consumed a broadcast of another step", me, s, v)
            mem.mark_consumed(("ag", par, k, u))
            mem.store(me, ("ag", par, k, u), SENT)
            yield


def run(S, steps, seed, fence=True, latency=None, max_ticks=400000):
    rng = random.Random(seed)
    latency = latency or (lambda th, addr, val, r: r.randint(1, 40))
    mem = Mem(rng, latency)
    threads = {(k, w): thread_program(k, w, S, steps, mem, fence) for k in range(S) for w in range(4)}
    alive = dict(threads)
    ticks = 0
    while alive:
        ticks += 1
        if ticks > max_ticks:
            return mem, "deadlock"
        key = rng.choice(list(alive))
        try:
            next(alive[key])
        except StopIteration:
            del alive[key]
        if rng.random() < 0.5:
            mem.tick()
    while mem.flight:
        mem.tick()
    return mem, "done"


@pytest.mark.parametrize("S", [2, 3, 4, 6, 8])
def test_exchange_protocol_is_safe_under_random_schedules_and_store_reordering(S):
    for seed in range(20):
        mem, status = run(S, steps=9, seed=100 * S + seed)
        assert status == "done", (S, seed, status)
        assert not mem.violations, (S, seed, mem.violations[:3])
        assert not mem.cells, "every slot is back to the sentinel at the end (the next launch starts clean)"


def test_the_one_waitcnt_is_necessary():
    """Without vmcnt(0) in front of the reduce-scatter stores a slow RESET can be overtaken by the refill two steps later (the
    reset then erases fresh data, or the refill lands on unconsumed data): the model must see it."""
    slow_resets = lambda th, addr, val, r: (600 if val is SENT else 1)
    bad = 0
    for seed in range(4):
        try:
            mem, status = run(4, steps=6, seed=seed, fence=False, latency=slow_resets, max_ticks=60000)
        except AssertionError:
            bad += 1
            continue
        bad += int(status != "done" or bool(mem.violations))
    assert bad == 4
    for seed in range(4):                                        # ... and with it the same adversary is harmless
        mem, status = run(4, steps=6, seed=seed, fence=True, latency=slow_resets, max_ticks=400000)
        assert status == "done" and not mem.violations
