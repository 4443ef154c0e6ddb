"""CPU replay of the in-order load queue of siren_bwd16w_kernel's stream loop (fenerf_siren_bwd16w.hip, "The in-order load queue"):
the kernel's `s_waitcnt vmcnt(N)` immediates are compile-time numbers; this test reads them (and the schedule they are derived
from) out of the built library and checks, for every stage shape the kernel instantiates, that a wait never allows more loads in
flight than were issued behind the ring chunk it waits for -- i.e. that the chunk whose A operands are about to be read has landed --
and that the tape blocks are covered before their epilogue reads them.  No GPU needed: the numbers are host-callable."""
import ctypes as C
import os

import pytest

from fenerf_amd import _lib


def sched(what, qb, step):
    f = _lib.lib().fenerf_internal_bwd16w_schedule
    f.restype, f.argtypes = C.c_int, [C.c_int, C.c_int, C.c_int]
    return int(f(what, qb, step))


def stage_queue(qb, nbody, two_step, prefix_loads, t16=False):
    """Replays one stage.  A load is identified by a tag; the queue is the list of tags in issue order.  Per step: [wait + barrier]
    (every step, or every even step), the ring DMA of chunk s + DPF, then the tape DMAs of the epilogue items scheduled in the step.
    prefix_loads: loads issued just before the stage (the previous stage's tail, the FiLM DMAs of the next layer) -- unknown to the
    wait formulas, which must stay safe with any number of them.  t16: the 16-bit tape's schedule (one tape DMA per body; hooks 4-6)."""
    o = 4 if t16 else 0
    DPF = sched(3, 0, 0)
    queue = [("ring", c) for c in range(DPF)] if prefix_loads is None else []
    issued = []                                                     # (tag) in issue order
    for c in range(DPF):                                            # chunks 0 .. DPF-1 of the stage are in flight when it starts
        issued.append(("ring", c))
    for k in range(prefix_loads or 0):                              # ... possibly followed by loads the formulas do not count
        issued.append(("other", k))
    checks = []
    for s in range(nbody * qb):
        need = None
        if not two_step:
            n, need = sched(o + 0, qb, s), s + 1
        elif s % 2 == 0:
            n, need = sched(o + 1, qb, s), s + 2
        if need is not None:
            # vmcnt(n): at most the n youngest loads may still be in flight; everything older has landed
            landed = issued[:len(issued) - n] if n > 0 else list(issued)
            for c in range(need + 1):
                if c < nbody * qb + DPF:                           # chunks beyond the stage belong to the next one (same stream)
                    checks.append((s, c, ("ring", c) in landed))
        issued.append(("ring", s + DPF))
        for _ in range(sched(o + 2, qb, s % qb)):
            issued.append(("tape", s))
    return checks, issued


@pytest.mark.parametrize("t16", [False, True])
@pytest.mark.parametrize("two_step", [False, True])
@pytest.mark.parametrize("qb,nbody", [(4, 8), (5, 8), (2, 4), (3, 4), (1, 2), (2, 2), (1, 1), (4, 1), (2, 1), (2, 3), (3, 6), (4, 6)])
def test_ring_waits_never_run_ahead_of_the_dma(qb, nbody, two_step, t16):
    o = 4 if t16 else 0
    if two_step and (qb * nbody) % 2:
        pytest.skip("one barrier per two steps is only instantiated for stages of an even number of steps")
    assert sum(sched(o + 2, qb, c) for c in range(qb)) == (1 if t16 else 2)      # tape DMAs per body
    for prefix in (0, 2, 6):
        checks, issued = stage_queue(qb, nbody, two_step, prefix, t16)
        bad = [(s, c) for s, c, ok in checks if not ok]
        assert not bad, f"qb={qb} nbody={nbody} prefix={prefix}: chunk not covered by the wait at (step, chunk) {bad[:5]}"
    # and the waits are not needlessly strict in the steady state: with nothing uncounted in the queue, from step DPF - 1 on the wait
    # leaves exactly the DMAs of the chunks behind the awaited one (and the tape DMAs issued since) in flight
    DPF = sched(3, 0, 0)
    for s in range(DPF - 1, nbody * qb):
        if two_step and s % 2:
            continue
        n = sched(o + (1 if two_step else 0), qb, s)
        behind = (DPF - 3 if two_step else DPF - 2) + sum(sched(o + 2, qb, (s - j) % qb) for j in range(1, DPF - 1 if two_step else DPF))
        assert n == behind


@pytest.mark.parametrize("t16", [False, True])
def test_tape_blocks_land_two_bodies_before_use(t16):
    """A tape block is DMA'd at an E item of body b - 1 and read at the same item position of body b + 1 (2 qb steps later).  For
    2 qb >= DPF the ring wait of the step that reads it already covers it: the block was issued behind ring chunk s_t + DPF only, and
    step s_t + 2 qb waits for chunk s_t + 2 qb + 1 >= s_t + DPF + 1, which was issued AFTER the tape block.  (Smaller bodies carry an
    explicit vmcnt(2 qb) in the kernel.)"""
    o = 4 if t16 else 0
    DPF = sched(3, 0, 0)
    for qb in (4, 5):
        assert 2 * qb >= DPF
        nbody = 8
        for s_t in range(0, (nbody - 2) * qb):
            if sched(o + 2, qb, s_t % qb) == 0:
                continue
            # 16-bit tape: the body's one DMA is issued by item E(1) (chunk c1 of the body) and read first by E(0) of the body two on,
            # i.e. c1 steps EARLIER than two bodies later
            c1 = 2 * min(qb, 4) // 4
            s_use = s_t + 2 * qb - (c1 if t16 else 0)
            if t16:
                assert s_use - s_t >= DPF, "the ring waits only cover a block issued at least DPF steps before its first read"
            # loads issued after the tape DMA of step s_t up to the wait of step s_use (every-step barriers)
            younger = sum(1 + sched(o + 2, qb, t % qb) for t in range(s_t + 1, s_use))
            assert sched(o + 0, qb, s_use) <= younger, (qb, s_t)
            # two-step barriers: the covering wait is the last even step <= s_use
            s_w = s_use - (s_use % 2)
            younger2 = sum(1 + sched(o + 2, qb, t % qb) for t in range(s_t + 1, s_w))
            assert sched(o + 1, qb, s_w) <= younger2, (qb, s_t)


def test_hook_is_not_part_of_the_public_header():
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "fenerf.h")).read()
    assert "fenerf_internal_" not in hdr
