"""The pair rule of the suffix sort (bwt_fwd.hip k_pair_*; divsufsort.cpp:1427-1520 induces where prefix doubling would double) as a
Python model (tests/pair_rule_model.py) against a brute-force suffix sort on repeat-heavy texts: periods, two and many copies of
segments at arbitrary distances, runs, nested periods, Fibonacci words, copies with a few edits -- with the rule applied in front of
different rounds and after different first-key depths.  CPU only."""
import random

import functools

import pair_rule_model
from pair_rule_model import suffix_array


def _brute(t):
    return sorted(range(len(t)), key=lambda i: bytes(t[i:]))


def _gen(rng):
    kind = rng.randrange(8)
    sigma = rng.choice([1, 2, 2, 3, 4, 8, 26])
    rb = lambda k: bytes(rng.randrange(sigma) for _ in range(k))
    n = rng.randrange(1, 400)
    if kind == 0:
        return rb(n)
    if kind == 1:                                        # one period, a short tail
        p = rng.randrange(1, 40)
        return (rb(p) * (n // p + 1))[:n] + rb(rng.randrange(5))
    if kind == 2:                                        # two copies at a distance
        seg = rb(rng.randrange(1, 150))
        return rb(rng.randrange(30)) + seg + rb(rng.randrange(60)) + seg + rb(rng.randrange(30))
    if kind == 3:                                        # several copies of several segments
        segs = [rb(rng.randrange(1, 60)) for _ in range(3)]
        return b"".join(rng.choice(segs) for _ in range(rng.randrange(2, 12)))
    if kind == 4:                                        # runs
        return b"".join(bytes([rng.randrange(sigma)]) * rng.randrange(1, 50) for _ in range(rng.randrange(1, 12)))
    if kind == 5:                                        # nested periods
        a = rb(rng.randrange(1, 6))
        b = a * rng.randrange(1, 6) + rb(rng.randrange(1, 4))
        return (b * rng.randrange(1, 12))[:500]
    if kind == 6:                                        # Fibonacci word
        x, y = bytes([0]), bytes([1])
        while len(y) < n:
            x, y = y, y + x
        return y[:n]
    segs = [rb(rng.randrange(1, 30)) for _ in range(2)]  # copies with a few edits
    s = bytearray(b"".join(rng.choice(segs) for _ in range(rng.randrange(2, 20))))
    for _ in range(rng.randrange(3)):
        if s:
            s[rng.randrange(len(s))] = rng.randrange(sigma)
    return bytes(s)


def _fib(n):
    x, y = bytes([0]), bytes([1])
    while len(y) < n:
        x, y = y, y + x
    return y[:n]


def _thue_morse(n):
    return bytes(bin(i).count("1") & 1 for i in range(n))


def _period_doubling(n):
    s = bytes([0])
    while len(s) < n:
        s = b"".join(bytes([0, 1]) if c == 0 else bytes([0, 0]) for c in s)
    return s[:n]


def test_self_similar_words_any_schedule_of_pair_rounds():
    """the Fibonacci, Thue-Morse and period-doubling words -- the classic worst cases of prefix doubling -- under every
    schedule of pair rounds, also pair rounds that FOLLOW each other (the host's rule keeps doubling rounds between them; the rule itself
    is sound whatever the schedule: it only finishes groups whose order is decided)"""
    for word in (_fib, _thue_morse, _period_doubling):
        for n in (1, 2, 3, 55, 89, 233, 610, 1000):
            t = word(n)
            want = _brute(t)
            for d0 in (1, 3, 7):
                for pr in ((), (2,), (2, 3, 4), (3, 4, 5, 6, 7, 8), (3, 6, 9, 12), tuple(range(2, 60))):
                    sa, _ = suffix_array(t, d0, pr)
                    assert sa == want, (word.__name__, n, d0, pr)


def test_pair_rounds_do_not_replace_the_doubling_rounds_of_a_self_similar_word():
    """Why the host keeps two doubling rounds between two pair rounds (bwt_fwd.hip build_sa) and the Fibonacci word stays the sort's
    worst case (profiles/r06_structured_inputs.txt): on these words a pair round finishes part of the list, but what it leaves still needs
    every doubling round -- log2(n / depth) of them, each over most of the block -- so denser schedules only ADD rounds.  (A period, two
    copies, a run: one pair round finishes them, test_pair_rule_removes_the_rounds_of_a_long_repeat.)"""
    for word in (_fib, _thue_morse, _period_doubling):
        t = word(4000)
        want = _brute(t)
        rounds = {}
        for name, pr in (("none", ()), ("every third", tuple(range(3, 60, 3))), ("every second", tuple(range(3, 60, 2))), ("from 3 on", tuple(range(3, 60)))):
            sa, rounds[name] = suffix_array(t, 7, pr)
            assert sa == want, (word.__name__, name)
        doubling = rounds["none"] - 1
        assert doubling >= 8                                   # ~ log2(4000 / 7)
        assert rounds["every second"] >= rounds["every third"] >= rounds["none"], rounds
        assert rounds["from 3 on"] >= 60 or rounds["from 3 on"] >= rounds["every third"], rounds     # back-to-back pair rounds: never fewer rounds


def test_pair_rule_keeps_the_suffix_order_on_repeat_heavy_texts():
    rng = random.Random(5)
    for it in range(900):
        t = _gen(rng)
        d0 = rng.choice([1, 2, 3, 7])
        pr = rng.choice([(2,), (2, 3), (3, 6, 9), (3, 7, 11), tuple(range(2, 100, 2))])
        sa, _ = suffix_array(t, d0, pr)
        assert sa == _brute(t), (it, t, d0, pr)


def test_every_variant_of_the_rule_keeps_the_suffix_order():
    """stretches carried through mixed groups or not (k_pair_repair), one or two passes over the stretches (the chained verdict)"""
    rng = random.Random(9)
    orig = pair_rule_model.pair_round
    try:
        for rep, iters in ((False, 1), (False, 2), (True, 1), (True, 3)):
            pair_rule_model.pair_round = functools.partial(orig, repair=rep, iters=iters)
            for it in range(250):
                t = _gen(rng)
                sa, _ = suffix_array(t, rng.choice([1, 2, 3]), rng.choice([(2,), (3, 6, 9), tuple(range(2, 100, 2))]))
                assert sa == _brute(t), (rep, iters, it, t)
    finally:
        pair_rule_model.pair_round = orig


def test_a_segment_with_inner_repeats_repeated():
    """the case k_pair_repair exists for: a phrase that occurs twice inside a segment that is itself repeated -- groups that mix two repeats"""
    rng = random.Random(3)
    rb = lambda k: bytes(rng.randrange(26) for _ in range(k))
    for copies in (2, 3, 6):
        phrase = rb(40)
        seg = rb(100) + phrase + rb(120) + phrase + rb(80)
        t = rb(200) + seg * copies + rb(100)
        sa, _ = suffix_array(t, 2, (3, 6, 9))
        assert sa == _brute(t), copies


def test_pair_rule_removes_the_rounds_of_a_long_repeat():
    rng = random.Random(1)
    rb = lambda k: bytes(rng.randrange(26) for _ in range(k))
    seg = rb(300)
    for t in (rb(500) + seg * 8 + rb(500), rb(500) + seg + rb(700) + seg + rb(100)):
        sa0, r0 = suffix_array(t, 2, ())
        sa1, r1 = suffix_array(t, 2, (3,))
        assert sa0 == sa1 == _brute(t)
        assert r0 >= 9 and r1 <= 4, (r0, r1)
