"""CPU model of the context codes of the suffix sort's round 0 (jampack_amd/csrc/bwt_fwd.hip: k_ctx_select / k_ctx_plan / k_key_final /
k_pack_keys_var / k_pack_keys_o2) -- the argument the GPU kernels rest on, checked by brute force:

  * every row of the code table is an ALPHABETIC prefix code over all occurring bytes (the weight-balanced splitting, any weights >= 1);
  * a key's first symbol is coded without context, every other one by the row its context selects -- a function of the one or two bytes in
    front of it only (order 2: the key's second symbol again without context, as k_pack_keys_o2 does);
  * key(i) = the first W bits of suffix i's code string (zero bits past the end of the text), depth(i) = the whole symbols in those bits.

Then for any two suffixes: key(i) < key(j) implies suffix i < suffix j; equal keys of two suffixes whose depth does not reach the end of the
text have equal depths and share that many symbols.  (divsufsort.cpp:1427-1520 sorts the suffixes themselves; the keys only have to agree.)"""
import numpy as np
import pytest


def wb_code(w):
    """weight-balanced alphabetic code (k_key_plan / k_ctx_plan's wb_walk): list of (code, length)"""
    s = len(w)
    cpre = [0]
    for x in w:
        cpre.append(cpre[-1] + int(x))
    out = [None] * s

    def rec(l, r, code, ln):
        if r - l == 1:
            out[l] = (code, max(ln, 1))
            return
        tgt2 = cpre[l] + cpre[r]
        lo, hi = l + 1, r - 1
        while lo < hi:
            mid = (lo + hi) >> 1
            if 2 * cpre[mid] >= tgt2:
                hi = mid
            else:
                lo = mid + 1
        m = lo
        if m - 1 > l and abs(2 * cpre[m - 1] - tgt2) < abs(2 * cpre[m] - tgt2):
            m -= 1
        rec(l, m, code << 1, ln + 1)
        rec(m, r, (code << 1) | 1, ln + 1)

    rec(0, s, 0, 0)
    return out


def build_tables(t, sigma, order, nclass, rng):
    """rows: 0 = no context, 1 + c = behind byte c, 1 + sigma + k = behind the k-th chosen pair; weights from the text's own counts + 1"""
    n = len(t)
    rows = [wb_code(np.bincount(t, minlength=sigma) + 1)]
    for c in range(sigma):
        cnt = np.ones(sigma, dtype=np.int64)
        for i in range(1, n):
            if t[i - 1] == c:
                cnt[t[i]] += 8
        rows.append(wb_code(cnt))
    pairmap = {}
    if order >= 2:
        pairs = [(a, b) for a in range(sigma) for b in range(sigma)]
        for k, idx in enumerate(rng.permutation(len(pairs))[:nclass]):       # ANY choice of pairs is valid: take a random one
            a, b = pairs[idx]
            cnt = np.ones(sigma, dtype=np.int64)
            for i in range(2, n):
                if t[i - 2] == a and t[i - 1] == b:
                    cnt[t[i]] += 8
            pairmap[(a, b)] = 1 + sigma + k
            rows.append(wb_code(cnt))
    return rows, pairmap


def key_of(t, i, W, rows, pairmap, sigma, order):
    n = len(t)
    bits, used, depth = 0, 0, 0
    for k in range(W + 1):
        p = i + k
        if p >= n:
            break
        if k == 0 or order == 0 or (order == 2 and k == 1):
            row = 0
        elif order == 2 and (t[p - 2], t[p - 1]) in pairmap:
            row = pairmap[(t[p - 2], t[p - 1])]
        else:
            row = 1 + t[p - 1]
        code, ln = rows[row][t[p]]
        if used + ln <= W:
            bits |= code << (W - used - ln)
            used += ln
            depth += 1
            if used == W:
                break
        else:
            bits |= code >> (ln - (W - used))
            break
    return bits, depth


@pytest.mark.parametrize("order", [0, 1, 2])
def test_keys_of_context_codes_compare_like_the_suffixes(order):
    rng = np.random.default_rng(100 + order)
    for trial in range(60):
        sigma = int(rng.integers(2, 7))
        n = int(rng.integers(20, 120))
        p = rng.random(sigma) ** 3 + 0.02
        t = rng.choice(sigma, n, p=p / p.sum())
        if trial % 3 == 0:
            t[n // 3: n // 3 + 10] = t[0]                        # a run
            t[n - 12:] = t[5: 17]                                # a copy that ends with the text
        t = [int(x) for x in t]
        W = int(rng.integers(6, 25))
        rows, pairmap = build_tables(np.array(t), sigma, order, int(rng.integers(1, sigma * sigma + 1)), rng)
        for row in rows:                                         # alphabetic and prefix-free
            strs = [format(c, "0%db" % ln) for c, ln in row]
            assert strs == sorted(strs) and all(not b.startswith(a) for a in strs for b in strs if a is not b)
        ks = [key_of(t, i, W, rows, pairmap, sigma, order) for i in range(n)]
        for i in range(n):
            for j in range(n):
                if i == j:
                    continue
                (ki, di), (kj, dj) = ks[i], ks[j]
                if ki < kj:
                    assert t[i:] < t[j:], (order, trial, i, j)
                elif ki == kj and i + di < n and j + dj < n:     # neither reaches the end of the text (those are groups of their own)
                    assert di == dj and t[i: i + di] == t[j: j + dj], (order, trial, i, j)
