"""CPU model of the suffix sort's "pair rule" (jampack_amd/csrc/bwt_fwd.hip, k_pair_*), used by tests/test_pair_rule_model.py.

Prefix doubling keeps every group of still-tied suffixes contiguous in suffix-array order, its members in DESCENDING text position
(round 0 is a stable sort fed in descending position, every later sort is stable).  A long repeat -- T[u .. u+L) == T[v .. v+L) --
leaves L groups {u+q, v+q} that prefix doubling resolves only when its distance exceeds L - q: log2(L) rounds over all of them
(divsufsort.cpp:1427-1520 has no such cliff: it induces the order of most suffixes from their successors).  The pair rule is that
induction step, applied to the active list between two doubling rounds:

  * for a member s that is not the first of its group, P[s] = s' - s, s' = the member in front of it (the next higher position);
    P = 0 everywhere else.  s and s + P[s] share at least their first byte (they are in one group), therefore
        order(s, s + p) = order(s + 1, s + 1 + p).
  * along a maximal stretch of positions [a, x] with the same non-zero P = p the argument repeats: every pair (y, y + p) of the
    stretch is ordered like (x + 1, x + 1 + p).  That pair is decided NOW if the two suffixes lie in different groups (compare
    their ranks) or if x + 1 + p is the end of the text (the empty suffix is the smaller one); otherwise the stretch stays open.
  * a second pass over the stretches also decides an end pair that lies in ONE group with other members between its two suffixes (a
    group that mixes two repeats): by the chain of neighbouring pairs between them, if all carry one verdict of the first pass.
  * repair: the induction only needs T[z] = T[z + p]; where a group mixes two repeats its members' neighbours are nearer than p and
    would end every stretch in an open pair, so a stretch is carried THROUGH such positions while z and z + p stay in one group (they
    give up their own pair: their verdict stays open and their group waits for the doubling rounds).
  * a group all of whose neighbouring pairs carry the same decided verdict is totally ordered by position: its members become
    singletons with ranks G, G+1, ...; every other group is left exactly as it was (the doubling distance does not change).

`suffix_array(t, depth0, pair_rounds)` runs prefix doubling with first key depth `depth0` and the pair rule in front of the rounds
listed in `pair_rounds`; it returns (SA, rounds used).  The test compares SA with a brute-force sort.
"""


def _groups(lst):
    """lst: list of (pos, grp) in SA order -> list of index ranges [b, e) of equal grp"""
    out = []
    b = 0
    for i in range(1, len(lst) + 1):
        if i == len(lst) or lst[i][1] != lst[b][1]:
            out.append((b, i))
            b = i
    return out


def pair_round(t, n, active, isa, sa_out, iters=2, repair=True):
    """one application of the pair rule: resolves the groups it can, returns the new active list"""
    P = [0] * (n + 1)
    for b, e in _groups(active):
        for j in range(b + 1, e):
            assert active[j - 1][0] > active[j][0], "members of a group are kept in descending position"
            P[active[j][0]] = active[j - 1][0] - active[j][0]
    # repair: a stretch is carried THROUGH positions whose own neighbour is nearer (members of a group that mixes two repeats) as long as
    # z and z + p stay in one group; such a position gives up its own pair (its verdict stays open)
    carried = [False] * (n + 1)
    if repair:
        P0 = list(P)
        best = [0] * (n + 1)
        for x in range(n - 1):
            p = P0[x]
            if p == 0 or P0[x + 1] == p:
                continue
            z = x + 1
            while z + p < n and P0[z] != p and isa[z] == isa[z + p]:
                best[z] = max(best[z], p)                   # the largest distance carried through z wins (the outer repeat, not an inner one)
                z += 1
        for z in range(n):
            if best[z]:
                P[z] = best[z]
                carried[z] = True
    # verdict of the stretch that ends at x (P[x] != P[x+1]): 1 = the lower position is the smaller suffix, 2 = the higher, 0 = open.
    # Passes after the first also decide a stretch whose end pair lies in ONE group with other members between the two: by the chain of
    # neighbouring pairs from x + 1 up to x + 1 + p, if all of them carry one verdict of the pass before.
    V = [0] * (n + 1)
    for it in range(iters):
        Vn = [0] * (n + 1)
        nxt = 0
        for y in range(n - 1, -1, -1):
            p = P[y]
            if p == 0:
                continue
            if P[y + 1] != p:                               # y ends its stretch
                a, b2 = y + 1, y + 1 + p
                if b2 >= n:
                    nxt = 2                                  # the suffix at b2 is empty: smaller
                elif isa[a] != isa[b2]:
                    nxt = 1 if isa[a] < isa[b2] else 2
                else:
                    nxt = 0
                    if it > 0:
                        z, v, ok = a, 0, True
                        for _ in range(8):
                            if z >= b2:
                                break
                            if P[z] == 0 or carried[z] or V[z] == 0 or (v and V[z] != v):
                                ok = False
                                break
                            v = V[z]
                            z += P[z]
                        if ok and z == b2:
                            nxt = v
            Vn[y] = 0 if carried[y] else nxt
        V = Vn
    out = []
    for b, e in _groups(active):
        G = active[b][1]
        vs = {V[active[j][0]] for j in range(b + 1, e)}
        if len(vs) == 1 and 0 not in vs:
            v = vs.pop()
            for j in range(b, e):
                # members are in descending position: verdict 2 (higher position smaller) = list order, 1 = reversed
                r = G + (j - b) if v == 2 else G + (e - 1 - j)
                isa[active[j][0]] = r
                sa_out[r] = active[j][0]
        else:
            out.extend(active[b:e])
    return out


def suffix_array(t, depth0=2, pair_rounds=(2, 4, 6, 8)):
    n = len(t)
    if n == 0:
        return [], 0
    sa_out = [-1] * n
    isa = [0] * n
    # round 0: stable sort by the first depth0 bytes (zero padded), fed in descending position; a suffix shorter than depth0 is a
    # group of its own
    order = sorted(range(n - 1, -1, -1), key=lambda i: bytes(t[i:i + depth0]).ljust(depth0, b"\0"))
    grp = [0] * n
    for j, s in enumerate(order):
        if j == 0:
            grp[j] = 0
        else:
            p = order[j - 1]
            same = bytes(t[s:s + depth0]).ljust(depth0, b"\0") == bytes(t[p:p + depth0]).ljust(depth0, b"\0")
            same = same and s + depth0 <= n and p + depth0 <= n
            grp[j] = grp[j - 1] if same else j
    active = []
    for j, s in enumerate(order):
        isa[s] = grp[j]
        single = (j + 1 == n or grp[j + 1] != grp[j]) and grp[j] == j
        if single:
            sa_out[j] = s
        else:
            active.append((s, grp[j]))
    h = depth0
    rounds = 1
    while active:
        rounds += 1
        if rounds in pair_rounds:
            active = pair_round(t, n, active, isa, sa_out)
            continue
        # one doubling round: every group sorted (stably) by the rank at distance h; reads before writes
        k2 = {s: (isa[s + h] + 1 if s + h < n else 0) for s, _ in active}
        new_active = []
        writes = []
        for b, e in _groups(active):
            G = active[b][1]
            mem = sorted(active[b:e], key=lambda m: k2[m[0]])
            r = G
            for j, (s, _) in enumerate(mem):
                if j and k2[s] != k2[mem[j - 1][0]]:
                    r = G + j
                writes.append((s, r))
            for j, (s, _) in enumerate(mem):
                rr = writes[len(writes) - len(mem) + j][1]
                alone = (j == 0 or writes[len(writes) - len(mem) + j - 1][1] != rr) and (j + 1 == len(mem) or writes[len(writes) - len(mem) + j + 1][1] != rr)
                if alone:
                    sa_out[rr] = s
                else:
                    new_active.append((s, rr))
        for s, r in writes:
            isa[s] = r
        active = new_active
        h *= 2
        assert rounds < 200
    return sa_out, rounds
