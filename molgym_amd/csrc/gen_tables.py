#!/usr/bin/env python3
"""Generates cg_tables.inc: sparse Clebsch-Gordan tables for maxl = 4.

Index conventions shared with the kernels:
  idx(l, m)  = l*l + m + l                    in [0, 25)
  block list of output degree l: all (l1, l2), l1 outer / l2 inner, with
      |l1-l2| <= l <= min(l1+l2, 4)           (5, 12, 16, 17, 15 blocks)
  forward row  (l, blkpos, m): terms (idx1, idx2, coef) with m1 + m2 = m
  transposed   (idx1, idx2):   terms (l, blkpos, coef)
Run: python molgym_amd/csrc/gen_tables.py  (writes next to itself).
"""
import math
import os
import sys

MAXL = 4


def _slot_cycles(offs):
    """LDS cycles of one 8-byte gather instruction: lane groups {0-31}, {32-63}, bank = dword address mod 64, identical
    addresses broadcast, every further distinct address on a busy bank costs a cycle (offs: dword offsets, None = padding)"""
    tot = 0
    for grp in (offs[:32], offs[32:]):
        banks = {}
        for o in grp:
            if o is None:
                continue
            banks.setdefault(o % 64, set()).add(o)
            banks.setdefault((o + 1) % 64, set()).add(o)
        tot += max((len(v) for v in banks.values()), default=1)
    return tot


def deconflict(ents, gm, to_dw, seed):
    """ents[t] = the (offset, coefficient) terms of lane t of a 64-lane group (a lane sums them: any order); returns slots[j][t]
    for j < gm with every lane's terms spread over the slots so that the 64 addresses of a slot hit as few common LDS banks as
    a local search finds (the given order cost 2.1 - 2.3x the conflict-free cycles).  Padding slots (coefficient 0) take the
    address of another lane of their lane group in that slot: a broadcast, never a conflict."""
    import random
    rng = random.Random(seed)
    n = len(ents)
    assign = [list(range(len(e))) + [None] * (gm - len(e)) for e in ents]  # assign[t][j] = index into ents[t] or None
    def offs(j):
        return [to_dw(ents[t][assign[t][j]][0]) if t < n and assign[t][j] is not None else None for t in range(64)]
    cost = [_slot_cycles(offs(j)) for j in range(gm)]
    if gm > 1:
        for _ in range(10000 * gm):
            t = rng.randrange(n)
            j1, j2 = rng.sample(range(gm), 2)
            if assign[t][j1] is None and assign[t][j2] is None:
                continue
            before = cost[j1] + cost[j2]
            assign[t][j1], assign[t][j2] = assign[t][j2], assign[t][j1]
            c1, c2 = _slot_cycles(offs(j1)), _slot_cycles(offs(j2))
            if c1 + c2 <= before:
                cost[j1], cost[j2] = c1, c2
            else:
                assign[t][j1], assign[t][j2] = assign[t][j2], assign[t][j1]
    slots = []
    for j in range(gm):
        row = []
        for t in range(64):
            if t < n and assign[t][j] is not None:
                row.append(ents[t][assign[t][j]])
            else:
                row.append(None)
        for half in (range(0, 32), range(32, 64)):
            real = [row[t][0] for t in half if row[t] is not None]
            for t in half:
                if row[t] is None:
                    row[t] = (real[0] if real else 0, 0.0)
        slots.append(row)
    return slots


def _write_cycles(addrs):
    """LDS array cycles of one 8-byte store instruction (ds_write_b64): four groups of 16 CONTIGUOUS lanes, bank = dword address
    mod 32, a lane covers two banks; identical addresses count once (addrs: 64 dword addresses)"""
    tot = 0
    for g in range(4):
        banks = {}
        for a in addrs[16 * g:16 * g + 16]:
            banks.setdefault(a % 32, set()).add(a)
            banks.setdefault((a + 1) % 32, set()).add(a)
        tot += max(len(v) for v in banks.values())
    return tot


def order_for_stores(items, counts, gmax, addr_fns, pad_addr, seed, iters=60000):
    """A lane of a 64-lane group stores its result(s) at addresses the data layout dictates; WHICH lane of WHICH group takes which
    item is free as long as the item's term count fits the group's slot count gmax[g].  items (in the sorted work order: feasible)
    -> list of groups of 64 (None = padding lane, stores at pad_addr) such that every group's store instructions (one per function
    in addr_fns: item -> dword address of an 8-byte store) hit as few common banks per 16-lane store group as a local search over
    pairwise exchanges finds.  Round 6: the stores of the rebuilt adjoint moments / of the projected rows cost 3.5x / 2.4x their
    conflict-free cycles in the sorted order (tools/lds_model.py)."""
    import random
    rng = random.Random(seed)
    ng = len(gmax)
    slots = list(items) + [None] * (64 * ng - len(items))
    cnt = lambda x: 0 if x is None else counts[x]  # noqa: E731

    def gcost(g):
        # search objective: the number of EXTRA distinct addresses per bank, summed over the group's 16-lane store groups and
        # stores (0 = conflict-free; the cycle count itself, a maximum per store group, is flat almost everywhere)
        tot = 0
        for f in addr_fns:
            for sg in range(4):
                banks = {}
                for x in slots[64 * g + 16 * sg:64 * g + 16 * sg + 16]:
                    a = f(x) if x is not None else pad_addr
                    banks.setdefault(a % 32, set()).add(a)
                    banks.setdefault((a + 1) % 32, set()).add(a)
                tot += sum(len(v) - 1 for v in banks.values())
        return tot
    def sub_bad(sg):
        """lanes of 16-lane store group sg that sit on a bank another lane of the group uses too (for any of the stores)"""
        bad = set()
        for f in addr_fns:
            banks = {}
            for t in range(16 * sg, 16 * sg + 16):
                a = f(slots[t]) if slots[t] is not None else pad_addr
                for b in (a % 32, (a + 1) % 32):
                    banks.setdefault(b, {}).setdefault(a, []).append(t)
            for by_addr in banks.values():
                if len(by_addr) > 1:
                    for ts in by_addr.values():
                        bad.update(ts)
        return sorted(bad)
    cost = [gcost(g) for g in range(ng)]
    floor = 0
    for it in range(iters):
        if sum(cost) <= floor:
            break
        # targeted move: a lane that takes part in a conflict, exchanged with a random lane of another 16-lane store group
        sg = rng.randrange(4 * ng)
        bad = sub_bad(sg)
        if not bad:
            continue
        a, b = rng.choice(bad), rng.randrange(64 * ng)
        ga, gb = a // 64, b // 64
        if a // 16 == b // 16 or cnt(slots[a]) > gmax[gb] or cnt(slots[b]) > gmax[ga]:
            continue
        before = cost[ga] + (cost[gb] if gb != ga else 0)
        slots[a], slots[b] = slots[b], slots[a]
        ca, cb = gcost(ga), (gcost(gb) if gb != ga else 0)
        if ca + cb <= before:
            cost[ga] = ca
            if gb != ga:
                cost[gb] = cb
        else:
            slots[a], slots[b] = slots[b], slots[a]
    return [slots[64 * g:64 * g + 64] for g in range(ng)]


def cg(l1, m1, l2, m2, l, m):
    if m1 + m2 != m or l < abs(l1 - l2) or l > l1 + l2:
        return 0.0
    if abs(m1) > l1 or abs(m2) > l2 or abs(m) > l:
        return 0.0
    f = math.factorial
    pref = (2 * l + 1) * f(l1 + l2 - l) * f(l1 - l2 + l) * f(-l1 + l2 + l) / f(l1 + l2 + l + 1)
    pref *= f(l + m) * f(l - m) * f(l1 - m1) * f(l1 + m1) * f(l2 - m2) * f(l2 + m2)
    tot = 0.0
    for k in range(0, l1 + l2 + l + 1):
        d = [k, l1 + l2 - l - k, l1 - m1 - k, l2 + m2 - k, l - l2 + m1 + k, l - l1 - m2 + k]
        if min(d) < 0:
            continue
        den = 1
        for x in d:
            den *= f(x)
        tot += (-1.0)**k / den
    return math.sqrt(pref) * tot


def idx(l, m):
    return l * l + m + l


def blocks(l):
    return [(l1, l2) for l1 in range(MAXL + 1) for l2 in range(MAXL + 1) if abs(l1 - l2) <= l <= min(l1 + l2, MAXL)]


def main():
    # (the bank-conflict search below takes ~20 s: skipped when the generated file carries the hash of this version of the script)
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    out_path = os.path.join(here, 'cg_tables.inc')
    gen_hash = hashlib.sha1(open(os.path.abspath(__file__), 'rb').read()).hexdigest()
    if '--force' not in sys.argv and os.path.exists(out_path) and ('generator sha1: ' + gen_hash) in open(out_path).readline():
        return
    nblk = [len(blocks(l)) for l in range(MAXL + 1)]
    row_base, rb = [], 0
    for l in range(MAXL + 1):
        row_base.append(rb)
        rb += nblk[l] * (2 * l + 1)
    nrows = rb
    row_start, terms = [], []
    for l in range(MAXL + 1):
        for (l1, l2) in blocks(l):
            for m in range(-l, l + 1):
                row_start.append(len(terms))
                for m1 in range(-l1, l1 + 1):
                    m2 = m - m1
                    if abs(m2) > l2:
                        continue
                    c = cg(l1, m1, l2, m2, l, m)
                    if abs(c) > 1e-14:
                        terms.append((idx(l1, m1), idx(l2, m2), c))
    row_start.append(len(terms))
    # transposed
    t_start, t_terms, t_keys = [], [], []
    for i1 in range(25):
        l1 = int(math.isqrt(i1))
        m1 = i1 - l1 * l1 - l1
        for i2 in range(25):
            l2 = int(math.isqrt(i2))
            m2 = i2 - l2 * l2 - l2
            t_start.append(len(t_terms))
            for l in range(abs(l1 - l2), min(l1 + l2, MAXL) + 1):
                m = m1 + m2
                if abs(m) > l:
                    continue
                c = cg(l1, m1, l2, m2, l, m)
                if abs(c) > 1e-14:
                    t_terms.append((l, blocks(l).index((l1, l2)), c))
                    t_keys.append(i1 * 25 + i2)
    t_start.append(len(t_terms))
    assert len(terms) == len(t_terms)

    # incidence lists: for every index i the terms it takes part in, as (output row, partner index, c) -- the
    # adjoint of sq[row] = sum c x[i1] x[i2] w.r.t. x[i] is sum over this list of c * dsq[row] * conj(x[partner])
    inc = [[] for _ in range(25)]
    for r in range(nrows):
        for q in range(row_start[r], row_start[r + 1]):
            i1, i2, c = terms[q]
            inc[i1].append((r, i2, c))
            inc[i2].append((r, i1, c))
    inc_start = [0]
    for i in range(25):
        inc_start.append(inc_start[-1] + len(inc[i]))
    inc_flat = [e for lst in inc for e in lst]
    assert len(inc_flat) == 2 * len(terms)

    # per forward term: everything the scatter form of the adjoint needs in one word --
    # i1 | i2 << 5 | l << 10 | (m index) << 13 | (block position) << 17
    term_pk = []
    r = 0
    for l in range(MAXL + 1):
        for bp, (l1, l2) in enumerate(blocks(l)):
            for mi in range(2 * l + 1):
                for q in range(row_start[r], row_start[r + 1]):
                    i1, i2, _ = terms[q]
                    term_pk.append(i1 | (i2 << 5) | (l << 10) | (mi << 13) | (bp << 17))
                r += 1
    assert len(term_pk) == len(terms)

    # gather form of the adjoint (key (i1, i2) -> terms): position of the term's AGGREGATE entry inside one channel's
    # slice of an atom's concatenated row block -- slice = for l, for m: [ag blocks | in | sq blocks] (2 nblk_l + 1
    # complex per row) -- and the distance nblk_l + 1 from it to the POWER entry: off | (nblk_l + 1) << 16
    slice_base, sb = [], 0
    for l in range(MAXL + 1):
        slice_base.append(sb)
        sb += (2 * l + 1) * (2 * nblk[l] + 1)
    assert sb == 775
    g_pk = []
    for key in range(625):
        i1, i2 = key // 25, key % 25
        l1, l2 = int(math.isqrt(i1)), int(math.isqrt(i2))
        m = (i1 - l1 * l1 - l1) + (i2 - l2 * l2 - l2)
        for q in range(t_start[key], t_start[key + 1]):
            l, bp, _ = t_terms[q]
            off = slice_base[l] + (m + l) * (2 * nblk[l] + 1) + bp
            g_pk.append(off | ((nblk[l] + 1) << 16))

    # work orders for the one-wave-per-(atom, channel) kernels: rows / keys sorted by DEcreasing term count, so that the
    # 64 lanes of one pass have similar trip counts (the pass runs max-of-the-group predicated iterations)
    def order(counts):
        idx = sorted(range(len(counts)), key=lambda i: -counts[i])
        gmax = [max(counts[i] for i in idx[g:g + 64]) for g in range(0, len(idx), 64)]
        return idx, gmax
    row_cnt = [row_start[r + 1] - row_start[r] for r in range(nrows)]
    row_perm, row_gmax = order(row_cnt)
    key_cnt = [t_start[k + 1] - t_start[k] for k in range(625)]
    key_perm, key_gmax = order(key_cnt)
    pairs = [(x, y) for x in range(25) for y in range(x, 25)]
    pair_cnt = [key_cnt[x * 25 + y] + (key_cnt[y * 25 + x] if x != y else 0) for x, y in pairs]
    pidx, pair_gmax = order(pair_cnt)
    pair_perm = [pairs[i][0] * 25 + pairs[i][1] for i in pidx]

    # RESOLVED lane-major tables of the adjoint rebuilds (k_catbuild_bwd_mfma): lane = one entry of the 25 x 25 adjoint
    # moment matrix, groups of 64 lanes in the sorted work orders above, group g padded to its maximum count G[g].
    # Slot (g, j) of lane t is ONE word pair {slice offset, coefficient} at [(slot_base[g] + j) * 64 + t]: consecutive
    # lanes read consecutive 8-byte words (conflict-free ds_read_b64, no dependent index look-ups); padding slots have
    # coefficient 0.  bk_*: aggregate block, entries (x, y) in key order; bp_*: symmetrised power block
    # S[x][y] = dP[x][y] + dP[y][x], pairs x <= y (x == y counted from both sides: coefficient doubled), offsets point at
    # the POWER entry (aggregate offset + nblk_l + 1).  *_pos: where the lane's result goes in the LDS matrix (float
    # index x * 52 + 2 y; pairs: both mirror positions, 16 bits each), CG_POS_DUMP for padding lanes.
    LD = 52   # forward kernel: floats per row of its moment matrices (CGM_LD, cg_mfma.inc)
    # adjoint kernel: its matrix is read as an MFMA A operand (lane (i, q) reads row i, column q + 4 s: ds_read_b32, banks mod
    # 32 over 32-lane groups): with 52 floats per row the rows i and i + 8 meet on one bank (52 i mod 32 has 8 values) -- a
    # 2-way conflict on every one of the 52 reads per item; any stride = 2 mod 4 spreads the 16 rows over 16 bank pairs
    LDB = 54
    # padding lanes store their (zero) result into a spare word of the wave's LDS block: float index of entry 775 of the
    # slice buffer that follows the 26 x LDB matrix (k_catbuild_bwd_mfma static_asserts this layout) -- no branch
    POS_DUMP = 26 * LDB + 2 * 775
    def resolved(lanes, counts, gmax, entries_of, pos_of, pad_pos, addr_fns):
        off, cf, pos = [], [], []
        groups = order_for_stores(lanes, counts, gmax, addr_fns, POS_DUMP, 3000)
        for g, gm in enumerate(gmax):
            grp = groups[g]
            ents = [entries_of(k) if k is not None else [] for k in grp]
            for row in deconflict(ents, gm, lambda o: 2 * o, 1000 + g):  # (complex offsets: 8 bytes)
                for o, c in row:
                    off.append(o)
                    cf.append(c)
            pos += [pos_of(k) if k is not None else pad_pos for k in grp]
        return off, cf, pos
    def key_entries(key):
        return [(g_pk[q] & 0xffff, t_terms[q][2]) for q in range(t_start[key], t_start[key + 1])]
    def pair_entries(key):
        x, y = key // 25, key % 25
        ent = [((g_pk[q] & 0xffff) + (g_pk[q] >> 16), t_terms[q][2] * (2.0 if x == y else 1.0))
               for q in range(t_start[key], t_start[key + 1])]
        if x != y:
            kb = y * 25 + x
            ent += [((g_pk[q] & 0xffff) + (g_pk[q] >> 16), t_terms[q][2]) for q in range(t_start[kb], t_start[kb + 1])]
        return ent
    pos_xy = lambda k: (k // 25) * LDB + 2 * (k % 25)  # noqa: E731
    pos_yx = lambda k: (k % 25) * LDB + 2 * (k // 25)  # noqa: E731
    bk_off, bk_c, bk_pos = resolved(key_perm, {k: key_cnt[k] for k in range(625)}, key_gmax, key_entries, pos_xy, POS_DUMP, [pos_xy])
    bp_off, bp_c, bp_pos = resolved(pair_perm, {pairs[i][0] * 25 + pairs[i][1]: pair_cnt[i] for i in range(len(pairs))}, pair_gmax,
                                    pair_entries, lambda k: pos_xy(k) | (pos_yx(k) << 16), POS_DUMP | (POS_DUMP << 16), [pos_xy, pos_yx])
    assert len(bk_off) == 64 * sum(key_gmax) and len(bp_off) == 64 * sum(pair_gmax)

    # forward projection, staged per part of the channel slice (parts: l <= 2, l = 3, l = 4): one word per output row,
    # t0 | count << 11 | (slice position of the aggregate entry) << 15 | (nblk_l + 1) << 25, rows of a part sorted by
    # decreasing term count and padded to groups of 64 lanes (padding: count 0, position 1023)
    row_info = []
    r = 0
    for l in range(MAXL + 1):
        for bp in range(nblk[l]):
            for mi in range(2 * l + 1):
                pos = slice_base[l] + mi * (2 * nblk[l] + 1) + bp
                row_info.append((l, row_start[r], row_start[r + 1] - row_start[r], pos, nblk[l] + 1))
                r += 1
    rowS, rowS_gmax, rowS_part = [], [], []
    # RESOLVED lane-major table of the forward projection (k_catbuild_mfma): slot (g, j) of lane t = {float index of the moment
    # G[i1][i2] (= i1 * 52 + 2 * i2; the power moment A[i1] A[i2] sits at the same index of a second matrix), coefficient};
    # fw_pos: where the lane's two results go inside the LDS stage of its part, aggregate | power << 16 (part-relative slice
    # positions), CG_FW_DUMP for padding lanes
    FW_DUMP = 511
    fw_off, fw_c, fw_pos, fw_lbm = [], [], [], []  # fw_lbm: the lane's row as l | block position << 3 | m index << 8 (0xffff: padding)
    part_base = [0, 251, 496]
    for part, ls in enumerate(((0, 1, 2), (3, ), (4, ))):
        rows = sorted((ri for ri in row_info if ri[0] in ls), key=lambda ri: -ri[2])
        part_gmax = [max(ri[2] for ri in rows[g:g + 64]) for g in range(0, len(rows), 64)]
        # which lane of which group takes which row: ordered against the bank conflicts of the two 8-byte stores into the stage
        # (aggregate entry at the row's part-relative slice position, power entry nblk_l + 1 behind it)
        part_groups = order_for_stores(list(range(len(rows))), {i: rows[i][2] for i in range(len(rows))}, part_gmax,
                                       [lambda i: 2 * (rows[i][3] - part_base[part]), lambda i: 2 * (rows[i][3] - part_base[part] + rows[i][4])],
                                       2 * FW_DUMP, 4000 + part)
        for gi, gm in enumerate(part_gmax):
            grp = [rows[i] if i is not None else None for i in part_groups[gi]]
            rowS_gmax.append(gm)
            rowS_part.append(part)
            for ri in grp:
                rowS.append((ri[1] | (ri[2] << 11) | (ri[3] << 15) | (ri[4] << 25)) if ri is not None else (1023 << 15))
            ents = [[(terms[ri[1] + j][0] * LD + 2 * terms[ri[1] + j][1], terms[ri[1] + j][2]) for j in range(ri[2])] if ri is not None else []
                    for ri in grp]
            for row in deconflict(ents, gm, lambda o: o, 2000 + len(rowS_gmax)):  # (float offsets of 8-byte entries)
                for o, c in row:
                    fw_off.append(o)
                    fw_c.append(c)
            for t in range(64):
                if grp[t] is not None:
                    pos = grp[t][3] - part_base[part]
                    fw_pos.append(pos | ((pos + grp[t][4]) << 16))
                    l = grp[t][0]
                    rel = grp[t][3] - slice_base[l]          # = mi * (2 nblk + 1) + bp
                    fw_lbm.append(l | ((rel % (2 * nblk[l] + 1)) << 3) | ((rel // (2 * nblk[l] + 1)) << 8))
                else:
                    fw_pos.append(FW_DUMP | (FW_DUMP << 16))
                    fw_lbm.append(0xffff)

    out = []
    w = out.append
    w('// GENERATED by gen_tables.py (generator sha1: ' + gen_hash + ') -- do not edit.  Sparse real Clebsch-Gordan tables, maxl = 4.')
    w(f'#define CG_NNZ {len(terms)}')
    w(f'#define CG_NROWS {nrows}')
    w('static const int h_cg_nblk[5] = {' + ', '.join(map(str, nblk)) + '};')
    w('#define CG_NBLK_SEL(l) ((l) == 0 ? %d : (l) == 1 ? %d : (l) == 2 ? %d : (l) == 3 ? %d : %d)' % tuple(nblk))
    w('static const int h_cg_row_base[5] = {' + ', '.join(map(str, row_base)) + '};')
    w(f'static const unsigned short h_cg_row_start[{nrows + 1}] = {{' + ', '.join(map(str, row_start)) + '};')
    w(f'static const unsigned char h_cg_t_i1[{len(terms)}] = {{' + ', '.join(str(t[0]) for t in terms) + '};')
    w(f'static const unsigned char h_cg_t_i2[{len(terms)}] = {{' + ', '.join(str(t[1]) for t in terms) + '};')
    w(f'static const float h_cg_t_c[{len(terms)}] = {{' + ', '.join(f'{t[2]:.9e}f' for t in terms) + '};')
    w(f'static const unsigned short h_cgT_start[626] = {{' + ', '.join(map(str, t_start)) + '};')
    w(f'static const unsigned char h_cgT_l[{len(terms)}] = {{' + ', '.join(str(t[0]) for t in t_terms) + '};')
    w(f'static const unsigned char h_cgT_blk[{len(terms)}] = {{' + ', '.join(str(t[1]) for t in t_terms) + '};')
    w(f'static const float h_cgT_c[{len(terms)}] = {{' + ', '.join(f'{t[2]:.9e}f' for t in t_terms) + '};')
    w(f'static const unsigned short h_cgI_start[26] = {{' + ', '.join(map(str, inc_start)) + '};')
    w(f'static const unsigned int h_cgI_pk[{len(inc_flat)}] = {{' + ', '.join(str(e[0] | (e[1] << 16)) for e in inc_flat) + '};')
    w(f'static const float h_cgI_c[{len(inc_flat)}] = {{' + ', '.join(f'{e[2]:.9e}f' for e in inc_flat) + '};')
    w(f'static const unsigned short h_cgT_key[{len(terms)}] = {{' + ', '.join(map(str, t_keys)) + '};')
    w('static const int h_cg_slice_base[5] = {' + ', '.join(map(str, slice_base)) + '};')
    w('#define CG_ROWS_GMAX {' + ', '.join(map(str, rowS_gmax)) + '}')
    w('#define CG_ROWS_PART {' + ', '.join(map(str, rowS_part)) + '}')
    w(f'#define CG_ROWS_NGRP {len(rowS_gmax)}')
    w('#define CG_KEY_GMAX {' + ', '.join(map(str, key_gmax)) + '}')
    w(f'#define CG_KEY_NGRP {len(key_gmax)}')
    w(f'#define CG_NPAIRS {len(pair_perm)}')
    w('#define CG_PAIR_GMAX {' + ', '.join(map(str, pair_gmax)) + '}')
    w(f'#define CG_PAIR_NGRP {len(pair_gmax)}')
    w(f'#define CG_FW_SLOTS {sum(rowS_gmax)}')
    w(f'#define CG_FW_DUMP {FW_DUMP}')
    w(f'static const unsigned short h_cgFW_off[{len(fw_off)}] = {{' + ', '.join(map(str, fw_off)) + '};')
    w(f'static const float h_cgFW_c[{len(fw_c)}] = {{' + ', '.join(f'{c:.9e}f' for c in fw_c) + '};')
    w(f'static const unsigned int h_cgFW_pos[{len(fw_pos)}] = {{' + ', '.join(map(str, fw_pos)) + '};')
    w(f'static const unsigned int h_cgFW_lbm[{len(fw_lbm)}] = {{' + ', '.join(map(str, fw_lbm)) + '};')
    w('#define CG_ROWS_SLOT0 {' + ', '.join(str(sum(rowS_gmax[:g])) for g in range(len(rowS_gmax))) + '}')
    w(f'#define CG_POS_DUMP {POS_DUMP}')
    w(f'#define CG_BWD_LD {LDB}')
    w(f'#define CG_BK_SLOTS {sum(key_gmax)}')
    w(f'#define CG_BP_SLOTS {sum(pair_gmax)}')
    w(f'static const unsigned short h_cgBK_off[{len(bk_off)}] = {{' + ', '.join(map(str, bk_off)) + '};')
    w(f'static const float h_cgBK_c[{len(bk_c)}] = {{' + ', '.join(f'{c:.9e}f' for c in bk_c) + '};')
    w(f'static const unsigned int h_cgBK_pos[{len(bk_pos)}] = {{' + ', '.join(map(str, bk_pos)) + '};')
    w(f'static const unsigned short h_cgBP_off[{len(bp_off)}] = {{' + ', '.join(map(str, bp_off)) + '};')
    w(f'static const float h_cgBP_c[{len(bp_c)}] = {{' + ', '.join(f'{c:.9e}f' for c in bp_c) + '};')
    w(f'static const unsigned int h_cgBP_pos[{len(bp_pos)}] = {{' + ', '.join(map(str, bp_pos)) + '};')
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'cg_tables.inc')
    text = '\n'.join(out) + '\n'
    if os.path.exists(path) and open(path).read() == text:
        return  # unchanged: keep the mtime so the library is not rebuilt
    with open(path, 'w') as fh:
        fh.write(text)
    print(f'wrote {path}: nnz={len(terms)} rows={nrows} nblk={nblk}', file=sys.stderr)


if __name__ == '__main__':
    main()
