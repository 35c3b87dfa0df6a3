"""Clebsch-Gordan coefficients against an independent implementation.

`oracle/so3.py::clebsch_gordan` and `molgym_amd/csrc/gen_tables.py::cg` are the same Racah sum typed twice; a
common-mode slip would pass every oracle-vs-kernel test.  Here both -- the oracle function over its whole domain
and every entry of the GENERATED kernel tables (forward CSR, transposed lists, incidence lists) -- are checked
against sympy.physics.wigner.clebsch_gordan (exact rational arithmetic), including completeness: every non-zero
coefficient with l <= 4 must be present exactly once."""
import math
import os
import re

import numpy as np
import pytest

sympy = pytest.importorskip('sympy')
from sympy.physics.wigner import clebsch_gordan as cg_sympy  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAXL = 4
_CACHE = {}


def exact(l1, m1, l2, m2, l, m):
    key = (l1, m1, l2, m2, l, m)
    if key not in _CACHE:
        _CACHE[key] = float(cg_sympy(l1, l2, l, m1, m2, m))
    return _CACHE[key]


def test_oracle_clebsch_gordan_matches_sympy():
    from oracle.so3 import clebsch_gordan
    worst, nonzero = 0.0, 0
    for l1 in range(MAXL + 1):
        for l2 in range(MAXL + 1):
            for l in range(abs(l1 - l2), l1 + l2 + 1):
                for m1 in range(-l1, l1 + 1):
                    for m2 in range(-l2, l2 + 1):
                        m = m1 + m2
                        if abs(m) > l:
                            assert clebsch_gordan(l1, m1, l2, m2, l, m) == 0.0
                            continue
                        want = exact(l1, m1, l2, m2, l, m)
                        worst = max(worst, abs(clebsch_gordan(l1, m1, l2, m2, l, m) - want))
                        nonzero += want != 0.0
    assert worst < 1e-14, worst
    assert nonzero > 2000  # the full l <= l1 + l2 <= 8 domain


def _tables():
    text = open(os.path.join(ROOT, 'molgym_amd', 'csrc', 'cg_tables.inc')).read()
    out = {}
    for name, body in re.findall(r'static const [\w ]+ (h_\w+)\[\d+\] = \{([^}]*)\};', text):
        vals = [v.strip().rstrip('f') for v in body.split(',')]
        out[name] = np.array([float(v) for v in vals])
    return out


def _lm(i):
    l = int(math.isqrt(i))
    return l, i - l * l - l


def _blocks(l):
    return [(l1, l2) for l1 in range(MAXL + 1) for l2 in range(MAXL + 1) if abs(l1 - l2) <= l <= min(l1 + l2, MAXL)]


def test_generated_kernel_tables_match_sympy():
    import subprocess
    import sys
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'molgym_amd', 'csrc', 'gen_tables.py')])
    T = _tables()
    nblk = [len(_blocks(l)) for l in range(MAXL + 1)]
    assert list(T['h_cg_nblk'].astype(int)) == nblk == [5, 12, 16, 17, 15]
    row_start = T['h_cg_row_start'].astype(int)
    i1s, i2s, cs = T['h_cg_t_i1'].astype(int), T['h_cg_t_i2'].astype(int), T['h_cg_t_c']
    seen = {}
    row = 0
    for l in range(MAXL + 1):
        assert int(T['h_cg_row_base'][l]) == row
        for blk, (l1, l2) in enumerate(_blocks(l)):
            for m in range(-l, l + 1):
                for q in range(row_start[row], row_start[row + 1]):
                    a, b = _lm(i1s[q]), _lm(i2s[q])
                    assert a[0] == l1 and b[0] == l2 and a[1] + b[1] == m, (row, q)
                    want = exact(l1, a[1], l2, b[1], l, m)
                    assert abs(cs[q] - want) < 1e-7 * max(1.0, abs(want)), (l1, a[1], l2, b[1], l, m, cs[q], want)
                    key = (l1, a[1], l2, b[1], l)
                    assert key not in seen
                    seen[key] = (row, cs[q])
                row += 1
    assert row == len(row_start) - 1 == 375
    # completeness: every non-zero coefficient with l1, l2, l <= 4 is in the table
    expected = 0
    for l1 in range(MAXL + 1):
        for l2 in range(MAXL + 1):
            for l in range(abs(l1 - l2), min(l1 + l2, MAXL) + 1):
                for m1 in range(-l1, l1 + 1):
                    for m2 in range(-l2, l2 + 1):
                        if abs(m1 + m2) <= l and abs(exact(l1, m1, l2, m2, l, m1 + m2)) > 1e-14:
                            expected += 1
                            assert (l1, m1, l2, m2, l) in seen
    assert expected == len(cs) == 1392
    # transposed lists: key (i1, i2) -> (l, block position, c)
    t_start = T['h_cgT_start'].astype(int)
    tl, tb, tc, tk = T['h_cgT_l'].astype(int), T['h_cgT_blk'].astype(int), T['h_cgT_c'], T['h_cgT_key'].astype(int)
    for key in range(625):
        (l1, m1), (l2, m2) = _lm(key // 25), _lm(key % 25)
        for q in range(t_start[key], t_start[key + 1]):
            assert tk[q] == key
            assert _blocks(tl[q])[tb[q]] == (l1, l2)
            assert abs(tc[q] - exact(l1, m1, l2, m2, tl[q], m1 + m2)) < 1e-7
    assert t_start[625] == 1392
    # incidence lists: index i -> (output row, partner, c), two entries per term
    i_start, pk, ic = T['h_cgI_start'].astype(int), T['h_cgI_pk'].astype(np.int64), T['h_cgI_c']
    assert i_start[25] == 2 * 1392
    rows_of = {v[0]: None for v in seen.values()}
    for i in range(25):
        for q in range(i_start[i], i_start[i + 1]):
            r, partner = int(pk[q]) & 0xffff, int(pk[q]) >> 16
            assert r in rows_of
            hits = [t for t in range(row_start[r], row_start[r + 1])
                    if (i1s[t], i2s[t]) in ((i, partner), (partner, i)) and abs(cs[t] - ic[q]) < 1e-12]
            assert hits, (i, q)


def _slice_offsets():
    """(l, block position, m) -> position of the AGGREGATE entry inside one channel's slice of an atom's concatenated
    rows: for l, for m: [ag blocks | in | sq blocks] (2 nblk_l + 1 complex per row), as cg_mfma.inc lays it out"""
    nblk = [len(_blocks(l)) for l in range(MAXL + 1)]
    out, base = {}, 0
    for l in range(MAXL + 1):
        w = 2 * nblk[l] + 1
        for mi in range(2 * l + 1):
            for bp in range(nblk[l]):
                out[(l, bp, mi - l)] = (base + mi * w + bp, nblk[l] + 1)
        base += (2 * l + 1) * w
    assert base == 775
    return out


def test_resolved_adjoint_tables_match_sympy():
    """The lane-major {offset, coefficient} tables the backward CG kernel keeps in LDS (gen_tables.py `resolved`):
    every entry (x, y) of the adjoint moment matrix must gather exactly the terms
    dG[x][y] = sum_l <l1 m1 l2 m2 | l m1+m2> d ag[l, block(l1, l2), m1+m2] (aggregate block) and the symmetrised
    S[x][y] = dP[x][y] + dP[y][x] (power block, diagonal counted twice), each pair of entries exactly once."""
    T = _tables()
    text = open(os.path.join(ROOT, 'molgym_amd', 'csrc', 'cg_tables.inc')).read()
    gmax = {k: [int(v) for v in re.search(r'#define %s \{([^}]*)\}' % k, text).group(1).split(',')]
            for k in ('CG_KEY_GMAX', 'CG_PAIR_GMAX')}
    off_of = _slice_offsets()
    LD = int(re.search(r'#define CG_BWD_LD (\d+)', text).group(1))  # row stride of the adjoint kernel's matrix (54 since round 6)
    assert LD >= 52 and LD % 4 == 2
    dump = int(re.search(r'#define CG_POS_DUMP (\d+)', text).group(1))
    assert dump >= 26 * LD  # outside the matrix (and its zero row)

    def want_key(x, y, power):
        (l1, m1), (l2, m2) = _lm(x), _lm(y)
        out = {}
        for l in range(abs(l1 - l2), min(l1 + l2, MAXL) + 1):
            if abs(m1 + m2) > l:
                continue
            c = exact(l1, m1, l2, m2, l, m1 + m2)
            if abs(c) > 1e-14:
                o, dist = off_of[(l, _blocks(l).index((l1, l2)), m1 + m2)]
                out[o + (dist if power else 0)] = out.get(o + (dist if power else 0), 0.0) + c
        return out

    def lanes(prefix, gm):
        off, cf, pos = T[prefix + '_off'].astype(int), T[prefix + '_c'], T[prefix + '_pos'].astype(np.int64)
        assert len(off) == len(cf) == 64 * sum(gm) and len(pos) == 64 * len(gm)
        slot = 0
        for g, n in enumerate(gm):
            for t in range(64):
                got = {}
                for j in range(n):
                    o, c = off[(slot + j) * 64 + t], cf[(slot + j) * 64 + t]
                    if c != 0.0:
                        got[o] = got.get(o, 0.0) + c
                yield int(pos[g * 64 + t]), got
            slot += n

    seen = set()
    for pos, got in lanes('h_cgBK', gmax['CG_KEY_GMAX']):
        if pos == dump:
            assert not got
            continue
        x, y = pos // LD, (pos % LD) // 2
        assert pos == x * LD + 2 * y and (x, y) not in seen and x < 25 and y < 25
        seen.add((x, y))
        want = want_key(x, y, False)
        assert set(got) == set(want) and all(abs(got[o] - want[o]) < 1e-7 for o in want), (x, y)
    assert len(seen) == 625
    seen = set()
    for pos, got in lanes('h_cgBP', gmax['CG_PAIR_GMAX']):
        if pos == (dump | (dump << 16)):
            assert not got
            continue
        p1, p2 = pos & 0xffff, pos >> 16
        x, y = p1 // LD, (p1 % LD) // 2
        assert p1 == x * LD + 2 * y and p2 == y * LD + 2 * x and x <= y and (x, y) not in seen
        seen.add((x, y))
        want = want_key(x, y, True)
        for o, c in want_key(y, x, True).items():
            want[o] = want.get(o, 0.0) + c   # x == y: the same terms again, i.e. doubled
        assert set(got) == set(want) and all(abs(got[o] - want[o]) < 1e-7 for o in want), (x, y)
    assert len(seen) == 325


def test_resolved_forward_table_matches_sympy():
    """The lane-major {moment index, coefficient} table of the forward projection: every output row (l, block, m) of the
    concatenated channels gathers exactly sum_{m1 + m2 = m} <l1 m1 l2 m2 | l m> G[(l1, m1)][(l2, m2)], lands at its slice
    position (aggregate) / nblk_l + 1 further (power), every row exactly once."""
    T = _tables()
    text = open(os.path.join(ROOT, 'molgym_amd', 'csrc', 'cg_tables.inc')).read()
    gmax = [int(v) for v in re.search(r'#define CG_ROWS_GMAX \{([^}]*)\}', text).group(1).split(',')]
    part = [int(v) for v in re.search(r'#define CG_ROWS_PART \{([^}]*)\}', text).group(1).split(',')]
    dump = int(re.search(r'#define CG_FW_DUMP (\d+)', text).group(1))
    off, cf, pos = T['h_cgFW_off'].astype(int), T['h_cgFW_c'], T['h_cgFW_pos'].astype(np.int64)
    assert len(off) == len(cf) == 64 * sum(gmax) and len(pos) == 64 * len(gmax)
    LD, part_base = 52, [0, 251, 496]
    where = {}   # slice position of the aggregate entry -> (l, block position, m, nblk + 1)
    for (l, bp, m), (o, dist) in _slice_offsets().items():
        where[o] = (l, bp, m, dist)
    seen, slot = set(), 0
    for g, n in enumerate(gmax):
        for t in range(64):
            p = int(pos[g * 64 + t])
            got = {}
            for j in range(n):
                o, c = off[(slot + j) * 64 + t], cf[(slot + j) * 64 + t]
                if c != 0.0:
                    got[o] = got.get(o, 0.0) + c
            if p == (dump | (dump << 16)):
                assert not got and int(T['h_cgFW_lbm'][g * 64 + t]) == 0xffff
                continue
            ag = (p & 0xffff) + part_base[part[g]]
            l, bp, m, dist = where[ag]
            lbm = int(T['h_cgFW_lbm'][g * 64 + t])  # the same row as (l, block position, m index): the heads' mixer uses it
            assert (lbm & 7, (lbm >> 3) & 31, lbm >> 8) == (l, bp, m + l)
            assert (p >> 16) + part_base[part[g]] == ag + dist and ag not in seen
            assert (l <= 2 and part[g] == 0) or l - 2 == part[g]
            seen.add(ag)
            l1, l2 = _blocks(l)[bp]
            want = {}
            for m1 in range(-l1, l1 + 1):
                m2 = m - m1
                if abs(m2) <= l2:
                    c = exact(l1, m1, l2, m2, l, m)
                    if abs(c) > 1e-14:
                        want[(l1 * l1 + m1 + l1) * LD + 2 * (l2 * l2 + m2 + l2)] = c
            assert set(got) == set(want) and all(abs(got[o] - want[o]) < 1e-7 for o in want), (l, bp, m)
        slot += n
    assert len(seen) == 375
