"""Offline: LDS bank-conflict cycles of the gathers of the resolved CG tables (ds_read_b64: lane groups {0-31}, {32-63}, bank =
dword address mod 64, identical addresses broadcast, every further distinct address on a busy bank costs a cycle).
usage: python tools/lds_conflicts.py [cg_tables.inc]"""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else 'molgym_amd/csrc/cg_tables.inc'
text = open(path).read()


def arr(name):
    m = re.search(name + r'\[\d+\] = \{([^}]*)\}', text)
    return [int(float(x)) if '.' not in x and 'e' not in x else float(x.rstrip('f')) for x in m.group(1).split(',')]


def macro(name):
    m = re.search(r'#define ' + name + r' \{([^}]*)\}', text)
    return [int(x) for x in m.group(1).split(',')]


def cycles(offs_dw):
    """offs_dw: 64 dword offsets of 8-byte reads -> LDS cycles of the instruction (2 when conflict-free)"""
    tot = 0
    for grp in (offs_dw[:32], offs_dw[32:]):
        banks = {}
        for o in grp:
            for b in (o % 64, (o + 1) % 64):
                banks.setdefault(b, set()).add(o)
        tot += max(len(v) for v in banks.values())
    return tot


def table(name_off, gmax, to_dw):
    off = arr(name_off)
    slot = 0
    tot = ideal = 0
    for gm in gmax:
        for j in range(gm):
            row = off[(slot + j) * 64:(slot + j + 1) * 64]
            tot += cycles([to_dw(o) for o in row])
            ideal += 2
        slot += gm
    return tot, ideal


if __name__ == '__main__':
    for nm, arrn, gm, f in (('forward projection (sG / sP gathers)', 'h_cgFW_off', 'CG_ROWS_GMAX', lambda o: o),
                            ('adjoint, aggregate block', 'h_cgBK_off', 'CG_KEY_GMAX', lambda o: 2 * o),
                            ('adjoint, power block', 'h_cgBP_off', 'CG_PAIR_GMAX', lambda o: 2 * o)):
        t, i = table(arrn, macro(gm), f)
        print(f'{nm}: {t} LDS cycles for {i} conflict-free ({t / i:.2f}x)')
