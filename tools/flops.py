#!/usr/bin/env python3
"""Algorithmic FLOP count of one CovariantAC.step forward, per sample.

Counting convention (the figure `roofline.achieved` in bench.py is computed from): real flops,
complex MAC = 8, complex multiply = 6, CG projection counted DENSE (m1*m2*sum_l(2l+1)*4 per channel),
the aggregation over neighbours counted over n atoms.  `n` is the number of atoms the sums run over:
n = canvas_size gives the dense (padded) count the reference itself executes; n = real atoms of a
sample gives the ragged count the HIP kernels execute.  fwd+bwd = 3 x forward.
"""
import sys

L, C, CE, W, G = 4, 10, 4, 128, 3
NBLK = [5, 12, 16, 17, 15]


def forward_flops(n, num_zs, terms=False):
    co = num_zs * CE
    lev_cout = [C, C, co]
    M = [2 * l + 1 for l in range(L + 1)]
    t = {}
    t['ylm'] = n * n * sum((2 * l - 1) * 3 * 6 + (2 * l - 1) * 3 * (2 * l + 1) * 4 for l in range(2, L + 1))
    t['radial'] = n * n * 3 * ((L + 1) * 2 * 32 * 2 * C + 640)
    # edge: dot products + channel mix
    tau_atom = [[C], [C] * 5, [C] * 5]
    edge = 0
    for k in range(3):
        edge += n * n * sum(tt * M[l] * 8 for l, tt in enumerate(tau_atom[k]))
        c_prev = 0 if k == 0 else C
        c_in = c_prev + sum(tau_atom[k]) + C
        edge += n * n * (L + 1) * c_in * C * 8
    t['edge'] = edge
    # CG aggregate (kron over neighbours + dense projection) and CG power
    ag = sq = mix = 0
    for k in range(3):
        parts = tau_atom[k]
        for l1 in range(L + 1):
            for l2 in range(len(parts)):
                lo, hi = abs(l1 - l2), min(l1 + l2, L)
                if lo > hi:
                    continue
                proj = C * M[l1] * M[l2] * sum(M[l] for l in range(lo, hi + 1)) * 4
                ag += n * (n * C * M[l1] * M[l2] * 8 + proj)
        for l1 in range(len(parts)):
            for l2 in range(len(parts)):
                lo, hi = abs(l1 - l2), min(l1 + l2, L)
                if lo > hi:
                    continue
                proj = C * M[l1] * M[l2] * sum(M[l] for l in range(lo, hi + 1)) * 4
                sq += n * (C * M[l1] * M[l2] * 6 + proj)
        tau_cat = [3 * C, C, C, C, C] if k == 0 else [C * (2 * b + 1) for b in NBLK]
        mix += n * sum(tau_cat[l] * lev_cout[k] * M[l] * 8 for l in range(L + 1))
    t['cg_aggregate'], t['cg_power'], t['atom_mix'] = ag, sq, mix
    nlat, nlat_e = (L + 2) * co * 2, (L + 2) * CE * 2
    heads = n * (2 * nlat * W + 2 * W * 1) + n * (2 * nlat * W + 2 * W * W)  # focus, trans per atom
    heads += (2 * nlat * W + 2 * W * num_zs) + (2 * nlat_e * W + 2 * W * 2 * G) + (2 * W * W + 2 * W)
    heads += n * co * 25 * 6
    t['heads'] = heads
    tau_m = [CE * (b + 2) for b in NBLK]
    mixer = CE * 25 * 2
    for l1 in range(L + 1):
        for l2 in range(L + 1):
            lo, hi = abs(l1 - l2), min(l1 + l2, L)
            mixer += CE * M[l1] * M[l2] * 6 + CE * M[l1] * M[l2] * sum(M[l] for l in range(lo, hi + 1)) * 4
    mixer += sum(tau_m[l] * CE * M[l] * 8 for l in range(L + 1))
    t['mixer'] = mixer
    t['so3_logz'] = 1730 * (25 * CE * 8 + 200)
    total = sum(t.values())
    return (total, t) if terms else total


def step_flops(natoms_list, num_zs):
    """fwd+bwd flops of a mini-batch with the given real atom counts (ragged convention)."""
    return 3 * sum(forward_flops(int(n), num_zs) for n in natoms_list)


if __name__ == '__main__':
    for n, z in ((7, 3), (12, 5), (20, 5), (40, 5)):
        tot, terms = forward_flops(n, z, terms=True)
        print(f'N={n} Z={z}: fwd {tot / 1e6:.1f} MFLOP, fwd+bwd {3 * tot / 1e6:.1f} MFLOP  ' +
              ' '.join(f'{k}={v / 1e6:.2f}' for k, v in terms.items()))
