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


def family_flops(natoms_list, num_zs, fused_small=False):
    """Algorithmic flops of ONE fwd+bwd step (ragged: real atoms only) per kernel FAMILY -- the names are the library's
    live-timing spans (mg_profile_report).  Every dense product is counted once forward, once for the adjoint w.r.t. its
    input and once for its weight gradient (`k_gemm_dw`: always the deferred GEMM launches); the heads' dense layers run inside
    `k_heads_fwd` / `k_heads_bwd`.  Where the forward products and input adjoints run depends on the path:
    fused_small (the SF6-sized mini-batches: level0.inc / edge_level.inc): `k_level0` = radial Linears of all levels + DotMatrix
    and both mixes of level 0, `k_edge_level` = DotMatrix + edge mixes of levels 1, 2, `k_gemm_rows` = the atom cat-mixes of
    levels 1, 2 only; otherwise everything is in `k_gemm_rows` (row / column / shared-input GEMM launches) and `k_dot`."""
    co = num_zs * CE
    M = [2 * l + 1 for l in range(L + 1)]
    nlat, nlat_e = (L + 2) * co * 2, (L + 2) * CE * 2
    out = dict(k_gemm_rows=0.0, k_gemm_dw=0.0, k_heads_fwd=0.0, k_heads_bwd=0.0, k_dot=0.0, k_level0=0.0, k_edge_level=0.0)
    for n in natoms_list:
        n = int(n)
        radial = n * n * 3 * (L + 1) * 2 * 32 * 2 * C
        edge_mix, dot, atom_mix = [0, 0, 0], [0, 0, 0], [0, 0, 0]
        for k in range(3):
            parts = [C] if k == 0 else [C] * 5
            dot[k] = n * n * sum(tt * M[l] * 8 for l, tt in enumerate(parts))
            c_in = (0 if k == 0 else C) + sum(parts) + C
            edge_mix[k] = n * n * (L + 1) * c_in * C * 8
            tau_cat = [3 * C, C, C, C, C] if k == 0 else [C * (2 * b + 1) for b in NBLK]
            atom_mix[k] = n * sum(tau_cat[l] * (co if k == 2 else C) * M[l] * 8 for l in range(L + 1))
        enc = radial + sum(edge_mix) + sum(atom_mix)
        per_atom_mlp = n * (2 * nlat * W + 2 * W) + n * (2 * nlat * W + 2 * W * W)
        per_sample_mlp = (2 * nlat * W + 2 * W * num_zs) + (2 * nlat_e * W + 2 * W * 2 * G) + (2 * W * W + 2 * W)
        mixer_mix = sum(CE * (b + 2) * CE * M[l] * 8 for l, b in enumerate(NBLK))
        mixer_cg = CE * 25 * 2 + sum(CE * M[a] * M[b] * 6 + CE * M[a] * M[b] * sum(M[l] for l in range(abs(a - b), min(a + b, L) + 1)) * 4
                                     for a in range(L + 1) for b in range(L + 1))
        head_dense = per_atom_mlp + per_sample_mlp + mixer_mix
        head_other = n * co * 25 * 6 + mixer_cg + 1730 * (25 * CE * 8 + 200)
        if fused_small:
            out['k_level0'] += 2 * (radial + edge_mix[0] + atom_mix[0]) + 3 * dot[0]
            out['k_edge_level'] += 2 * (edge_mix[1] + edge_mix[2]) + 3 * (dot[1] + dot[2])
            out['k_gemm_rows'] += 2 * (atom_mix[1] + atom_mix[2])
        else:
            out['k_gemm_rows'] += 2 * enc                  # forward + input adjoint
            out['k_dot'] += 3 * sum(dot)                   # forward + two input adjoints
        out['k_gemm_dw'] += enc + head_dense               # every weight gradient
        out['k_heads_fwd'] += head_dense + head_other
        out['k_heads_bwd'] += head_dense + 2 * head_other  # input adjoints (the non-linear parts cost ~2x backward)
    return out


if __name__ == '__main__':
    for n, z in ((7, 3), (12, 5), (20, 5), (40, 5)):
        tot, terms = forward_flops(n, z, terms=True)
        print(f'N={n} Z={z}: fwd {tot / 1e6:.1f} MFLOP, fwd+bwd {3 * tot / 1e6:.1f} MFLOP  ' +
              ' '.join(f'{k}={v / 1e6:.2f}' for k, v in terms.items()))
