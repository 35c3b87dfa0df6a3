#!/usr/bin/env python3
"""tests/golden/test_agent_molecules.json: the three molecules the reference's own agent tests run on
(/root/reference/tests/agents/covariant/resources/{h2o,ch3,ch4}.xyz, read at test_agent.py:61-63,98-100,121-123)
as plain data -- symbols, atomic numbers, positions -- so that the same property tests can run on the GPU box, where
/root/reference does not exist.  Usage: python oracle/make_molecules.py  (build container only)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = '/root/reference/tests/agents/covariant/resources'
NUMBERS = {'H': 1, 'C': 6, 'O': 8}


def read_xyz(path):
    lines = open(path).read().splitlines()
    n = int(lines[0])
    atoms = [ln.split() for ln in lines[2:2 + n]]
    return {'symbols': [a[0] for a in atoms], 'numbers': [NUMBERS[a[0]] for a in atoms],
            'positions': [[float(x) for x in a[1:4]] for a in atoms]}


def main():
    out = {name: read_xyz(os.path.join(SRC, name + '.xyz')) for name in ('h2o', 'ch3', 'ch4')}
    # the fixed configuration of CovariantAgentTest.setUp (test_agent.py:23-41)
    out['setup'] = dict(canvas_size=5, zs=[0, 1, 6, 8], min_max_distance=[0.9, 1.8], network_width=64, bag_scale=1,
                        beta=100, maxl=4, num_cg_levels=3, num_channels_hidden=10, num_channels_per_element=4,
                        num_gaussians=3, formula=[[1, 1]])
    with open(os.path.join(ROOT, 'tests', 'golden', 'test_agent_molecules.json'), 'w') as fh:
        json.dump(out, fh, indent=1)


if __name__ == '__main__':
    main()
