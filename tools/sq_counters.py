#!/usr/bin/env python3
"""Per-kernel averages of arbitrary --pmc counters from one or more rocprofv3 databases (debug aid).
usage: sq_counters.py out.json a_results.db [b_results.db ...]"""
import json
import sqlite3
import sys

out = {}
for db in sys.argv[2:]:
    c = sqlite3.connect(db)
    for n, cn, cnt, tot in c.execute('select name, counter_name, count(*), sum(counter_value) from pmc_events group by name, counter_name'):
        out.setdefault(n.split('(')[0].replace('void ', ''), {})[cn] = tot / max(cnt, 1)
json.dump(out, open(sys.argv[1], 'w'), indent=1, sort_keys=True)
