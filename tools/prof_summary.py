#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite .db) as a text table for profiles/."""
import sqlite3
import sys


def main(db, out, note=''):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                          "group by name order by sum(duration) desc"))
    tot = float(sum(r[2] for r in rows))
    with open(out, 'w') as f:
        f.write('# rocprofv3 --kernel-trace --stats summary (durations in microseconds; source: %s)\n' % db.split('/')[-1])
        if note:
            f.write('# %s\n' % note)
        f.write('# total kernel time: %.3f ms\n' % (tot / 1e6))
        f.write('%-100s %8s %12s %10s %10s %10s %6s\n' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', 'pct'))
        for n, k, s, a, mn, mx in rows:
            f.write('%-100s %8d %12.1f %10.2f %10.2f %10.2f %6.2f\n' % (n[:100], k, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))


def write_json(db, out_json, commit):
    """Per-kernel call count and average duration (us) of the whole traced run, for bench.py's `roofline.replayed_region`."""
    import json
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, count(*), avg(duration) from kernels group by name order by sum(duration) desc"))
    js = {'_meta': {'commit': commit, 'source': 'rocprofv3 --kernel-trace of `python bench.py --no-cpu-baseline --no-secondary` (hipGraph replay)'}}
    for n, k, a in rows[:16]:
        key = n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:70]
        js[key] = {'calls': k, 'avg_us': round(a / 1e3, 2)}
    json.dump(js, open(out_json, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    import os
    main(sys.argv[1], sys.argv[2], ' '.join(sys.argv[3:]))
    if os.environ.get('AMS_PROF_JSON'):
        write_json(sys.argv[1], os.environ['AMS_PROF_JSON'], os.environ.get('AMS_COMMIT', 'unknown'))
