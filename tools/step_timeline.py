"""Print one training step's kernel timeline from a rocprofv3 results db (debug aid).

usage: python tools/step_timeline.py <results.db> [step_index_from_end=2] [--all]
LSTM step kernels are collapsed into one line per recurrence (span, count, mean duration, mean gap).
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    back = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith('-') else 2
    c = db.cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if 'kernel_dispatch' in t][0]
    ks = [t for t in tabs if 'kernel_symbol' in t][0]
    rows = c.execute("select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.stream_id from %s d join %s s "
                     "on d.kernel_id=s.id order by d.start" % (kd, ks)).fetchall()
    idx = [i for i, r in enumerate(rows) if 'amsgrad' in r[0] or 'rmsprop' in r[0] or 'momentum' in r[0]]
    if len(idx) <= back:                                       # no optimizer (inference): a step ends with its last overlap-add / k-means select
        for anchor in ('overlap_add_kernel', 'kmeans_select_kernel'):
            idx = [i for i, r in enumerate(rows) if anchor in r[0]]
            if len(idx) > back:
                break
    a, b = idx[-back - 1], idx[-back]
    t0 = rows[a][2]
    print('# step span %.1f us' % ((rows[b][2] - t0) / 1e3))
    run = []

    def flush():
        if not run:
            return
        durs = [(r[2] - r[1]) / 1e3 for r in run]
        gaps = [(run[i + 1][1] - run[i][2]) / 1e3 for i in range(len(run) - 1)]
        print('%9.1f %8.1f  %-40s n=%d mean_dur=%.2f mean_gap=%.2f' % (
            (run[0][1] - t0) / 1e3, (run[-1][2] - run[0][1]) / 1e3, 'lstm_step x', len(run), sum(durs) / len(durs),
            sum(gaps) / max(1, len(gaps))))
        run.clear()

    for r in rows[a + 1:b + 1]:
        n = r[0]
        if 'lstm_step' in n:
            if run and (('fwd' in n) != ('fwd' in run[-1][0])):
                flush()
            run.append(r)
            continue
        if r[5] == (run[0][5] if run else None):
            flush()
        short = n.split('(')[0].replace('(anonymous namespace)::', '').replace('void ', '')
        if '_ZN' in short:
            short = short[short.find('_N_') + 5:][:44]
        if 'gemm_f32' in n:
            short = n[n.index('gemm_f32'):n.index('gemm_f32') + 24]
        print('%9.1f %8.1f  %-44s grid=%s,%s st=%s' % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, short[:44], r[3], r[4], r[5]))
    flush()
    # idle accounting: time inside the step during which NO kernel was running, and the gaps that make it up
    seg = sorted((r[1], r[2], r[0]) for r in rows[a + 1:b + 1])
    idle, cur_end, prev_name, gaps = 0.0, seg[0][0], '', []
    for st, en, name in seg:
        if st > cur_end:
            g = (st - cur_end) / 1e3
            idle += g
            if g >= 3.0:
                gaps.append((g, (cur_end - t0) / 1e3, prev_name, name))
        if en > cur_end:
            cur_end, prev_name = en, name
    print('# idle (no kernel running): %.1f us in %d gaps >= 3 us' % (idle, len(gaps)))
    for g, at, pn, nn in sorted(gaps, reverse=True)[:25]:
        sh = lambda n: (n[n.index('gemm_f32'):n.index('gemm_f32') + 24] if 'gemm_f32' in n else n.split('(')[0].replace('(anonymous namespace)::', '').replace('void ', ''))[:40]  # noqa: E731
        print('#   %6.1f us at %8.1f  after %-40s before %s' % (g, at, sh(pn), sh(nn)))


if __name__ == '__main__':
    main()
