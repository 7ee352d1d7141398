"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command).

usage: python tools/pmc_summary.py <fetch_results.db> <write_results.db> <out.txt> [<out.json>] ["command description"]

Units/corrections (MI355X_MICROARCH.md, HBM section): both counters are KILOBYTES per dispatch; on gfx950 FETCH_SIZE tallies
the 128-byte fabric requests of wide coalesced reads at 64 B, so it is DOUBLED here; WRITE_SIZE is taken as reported
(uncalibrated).  Infinity-Cache hits are counted, so 'traffic' is memory-side L2 traffic, an upper bound on DRAM bytes.
"""
import json
import re
import sqlite3
import sys


# launch grid (workgroups) -> what that launch is in the B=64 front_DPCL step (bench.py): the product family runs many shapes under
# one kernel name, and traffic only means something per shape
# Round 5: persistent / stream-K grids make several shapes launch the same number of workgroups, so a launch is named by its POSITION in
# the step's product sequence (host issue order of an eager step = dispatch order; a step starts at the front conv, variant <2, 0, ...>):
STEP_SEQUENCE = ['front conv 15360x256x1024 (stream-K)', 'projection L0 5120x2400x256', 'projection L1 5120x2400x600',
                 'projection L2 5120x2400x600', 'dense fwd 5120x10240x600', 'dense dX 5120x600x10240', 'dense dW block 0 600x6656x5120 (capped)',
                 'LSTM dX L2 5120x600x2400', 'LSTM dWx L2 600x2400x5120 (capped)', 'LSTM dU L2 2x300x1200x5119 (capped)',
                 'dense dW block 1 600x1792x5120 (capped)', 'LSTM dX L1 5120x600x2400', 'LSTM dWx L1 600x2400x5120 (capped)',
                 'LSTM dU L1 2x300x1200x5119 (capped)', 'dense dW block 2 600x1792x5120 (capped)', 'LSTM dU L0 2x300x1200x5119',
                 'LSTM dWx L0 256x2400x5120']
# the variant each position is launched as (A loader, B loader, ..., two accumulator sets): a step whose sequence differs (the first
# pass of a model, bench.py's side-stream-off 'alone' step) is left unlabelled instead of mislabelled
# (round 6: the four forward products run from pre-split operand images, csrc/gemm_ps.hip: variant '<ps>')
STEP_VARIANT = ['<2, 0', '<ps>', '<ps>', '<ps>', '<ps>', '<0, 1', '<1, 0, 3, 0, false', '<0, 1', '<1, 0, 3, 0, false', '<1, 0, 3, 0, false',
                '<1, 0, 3, 0, false', '<0, 1', '<1, 0, 3, 0, false', '<1, 0, 3, 0, false', '<1, 0, 3, 0, false', '<1, 0, 3, 0, true', '<1, 0, 3, 0, true']


def step_label(variant, pos):
    """(label, next position) of a product launch of kernel variant `variant` ('<..>' template arguments) at sequence position pos."""
    if variant.startswith('<2, 0'):
        pos = 0
    if pos is None or pos >= len(STEP_SEQUENCE) or not variant.startswith(STEP_VARIANT[pos]):
        return '', None
    return STEP_SEQUENCE[pos], pos + 1


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    wg = 'workgroup_size_x' if 'workgroup_size_x' in cols else '256'
    did = 'dispatch_id' if 'dispatch_id' in cols else 'rowid'
    rows = db.execute("select kernel_name, grid_size_x, %s, sum(value), %s from counters_collection where counter_name=? group by %s order by %s"
                      % (wg, did, did, did), (counter,)).fetchall()
    out = {}
    pos = None
    for name, gx, wx, v, _ in rows:
        # the product family runs many shapes under one name: keep them apart by their position in the step (see STEP_SEQUENCE)
        m = re.search(r'gemm_(f32|x6|x3)_kernel<[^>]*>|gemm_ps_kernel', name)
        if m is None:
            key = name
        else:
            nwg = int(gx) // max(1, int(wx))
            what, pos = step_label(m.group(0).split('kernel')[1] or '<ps>', pos)
            key = '%s grid=%d%s' % (m.group(0), nwg, (' [' + what + ']') if what else '')
        out.setdefault(key, []).append(float(v))
    return out


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    if ' grid=' in n:
        return n[:120]
    return n.split('(')[0][:70]


def main():
    f = per_kernel(sys.argv[1], 'FETCH_SIZE')
    w = per_kernel(sys.argv[2], 'WRITE_SIZE')
    desc = sys.argv[5] if len(sys.argv) > 5 else ''
    names = sorted(set(f) | set(w), key=lambda n: -(2 * sum(f.get(n, [0])) + sum(w.get(n, [0]))))
    lines = ['# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only); MB per launch',
             '# read = 2 x FETCH_SIZE (gfx950 correction), write = WRITE_SIZE as reported; ' + desc,
             '%-120s %7s %12s %12s %12s' % ('kernel', 'calls', 'read_MB', 'write_MB', 'total_MB')]
    js = {}
    for n in names:
        fv, wv = f.get(n, []), w.get(n, [])
        calls = max(len(fv), len(wv))
        rd = 2.0 * sum(fv) / max(1, len(fv)) * 1024 / 1e6
        wr = sum(wv) / max(1, len(wv)) * 1024 / 1e6
        lines.append('%-120s %7d %12.3f %12.3f %12.3f' % (short(n), calls, rd, wr, rd + wr))
        js[short(n)] = {'calls': calls, 'read_MB_per_launch': round(rd, 4), 'write_MB_per_launch': round(wr, 4)}
    open(sys.argv[3], 'w').write('\n'.join(lines) + '\n')
    if len(sys.argv) > 4 and sys.argv[4]:
        import os
        js['_meta'] = {'commit': os.environ.get('AMS_COMMIT', 'unknown'), 'command': desc}
        json.dump(js, open(sys.argv[4], 'w'), indent=1, sort_keys=True)
    print('\n'.join(lines[:24]))


if __name__ == '__main__':
    main()
