"""Per-kernel matrix-pipe utilisation from a rocprofv3 PMC pass (tools/pmc_mfma.sh).

usage: python tools/pmc_mfma_summary.py <results.db> <out.txt> [<out.json>]

MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8) * CUs * 4): the fraction of SIMD-cycles of the dispatch in which the MFMA
ALU was busy.  SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs (checked: the dense forward product reports 184 M = 3 x 62.9 GF /
32768 flops x 32 cycles); GRBM_GUI_ACTIVE is summed over the 8 XCDs (checked against the kernel-trace durations: 2.82 M / 8 = 353 k
cycles = 147 us at 2.4 GHz for launches that take 142 us untraced) -- hence the / 8.  PMC mode
serialises dispatches, so rings and side-stream products are measured ALONE here, not overlapped as in the step).  MOPS_* = MFMA
operations by input type in units of 512 flops (the hardware counter's unit on gfx94x/gfx950)."""
import json
import os
import re
import os as _os
import sqlite3
import sys
sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))

CUS = 256
XCDS = 8
NAMES = ['SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CU_CYCLES', 'SQ_INSTS_VALU_MFMA_MOPS_F16', 'SQ_INSTS_VALU_MFMA_MOPS_BF16',
         'SQ_INSTS_VALU_MFMA_MOPS_F32', 'GRBM_GUI_ACTIVE']


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    wg = 'workgroup_size_x' if 'workgroup_size_x' in cols else '256'
    did = 'dispatch_id' if 'dispatch_id' in cols else 'rowid'
    rows = db.execute("select %s, kernel_name, grid_size_x, %s, counter_name, value from counters_collection" % (did, wg)).fetchall()
    from pmc_summary import step_label
    disp = {}
    pos = None
    for d, name, gx, wx, cn, v in sorted(rows, key=lambda r: r[0]):
        if d not in disp:
            m = re.search(r'gemm_(f32|x6|x3)_kernel<[^>]*>|gemm_ps_kernel', name)
            if m is None:
                key = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60]
            else:
                # a product launch is named by its position in the step's product sequence (tools/pmc_summary.py: STEP_SEQUENCE)
                what, pos = step_label(m.group(0).split('kernel')[1] or '<ps>', pos)
                key = '%s grid=%d%s' % (m.group(0), int(gx) // max(1, int(wx)), (' [' + what + ']') if what else '')
            disp[d] = {'key': key}
        e = disp[d]
        e[cn] = e.get(cn, 0.0) + float(v)
    agg = {}
    for e in disp.values():
        a = agg.setdefault(e['key'], {'calls': 0})
        a['calls'] += 1
        for n in NAMES:
            a[n] = a.get(n, 0.0) + e.get(n, 0.0)
    lines = ['# rocprofv3 --pmc %s --kernel-trace (one pass; dispatches serialised by the profiler)' % ' '.join(NAMES),
             '# MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * %d CUs * 4 SIMDs); Mops = MFMA ops per launch in units of 512 flops' % CUS,
             '%-110s %6s %9s %12s %12s %12s %12s' % ('kernel', 'calls', 'MfmaUtil', 'busy_Mcyc', 'Mops_f16', 'Mops_bf16', 'Mops_f32')]
    js = {}
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)):
        act = a.get('GRBM_GUI_ACTIVE', 0.0)
        util = a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (act / XCDS * CUS * 4) if act else 0.0
        c = a['calls']
        lines.append('%-110s %6d %9.3f %12.2f %12.0f %12.0f %12.0f' % (k, c, util, a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / c / 1e6,
                                                                   a.get('SQ_INSTS_VALU_MFMA_MOPS_F16', 0.0) / c, a.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0.0) / c,
                                                                   a.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0.0) / c))
        js[k] = {'calls': c, 'mfma_util': round(util, 4), 'gui_active_cycles_per_launch_per_xcd': round(act / c / XCDS),
                 'mfma_busy_cycles_per_launch': round(a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / c)}
    open(sys.argv[2], 'w').write('\n'.join(lines[:40]) + '\n')
    if len(sys.argv) > 3:
        js['_meta'] = {'commit': os.environ.get('AMS_COMMIT', 'unknown')}
        json.dump(js, open(sys.argv[3], 'w'), indent=1, sort_keys=True)
    print('\n'.join(lines[:24]))


if __name__ == '__main__':
    main()
