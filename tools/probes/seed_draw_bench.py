"""Host timing of the reference k-means seed draw (csrc/host/mt_choice.c) against numpy's own calls, at the benchmark shape
(640 rows = 64 mixtures x 10 tries, l = 20480 bins, C = 2).  No GPU involved."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'adaptive-multispeaker-separation_amd'))


def one():
    import types
    from ams_hip import kmeans_host as kh
    ns = types.SimpleNamespace(nb_clusters=2, seeding='reference')
    kh._REFERENCE_SEEDS.ahead = False
    kh.KMeans._draw(ns, 64, 20480)
    ts = []
    for _ in range(10):
        t = time.time()
        kh.KMeans._draw(ns, 640, 20480)
        ts.append(time.time() - t)
    print('%-44s %.2f ms per batch of 640 rows (median of 10)' % (os.environ.get('LABEL', ''), 1e3 * sorted(ts)[5]))


if __name__ == '__main__':
    if len(sys.argv) > 1:
        one()
        sys.exit(0)
    t = time.time()
    for _ in range(640):
        np.random.choice(20480, size=2, replace=False)
    print('%-44s %.2f ms per batch of 640 rows' % ('numpy, one choice() per row', 1e3 * (time.time() - t)))
    for label, env in (('C, scalar', dict(AMS_MT_CHOICE_SCALAR='1')), ('C, AVX2 (default)', {})):
        subprocess.run([sys.executable, os.path.abspath(__file__), 'one'], env=dict(os.environ, LABEL=label, **env))
