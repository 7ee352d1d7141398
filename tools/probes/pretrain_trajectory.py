"""Per-step cost of the cfg2 pre-training step (path A, B=64, full geometry), eager vs hipGraph replay, same seeds and batches.
Answers VERDICT r01: `pretraining_A_graph` reported cost 2.38e7 vs 3.9e3 eager after a different number of steps -- replay bug or
an SDR-term blow-up of the recipe itself (lr 1e-3 AMSGrad on `mix * (non_mix / mix)` masks + sdr loss)?

    python tools/probes/pretrain_trajectory.py [--steps 20]  ->  one JSON line per mode with the whole trajectory
"""
import argparse
import contextlib
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in (ROOT, os.path.join(ROOT, 'adaptive-multispeaker-separation_amd'), os.path.join(ROOT, 'tools')):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def run(graph, steps, B):
    import torch
    import utils.ops
    from bench_configs import _args
    from utils.trainer import Adapt_Pretrainer
    utils.ops.rng.seed(42)
    L = 20480
    a = _args(window_size=1024, filters=256, max_pool=256, hop_size=256, chunk_size=L, batch_size=B, nb_speakers=2, loss='sdr+l2',
              separation='mask', beta=0.0, regularization=0.0, overlap_coef=1.0, learning_rate=1e-3, hip_graph=graph)
    a.pop('type')
    tr = Adapt_Pretrainer(**a)
    dist, tfds = tr.prepare()
    costs = []
    with tr.graph.as_default():
        feed = {tfds.handle: tfds.get_handle(tfds.TRAIN), tfds.chunk_size: L}
        tfds.initialize(tfds.TRAIN)
        for i in range(steps):
            costs.append(float(tr.model.train(feed, i)))
    torch.cuda.synchronize()
    return costs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--batch', type=int, default=64)
    args = ap.parse_args()
    os.environ.setdefault('AMS_LOG_DIR', tempfile.mkdtemp(prefix='ams_pt_log_'))
    out = {}
    for graph in (False, True):
        with contextlib.redirect_stdout(sys.stderr):
            out['graph' if graph else 'eager'] = run(graph, args.steps, args.batch)
    e, g = out['eager'], out['graph']
    out['max_rel_diff'] = max(abs(x - y) / max(abs(x), 1e-30) for x, y in zip(e, g))
    print(json.dumps(out))


if __name__ == '__main__':
    main()
