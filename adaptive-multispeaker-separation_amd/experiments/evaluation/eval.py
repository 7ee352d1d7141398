# coding: utf-8
"""python -m experiments.evaluation.eval --model_folder ... --sortofmodel {STFT,front,pretraining}[_enhanced]_{DPCL,L41}
(reference experiments/evaluation/eval.py): run an inference trainer over the test split and accumulate the BSS-eval
improvement of the separated sources over the unprocessed mixture."""
import numpy as np

from models.L41 import L41Model
from models.dpcl import DPCL
from utils.bss_eval import bss_eval_sources_cupy
from utils.trainer import (MyArgs, STFT_Separator_Enhanced_Inference, STFT_Separator_Inference, Front_Separator_Enhanced_Inference,
                           Front_Separator_Inference, Pretrained_Inference)


def pick(sortofmodel):
    if 'STFT' in sortofmodel:
        inferencer = STFT_Separator_Enhanced_Inference if 'enhanced' in sortofmodel else STFT_Separator_Inference
    elif 'front' in sortofmodel:
        inferencer = Front_Separator_Enhanced_Inference if 'enhanced' in sortofmodel else Front_Separator_Inference
    elif 'pretraining' in sortofmodel:
        inferencer = Pretrained_Inference
    else:
        raise SystemExit(0)                                           # eval.py:34-35
    sep = L41Model if 'L41' in sortofmodel else (DPCL if 'DPCL' in sortofmodel else None)
    return inferencer, sep


def evaluate(batches, nsrc=2, verbose=True):
    """batches: iterable of (mix [B,L], non_mix [B,S,L], separated [B,S,L]) -> (mean improvements, per-utterance array)."""
    sdr = sir = sar = 0.0
    i = 0
    arr = []
    for mix, non_mix, separated in batches:
        for m, n_m, s in zip(list(mix), list(non_mix), list(separated)):
            # device tensors stay on the device (the metric kernels read them in place); numpy inputs are uploaded
            mix_stack = m.unsqueeze(0).expand(nsrc, -1) if hasattr(m, 'unsqueeze') else np.array([m] * nsrc)
            no_separation = bss_eval_sources_cupy(n_m, mix_stack, nsrc=nsrc)
            separation = bss_eval_sources_cupy(n_m, s, nsrc=nsrc)
            sdr_ = np.mean(separation[0] - no_separation[0])
            sir_ = np.mean(separation[1] - no_separation[1])
            sar_ = np.mean(separation[2] - no_separation[2])
            if not np.all(np.isfinite([sdr_, sir_, sar_])):            # eval.py:61-62
                continue
            arr.append((no_separation[0], separation[0]))
            sdr += sdr_
            sir += sir_
            sar += sar_
            i += 1
            if verbose:
                print(sdr / float(i), sir / float(i), sar / float(i))
    n = float(max(i, 1))
    return (sdr / n, sir / n, sar / n), np.array(arr)


if __name__ == '__main__':
    p = MyArgs()
    p.parser.add_argument('--model_folder', help='Path to the Model folder to load', required=True)
    p.parser.add_argument('--sortofmodel', help='Sort of model', required=True)
    p.parser.add_argument('--out', help='Use out-of-set dataset for testing', action="store_true")
    p.add_adapt_args()
    p.add_separator_args()
    args = p.get_args()
    inferencer, sep = pick(args.sortofmodel)
    inferencer = inferencer(sep, 'inference', **vars(args))

    def _gen():
        for mix, non_mix, separated in inferencer.inference():
            yield mix, non_mix, separated
    means, arr = evaluate(_gen(), nsrc=args.nb_speakers)
    np.save(inferencer.model.runID, arr)
