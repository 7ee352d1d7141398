# coding: utf-8
"""python -m experiments.evaluation.tf_eval --model_folder ... --model {pretraining,front_L41,front_L41_enhance,STFT_L41,...}
(reference experiments/evaluation/tf_eval.py:1-38): the in-graph SDR-improvement evaluation -- run an inference recipe over the
test split through Trainer.sdr_improvement() (utils/trainer.py:231-260), which yields [x_mix, x_non_mix, sdr_imp] per batch
(models/network.py:239-242), and report the batch-size-weighted running mean, skipping NaN batches (tf_eval.py:31-35).

The reference maps only four of the nine --model choices (tf_eval.py:16-23) and names a class that does not exist for one of them
(`STFT_inference`, tf_eval.py:19: a NameError there); this mirror maps that choice to STFT_Separator_Inference, the evident intent,
and exits with a message for the choices the reference leaves unbound (an UnboundLocalError there)."""
from __future__ import print_function

import numpy as np

from models.L41 import L41Model
from utils.trainer import (MyArgs, Front_Separator_Inference, STFT_Separator_Inference, Front_Separator_Enhanced_Inference,
                           Pretrained_Inference)

INFERENCERS = {'front_L41': Front_Separator_Inference, 'STFT_L41': STFT_Separator_Inference,
               'front_L41_enhance': Front_Separator_Enhanced_Inference, 'pretraining': Pretrained_Inference}


def running_sdr(batches, batch_size, verbose=True):
    """tf_eval.py:27-38: batches yields (x_mix, x_non_mix, sdr_imp) -> (weighted mean SDR improvement, batches counted)."""
    sdr, i = 0.0, 0
    for _, _, sdr_ in batches:
        sdr_ = float(np.asarray(sdr_.detach().cpu() if hasattr(sdr_, 'detach') else sdr_).reshape(-1)[0])
        if np.isnan(sdr_):
            continue
        sdr += sdr_ * batch_size
        i += 1
        if verbose:
            print(sdr / float(i * batch_size), sdr_)
    return (sdr / float(i * batch_size) if i else float('nan')), i


def build(argv=None):
    """Parse the reference's command line and construct the inferencer it selects (tf_eval.py:7-25)."""
    p = MyArgs()
    p.parser.add_argument('--model_folder', help='Path to the Model folder to load', required=True)
    p.select_inferencer()
    p.add_adapt_args()
    p.add_separator_args()
    args = p.get_args(argv)
    if args.model not in INFERENCERS:
        raise SystemExit('tf_eval: --model %s has no inferencer in the reference either (experiments/evaluation/tf_eval.py:16-23)'
                         % args.model)
    return INFERENCERS[args.model](L41Model, 'inference', **vars(args)), args


def main(argv=None):
    inferencer, args = build(argv)
    sdr, n = running_sdr(inferencer.sdr_improvement(), args.batch_size)
    print('SDR =', sdr)
    return sdr, n


if __name__ == '__main__':
    main()
