# coding: utf-8
"""python -m experiments.training.pretraining  (reference experiments/training/pretraining.py)."""
from utils.trainer import MyArgs, Adapt_Pretrainer


if __name__ == '__main__':
    p = MyArgs()
    p.add_adapt_args()
    args = p.get_args()
    trainer = Adapt_Pretrainer(pretraining=True, **vars(args))
    trainer.train()
