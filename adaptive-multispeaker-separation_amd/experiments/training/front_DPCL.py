# coding: utf-8
"""python -m experiments.training.front_DPCL  (reference experiments/training/front_DPCL.py)."""
from utils.trainer import MyArgs, Front_Separator_Trainer
from models.dpcl import DPCL

if __name__ == '__main__':
    p = MyArgs()
    p.parser.add_argument('--model_folder', help='Path to the model folder to load', required=True)
    p.parser.add_argument('--model_previous', help='Path to previous folder to load', required=False, default=None)
    p.add_separator_args()
    args = p.get_args()
    trainer = Front_Separator_Trainer(DPCL, 'front_DPCL', pretraining=False, **vars(args))
    trainer.train()
