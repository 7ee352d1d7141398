# coding: utf-8
"""python -m experiments.training.front_L41  (reference experiments/training/front_L41.py)."""
from utils.trainer import MyArgs, Front_Separator_Trainer
from models.L41 import L41Model

if __name__ == '__main__':
    p = MyArgs()
    p.parser.add_argument('--model_folder', help='Path to the model folder to load', required=True)
    p.parser.add_argument('--model_previous', help='Path to previous folder to load', required=False, default=None)
    p.add_separator_args()
    args = p.get_args()
    trainer = Front_Separator_Trainer(L41Model, 'front_L41', pretraining=False, **vars(args))
    trainer.train()
