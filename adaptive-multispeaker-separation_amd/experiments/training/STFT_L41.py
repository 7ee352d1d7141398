# coding: utf-8
"""python -m experiments.training.STFT_L41  (reference experiments/training/STFT_L41.py)."""
from utils.trainer import MyArgs, STFT_Separator_Trainer
from models.L41 import L41Model

if __name__ == '__main__':
    p = MyArgs()
    p.parser.add_argument('--model_folder', help='Path to the Model folder to load', required=False, default=None)
    p.add_stft_args()
    p.add_separator_args()
    args = p.get_args()
    trainer = STFT_Separator_Trainer(L41Model, 'STFT_L41', **vars(args))
    trainer.train()
