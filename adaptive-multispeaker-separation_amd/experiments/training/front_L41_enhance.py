# coding: utf-8
"""python -m experiments.training.front_L41_enhance  (reference experiments/training/front_L41_enhance.py)."""
from utils.trainer import MyArgs, Front_Separator_Enhance_Trainer
from models.L41 import L41Model

if __name__ == '__main__':
    p = MyArgs()
    p.parser.add_argument('--model_folder', help='Path to the model folder to load', required=True)
    p.add_separator_args()
    p.add_enhance_layer_args()
    args = p.get_args()
    trainer = Front_Separator_Enhance_Trainer(L41Model, 'front_L41_enhance', pretraining=False, **vars(args))
    trainer.train()
