# coding: utf-8
"""python -m experiments.training.front_DPCL_enhance_finetuning  (reference experiments/training/front_DPCL_enhance_finetuning.py)."""
from utils.trainer import MyArgs, Front_Separator_Enhance_Finetuning_Trainer
from models.dpcl import DPCL

if __name__ == '__main__':
    p = MyArgs()
    p.parser.add_argument('--model_folder', help='Path to the model folder to load', required=True)
    p.add_adapt_args()
    p.add_separator_args()
    p.add_enhance_layer_args()
    p.add_finetuning_args()
    args = p.get_args()
    trainer = Front_Separator_Enhance_Finetuning_Trainer(DPCL, 'front_DPCL_finetuning', pretraining=False, **vars(args))
    trainer.train()
