# coding: utf-8
"""python -m experiments.training.STFT_DPCL_finetuning  (reference experiments/training/STFT_DPCL_finetuning.py)."""
from utils.trainer import MyArgs, STFT_Separator_FineTune_Trainer
from models.dpcl import DPCL

if __name__ == '__main__':
    p = MyArgs()
    p.parser.add_argument('--model_folder', help='Path to the model folder to load', required=True)
    p.add_stft_args()
    p.add_separator_args()
    p.add_enhance_layer_args()
    p.add_finetuning_args()
    args = p.get_args()
    trainer = STFT_Separator_FineTune_Trainer(DPCL, 'STFT_DPCL_finetuning', **vars(args))
    trainer.train()
