# coding: utf-8
"""Table of the reference's training entry points (experiments/training/*.py): which Trainer, which separator, the `type` string
it passes, which argument groups it registers.  Every `python -m experiments.training.<name>` module is a three-line stub that
calls `main(<name>)`; behaviour (flags, defaults, type strings -- including the reference's reuse of 'front_L41_finetuning' for the
DPCL fine-tuning script, front_DPCL_finetuning.py:17) is defined here once."""
import utils.trainer as T


#   name: (trainer, separator, type string, --model_folder required (None = no such flag), --model_previous, argument groups, pretraining kw)
RECIPES = {
    'pretraining': ('Adapt_Pretrainer', None, None, None, False, ('adapt',), True),
    'STFT_DPCL': ('STFT_Separator_Trainer', 'DPCL', 'STFT_DPCL', False, False, ('stft', 'separator'), None),
    'STFT_L41': ('STFT_Separator_Trainer', 'L41Model', 'STFT_L41', False, False, ('stft', 'separator'), None),
    'STFT_DPCL_enhance': ('STFT_Separator_enhance_Trainer', 'DPCL', 'STFT_DPCL_enhance', True, False, ('stft', 'separator', 'enhance_layer'), None),
    'STFT_L41_enhance': ('STFT_Separator_enhance_Trainer', 'L41Model', 'STFT_L41_enhance', True, False, ('stft', 'separator', 'enhance_layer'), None),
    'STFT_DPCL_finetuning': ('STFT_Separator_FineTune_Trainer', 'DPCL', 'STFT_DPCL_finetuning', True, False, ('stft', 'finetuning', 'separator'), None),
    'STFT_L41_finetuning': ('STFT_Separator_FineTune_Trainer', 'L41Model', 'STFT_L41_finetuning', True, False, ('stft', 'finetuning', 'separator'), None),
    'front_DPCL': ('Front_Separator_Trainer', 'DPCL', 'front_DPCL', True, True, ('separator',), False),
    'front_L41': ('Front_Separator_Trainer', 'L41Model', 'front_L41', True, True, ('separator',), False),
    'front_DPCL_enhance': ('Front_Separator_Enhance_Trainer', 'DPCL', 'front_DPCL_enhance', True, False, ('separator', 'enhance_layer'), False),
    'front_L41_enhance': ('Front_Separator_Enhance_Trainer', 'L41Model', 'front_L41_enhance', True, False, ('separator', 'enhance_layer'), False),
    'front_DPCL_finetuning': ('Front_Separator_Finetuning_Trainer', 'DPCL', 'front_L41_finetuning', True, False, ('adapt', 'separator'), False),
    'front_L41_finetuning': ('Front_Separator_Finetuning_Trainer', 'L41Model', 'front_L41_finetuning', True, False, ('adapt', 'separator'), False),
    'front_DPCL_enhance_finetuning': ('Front_Separator_Enhance_Finetuning_Trainer', 'DPCL', 'front_DPCL_finetuning', True, False, ('adapt', 'separator', 'finetuning', 'enhance_layer'), False),
    'front_L41_enhance_finetuning': ('Front_Separator_Enhance_Finetuning_Trainer', 'L41Model', 'front_L41_finetuning', True, False, ('adapt', 'separator', 'finetuning', 'enhance_layer'), False),
}


def build_parser(name):
    trainer, sep, typ, need_folder, has_prev, groups, pre = RECIPES[name]
    p = T.MyArgs()
    if need_folder is not None:
        p.parser.add_argument('--model_folder', help='Path to the model folder to load', required=need_folder, default=None)
    if has_prev:
        p.parser.add_argument('--model_previous', help='Path to previous folder to load', required=False, default=None)
    for g in groups:
        getattr(p, 'add_%s_args' % g)()
    return p


def make_trainer(name, argv=None):
    trainer, sep, typ, need_folder, has_prev, groups, pre = RECIPES[name]
    p = build_parser(name)
    args = p.get_args() if argv is None else p.get_args(argv)
    kw = dict(vars(args))
    if pre is not None:
        kw['pretraining'] = pre
    cls = getattr(T, trainer)
    if sep is None:
        return cls(**kw)
    if sep == 'DPCL':
        from models.dpcl import DPCL as separator
    else:
        from models.L41 import L41Model as separator
    return cls(separator, typ, **kw)


def main(name):
    make_trainer(name).train()
