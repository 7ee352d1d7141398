"""python -m experiments.training.front_DPCL_enhance_finetuning -- see experiments/training/_recipes.py."""
from experiments.training._recipes import main

if __name__ == '__main__':
    main('front_DPCL_enhance_finetuning')
