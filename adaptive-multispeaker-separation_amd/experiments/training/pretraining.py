"""python -m experiments.training.pretraining -- see experiments/training/_recipes.py."""
from experiments.training._recipes import main

if __name__ == '__main__':
    main('pretraining')
