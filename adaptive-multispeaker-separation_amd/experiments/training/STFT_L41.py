"""python -m experiments.training.STFT_L41 -- see experiments/training/_recipes.py."""
from experiments.training._recipes import main

if __name__ == '__main__':
    main('STFT_L41')
