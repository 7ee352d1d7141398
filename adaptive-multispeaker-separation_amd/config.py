"""Global constants (reference config.py:6-47 semantics: seed, fs, log_dir under the working tree)."""
import os

seed = 42

workdir = os.path.dirname(os.path.abspath(__file__))
log_dir = os.environ.get('AMS_LOG_DIR', os.path.join(workdir, 'log'))
model_root = workdir

# audio
fs = 8000
