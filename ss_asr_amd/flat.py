"""Binds the reference's flat module names to this package.  The reference's scripts import their siblings by bare
name (`import trainer`, src/train.py:9; `from asr import ASR`, src/trainer.py:20-31); importing THIS module installs
those names in sys.modules -- `trainer` is then ss_asr_amd.trainer, and so on -- so that the reference's unmodified
entry points run against the MI355X implementation (ss_asr_amd/run_reference.py)."""
import importlib
import sys

NAMES = ('asr', 'trainer', 'ASRDataset', 'preprocess', 'postprocess', 'TrackerHandler', 'LogHandler',
         'text_autoencoder', 'discriminator', 'speech_autoencoder')


def install():
    for name in NAMES:
        sys.modules[name] = importlib.import_module('ss_asr_amd.' + name)


install()
