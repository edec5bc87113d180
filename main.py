#!/usr/bin/env python
# coding: utf-8
"""Entry point with the reference's command line (reference main.py:12-49): same flags, same YAML, same
Solver.load_data() / set_model() / exec() sequence.  Only deltas: yaml.safe_load (PyYAML >= 6 rejects a bare
yaml.load) and one process per GPU when launched under torchrun."""
import argparse
import os
import random
import sys

import numpy as np
import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


# (flag, default, type, what it does) -- the reference's command-line surface
FLAGS = (
    ('--config', None, str, 'YAML with the asr_model / solver sections'),
    ('--name', None, str, 'run name (default: <config stem>_sd<seed>)'),
    ('--logdir', 'log/', str, 'where scalars / tensorboard events go'),
    ('--ckpdir', 'result/', str, 'where checkpoints and decode outputs go'),
    ('--load', None, str, 'checkpoint to resume from'),
    ('--seed', 0, int, 'seed of Python / numpy / torch RNGs'),
    ('--njobs', 1, int, 'accepted for compatibility (the beam is batched on the device)'),
)
SWITCHES = (
    ('--cpu', 'refused: there is no CPU path in this build'),
    ('--test', 'beam-search decoding of the test set (Tester)'),
    ('--no-msg', 'quiet'),
    ('--rnnlm', 'refused: the RNN-LM is out of scope'),
)


def parse(argv=None):
    ap = argparse.ArgumentParser(description='LAS training / decoding on MI355X')
    for flag, default, typ, what in FLAGS:
        ap.add_argument(flag, default=default, type=typ, help=what)
    for flag, what in SWITCHES:
        ap.add_argument(flag, action='store_true', help=what)
    a = ap.parse_args(argv)
    a.gpu, a.verbose = not a.cpu, not a.no_msg
    return a


def main(argv=None):
    paras = parse(argv)
    with open(paras.config, 'r') as f:
        config = yaml.safe_load(f)
    random.seed(paras.seed)
    np.random.seed(paras.seed)
    torch.manual_seed(paras.seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(paras.seed)
    import src.solver as S
    kind = S.RNNLM_Trainer if paras.rnnlm else (S.Tester if paras.test else S.Trainer)
    solver = kind(config, paras)
    for stage in (solver.load_data, solver.set_model, solver.exec):
        stage()
    return solver


if __name__ == '__main__':
    main()
