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


def parse(argv=None):
    ap = argparse.ArgumentParser(description='Training E2E asr.')
    ap.add_argument('--config', type=str, help='Path to experiment config.')
    ap.add_argument('--name', default=None, type=str, help='Name for logging.')
    ap.add_argument('--logdir', default='log/', type=str, help='Logging path.')
    ap.add_argument('--ckpdir', default='result/', type=str, help='Checkpoint/Result path.')
    ap.add_argument('--load', default=None, type=str, help='Load pre-trained model')
    ap.add_argument('--seed', default=0, type=int, help='Random seed for reproducable results.')
    ap.add_argument('--njobs', default=1, type=int, help='Number of threads for decoding.')
    ap.add_argument('--cpu', action='store_true', help='Disable GPU training.')
    ap.add_argument('--test', action='store_true', help='Test the model.')
    ap.add_argument('--no-msg', action='store_true', help='Hide all messages.')
    ap.add_argument('--rnnlm', action='store_true', help='Option for training RNNLM.')
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
    if paras.rnnlm:
        from src.solver import RNNLM_Trainer as Solver
    elif paras.test:
        from src.solver import Tester as Solver
    else:
        from src.solver import Trainer as Solver
    solver = Solver(config, paras)
    solver.load_data()
    solver.set_model()
    solver.exec()
    return solver


if __name__ == '__main__':
    main()
