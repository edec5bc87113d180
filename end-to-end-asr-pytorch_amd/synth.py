"""Synthetic utterance batches in the reference's data contract:
x (B,T,D) float32 zero-padded after each utterance, lengths descending; y (B,L+2) = [0, tok..., 1, 0...]."""
import math
import numpy as np
import torch


def total_downsample(sample_rate):
    r = 1
    for v in str(sample_rate).split('_'):
        r *= int(v)
    return r


def make_batch(step, B, T_max, D, V, L_max, time_reduction=1, seed=1234, min_frac=0.6, l_frac=0.5, full_length=False, ctc=True):
    """Per-step seed s = seed + step; T_b ~ U{ceil(min_frac*T_max)..T_max} with one forced to T_max; labels
    n_b ~ U{ceil(l_frac*L_max)..L_max}, tokens ~ U{2..V-1}.  SURVEY.md 8d: with a CTC loss in the step (`ctc`) a label is
    cut to the longest prefix that is CTC-feasible, n_b + 1 (<eos>) + repeats <= T'_b; an attention-only step keeps every
    label as drawn."""
    rng = np.random.RandomState(seed + step)
    lens = [T_max] + [T_max if full_length else int(rng.randint(math.ceil(min_frac * T_max), T_max + 1)) for _ in range(B - 1)]
    lens = sorted(lens, reverse=True)
    x = np.zeros((B, T_max, D), np.float32)
    for b, l in enumerate(lens):
        x[b, :l] = rng.randn(l, D).astype(np.float32)
        # a real frame must not sum to exactly 0 (lengths are inferred from zero frames, solver.py:134)
    ns = [int(rng.randint(max(1, math.ceil(l_frac * L_max)), L_max + 1)) for _ in range(B)]
    y = np.zeros((B, max(ns) + 2), np.int64)
    for b, n in enumerate(ns):
        tok = rng.randint(2, V, size=n)
        if ctc:
            tp = lens[b] // time_reduction
            need = np.arange(1, n + 1) + 1 + np.concatenate([[0], np.cumsum(tok[1:] == tok[:-1])])   # prefix k: k + 1 + repeats
            n = max(1, int((need <= tp).sum()))
        y[b, 1:n + 1] = tok[:n]
        y[b, n + 1] = 1
    return torch.from_numpy(x), torch.from_numpy(y), lens


class SyntheticSet:
    """Iterable of (x (1,B,T,D), y (1,B,L+2)) like the reference's bucketed DataLoader (dataset.py:155)."""

    def __init__(self, n_batches, B, T_max, D, V, L_max, time_reduction, seed=1234, rank=0, world=1):
        self.n, self.args = n_batches, (B, T_max, D, V, L_max, time_reduction)
        self.seed, self.rank, self.world = seed, rank, world
        self.epoch = 0

    def __len__(self):
        return self.n

    def __iter__(self):
        B, T_max, D, V, L_max, tr = self.args
        for i in range(self.n):
            step = (self.epoch * self.n + i) * self.world + self.rank
            x, y, _ = make_batch(step, B, T_max, D, V, L_max, tr, self.seed)
            yield x.unsqueeze(0), y.unsqueeze(0)
        self.epoch += 1
