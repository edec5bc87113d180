"""Bucketed datasets in the reference's on-disk formats, plus a synthetic source.

TIMIT:        {split}_x.pkl list[np.float32 (T,D)], {split}_y.pkl list[list[int]]   (reference dataset.py:23-52)
LibriSpeech:  <split>.csv with columns file_path,length,label ('_'-joined ints) + per-utterance .npy
              (reference dataset.py:57-115), incl. the half-batch rule (T>800 or L>150, dataset.py:88-92)
synthetic:    config['solver']['dataset'] == 'synthetic' (bench / smoke; no files)
Every source yields x (1,B,T,D) zero-padded, utterances sorted by length descending, y (1,B,L+2)."""
import os
import pickle
import random

import numpy as np
import torch

from .synth import SyntheticSet, total_downsample

HALF_BATCHSIZE_TIME = 800       # reference dataset.py:12-13
HALF_BATCHSIZE_LABEL = 150


def _pad_x(xs, pad_len=0):
    T = pad_len or max(len(v) for v in xs)
    out = np.zeros((len(xs), T, xs[0].shape[-1]), np.float32)
    for i, v in enumerate(xs):
        n = min(len(v), T)
        out[i, :n] = v[:n]
    return out


def _pad_y(ys, max_len=0):
    L = max_len or max(len(v) for v in ys)
    out = np.zeros((len(ys), L), np.int64)
    for i, v in enumerate(ys):
        out[i, :len(v)] = np.asarray(v)
    return out


class _Buckets:
    """Pre-bucketed batches.  Data parallel (world > 1): every bucket is dealt round-robin over the ranks by utterance
    (dist.shard_bucket's rule: rank r takes utterances r, r + world, ... of the length-sorted bucket, which balances the
    frame counts and keeps each shard sorted); all ranks walk the buckets in the same order (the shuffle draws from
    Python's `random`, which main.py seeds identically on every rank).  `last_global_B` is the size of the whole bucket
    the last yielded shard came from: shards of one bucket can differ in size (half-batch rule, trailing bucket), and
    the Trainer weights its gradient by B_local / B_global."""

    def __init__(self, shuffle, rank=0, world=1):
        self.shuffle = shuffle
        self.items = []
        self.rank, self.world = int(rank), max(1, int(world))
        self.last_global_B = None

    def __len__(self):
        return len(self.items)

    def bucket_size(self, i):
        return len(self.items[i][0])

    def __iter__(self):
        order = list(range(len(self.items)))
        if self.shuffle:
            random.shuffle(order)
        for i in order:
            B = self.bucket_size(i)
            idx = list(range(self.rank, B, self.world)) if self.world > 1 else None
            x, y = self.get(i, idx)
            self.last_global_B = B
            yield torch.from_numpy(x).unsqueeze(0), torch.from_numpy(y).unsqueeze(0)


class TimitBuckets(_Buckets):
    def __init__(self, path, sets, bucket_size, max_timestep=0, max_label_len=0, shuffle=False, rank=0, world=1):
        super().__init__(shuffle, rank, world)
        x, y = [], []
        for s in sets:
            with open(os.path.join(path, s + '_x.pkl'), 'rb') as fp:
                x += pickle.load(fp)
            with open(os.path.join(path, s + '_y.pkl'), 'rb') as fp:
                y += pickle.load(fp)
        assert len(x) == len(y)
        order = list(reversed(np.argsort([len(t) for t in x])))          # longest first (dataset.py:38-40)
        for b in range(0, len(order), bucket_size):
            idx = order[b:b + bucket_size]
            T = min(max_timestep, len(x[idx[0]])) if max_timestep else 0
            L = min(max_label_len, max(len(y[i]) for i in idx)) if max_label_len else 0
            self.items.append((_pad_x([x[i] for i in idx], T), _pad_y([y[i] for i in idx], L)))

    def get(self, i, idx=None):
        x, y = self.items[i]
        return (x, y) if idx is None else (x[idx], y[idx])


class LibriBuckets(_Buckets):
    def __init__(self, path, sets, bucket_size, max_timestep=0, max_label_len=0, drop=False, shuffle=False, rank=0, world=1):
        super().__init__(shuffle, rank, world)
        import pandas as pd
        self.root = path
        tab = pd.concat([pd.read_csv(os.path.join(path, s + '.csv')) for s in sets], ignore_index=True)
        tab = tab.sort_values(by=['length'], ascending=False)
        if drop and max_timestep > 0:
            tab = tab[tab.length < max_timestep]
        if drop and max_label_len > 0:
            tab = tab[tab.label.str.count('_') + 1 < max_label_len]
        files, lens = tab['file_path'].tolist(), tab['length'].tolist()
        labels = [list(map(int, l.split('_'))) for l in tab['label'].tolist()]
        cur = []
        for f, n, l in zip(files, lens, labels):
            cur.append((f, n, l))
            if len(cur) == bucket_size:
                if bucket_size >= 2 and (max(c[1] for c in cur) > HALF_BATCHSIZE_TIME or
                                         max(len(c[2]) for c in cur) > HALF_BATCHSIZE_LABEL):
                    self.items += [cur[:bucket_size // 2], cur[bucket_size // 2:]]
                else:
                    self.items.append(cur)
                cur = []
        if cur:
            self.items.append(cur)

    def bucket_size(self, i):
        return len(self.items[i])

    def get(self, i, idx=None):
        items = self.items[i] if idx is None else [self.items[i][k] for k in idx]
        if not items:                                       # a bucket smaller than the world: nothing for this rank, but
            D = np.load(os.path.join(self.root, self.items[i][0][0]), mmap_mode='r').shape[-1]   # the feature dim stays the
            return np.zeros((0, 1, D), np.float32), np.zeros((0, 2), np.int64)                   # data's (it sizes the model)
        xs = [np.load(os.path.join(self.root, f)).astype(np.float32) for f, _, _ in items]
        return _pad_x(xs), _pad_y([l for _, _, l in items])


def LoadDataset(split, text_only, data_path, batch_size, max_timestep, max_label_len, use_gpu, n_jobs, dataset,
                train_set, dev_set, test_set, dev_batch_size, decode_beam_size, **kwargs):
    """Same signature as reference dataset.py:119 (extra YAML keys are swallowed)."""
    if text_only:
        raise NotImplementedError('text-only sets feed the CLM / RNN-LM, which are out of scope (SURVEY.md §2.1)')
    if split == 'train':
        bs, shuffle, sets, drop = batch_size, True, train_set, True
    elif split == 'dev':
        bs, shuffle, sets, drop = dev_batch_size, False, dev_set, True
    elif split == 'test':
        bs, shuffle, sets, drop = (1 if decode_beam_size > 1 else dev_batch_size), False, test_set, False
    else:
        raise NotImplementedError(split)
    name = dataset.upper()
    if name == 'TIMIT':
        return TimitBuckets(data_path, sets, bs, max_timestep, max_label_len, shuffle, kwargs.get('rank', 0), kwargs.get('world', 1))
    if name == 'LIBRISPEECH':
        return LibriBuckets(data_path, sets, bs, max_timestep, max_label_len, drop, shuffle, kwargs.get('rank', 0),
                            kwargs.get('world', 1))
    if name == 'SYNTHETIC':
        s = kwargs.get('synthetic', {})
        n = s.get('n_batches', 8) if split == 'train' else s.get('n_dev_batches', 1)
        return SyntheticSet(n, bs, s.get('T_max', 300), s.get('D', 39), s.get('V', 63), s.get('L_max', 40),
                            s.get('time_reduction', 4), seed=s.get('seed', 1234) + (0 if split == 'train' else 7919),
                            rank=kwargs.get('rank', 0), world=kwargs.get('world', 1))
    raise ValueError('Unsupported Dataset: ' + dataset)
