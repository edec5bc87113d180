"""CPU: host-side logic of the package (no GPU, no compute calls into the HIP library)."""
import importlib
import os
import pickle
import numpy as np
import pytest
import torch
import yaml

pkg = importlib.import_module('end-to-end-asr-pytorch_amd')
synth = importlib.import_module('end-to-end-asr-pytorch_amd.synth')
dataset = importlib.import_module('end-to-end-asr-pytorch_amd.dataset')
post = importlib.import_module('end-to-end-asr-pytorch_amd.postprocess')
asr = importlib.import_module('end-to-end-asr-pytorch_amd.asr')
ops = importlib.import_module('end-to-end-asr-pytorch_amd.ops')

TIMIT_YAML = """
asr_model:
  optimizer: {type: 'Adam', learning_rate: 0.0001, joint_ctc: 0.0}
  encoder: {enc_type: 'BiRNN', sample_rate: '2_2_1', sample_style: 'concat', dim: '256_256_256', dropout: '0_0_0', rnn_cell: 'LSTM'}
  attention: {att_mode: 'dot', dim: 256, proj: True, num_head: 1}
  decoder: {dim: 256, layer: 1, dropout: 0, rnn_cell: 'LSTMCell'}
"""


def test_param_count_matches_reference_timit_config():
    """SURVEY.md §8 A2 [probe]: the reference's Seq2Seq for config/timit_example.yaml (D=39, V=63) has 9,500,991
    parameters; the same YAML keys must size the same tensors here."""
    cfg = yaml.safe_load(TIMIT_YAML)['asr_model']
    shapes = asr.param_shapes(torch.zeros(8, 300, 39), 63, cfg)
    assert sum(int(np.prod(s)) for s in shapes.values()) == 9500991
    assert shapes['encoder.layer0.layer.weight_ih_l0_reverse'] == (1024, 39)
    assert shapes['encoder.proj1.weight'] == (1024, 1024)
    assert shapes['decoder.layer0.weight_ih'] == (1024, 768)
    assert shapes['char_trans.weight'] == (63, 256)


def test_config_surface_errors():
    cfg = yaml.safe_load(TIMIT_YAML)['asr_model']
    bad = dict(cfg, encoder=dict(cfg['encoder'], dim='256_256'))
    with pytest.raises(AssertionError):
        asr.param_shapes(torch.zeros(1, 10, 39), 63, bad)
    vgg = dict(cfg, encoder=dict(cfg['encoder'], enc_type='VGGBiRNN'))
    sh = asr.param_shapes(torch.zeros(1, 10, 39), 63, vgg)              # 3x13 MFCC -> 3 input channels, 3*128 features
    assert sh['encoder.vgg_extractor.conv1.weight'] == (64, 3, 3, 3)
    assert sh['encoder.vgg_extractor.conv4.weight'] == (128, 128, 3, 3)
    assert sh['encoder.layer0.layer.weight_ih_l0'][1] == 384
    with pytest.raises(ValueError):                                     # check_dim, asr.py:522-531
        asr.param_shapes(torch.zeros(1, 10, 41), 63, vgg)
    bad = dict(cfg, encoder=dict(cfg['encoder'], rnn_cell='GRU'))
    with pytest.raises(NotImplementedError):
        asr.param_shapes(torch.zeros(1, 10, 39), 63, bad)
    bad = dict(cfg, attention=dict(cfg['attention'], att_mode='foo'))
    with pytest.raises(ValueError):
        asr.param_shapes(torch.zeros(1, 10, 39), 63, bad)


def test_flat_param_layout_alignment():
    cfg = yaml.safe_load(TIMIT_YAML)['asr_model']
    cfg['encoder'].update(dim='8_8_8')
    cfg['attention'].update(dim=6, att_mode='loc')
    cfg['decoder'].update(dim=8)
    m = asr.Seq2Seq(torch.zeros(2, 20, 5), 9, cfg, device='cpu')
    for name, (off, k, shape) in m.param_slices.items():
        if not name.endswith('_reverse'):
            assert off % 64 == 0, name
    # per-direction twins are adjacent so the kernel-facing concatenation is a plain view
    o0, k0, _ = m.param_slices['encoder.layer0.layer.weight_ih_l0']
    o1, _, _ = m.param_slices['encoder.layer0.layer.weight_ih_l0_reverse']
    assert o1 == o0 + k0
    cat = m._cat('encoder.layer0.layer', 'weight_ih', (2 * 32, 5))
    assert torch.equal(cat[:32], m.P('encoder.layer0.layer.weight_ih_l0').data)
    assert torch.equal(cat[32:], m.P('encoder.layer0.layer.weight_ih_l0_reverse').data)
    assert m.P('decoder.layer0.bias_ih').data[8:16].eq(1).all()        # forget-gate bias = 1 (asr.py:144-153)
    assert m.P('embed.weight').grad.data_ptr() == m.flat_grads.data_ptr() + 4 * m.param_slices['embed.weight'][0]
    names = [n for n, _ in m.named_parameters()]
    assert 'attention.loc_conv.weight' in names and 'ctc_layer.weight' not in names


def test_no_cpu_fallback():
    with pytest.raises(pkg.LasError):
        ops.infer_lengths(torch.zeros(2, 3, 4))


def test_synthetic_batch_contract():
    x, y, lens = synth.make_batch(3, 6, 50, 7, 11, 9, time_reduction=4)
    assert x.shape == (6, 50, 7) and lens == sorted(lens, reverse=True) and lens[0] == 50
    for b, l in enumerate(lens):
        assert (x[b, l:] == 0).all() and (x[b, :l].abs().sum(-1) != 0).all()
        n = int((y[b] != 0).sum()) - 1
        assert y[b, 0] == 0 and y[b, n + 1] == 1 and (y[b, n + 2:] == 0).all() and (y[b, 1:n + 1] >= 2).all()
        rep = int((y[b, 2:n + 2] == y[b, 1:n + 1]).sum())
        assert (n + 1) + rep <= l // 4                 # CTC-feasible: tokens + <eos> + repeats <= T' 
    x2, y2, _ = synth.make_batch(3, 6, 50, 7, 11, 9, time_reduction=4)
    assert torch.equal(x, x2) and torch.equal(y, y2)
    # attention-only workloads keep every label as drawn (no clamp), CTC ones only cut what is infeasible
    _, y3, _ = synth.make_batch(3, 6, 50, 7, 11, 30, time_reduction=4, ctc=False)
    _, y4, _ = synth.make_batch(3, 6, 50, 7, 11, 30, time_reduction=4, ctc=True)
    n3, n4 = (y3 != 0).sum(-1) - 1, (y4 != 0).sum(-1) - 1
    assert int(n3.min()) >= 15 and int(n4.max()) <= 50 // 4 - 1 and (n4 <= n3).all()
    assert synth.total_downsample('2_2_1_1_1') == 4


def test_timit_buckets(tmp_path):
    rng = np.random.RandomState(0)
    xs = [rng.randn(n, 4).astype(np.float32) for n in [5, 9, 7, 3, 8]]
    ys = [[0, 2, 3, 1], [0, 4, 1], [0, 2, 2, 2, 1], [0, 5, 1], [0, 3, 1]]
    pickle.dump(xs, open(tmp_path / 'train_x.pkl', 'wb'))
    pickle.dump(ys, open(tmp_path / 'train_y.pkl', 'wb'))
    ds = dataset.TimitBuckets(str(tmp_path), ['train'], 2)
    assert len(ds) == 3
    x0, y0 = ds.get(0)
    assert x0.shape == (2, 9, 4) and np.allclose(x0[0], xs[1]) and np.allclose(x0[1, :8], xs[4]) and (x0[1, 8:] == 0).all()
    assert y0.tolist() == [[0, 4, 1], [0, 3, 1]]
    batches = list(ds)
    assert batches[0][0].shape[0] == 1 and batches[0][0].dim() == 4 and batches[0][1].dim() == 3


def test_libri_half_batch_rule(tmp_path):
    import pandas as pd
    rows = []
    for i, n in enumerate([900, 850, 500, 400]):
        np.save(tmp_path / f'u{i}.npy', np.ones((n, 3), np.float32))
        rows.append(dict(file_path=f'u{i}.npy', length=n, label='0_5_6_1'))
    pd.DataFrame(rows).to_csv(tmp_path / 'train.csv', index=False)
    ds = dataset.LibriBuckets(str(tmp_path), ['train'], 2)
    assert [len(b) for b in ds.items] == [1, 1, 2]        # first bucket exceeds 800 frames -> halved (dataset.py:88-92)
    x, y = ds.get(2)
    assert x.shape == (2, 500, 3) and y.tolist() == [[0, 5, 6, 1], [0, 5, 6, 1]]


def test_mapper_and_metrics():
    m = post.Mapper(mapping={'<sos>': 0, '<eos>': 1, 'a': 2, 'b': 3, 'c': 4})
    assert m.unit == 'char' and m.get_dim() == 5
    assert m.translate([2, 3, 1, 4], return_string=True) == 'ab'
    assert post.edit_distance('kitten', 'sitting') == 3
    assert post.cal_acc(np.array([[2, 3, 4], [2, 2, 2]]), np.array([[2, 3, 0], [3, 2, 1]])) == pytest.approx((1.0 + 1 / 3) / 2)
    ph = post.Mapper(mapping={'<sos>': 0, '<eos>': 1, 'ao': 2, 'h#': 3})
    assert ph.unit == 'phone' and ph.translate([2, 3, 1], return_string=True) == 'aa h# '
    assert post.cal_cer(np.array([[2, 3, 1]]), np.array([[2, 2, 1]]), m) == 1.0


def test_main_cli_flags():
    import main
    a = main.parse(['--config', 'config/x.yaml', '--seed', '3', '--cpu', '--no-msg'])
    assert a.config == 'config/x.yaml' and a.seed == 3 and a.gpu is False and a.verbose is False
    assert a.logdir == 'log/' and a.ckpdir == 'result/' and a.njobs == 1 and not a.test and not a.rnnlm


def _write_libri_dir(d, root):
    """Rebuild the csv + .npy directory the reference's LoadDataset saw when tools/gen_golden.py::g8_libri ran."""
    D = int(d['D'])
    for split in ['train', 'dev', 'test']:
        os.makedirs(os.path.join(root, split), exist_ok=True)
        with open(os.path.join(root, split + '.csv'), 'w') as f:
            f.write('file_path,length,label\n')
            for row in d[f'{split}.csv']:
                f.write(str(row) + '\n')
                fp, n, _ = str(row).split(',')
                # content: a deterministic function of the path (the golden stores per-frame sums of two buckets only)
                np.save(os.path.join(root, fp), np.full((int(n), D), 1.0 + (hash_path(fp) % 7), np.float32))


def hash_path(fp):
    return sum(ord(c) for c in fp)


SOLVER_LIBRI = dict(batch_size=4, max_timestep=1200, max_label_len=400, use_gpu=False, n_jobs=0, dataset='librispeech',
                    train_set=['train'], dev_set=['dev'], test_set=['test'], dev_batch_size=4, decode_beam_size=1, dev_step=10)


def test_libri_buckets_match_reference_loader(tmp_path):
    """N4 pinned: bucket membership AND order of LibriBuckets against the reference's LoadDataset (tests/golden/g8_*, made
    by tools/gen_golden.py from /root/reference/src/dataset.py:57-155): half-batch rule by length (> 800) and by label
    length (> 150), drop filters for train/dev but not test, ties in length, trailing partial bucket, single-utterance test
    buckets when decode_beam_size > 1, zero-padded labels."""
    d = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g8_libri_buckets.npz'))
    root = str(tmp_path)
    _write_libri_dir(d, root)
    for split in ['train', 'dev', 'test']:
        ds = dataset.LoadDataset(split, text_only=False, data_path=root, **SOLVER_LIBRI)
        assert len(ds) == int(d[f'{split}.n_buckets']), split
        for i in range(len(ds)):
            assert [f for f, _, _ in ds.items[i]] == [str(v) for v in d[f'{split}.bucket{i}.files']], (split, i)
            x, y = ds.get(i)
            assert list(x.shape) == list(d[f'{split}.bucket{i}.xshape'])
            np.testing.assert_array_equal(y, d[f'{split}.bucket{i}.y'])
            lens = [(x[b].sum(-1) != 0).sum() for b in range(x.shape[0])]
            assert lens == sorted(lens, reverse=True)
    ds = dataset.LoadDataset('test', text_only=False, data_path=root, **dict(SOLVER_LIBRI, decode_beam_size=5))
    assert [b[0][0] for b in ds.items] == [str(v) for v in d['test_beam.files']] and all(len(b) == 1 for b in ds.items)
    # the train split saw the half-batch rule (by length and by label length) and both drop filters
    sizes = [len(b) for b in dataset.LoadDataset('train', text_only=False, data_path=root, **SOLVER_LIBRI).items]
    assert sizes == [2, 2, 4, 2, 2, 4, 3]


@pytest.mark.parametrize('kind', ['timit', 'libri'])
def test_buckets_are_sharded_over_ranks(tmp_path, kind):
    """Data parallel on real data: at world = 2 (and 3: unequal shards) the ranks' shards of every bucket are
    disjoint, cover it, stay sorted by length, and all ranks walk the buckets in the same (shuffled) order."""
    import random
    root = str(tmp_path)
    if kind == 'libri':
        d = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g8_libri_buckets.npz'))
        _write_libri_dir(d, root)
        mk = lambda r, w: dataset.LoadDataset('train', text_only=False, data_path=root, rank=r, world=w, **SOLVER_LIBRI)
    else:
        rng = np.random.RandomState(0)
        xs = [rng.randn(n, 4).astype(np.float32) for n in rng.randint(5, 40, size=11)]
        ys = [[0] + rng.randint(2, 9, size=rng.randint(1, 5)).tolist() + [1] for _ in xs]
        pickle.dump(xs, open(tmp_path / 'train_x.pkl', 'wb'))
        pickle.dump(ys, open(tmp_path / 'train_y.pkl', 'wb'))
        mk = lambda r, w: dataset.LoadDataset('train', text_only=False, data_path=root, rank=r, world=w,
                                              **dict(SOLVER_LIBRI, dataset='timit', max_timestep=0, max_label_len=0))
    for world in (2, 3):
        random.seed(5)
        full = [(x[0], y[0]) for x, y in mk(0, 1)]
        shards = []
        for r in range(world):
            random.seed(5)                                   # main.py seeds `random` identically on every rank
            ds = mk(r, world)
            got = []
            for x, y in ds:
                got.append((x[0], y[0], ds.last_global_B))
            shards.append(got)
        assert all(len(s) == len(full) for s in shards)
        for i, (fx, fy) in enumerate(full):
            B = fx.shape[0]
            assert all(s[i][2] == B for s in shards)
            assert sum(s[i][0].shape[0] for s in shards) == B
            for r in range(world):
                sx, sy, _ = shards[r][i]
                idx = list(range(r, B, world))
                assert sx.shape[0] == len(idx)
                for k, b in enumerate(idx):
                    n = int((fx[b].abs().sum(-1) != 0).sum())
                    assert torch.equal(sx[k, :n], fx[b, :n]) and (sx[k, n:] == 0).all()
                    L = int((fy[b] != 0).sum()) + 1
                    assert torch.equal(sy[k, :L], fy[b, :L])


def test_draw_att_shapes():
    att = [np.arange(2 * 5 * 3, dtype=np.float32).reshape(2, 5, 3)]
    maps = post.draw_att(att, np.array([[4, 1, 2, 2, 2], [3, 3, 3, 3, 3]]))
    assert maps[0].shape == (3, 2, 3) and maps[1].shape == (3, 5, 3)          # cut at <eos> (postprocess.py:149-155)
    assert np.array_equal(maps[0][1], att[0][0, :2])
