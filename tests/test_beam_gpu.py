"""GPU parity of the decode path (N3): CTC prefix scorer, row log-softmax / top-k, and the joint CTC/attention beam
search through the C ABI, against golden vectors from the imported reference (g5_ctc_prefix, g6_beam_*) and the CPU
oracle.  f32 mode: prefix scores / per-token hypothesis scores within 1e-4 (float32 log-space sums over T' frames;
libm vs device expf/log1pf differ by ulps), identical token sequences in identical order."""
import argparse
import ctypes
import importlib
import os
import sys
import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
sys.path.insert(0, os.path.join(ROOT, 'tools'))


@pytest.fixture(scope='module')
def mods():
    importlib.import_module('end-to-end-asr-pytorch_amd')
    m = lambda n: importlib.import_module('end-to-end-asr-pytorch_amd.' + n)
    return m('ops'), m('asr'), m('_lib'), m('beam')


def prefix_score(lib, lp, r_prev, last, plen, cand):
    L_ = lib.lib()
    T, V = lp.shape
    N, K = cand.shape
    f = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=DEV)
    i = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.int32, device=DEV)
    lp_d, rp_d, la_d, pl_d, ca_d = f(lp), f(r_prev), i(last), i(plen), i(cand)
    psi = torch.empty(N, K, device=DEV)
    r = torch.empty(N, K, T, 2, device=DEV)
    P, I = lib.P, lib.I
    lib.check(L_.las_ctc_prefix_score(P(lp_d.data_ptr()), I(T), I(V), P(rp_d.data_ptr()), P(la_d.data_ptr()), P(pl_d.data_ptr()),
                                      P(ca_d.data_ptr()), I(N), I(K), P(psi.data_ptr()), P(r.data_ptr()), lib.cur_stream()), 'prefix')
    torch.cuda.synchronize()
    return psi.cpu().numpy(), r.cpu().numpy()


def test_ctc_prefix_golden(mods):
    ops, asr, lib, beam = mods
    d = np.load(os.path.join(GOLDEN, 'g5_ctc_prefix.npz'))
    lp = d['lp'][0]
    T, V = lp.shape
    L_ = lib.lib()
    lp_d = torch.tensor(lp, device=DEV)
    r0 = torch.empty(T, 2, device=DEV)
    lib.check(L_.las_ctc_prefix_init(lib.P(lp_d.data_ptr()), lib.I(T), lib.I(V), lib.P(r0.data_ptr()), lib.cur_stream()), 'init')
    np.testing.assert_allclose(r0.cpu().numpy(), d['r0'], atol=1e-5)
    cand = np.array([[1, 2, 3, 4, 5]])
    psi1, r1 = prefix_score(lib, lp, d['r0'][None], [0], [0], cand)
    np.testing.assert_allclose(psi1[0], d['psi1'], atol=1e-4)
    np.testing.assert_allclose(r1[0], d['r1'], atol=1e-4)
    psi2, r2 = prefix_score(lib, lp, d['r1'][2][None], [3], [1], cand)            # repeated token 3 among the candidates
    np.testing.assert_allclose(psi2[0], d['psi2'], atol=1e-4)
    np.testing.assert_allclose(r2[0], d['r2'], atol=1e-4)
    psi3, r3 = prefix_score(lib, lp, d['r2'][2][None], [3], [2], np.array([[2, 3, 5]]))
    np.testing.assert_allclose(psi3[0], d['psi3'], atol=1e-4)
    np.testing.assert_allclose(r3[0], d['r3'], atol=1e-4)


def test_ctc_prefix_batched_vs_oracle(mods):
    """N hypotheses x K candidates in one launch, prefixes of different lengths, T'=120, V=40."""
    from oracle import beam_ref as Bm
    ops, asr, lib, beam = mods
    rng = np.random.RandomState(5)
    T, V, N, K = 120, 40, 6, 9
    lp = torch.log_softmax(torch.tensor(rng.randn(T, V) * 2.0), -1).numpy().astype(np.float32)
    r0 = Bm.ctc_prefix_init(lp)
    prefixes = [[], [7], [7, 7], [3, 9, 9, 12], [5, 6], [30, 2, 2, 2, 8]]
    r_prev = []
    for g in prefixes:                               # states reached by scoring the prefix token by token
        r = r0
        for i, tok in enumerate(g):
            _, rr = Bm.ctc_prefix_cheap(lp, g[:i], r, [tok])
            r = rr[0]
        r_prev.append(r)
    cand = np.stack([rng.permutation(V)[:K] for _ in range(N)])
    for n, g in enumerate(prefixes):
        if g:
            cand[n, 0] = g[-1]                       # the repeated-token case in every non-empty row
    psi, r = prefix_score(lib, lp, np.stack(r_prev), [g[-1] if g else 0 for g in prefixes], [len(g) for g in prefixes], cand)
    for n, g in enumerate(prefixes):
        want_psi, want_r = Bm.ctc_prefix_cheap(lp, g, r_prev[n], [int(v) for v in cand[n]])
        np.testing.assert_allclose(psi[n], want_psi, atol=2e-4, rtol=1e-5)
        np.testing.assert_allclose(r[n], want_r, atol=2e-4, rtol=1e-5)


def test_log_softmax_topk_rows(mods):
    ops, asr, lib, beam = mods
    L_ = lib.lib()
    g = torch.Generator().manual_seed(3)
    for R, V, k in [(1, 9, 4), (5, 31, 7), (20, 5000, 30)]:
        x = torch.randn(R, V, generator=g).to(DEV)
        out = torch.empty_like(x)
        lib.check(L_.las_log_softmax_rows(lib.P(x.data_ptr()), lib.I(R), lib.I(V), lib.P(out.data_ptr()), lib.cur_stream()), 'lsm')
        np.testing.assert_allclose(out.cpu().numpy(), torch.log_softmax(x.cpu(), -1).numpy(), atol=2e-6)
        vals = torch.empty(R, k, device=DEV)
        idx = torch.empty(R, k, dtype=torch.int32, device=DEV)
        lib.check(L_.las_topk_rows(lib.P(out.data_ptr()), lib.I(R), lib.I(V), lib.I(k), lib.P(vals.data_ptr()), lib.P(idx.data_ptr()),
                                   lib.cur_stream()), 'topk')
        tv, ti = out.cpu().topk(k)
        assert np.array_equal(idx.cpu().numpy(), ti.numpy())
        np.testing.assert_array_equal(vals.cpu().numpy(), tv.numpy())


@pytest.mark.parametrize('name', ['loc_ctc_b1', 'loc_ctc_b3', 'dot_att_b1', 'dot_att_b3'])
def test_beam_decode_golden(mods, name):
    ops, asr, lib, beam = mods
    from gen_golden import TINY
    d = np.load(os.path.join(GOLDEN, f'g6_beam_{name}.npz'))
    cfg = TINY[name.rsplit('_', 1)[0]]
    x = torch.tensor(d['x'], device=DEV)
    ops.set_precision('f32')
    try:
        model = asr.Seq2Seq(x, int(d['V']), cfg, device=DEV)
        model.load_reference_state({k[2:]: d[k] for k in d.files if k.startswith('w.')})
        model.eval()
        hyps = model.beam_decode(x, int(d['steps']), [x.shape[1]], int(d['beam']))
        torch.cuda.synchronize()
    finally:
        ops.set_precision('bf16')
    assert int(model.status.item()) == 0
    assert len(hyps) == int(d['n_hyps'])
    for i, h in enumerate(hyps):
        assert h.outIndex == d[f'hyp{i}.seq'].tolist(), (i, h.outIndex)
        np.testing.assert_allclose(np.array(h.output_scores), d[f'hyp{i}.scores'], atol=1e-4)


@pytest.mark.parametrize('mode,ctc,beam', [('loc', 0.3, 5), ('dot', 0.0, 4), ('loc', 0.5, 20)])
def test_beam_decode_vs_oracle(mods, mode, ctc, beam):
    """A wider model (H=32, A=70 -> 2 attention lanes, T'=40, V=45) and beams up to the reference's default of 20."""
    from oracle import las_ref as R, beam_ref as Bm
    ops, asr, lib, bm = mods
    cfg = dict(optimizer=dict(type='Adam', learning_rate=1e-3, joint_ctc=ctc),
               encoder=dict(enc_type='BiRNN', sample_rate='2_1', sample_style='concat', dim='32_32', dropout='0_0', rnn_cell='LSTM'),
               attention=dict(att_mode=mode, dim=70, proj=True, num_head=1),
               decoder=dict(dim=32, layer=1, dropout=0, rnn_cell='LSTMCell'))
    torch.manual_seed(11)
    V, T, D = 45, 80, 13
    x = torch.randn(1, T, D)
    ops.set_precision('f32')
    try:
        model = asr.Seq2Seq(x, V, cfg, device=DEV)
        with torch.no_grad():
            model.P('char_trans.weight').mul_(5.0)
            if ctc > 0:
                model.P('ctc_layer.weight').mul_(3.0)
        model.eval()
        hyps = model.beam_decode(x.to(DEV), 12, [T], beam)
        torch.cuda.synchronize()
    finally:
        ops.set_precision('bf16')
    W = {k: v.detach().cpu() for k, v in model.named_parameters()}
    want = Bm.beam_decode(W, R.parse_cfg(cfg), x, 12, beam)
    assert len(hyps) == len(want)
    for h, (seq, scores) in zip(hyps, want):
        assert h.outIndex == seq
        np.testing.assert_allclose(np.array(h.output_scores), np.array(scores), atol=2e-4)


def test_tester_end_to_end(mods, tmp_path):
    """Trainer (2 steps, checkpoint at the dev check) -> Tester.load_data/set_model/exec: decode files written, one line
    per utterance and beam entry (reference solver.py:293-390)."""
    ops, asr, lib, beam = mods
    solver = importlib.import_module('end-to-end-asr-pytorch_amd.solver')
    mp = dict(optimizer=dict(type='Adam', learning_rate=1e-3, joint_ctc=0.5),
              encoder=dict(enc_type='BiRNN', sample_rate='2_2', sample_style='concat', dim='16_16', dropout='0_0', rnn_cell='LSTM'),
              attention=dict(att_mode='loc', dim=12, proj=True, num_head=1),
              decoder=dict(dim=16, layer=1, dropout=0, rnn_cell='LSTMCell'))
    config = dict(asr_model=mp, clm=dict(enable=False),
                  solver=dict(dataset='synthetic', data_path='', n_jobs=0, max_timestep=0, max_label_len=0, train_set=['train'],
                              batch_size=4, apex=False, total_steps=2, tf_start=1.0, tf_end=1.0, dev_set=['dev'], dev_batch_size=2,
                              dev_step=1, test_set=['test'], decode_beam_size=3, max_decode_step_ratio=0.2, decode_ctc_weight=0.3,
                              decode_lm_weight=0.0,
                              synthetic=dict(T_max=40, D=13, V=11, L_max=4, time_reduction=4, n_batches=2, n_dev_batches=1)))
    paras = argparse.Namespace(gpu=True, name='t', config='t.yaml', seed=0, ckpdir=str(tmp_path / 'ckpt'),
                               logdir=str(tmp_path / 'log'), load=None, verbose=False, njobs=1)
    torch.manual_seed(0)
    tr = solver.Trainer(config, paras)
    tr.load_data(); tr.set_model(); tr.exec()
    assert os.path.exists(os.path.join(tr.ckpdir, 'asr'))
    te = solver.Tester(config, paras)
    te.load_data(); te.set_model()
    n = te.exec()
    best = open(os.path.join(te.ckpdir, te.decode_file + '.txt')).read().splitlines()
    nbest = open(os.path.join(te.ckpdir, te.decode_file + '_nbest.txt')).read().splitlines()
    assert n >= 1 and len(best) == n and n <= len(nbest) <= 3 * n
    assert te.decode_file.endswith('_ctc0.3') and all('\t' in l for l in best)
