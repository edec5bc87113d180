"""Joint CTC/attention beam search (reference src/asr.py:155-258 `Seq2Seq.beam_decode`, src/ctc.py `CTCPrefixScore`,
src/postprocess.py:44-119 `Hypothesis`), with every hypothesis of the beam advanced in ONE batched decode step on the
device: the decoder step kernels run with batch = live hypotheses, the CTC prefix scorer with one thread per
(hypothesis, candidate) pair.  The host keeps the reference's bookkeeping (expansion, <eos> handling, average-score
ranking) on the few numbers it needs per step: the top-k scores/ids and the candidate lists.
"""
import ctypes
import torch

from . import _lib, ops
from ._lib import P, I, F, ptr, check, cur_stream
from .decoder import DecDims, DecState, make_params, weight_names, s_dtype, LOC_C

CTC_BEAM_RATIO = 1.5          # asr.py:15


class Hypothesis:
    """Result record with the reference's accessors (postprocess.py:48-119)."""

    def __init__(self, seq, scores):
        self.output_seq, self.output_scores = list(seq), list(scores)

    def avgScore(self):
        assert len(self.output_scores) != 0
        return sum(self.output_scores) / len(self.output_scores)

    @property
    def outIndex(self):
        return [int(i) for i in self.output_seq]


class _Live:
    __slots__ = ('seq', 'scores', 'slot')

    def __init__(self, seq, scores, slot):
        self.seq, self.scores, self.slot = seq, scores, slot          # slot: row of the device state tensors

    def avg(self):
        return sum(self.scores) / len(self.scores)


def beam_decode(model, audio_feature, decode_step, state_len, decode_beam_size):
    """Returns the top `decode_beam_size` Hypothesis objects of ONE utterance (asr.py:155-258)."""
    L_ = _lib.lib()
    assert audio_feature.shape[0] == 1
    if getattr(model, 'decode_lm_weight', 0) > 0:
        raise NotImplementedError('RNN-LM fusion (asr.py:232-235) is outside the LAS path (SURVEY.md §2.1)')
    if not model.joint_att:
        return []                                      # as the reference: nothing is decoded without the attention decoder
    beam = int(decode_beam_size)
    dev = audio_feature.device
    f32 = dict(dtype=torch.float32, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    with torch.no_grad():
        lens_host = [int(v) for v in state_len]
        lens_dev = torch.tensor(lens_host, **i32)
        enc, enc_len_dev, enc_len = model.encode(audio_feature.float(), lens_dev, lens_host)
        enc = enc.contiguous()
        Tp, E = int(enc.shape[1]), int(enc.shape[2])
        if decode_step == 0:
            decode_step = int(enc_len[0])
        V, C, NL, A = model.char_dim, model.dec_dim, model.dec_layers, model.A
        loc = model.att_mode == 'loc'
        lam = float(model.ctc_weight)
        joint_ctc = bool(model.joint_ctc)
        K = min(int(CTC_BEAM_RATIO * beam), V)
        kb = min(beam, V)
        names = weight_names(NL, loc)
        W = {n: model.P(n).detach().contiguous() for n in names}
        params = make_params(W, NL, loc)
        psi = ops.linear(enc, model.P('attention.psi.weight'), model.P('attention.psi.bias'), act=1).contiguous()
        lp = r_prev = None
        if joint_ctc:
            logit = ops.linear(enc, model.P('ctc_layer.weight'), model.P('ctc_layer.bias'))[0].contiguous()     # [T',V]
            lp = torch.empty_like(logit)
            check(L_.las_log_softmax_rows(ptr(logit), I(Tp), I(V), ptr(lp), cur_stream()), 'las_log_softmax_rows')
            r_prev = torch.empty(1, Tp, 2, **f32)
            check(L_.las_ctc_prefix_init(ptr(lp), I(Tp), I(V), ptr(r_prev), cur_stream()), 'las_ctc_prefix_init')
        # the utterance's encoding, replicated once for the widest beam (the step kernels index enc/psi per hypothesis)
        encB = enc.expand(beam, Tp, E).contiguous()
        psiB = psi.expand(beam, Tp, A).contiguous()
        lenB = enc_len_dev.expand(beam).contiguous()
        # device state of the live hypotheses (row = hypothesis)
        h = torch.zeros(NL, 1, C, **f32)
        c = torch.zeros(NL, 1, C, **f32)
        att = torch.zeros(1, Tp, **f32)
        if loc:
            att[0, :enc_len[0]] = 1.0 / enc_len[0]                     # Attention.forward's first-call init (asr.py:444-449)
        tok = torch.zeros(1, **i32)
        plen = torch.zeros(1, **i32)
        prev_ctc = torch.zeros(1, **f32)
        live = [_Live([], [], 0)]
        final = []
        for t in range(decode_step):
            N = len(live)
            d = DecDims(N, Tp, E, A, C, NL, V, 1, int(loc), ops._prec)
            hs = torch.empty(NL, 2, N, C, **f32)
            cs = torch.empty(NL, 2, N, C, **f32)
            hs[:, 0] = h
            cs[:, 0] = c
            attb = torch.empty(2, N, Tp, **f32)
            attb[0] = att
            S = dict(tok=tok.contiguous(), xin=torch.empty(1, N, C + E, **f32), q=torch.empty(1, N, A, **f32), att=attb, hs=hs,
                     cs=cs, gates=torch.empty(NL, 1, N, 4 * C, **f32), ebuf=torch.empty(N, Tp, **f32),
                     logits_step=torch.empty(N, V, **f32))
            if loc:
                S['f'] = torch.empty(1, N, LOC_C, Tp, **f32)
                S['s'] = torch.empty(1, N, Tp, A, dtype=s_dtype(d.prec), device=dev)
            st = DecState()
            for k, v in S.items():
                setattr(st, k, v.data_ptr())
            logits = torch.empty(N, V, **f32)
            check(L_.las_decoder_step(ctypes.byref(d), ctypes.byref(params), ptr(encB), ptr(psiB), ptr(lenB), ctypes.byref(st),
                                      ptr(logits), cur_stream()), 'las_decoder_step')
            cur = torch.empty_like(logits)
            check(L_.las_log_softmax_rows(ptr(logits), I(N), I(V), ptr(cur), cur_stream()), 'las_log_softmax_rows')
            cand = psi_c = r_out = None
            if joint_ctc:
                cv = torch.empty(N, K, **f32)
                cand = torch.empty(N, K, **i32)
                check(L_.las_topk_rows(ptr(cur), I(N), I(V), I(K), ptr(cv), ptr(cand), cur_stream()), 'las_topk_rows')
                psi_c = torch.empty(N, K, **f32)
                r_out = torch.empty(N, K, Tp, 2, **f32)
                check(L_.las_ctc_prefix_score(ptr(lp), I(Tp), I(V), ptr(r_prev), ptr(tok), ptr(plen), ptr(cand), I(N), I(K),
                                              ptr(psi_c), ptr(r_out), cur_stream()), 'las_ctc_prefix_score')
                check(L_.las_beam_combine(ptr(cur), I(N), I(V), ptr(cand), ptr(psi_c), ptr(prev_ctc), I(K), F(lam),
                                          cur_stream()), 'las_beam_combine')
            topv = torch.empty(N, kb, **f32)
            topi = torch.empty(N, kb, **i32)
            check(L_.las_topk_rows(ptr(cur), I(N), I(V), I(kb), ptr(topv), ptr(topi), cur_stream()), 'las_topk_rows')
            # ---- host bookkeeping on N x beam numbers (Hypothesis.addTopk, postprocess.py:71-104)
            pack = [topv, topi.float()] + ([cand.float()] if joint_ctc else [])       # one D2H (ids < 2^24 are exact in fp32)
            host = torch.cat(pack, dim=1).cpu()
            topv_h = host[:, :kb].tolist()
            topi_h = host[:, kb:2 * kb].to(torch.int64).tolist()
            cand_h = host[:, 2 * kb:].to(torch.int64).tolist() if joint_ctc else None
            nxt = []
            for hyp in live:
                n = hyp.slot
                term = None
                for i in range(kb):
                    tk, sc = int(topi_h[n][i]), float(topv_h[n][i])
                    if tk == 1:
                        term = sc
                        continue
                    j = cand_h[n].index(tk) if joint_ctc else 0
                    nxt.append((_Live(hyp.seq + [tk], hyp.scores + [sc], None), n, j))
                if term is not None:
                    final.append(Hypothesis(hyp.seq + [1], hyp.scores + [term]))
                    if beam == 1:
                        return final
            nxt.sort(key=lambda o: o[0].avg(), reverse=True)            # stable, as list.sort in the reference
            nxt = nxt[:beam]
            live = []
            if not nxt:
                break
            parents = torch.tensor([p for _, p, _ in nxt], dtype=torch.long, device=dev)
            h = hs[:, 1].index_select(1, parents)
            c = cs[:, 1].index_select(1, parents)
            att = attb[1].index_select(0, parents)
            tok = torch.tensor([o.seq[-1] for o, _, _ in nxt], **i32)
            plen = torch.tensor([len(o.seq) for o, _, _ in nxt], **i32)
            if joint_ctc:
                js = torch.tensor([j for _, _, j in nxt], dtype=torch.long, device=dev)
                r_prev = r_out[parents, js].contiguous()
                prev_ctc = psi_c[parents, js].contiguous()
            for slot, (o, _, _) in enumerate(nxt):
                o.slot = slot
                live.append(o)
        final += [Hypothesis(o.seq, o.scores) for o in live]
        final.sort(key=lambda o: o.avgScore(), reverse=True)
        return final[:beam]
