"""CPU oracle for the LAS training hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

A functional, torch-CPU fp32 restatement of the reference's train-step
arithmetic.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this module; the product package never does.

Pinned by golden vectors generated from the imported reference
(tools/gen_golden.py -> tests/golden/*.npz, checked in tests/test_oracle.py).

Reference lines restated (all under /root/reference):
  infer_lengths      src/solver.py:134-136
  lstm_dir / bilstm  src/asr.py:466-501   (nn.LSTM packed semantics; gate order i,f,g,o)
  listener           src/asr.py:266-317   (RNNLayer -> tanh(Linear) per layer)
  vgg_extractor      src/asr.py:507-558   (4x conv3x3+ReLU, 2x MaxPool2d(2); time /4, freq /4)
  attention_*        src/asr.py:370-462   (dot / loc, softmax scale 2.0, mask -inf)
  speller_step       src/asr.py:320-357
  seq2seq_forward    src/asr.py:58-112
  joint_loss         src/solver.py:144-164
  clip_grad_norm     src/solver.py:178   (torch.nn.utils.clip_grad_norm_ semantics)
  adam / adadelta    src/solver.py:105-106 (torch.optim.{Adam,Adadelta}(lr, eps=1e-8) defaults)
  ctc_*              torch.nn.CTCLoss(blank=0,'mean') as called at src/solver.py:93,160
                     (third-party ATen ctc_loss_cpu; restated explicitly in ctc_ref.c / ctc_numpy)

Weights use the reference's state_dict names, e.g.
  encoder.layer0.layer.weight_ih_l0[_reverse], encoder.proj0.weight,
  attention.{phi,psi,loc_conv,loc_proj,gen_energy}.*, decoder.layer0.weight_ih,
  embed.weight, char_trans.*, ctc_layer.*
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

LOC_C, LOC_K = 10, 100      # asr.py:395-396
ATT_SCALE = 2.0             # asr.py:410
GRAD_CLIP = 5.0             # solver.py:20


# ----------------------------------------------------------------------------- config helpers
def parse_cfg(model_para):
    enc = model_para['encoder']
    dims = [int(v) for v in str(enc['dim']).split('_')]
    srs = [int(v) for v in str(enc['sample_rate']).split('_')]
    bidir = 'Bi' in enc['enc_type']
    style = enc['sample_style']
    ctc_w = float(model_para['optimizer']['joint_ctc'])
    return dict(dims=dims, srs=srs, bidir=bidir, style=style, ctc_w=ctc_w, vgg='VGG' in enc['enc_type'],
                att_mode=model_para['attention']['att_mode'].lower(),
                dec_layers=int(model_para['decoder']['layer']), dec_dim=int(model_para['decoder']['dim']))


def infer_lengths(x):
    """solver.py:134  frames whose feature-sum != 0."""
    return [int(v) for v in (x.sum(-1) != 0).sum(-1)]


# ----------------------------------------------------------------------------- encoder
def lstm_dir(x, lens, w_ih, w_hh, b_ih, b_hh, reverse, bf16_operands=False):
    """One direction of a packed LSTM, explicit time loop.  x (B,T,I) -> (B,T,H), zeros at t>=len.
    bf16_operands (forward checks only): the operands of both matrix products are rounded to bf16 (RNE), sums and
    state stay fp32 -- the arithmetic of the product's bf16 mode, so that mode can be checked to 1e-3 instead of 5e-2."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    r = _rb if bf16_operands else (lambda t_: t_)
    w_hh = r(w_hh)
    xp = r(x) @ r(w_ih).t() + (b_ih + b_hh)
    lens_t = torch.as_tensor(lens)
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    outs = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        g = xp[:, t] + r(h) @ w_hh.t()
        i, f, gg, o = g.chunk(4, dim=-1)
        c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h_new = torch.sigmoid(o) * torch.tanh(c_new)
        m = (t < lens_t).to(x.dtype).unsqueeze(1)
        c = m * c_new + (1 - m) * c          # state only advances inside the utterance
        h = m * h_new + (1 - m) * h
        outs[t] = m * h_new
    return torch.stack(outs, dim=1)


def lstm_dir_fast(x, lens, w_ih, w_hh, b_ih, b_hh, reverse):
    """Same arithmetic through torch's packed LSTM op (what the reference's nn.LSTM delegates to)."""
    from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
    B, T, _ = x.shape
    H = w_hh.shape[1]
    if reverse:
        idx = torch.zeros(B, T, dtype=torch.long)
        for b, l in enumerate(lens):
            idx[b, :l] = torch.arange(l - 1, -1, -1)
            idx[b, l:] = torch.arange(l, T)
        x = torch.gather(x, 1, idx.unsqueeze(-1).expand(-1, -1, x.shape[-1]))
    pk = pack_padded_sequence(x, torch.as_tensor(lens), batch_first=True)
    h0 = x.new_zeros(1, B, H)
    out, _, _ = torch._VF.lstm(pk.data, pk.batch_sizes, (h0, h0.clone()), [w_ih, w_hh, b_ih, b_hh],
                               True, 1, 0.0, False, False)
    y, _ = pad_packed_sequence(torch.nn.utils.rnn.PackedSequence(out, pk.batch_sizes), batch_first=True, total_length=T)
    if reverse:
        y = torch.gather(y, 1, idx.unsqueeze(-1).expand(-1, -1, H))
    return y


def rnn_layer(x, lens, W, prefix, sr, style, bidir, fast=False, bf16_operands=False):
    """asr.py:476-501.  Returns (y, out_lens)."""
    f = lstm_dir_fast if fast else lstm_dir
    if bf16_operands:
        f = lambda *a: lstm_dir(*a, bf16_operands=True)
    T = max(lens)                                  # pad_packed_sequence trims to the longest
    x = x[:, :T]
    p = prefix + '.layer.'
    y = f(x, lens, W[p + 'weight_ih_l0'], W[p + 'weight_hh_l0'], W[p + 'bias_ih_l0'], W[p + 'bias_hh_l0'], False)
    if bidir:
        yr = f(x, lens, W[p + 'weight_ih_l0_reverse'], W[p + 'weight_hh_l0_reverse'],
               W[p + 'bias_ih_l0_reverse'], W[p + 'bias_hh_l0_reverse'], True)
        y = torch.cat([y, yr], dim=-1)
    if sr > 1:
        B, T, Fd = y.shape
        if style == 'drop':
            y = y[:, ::sr]
        elif style == 'concat':
            if T % sr:
                y = y[:, :T - T % sr]
            y = y.reshape(B, T // sr, Fd * sr)
        else:
            raise ValueError(style)
        lens = [int(l / sr) for l in lens]
    return y, list(lens)


def vgg_dims(d):
    """check_dim, asr.py:522-531: (in_channel, freq_dim, out_dim)."""
    if d % 13 == 0:
        return d // 13, 13, (13 // 4) * 128
    if d % 40 == 0:
        return d // 40, 40, (40 // 4) * 128
    raise ValueError('Acoustic feature dimension for VGG should be 13/26/39(MFCC) or 40/80/120(Fbank) but got %d' % d)


def _rb(t):
    return t.to(torch.bfloat16).to(torch.float32)


class _ConvBf16Operands(torch.autograd.Function):
    """conv3x3(pad 1) whose matrix-product operands are rounded to bf16 (RNE) with fp32 accumulation, forward and
    backward: the arithmetic of the product's bf16 mode.  Used only to check that mode tightly: with the reference's
    pure-fp32 arithmetic the ReLU masks / max-pool winners of near-ties differ, which moves whole gradient terms."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return F.conv2d(_rb(x), _rb(w), b, padding=1)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = torch.nn.grad.conv2d_input(x.shape, _rb(w), _rb(g), padding=1)
        gw = torch.nn.grad.conv2d_weight(_rb(x), w.shape, _rb(g), padding=1)
        return gx, gw, g.sum((0, 2, 3))


def vgg_extractor(x, lens, W, prefix='encoder.vgg_extractor', bf16_operands=False):
    """asr.py:533-558.  x (B,T,D) -> (B, T//4, 128*(F//4)); lengths //4; the T%4 tail frames are dropped."""
    conv = _ConvBf16Operands.apply if bf16_operands else (lambda h, w, b: F.conv2d(h, w, b, padding=1))
    B, T, D = x.shape
    cin, fd, od = vgg_dims(D)
    lens = [v // 4 for v in lens]
    if T % 4:
        x = x[:, :T - T % 4]
    h = x.reshape(B, x.shape[1], cin, fd).transpose(1, 2)
    h = F.relu(conv(h, W[f'{prefix}.conv1.weight'], W[f'{prefix}.conv1.bias']))
    h = F.relu(conv(h, W[f'{prefix}.conv2.weight'], W[f'{prefix}.conv2.bias']))
    h = F.max_pool2d(h, 2, stride=2)
    h = F.relu(conv(h, W[f'{prefix}.conv3.weight'], W[f'{prefix}.conv3.bias']))
    h = F.relu(conv(h, W[f'{prefix}.conv4.weight'], W[f'{prefix}.conv4.bias']))
    h = F.max_pool2d(h, 2, stride=2)
    h = h.transpose(1, 2)
    return h.reshape(B, h.shape[1], od), lens


def listener(x, lens, W, cfg, fast=False):
    """asr.py:311-317."""
    if cfg.get('vgg'):
        x, lens = vgg_extractor(x, lens, W)
    for l, (sr, _) in enumerate(zip(cfg['srs'], cfg['dims'])):
        x, lens = rnn_layer(x, lens, W, f'encoder.layer{l}', sr, cfg['style'], cfg['bidir'], fast)
        x = torch.tanh(F.linear(x, W[f'encoder.proj{l}.weight'], W[f'encoder.proj{l}.bias']))
    return x, lens


# ----------------------------------------------------------------------------- attention
def attention_init(enc, lens, W):
    """First-call work of Attention.forward, asr.py:412-419 (+ loc init :444-449)."""
    B, Tp, _ = enc.shape
    mask = torch.zeros(B, Tp, dtype=torch.bool)
    prev = enc.new_zeros(B, Tp)
    for b, l in enumerate(lens):
        mask[b, l:] = True
        prev[b, :l] = 1.0 / l
    psi = torch.tanh(F.linear(enc, W['attention.psi.weight'], W['attention.psi.bias']))
    return dict(mask=mask, psi=psi, prev=prev)


def attention_step(h, enc, st, W, mode, bf16_operands=False):
    """asr.py:421-457.  Returns (score (B,T'), context (B,E)); updates st['prev'] in loc mode.
    bf16_operands (checks of the product's bf16 mode only): the operands of the matrix products the HIP path runs on the
    matrix cores (phi h, W_lp f, the context's enc) are rounded to bf16 (RNE), everything else stays fp32; the cast is the
    identity for autograd, so the backward pass is the exact derivative of that forward."""
    r = _rb if bf16_operands else (lambda t_: t_)
    q = torch.tanh(F.linear(r(h), r(W['attention.phi.weight'])))
    if mode == 'dot':
        e = torch.einsum('bta,ba->bt', st['psi'], q)
    elif mode == 'loc':
        f = F.conv1d(st['prev'].unsqueeze(1), W['attention.loc_conv.weight'], padding=LOC_K)   # (B,C,T')
        u = torch.tanh(F.linear(r(f.transpose(1, 2)), r(W['attention.loc_proj.weight'])))     # (B,T',A)
        e = F.linear(torch.tanh(st['psi'] + q.unsqueeze(1) + u),
                     W['attention.gen_energy.weight'], W['attention.gen_energy.bias']).squeeze(2)
    else:
        raise ValueError(mode)
    e = e.masked_fill(st['mask'], -float('inf'))
    a = torch.softmax(e * ATT_SCALE, dim=-1)
    if mode == 'loc':
        st['prev'] = a
    ctx = torch.einsum('bt,bte->be', a, r(enc))
    return a, ctx


# ----------------------------------------------------------------------------- decoder
def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh, bf16_operands=False):
    r = _rb if bf16_operands else (lambda t_: t_)
    g = F.linear(r(x), r(w_ih), b_ih) + F.linear(r(h), r(w_hh), b_hh)
    i, f, gg, o = g.chunk(4, dim=-1)
    c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    return torch.sigmoid(o) * torch.tanh(c), c


def speller_step(inp, hs, cs, W, n_layers, masks=None, bf16_operands=False):
    """asr.py:352-357.  masks (dropout replay): masks[0] multiplies the cell-0 input (asr.py:353), masks[l] the
    recurrent state h_l of layer l >= 1 (asr.py:355: dropout sits on the hidden input, not on the layer input); each
    already scaled by 1/(1-p).  None = dropout 0 / eval."""
    p = 'decoder.layer'
    if masks is not None:
        inp = inp * masks[0]
    hs[0], cs[0] = lstm_cell(inp, hs[0], cs[0], W[p + '0.weight_ih'], W[p + '0.weight_hh'],
                             W[p + '0.bias_ih'], W[p + '0.bias_hh'], bf16_operands)
    for l in range(1, n_layers):
        hl = hs[l] * masks[l] if masks is not None else hs[l]
        hs[l], cs[l] = lstm_cell(hs[l - 1], hl, cs[l], W[f'{p}{l}.weight_ih'], W[f'{p}{l}.weight_hh'],
                                 W[f'{p}{l}.bias_ih'], W[f'{p}{l}.bias_hh'], bf16_operands)
    return hs[-1]


def seq2seq_forward(W, cfg, x, decode_step, teacher=None, lens=None, fast=False, use_teacher=None, sampled=None):
    """asr.py:58-112.  `use_teacher[t]` replays the per-step coin flips (asr.py:96); default all True.
    Without a teacher the argmax is fed back (asr.py:102); `sampled[t]` (LongTensor [B]) replays the token the
    scheduled-sampling draw of asr.py:99 produced after step t (it is not differentiated through, as in the reference)."""
    enc, enc_len = listener(x, lens, W, cfg, fast)
    B = x.shape[0]
    ctc_out = att_out = att_map = None
    if cfg['ctc_w'] > 0:
        ctc_out = F.linear(enc, W['ctc_layer.weight'], W['ctc_layer.bias'])
    if cfg['ctc_w'] < 1:
        C, nl = cfg['dec_dim'], cfg['dec_layers']
        emb = W['embed.weight']
        temb = emb[teacher] if teacher is not None else None
        hs = [x.new_zeros(B, C) for _ in range(nl)]
        cs = [x.new_zeros(B, C) for _ in range(nl)]
        st = attention_init(enc, enc_len, W)
        last = emb[torch.zeros(B, dtype=torch.long)]
        logits, maps = [], []
        for t in range(decode_step):
            a, ctx = attention_step(hs[0], enc, st, W, cfg['att_mode'])
            top = speller_step(torch.cat([last, ctx], dim=-1), hs, cs, W, nl)
            cur = F.linear(top, W['char_trans.weight'], W['char_trans.bias'])
            if temb is not None and (use_teacher is None or use_teacher[t]):
                last = temb[:, t + 1]
            elif sampled is not None and t in sampled:
                last = emb[torch.as_tensor(sampled[t]).long()]
            else:
                last = emb[torch.argmax(cur, dim=-1)]
            logits.append(cur)
            maps.append(a)
        att_out = torch.stack(logits, dim=1)
        att_map = torch.stack(maps, dim=1)
    return ctc_out, enc_len, att_out, att_map


# ----------------------------------------------------------------------------- losses
def att_ce_loss(att_pred, y, ans_len):
    """solver.py:144-155."""
    label = y[:, 1:ans_len + 1]
    B, L, V = att_pred.shape
    lp = torch.log_softmax(att_pred, dim=-1)
    nll = -lp.gather(-1, label.unsqueeze(-1)).squeeze(-1)
    nll = nll * (label != 0).to(nll.dtype)                 # ignore_index=0
    per_utt = nll.sum(-1) / (y != 0).sum(-1).to(nll.dtype)
    return per_utt.mean()


def ctc_numpy(logits, label, enc_len, tgt_len, blank=0):
    """Explicit log-space CTC forward-backward, numpy float64 internally, results cast to fp32.
    logits (B,T',V) raw; label (B,L) padded.  Returns nll (B,), log_alpha (B,T',2Lmax+1), grad wrt logits
    of sum_b nll_b * gscale_b with gscale=1 (B,T',V).  Mirrors ATen's ctc_loss semantics (zero_infinity=False)."""
    logits = np.asarray(logits, dtype=np.float64)
    B, Tp, V = logits.shape
    L = label.shape[1]
    S = 2 * L + 1
    m = logits.max(-1, keepdims=True)
    lp = logits - (m + np.log(np.exp(logits - m).sum(-1, keepdims=True)))
    NEG = -np.inf
    nll = np.zeros(B)
    la = np.full((B, Tp, S), NEG)
    grad = np.zeros((B, Tp, V))

    def lse(*a):
        mx = max(a)
        if mx == NEG:
            return NEG
        return mx + math.log(sum(math.exp(v - mx) for v in a))
    for b in range(B):
        T, n = int(enc_len[b]), int(tgt_len[b])
        s_n = 2 * n + 1
        ext = [blank if s % 2 == 0 else int(label[b, s // 2]) for s in range(s_n)]
        al = np.full((T, s_n), NEG)
        al[0, 0] = lp[b, 0, blank]
        if s_n > 1:
            al[0, 1] = lp[b, 0, ext[1]]
        for t in range(1, T):
            for s in range(s_n):
                v = [al[t - 1, s]]
                if s > 0:
                    v.append(al[t - 1, s - 1])
                if s > 1 and ext[s] != blank and ext[s] != ext[s - 2]:
                    v.append(al[t - 1, s - 2])
                al[t, s] = lse(*v) + lp[b, t, ext[s]]
        ll = lse(al[T - 1, s_n - 1], al[T - 1, s_n - 2]) if s_n > 1 else al[T - 1, 0]
        nll[b] = -ll
        la[b, :T, :s_n] = al
        be = np.full((T, s_n), NEG)
        be[T - 1, s_n - 1] = lp[b, T - 1, blank]
        if s_n > 1:
            be[T - 1, s_n - 2] = lp[b, T - 1, ext[s_n - 2]]
        for t in range(T - 2, -1, -1):
            for s in range(s_n):
                v = [be[t + 1, s]]
                if s + 1 < s_n:
                    v.append(be[t + 1, s + 1])
                if s + 2 < s_n and ext[s + 2] != blank and ext[s + 2] != ext[s]:
                    v.append(be[t + 1, s + 2])
                be[t, s] = lse(*v) + lp[b, t, ext[s]]
        if ll == NEG:
            grad[b, :T] = np.nan                 # ATen: inf nll -> NaN grads (zero_infinity=False)
            continue
        for t in range(T):
            occ = np.zeros(V)
            for s in range(s_n):
                ab = al[t, s] + be[t, s]
                if ab > NEG:
                    occ[ext[s]] += math.exp(ab - ll - lp[b, t, ext[s]])
            grad[b, t] = np.exp(lp[b, t]) - occ
    return nll.astype(np.float32), la.astype(np.float32), grad.astype(np.float32)


def ctc_mean_loss(ctc_pred, y, ans_len, enc_len):
    """solver.py:158-160 through torch's CPU CTC (third-party arithmetic the reference calls)."""
    label = y[:, 1:ans_len + 1].contiguous()
    tgt = (y != 0).sum(-1)
    lp = F.log_softmax(ctc_pred.transpose(0, 1), dim=-1)
    return F.ctc_loss(lp, label, torch.as_tensor(enc_len, dtype=torch.long), tgt, blank=0, reduction='mean')


def joint_loss(ctc_pred, att_pred, y, ans_len, enc_len, ctc_w):
    """solver.py:144-164."""
    att = att_ce_loss(att_pred, y, ans_len) if ctc_w < 1 else 0.0
    ctc = ctc_mean_loss(ctc_pred, y, ans_len, enc_len) if ctc_w > 0 else 0.0
    return (1 - ctc_w) * att + ctc_w * ctc, att, ctc


# ----------------------------------------------------------------------------- optimiser
def clip_grad_norm(grads, max_norm=GRAD_CLIP):
    """torch.nn.utils.clip_grad_norm_: total L2 norm, scale by max_norm/(norm+1e-6) clamped to 1."""
    total = torch.sqrt(sum((g.detach() ** 2).sum() for g in grads))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return float(total)


def adam_update(p, g, st, lr, step, b1=0.9, b2=0.999, eps=1e-8):
    st['m'] = b1 * st['m'] + (1 - b1) * g
    st['v'] = b2 * st['v'] + (1 - b2) * g * g
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    return p - (lr / bc1) * st['m'] / (st['v'].sqrt() / math.sqrt(bc2) + eps)


def adadelta_update(p, g, st, lr, rho=0.9, eps=1e-8):
    st['sq'] = rho * st['sq'] + (1 - rho) * g * g
    delta = (st['acc'] + eps).sqrt() / (st['sq'] + eps).sqrt() * g
    st['acc'] = rho * st['acc'] + (1 - rho) * delta * delta
    return p - lr * delta


class RefTrainStep:
    """Stateful oracle for Trainer.exec's step body (solver.py:127-182) at tf_rate=1, dropout 0."""

    def __init__(self, weights, model_para, fast=False):
        self.cfg = parse_cfg(model_para)
        self.W = {k: torch.as_tensor(np.asarray(v)).clone().float().requires_grad_(True) for k, v in weights.items()}
        self.opt = model_para['optimizer']['type']
        self.lr = float(model_para['optimizer']['learning_rate'])
        self.fast = fast
        self.n = 0
        if self.opt == 'Adam':
            self.state = {k: dict(m=torch.zeros_like(v), v=torch.zeros_like(v)) for k, v in self.W.items()}
        elif self.opt == 'Adadelta':
            self.state = {k: dict(sq=torch.zeros_like(v), acc=torch.zeros_like(v)) for k, v in self.W.items()}
        else:
            raise ValueError(self.opt)

    def forward_loss(self, x, y):
        x = torch.as_tensor(x).float()
        y = torch.as_tensor(y).long()
        lens = infer_lengths(x)
        ans_len = int((y != 0).sum(-1).max())
        ctc_pred, enc_len, att_pred, att_map = seq2seq_forward(self.W, self.cfg, x, ans_len, teacher=y, lens=lens, fast=self.fast)
        loss, att, ctc = joint_loss(ctc_pred, att_pred, y, ans_len, enc_len, self.cfg['ctc_w'])
        return loss, att, ctc, dict(ctc_pred=ctc_pred, att_pred=att_pred, att_map=att_map, enc_len=enc_len)

    def step(self, x, y):
        for v in self.W.values():
            v.grad = None
        loss, att, ctc, aux = self.forward_loss(x, y)
        loss.backward()
        used = [k for k, v in self.W.items() if v.grad is not None]
        gn = clip_grad_norm([self.W[k].grad for k in used])
        out = dict(loss=float(loss), att_loss=float(att), ctc_loss=float(ctc), grad_norm=gn,
                   grads={k: self.W[k].grad.clone() for k in used}, aux=aux)
        if not math.isnan(gn):                     # solver.py:179-182
            self.n += 1
            with torch.no_grad():
                for k in used:
                    p, g = self.W[k], self.W[k].grad
                    if self.opt == 'Adam':
                        p.copy_(adam_update(p, g, self.state[k], self.lr, self.n))
                    else:
                        p.copy_(adadelta_update(p, g, self.state[k], self.lr))
        return out
