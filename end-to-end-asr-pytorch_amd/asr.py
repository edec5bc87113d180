"""Host-side mirror of the reference's model interface (src/asr.py) on top of the HIP kernels.

Same constructor / forward signatures and the same state_dict names as the reference's Seq2Seq, Listener,
Attention and Speller, so checkpoints and the Trainer code carry over; every tensor op runs in liblas_hip.so.
All parameters (and their gradients) are views of one flat fp32 buffer each, which the fused optimiser and the
RCCL gradient all-reduce treat as a single vector.

Not built here (out of scope): multi-head attention (broken in the reference, asr.py:436), GRU cells.
"""
import math
import random

import torch
import torch.nn as nn

from . import ops
from .decoder import DecoderFn, weight_names
from . import vgg as _vgg


class FlatParams:
    """Carves nn.Parameters (and their .grad) out of two flat fp32 buffers."""

    def __init__(self):
        self.specs = []           # (name, shape)

    def add(self, name, *shape):
        self.specs.append((name, tuple(shape)))

    ALIGN = 64          # floats: every parameter starts 256-byte aligned (vector loads in the kernels)

    def build(self, module, device):
        """Adjacent per-direction LSTM parameters must stay contiguous (kernel-facing concatenated views), so
        alignment padding is only inserted where the next spec is not a `_reverse` twin."""
        offs, off = [], 0
        for i, (name, shape) in enumerate(self.specs):
            if not name.endswith('_reverse'):
                off = (off + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            offs.append(off)
            off += math.prod(shape)
        total = (off + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        module.flat_params = torch.zeros(total, dtype=torch.float32, device=device)
        module.flat_grads = torch.zeros(total, dtype=torch.float32, device=device)
        # bf16 shadow of the weights: operand of the GEMMs in bf16 mode, rewritten by the fused optimiser kernels
        module.flat_params16 = torch.zeros(total, dtype=torch.bfloat16, device=device)
        module.param_slices = {}
        for (name, shape), off in zip(self.specs, offs):
            k = math.prod(shape)
            p = nn.Parameter(module.flat_params[off:off + k].view(shape))
            p.grad = module.flat_grads[off:off + k].view(shape)
            p._bf16 = module.flat_params16[off:off + k].view(shape)
            module.param_slices[name] = (off, k, shape)
            # register under the reference's dotted name
            obj = module
            parts = name.split('.')
            for q in parts[:-1]:
                if not hasattr(obj, q):
                    obj.add_module(q, nn.Module())
                obj = getattr(obj, q)
            obj.register_parameter(parts[-1], p)
        module.n_params = sum(math.prod(s) for _, s in self.specs)


def param_shapes(example_input, output_dim, model_para):
    """{reference parameter name: shape} of the architecture a YAML describes (no device needed)."""
    class _Probe(Seq2Seq):
        def _finish(self, fp, device):
            self._specs = dict(fp.specs)
    return _Probe(example_input, output_dim, model_para, device='cpu')._specs


class Seq2Seq(nn.Module):
    """Listen-Attend-Spell with optional CTC head; reference src/asr.py:18-153."""

    def __init__(self, example_input, output_dim, model_para, device=None):
        super().__init__()
        device = torch.device(device if device is not None else 'cuda')
        enc, att, dec = model_para['encoder'], model_para.get('attention'), model_para.get('decoder')
        self.vgg = 'VGG' in enc['enc_type']
        if enc['rnn_cell'].upper() != 'LSTM':
            raise NotImplementedError('only LSTM encoder cells are built')
        self.dims = [int(v) for v in str(enc['dim']).split('_')]
        self.srs = [int(v) for v in str(enc['sample_rate']).split('_')]
        drops = [float(v) for v in str(enc['dropout']).split('_')]
        assert len(self.srs) == len(drops) == len(self.dims), 'Number of layer mismatch'      # asr.py:279-280
        # encoder `dropout` is accepted and has no effect, exactly as in the reference: it is handed to
        # nn.LSTM(num_layers=1, dropout=p) (asr.py:473), which applies dropout only BETWEEN stacked layers
        self.bidir = 'Bi' in enc['enc_type']
        if 'RNN' not in enc['enc_type']:
            raise ValueError('Unsupported Encoder Type: ' + enc['enc_type'])
        self.concat = enc['sample_style'] == 'concat'
        if enc['sample_style'] not in ('concat', 'drop'):
            raise ValueError('Unsupported Sample Style: ' + enc['sample_style'])
        self.ND = 2 if self.bidir else 1
        self.joint_ctc = model_para['optimizer']['joint_ctc'] > 0
        self.joint_att = model_para['optimizer']['joint_ctc'] < 1
        self.ctc_weight = model_para['optimizer']['joint_ctc']
        in_dim = int(example_input.shape[-1])

        fp = FlatParams()
        if self.vgg:                                                    # Listener.__init__, asr.py:285-288
            cin, _, in_dim = _vgg.check_dim(in_dim)
            for i, (co, ci) in enumerate(((64, cin), (64, 64), (128, 64), (128, 128)), 1):
                fp.add(f'encoder.vgg_extractor.conv{i}.weight', co, ci, 3, 3)
                fp.add(f'encoder.vgg_extractor.conv{i}.bias', co)
        self.enc_in = []
        sfx = ['', '_reverse'] if self.bidir else ['']
        for l, (H, sr) in enumerate(zip(self.dims, self.srs)):
            self.enc_in.append(in_dim)
            # kernel-facing concatenations are contiguous: [w_ih | w_ih_rev], [w_hh | w_hh_rev], ...
            for kind, shape in (('weight_ih', (4 * H, in_dim)), ('weight_hh', (4 * H, H)), ('bias_ih', (4 * H,)),
                                ('bias_hh', (4 * H,))):
                for s in sfx:
                    fp.add(f'encoder.layer{l}.layer.{kind}_l0{s}', *shape)
            out = H * self.ND * (sr if self.concat else 1)
            fp.add(f'encoder.proj{l}.weight', out, out)
            fp.add(f'encoder.proj{l}.bias', out)
            in_dim = out
        self.enc_out_dim = in_dim
        if self.joint_att:
            if att['num_head'] != 1:
                raise NotImplementedError('multi-head attention is broken in the reference (asr.py:436) and not built')
            if not att['proj']:
                raise NotImplementedError('attention without projection is not built')
            if dec['rnn_cell'] != 'LSTMCell':
                raise NotImplementedError('only LSTMCell decoders are built')
            self.dec_dropout = float(dec['dropout'])                  # Speller dropout, asr.py:327,353,355
            self.drop_calls = 0
            self.att_mode = att['att_mode'].lower()
            if self.att_mode not in ('dot', 'loc'):
                raise ValueError('Unsupported Attention Mode: ' + self.att_mode)
            self.A, self.dec_dim, self.dec_layers = int(att['dim']), int(dec['dim']), int(dec['layer'])
            E, C, A = self.enc_out_dim, self.dec_dim, self.A
            fp.add('attention.phi.weight', A, C)
            fp.add('attention.psi.weight', A, E)
            fp.add('attention.psi.bias', A)
            if self.att_mode == 'loc':
                fp.add('attention.loc_conv.weight', 10, 1, 201)
                fp.add('attention.loc_proj.weight', A, 10)
                fp.add('attention.gen_energy.weight', 1, A)
                fp.add('attention.gen_energy.bias', 1)
            for l in range(self.dec_layers):
                fp.add(f'decoder.layer{l}.weight_ih', 4 * C, (E + C) if l == 0 else C)
                fp.add(f'decoder.layer{l}.weight_hh', 4 * C, C)
                fp.add(f'decoder.layer{l}.bias_ih', 4 * C)
                fp.add(f'decoder.layer{l}.bias_hh', 4 * C)
            fp.add('embed.weight', output_dim, C)
            fp.add('char_trans.weight', output_dim, C)
            fp.add('char_trans.bias', output_dim)
            self.char_dim = output_dim
        if self.joint_ctc:
            fp.add('ctc_layer.weight', output_dim, self.enc_out_dim)
            fp.add('ctc_layer.bias', output_dim)
        self._finish(fp, device)

    def _finish(self, fp, device):
        fp.build(self, device)
        self.status = torch.zeros(1, dtype=torch.int32, device=device)
        self.ctc_branch = False          # set by Trainer.train_step for the duration of its forward (see forward())
        self.sample_seed = 0
        self.init_parameters()

    # -- parameter access ----------------------------------------------------------------------------------
    def P(self, name):
        obj = self
        for q in name.split('.'):
            obj = getattr(obj, q)
        return obj

    def _cat(self, prefix, kind, shape, grads=False, shadow=False):
        """Kernel-facing view over the adjacent per-direction parameters (or their gradients / bf16 shadow), no copy."""
        off, k, _ = self.param_slices[f'{prefix}.{kind}_l0']
        return (self.flat_params16 if shadow else self.flat_grads if grads else self.flat_params)[off:off + k * self.ND].view(shape)

    def grad_ready_offsets(self):
        """Flat-gradient offsets at which backward reports "everything from here on is final" (ops._GRAD_READY, fired by
        each encoder layer's BPTT node with its first gradient view), in the order backward fires them: top layer first.
        dist.exchange_without_backward replays them on a rank that has no backward to run."""
        return [self.param_slices[f'encoder.layer{l}.layer.weight_ih_l0'][0] for l in reversed(range(len(self.dims)))]

    def sync_bf16(self):
        """Refresh the weights' bf16 shadow after the fp32 weights were written by anything but the fused optimiser."""
        if self.flat_params.is_cuda:
            ops.cast_bf16(self.flat_params, out=self.flat_params16)

    def init_parameters(self):
        """Same scheme as reference asr.py:114-153 (LeCun normal; decoder forget-gate bias_ih = 1; embed N(0,1))."""
        with torch.no_grad():
            for name, p in self.named_parameters():
                if p.dim() == 1:
                    p.zero_()
                elif p.dim() == 2:
                    p.copy_(torch.randn(p.shape) * (1.0 / math.sqrt(p.shape[1])))
                else:
                    n = p.shape[1] * math.prod(p.shape[2:])
                    p.copy_(torch.randn(p.shape) * (1.0 / math.sqrt(n)))
            if self.joint_att:
                self.P('embed.weight').copy_(torch.randn(self.P('embed.weight').shape))
                for l in range(self.dec_layers):
                    b = self.P(f'decoder.layer{l}.bias_ih')
                    n = b.shape[0]
                    b[n // 4:n // 2].fill_(1.0)
        self.sync_bf16()

    def load_reference_state(self, state):
        """Copy weights given under the reference's state_dict names (numpy arrays or tensors)."""
        with torch.no_grad():
            for name, p in self.named_parameters():
                p.copy_(torch.as_tensor(state[name]).to(p.device, torch.float32).view(p.shape))
        self.sync_bf16()

    # -- encoder ---------------------------------------------------------------------------------------------
    def encode(self, x, lens_dev, lens_host):
        """Listener.forward, reference asr.py:311-317; x [B,T,D] batch-major -> enc [B,T',E], enc_len (host list)."""
        lens_dev = lens_dev.clone()
        lens_host = list(lens_host)
        if self.vgg:                                       # asr.py:312-313; time-major from here on
            h = _vgg.VGGFn.apply(x, True, *[self.P('encoder.vgg_extractor.' + n) for n in _vgg.NAMES])
            lens_host = [v // 4 for v in lens_host]        # view_input, asr.py:535
            lens_dev = torch.div(lens_dev, 4, rounding_mode='floor').to(torch.int32)
        else:
            h = ops.Transpose01Fn.apply(x)
        for l, (H, sr) in enumerate(zip(self.dims, self.srs)):
            pre = f'encoder.layer{l}.layer'
            I_ = self.enc_in[l]
            T_l = max(lens_host)                 # pad_packed_sequence trims to the longest utterance (asr.py:483)
            if h.shape[0] > T_l:
                h = ops.narrow_rows(h, T_l)
            leaves = []
            for kind in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh'):
                for s in (['', '_reverse'] if self.bidir else ['']):
                    leaves.append(self.P(f'{pre}.{kind}_l0{s}'))
            cats = (self._cat(pre, 'weight_ih', (self.ND * 4 * H, I_)), self._cat(pre, 'weight_hh', (self.ND, 4 * H, H)),
                    self._cat(pre, 'bias_ih', (self.ND * 4 * H,)), self._cat(pre, 'bias_hh', (self.ND * 4 * H,)),
                    self._cat(pre, 'weight_ih', (self.ND * 4 * H, I_), shadow=True))
            cat_grads = (self._cat(pre, 'weight_ih', (self.ND * 4 * H, I_), True), self._cat(pre, 'weight_hh', (self.ND, 4 * H, H), True),
                         self._cat(pre, 'bias_ih', (self.ND * 4 * H,), True), self._cat(pre, 'bias_hh', (self.ND * 4 * H,), True))
            h = ops.lstm_layer_leaves(h, lens_dev, cats, sr, self.concat, self.status, self.ND, leaves, cat_grads)
            if sr > 1:
                lens_host = [int(v / sr) for v in lens_host]              # asr.py:497
                lens_dev = torch.div(lens_dev, sr, rounding_mode='floor').to(torch.int32)
            h = ops.linear(h, self.P(f'encoder.proj{l}.weight'), self.P(f'encoder.proj{l}.bias'), act=1)
        return ops.Transpose01Fn.apply(h), lens_dev, lens_host

    # -- decoding --------------------------------------------------------------------------------------------
    def beam_decode(self, audio_feature, decode_step, state_len, decode_beam_size):
        """reference asr.py:155-258: top-N hypotheses of one utterance, joint CTC/attention scoring."""
        from .beam import beam_decode
        return beam_decode(self, audio_feature, decode_step, state_len, decode_beam_size)

    # -- full forward ----------------------------------------------------------------------------------------
    def forward(self, audio_feature, decode_step, tf_rate=0.0, teacher=None, state_len=None, state_len_dev=None, replay=None):
        """reference asr.py:58-112.  Returns (ctc_output [B,T',V]|None, encode_len list[int],
        att_output [B,L,V]|None, att_maps [ (B,L,T') ]|None).
        state_len_dev (not in the reference): the int32 device copy of a host list `state_len` when the caller has both
        (Trainer.train_step) -- building it here from the list is a pageable H2D copy, which makes the host wait for
        everything queued on the stream, i.e. for the whole previous step.
        replay (not in the reference; tests): dict(flips=[bool] * decode_step[, tokens={t: LongTensor[B]}]) -- the recorded
        coin flips of asr.py:96 instead of fresh ones and, for the steps whose flip said "sample", the recorded draws of
        asr.py:99 instead of the device sampler's.  A step fed a recorded draw is a teacher-forced step on that token
        (the draw is not differentiated through), which is how it is run."""
        x = audio_feature
        if not x.is_cuda:
            raise ops._lib.LasError('Seq2Seq.forward needs HIP device tensors (no CPU path)')
        if state_len is None:
            lens_dev = ops.infer_lengths(x)
            lens_host = lens_dev.cpu().tolist()
        elif torch.is_tensor(state_len):
            lens_dev = state_len.to(device=x.device, dtype=torch.int32)
            lens_host = state_len.cpu().tolist()
        else:
            lens_host = [int(v) for v in state_len]
            if state_len_dev is not None:
                lens_dev = state_len_dev.to(device=x.device, dtype=torch.int32)
            else:
                lens_dev = torch.tensor(lens_host, dtype=torch.int32, device=x.device)
        T = max(lens_host)                                   # pad_packed_sequence trims to the longest (asr.py:483)
        if T < x.shape[1] and not self.vgg:                  # the VGG front-end convolves over the padding too
            x = x[:, :T].contiguous()
        enc, enc_len_dev, enc_len = self.encode(x.float(), lens_dev, lens_host)
        ctc_output = att_output = att_maps = None
        if self.joint_ctc:
            if self.ctc_branch and self.joint_att and ops._BRANCH['enabled']:
                # Trainer.train_step only: the CTC head runs on a stream of its own beside the attend-and-spell loop; the
                # tensor carries the stream (`_branch`) and ops.joint_loss keeps the CTC loss on it and joins.  Autograd
                # runs a node's backward on the stream of its forward, so the head's backward stays there as well.
                cs = ops.branch_stream()
                ops.fork_to(cs)
                with torch.cuda.stream(cs):
                    ctc_output = ops.linear(enc, self.P('ctc_layer.weight'), self.P('ctc_layer.bias'))
                enc.record_stream(cs)
                ctc_output._branch = cs
                ops._BRANCH['pending'] = cs          # (joint_loss joins this stream even if the tag above gets lost)
            else:
                ctc_output = ops.linear(enc, self.P('ctc_layer.weight'), self.P('ctc_layer.bias'))
        if self.joint_att:
            L = int(decode_step)
            # one coin flip per step for the whole batch (asr.py:96); the flip after step t picks step t+1's input
            if teacher is not None:
                flips = [random.random() <= tf_rate for _ in range(L)]
                if replay is not None:
                    flips = [bool(v) for v in replay['flips']][:L]
                mode = [1] + [1 if f else 0 for f in flips[:L - 1]]
                y = teacher.to(device=x.device, dtype=torch.int64).contiguous()
                if y.shape[1] < L:
                    raise ValueError('teacher shorter than decode_step')
                if replay is not None and replay.get('tokens'):
                    y = y.clone()
                    for t, tok in replay['tokens'].items():      # the draw after step t feeds step t + 1
                        if t + 1 < L:
                            y[:, t + 1] = torch.as_tensor(tok).to(device=x.device, dtype=torch.int64)
                            mode[t + 1] = 1
            else:
                mode = [1] + [2] * (L - 1)
                y = None
            step_mode = None if all(m == 1 for m in mode) and y is not None else mode
            ops.twin(enc, make=True)          # enc's bf16 copy, attached to enc: the psi GEMM and the per-step attention backward read it
            psi = ops.linear(enc, self.P('attention.psi.weight'), self.P('attention.psi.bias'), act=1)
            loc = self.att_mode == 'loc'
            ws = [self.P(n) for n in weight_names(self.dec_layers, loc)]
            self.sample_seed += 1
            seed = (torch.initial_seed() + 1000003 * self.sample_seed) & 0x7fffffff
            if self.training and self.dec_dropout > 0:
                self.drop_calls += 1
                seed = (seed, self.dec_dropout, (torch.initial_seed() * 2654435761 + self.drop_calls) & 0xffffffff)
            h_top, att = DecoderFn.apply(enc, psi, enc_len_dev, y, L, self.dec_layers, loc, step_mode,
                                         dict(seed=seed, status=self.status), *ws)
            logits = ops.linear(h_top, self.P('char_trans.weight'), self.P('char_trans.bias'))     # [L,B,V]
            att_output = ops.Transpose01Fn.apply(logits)
            att_maps = [ops.transpose01(att)]
        self.last_enc_len_dev = enc_len_dev
        return ctc_output, enc_len, att_output, att_maps
