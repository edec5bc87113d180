"""Trainer with the reference's API (src/solver.py:23-291): Trainer(config, paras).load_data() / .set_model() /
.exec(), same YAML keys, same TensorBoard-style scalar names.  The step body runs entirely in liblas_hip.so
kernels; host<->device traffic per step is one H2D of the batch and ONE small D2H (lengths + ans_len), and the
logged scalars are read back one step late so logging never stalls the stream.

Differences from the reference, all recorded in DESIGN.md: scalars also go to <logdir>/scalars.jsonl
(tensorboardX is used when importable); the best model is saved as a state_dict (+config), not a pickled module;
`--load` resumes from such a checkpoint; `apex: True` selects the built-in fused Adam."""
import json
import math
import os
import time

import torch

from . import ops, dist as ldist
from .asr import Seq2Seq
from .dataset import LoadDataset
from .optim import FlatOptimizer
from .postprocess import Mapper, cal_acc, cal_cer, draw_att

VAL_STEP = 30            # reference solver.py:18-20
TRAIN_WER_STEP = 250
GRAD_CLIP = 5


class ScalarLog:
    """SummaryWriter-shaped sink: add_scalars / add_text / add_image."""

    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.f = open(os.path.join(logdir, 'scalars.jsonl'), 'a')
        self.tb = None
        try:
            from tensorboardX import SummaryWriter
            self.tb = SummaryWriter(logdir)
        except Exception:
            pass
        self.history = []

    def add_scalars(self, name, d, step):
        rec = {'step': int(step), 'name': name, 'values': {k: float(v) for k, v in d.items()}}
        self.history.append(rec)
        self.f.write(json.dumps(rec) + '\n')
        self.f.flush()
        if self.tb:
            self.tb.add_scalars(name, rec['values'], step)

    def add_text(self, name, txt, step):
        self.f.write(json.dumps({'step': int(step), 'name': name, 'text': str(txt)}) + '\n')
        self.f.flush()
        if self.tb:
            self.tb.add_text(name, txt, step)

    def add_image(self, name, img, step):
        self.f.write(json.dumps({'step': int(step), 'name': name, 'image_shape': [int(v) for v in getattr(img, 'shape', ())]}) + '\n')
        self.f.flush()
        if self.tb:
            self.tb.add_image(name, img, step)


class Solver:
    def __init__(self, config, paras):
        self.config, self.paras = config, paras
        self.world, self.rank, self.local_rank = ldist.init() if getattr(paras, 'gpu', True) else (1, 0, 0)
        if not (getattr(paras, 'gpu', True) and torch.cuda.is_available()):
            raise ops._lib.LasError('this build has no CPU path: a HIP device is required (the reference CPU path is '
                                    'restated only as the test oracle under oracle/)')
        self.device = torch.device('cuda', self.local_rank % torch.cuda.device_count())
        torch.cuda.set_device(self.device)
        self.exp_name = paras.name
        if self.exp_name is None:
            self.exp_name = '_'.join([paras.config.split('/')[-1].replace('.yaml', ''), 'sd' + str(paras.seed)])
        os.makedirs(paras.ckpdir, exist_ok=True)
        self.ckpdir = os.path.join(paras.ckpdir, self.exp_name)
        os.makedirs(self.ckpdir, exist_ok=True)
        if str(config['solver'].get('dataset', '')).upper() == 'SYNTHETIC':
            V = config['solver'].get('synthetic', {}).get('V', 63)
            m = {'<sos>': 0, '<eos>': 1}
            m.update({'t%d' % i: i for i in range(2, V)})
            self.mapper = Mapper(mapping=m)
            self.mapper.unit = 'word'
        else:
            self.mapper = Mapper(config['solver']['data_path'])

    def verbose(self, msg):
        if self.paras.verbose and self.rank == 0:
            print('[INFO]', msg)

    def progress(self, msg):
        if self.paras.verbose and self.rank == 0:
            print(msg + '                              ', end='\r')


class Trainer(Solver):
    """Handler for the complete training progress; reference src/solver.py:52-291."""

    def __init__(self, config, paras):
        super().__init__(config, paras)
        self.logdir = os.path.join(paras.logdir, self.exp_name)
        self.log = ScalarLog(self.logdir) if self.rank == 0 else None
        s = config['solver']
        self.valid_step, self.max_step = s['dev_step'], s['total_steps']
        self.tf_start, self.tf_end = s['tf_start'], s['tf_end']
        self.apex = s.get('apex', False)
        self.best_val_ed = 2.0
        self.step = 0
        if config.get('clm', {}).get('enable', False):
            raise NotImplementedError('CLM adversarial training is out of scope (SURVEY.md §2.1)')
        self._pending = None
        self.len_stream = True           # length inference + its D2H on a stream of their own (train_step; tests switch it off)

    # ------------------------------------------------------------------------------------------------ data
    def load_data(self):
        self.verbose('Loading data from ' + str(self.config['solver'].get('data_path')))
        kw = dict(self.config['solver'])
        # data parallel: every training bucket is dealt over the ranks by utterance (dist.shard_bucket); the dev set is
        # small and evaluated whole on every rank (rank 0 logs and saves)
        self.train_set = LoadDataset('train', text_only=False, use_gpu=self.paras.gpu, rank=self.rank, world=self.world, **kw)
        self.dev_set = LoadDataset('dev', text_only=False, use_gpu=self.paras.gpu, **kw)
        for self.sample_x, _ in self.train_set:          # one example sizes the model (solver.py:79)
            break
        if len(self.sample_x.shape) == 4:
            self.sample_x = self.sample_x[0]

    # ------------------------------------------------------------------------------------------------ model
    def set_model(self):
        self.verbose('Init ASR model. Note: validation is done through greedy decoding w/ attention decoder.')
        mp = self.config['asr_model']
        self.asr_model = Seq2Seq(self.sample_x, self.mapper.get_dim(), mp, device=self.device)
        self.ctc_weight = mp['optimizer']['joint_ctc']
        self.asr_opt = FlatOptimizer(self.asr_model, mp['optimizer']['type'], mp['optimizer']['learning_rate'], eps=1e-8,
                                     world_size=self.world)
        if self.paras.load:
            ck = torch.load(self.paras.load, map_location=self.device, weights_only=True)
            self.asr_model.load_reference_state(ck['model'])
            if 'opt' in ck:
                self.asr_opt.load_state_dict(ck['opt'])
            self.step = int(ck.get('step', 0))
            self.best_val_ed = float(ck.get('best_val_ed', 2.0))
        ldist.broadcast_params(self.asr_model.flat_params)
        self.asr_model.sync_bf16()

    # ------------------------------------------------------------------------------------------------ one step
    def train_step(self, x, y, tf_rate, host_lens=None, shard_weight=1.0, inputs_ready=None):
        """The body of the reference's training loop, solver.py:127-182.  x (B,T,D) / y (B,L+2) on the device.
        Returns device scalars (loss, att, ctc) and the predictions; nothing here waits on the GPU except the
        single small read-back of lengths.  `host_lens=(state_len, ans_len)` skips that read-back when the caller
        already knows the lengths on the host (bench.py's kernel-timing pass uses it so that no event bracket
        contains a host bubble; the timed region of the benchmark does NOT).  `shard_weight` = B_local * world /
        B_global (1 for equal shards): both losses are batch means, so the global-batch gradient is the B_local-weighted
        mean of the ranks' gradients.
        `inputs_ready`: None = x / y were produced on the current stream, so the length inference and its read-back
        queue up behind everything on it (the previous step included).  True (x / y are complete) or a torch.cuda.Event
        recorded behind their producer (an H2D copy on another stream, `exec`): the length inference and the read-back
        run on a stream of their own, the host does not wait for the previous step's tail and enqueues this step's
        kernels while that one still runs -- the step has no host bubble at its start."""
        if x.shape[0] == 0:                                       # a bucket smaller than the world: nothing on this rank,
            ldist.exchange_without_backward(self.asr_model)       # but it joins the exchange (the SAME collective sequence as
            self.asr_opt.step(zero_grad=True)                     # its peers' backward issues) and the (global) update
            z = torch.zeros((), device=self.device)
            return z, z, z, None, 0
        if inputs_ready is not None and inputs_ready is not True:
            torch.cuda.current_stream().wait_event(inputs_ready)
        if inputs_ready is not None and host_lens is None and self.len_stream:
            ls = ops.length_stream()
            if inputs_ready is not True:
                ls.wait_event(inputs_ready)
            with torch.cuda.stream(ls):
                lens = ops.infer_lengths(x)                           # solver.py:134, on the device
                ntok = ops.count_nonzero(y)                           # solver.py:136,159
                host = torch.cat([lens, ntok.max().view(1)]).cpu().tolist()  # the one D2H sync of the step (this stream only)
            for t_ in (lens, ntok):                                   # complete (the host has waited for them)
                t_.record_stream(torch.cuda.current_stream())
            for t_ in (x, y):
                t_.record_stream(ls)
            state_len, ans_len = host[:-1], int(host[-1])
        else:
            lens = ops.infer_lengths(x)                               # solver.py:134, on the device
            ntok = ops.count_nonzero(y)                               # solver.py:136,159
            if host_lens is None:
                host = torch.cat([lens, ntok.max().view(1)]).cpu().tolist()  # the one D2H sync of the step
                state_len, ans_len = host[:-1], int(host[-1])
            else:
                state_len, ans_len = list(host_lens[0]), int(host_lens[1])
        self.asr_model.ctc_branch = True                          # CTC head + loss beside the attend-and-spell loops (ops.branch_stream)
        try:
            ctc_pred, enc_len, att_pred, _ = self.asr_model(x, ans_len, tf_rate=tf_rate, teacher=y, state_len=state_len,
                                                             state_len_dev=lens)
        finally:
            self.asr_model.ctc_branch = False
        loss, att_loss, ctc_loss = ops.joint_loss(att_pred, ctc_pred, y, ntok, self.asr_model.last_enc_len_dev, ans_len,
                                                  self.ctc_weight)
        ldist.backward_with_overlap(loss if shard_weight == 1.0 else loss * shard_weight, self.asr_model)   # solver.py:177
        self.asr_opt.step(zero_grad=True)                         # clip 5, NaN guard, update (solver.py:178-182)
        return loss, att_loss, ctc_loss, att_pred, ans_len

    def exec(self):
        self.verbose('Training set total ' + str(len(self.train_set)) + ' batches.')
        if self.device.type == 'cuda':                   # the dependency chain on a high-priority stream (ops.main_stream)
            ms = ops.main_stream()
            ms.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(ms):
                self._exec()
            torch.cuda.current_stream().wait_stream(ms)
        else:
            self._exec()

    def _exec(self):
        self.asr_opt.zero_grad()
        while self.step < self.max_step:
            for x, y in self.train_set:
                self.progress('Training step - ' + str(self.step))
                tf_rate = self.tf_start - self.step * (self.tf_start - self.tf_end) / self.max_step
                assert len(x.shape) == 4, 'Bucketing should cause acoustic feature to have shape 1xBxTxD'
                assert len(y.shape) == 3, 'Bucketing should cause label have to shape 1xBxT'
                ready = None
                if self.device.type == 'cuda':                   # H2D on a copy stream: not queued behind the previous step
                    with torch.cuda.stream(ops.copy_stream()):
                        x = x.squeeze(0).to(device=self.device, dtype=torch.float32, non_blocking=True)
                        y = y.squeeze(0).to(device=self.device, dtype=torch.long, non_blocking=True)
                        ready = torch.cuda.Event()
                        ready.record(ops.copy_stream())
                    for t_ in (x, y):
                        t_.record_stream(torch.cuda.current_stream())
                else:
                    x = x.squeeze(0).to(device=self.device, dtype=torch.float32, non_blocking=True)
                    y = y.squeeze(0).to(device=self.device, dtype=torch.long, non_blocking=True)
                gB = getattr(self.train_set, 'last_global_B', None) or int(x.shape[0]) * self.world
                loss, att_loss, ctc_loss, att_pred, ans_len = self.train_step(x, y, tf_rate, inputs_ready=ready,
                                                                              shard_weight=int(x.shape[0]) * self.world / gB)
                self._log_train(loss, att_loss, ctc_loss, att_pred, y, ans_len)
                if self.step % self.valid_step == 0:
                    self.valid()
                self.step += 1
                if self.step > self.max_step:
                    break
        self._flush_log()

    # ------------------------------------------------------------------------------------------------ logging
    def _log_train(self, loss, att_loss, ctc_loss, att_pred, y, ans_len):
        """Scalars of solver.py:185-190, read back one step late (pinned, non-blocking)."""
        vals = [loss.detach().view(1), att_loss.view(1), ctc_loss.view(1), self.asr_opt.norm3,
                self.asr_model.status.float()]             # persistent-kernel hand-off status rides along: no extra sync
        if att_pred is not None:
            pred = ops.argmax_rows(att_pred)
            vals.append(ops.token_acc(pred, y).view(1))
        dev = torch.cat([v.float() for v in vals])
        buf = torch.empty(dev.shape, dtype=torch.float32, pin_memory=True)
        buf.copy_(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        cer = None
        if self.step % TRAIN_WER_STEP == 0 and att_pred is not None and self.rank == 0:
            label = y[:, 1:ans_len + 1]
            cer = cal_cer(pred.cpu().numpy(), label.cpu().numpy(), mapper=self.mapper)
        self._flush_log()
        self._pending = (self.step, buf, ev, att_pred is not None, cer)

    def _flush_log(self):
        if self._pending is None:
            return
        step, buf, ev, has_att, cer = self._pending
        self._pending = None
        ev.synchronize()
        v = buf.tolist()
        self.last_scalars = dict(step=step, loss=v[0], att=v[1], ctc=v[2], grad_norm=v[3], skipped=bool(v[5]))
        self._check_status(int(v[6]), 'train step %d' % step)
        if v[5]:
            self.verbose('Error : grad norm is NaN @ step ' + str(step))
        if self.log is None:
            return
        d = {}
        if self.ctc_weight < 1:
            d['train_att'] = v[1]
        if self.ctc_weight > 0:
            d['train_ctc'] = v[2]
        d['train_full'] = v[0]
        self.log.add_scalars('loss', d, step)
        if has_att:
            self.log.add_scalars('acc', {'train': v[7]}, step)
        if cer is not None:
            self.log.add_scalars('error rate', {'train': cer}, step)

    def _check_status(self, code=None, where=''):
        """A persistent LSTM kernel whose hand-off spin ran out (LAS_E_TIMEOUT: its workgroups were not all co-resident,
        e.g. two processes on one GPU) leaves its outputs partly unwritten: never train on, validate with or checkpoint
        that.  `code` None reads the flag from the device (one small sync: validation / decoding only)."""
        if code is None:
            code = int(self.asr_model.status.item())
        if code != 0:
            raise ops._lib.LasError('persistent LSTM kernel reported status %d (%s) during %s: results are invalid'
                                    % (code, 'hand-off spin timeout' if code == -4 else 'error', where))

    def write_log(self, val_name, val_dict):
        if self.log is None:
            return
        if 'att' in val_name:
            self.log.add_image(val_name, val_dict, self.step)
        elif 'txt' in val_name or 'hyp' in val_name:
            self.log.add_text(val_name, val_dict, self.step)
        else:
            self.log.add_scalars(val_name, val_dict, self.step)

    # ------------------------------------------------------------------------------------------------ validation
    def valid(self):
        """Greedy-decoding validation, reference solver.py:211-291: eval mode for its duration
        (solver.py:211,287: no Speller dropout), no-teacher argmax feedback for ans_len + VAL_STEP steps, dev losses,
        error rate / accuracy, attention maps + hypotheses + references of the LAST bucket (solver.py:266-276), best
        checkpoint + best_hyp.txt."""
        was_training = self.asr_model.training
        self.asr_model.eval()
        try:
            self._valid_body()
        finally:
            self.asr_model.train(was_training)

    def _valid_body(self):
        val_ctc = val_att = val_acc = val_cer = 0.0
        val_len = 0
        all_pred, all_true = [], []
        pred = label = att_maps = None
        with torch.no_grad():
            for x, y in self.dev_set:
                if len(x.shape) == 4:
                    x = x.squeeze(0)
                if len(y.shape) == 3:
                    y = y.squeeze(0)
                x = x.to(device=self.device, dtype=torch.float32)
                y = y.to(device=self.device, dtype=torch.long)
                lens = ops.infer_lengths(x)
                ntok = ops.count_nonzero(y)
                host = torch.cat([lens, ntok.max().view(1)]).cpu().tolist()
                state_len, ans_len = host[:-1], int(host[-1])
                ctc_pred, enc_len, att_pred, att_maps = self.asr_model(x, ans_len + VAL_STEP, state_len=state_len)
                B = int(x.shape[0])
                att_cut = att_pred[:, :ans_len].contiguous() if att_pred is not None else None
                _, a_l, c_l = ops.joint_loss(att_cut, ctc_pred, y, ntok, self.asr_model.last_enc_len_dev, ans_len,
                                             self.ctc_weight)
                label = y[:, 1:ans_len + 1].cpu().numpy()
                if att_pred is not None:
                    pred = ops.argmax_rows(att_pred).cpu().numpy()
                    val_att += float(a_l) * B
                    t1, t2 = cal_cer(pred, label, mapper=self.mapper, get_sentence=True)
                    all_pred += t1
                    all_true += t2
                    val_acc += cal_acc(pred, label) * B
                    val_cer += cal_cer(pred, label, mapper=self.mapper) * B
                if ctc_pred is not None:
                    val_ctc += float(c_l) * B
                val_len += B
        self._check_status(None, 'validation')
        val_loss = (1 - self.ctc_weight) * val_att + self.ctc_weight * val_ctc
        loss_log = {k: v / val_len for k, v in zip(['dev_full', 'dev_ctc', 'dev_att'], [val_loss, val_ctc, val_att]) if v > 0.0}
        self.write_log('loss', loss_log)
        if self.ctc_weight < 1:
            self.write_log('error rate', {'dev': val_cer / val_len})
            self.write_log('acc', {'dev': val_acc / val_len})
            # attention maps / hypotheses / references of the last bucket (solver.py:266-276)
            val_hyp, val_txt = cal_cer(pred, label, mapper=self.mapper, get_sentence=True)
            for idx, attmap in enumerate(draw_att(att_maps, pred)):
                self.write_log('att_' + str(idx), attmap)
                self.write_log('hyp_' + str(idx), val_hyp[idx])
                self.write_log('txt_' + str(idx), val_txt[idx])
            if val_cer / val_len < self.best_val_ed and self.rank == 0:
                self.best_val_ed = val_cer / val_len
                self.verbose('Best val er       : {:.4f}       @ step {}'.format(self.best_val_ed, self.step))
                self.save_checkpoint(os.path.join(self.ckpdir, 'asr'))
                with open(os.path.join(self.ckpdir, 'best_hyp.txt'), 'w') as f:
                    for t1, t2 in zip(all_pred, all_true):
                        f.write(t1 + ',' + t2 + '\n')

    def save_checkpoint(self, path):
        sd = {k: v.detach().clone() for k, v in self.asr_model.named_parameters()}
        torch.save({'model': sd, 'opt': self.asr_opt.state_dict(), 'step': self.step, 'best_val_ed': self.best_val_ed,
                    'config': json.dumps(self.config['asr_model'])}, path)


class Tester(Solver):
    """Handler for the complete inference progress; reference src/solver.py:293-441.  Beam search keeps the
    reference's batch size of 1 utterance, but every hypothesis of the beam (and every CTC candidate) advances in one
    batched device step (beam.py), so --njobs is accepted and unused."""

    def __init__(self, config, paras):
        super().__init__(config, paras)
        self.verbose('During beam decoding, batch size is set to 1 (the beam is batched on the device).')
        self.njobs = getattr(paras, 'njobs', 1)
        s = config['solver']
        self.decode_step_ratio = s['max_decode_step_ratio']
        self.decode_beam_size = s['decode_beam_size']
        self.decode_file = '_'.join(['decode', 'beam', str(s['decode_beam_size']), 'len', str(s['max_decode_step_ratio'])])
        self.log = None
        self.step = 0
        self.best_val_ed = -1.0                      # never overwrite the checkpoint from the Tester's dev check

    def write_log(self, name, d):
        if isinstance(d, dict):                      # the Tester has no TensorBoard: scalars and texts go to stdout,
            self.verbose('{}: {}'.format(name, {k: round(float(v), 4) for k, v in d.items()}))
        elif isinstance(d, str):                     # attention images are dropped (the reference's Tester.valid logs none)
            self.verbose('{}: {}'.format(name, d))

    def load_data(self):
        self.verbose('Loading testing data ' + str(self.config['solver']['test_set']) + ' from ' +
                     str(self.config['solver'].get('data_path')))
        kw = dict(self.config['solver'])
        self.test_set = LoadDataset('test', text_only=False, use_gpu=self.paras.gpu, **kw)
        self.dev_set = LoadDataset('dev', text_only=False, use_gpu=self.paras.gpu, **kw)
        for self.sample_x, _ in self.test_set:
            break
        if len(self.sample_x.shape) == 4:
            self.sample_x = self.sample_x[0]

    def set_model(self):
        """Load the saved ASR (state_dict checkpoint written by Trainer.save_checkpoint)."""
        path = os.path.join(self.ckpdir, 'asr')
        self.verbose('Load ASR model from ' + path)
        ck = torch.load(path, map_location=self.device, weights_only=True)
        mp = json.loads(ck['config']) if 'config' in ck else self.config['asr_model']
        self.asr_model = Seq2Seq(self.sample_x, self.mapper.get_dim(), mp, device=self.device)
        self.asr_model.load_reference_state(ck['model'])
        self.ctc_weight = mp['optimizer']['joint_ctc']
        s = self.config['solver']
        dcw = s.get('decode_ctc_weight', 0)
        if dcw > 0:                                   # solver.py:317-323
            assert self.asr_model.joint_ctc, 'The ASR was not trained with CTC'
            self.verbose('Joint CTC decoding is enabled with weight = ' + str(dcw))
            self.decode_file += '_ctc{:}'.format(dcw)
            self.asr_model.ctc_weight = dcw
        self.asr_model.joint_ctc = dcw > 0
        if s.get('decode_lm_weight', 0) > 0:
            raise NotImplementedError('RNN-LM fusion (solver.py:326-333) is outside the LAS path (SURVEY.md §2.1)')
        self.asr_model.decode_lm_weight = 0
        self.asr_model.eval()
        self.verbose('Checking models performance on dev set ' + str(s['dev_set']) + '...')
        self.valid()

    valid = Trainer.valid                             # greedy attention decoding on the dev set (solver.py:389-441)
    _valid_body = Trainer._valid_body
    _check_status = Trainer._check_status

    def save_checkpoint(self, path):                  # the Tester never saves
        pass

    def exec(self):
        """Beam-search inference over the test set, solver.py:342-355."""
        self.verbose('Start decoding with beam search, beam size = ' + str(self.decode_beam_size))
        self.verbose('Number of utts to decode : {}'.format(len(self.test_set)))
        n = 0
        for x, y in self.test_set:
            if len(x.shape) == 4:
                x = x.squeeze(0)
            if len(y.shape) == 3:
                y = y.squeeze(0)
            for b in range(x.shape[0]):
                self.beam_decode(x[b:b + 1], y[b].tolist())
                n += 1
        self.verbose('Decode done, best results at {}.'.format(os.path.join(self.ckpdir, self.decode_file + '.txt')))
        self.verbose('Top {} results at {}.'.format(self.decode_beam_size,
                                                    os.path.join(self.ckpdir, self.decode_file + '_nbest.txt')))
        return n

    def write_hyp(self, hyps, y):
        """Record decoding results, solver.py:357-369."""
        gt = self.mapper.translate(y, return_string=True)
        with open(os.path.join(self.ckpdir, self.decode_file + '.txt'), 'a') as f:
            f.write(gt + '\t' + self.mapper.translate(hyps[0].outIndex, return_string=True) + '\n')
        with open(os.path.join(self.ckpdir, self.decode_file + '_nbest.txt'), 'a') as f:
            for hyp in hyps:
                f.write(gt + '\t' + self.mapper.translate(hyp.outIndex, return_string=True) + '\n')

    def beam_decode(self, x, y):
        """solver.py:372-390."""
        x = x.to(device=self.device, dtype=torch.float32)
        state_len = ops.infer_lengths(x).cpu().tolist()
        x = x[:, :max(state_len)].contiguous() if not self.asr_model.vgg else x
        max_decode_step = int(math.ceil(state_len[0] * self.decode_step_ratio))
        hyps = self.asr_model.beam_decode(x, max_decode_step, state_len, self.decode_beam_size)
        self._check_status(None, 'beam decoding')
        self.write_hyp(hyps, y)
        return hyps


class RNNLM_Trainer(Solver):
    def __init__(self, config, paras):
        raise NotImplementedError('RNN-LM training (src/solver.py:444-537) is outside the LAS hot path (SURVEY.md §2.1)')
