"""Token mapping and training-time metrics (host side); same contracts as reference src/postprocess.py:7-41,
121-147.  The per-step accuracy runs on the device (ops.token_acc); CER is host-side and only computed every
TRAIN_WER_STEP steps, as in the reference (solver.py:188-190)."""
import os
import pickle

# TIMIT 61 -> 39 phone folding used for scoring (standard Lee & Hon mapping; reference postprocess.py:164-168)
_FOLD = {"bcl": "h#", "dcl": "h#", "gcl": "h#", "pcl": "h#", "tcl": "h#", "kcl": "h#", "zh": "sh", "em": "m",
         "en": "n", "eng": "ng", "nx": "n", "hv": "hh", "el": "l", "ao": "aa", "ux": "uw", "ax": "ah", "ix": "ih",
         "axr": "er", "ax-h": "ah", "pau": "h#", "epi": "h#"}


def fold_phones(seq):
    return [_FOLD.get(p, p) for p in seq]


def trim_eos(seq):
    out = []
    for c in seq:
        out.append(int(c))
        if int(c) == 1:
            break
    return out


class Mapper:
    """index -> token, unit detection as reference postprocess.py:9-22."""

    def __init__(self, file_path=None, mapping=None):
        if mapping is None:
            with open(os.path.join(file_path, 'mapping.pkl'), 'rb') as fp:
                mapping = pickle.load(fp)
        self.mapping = mapping
        self.r_mapping = {v: k for k, v in mapping.items()}
        symbols = ''.join(str(k) for k in mapping.keys())
        if '▁' in symbols:
            self.unit = 'subword'
        elif '#' in symbols:
            self.unit = 'phone'
        elif len(mapping) < 50:
            self.unit = 'char'
        else:
            self.unit = 'word'

    def get_dim(self):
        return len(self.mapping)

    def translate(self, seq, return_string=False):
        toks = [self.r_mapping[c] for c in trim_eos(seq)]
        if not return_string:
            return toks
        strip = lambda s: s.replace('<sos>', '').replace('<eos>', '')
        if self.unit == 'subword':
            return strip(''.join(toks)).replace('▁', ' ').lstrip()
        if self.unit == 'word':
            return strip(' '.join(toks)).lstrip()
        if self.unit == 'phone':
            return strip(' '.join(fold_phones(toks)))
        return strip(''.join(toks))


def edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def cal_acc(pred_ids, label):
    """Mean over utterances of token accuracy up to the first 0 label (reference postprocess.py:121-133).
    pred_ids, label: integer arrays [B, L]."""
    accs = []
    for p, l in zip(pred_ids, label):
        correct = total = 0
        for pp, ll in zip(p, l):
            if ll == 0:
                break
            correct += int(pp == ll)
            total += 1
        accs.append(correct / max(total, 1))
    return sum(accs) / len(accs)


def cal_cer(pred_ids, label, mapper, get_sentence=False):
    """Word-level edit distance on space-split strings / reference length (reference postprocess.py:135-146)."""
    pred = [mapper.translate(p, return_string=True) for p in pred_ids]
    lab = [mapper.translate(l, return_string=True) for l in label]
    if get_sentence:
        return pred, lab
    eds = [float(edit_distance(p.split(' '), l.split(' '))) / len(l.split(' ')) for p, l in zip(pred, lab)]
    return sum(eds) / len(eds)


def draw_att(att_list, pred_ids):
    """3-channel attention images of the first head, one per utterance, cut at the hypothesis' <eos>
    (reference postprocess.py:149-155).  att_list[0]: (B, L, T') tensor / array; pred_ids: integer array [B, L]."""
    import numpy as np
    att0 = att_list[0]
    att0 = att0.detach().cpu().numpy() if hasattr(att0, 'detach') else np.asarray(att0)
    maps = []
    for att, hyp in zip(att0, pred_ids):
        n = len(trim_eos(hyp))
        maps.append(np.stack([att, att, att], axis=0)[:, :n, :])
    return maps
