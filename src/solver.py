"""`src.solver` as main.py imports it (reference main.py:38-44); re-exports the MI355X-native Trainer."""
import importlib as _il

_m = _il.import_module('end-to-end-asr-pytorch_amd.solver')
Solver, Trainer, Tester, RNNLM_Trainer = _m.Solver, _m.Trainer, _m.Tester, _m.RNNLM_Trainer
VAL_STEP, TRAIN_WER_STEP, GRAD_CLIP = _m.VAL_STEP, _m.TRAIN_WER_STEP, _m.GRAD_CLIP
