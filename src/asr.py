"""`src.asr` of the reference; re-exports the MI355X-native Seq2Seq (same constructor / forward signature)."""
import importlib as _il

Seq2Seq = _il.import_module('end-to-end-asr-pytorch_amd.asr').Seq2Seq
