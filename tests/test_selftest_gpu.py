"""The start-up self-test of the tagged-granule hand-offs (csrc/selftest.hip, _lib._granule_selftest)."""
import ctypes
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_granule_selftest_ran_and_passes_at_length():
    _lib = importlib.import_module('end-to-end-asr-pytorch_amd._lib')
    L = _lib.lib()
    st = _lib.SELFTEST
    assert st is not None and st['ok'], st                      # the import-time run: 2 000 rounds per flavour
    assert st['finished'] == 2 * 2 * 64 * 64 and st['torn'] == 0 and st['timeouts'] == 0
    # ten times as many rounds: 128 workgroups x 64 lanes x 20 000 round trips per flavour, every polled granule checked
    ws = torch.empty(int(L.las_granule_selftest_bytes()), dtype=torch.uint8, device='cuda')
    res = (ctypes.c_uint * 3)()
    _lib.check(L.las_granule_selftest(_lib.I(20000), _lib.P(ws.data_ptr()), res, _lib.cur_stream()), 'las_granule_selftest')
    assert (int(res[0]), int(res[1]), int(res[2])) == (0, 2 * 2 * 64 * 64, 0)


def test_a_failed_selftest_selects_the_fallback_kernels(monkeypatch):
    """The consequence of a failure, without a device that fails: _granule_selftest with a library whose test reports a torn
    observation must set the two documented fallbacks (and only warn)."""
    _lib = importlib.import_module('end-to-end-asr-pytorch_amd._lib')
    real = _lib.lib()

    class Fake:
        def las_granule_selftest_bytes(self):
            return real.las_granule_selftest_bytes()

        def las_granule_selftest(self, iters, ws, res, stream):
            res[0], res[1], res[2] = 3, 2 * 2 * 64 * 64, 0
            return 0
    import os
    monkeypatch.delenv('LAS_LSTM_NO_GR', raising=False)
    monkeypatch.delenv('LAS_DEC_NO_PK', raising=False)
    saved = _lib.SELFTEST
    try:
        with pytest.warns(UserWarning, match='self-test failed'):
            _lib._granule_selftest(Fake())
        assert not _lib.SELFTEST['ok'] and os.environ.get('LAS_LSTM_NO_GR') == '1' and os.environ.get('LAS_DEC_NO_PK') == '1'
    finally:
        _lib.SELFTEST = saved
        os.environ.pop('LAS_LSTM_NO_GR', None)
        os.environ.pop('LAS_DEC_NO_PK', None)
