import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


# LAS_POISON=1: every torch.empty / empty_like on the GPU comes back filled with NaN (floats) or 0x7f bytes (integers), so a
# kernel that relies on "fresh memory is zero" fails loudly instead of depending on what the caching allocator hands out.
if os.environ.get('LAS_POISON'):
    import torch
    _empty, _empty_like = torch.empty, torch.empty_like

    def _poison(t):
        if t.is_cuda and t.numel():
            if t.is_floating_point():
                t.fill_(float('nan'))
            else:
                t.view(torch.uint8).fill_(0x7f) if t.is_contiguous() else None
        return t

    torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
    torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))
