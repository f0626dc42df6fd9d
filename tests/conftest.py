import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')
    config.addinivalue_line('markers', 'slow: takes tens of seconds on CPU')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return load


def fixture_xy(fx):
    """(x, y) float32 / int64 CPU tensors of a golden fixture: stored in it, or rebuilt from
    the recipe it carries (the bench-shape fixtures hold the recipe of their synthetic batch
    instead of 8 MB of frames; `x_abs_sum` guards the rebuild)."""
    import numpy as np
    import torch
    if 'x' in fx.files:
        return torch.from_numpy(fx['x']), torch.from_numpy(fx['y'])
    if 'recipe_config4' in fx.files:
        from ss_asr_amd.synthetic import config4_batch
        x, y, lens = config4_batch(batch_size=int(fx['recipe_batch_size']), feat_dim=int(fx['dims'][4]),
                                   seed=int(fx['recipe_seed']))
        assert lens == [int(v) for v in fx['lens']] and np.array_equal(y.numpy(), fx['y'])
        assert abs(float(x.double().abs().sum()) - float(fx['x_abs_sum'])) < 1e-6 * float(fx['x_abs_sum'])
        return x, y
    from ss_asr_amd.synthetic import config2_batches
    x, y, lens = config2_batches(int(fx['recipe_n_batches']), batch_size=int(fx['recipe_batch_size']),
                                 feat_dim=int(fx['dims'][4]), seed=int(fx['recipe_corpus_seed']))[int(fx['recipe_pick'])]
    assert lens == [int(v) for v in fx['lens']]
    assert np.array_equal(y.numpy(), fx['y'])
    assert abs(float(x.double().abs().sum()) - float(fx['x_abs_sum'])) < 1e-6 * float(fx['x_abs_sum'])
    return x, y


def fixture_att(fx, att):
    """The rows of an attention map [B, U, T'] that the fixture holds (all, or `att_rows`)."""
    return att[fx['att_rows']] if 'att_rows' in fx.files else att
