import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def check_pinned(g, prefix, named, rtol=0.0, atol=2e-6, sum_rtol=1e-4, norm_rel=None, outlier_frac=0.0, outlier_atol=0.0):
    """Compare tensors with a fixture written by gen_goldens._pin: every stride-th element + the whole tensor's sum.
    norm_rel: compare the sampled vector in relative L2 norm instead of element by element (gradients whose entries span
    many orders of magnitude)."""
    for k, t in named:
        flat = t.detach().double().reshape(-1).cpu()
        stride = int(g[f"{prefix}_stride.{k}"])
        if norm_rel is not None:
            ref = g[f"{prefix}.{k}"].astype(np.float64)
            err = np.linalg.norm(flat[::stride].numpy() - ref)
            assert err <= norm_rel * np.linalg.norm(ref) + atol, f"{prefix}.{k}: {err} vs {np.linalg.norm(ref)}"
            continue
        got, ref = flat[::stride].numpy(), g[f"{prefix}.{k}"].astype(np.float64)
        if outlier_frac > 0.0:
            # Adam divides by sqrt(v): an entry whose gradient is at rounding-noise level moves by up to +-lr per step in a
            # direction that noise decides, so a small fraction of entries may sit up to `outlier_atol` away
            bad = np.abs(got - ref) > atol + rtol * np.abs(ref)
            assert bad.mean() <= outlier_frac and np.all(np.abs(got - ref)[bad] <= outlier_atol), f"{prefix}.{k}: {bad.sum()} outliers"
        else:
            np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol, err_msg=f"{prefix}.{k}")
        ref_sum, ref_l2 = float(g[f"{prefix}_sum.{k}"]), float(g[f"{prefix}_l2.{k}"])
        assert abs(float(flat.sum()) - ref_sum) <= sum_rtol * (ref_l2 * flat.numel() ** 0.5 + 1e-12) + atol * flat.numel() ** 0.5, f"{prefix}_sum.{k}"


def keep_first_step_gradients(opt, named):
    """Wrap opt.step so that the gradients present at its FIRST call are kept: -> dict filled in place."""
    first, plain = {}, opt.step

    def step(*a, **k):
        if not first:
            first.update({key: p.grad.detach().clone() for key, p in named()})
        return plain(*a, **k)

    opt.step = step
    return first
