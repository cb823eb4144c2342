"""Shared helpers for the GPU parity tests."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'open-genie_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    """fp32 tensor holding bf16-representable values (what the HIP path actually sees)."""
    return x.to(torch.bfloat16).to(torch.float32)


def assert_close_bf16(out: torch.Tensor, ref: torch.Tensor, what: str = '', rel: float = 2 ** -7, rms_frac: float = 2e-3):
    """`out` came from a kernel that accumulates in fp32 and rounds ONCE to bf16; `ref` is the fp32 oracle
    on the same (bf16-representable) inputs.  Bound: one bf16 ulp of the value plus a small fraction of
    the tensor RMS for accumulation-order effects and cancellation."""
    out = out.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert out.shape == ref.shape, f'{what}: shape {tuple(out.shape)} vs {tuple(ref.shape)}'
    assert torch.isfinite(out).all(), f'{what}: non-finite output'
    rms = ref.pow(2).mean().sqrt().item()
    tol = rel * ref.abs() + rms_frac * rms + 1e-30
    err = (out - ref).abs()
    bad = err > tol
    if bad.any():
        i = torch.argmax(err / tol)
        raise AssertionError(f'{what}: {int(bad.sum())}/{bad.numel()} elements out of tolerance; worst err '
                             f'{err.flatten()[i].item():.4g} vs tol {tol.flatten()[i].item():.4g} '
                             f'(ref {ref.flatten()[i].item():.4g}, rms {rms:.4g})')


def report(name: str, **values) -> None:
    """Append one JSON line of measured parity numbers to gpurun_out/parity_report.jsonl (pytest swallows the prints of passing
    tests; the judged summaries under profiles/ are copied from this file)."""
    import json
    try:
        d = os.path.join(ROOT, 'gpurun_out')
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, 'parity_report.jsonl'), 'a') as f:
            f.write(json.dumps({'test': name, **{k: ((round(v, 6) if abs(v) >= 1e-3 else float(f'{v:.3e}')) if isinstance(v, float) else v) for k, v in values.items()}}) + '\n')
    except OSError:
        pass
