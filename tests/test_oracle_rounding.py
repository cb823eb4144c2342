"""The oracle's bf16-at-stores mode (oracle.set_rounding): CPU checks that it is OFF by default (the pinned fp32 arithmetic is
untouched), that it rounds exactly at operator outputs / stored gradients, and that it stays a small perturbation of the fp32
result.  The GPU parity tests use it as the tight bound for model-level comparisons (tests/test_gpu_tokenizer.py)."""
import torch

from util import bf16_round  # noqa: F401  (path set-up)
from oracle import genie_oracle as O


def _is_bf16(t):
    return torch.equal(t, t.to(torch.bfloat16).to(t.dtype))


def _sd_block(c_in, c_out, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    return {'main.0.weight': 1 + 0.1 * r(c_in), 'main.0.bias': 0.1 * r(c_in), 'main.2.weight': bf16_round(0.05 * r(c_out, c_in, 3, 3, 3)), 'main.2.bias': 0.1 * r(c_out),
            'main.4.weight': 1 + 0.1 * r(c_out), 'main.4.bias': 0.1 * r(c_out), 'main.6.weight': bf16_round(0.05 * r(c_out, c_out, 3, 3, 3)), 'main.6.bias': 0.1 * r(c_out),
            'res.1.weight': bf16_round(0.2 * r(c_out, c_in, 1, 1, 1)), 'res.1.bias': 0.1 * r(c_out)}


def test_rounding_mode_is_off_by_default_and_restored():
    assert O._ROUNDING is None
    with O.rounding('bf16_at_stores'):
        assert O._ROUNDING == 'bf16_at_stores'
        with O.rounding(None):
            assert O._ROUNDING is None
        assert O._ROUNDING == 'bf16_at_stores'
    assert O._ROUNDING is None
    try:
        O.set_rounding('fp8')
    except ValueError:
        pass
    else:
        raise AssertionError('unknown mode accepted')


def test_residual_block_rounds_at_stores_only():
    torch.manual_seed(1)
    sd = _sd_block(8, 16)
    x = bf16_round(torch.randn(2, 8, 3, 6, 6))
    ref = O.video_residual_block(x, sd, '', 8, 16)
    assert not _is_bf16(ref)                                   # the default arithmetic is fp32 end to end
    with O.rounding('bf16_at_stores'):
        xr = x.clone().requires_grad_(True)
        sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        out = O.video_residual_block(xr, sdr, '', 8, 16)
        assert _is_bf16(out.detach())
        dy = bf16_round(torch.randn_like(out))
        out.backward(dy)
    # a small perturbation of the fp32 result (three stored intermediates, relative 2^-9 each)
    rel = ((out.detach() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    assert 0 < rel < 1e-2, rel
    # fp32 gradients for comparison
    x32 = x.clone().requires_grad_(True)
    sd32 = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    O.video_residual_block(x32, sd32, '', 8, 16).backward(dy)
    # parameter gradients are NOT rounded (the kernels accumulate them in fp32) but stay close to the fp32 ones
    for k in sd:
        g, g32 = sdr[k].grad, sd32[k].grad
        assert ((g - g32).pow(2).mean().sqrt() / g32.pow(2).mean().sqrt()).item() < 2e-2, k
    assert not _is_bf16(sdr['main.2.weight'].grad)
    # the input gradient is the SUM of a stored main-branch gradient and the shortcut's fp32 backward-data result: rounded by whoever stores x
    assert ((xr.grad - x32.grad).pow(2).mean().sqrt() / x32.grad.pow(2).mean().sqrt()).item() < 2e-2


def test_group_norm_silu_pair_is_one_store():
    """'group_norm' followed by 'silu' in a blueprint is one pass on the HIP path: one rounding, not two."""
    torch.manual_seed(2)
    desc = (('group_norm', {'num_groups': 2, 'num_channels': 8}), ('silu', {}))
    sd = {'enc_layers.0.weight': 1 + 0.1 * torch.randn(8), 'enc_layers.0.bias': 0.1 * torch.randn(8)}
    x = bf16_round(torch.randn(2, 8, 2, 4, 4))
    exact = O.silu(O.group_norm(x, 2, sd['enc_layers.0.weight'], sd['enc_layers.0.bias']))
    with O.rounding('bf16_at_stores'):
        got = O.tokenizer_encode(x, sd, desc)
    assert torch.equal(got, exact.to(torch.bfloat16).float())
    assert torch.equal(O.tokenizer_encode(x, sd, desc), exact)          # and untouched when the mode is off


def test_space_time_block_rounding_mode_runs_and_stays_close():
    torch.manual_seed(3)
    c, nh, dh = 32, 2, 16
    sd = {}
    for a, kind in (('space_attn.', '2d'), ('temp_attn.', '1d')):
        sd[a + 'norm.weight'] = 1 + 0.1 * torch.randn(c)
        sd[a + 'norm.bias'] = 0.1 * torch.randn(c)
        sd[a + 'embed.freq'] = O.rotary_freq(c, kind)
    sd['ffn.1.net.0.weight'] = 1 + 0.1 * torch.randn(c)
    sd['ffn.1.net.0.bias'] = 0.1 * torch.randn(c)
    sd['ffn.1.net.1.0.weight'] = bf16_round(0.05 * torch.randn(c, c, 3, 3, 3))
    x = bf16_round(torch.randn(2, 3, 4, 4, c))
    ref = O.space_time_block(x, sd, '', nh, dh)
    with O.rounding('bf16_at_stores'):
        xr = x.clone().requires_grad_(True)
        out = O.space_time_block(xr, sd, '', nh, dh)
        out.sum().backward()
    assert _is_bf16(out.detach())
    assert ((out.detach() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item() < 2e-2
