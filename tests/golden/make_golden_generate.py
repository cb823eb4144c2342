"""Golden fixture for MaskGIT token ids: the REAL reference's ``DynamicsModel.generate`` (reference genie/dynamics.py:101-165) on
the reference's own test configuration (test/test_dynamics.py:9-35: 4 x space-time_attn with 4 heads x 16, 16 tokens, 4 actions,
(2, 10, 16, 16) context; ``n_embd`` dropped because the reference's constructor rejects it), with the one
non-reproducible call -- ``torch.multinomial`` (dynamics.py:141) -- replaced for the duration of the call by the inverse-CDF draw
from pre-drawn uniforms (oracle/genie_oracle.py::sample_from_uniform).  Everything else (schedule, softmax, confidence, -inf
masking, top-k, scatter, the never-fed-back context) is the reference's code, run as is.

    python tests/golden/make_golden_generate.py          (build container only: needs /root/reference)
"""
import copy
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import genie_oracle as O  # noqa: E402
from oracle.ref_import import import_reference  # noqa: E402

REF_TEST_DESC = (('space-time_attn', {'n_rep': 4, 'n_head': 4, 'd_head': 16, 'transpose': False}),)


def reference_generate(model, tokens, act, uniforms, steps, **kw):
    """model.generate(...) of the real reference with torch.multinomial swapped for the injected-noise draw."""
    calls = {'n': 0}
    real = torch.multinomial

    def injected(prob, num_samples=1, **_):
        assert num_samples == 1
        u = uniforms[calls['n']]
        calls['n'] += 1
        return O.sample_from_uniform(prob, u)[:, None]

    torch.multinomial = injected
    try:
        return model.generate(tokens, act, steps=steps, **kw), calls['n']
    finally:
        torch.multinomial = real


def main():
    ref = import_reference()
    out = {}
    for name, (seed, steps, which, temp) in {'linear5': (200, 5, 'linear', 1.), 'cosine8_t07': (201, 8, 'cosine', 0.7)}.items():
        torch.manual_seed(seed)
        m = ref.DynamicsModel(copy.deepcopy(REF_TEST_DESC), tok_vocab=16, act_vocab=4, embed_dim=64).eval()
        with torch.no_grad():
            for n_, p in m.named_parameters():
                if 'freq' in n_:
                    continue
                if p.dim() >= 2:
                    p.copy_(p.to(torch.bfloat16).float())                    # bf16-representable weights (what the HIP path multiplies)
        tok, act = torch.randint(0, 16, (2, 10, 16, 16)), torch.randint(0, 4, (2, 10))
        u = torch.rand(steps, 2 * 256)
        gen, ncalls = reference_generate(m, tok, act, u, steps, which=which, temp=temp)
        with torch.no_grad():
            _, last = m(torch.cat([tok, torch.zeros(2, 1, 16, 16, dtype=tok.dtype)], 1), torch.cat([act, torch.zeros(2, 1, dtype=act.dtype)], 1))
        out[name] = dict(desc=REF_TEST_DESC, tok_vocab=16, act_vocab=4, embed_dim=64, tokens=tok, act=act, uniforms=u, steps=steps, which=which,
                         temp=temp, sd={k: v.detach().clone() for k, v in m.state_dict().items()}, gen=gen.detach().clone(),
                         last_logits=last.detach().clone(), multinomial_calls=ncalls)
        # torch.topk's order among EQUAL confidences is unspecified: a fixture whose selection boundary is tied would pin an
        # implementation detail of the CPU sort, not the algorithm -- refuse to write one
        tr = []
        O.dynamics_generate(tok, act, out[name]['sd'], REF_TEST_DESC, u, steps=steps, which=which, temp=temp, trace=tr)
        for t_ in tr:
            c = t_['conf'].masked_fill(~t_['mask_before'], -1.)
            if t_['k'] < c.shape[1]:
                top = c.topk(t_['k'] + 1, -1).values
                assert (top[:, -2] > top[:, -1]).all(), 'tied confidences at the top-k boundary'
        print(name, 'calls', ncalls, 'distinct ids', gen[:, -1].unique().numel(), 'logit std', float(last.std()))
    path = os.path.join(HERE, 'dynamics_generate.pt')
    torch.save(out, path)
    print(f'dynamics_generate.pt: {os.path.getsize(path) / 1024:.0f} KiB')


if __name__ == '__main__':
    main()
