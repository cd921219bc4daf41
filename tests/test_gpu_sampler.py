"""`-m gpu`: device neighbour sampling of the init embeddings (csrc/rr_sample.hip) against torch.multinomial, the
reference's sampler (rrnco/models/env_embeddings/atsp.py:55-67)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_gumbel_topk_sampler_has_the_multinomial_law():
    from rrnco_amd.models.encoder import ATSPInitEmbedding
    torch.manual_seed(0)
    N, K, R = 20, 5, 20000
    D1 = torch.rand(1, N, N)
    D1[0].fill_diagonal_(0.0)
    D = D1.repeat(R, 1, 1).cuda()
    idx = ATSPInitEmbedding.sample_indices(D, K)                       # [R, N, K] on the device kernel
    assert idx.shape == (R, N, K) and idx.dtype == torch.int64
    srt = idx.sort(-1).values
    assert bool((srt[..., 1:] != srt[..., :-1]).all())                 # without replacement
    assert bool((idx != torch.arange(N, device="cuda")[None, :, None]).float().mean() > 0.999)    # the diagonal has weight ~1e-6
    # reference law: torch.multinomial on the same probabilities
    pd = D1[0].clone(); pd.fill_diagonal_(1e6)
    inv = 1 / (pd + 1e-6)
    prob = inv / inv.sum(-1, keepdim=True)
    ref = torch.stack([torch.multinomial(prob, K, replacement=False) for _ in range(R)])          # [R, N, K]
    onehot = lambda t: torch.zeros(R, N, N).scatter_(2, t.cpu(), 1.0).mean(0)                     # noqa: E731  inclusion frequency
    f_hip, f_ref = onehot(idx), onehot(ref)
    # first draw: exactly the categorical law
    first = torch.zeros(N, N).index_put_((torch.arange(N).repeat_interleave(R), idx[:, :, 0].t().reshape(-1).cpu()),
                                         torch.ones(N * R), accumulate=True) / R
    assert float((first - prob).abs().max()) < 5 * (0.25 / R) ** 0.5 + 1e-3
    # inclusion frequencies of the K-subset agree with multinomial's within sampling noise (two empirical estimates)
    assert float((f_hip - f_ref).abs().max()) < 6 * (0.5 / R) ** 0.5


def test_sampler_is_reproducible_under_the_torch_seed_and_differs_between_draws():
    from rrnco_amd.models.encoder import ATSPInitEmbedding
    D = torch.rand(3, 50, 50, generator=torch.Generator().manual_seed(1)).cuda()
    torch.manual_seed(5)
    a, b = ATSPInitEmbedding.sample_indices(D, 25), ATSPInitEmbedding.sample_indices(D, 25)
    torch.manual_seed(5)
    c = ATSPInitEmbedding.sample_indices(D, 25)
    assert torch.equal(a, c) and not torch.equal(a, b)
