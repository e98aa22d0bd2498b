"""The one-kernel inference head against the same module run layer by layer in torch (the
layers mirror the reference's MLP, spiro/dtdg/models/decoder/torch/simpleDecoder.py:45-79),
and against the numpy oracle of that MLP.  fp32, 1e-5 relative to the largest output."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from spiral_amd.dtdg.models.decoder.torch.simpleDecoder import MLP

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _fused_head_on():
    """The one-kernel head is opt-in (MLP.FUSE_EVAL); these tests are about that path."""
    old = MLP.FUSE_EVAL
    MLP.FUSE_EVAL = True
    yield
    MLP.FUSE_EVAL = old


def _randomise(mlp, seed):
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in mlp.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.num_features, generator=gen))
                m.running_var.copy_(torch.rand(m.num_features, generator=gen) + 0.5)
                m.weight.copy_(torch.randn(m.num_features, generator=gen))
                m.bias.copy_(torch.randn(m.num_features, generator=gen))


@pytest.mark.parametrize("n,i,dims,out,act", [
    (5000, 128, [10, 20, 10, 5], 1, "linear"),     # the head of the reference's examples
    (4097, 32, [32], 3, "sigmoid"),
    (10_000, 256, [], 7, "tanh"),                  # a single Linear
    (6000, 64, [8, 8, 8, 8, 8, 8, 8], 2, "relu"),  # eight layers
    (70_000, 128, [17, 31], 32, "linear"),
    (5001, 96, [10, 20], 1, "linear"),
    (4100, 160, [32, 5], 4, "sigmoid"),
    (4500, 512, [10, 20], 1, "linear"),            # more than 256 inputs: two chunks per row
    (4200, 320, [16], 2, "tanh"),
    (5000, 384, [32], 3, "linear"),
    (4300, 448, [5], 1, "sigmoid"),
    (4400, 512, [32, 8], 1, "linear"),             # 512 x 32 first layer
    (70_001, 512, [10, 20, 10, 5], 1, "linear"),   # many tiles per wave, ragged last tile
    (4100, 288, [10], 1, "linear"),                # a width only the all-VALU kernel takes
])
def test_fused_head_equals_layerwise(dev, n, i, dims, out, act):
    torch.manual_seed(n % 97)
    mlp = MLP(i, dims, output_dim=out, activation=act).to(dev).eval()
    _randomise(mlp, seed=i)
    x = torch.randn(n, i, device=dev)
    ids = torch.arange(n, device=dev)
    with torch.no_grad():
        assert mlp._fused_eval_ok(x)
        got = mlp((x, ids))
        want = mlp.mlp(x)
    assert got.shape == want.shape
    assert ((got - want).abs().max() / want.abs().max()).item() < 1e-5
    # a strided view of a wider matrix (the recurrent output is such a view in places)
    wide = torch.randn(n, i + 64, device=dev)
    with torch.no_grad():
        got2, want2 = mlp((wide[:, 32:32 + i], ids)), mlp.mlp(wide[:, 32:32 + i])
    assert ((got2 - want2).abs().max() / want2.abs().max()).item() < 1e-5


def test_fused_head_follows_parameter_updates_and_modes(dev):
    n, i = 8192, 128
    mlp = MLP(i, [10, 20, 10, 5]).to(dev).eval()
    _randomise(mlp, seed=1)
    x = torch.randn(n, i, device=dev)
    ids = torch.arange(n, device=dev)
    with torch.no_grad():
        a = mlp((x, ids))
        mlp.layers[0].weight.mul_(1.5)                       # in-place update, as an optimizer does
        mlp.layers[1].running_mean.add_(0.25)
        b, want = mlp((x, ids)), mlp.mlp(x)
    assert not torch.allclose(a, b)
    assert ((b - want).abs().max() / want.abs().max()).item() < 1e-5
    # with autograd on, or in training mode, the layers run as torch modules
    assert not mlp._fused_eval_ok(x)
    mlp.train()
    with torch.no_grad():
        assert not mlp._fused_eval_ok(x)
    # widths beyond the kernel's registers fall back to the layers as well
    big = MLP(i, [64]).to(dev).eval()
    with torch.no_grad():
        assert not big._fused_eval_ok(x)
        assert big((x, ids)).shape == (n, 1)


def test_fused_head_against_the_oracle(dev):
    n, i = 4500, 128
    mlp = MLP(i, [10, 20, 10, 5]).to(dev).eval()
    _randomise(mlp, seed=3)
    x = torch.randn(n, i, device=dev)
    with torch.no_grad():
        got = mlp((x, torch.arange(n, device=dev))).cpu().numpy()
    cpu = lambda t: t.detach().cpu().numpy()
    layers = []
    for m in mlp.layers:
        if isinstance(m, torch.nn.Linear):
            layers.append({"w": cpu(m.weight), "b": cpu(m.bias)})
        elif isinstance(m, torch.nn.BatchNorm1d):
            layers.append({"gamma": cpu(m.weight), "beta": cpu(m.bias), "mean": cpu(m.running_mean),
                           "var": cpu(m.running_var)})
    want = orc.mlp_forward(cpu(x), layers)
    assert np.abs(got - want).max() / np.abs(want).max() < 1e-5
