"""GPU parity of the stand-alone encoder classes (mctsq_encoder_forward through the C ABI) against the
reference's own ConvAttentionEnc / GRUAttentionEnc / TransformerEnc / GasEUAEncoder
(tests/golden/encoders.npz): seeded default initialisation bit-equal, outputs within 1e-5 of max|y|
(the north-star tolerance for fp32 results), plus a ragged / large batch against the fp64 oracle."""
import numpy as np
import pytest
import torch

from oracle import qnet_numpy
from tests.test_boundary_cpu import ORACLE_FN, case_state_dict, encoder_cases

pytestmark = pytest.mark.gpu


def build(tag, meta):
    from src.mctsq_config_dqn_per import ConvAttentionEnc, GasEUAEncoder, GRUAttentionEnc, TransformerEnc
    F = meta[tag]["shape"][2]
    torch.manual_seed(meta[tag]["seed"])
    if tag.startswith("conv"):
        return ConvAttentionEnc(F, 64)
    if tag.startswith("gru"):
        return GRUAttentionEnc(F, 64)
    if tag == "transformer":
        return TransformerEnc(F, 64, 4, 2)
    return GasEUAEncoder(F, 64, "mlp" if tag == "gas_mlp" else "gru")


@pytest.mark.parametrize("tag", ["conv", "conv6", "gru", "gru1", "transformer", "gas_mlp", "gas_gru"])
def test_encoder_class_matches_reference(tag):
    z, meta = encoder_cases()
    enc = build(tag, meta)
    ref_sd = case_state_dict(z, meta, tag)
    sd = enc.state_dict()
    assert sorted(sd) == sorted(ref_sd)
    for k, v in sd.items():                                  # same seed -> the reference's initial weights
        assert np.array_equal(v.numpy(), ref_sd[k]), k
    x, y = z[tag + "/x"], z[tag + "/y"]
    out = enc(torch.from_numpy(x))                            # CPU tensor in -> CPU tensor out
    assert out.device.type == "cpu" and out.shape == y.shape
    assert np.abs(out.numpy() - y).max() <= 1e-5 * np.abs(y).max()
    out_dev = enc(torch.from_numpy(x).cuda())
    assert out_dev.is_cuda and torch.equal(out_dev.cpu(), out)


@pytest.mark.parametrize("tag,B", [("conv", 1), ("gru", 37), ("gru1", 4096), ("conv6", 1000), ("transformer", 70)])
def test_encoder_ragged_batches_vs_fp64_oracle(tag, B):
    z, meta = encoder_cases()
    enc = build(tag, meta)
    _, S, F = meta[tag]["shape"]
    x = np.random.RandomState(B).random_sample((B, S, F)).astype(np.float32)
    sd = {k: np.asarray(v.numpy(), np.float64) for k, v in enc.state_dict().items()}
    ref = ORACLE_FN[tag](sd, x.astype(np.float64))
    got = enc(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()


def test_encoder_argument_errors():
    from src.mctsq_config_dqn_per import GasEUAEncoder, GRUAttentionEnc, TransformerEnc
    from mcts_q_amd import MctsqError
    with pytest.raises(ValueError):
        GasEUAEncoder(2, 64, "conv")
    with pytest.raises(MctsqError):
        TransformerEnc(6, 64, 8, 2)
    with pytest.raises(MctsqError):
        GRUAttentionEnc(6, 32)
    enc = GRUAttentionEnc(6, 64)
    with pytest.raises(ValueError):
        enc(torch.zeros(4, 10, 5))
    assert enc(torch.zeros(0, 10, 6).cuda()).shape == (0, 64)
    with pytest.raises(ValueError):
        GasEUAEncoder(2, 64, "mlp")(torch.zeros(4, 5, 2))
