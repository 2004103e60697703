"""Structural pins for oracle/unet_oracle.py (diffusers is absent: parity unpinned beyond these)."""
import torch

from oracle import unet_oracle as U


def _count(cfg):
    return sum(torch.Size(s).numel() for s in U.param_shapes(cfg).values())


def test_parameter_counts_match_published_sizes():
    assert _count(U.make_config(64, 3, 3)) == 113_673_219
    assert _count(U.make_config(64, 3, 6)) == 113_676_678
    assert _count(U.make_config(128, 3, 6)) == 116_320_390
    assert _count(U.make_config(64, 4, 8, latent=True)) == 113_680_136


def test_flops_per_image():
    assert abs(U.flops_per_image(U.make_config(64, 3, 3), 64) / 1e9 - 31.03) < 0.01
    assert abs(U.flops_per_image(U.make_config(128, 3, 6), 128) / 1e9 - 103.39) < 0.01


def test_state_dict_keys():
    keys = list(U.param_shapes(U.make_config(64, 3, 6)))
    for k in ("conv_in.weight", "time_embedding.linear_1.weight", "time_embedding.linear_2.bias",
              "down_blocks.0.resnets.1.time_emb_proj.weight", "down_blocks.2.resnets.0.conv_shortcut.weight",
              "down_blocks.4.attentions.1.to_out.0.bias", "down_blocks.4.downsamplers.0.conv.weight",
              "mid_block.attentions.0.group_norm.weight", "up_blocks.1.attentions.2.to_q.weight",
              "up_blocks.4.upsamplers.0.conv.bias", "up_blocks.5.resnets.2.conv_shortcut.weight",
              "conv_norm_out.bias", "conv_out.weight"):
        assert k in keys, k
    assert "down_blocks.5.downsamplers.0.conv.weight" not in keys
    assert "up_blocks.5.upsamplers.0.conv.weight" not in keys
    assert "down_blocks.0.resnets.0.conv_shortcut.weight" not in keys


def test_forward_shapes_and_timestep_forms():
    cfg = U.make_config(64, 3, 6)
    sd = U.init_params(cfg, seed=1, perturb_norm=0.1)
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(0))
    y1 = U.forward(sd, cfg, x, torch.tensor([0.5, 0.5]))
    y2 = U.forward(sd, cfg, x, torch.tensor(0.5))
    y3 = U.forward(sd, cfg, x, 0.5)
    assert y1.shape == (2, 6, 64, 64)
    assert torch.equal(y1, y2) and torch.equal(y1, y3)
    # samples are independent (GroupNorm / attention are per-sample): SURVEY section 8e
    y_single = U.forward(sd, cfg, x[1:2], 0.5)
    assert torch.allclose(y_single, y1[1:2], atol=1e-5)
