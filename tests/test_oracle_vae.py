"""The VAE-decoder oracle (oracle/vae_oracle.py): structural pins only -- parity is unpinned (no diffusers, no
checkpoint offline; see the oracle's header)."""
import torch

from oracle import vae_oracle as V


def test_param_count_and_keys():
    cfg = V.make_config()
    # decoder of stabilityai/sd-vae-ft-mse: 49,490,179 parameters, + post_quant_conv 4*4 + 4
    assert V.num_params(cfg) == 49_490_179 + 20
    keys = list(V.param_shapes(cfg))
    assert keys[0] == "post_quant_conv.weight" and keys[2] == "decoder.conv_in.weight"
    assert "decoder.mid_block.attentions.0.to_out.0.weight" in keys
    assert "decoder.up_blocks.2.resnets.0.conv_shortcut.weight" in keys       # 512 -> 256
    assert "decoder.up_blocks.3.resnets.0.conv_shortcut.weight" in keys       # 256 -> 128
    assert "decoder.up_blocks.0.resnets.0.conv_shortcut.weight" not in keys
    assert "decoder.up_blocks.3.upsamplers.0.conv.weight" not in keys
    assert keys[-1] == "decoder.conv_out.bias"


def test_flops_match_survey():
    # ~1.26 TMAC = ~2.5 TFLOP per 512x512 image
    assert abs(V.flops_per_image(V.make_config(), 64) / 1e12 - 2.51) < 0.02


def test_decode_shapes_and_module_state_dict_agree():
    from bndm_amd.vae import AutoencoderKL
    cfg = V.make_config(block_out_channels=(128, 256), layers_per_block=1)
    sd = V.init_params(cfg, seed=2)
    y = V.vae_decode(sd, cfg, torch.randn(1, 4, 16, 16))
    assert y.shape == (1, 3, 32, 32) and torch.isfinite(y).all()
    m = AutoencoderKL(block_out_channels=(128, 256), layers_per_block=1)
    assert sorted(m.state_dict()) == sorted(sd) and all(m.state_dict()[k].shape == sd[k].shape for k in sd)
    m.load_state_dict(dict(sd, **{"encoder.conv_in.weight": torch.zeros(1)}))       # encoder keys are ignored
