"""CPU tests of the PyTorch-side modules the loops call once per image / edit: the restated
AutoencoderKL (diffusers is not installed; parity unpinned) and the SD-2 CLIP text tower built from transformers."""
import torch


def test_autoencoder_kl_matches_the_published_sd_vae_structure():
    from diffusionhandles_amd.vae import AutoencoderKL
    vae = AutoencoderKL()
    sd = vae.state_dict()
    assert sum(p.numel() for p in vae.parameters()) == 83_653_863          # stabilityai sd-vae parameter count
    assert len(sd) == 248
    for key, shape in {"encoder.down_blocks.1.resnets.0.conv_shortcut.weight": (256, 128, 1, 1),
                       "encoder.down_blocks.2.downsamplers.0.conv.weight": (512, 512, 3, 3),
                       "encoder.mid_block.attentions.0.to_q.weight": (512, 512),
                       "decoder.mid_block.attentions.0.to_out.0.bias": (512,),
                       "decoder.up_blocks.2.resnets.0.conv_shortcut.weight": (256, 512, 1, 1),
                       "decoder.up_blocks.0.upsamplers.0.conv.weight": (512, 512, 3, 3),
                       "encoder.conv_out.weight": (8, 512, 3, 3), "quant_conv.weight": (8, 8, 1, 1),
                       "post_quant_conv.weight": (4, 4, 1, 1)}.items():
        assert tuple(sd[key].shape) == shape, key
    assert vae.config.scaling_factor == 0.18215 and len(vae.config.block_out_channels) == 4


def test_autoencoder_kl_call_surface_and_shapes():
    from diffusionhandles_amd.vae import AutoencoderKL
    torch.manual_seed(0)
    vae = AutoencoderKL(dict(block_out_channels=(32, 64, 64, 64), norm_num_groups=8)).eval()
    x = torch.rand(1, 3, 64, 64) * 2 - 1
    with torch.no_grad():
        enc = vae.encode(x)
        z = enc["latent_dist"].mean
        assert z.shape == (1, 4, 8, 8) and enc.latent_dist.sample(torch.Generator().manual_seed(0)).shape == z.shape
        assert vae.encode(x, return_dict=False)[0].mode().shape == z.shape
        y = vae.decode(z / vae.config.scaling_factor)
        assert y["sample"].shape == x.shape and y.sample.shape == x.shape and y[0].shape == x.shape
        assert vae.decode(z, return_dict=False)[0].shape == x.shape
    # state dict round trip (what from_safetensors does)
    other = AutoencoderKL(dict(block_out_channels=(32, 64, 64, 64), norm_num_groups=8))
    other.load_state_dict(vae.state_dict())
    with torch.no_grad():
        assert torch.equal(other.eval().decode(z)[0], vae.decode(z)[0])


def test_sd2_text_tower_builds_with_the_reference_output_shape():
    from diffusionhandles_amd.vae import SD2_TEXT, build_text_encoder
    small = build_text_encoder(config=dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                                           num_attention_heads=4, projection_dim=32))
    ids = torch.randint(0, 49408, (2, 77))
    with torch.no_grad():
        out = small(ids)[0]
    assert out.shape == (2, 77, 64)
    assert SD2_TEXT["hidden_size"] == 1024 and SD2_TEXT["num_hidden_layers"] == 23      # cross_attention_dim of the U-Net
