"""CPU tests of the PyTorch-side modules the loops call once per image / edit: the restated
AutoencoderKL (diffusers is not installed; parity unpinned) and the SD-2 CLIP text tower built from transformers."""
import pytest
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


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1e-2), (torch.bfloat16, 4e-2)])
def test_native_vae_decoder_matches_the_torch_restatement(dtype, tol):
    """csrc/vae_engine.cpp (MFMA implicit-GEMM convolutions, engine GroupNorm, the 512-dim attention head as two GEMMs
    around a row softmax) against diffusionhandles_amd.vae.AutoencoderKL.decode in fp32 on the same seeded weights, at the
    full SD VAE size (64x64 latent -> 512x512 image) and at a 32x32 latent.  Relative L2 of the image; fp16 measured 2e-3."""
    from diffusionhandles_amd.vae import AutoencoderKL, HipVAEDecoder
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    vae = AutoencoderKL().to(dev).eval()
    with torch.no_grad():
        for p in vae.parameters():
            p.copy_(p.to(dtype).float())
    for lat, B in ((64, 2), (32, 1)):
        dec = HipVAEDecoder(latent_size=lat, dtype=dtype).load_state_dict(vae.state_dict())
        z = torch.randn(B, 4, lat, lat, generator=torch.Generator().manual_seed(lat)).to(dev) * 3.0
        with torch.no_grad():
            ref = vae.decode(z)["sample"]
        got = dec.decode(z)["sample"]
        assert got.shape == ref.shape == (B, 3, 8 * lat, 8 * lat)
        err = ((got - ref).norm() / ref.norm()).item()
        print(f"native VAE decode {dtype} latent {lat}: rel L2 {err:.3e}, ref rms {ref.pow(2).mean().sqrt().item():.3f}")
        assert err < tol
        assert torch.equal(dec.decode(z, return_dict=False)[0], got)          # deterministic, tuple form


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1e-2), (torch.bfloat16, 4e-2)])
def test_native_vae_encoder_matches_the_torch_restatement(dtype, tol):
    """The encoder half on the same kernels (stride-2 convolutions padded bottom / right only, as diffusers' Downsample2D with
    padding 0) against diffusionhandles_amd.vae.AutoencoderKL.encode in fp32 on the same seeded weights: the mean of the latent
    distribution at 512x512 and 256x256 (relative L2), and the NativeDecodeVAE wrapper routes encode and decode natively."""
    from diffusionhandles_amd.vae import AutoencoderKL, HipVAEEncoder, NativeDecodeVAE
    dev = torch.device("cuda:0")
    torch.manual_seed(4)
    vae = AutoencoderKL().to(dev).eval()
    with torch.no_grad():
        for p in vae.parameters():
            p.copy_(p.to(dtype).float())
    for lat, B in ((64, 2), (32, 1)):
        enc = HipVAEEncoder(latent_size=lat, dtype=dtype).load_state_dict(vae.state_dict())
        img = (torch.rand(B, 3, 8 * lat, 8 * lat, generator=torch.Generator().manual_seed(lat)).to(dev) * 2 - 1)
        with torch.no_grad():
            ref = vae.encode(img)["latent_dist"]
        got = enc.encode(img)["latent_dist"]
        assert got.mean.shape == ref.mean.shape == (B, 4, lat, lat)
        err = ((got.mean - ref.mean).norm() / ref.mean.norm()).item()
        print(f"native VAE encode {dtype} latent {lat}: rel L2 {err:.3e}, ref rms {ref.mean.pow(2).mean().sqrt().item():.3f}")
        assert err < tol
        assert torch.equal(enc.encode(img, return_dict=False)[0].mean, got.mean)          # deterministic, tuple form
    wrap = NativeDecodeVAE(vae, latent_size=32, dtype=dtype).to(dev)
    assert wrap._enc is None                                     # the encoder engine is built by the first encode()
    img = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(9)).to(dev) * 2 - 1
    z = wrap.encode(img)["latent_dist"].mean
    assert wrap._enc is not None
    # another resolution than the engines were built for: both directions fall back to the module
    z16 = torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(10)).to(dev)
    with torch.no_grad():
        # (a relative-L2 bound, not equality: MIOpen may pick another convolution algorithm on the second call of the module)
        d_a, d_b = wrap.decode(z16)["sample"], vae.decode(z16)["sample"]
        assert ((d_a - d_b).norm() / d_b.norm()).item() < 1e-3
        e_a = wrap.encode(img[..., :128, :128])["latent_dist"].mean
        e_b = vae.encode(img[..., :128, :128])["latent_dist"].mean
        assert ((e_a - e_b).norm() / e_b.norm()).item() < 1e-3
    with torch.no_grad():
        zr = vae.encode(img)["latent_dist"].mean
    assert ((z - zr).norm() / zr.norm()).item() < tol
    rec = wrap.decode(z)["sample"]
    with torch.no_grad():
        recr = vae.decode(zr)["sample"]
    assert ((rec - recr).norm() / recr.norm()).item() < 2 * tol


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1e-2), (torch.bfloat16, 5e-2)])
def test_native_text_encoder_matches_transformers_clip(dtype, tol):
    """csrc/text_engine.cpp (LayerNorm, MFMA GEMMs, causal flash attention, erf-GELU) against transformers' CLIPTextModel in
    fp32 with the SD-2 text configuration (23 layers, width 1024, 16 heads) on the same seeded weights: last_hidden_state of
    two 77-token prompts, relative L2; causality: changing a later token leaves the earlier positions untouched."""
    from diffusionhandles_amd.vae import HipTextEncoder, build_text_encoder
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    ref_model = build_text_encoder().to(dev).eval()
    with torch.no_grad():
        for p in ref_model.parameters():
            p.copy_(p.to(dtype).float())
    enc = HipTextEncoder(dtype=dtype, max_batch=2).load_state_dict(ref_model.state_dict())
    ids = torch.randint(0, 49408, (2, 77), generator=torch.Generator().manual_seed(5)).to(dev)
    with torch.no_grad():
        ref = ref_model(ids)[0]
    got = enc(ids)[0]
    assert got.shape == ref.shape == (2, 77, 1024)
    err = ((got - ref).norm() / ref.norm()).item()
    print(f"native text encoder {dtype}: rel L2 {err:.3e}, ref rms {ref.pow(2).mean().sqrt().item():.3f}")
    assert err < tol
    ids2 = ids.clone(); ids2[:, 40:] = (ids2[:, 40:] + 7) % 49408
    got2 = enc(ids2)[0]
    assert torch.equal(got2[:, :40], got[:, :40]) and not torch.equal(got2[:, 40:], got[:, 40:])
    short = enc(ids[:1, :20])[0]                                # fewer tokens than the maximum
    with torch.no_grad():
        ref_s = ref_model(ids[:1, :20])[0]
    assert ((short - ref_s).norm() / ref_s.norm()).item() < tol


@pytest.mark.gpu
def test_diffuser_with_native_text_tower_and_vae():
    """GuidedStableDiffuser(text_encoder="sd2-native", vae="sd-native"): the prompt embedding comes from the native CLIP tower
    (compared with the transformers module built from the same seed) and images go through the native VAE both ways."""
    from diffusionhandles_amd import conf as C
    from diffusionhandles_amd.guided_stable_diffuser import GuidedStableDiffuser
    from diffusionhandles_amd.unet import HipUNet
    from diffusionhandles_amd.vae import HipTextEncoder, NativeDecodeVAE, build_text_encoder
    dev = torch.device("cuda:0")
    hip = HipUNet(dtype=torch.float16, max_batch=2)
    hip.init_synthetic(0)
    conf = C.load_default().guided_diffuser
    gd = GuidedStableDiffuser(conf, unet=hip, text_encoder="sd2-native", vae="sd-native", synthetic_seed=3).to(dev)
    assert isinstance(gd.text_encoder, HipTextEncoder) and isinstance(gd.vae, NativeDecodeVAE)
    # the diffuser seeds its random-weight text tower with 1000 + synthetic_seed (the same weights in every process: two runs of
    # a driver must agree byte for byte) without touching the caller's RNG stream
    torch.manual_seed(77)
    probe, probe_dev = torch.rand(1).item(), torch.rand(4, device=dev).cpu()
    torch.manual_seed(77)
    GuidedStableDiffuser(conf, unet=hip, text_encoder="sd2", vae="sd", synthetic_seed=3).to(dev)
    assert torch.rand(1).item() == probe, "building the random-weight modules consumed the caller's RNG"
    assert torch.equal(torch.rand(4, device=dev).cpu(), probe_dev), "building the random-weight modules re-seeded the DEVICE generator"
    torch.manual_seed(1000 + 3)
    ref = build_text_encoder().to(dev).eval()
    emb = gd._encode(["a sphere on a plane", ""])
    ids = gd.tokenizer(["a sphere on a plane", ""], padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
    with torch.no_grad():
        want = ref(ids.to(dev))[0]
    assert emb.shape == want.shape == (2, 77, 1024)
    assert ((emb - want).norm() / want.norm()).item() < 1e-2
    img = torch.rand(1, 3, 512, 512, device=dev) * 2 - 1
    z = gd.vae.encode(img)["latent_dist"].mean
    assert z.shape == (1, 4, 64, 64) and torch.isfinite(z).all()
    rec = gd.vae.decode(z)["sample"]
    assert rec.shape == (1, 3, 512, 512) and torch.isfinite(rec).all()


@pytest.mark.gpu
def test_native_text_encoder_long_causal_sequence():
    """Causal attention over more than 448 tokens (8+ key tiles, where the non-causal forward splits the keys over wave groups):
    576 tokens, two layers, against transformers' CLIPTextModel; finite and causal."""
    from diffusionhandles_amd.vae import HipTextEncoder, build_text_encoder
    dev = torch.device("cuda:0")
    cfg = dict(num_hidden_layers=2, max_position_embeddings=600)
    torch.manual_seed(13)
    ref_model = build_text_encoder(config=cfg).to(dev).eval()
    with torch.no_grad():
        for p in ref_model.parameters():
            p.copy_(p.half().float())
    enc = HipTextEncoder(cfg, max_batch=1).load_state_dict(ref_model.state_dict())
    ids = torch.randint(0, 49408, (1, 576), generator=torch.Generator().manual_seed(6)).to(dev)
    with torch.no_grad():
        ref = ref_model(ids)[0]
    got = enc(ids)[0]
    assert torch.isfinite(got).all()
    err = ((got - ref).norm() / ref.norm()).item()
    print(f"native text encoder, 576 tokens: rel L2 {err:.3e}")
    assert err < 1e-2
    ids2 = ids.clone(); ids2[:, 500:] = (ids2[:, 500:] + 3) % 49408
    assert torch.equal(enc(ids2)[0][:, :500], got[:, :500])


@pytest.mark.gpu
def test_native_aux_error_paths():
    """The native text tower / VAE refuse what they were not built for instead of computing something else."""
    from diffusionhandles_amd.vae import AutoencoderKL, HipTextEncoder, HipVAEDecoder, HipVAEEncoder, build_text_encoder
    dev = torch.device("cuda:0")
    small = dict(num_hidden_layers=1)
    m = build_text_encoder(config=small).to(dev).eval()
    enc = HipTextEncoder(small, max_batch=1).load_state_dict(m.state_dict())
    ids = torch.zeros(2, 77, dtype=torch.int64, device=dev)
    with pytest.raises(RuntimeError):
        enc(ids)                                                  # batch 2 > max_batch 1
    with pytest.raises(NotImplementedError):
        enc(ids[:1], attention_mask=torch.ones(1, 77, device=dev))
    with pytest.raises(NotImplementedError):
        HipTextEncoder(dict(small, hidden_act="quick_gelu"))
    vae = AutoencoderKL().to(dev).eval()
    ve = HipVAEEncoder(latent_size=16).load_state_dict(vae.state_dict())
    with pytest.raises(ValueError):
        ve.encode(torch.zeros(1, 3, 256, 256, device=dev))        # built for 128x128 images
    with pytest.raises(NotImplementedError):
        ve.decode(torch.zeros(1, 4, 16, 16, device=dev))
    vd = HipVAEDecoder(latent_size=16).load_state_dict(vae.state_dict())
    with pytest.raises(ValueError):
        vd.decode(torch.zeros(1, 4, 32, 32, device=dev))
    import ctypes
    from diffusionhandles_amd import _lib
    out = torch.zeros(1, 16, 16, 8, device=dev)
    img = torch.zeros(1, 128, 128, 3, device=dev)
    rc = _lib.lib().dh_vae_encoder_encode(vd._h, _lib.ptr(img), 1, _lib.ptr(out), _lib.stream_ptr())
    assert rc != 0 and b"decoder" in _lib.lib().dh_last_error()   # a decoder handle is not an encoder
