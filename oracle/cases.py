"""ORACLE support — the small network configurations used by the golden fixtures and parity tests.

They keep every structural feature of the shipped configs (GroupNorm32, two Swin blocks per stage with
a shifted second block, 8x8 windows with 32-channel heads, FiLM ResBlocks, skip concats whose group
boundaries straddle the two sources, VQ-f4 autoencoder with a mid attention block) at a size the CPU
oracle evaluates in well under a second.
"""
TINY_UNET = dict(image_size=16, in_channels=3, model_channels=32, out_channels=3, attention_resolutions=[16, 8], dropout=0,
                 channel_mult=[1, 2], num_res_blocks=[1, 1], conv_resample=True, dims=2, use_fp16=False, num_head_channels=32,
                 use_scale_shift_norm=True, resblock_updown=False, swin_depth=2, swin_embed_dim=64, window_size=8, mlp_ratio=2,
                 cond_lq=True, lq_size=16)
# feature-extractor + mask variant (inpaint / faceir style conditioning, models/unet.py:689-702)
TINY_UNET_FE = dict(TINY_UNET, lq_size=64, cond_mask=True)
TINY_AE = dict(embed_dim=3, n_embed=512,
               ddconfig=dict(double_z=False, z_channels=3, resolution=64, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 4],
                             num_res_blocks=1, attn_resolutions=[], dropout=0.0, padding_mode="zeros"))
TINY_DIFFUSION = dict(sf=4, schedule_name="exponential", schedule_kwargs=dict(power=0.3), etas_end=0.99, steps=4, min_noise_level=0.2,
                      kappa=2.0, weighted_mse=False, predict_type="xstart", timestep_respacing=None, scale_factor=1.0,
                      normalize_input=True, latent_flag=True)
# sf=1 variant used together with TINY_UNET_FE (inpainting-like: lq at image resolution, latent at /4)
TINY_DIFFUSION_SF1 = dict(TINY_DIFFUSION, sf=1)
# faceir-style: 8-channel latent, conditioning through the feature extractor without a mask
TINY_UNET_FE8 = dict(TINY_UNET, in_channels=8, out_channels=8, lq_size=64)
TINY_AE8 = dict(embed_dim=8, n_embed=256,
                ddconfig=dict(double_z=False, z_channels=8, resolution=64, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 4],
                              num_res_blocks=[1, 2, 1], attn_resolutions=[], dropout=0.0, padding_mode="zeros"))
