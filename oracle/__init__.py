"""CPU fp32 oracle for the InstantIR denoising path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
package.  `instantir_amd/` never does: the shipped path is the HIP library and fails loudly when
it is missing.

What it is: a plain-PyTorch (CPU, float32) restatement of the reference's hot path, written from
the reference's text, every function citing the file:line it follows.

Parity pinning status
---------------------
* Pinned by goldens generated from the reference's own importable modules in the build container
  (`tests/golden/make_reference_goldens.py`): Resampler, AdaLayerNorm, AttnProcessor2_0,
  TA_IPAttnProcessor2_0, MultiIPAdapterImageProjection (module/ip_adapter/*), and
  LCMSingleStepScheduler (schedulers/lcm_single_step_scheduler.py; tables, c_skip/c_out, step,
  add_noise, set_timesteps).
* PARITY UNPINNED (third-party arithmetic that is not in /root/reference and not installed here:
  diffusers==0.28.1 UNet2DConditionModel / AutoencoderKL / DDPMScheduler / DDIMScheduler,
  peft==0.10.0 LoRA): the UNet, Aggregator trunk, VAE, DDPM/DDIM restatements follow the in-tree
  text copies (`module/min_sdxl.py`, `module/unet/unet_2d_ZeroSFT*.py`, `module/diffusers_vae/*`,
  `train_previewer_lora.py:194-254`) and the published DDPM/DDIM formulas; the reference holds no
  test or fixture for them.
"""
