import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantir_amd import lib, weights as W
from instantir_amd.config import UNetConfig
from instantir_amd.pipeline import InstantIRPipeline
from instantir_amd.schedulers import DDIMScheduler, LCMSingleStepScheduler
lib.load()
cfg = UNetConfig.tiny()
sd = W.synth_state_dict(W.unet_specs(cfg), 11); sda = W.synth_state_dict(W.aggregator_specs(cfg), 12); lora = W.synth_state_dict(W.lora_specs(cfg), 13)
g = torch.Generator().manual_seed(0)
B, H = 2, 16
lq = torch.randn(B, 4, H, H, generator=g) * 0.8
pe = torch.randn(B, cfg.text_len, cfg.cross_attention_dim, generator=g)
pooled = torch.randn(B, cfg.pooled_dim, generator=g)
img = torch.randn(2, B, cfg.resampler.seq_len, cfg.resampler.embedding_dim, generator=g)
noise = torch.randn(B, 4, H, H, generator=g)
prefetch, overlap, npipes = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
for it in range(npipes):
    pipe = InstantIRPipeline(cfg, sd, scheduler=DDIMScheduler(), device="cuda:0")
    pipe.aggregator.load_state_dict(sda); pipe.prepare_previewers(lora, lora_alpha=8)
    pipe.overlap_streams = bool(overlap); pipe.use_graphs = it > 0 or npipes == 1
    pipe._build()
    for net in (pipe._unet, pipe._unet_prev, pipe._agg): net.prefetch = bool(prefetch)
    out = pipe(image=lq, prompt_embeds=pe, pooled_prompt_embeds=pooled, ip_adapter_image_embeds=[img], output_type="latent",
               num_inference_steps=3, guidance_scale=7.0, init_noise=noise,
               previewer_scheduler=LCMSingleStepScheduler.from_config(pipe.scheduler.config)).images
    torch.cuda.synchronize()
    print("pipe", it, "ok", float(out.abs().mean()), flush=True)
