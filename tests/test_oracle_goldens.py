"""CPU: the oracle restatement against vectors produced by the reference's own modules
(tests/golden/make_reference_goldens.py) and the known answers recorded in SURVEY.md section 8c."""
import os

import numpy as np
import torch

from instantir_amd.config import ResamplerConfig
from oracle import nets, sched


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    return {k: z[k] for k in z.files}


def _params(z, prefix):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in z.items() if k.startswith(prefix)}


def test_resampler_and_image_projection(golden_dir):
    z = _load(golden_dir, "ip_adapter.npz")
    P = {"encoder_hid_proj.image_projection_layers.0." + k: v for k, v in _params(z, "rs.").items()}
    rc = ResamplerConfig(dim=128, depth=2, dim_head=64, heads=2, num_queries=16, embedding_dim=64, output_dim=128,
                         seq_len=21)
    y = nets.image_projection(P, [torch.from_numpy(z["rs_in"])], rc)[0]
    assert y.shape == (6, 16, 128)
    np.testing.assert_allclose(y.numpy(), z["rs_out"], rtol=1e-4, atol=2e-5)


def test_ada_layer_norm(golden_dir):
    z = _load(golden_dir, "ip_adapter.npz")
    P = {"n." + k: v for k, v in _params(z, "aln.").items()}
    y = nets.ada_layer_norm(P, "n", torch.from_numpy(z["aln_x"]), torch.from_numpy(z["aln_t"]))
    np.testing.assert_allclose(y.numpy(), z["aln_out"], rtol=1e-5, atol=1e-5)


def test_attn_processor2_0(golden_dir):
    z = _load(golden_dir, "ip_adapter.npz")
    P = {"a." + k: v for k, v in _params(z, "sa.").items()}
    y = nets.attn_self(P, "a", torch.from_numpy(z["sa_x"]), heads=2)
    np.testing.assert_allclose(y.numpy(), z["sa_out"], rtol=1e-4, atol=1e-5)


def test_ta_ip_attn_processor(golden_dir):
    z = _load(golden_dir, "ip_adapter.npz")
    P = {"a." + k: v for k, v in _params(z, "ca.").items()}
    y = nets.attn_ta_ip(P, "a", torch.from_numpy(z["ca_x"]), torch.from_numpy(z["ca_ctx"]),
                        torch.from_numpy(z["ca_ip"]), torch.from_numpy(z["ca_temb"]), heads=2)
    np.testing.assert_allclose(y.numpy(), z["ca_out"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(z["ca_out_cat"], z["ca_out"], rtol=0, atol=0)   # both input forms agree


def test_lcm_tables_bit_exact(golden_dir):
    z = _load(golden_dir, "lcm_scheduler.npz")
    acp = sched.make_alphas_cumprod()
    assert np.array_equal(acp.numpy(), z["alphas_cumprod"])          # float32 table, bit for bit
    for n in (1, 2, 4, 8):
        got = sched.lcm_timesteps(n)
        assert got.dtype == np.int64 and np.array_equal(got, z[f"set_timesteps_{n}"])
    cs, co = sched.lcm_scalings(torch.from_numpy(z["t"]))
    assert np.array_equal(cs.numpy(), z["c_skip"]) and np.array_equal(co.numpy(), z["c_out"])


def test_lcm_step_and_add_noise(golden_dir):
    z = _load(golden_dir, "lcm_scheduler.npz")
    acp = sched.make_alphas_cumprod()
    x, e = torch.from_numpy(z["x"]), torch.from_numpy(z["eps"])
    for i, t in enumerate(z["t"]):
        got = sched.lcm_step(acp, e, int(t), x)
        assert np.array_equal(got.numpy(), z["step"][i])
        got = sched.add_noise(acp, x, e, [int(t)] * 2)
        assert np.array_equal(got.numpy(), z["add_noise"][i])


def test_survey_known_answers():
    """SURVEY.md section 8c known-answer values (computed there by the reference's scheduler)."""
    acp = sched.make_alphas_cumprod()
    for t, v in [(1, 0.9982960224), (34, 0.9678812623), (499, 0.2776694298), (958, 0.0075347675),
                 (999, 0.0046600951)]:
        assert abs(float(acp[t]) - v) < 5e-10 * max(1, 1 / v) + 1e-9
    cs, co = sched.lcm_scalings(torch.tensor([958, 34, 1]))
    np.testing.assert_allclose(cs.numpy(), [2.724012e-09, 2.162625e-06, 2.493766e-03], rtol=2e-6)
    np.testing.assert_allclose(co.numpy(), [1.0, 0.99999893, 0.99875236], rtol=1e-7)
    assert list(sched.lcm_timesteps(4)) == [999, 759, 519, 279]
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 4, 2, 2, generator=g)
    e = torch.randn(1, 4, 2, 2, generator=g)
    for t, want in [(958, [2.554154, 6.188374, -9.390113, -14.105405]),
                    (34, [-0.897961, -0.862384, -0.357927, -0.58557]),
                    (1, [-1.072388, -1.084799, -0.274486, -0.467532])]:
        got = sched.lcm_step(acp, e, t, x).flatten()[:4].numpy()
        np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-6)


def test_leading_timetable():
    """N=30 -> 958, 925, ..., 34, 1 (int64)."""
    ts = sched.leading_timesteps(30)
    assert ts.dtype == np.int64 and ts[0] == 958 and ts[1] == 925 and ts[-2] == 34 and ts[-1] == 1 and len(ts) == 30
    assert list(sched.leading_timesteps(4)) == [751, 501, 251, 1]


# ------------------------------------------------------------------------------------------------------------------
# module/min_sdxl.py (the in-tree SDXL UNet text) and SFT of module/aggregator.py: fixtures produced by running the
# reference's own classes (tests/golden/make_reference_goldens.py: gen_min_sdxl, gen_sft).  Parameters are re-created
# from (seed, inventory) with tests/golden/seeded.py; inputs and outputs are the committed arrays.
# ------------------------------------------------------------------------------------------------------------------
def _seeded_params(z, tag, prefix, zero=()):
    from golden.seeded import seeded_fill, unpack_inventory
    names, shapes = unpack_inventory(z[tag + "__inv"] if tag else z["inv"])
    sd = seeded_fill(names, shapes, int(z[tag + "__seed"] if tag else z["seed"]), zero)
    return {prefix + k: v for k, v in sd.items()}


TOL = dict(rtol=1e-4, atol=1e-4)     # oracle vs reference-produced vectors at <= 1e-4


def test_min_sdxl_timesteps_and_embedding(golden_dir):
    z = _load(golden_dir, "min_sdxl.npz")
    np.testing.assert_allclose(nets.sinusoid(torch.from_numpy(z["ts_t"]), 320).numpy(), z["ts_320"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(nets.sinusoid(torch.tensor([1024., 768., 0., 3.5]), 256).numpy(), z["ts_256"],
                               rtol=1e-6, atol=1e-6)
    P = _seeded_params(z, "te", "e.")
    x = torch.from_numpy(z["te_x"])
    y = nets.linear(P, "e.linear_2", torch.nn.functional.silu(nets.linear(P, "e.linear_1", x)))
    np.testing.assert_allclose(y.numpy(), z["te__out"], **TOL)


def test_min_sdxl_resnet_and_resamplers(golden_dir):
    z = _load(golden_dir, "min_sdxl.npz")
    x, temb = torch.from_numpy(z["x64"]), torch.from_numpy(z["temb"])
    for tag in ("res_sc", "res_id"):
        y = nets.resnet(_seeded_params(z, tag, "r."), "r", x, temb, 32)
        np.testing.assert_allclose(y.numpy(), z[tag + "__out"], **TOL)
    y = nets.conv2d(_seeded_params(z, "down", "d."), "d.conv", x, stride=2, padding=1)
    np.testing.assert_allclose(y.numpy(), z["down__out"], **TOL)
    up = torch.nn.functional.interpolate(x, scale_factor=2.0, mode="nearest")
    y = nets.conv2d(_seeded_params(z, "up", "u."), "u.conv", up)
    np.testing.assert_allclose(y.numpy(), z["up__out"], **TOL)


def test_min_sdxl_geglu_ff_transformer(golden_dir):
    z = _load(golden_dir, "min_sdxl.npz")
    tok, ctx = torch.from_numpy(z["tok"]), torch.from_numpy(z["ctx"])
    np.testing.assert_allclose(nets.geglu(_seeded_params(z, "geglu", "g."), "g", tok).numpy(), z["geglu__out"], **TOL)
    np.testing.assert_allclose(nets.feed_forward(_seeded_params(z, "ff", "f."), "f", tok).numpy(), z["ff__out"], **TOL)
    y = nets.transformer_block(_seeded_params(z, "btb", "b."), "b", tok, ctx, None, None, heads=2)
    np.testing.assert_allclose(y.numpy(), z["btb__out"], **TOL)
    y = nets.transformer2d(_seeded_params(z, "t2d", "t."), "t", torch.from_numpy(z["x128"]), 2, ctx, None, None, 2, 32)
    np.testing.assert_allclose(y.numpy(), z["t2d__out"], **TOL)


def test_min_sdxl_block_wrappers(golden_dir):
    z = _load(golden_dir, "min_sdxl.npz")
    x64, x128, temb, ctx = (torch.from_numpy(z[k]) for k in ("x64", "x128", "temb", "ctx"))
    # DownBlock2D: 2 resnets + downsampler, outputs [r0, r1, down]
    x, outs = nets.down_block(_seeded_params(z, "dblk", "d."), "d", x64, temb, None, None, 2, 0, 1, 32, True)
    for i, got in enumerate([x] + outs):
        np.testing.assert_allclose(got.numpy(), z[f"dblk__out{i}"], **TOL)
    # CrossAttnDownBlock2D with / without downsampler
    x, outs = nets.down_block(_seeded_params(z, "cadb", "d."), "d", x64, temb, ctx, None, 2, 1, 2, 32, True)
    for i, got in enumerate([x] + outs):
        np.testing.assert_allclose(got.numpy(), z[f"cadb__out{i}"], **TOL)
    x, outs = nets.down_block(_seeded_params(z, "cadb_nods", "d."), "d", x128, temb, ctx, None, 2, 1, 2, 32, False)
    assert len(outs) == 2
    for i, got in enumerate([x] + outs):
        np.testing.assert_allclose(got.numpy(), z[f"cadb_nods__out{i}"], **TOL)
    # CrossAttnUpBlock2D: skips consumed from the end of the tuple
    skips = [torch.from_numpy(z["caub_skip0"])] + [torch.from_numpy(a) for a in z["caub_skips"]]
    y = nets.up_block(_seeded_params(z, "caub", "u."), "u", x128, skips, temb, ctx, None, 1, 2, 32, True)
    np.testing.assert_allclose(y.numpy(), z["caub__out"], **TOL)
    skips = [torch.from_numpy(a) for a in z["upb_skips"]]
    y = nets.up_block(_seeded_params(z, "upb", "u."), "u", x128, skips, temb, None, None, 0, 1, 32, False)
    np.testing.assert_allclose(y.numpy(), z["upb__out"], **TOL)
    y = nets.mid_block(_seeded_params(z, "mid", "m."), "m", torch.from_numpy(z["xm"]), temb, ctx, None, 10, 2, 32)
    np.testing.assert_allclose(y.numpy(), z["mid__out"], **TOL)


def test_min_sdxl_full_unet_forward(golden_dir):
    """The reference's hard-coded SDXL-base UNet2DConditionModel (module/min_sdxl.py:789-915), one forward on a
    16x16 latent with 2.6 B seeded parameters, against oracle.nets.unet_forward (no IP branch)."""
    from instantir_amd.config import UNetConfig
    z = _load(golden_dir, "min_sdxl.npz")
    P = _seeded_params(z, "unet", "")
    y = nets.unet_forward(P, UNetConfig.sdxl(), torch.from_numpy(z["unet_sample"]), int(z["unet_t"]),
                          torch.from_numpy(z["unet_ctx"]), torch.from_numpy(z["unet_pooled"]),
                          torch.from_numpy(z["unet_time_ids"]), None)
    ref = z["unet__out0"]
    np.testing.assert_allclose(y.numpy(), ref, rtol=1e-4, atol=1e-4 * float(np.abs(ref).max()))


def test_sft_head(golden_dir):
    z = _load(golden_dir, "sft.npz")
    P = _seeded_params(z, "", "s.")
    y = nets.sft(P, "s", torch.from_numpy(z["c"]), torch.from_numpy(z["h"]))
    np.testing.assert_allclose(y.numpy(), z["out"], **TOL)
