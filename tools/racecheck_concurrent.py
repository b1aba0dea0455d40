"""Race screen UNDER CONCURRENCY: each kernel is launched REP times on one stream while a second stream keeps the chip busy with
other kernels (the step runs two streams: main UNet encoder beside previewer UNet + Aggregator).  Every kernel is deterministic,
so the REP outputs must be bit-identical; a kernel whose result depends on timing (a missing wait / barrier) shows up here even if
it is reproducible when it runs alone."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from instantir_amd import ops
from instantir_amd.packing import pair_rows, conv_weight_nhwc
dev = torch.device("cuda:0")
REP = int(os.environ.get("REP", "24"))
NOISE = os.environ.get("NOISE", "1") != "0"
g = torch.Generator().manual_seed(1)
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).half().to(dev)
side = torch.cuda.Stream()
# noise: a mix of the step's kernels
nx, nw, no = rnd(4096, 1280), rnd(2560, 1280, scale=0.03), torch.empty(4096, 2560, dtype=torch.half, device=dev)
nq, nvt, nao = rnd(2 * 2048, 2 * 640), rnd(640, 2 * 2048), torch.empty(2 * 2048, 640, dtype=torch.half, device=dev)
ncx, ncw, nco = rnd(2, 64, 64, 640), rnd(640, 3, 3, 640, scale=0.02), torch.empty(2 * 64 * 64, 640, dtype=torch.half, device=dev)
def noise(n):
    with torch.cuda.stream(side):
        for _ in range(n):
            ops.gemm(nx, nw, no)
            ops.attention(nq[:, :640], nao, [(nq[:, 640:], 2048, nvt, 2048, 2048)], 2, 10, 2048)
            ops.conv2d(ncx, ncw, nco)
bad = []
def screen(name, launch, shapes, dtype=torch.half):
    first = None; nd = 0; where = None
    for it in range(REP):
        dts = list(dtype) if isinstance(dtype, (list, tuple)) else [dtype] * len(shapes)
        outs = [torch.zeros(s, dtype=dt_, device=dev) for s, dt_ in zip(shapes, dts)]
        if NOISE: noise(3)
        launch(*outs)
        torch.cuda.synchronize()
        cat = torch.cat([o.flatten().float() for o in outs])
        if first is None: first = cat
        elif not torch.equal(cat, first):
            nd += 1
            idx = (cat != first).nonzero().flatten()
            where = (int(idx.numel()), idx[:4].tolist(), float((cat - first).abs().max()))
    if nd: bad.append(name)
    print(f"{name:60s} nondeterministic_runs={nd}/{REP - 1} {where or ''}", flush=True)

for (M, N, K) in [(2048, 1280, 1280), (8192, 640, 640), (2048, 1280, 5120), (2048, 3840, 1280)]:
    x, w, b, res = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N), rnd(M, N)
    for tile in (0, 21, 22, 24, 25, 35, 55):
        screen(f"gemm {M}x{N}x{K} tile {tile} bias+res", lambda o: ops.gemm(x, w, o, bias=b, res=res, tile=tile), [(M, N)])
M, N, K = 2048, 10240, 1280
x, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
wp, bp = pair_rows(w[:N // 2], w[N // 2:]), pair_rows(b[:N // 2], b[N // 2:])
screen("gemm GEGLU 2048x10240x1280", lambda o: ops.gemm(x, wp, o, bias=bp, epi=ops.EPI_GEGLU), [(M, N // 2)])
C = 1280
w3 = rnd(3 * C, C, scale=C ** -0.5)
screen("gemm q|k|v 2048x3840x1280 (V transposed)", lambda qk, vt: ops.gemm(x, w3, qk, out_t=(vt, 2 * C)), [(M, 2 * C), (C, M)])
# LayerNorm fold: producer partials, consumer
parts = ops.ln_parts(M, C, C)
h0, wo = rnd(M, C), rnd(C, C, scale=C ** -0.5)
st = torch.zeros(parts, M, 2, device=dev)
hh = torch.empty(M, C, dtype=torch.half, device=dev)
def prod(o, s):
    ops.gemm(x, wo, o, res=h0, ln_out=st); s.copy_(st)
screen("gemm producer with ln_out (output + partials)", prod, [(M, C)], ) if False else None
for it in range(1):
    outs = []
    first = None; nd = 0
    for r_ in range(REP):
        st.zero_(); 
        if NOISE: noise(3)
        ops.gemm(x, wo, hh, res=h0, ln_out=st); torch.cuda.synchronize()
        cat = torch.cat([hh.flatten().float(), st.flatten()])
        if first is None: first = cat
        elif not torch.equal(cat, first): nd += 1
    if nd: bad.append("ln producer")
    print(f"{'gemm producer with ln_out (output + partials)':60s} nondeterministic_runs={nd}/{REP - 1}", flush=True)
gam, bet = rnd(C) + 1, rnd(C)
f = ops.LnFold(w3, gam, bet)
screen("gemm consumer with ln_in (q|k|v)", lambda qk, vt: ops.gemm(hh, f.w, qk, bias=f.bias, out_t=(vt, 2 * C), ln_in=(st, f.colsum, 1e-5)), [(M, 2 * C), (C, M)])
f1 = ops.LnFold(w, gam, bet, bias=b, pair=pair_rows)
screen("gemm consumer with ln_in (GEGLU)", lambda o: ops.gemm(hh, f1.w, o, bias=f1.bias, epi=ops.EPI_GEGLU, ln_in=(st, f1.colsum, 1e-5)), [(M, N // 2)])
for (R, H, Cin, Cout) in [(2, 32, 1280, 1280), (2, 64, 640, 640), (2, 128, 320, 320), (2, 32, 2560, 1280)]:
    cx, cw, cb = rnd(R, H, H, Cin), rnd(Cout, 3, 3, Cin, scale=(9 * Cin) ** -0.5), rnd(Cout)
    for tile in (0,):
        screen(f"conv3x3 {R}x{H}x{H} {Cin}->{Cout} tile {tile}", lambda o: ops.conv2d(cx, cw, o, bias=cb, tile=tile), [(R * H * H, Cout)])
for (B, heads, T, kvs) in [(2, 20, 1024, [1024]), (2, 10, 4096, [4096]), (2, 20, 1024, [77, 64]), (2, 10, 4096, [77, 64]), (2, 10, 8192, [8192])]:
    Cc = heads * 64
    q = rnd(B * T, Cc); segs = []
    for Tk in kvs:
        pad = (Tk + 7) // 8 * 8
        segs.append((rnd(B * Tk, Cc), Tk, rnd(Cc, B * pad), pad, Tk))
    screen(f"attention B={B} h={heads} T={T} kv={kvs}", lambda o: ops.attention(q, o, segs, B, heads, T), [(B * T, Cc)])
for (R, HW, Cg) in [(2, 1024, 1280), (2, 4096, 640), (2, 16384, 320), (2, 1024, 2560)]:
    gx, gg, gb = rnd(R * HW, Cg), rnd(Cg) + 1, rnd(Cg)
    ws = ops.gn_workspace(dev, R, 32)
    screen(f"groupnorm R={R} HW={HW} C={Cg}", lambda o: ops.groupnorm(gx, o, R, HW, gg, gb, 1e-5, True, 32, ws), [(R * HW, Cg)])
# conv variants: stride 2, nearest-2x upsample folded, 1x1, SFT pair epilogue, loader-wave tile
cx, cw, cb = rnd(2, 64, 64, 640), rnd(640, 3, 3, 640, scale=0.02), rnd(640)
screen("conv3x3 stride 2 2x64x64 640->640", lambda o: ops.conv2d(cx, cw, o, bias=cb, stride=2), [(2 * 32 * 32, 640)])
screen("conv3x3 upsample 2x 2x64x64 640->640", lambda o: ops.conv2d(cx, cw, o, bias=cb, upsample=True), [(2 * 128 * 128, 640)])
c1w = rnd(1280, 1, 1, 640, scale=0.04)
screen("conv1x1 2x64x64 640->1280", lambda o: ops.conv2d(cx, c1w, o, ksize=1), [(2 * 64 * 64, 1280)])
sx, sw, sb, sres = rnd(2, 32, 32, 256), pair_rows(rnd(1280, 3, 3, 256, scale=0.02), rnd(1280, 3, 3, 256, scale=0.02)), pair_rows(rnd(1280), rnd(1280)), rnd(2 * 32 * 32, 1280)
screen("conv3x3 SFT epilogue 2x32x32 256->2x1280", lambda o: ops.conv2d(sx, sw, o, bias=sb, res=sres, epi=ops.EPI_SFT), [(2 * 32 * 32, 1280)])
l2x, l2w = rnd(2, 32, 32, 1280), rnd(1280, 3, 3, 1280, scale=0.01)
screen("conv3x3 2x32x32 1280->1280 tile 55 (loader waves)", lambda o: ops.conv2d(l2x, l2w, o, tile=55), [(2 * 32 * 32, 1280)])
# pointwise / glue
ca, cadd, csc = rnd(8192, 640), rnd(8192, 640), torch.tensor([0.7, 1.0], device=dev)
screen("copy_add 8192x640 -> cat buffer", lambda o: ops.copy_add(ca, o, 640, add=cadd, add_scale=csc, rows_per_scale=4096), [(8192, 1280)])
sil = rnd(2, 1280)
screen("silu", lambda o: ops.silu(sil, o), [(2, 1280)])
# the small launches of a step (few workgroups: the occupancy of the finalize kernel that failed)
xl = (torch.randn(1, 4, 128, 128, generator=g) * 0.8).to(dev)
screen("pack_latent (rep 2)", lambda o: ops.pack_latent(xl, o, rep=2), [(2 * 128 * 128, 64)])
eps2 = rnd(2 * 128 * 128, 64)
coef = torch.tensor([7.0, 0.9, 0.3, 0.8, 0.5, 0.0, 0.0, 0.0], device=dev)
screen("sched_step (CFG + DDIM)", lambda o: ops.sched_step(eps2, 1, coef, xl, o), [(1, 4, 128, 128)], dtype=torch.float32)
lcoef = torch.tensor([0.1, 0.9, 0.3, 0.95], device=dev)
screen("lcm_step", lambda o, p32: ops.lcm_step(eps2, 1, 2, lcoef, xl, o, p32), [(2 * 128 * 128, 64), (2, 4, 128, 128)]) if False else None
tv = torch.tensor([[499.0], [499.0]], device=dev)
screen("sinusoid", lambda o: ops.sinusoid(tv, o, 320), [(2, 320)])
e1, tw, tb_ = rnd(2, 1280), rnd(1280, 320, scale=0.05), rnd(1280)
sn_ = rnd(2, 320)
screen("gemm M=2 (time embedding)", lambda o: ops.gemm(sn_, tw, o, bias=tb_, act=ops.ACT_SILU), [(2, 1280)])
lnx, lng, lnb = rnd(2 * 64, 2048), rnd(2048) + 1, rnd(2048)
screen("layernorm 128x2048 (ip tokens)", lambda o: ops.layernorm(lnx, o, lng, lnb, 1e-6), [(128, 2048)])
gsm, gsg, gsb = rnd(2 * 64, 1280), rnd(1280) + 1, rnd(1280)
wsm = ops.gn_workspace(dev, 2, 32)
screen("groupnorm R=2 HW=64 C=1280 (8x8 map)", lambda o: ops.groupnorm(gsm, o, 2, 64, gsg, gsb, 1e-5, True, 32, wsm), [(128, 1280)])
lx, lg, lb = rnd(2048, 1280), rnd(1280) + 1, rnd(1280)
screen("layernorm 2048x1280", lambda o: ops.layernorm(lx, o, lg, lb, 1e-5), [(2048, 1280)])
# ---- round 3 additions: the kernels the first screen left out, and the 8-wave GEMM
for (M8, N8, K8, geglu8) in [(2048, 10240, 1280, True), (4096, 10240, 1280, True), (8192, 5120, 640, True), (2048, 10240, 1280, False)]:
    x8, w8, b8 = rnd(M8, K8), rnd(N8, K8, scale=K8 ** -0.5), rnd(N8)
    if geglu8:
        w8, b8 = pair_rows(w8[:N8 // 2], w8[N8 // 2:]), pair_rows(b8[:N8 // 2], b8[N8 // 2:])
    screen(f"gemm8 (256x320, 8 waves) {M8}x{N8}x{K8} {'GEGLU' if geglu8 else 'plain'}",
           lambda o: ops.gemm(x8, w8, o, bias=b8, tile=91, epi=ops.EPI_GEGLU if geglu8 else ops.EPI_PLAIN), [(M8, N8 // 2 if geglu8 else N8)])
# to_q + cross-attention in one launch (IIR_EPI_XATTN), level-2 and level-1 geometry
for (Rx, Tx, hx) in [(2, 1024, 20), (2, 4096, 10)]:
    Cx = hx * 64
    xa_a, xa_w = rnd(Rx * Tx, Cx), rnd(Cx, Cx, scale=Cx ** -0.5 * ops.attn_q_factor())
    xa_segs = []
    for Lx in (77, 64):
        tpx = (Lx + 7) // 8 * 8
        xa_segs.append((rnd(Rx * Lx, Cx), Lx, rnd(Cx, Rx * tpx), tpx, Lx))
    screen(f"gemm + cross-attention epilogue R={Rx} T={Tx} heads={hx}",
           lambda o: ops.gemm(xa_a, xa_w, o, epi=ops.EPI_XATTN, xattn=(xa_segs, Tx)), [(Rx * Tx, Cx)])
# both operands fp8 (configs[4] build): GEMM (4-wave one-per-CU tile, 8-wave tile, fp8 result), LayerNorm and attention storing fp8
for (Mf, Nf, Kf, tile_f, geglu_f, out8) in [(2048, 1280, 1280, 0, False, False), (4096, 1280, 5120, 0, False, False), (4096, 10240, 1280, 91, True, True),
                                          (16384, 5120, 640, 0, True, True)]:
    fa8, fsa = ops.quantize_fp8_tensor(rnd(Mf, Kf))
    fw, fb = rnd(Nf, Kf, scale=Kf ** -0.5), rnd(Nf)
    if geglu_f:
        fw, fb = pair_rows(fw[:Nf // 2], fw[Nf // 2:]), pair_rows(fb[:Nf // 2], fb[Nf // 2:])
    fw8 = ops.Fp8Weight(*ops.quantize_fp8_rows(fw))
    screen(f"gemm fp8 x fp8 {Mf}x{Nf}x{Kf} tile {tile_f}{' GEGLU' if geglu_f else ''}{' -> fp8' if out8 else ''}",
           lambda o: ops.gemm_fp8(fa8, fw8, o, a_scale=fsa, bias=fb, tile=tile_f, epi=ops.EPI_GEGLU if geglu_f else ops.EPI_PLAIN),
           [(Mf, Nf // 2 if geglu_f else Nf)], dtype=torch.uint8 if out8 else torch.half)
screen("layernorm 2048x1280 -> fp8", lambda o: ops.layernorm(lx, o, lg, lb, 1e-5), [(2048, 1280)], dtype=torch.uint8)
tx = rnd(154, 1280)
screen("transpose 154x1280 -> (1280, 160)", lambda o: ops.transpose(tx, o, 160), [(1280, 160)])
# adaLN batch: 8 jobs of the step's geometry (2 rows x 64 IP tokens, C = 1280 / 640), half of them transposed
ada_x = [rnd(128, Cj) for Cj in (1280, 640) * 4]
ada_sh, ada_sc = rnd(2, 2 * 1280 * 8), None
def ada_launch(*outs):
    jobs = []
    for j, (xj, oj) in enumerate(zip(ada_x, outs)):
        Cj = xj.shape[1]
        jobs.append((xj, oj, ada_sh[:, j * 2560:j * 2560 + Cj], ada_sh[:, j * 2560 + 1280:j * 2560 + 1280 + Cj], j % 2 == 1))
    tab = ops.adaln_job_table(jobs, dev)
    ops.adaln_batch(tab, len(jobs), 128, 1280, ada_sh.stride(0), 64, 64, 64)
    torch.cuda.synchronize()
screen("adaln_batch 8 jobs (4 transposed)", ada_launch, [((Cj, 128) if j % 2 == 1 else (128, Cj)) for j, Cj in enumerate((1280, 640) * 4)])
sm_s = (torch.randn(4096, 4096, generator=g) * 3).to(dev)
screen("softmax_rows_f32 4096x4096 -> bf16", lambda o: ops.softmax_rows_f32(sm_s, o), [(4096, 4096)], dtype=torch.bfloat16)
bl_a = torch.randn(1, 3, 1024, 1024, generator=g).to(dev)
bl_b0 = torch.randn(1, 3, 1024, 1024, generator=g).to(dev)
screen("blend_tiles (vertical, 256 rows)", lambda o: (o.copy_(bl_b0), ops.blend_tiles(bl_a, o, 256, True)), [(1, 3, 1024, 1024)], dtype=torch.float32)
screen("lcm_step (rep 2, with fp32 preview)", lambda o, p32: ops.lcm_step(eps2, 1, 2, lcoef, xl, o, p32), [(2 * 128 * 128, 64), (2, 4, 128, 128)], dtype=[torch.half, torch.float32])
fac = torch.ones(1, device=dev)
screen("cfg_rescale_factor", lambda o: (ops.cfg_rescale_factor(eps2, 1, coef, xl, 0.7, fac), o.copy_(fac)), [(1,)], dtype=torch.float32)
screen("unpack_latent", lambda o: ops.unpack_latent(eps2, o), [(2, 4, 128, 128)], dtype=torch.float32)
print("kernels with run-to-run differences:", bad)
