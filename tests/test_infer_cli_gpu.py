"""GPU: the infer CLI end to end on synthetic tiny weights (infer.py:114-225 behaviour: resize_img, batching, skip of
finished outputs, resize back to the input size, one output file per input)."""
import os

import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu


def test_infer_cli_synthetic_tiny(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import numpy as np
    from instantir_amd.infer import build_parser, main
    src, out = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    rng = np.random.default_rng(0)
    for n in ("a.png", "b.png", "c.png"):
        Image.fromarray(rng.integers(0, 255, (96, 96, 3), dtype=np.uint8)).save(src / n)
    args = build_parser().parse_args(["--test_path", str(src), "--out_path", str(out), "--synthetic", "tiny", "--num_inference_steps", "2",
                                      "--width", "128", "--height", "128", "--batch_size", "2", "--cfg", "5.0", "--creative_start", "0.5"])
    import instantir_amd.infer as cli
    orig = cli.resize_img
    cli.resize_img = lambda im, **kw: orig(im, max_side=128, min_side=128, **kw)      # keep the tiny nets tiny
    try:
        main(args, torch.device("cuda:0"))
        files = sorted(os.listdir(out))
        assert files == ["a.png", "b.png", "c.png"]
        for f in files:
            im = Image.open(out / f)
            assert im.size == (128, 128) and im.mode == "RGB"
        before = {f: os.path.getmtime(out / f) for f in files}
        main(args, torch.device("cuda:0"))                                        # second run: everything is skipped
        assert {f: os.path.getmtime(out / f) for f in files} == before
    finally:
        cli.resize_img = orig
