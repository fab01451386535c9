"""GPU parity of the image preprocessing kernels (medplib_amd/preprocess.py through the C ABI) — all BIT-EXACT: the resize against
the oracle's restatement of PIL's 8-bit bilinear resampler (itself pinned to the real PIL in tests/test_preprocess.py) and against
the golden file produced by executing the reference's own preprocessing code; the SAM / CLIP normalisations are per-value tables, so
the float outputs are compared with equality too.  Large sizes: against PIL itself when it is importable on the box, plus the
size-independent property that a constant image stays constant."""
import os

import numpy as np
import pytest
import torch

from oracle import preprocess as P

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "preprocess_reference.npz")


def test_resize_bit_exact_vs_oracle(dev):
    from medplib_amd import preprocess as PP
    rng = np.random.default_rng(3)
    for h, w, t in [(480, 640, 256), (640, 480, 336), (100, 37, 336), (333, 777, 336), (256, 256, 256), (200, 256, 256),
                    (37, 100, 256), (50, 50, 336), (1, 9, 256), (9, 1, 336), (513, 1027, 256)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        got = PP.ResizeLongestSide(t).apply_image(torch.from_numpy(img).to(dev)).cpu().numpy()
        assert np.array_equal(got, P.resize_longest_side(img, t)), (h, w, t)
        m = (rng.random((h, w)) > 0.5).astype(np.uint8)
        gm = PP.ResizeLongestSide(t).apply_image(torch.from_numpy(m).to(dev)).cpu().numpy()
        assert np.array_equal(gm, P.resize_longest_side(m, t)), (h, w, t)


def test_preprocess_against_executed_reference_golden(dev):
    from medplib_amd import preprocess as PP
    z = np.load(GOLD)
    for i in range(int(z["n_cases"])):
        img = torch.from_numpy(z[f"img{i}"]).to(dev)
        s, rs = PP.preprocess_sam(img)
        assert s.dtype == torch.float32 and np.array_equal(s.cpu().numpy(), z[f"sam_out{i}"])
        assert tuple(rs) == z[f"sam_resized{i}"].shape[:2]
        c = PP.preprocess_clip(img)
        assert np.array_equal(c.cpu().numpy(), z[f"clip_out{i}"])
        r = PP.preprocess_region_mask(torch.from_numpy(z[f"mask{i}"]).to(dev))
        assert np.array_equal(r.cpu().numpy(), z[f"region_mask{i}"])
        # bf16 output = the fp32 value rounded once (what the model's own cast does, MedPLIB.py image casts)
        cb = PP.preprocess_clip(img, out_dtype=torch.bfloat16)
        assert torch.equal(cb.cpu(), torch.from_numpy(z[f"clip_out{i}"]).to(torch.bfloat16))


def test_large_image_vs_pil_and_constant_property(dev):
    from medplib_amd import preprocess as PP
    rng = np.random.default_rng(5)
    h, w = 3000, 4000                                  # a 12-megapixel photograph
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    got = PP.ResizeLongestSide(336).apply_image(torch.from_numpy(img).to(dev)).cpu().numpy()
    try:
        from PIL import Image
        nh, nw = P.get_preprocess_shape(h, w, 336)
        assert np.array_equal(got, np.array(Image.fromarray(img).resize((nw, nh), Image.BILINEAR)))
    except ImportError:
        assert np.array_equal(got, P.resize_longest_side(img, 336))
    const = np.full((h, w, 3), 200, dtype=np.uint8)
    out = PP.ResizeLongestSide(256).apply_image(torch.from_numpy(const).to(dev))
    assert int(out.min()) == 200 and int(out.max()) == 200     # the normalised coefficients of every window sum to 1 << 22


def test_preprocessed_images_feed_the_model(dev):
    """The preprocessed tensors are the `images` / `images_clip` entries of the batch dict: shapes, dtypes and the centre padding."""
    from medplib_amd import preprocess as PP
    rng = np.random.default_rng(9)
    img = torch.from_numpy(rng.integers(0, 256, (300, 200, 3), dtype=np.uint8)).to(dev)
    s, rs = PP.preprocess_sam(img)
    assert s.shape == (3, 256, 256) and rs == (256, 171)
    left = (256 - 171) // 2
    assert float(s[:, :, :left].abs().max()) == 0.0 and float(s[:, :, left + 171:].abs().max()) == 0.0
    c = PP.preprocess_clip(img)
    assert c.shape == (3, 336, 336)
    tab = P.clip_value_table()
    assert abs(float(c[0, 0, 0]) - float(tab[0, 122])) == 0.0      # the CLIP pad value is the integer mean pushed through the table
