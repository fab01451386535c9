"""Image preprocessing, CPU side: the oracle's restatement of PIL's 8-bit bilinear resampler against the REAL PIL
in this image (byte equality on random images / sizes), against the golden file written by executing the reference's own
`LazySupervisedDataset.preprocess` / `ResizeLongestSide.get_preprocess_shape` (oracle/make_golden.py preprocess), and the product's
host-side coefficient routine (C, in the library) against the oracle's."""
import os

import numpy as np
import pytest

from oracle import preprocess as P

GOLD = os.path.join(os.path.dirname(__file__), "golden", "preprocess_reference.npz")


def test_oracle_resize_equals_pil_bytes():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(0)
    cases = [(480, 640, 256), (640, 480, 336), (100, 37, 336), (333, 777, 336), (256, 256, 256), (200, 256, 256), (37, 100, 256),
             (50, 50, 336), (1, 9, 256), (9, 1, 336), (700, 1100, 256)]
    for h, w, t in cases:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        nh, nw = P.get_preprocess_shape(h, w, t)
        ref = np.array(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
        assert np.array_equal(ref, P.pil_resize_bilinear(img, nh, nw)), (h, w, t)
        m = (rng.random((h, w)) > 0.5).astype(np.uint8) * 255
        refm = np.array(Image.fromarray(m).resize((nw, nh), Image.BILINEAR))
        assert np.array_equal(refm, P.pil_resize_bilinear(m, nh, nw)), (h, w, t)


def test_oracle_against_executed_reference_golden():
    z = np.load(GOLD)
    for i in range(int(z["n_cases"])):
        img, mask = z[f"img{i}"], z[f"mask{i}"]
        assert np.array_equal(P.resize_longest_side(img, 256), z[f"sam_resized{i}"])
        assert np.array_equal(P.resize_longest_side(img, 336), z[f"clip_resized{i}"])
        s, rs = P.preprocess_sam(img)
        assert s.dtype == np.float32 and np.array_equal(s, z[f"sam_out{i}"])          # float values identical (table of 256 values)
        assert tuple(rs) == z[f"sam_resized{i}"].shape[:2]
        assert np.array_equal(P.preprocess_clip(img), z[f"clip_out{i}"])
        assert np.array_equal(P.preprocess_region_mask(mask), z[f"region_mask{i}"])


def test_clip_pad_values_and_tables():
    assert P.clip_pad_values() == [122, 116, 104]                                      # (mean * 255).clamp(0, 255).to(int)
    t = P.sam_value_table()
    assert t.shape == (3, 256) and abs(float(t[0, 0]) + 123.675 / 58.395) < 1e-6


def test_library_coefficients_equal_oracle():
    """mp_pil_bilinear_coeffs (host C in the shipped library) vs the oracle's Python restatement of Resample.c."""
    from medplib_amd.preprocess import bilinear_coeffs_host, get_preprocess_shape
    for n_in, n_out in [(640, 256), (480, 192), (37, 124), (100, 336), (256, 256), (1600, 336), (7, 3), (3, 7), (1, 5), (1024, 1)]:
        b, c = bilinear_coeffs_host(n_in, n_out)
        bo, co = P.bilinear_coeffs(n_in, n_out)
        assert np.array_equal(b, bo) and np.array_equal(c, co), (n_in, n_out)
    for h, w, t in [(480, 640, 256), (1, 1, 336), (1023, 17, 256), (336, 336, 336)]:
        assert get_preprocess_shape(h, w, t) == P.get_preprocess_shape(h, w, t)
