#!/usr/bin/env python3
"""Golden vectors for BASELINE config C4 (window-7 attention on the 128 x 256 WeatherBench grid), produced by IMPORTING
the reference's own classes in this container (SURVEY.md §8c asked for exactly these and round 1 stopped at 28 x 28):

  * nsbench BasicLayer(window_size=7) (src/nsbench/models/swintransformer/swin_transformer.py:305-408; the dlwpbench block
    pads the wrong axes at this setting) on 32 x 64 and 128 x 256, embed 16, heads 4, constant and
    circular padding;
  * Pangu get_earth_position_index((2, 7, 7)) and get_shift_window_mask for the padded (2, 133, 259) canvas
    (src/dlwpbench/models/panguweather/utils/{earth_position_index,shift_window_mask}.py), and one shifted
    EarthSpecificBlock at (1, 128, 256) with window (2, 7, 7).

The 128 x 256 tensors are stored sub-sampled (every STRIDE-th token) and the inputs are re-generated from the recorded
seeds in the tests (a checksum guards the random stream); parameter gradients are stored in full.

    python tests/golden/make_c4_window7_golden.py        (writes tests/golden/c4_window7_golden.npz)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_pangu_golden  # noqa: E402
import make_swin_golden  # noqa: E402

STRIDE = 61


def seeded(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def main():
    out = {"stride": np.int32(STRIDE)}
    swin = make_swin_golden.load_reference()
    torch.manual_seed(777)
    for tag, (H, W, pm, B, seed) in {"32x64": (32, 64, "constant", 2, 11), "128x256": (128, 256, "constant", 1, 12),
                                     "128x256c": (128, 256, "circular", 1, 13)}.items():
        bl = swin.BasicLayer(dim=16, depth=2, num_heads=4, window_size=7, padding_mode=pm)
        with torch.no_grad():
            for n, p in bl.named_parameters():
                if "relative_position_bias_table" in n:
                    p.mul_(25.0)
        x = seeded(seed, B, H * W, 16).requires_grad_(True)
        y = bl(x, H, W)[0]
        gy = seeded(seed + 100, B, H * W, 16)
        y.backward(gy)
        out[f"bl_{tag}_cfg"] = np.array([H, W, B, seed, int(pm == "circular")], dtype=np.int32)
        out[f"bl_{tag}_xsum"] = np.float64(x.detach().double().abs().sum().item())
        out[f"bl_{tag}_y"] = y.detach()[:, ::STRIDE].numpy()
        out[f"bl_{tag}_gx"] = x.grad[:, ::STRIDE].numpy()
        out.update(make_swin_golden.params(bl, f"bl_{tag}_"))
        out.update(make_swin_golden.grads(bl, f"bl_{tag}_"))
    pangu = make_pangu_golden.load_reference()
    from models.panguweather.utils.earth_position_index import get_earth_position_index
    from models.panguweather.utils.shift_window_mask import get_shift_window_mask
    out["epi_2_7_7"] = get_earth_position_index((2, 7, 7)).numpy().astype(np.int32)
    mask = get_shift_window_mask((2, 133, 259), (2, 7, 7), (1, 3, 6))            # [n_lon, n_pl*n_lat, N, N]
    out["mask_shape"] = np.array(mask.shape, dtype=np.int32)
    out["mask_bits"] = np.packbits((mask != 0).numpy().reshape(-1))
    assert set(np.unique(mask.numpy())) <= {0.0, -100.0}
    torch.manual_seed(778)
    blk = pangu.EarthSpecificBlock(dim=8, input_resolution=(1, 128, 256), num_heads=2, window_size=(2, 7, 7), shift_size=None)
    blk.eval()
    with torch.no_grad():
        blk.attn.earth_position_bias_table.mul_(25.0)
    x = seeded(21, 1, 128 * 256, 8).requires_grad_(True)
    y = blk(x)
    gy = seeded(121, 1, 128 * 256, 8)
    y.backward(gy)
    out["pg_xsum"] = np.float64(x.detach().double().abs().sum().item())
    out["pg_y"] = y.detach()[:, ::STRIDE].numpy()
    out["pg_gx"] = x.grad[:, ::STRIDE].numpy()
    out.update(make_pangu_golden.params(blk, "pg_"))
    out.update(make_pangu_golden.grads(blk, "pg_"))
    path = os.path.join(HERE, "c4_window7_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
