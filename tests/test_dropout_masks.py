"""The numpy restatement of the kernels' dropout hashes (tests/dropout_masks.py) on its own: rates, scale, the quantised threshold and
the independence of neighbouring decisions.  The comparison with the kernels' masks is tests/test_gpu_dropout_parity.py."""
import numpy as np

from tests import dropout_masks as DM


def test_threshold_and_scale_quantisation():
    assert DM.drop_th8(0.) == 0 and DM.drop_th8(0.1) == 26 and DM.drop_th8(0.5) == 128
    assert DM.drop_th8(1e-4) == 1 and DM.drop_th8(0.999) == 255            # p > 0 always drops something and never everything
    assert abs(DM.inv_keep8(26) - 256. / 230.) < 1e-6


def test_element_masks_have_the_stated_rate_and_differ_per_stream_and_seed():
    a, scale = DM.elem_keep(0x1234567, 1, [0, 3], 50, 1024, 0.1)
    assert a.shape == (2, 50, 1024) and abs(float(a.mean()) - 230. / 256.) < 3e-3 and abs(scale - 256. / 230.) < 1e-6
    b, _ = DM.elem_keep(0x1234567, 0, [0, 3], 50, 1024, 0.1)
    c, _ = DM.elem_keep(0x1234568, 1, [0, 3], 50, 1024, 0.1)
    for other in (b, c):
        agree = float((a == other).mean())
        assert abs(agree - ((230. / 256.) ** 2 + (26. / 256.) ** 2)) < 5e-3, agree       # independent draws
    # the four decisions that share one hash word are independent of each other
    x = a.reshape(-1, 4).astype(np.float64)
    cc = np.corrcoef(x.T)
    assert np.abs(cc - np.eye(4)).max() < 2e-2, cc
    # a row of the batch does not depend on which other rows are asked for
    only3, _ = DM.elem_keep(0x1234567, 1, [3], 50, 1024, 0.1)
    assert np.array_equal(only3[0], a[1])


def test_attention_masks_have_the_stated_rate_per_head():
    keep, scale = DM.attn_keep(0x7654321, [1, 2], 2, 200, 0.1)
    assert keep.shape == (2, 2, 200, 200) and abs(scale - 256. / 230.) < 1e-6
    for b in range(2):
        for h in range(2):
            assert abs(float(keep[b, h].mean()) - 230. / 256.) < 6e-3
    assert float((keep[0, 0] == keep[0, 1]).mean()) < 0.86                  # heads draw different masks
    blk = keep[0, 0].reshape(50, 4, 50, 4).transpose(0, 2, 1, 3).reshape(2500, 16).astype(np.float64)
    cc = np.corrcoef(blk.T)
    assert np.abs(cc - np.eye(16)).max() < 8e-2, np.abs(cc - np.eye(16)).max()
