"""Host side of the byte-input feed: statistics, the host restatement of main.py's normalisation, .npy loading."""
import numpy as np
import pytest

from unet_amd import data


def test_channel_stats_match_the_reference_recipe():
    images, _ = data.synthetic_dataset(5, 32, 48, seed=1)
    mean, std = data.channel_stats(images, chunk=2)
    x = np.transpose(images, (0, 3, 1, 2)).astype(np.float32) / 255.0      # main.py:85-91
    np.testing.assert_allclose(mean, x.mean(axis=(0, 2, 3)), rtol=1e-5)
    np.testing.assert_allclose(std, x.std(axis=(0, 2, 3)), rtol=1e-4)
    y = data.normalize_host(images, mean, std)
    assert y.shape == (5, 3, 32, 48) and y.dtype == np.float32
    np.testing.assert_allclose(y.mean(axis=(0, 2, 3)), 0.0, atol=1e-4)
    np.testing.assert_allclose(y.std(axis=(0, 2, 3)), 1.0, atol=1e-3)


def test_masks_are_scaled_only_when_stored_as_0_255():
    _, m255 = data.synthetic_dataset(2, 16, 16, seed=2)
    _, m1 = data.synthetic_dataset(2, 16, 16, seed=2, mask_value=1)
    assert set(np.unique(data.masks_host(m255))) <= {0.0, 1.0}
    np.testing.assert_array_equal(data.masks_host(m255), data.masks_host(m1))


def test_npy_round_trip_and_validation(tmp_path):
    images, masks = data.synthetic_dataset(3, 16, 24, seed=3)
    np.save(tmp_path / "i.npy", images)
    np.save(tmp_path / "m.npy", masks)
    li, lm = data.load_npy(tmp_path / "i.npy", tmp_path / "m.npy")
    np.testing.assert_array_equal(li, images)
    np.testing.assert_array_equal(lm, masks)
    np.save(tmp_path / "f.npy", images.astype(np.float32))
    with pytest.raises(ValueError):
        data.load_npy(tmp_path / "f.npy", tmp_path / "m.npy")
    np.save(tmp_path / "m2.npy", masks[:, :8])
    with pytest.raises(ValueError):
        data.load_npy(tmp_path / "i.npy", tmp_path / "m2.npy")
    with pytest.raises(ValueError):
        data.channel_stats(images.astype(np.float32))
