"""Host-side logic of the trainer drop-in: dataset semantics (reference datasets.py:21-46), command
line (starter.py:17-35) and checkpoint directory naming (models/cvae.py:223-226).  No GPU."""
import os

import numpy as np
import pytest
import torch

from cvae_gan_amd import datasets as D
from cvae_gan_amd import starter


def test_label_rule():
    assert D.label_of("pose_05_0007.png") == 3            # int(name.split('_')[1]) - 2
    assert D.label_of("a_2_x") == 0


def test_bodymap_dataset_semantics(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    root = tmp_path / "data" / "Unit"
    (root / "GT_output").mkdir(parents=True)
    names = ["im_02_a.png", "im_04_b.png", "im_03_c.png"]
    rgb = np.zeros((8, 8, 3), dtype=np.uint8)
    rgb[..., 0], rgb[..., 1], rgb[..., 2] = 255, 128, 0    # R, G, B
    for n in names:
        Image.fromarray(rgb).save(str(root / "GT_output" / n))
    (root / "input_Unit.txt").write_text("\n".join(names) + "\n")
    ds = D.BodyMapDataset(str(tmp_path / "data"), "Unit", dim=4, max_size=-1, cls_num=5)
    assert len(ds) == 2                                    # the reference's [:max_size] with -1 drops the last name
    img, y = ds[1]
    assert img.shape == (3, 4, 4) and img.dtype == torch.float32
    # BGR order, (v/255 - .5) / .5
    assert abs(float(img[0].mean()) - (-1.0)) < 1e-6       # B = 0
    assert abs(float(img[1].mean()) - (128 / 255 - 0.5) / 0.5) < 1e-6
    assert abs(float(img[2].mean()) - 1.0) < 1e-6          # R = 255
    assert y.tolist() == [0, 0, 1, 0, 0]                   # "04" - 2
    full = D.BodyMapDataset(str(tmp_path / "data"), "Unit", dim=4, max_size=None, cls_num=5)
    assert len(full) == 3


def test_synthetic_fallback(tmp_path):
    ds, synthetic = D.make_dataset(str(tmp_path), "nothing-here", 16, 4, max_size=-1, synthetic_size=12)
    assert synthetic and len(ds) == 12
    img, y = ds[5]
    assert img.shape == (3, 16, 16) and float(img.min()) >= -1 and float(img.max()) <= 1
    assert int(y.argmax()) == 5 % 4 and float(y.sum()) == 1
    img2, _ = ds[5]
    assert torch.equal(img, img2)                          # deterministic


def test_cli_flags_match_reference():
    p = starter.build_parser()
    a = p.parse_args(["--dataset", "PoseUnit-stretch", "--data_dir", "./data", "--epoch", "10000", "--batch_size", "550",
                      "--data_size", "-1", "--z_dim", "32", "--y_dim", "16", "--pix_dim", "64", "--gpus", "0,1",
                      "--worker", "48", "--lrG", "3e-5", "--resume"])                       # train.sh
    assert (a.batch_size, a.z_dim, a.y_dim, a.pix_dim, a.resume, a.testmode) == (550, 32, 16, 64, True, False)
    assert a.lrG == 3e-5 and a.save_dir == "checkpoint" and a.result_dir == "result" and a.gan_type == "VAE"
    b = p.parse_args(["--dataset", "d", "--data_dir", "x", "--pkl", "./checkpoint/p/d", "--testmode"])   # test.sh
    assert b.testmode and b.pkl == "./checkpoint/p/d" and b.epoch == 200 and b.lrG == 1e-3
    with pytest.raises(SystemExit):
        p.parse_args(["--data_dir", "x"])                  # --dataset is required


def test_model_dir_format():
    from cvae_gan_amd.cvae import CVAE
    obj = CVAE.__new__(CVAE)
    obj.dataset, obj.pix_dim, obj.batch_size, obj.z_dim = "PoseUnit-stretch", 64, 550, 32
    assert obj.model_dir == "VAE_data_PoseUnit-stretch_pix_64_batch_550_embed_32"   # test.sh's --pkl path


def test_mnist_shaped_inputs():
    """BASELINE configuration 1's plumbing: 28 x 28 gray digits -> 32 x 32 x 3 in [-1, 1], black border."""
    g = torch.rand(5, 1, 28, 28)
    x = D.mnist_to_input(g, 32)
    assert x.shape == (5, 3, 32, 32) and float(x.min()) >= -1 and float(x.max()) <= 1
    assert torch.equal(x[:, 0], x[:, 1]) and torch.equal(x[:, 0], x[:, 2])
    assert torch.all(x[:, :, :2] == -1) and torch.all(x[:, :, :, -2:] == -1)
    assert torch.allclose(x[:, 0, 2:30, 2:30], (g[:, 0] - 0.5) / 0.5)
