"""Pins the CPU oracle (oracle/pretrain_oracle.py) to fixtures produced by RUNNING the reference
(tools/make_goldens.py, build container only).  No GPU, no /root/reference needed."""
import os

import numpy as np
import pytest
import torch

from oracle import pretrain_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


# ----------------------------------------------------------------------------- masks
def test_tube_masks_bit_exact():
    m = _load("masks.npz")
    for seed in (10, 0, 1, 2, 3):
        np.random.seed(seed)
        got = O.tube_mask((8, 14, 14), 0.9)
        assert got.dtype == np.float64 and got.shape == (1568,)
        assert np.array_equal(got.astype(np.uint8), m[f"tube_s{seed}"])
        assert got.sum() == 1408
    np.random.seed(10)
    tiny = np.stack([O.tube_mask((8, 2, 2), 0.75) for _ in range(2)])
    assert np.array_equal(tiny.astype(np.uint8), m["tube_tiny_s10"])
    np.random.seed(7)
    assert np.array_equal(O.tube_mask((16, 14, 14), 0.9).astype(np.uint8), m["tube_l32_s7"])


def test_tube_mask_seed10_known_answer():
    # the shipped pipeline re-seeds numpy to 10 before every draw.
    np.random.seed(10)
    vis = np.nonzero(O.tube_mask((8, 14, 14), 0.9)[:196] == 0)[0]
    assert vis.tolist() == [1, 5, 9, 11, 27, 41, 86, 89, 94, 113, 115, 120, 133, 153, 168, 180, 184, 186, 193, 195]


def test_bb_masks_bit_exact():
    m = _load("masks.npz")
    boxes = m["bb_boxes"]
    for seed in (10, 0):
        for i, b in enumerate(boxes):
            np.random.seed(seed)
            got = O.bb_mask((8, 14, 14), 0.9, 0.75, np.tile(b, (16, 1)))
            assert np.array_equal(got.astype(np.uint8), m[f"bb_s{seed}"][i]), (seed, i)
            assert got[:196].sum() == 176          # always exactly 176 / frame
            assert np.array_equal(got.reshape(8, 196), np.tile(got[:196], (8, 1)))
    np.random.seed(5)
    stream = np.stack([O.bb_mask((8, 14, 14), 0.9, 0.75, np.tile(b, (16, 1))) for b in boxes])
    assert np.array_equal(stream.astype(np.uint8), m["bb_stream_s5"])


# ----------------------------------------------------------------------------- tables
@pytest.mark.parametrize("n,d", [(1568, 768), (1568, 384), (32, 128), (32, 64), (3136, 1024)])
def test_sincos(n, d):
    g = _load("sincos.npz")
    t = O.sincos_table(n, d)
    assert t.shape == (1, n, d) and t.dtype == torch.float32
    assert np.array_equal(t[0, :4, :8].numpy(), g[f"t{n}x{d}_head"])
    assert np.array_equal(t[0, -4:, -8:].numpy(), g[f"t{n}x{d}_tail"])
    assert np.array_equal(t[0, min(777, n - 1), ::max(1, d // 16)].numpy(), g[f"t{n}x{d}_row777"])
    assert t.double().sum().item() == pytest.approx(float(g[f"t{n}x{d}_sum"]), rel=0, abs=1e-9)


def test_sincos_known_answer():
    t = O.sincos_table(1568, 768)
    np.testing.assert_allclose(t[0, 1, :4].numpy(), [0.84147096, 0.54030228, 0.82843077, 0.56009150], rtol=0, atol=1e-7)


def test_cosine_schedule():
    g = _load("sched.npz")
    assert np.array_equal(O.cosine_schedule(1.5e-4, 1e-5, 10, 7, warmup_epochs=3), g["s1"])
    assert np.array_equal(O.cosine_schedule(0.05, 0.05, 4, 5), g["s2"])
    assert np.array_equal(O.cosine_schedule(1.2e-3, 1e-5, 6, 11, warmup_epochs=2, warmup_steps=9), g["s3"])


def test_ingest_uint8_matches_reference_transforms():
    """fixture = the reference's Stack -> ToTorchFormatTensor(div=True) -> GroupNormalize -> view/transpose on PIL frames"""
    g = _load("ingest.npz")
    frames = torch.from_numpy(g["stacked"])[None]                # [1, H, W, T*3] uint8
    got = O.ingest_uint8(frames)[0]
    assert got.shape == g["clip"].shape
    assert np.array_equal(got.numpy(), g["clip"])                # same fp32 operations: bit-exact


# ----------------------------------------------------------------------------- tiny config, every tensor
@pytest.mark.parametrize("mode", ["small", "xavier"])
def test_tiny_full_parity(mode):
    g = _load(f"tiny_{mode}.npz")
    cfg = O.TINY
    P = O.keyed_params(cfg, mode)
    x = O.keyed_clips(2, cfg)
    mask = torch.from_numpy(_load("masks.npz")["tube_tiny_s10"]).bool()
    assert [k for k in P] == [str(s) for s in g["names"]]
    taps = {}
    with torch.no_grad():
        out = O.model_forward(x, mask, P, cfg, taps)
        taps["patch_embed"] = O.patch_embed(x, P, cfg)
    np.testing.assert_allclose(out.numpy(), g["output"], rtol=1e-4, atol=1e-5)
    assert out.double().sum().item() == pytest.approx(float(g["out_sum"]), rel=1e-5, abs=1e-4)
    for k, v in taps.items():
        np.testing.assert_allclose(v.numpy(), g["tap_" + k], rtol=1e-4, atol=1e-5, err_msg=k)
    if mode == "small":   # SURVEY.md §8c known answers
        assert out.double().sum().item() == pytest.approx(-5.921754358714679, abs=2e-4)
        np.testing.assert_allclose(out[0, 0, :3].numpy(), [0.19170940, 0.14276052, 0.03188009], atol=1e-6)
    np.testing.assert_array_equal(O.build_targets(x, mask, cfg).numpy(), g["labels"])

    st = O.AdamWState()
    losses, norms = [], []
    for s in range(3):
        loss, gn, grads = O.train_step(x, mask, P, cfg, st)
        losses.append(loss)
        norms.append(gn)
        if s == 0:
            for i, name in enumerate(P):
                l2 = float(torch.norm(grads[name].double()))
                assert l2 == pytest.approx(g["grad_stats"][i, 0], rel=2e-4, abs=1e-9), name
                n = min(16, grads[name].numel())
                np.testing.assert_allclose(grads[name].reshape(-1)[:n].numpy(), g["grad_head"][i, :n],
                                           rtol=2e-3, atol=1e-7, err_msg=name)
                if "grad_" + name in g.files:
                    np.testing.assert_allclose(grads[name].numpy(), g["grad_" + name], rtol=2e-3, atol=1e-7, err_msg=name)
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-5)
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=1e-4)
    for i, name in enumerate(P):   # parameters after three reference AdamW steps
        assert float(torch.norm(P[name].double())) == pytest.approx(g["param_stats_after3"][i, 0], rel=1e-5), name
        n = min(16, P[name].numel())
        np.testing.assert_allclose(P[name].reshape(-1)[:n].numpy(), g["param_head_after3"][i, :n], rtol=1e-4, atol=1e-7)
    decay = sum(0 if O.is_no_decay(k, v.shape) else 1 for k, v in P.items())
    assert sorted(g["group_sizes"].tolist()) == sorted([decay, len(P) - decay])


def test_library_ops_form_is_the_same_arithmetic(monkeypatch):
    """O.LIBRARY_OPS (tests/eager_gpu_baseline.py) routes the forward through F.linear / F.layer_norm / F.gelu / softmax: the fixture's output
    and first-step loss / gradient norm to the tolerances the elementary form is held to"""
    g = _load("tiny_small.npz")
    cfg = O.TINY
    P = O.keyed_params(cfg, "small")
    x = O.keyed_clips(2, cfg)
    mask = torch.from_numpy(_load("masks.npz")["tube_tiny_s10"]).bool()
    monkeypatch.setattr(O, "LIBRARY_OPS", True)
    with torch.no_grad():
        out = O.model_forward(x, mask, P, cfg)
    np.testing.assert_allclose(out.numpy(), g["output"], rtol=1e-4, atol=1e-5)
    loss, gn, _ = O.train_step(x, mask, P, cfg)
    assert loss == pytest.approx(float(g["losses"][0]), rel=1e-5) and gn == pytest.approx(float(g["grad_norms"][0]), rel=1e-4)


# ----------------------------------------------------------------------------- BASELINE config[0]: the engine itself
def test_vitb_engine_step_parity():
    """Fixture = the reference's own train_one_epoch, one step, ViT-B, B=2, tube masks (seeds 10 and 0)."""
    g = _load("engine_vitb.npz")
    m = _load("masks.npz")
    cfg = O.VIT_B
    P = O.keyed_params(cfg, "xavier")
    assert sum(v.numel() for v in P.values()) == 94_210_944
    x = O.keyed_clips(2, cfg)
    mask = torch.from_numpy(np.stack([m["tube_s10"], m["tube_s0"]])).bool()
    labels = O.build_targets(x, mask, cfg)
    np.testing.assert_allclose(labels[:, :6, :48].numpy(), g["labels_slice"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(labels[:, -3:, -24:].numpy(), g["labels_tail"], rtol=1e-5, atol=1e-6)
    assert (labels.double() ** 2).sum().item() == pytest.approx(float(g["labels_sqsum"]), rel=1e-6)
    st = O.AdamWState()
    loss, gn, grads = O.train_step(x, mask, P, cfg, st, lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
    assert loss == pytest.approx(float(g["loss"]), rel=1e-5)
    assert gn == pytest.approx(float(g["grad_norm"]), rel=1e-4)
    for i, name in enumerate(P):
        assert float(torch.norm(grads[name].double())) == pytest.approx(g["grad_stats"][i, 0], rel=1e-3, abs=1e-9), name
    loss2, _, _ = O.train_step(x, mask, P, cfg, st)   # second step sees the reference's post-step weights?
    # the fixture's follow-up steps used lr=1.5e-4, wd=0.05 (create_optimizer defaults) after step 1
    assert loss2 == pytest.approx(float(g["losses_after"][0]), rel=2e-5)


def test_vitb_bb_masks_parity():
    g = _load("vitb_bb.npz")
    m = _load("masks.npz")
    cfg = O.VIT_B
    P = O.keyed_params(cfg, "xavier")
    x = O.keyed_clips(2, cfg)
    mask = torch.from_numpy(m["bb_s10"][[0, 3]]).bool()
    with torch.no_grad():
        out = O.model_forward(x, mask, P, cfg)
        loss = O.mse_loss(out, O.build_targets(x, mask, cfg))
    np.testing.assert_allclose(out[:, :6, :48].numpy(), g["out_slice"], rtol=1e-3, atol=1e-5)
    assert float(loss) == pytest.approx(float(g["loss"]), rel=1e-5)


def test_vitb_bb_engine_step_matches_reference_loop():
    """one step of the reference's OWN motion-box epoch loop (train_one_epoch_BB, tests/golden/engine_vitb_bb.npz): the
    oracle's loss, gradient norm and per-tensor gradient norms"""
    g = _load("engine_vitb_bb.npz")
    m = _load("masks.npz")
    cfg = O.VIT_B
    P = O.keyed_params(cfg, "xavier")
    x = O.keyed_clips(2, cfg)
    mask = torch.from_numpy(m["bb_s10"][g["pick"]]).bool()
    loss, gn, grads = O.train_step(x, mask, P, cfg)
    assert loss == pytest.approx(float(g["loss"]), rel=1e-5) and gn == pytest.approx(float(g["grad_norm"]), rel=1e-4)
    for i, n in enumerate(str(s) for s in g["names"]):
        assert float(grads[n].double().norm()) == pytest.approx(g["grad_stats"][i, 0], rel=2e-3, abs=1e-7), n


# ----------------------------------------------------------------------------- "next" rows
def _vis_inputs():
    g = _load("vis.npz")
    cfg = O.VIT_B
    img = O.keyed_clips(1, cfg, base_seed=2000)
    np.random.seed(10)
    m = O.tube_mask(cfg.grid, 0.9)
    assert np.array_equal(m, g["mask"])
    mask = torch.from_numpy(m)[None].bool()
    outputs = torch.from_numpy(np.random.RandomState(77).standard_normal((1, int(mask.sum()), cfg.patch_dim)).astype(np.float32))
    return g, cfg, img, mask, outputs


def test_reconstruct_video_matches_reference_vis_arithmetic():
    g, cfg, img, mask, outputs = _vis_inputs()
    ori, rec, masked = O.reconstruct_video(img, mask, outputs, cfg)
    for name, t in (("ori_img", ori), ("rec_img", rec), ("img_mask", masked)):
        np.testing.assert_allclose(t[0, :, :2, :48, :48].numpy(), g[name + "_head"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(t[0, :, -2:, -48:, -48:].numpy(), g[name + "_tail"], rtol=1e-5, atol=1e-6)
        assert t.double().sum().item() == pytest.approx(float(g[name + "_sum"]), rel=1e-6)
        assert (t.double() ** 2).sum().item() == pytest.approx(float(g[name + "_sqsum"]), rel=1e-6)
        np.testing.assert_allclose(t.double().sum(dim=(0, 1, 3, 4)).numpy(), g[name + "_framesum"], rtol=1e-6)
    # visible tokens come back as the original pixels, masked ones are zero in the masked video
    keep = (masked != 0)
    assert torch.allclose(rec[keep], ori[keep], atol=1e-5)


@pytest.mark.parametrize("tag,ncls,nb", [("tiny", 10, 2), ("vitb", 400, 1)])
def test_finetune_forward_matches_reference(tag, ncls, nb):
    g = _load(f"finetune_{tag}.npz")
    cfg = O.TINY if tag == "tiny" else O.VIT_B
    P = O.finetune_keyed_params(cfg, ncls)
    x = O.keyed_clips(nb, cfg, base_seed=3000)
    with torch.no_grad():
        feat = O.finetune_forward(x, P, cfg, features_only=True)
        logits = O.finetune_forward(x, P, cfg)
    np.testing.assert_allclose(feat.numpy(), g["features"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=2e-4, atol=2e-5)


def test_clipped_steps_match_reference_scaler():
    """three steps with clip_grad=0.1 (active: the norm is 0.126): losses, norms reported before clipping and parameters
    afterwards against the reference's scaler + clip_grad_norm_ + create_optimizer (tests/golden/tiny_clip.npz)"""
    g = _load("tiny_clip.npz")
    cfg = O.TINY
    P = O.keyed_params(cfg, "xavier")
    x = O.keyed_clips(2, cfg)
    mask = torch.from_numpy(_load("masks.npz")["tube_tiny_s10"]).bool()
    st = O.AdamWState()
    out = [O.train_step(x, mask, P, cfg, st, clip_grad=float(g["clip_grad"]))[:2] for _ in range(3)]
    np.testing.assert_allclose([o[0] for o in out], g["losses"], rtol=1e-5)
    np.testing.assert_allclose([o[1] for o in out], g["norms"], rtol=1e-4)
    assert all(n > float(g["clip_grad"]) for n in g["norms"])          # the clip really bites
    for i, n in enumerate(str(s) for s in g["names"]):
        assert float(P[n].double().norm()) == pytest.approx(g["param_stats_after3"][i, 0], rel=1e-5, abs=1e-7), n


def test_vit_large_32_frames_matches_reference_classes():
    """BASELINE config 4's shapes (ViT-L widths, 32 frames -> 3136 tokens, 320 visible; 3 + 1 blocks): the oracle against the
    reference classes with their two position tables rebuilt for 32 frames (tests/golden/vitl32.npz)"""
    g = _load("vitl32.npz")
    cfg = O.OracleConfig(num_frames=32, enc_dim=1024, enc_depth=3, enc_heads=16, dec_dim=512, dec_depth=1, dec_heads=8)
    P = O.keyed_params(cfg, "xavier")
    x = O.keyed_clips(1, cfg)
    mask = torch.from_numpy(g["mask"]).bool()
    np.random.seed(7)
    assert np.array_equal(O.tube_mask(cfg.grid, 0.9)[None].astype(np.uint8), g["mask"])
    loss, gn, grads = O.train_step(x, mask, P, cfg)
    assert loss == pytest.approx(float(g["loss"]), rel=1e-5) and gn == pytest.approx(float(g["grad_norm"]), rel=1e-4)
    for i, n in enumerate(str(s) for s in g["names"]):
        assert float(grads[n].double().norm()) == pytest.approx(g["grad_stats"][i, 0], rel=2e-3, abs=1e-7), n
