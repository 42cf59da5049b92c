"""Host-side logic on CPU (no GPU, no /root/reference): drop-in API surface, mask generators and schedules against
the reference fixtures, flat parameter store layout, gradient-bucket plan, and the data-parallel gradient exchange on
two gloo ranks."""
import os
import socket
from functools import partial

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


# ------------------------------------------------------------------------------------------------ masks / schedules
def test_mask_generators_match_reference_fixtures():
    from mofo_amd.masking_generator import TubeMaskingGenerator, TubeMaskingGenerator_BB
    m = np.load(os.path.join(G, "masks.npz"))
    for seed in (10, 0, 1, 2, 3):
        np.random.seed(seed)
        got = TubeMaskingGenerator((8, 14, 14), 0.9)()
        assert got.dtype == np.float64 and np.array_equal(got.astype(np.uint8), m[f"tube_s{seed}"])
    np.random.seed(7)
    assert np.array_equal(TubeMaskingGenerator((16, 14, 14), 0.9)().astype(np.uint8), m["tube_l32_s7"])
    for seed in (10, 0):
        for i, b in enumerate(m["bb_boxes"]):
            np.random.seed(seed)
            got = TubeMaskingGenerator_BB((8, 14, 14), 0.9, 0.75)(np.tile(b, (16, 1)))
            assert np.array_equal(got.astype(np.uint8), m[f"bb_s{seed}"][i])
            assert got.sum() == 1408
    np.random.seed(5)
    gen = TubeMaskingGenerator_BB((8, 14, 14), 0.9, 0.75)
    stream = np.stack([gen(np.tile(b, (16, 1))) for b in m["bb_boxes"]])
    assert np.array_equal(stream.astype(np.uint8), m["bb_stream_s5"])
    assert "total patches 1568, mask patches 1408" in repr(gen)


def test_device_tube_mask_generator_distribution():
    """the oracle's restatement of the device-side tube-mask generator has the reference generator's
    structure and draws every patch with the same frequency (the GPU test compares the kernel with it bit for bit)"""
    from oracle import pretrain_oracle as O
    from mofo_amd.masking_generator import TubeMaskingGenerator
    host = TubeMaskingGenerator((8, 14, 14), 0.9)
    m = O.device_tube_masks(5, 0, 6, 8, 196, host.num_masks_per_frame)
    ref = host()
    assert m.shape == (6, ref.size) and (m.sum(1) == ref.sum()).all()
    assert (m.reshape(6, 8, 196) == m.reshape(6, 8, 196)[:, :1]).all()
    many = O.device_tube_masks(5, 0, 4000, 1, 196, 176).astype(np.float64)
    freq = many.mean(0)                                   # each patch masked with probability 176 / 196
    assert abs(freq.mean() - 176 / 196) < 1e-12
    assert np.abs(freq - 176 / 196).max() < 5 * np.sqrt((176 / 196) * (20 / 196) / 4000)
    pair = (many[:, :98] * many[:, 98:]).mean(0)          # pairs: P(both masked) = 176 * 175 / (196 * 195)
    assert abs(pair.mean() - 176 * 175 / (196 * 195)) < 2e-3
    assert np.array_equal(O.device_tube_masks(5, 3, 2, 8, 196, 176), O.device_tube_masks(5, 0, 5, 8, 196, 176)[3:])


def test_bb_mask_edge_cases():
    from mofo_amd.masking_generator import TubeMaskingGenerator_BB
    gen = TubeMaskingGenerator_BB((8, 14, 14), 0.9, 0.75)
    for box in ([0, 0, 0, 0], [223, 223, 224, 224], [0, 0, 224, 224], [-5, -5, -1, -1], [300, 300, 400, 400]):
        np.random.seed(1)
        msk = gen(np.tile(np.array(box), (16, 1)))
        assert msk.shape == (1568,) and msk[:196].sum() == 176      # always exactly 176 masked per frame -> 160 visible
        assert np.array_equal(msk.reshape(8, 196), np.tile(msk[:196], (8, 1)))


def test_cosine_scheduler_and_sincos():
    from mofo_amd import utils
    from mofo_amd.runtime import sincos_table
    g = np.load(os.path.join(G, "sched.npz"))
    assert np.array_equal(utils.cosine_scheduler(1.5e-4, 1e-5, 10, 7, warmup_epochs=3), g["s1"])
    assert np.array_equal(utils.cosine_scheduler(0.05, 0.05, 4, 5), g["s2"])
    assert np.array_equal(utils.cosine_scheduler(1.2e-3, 1e-5, 6, 11, warmup_epochs=2, warmup_steps=9), g["s3"])
    s = np.load(os.path.join(G, "sincos.npz"))
    for n, d in ((1568, 768), (1568, 384), (3136, 1024)):
        t = sincos_table(n, d)
        assert np.array_equal(t[:4, :8].numpy(), s[f"t{n}x{d}_head"]) and np.array_equal(t[-4:, -8:].numpy(), s[f"t{n}x{d}_tail"])


# ------------------------------------------------------------------------------------------------ API surface
def test_state_dict_schema_and_factories():
    from mofo_amd import modeling_pretrain as mp
    from oracle import pretrain_oracle as O
    m = mp.pretrain_videomae_base_patch16_224(decoder_depth=4)
    sd = m.state_dict()
    shapes = O.param_shapes(O.VIT_B)
    assert list(sd) == list(shapes) and all(tuple(sd[k].shape) == shapes[k] for k in sd)
    assert sum(v.numel() for v in sd.values()) == 94_210_944
    assert m.encoder.patch_embed.patch_size == (16, 16) and m.encoder.patch_embed.num_patches == 1568
    assert m.no_weight_decay() == {'pos_embed', 'cls_token', 'mask_token'}
    assert float(m.mask_token.abs().max()) <= 0.02 + 1e-9           # trunc_normal_(std=.02) clipped at +-std
    for blk in m.encoder.blocks[:1]:
        assert torch.all(blk.attn.q_bias == 0) and torch.all(blk.norm1.weight == 1) and torch.all(blk.mlp.fc1.bias == 0)
    mm = mp.create_model('pretrain_videomae_base_patch16_224', pretrained=False, drop_path_rate=0.0, drop_block_rate=None, decoder_depth=4)
    assert len(mm.decoder.blocks) == 4
    small = mp.pretrain_mae_small_patch16_224(decoder_depth=2)
    assert small.encoder.embed_dim == 384 and small.decoder.embed_dim == 192
    # a reference checkpoint's 'model' dict loads by name
    ref_like = {k: torch.randn(s) for k, s in shapes.items()}
    assert not m.load_state_dict(ref_like, strict=True).missing_keys


def test_unsupported_options_raise_loudly():
    from mofo_amd import modeling_pretrain as mp
    for kw in (dict(drop_path_rate=0.1), dict(drop_rate=0.1), dict(attn_drop_rate=0.1), dict(init_values=0.1),
               dict(use_learnable_pos_emb=True), dict(qk_scale=0.5)):
        with pytest.raises(NotImplementedError):
            mp.pretrain_videomae_base_patch16_224(decoder_depth=1, **kw)
    with pytest.raises(NotImplementedError):
        mp.PretrainVisionTransformer(qkv_bias=False)
    with pytest.raises(NotImplementedError, match="head_dim"):
        mp.PretrainVisionTransformerEncoder(embed_dim=768, num_heads=8, qkv_bias=True)
    m = mp.pretrain_videomae_base_patch16_224(decoder_depth=1)
    with pytest.raises(RuntimeError, match="GPU only"):
        m(torch.zeros(1, 3, 16, 224, 224), torch.zeros(1, 1568, dtype=torch.bool))


def test_optimizer_groups_and_errors():
    from mofo_amd import modeling_pretrain as mp
    from mofo_amd import optim_factory
    m = mp.pretrain_videomae_base_patch16_224(decoder_depth=4)
    groups, names = optim_factory.get_parameter_groups(m, 0.05, m.no_weight_decay())
    sizes = {k: len(v) for k, v in names.items()}
    assert sizes == {"no_decay": 151, "decay": 67}
    assert list(names)[0] == "no_decay" and "mask_token" in names["no_decay"] and "encoder.patch_embed.proj.weight" in names["decay"]
    assert all("attn.q_bias" not in n for n in names["decay"])

    class A:
        opt, lr, weight_decay, opt_eps, opt_betas = "sgd", 1e-3, 0.05, 1e-8, (0.9, 0.95)
    with pytest.raises(NotImplementedError):
        optim_factory.create_optimizer(A, m)


# ------------------------------------------------------------------------------------------------ flat store
def _tiny_model():
    from mofo_amd import modeling_pretrain as mp
    return mp.PretrainVisionTransformer(img_size=32, encoder_embed_dim=128, encoder_depth=5, encoder_num_heads=2, decoder_embed_dim=64,
                                        decoder_depth=1, decoder_num_heads=1, decoder_num_classes=1536, qkv_bias=True,
                                        norm_layer=partial(torch.nn.LayerNorm, eps=1e-6))


def _cpu_runtime(model):
    from mofo_amd.runtime import FlatStore, PretrainRuntime
    named = dict(model.named_parameters())
    store = FlatStore([(n, named[n]) for n in model._flat_order()], torch.device("cpu"), skip_decay=model.no_weight_decay())
    return PretrainRuntime(model._make_runtime.__func__(model, store).d, store), store


def test_flat_store_layout_and_views():
    model = _tiny_model()
    before = {k: v.clone() for k, v in model.state_dict().items()}
    rt, st = _cpu_runtime(model)
    assert st.total % 1024 == 0 and st.owns(full=True) and st.grads_attached()
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k])                               # values survived the move into the flat buffer
    # parameters ARE views: writing the flat buffer changes the module, load_state_dict writes the flat buffer
    st.params.zero_()
    assert all(float(p.abs().sum()) == 0 for p in model.parameters())
    model.load_state_dict(before)
    assert torch.equal(st.view("decoder.head.weight"), before["decoder.head.weight"])
    # fused qkv bias = (q_bias | zeros | v_bias), zero third is not a parameter
    fb = st.fused_bias("encoder.blocks.0.attn.q_bias")
    assert fb.numel() == 384 and torch.equal(fb[:128], before["encoder.blocks.0.attn.q_bias"]) and torch.all(fb[128:256] == 0)
    # decay flags: chunk_group 0 = decayed (>=2-D weights), 1 = everything else incl. padding
    cg = st.chunk_group.numpy()
    o = st.offset["encoder.blocks.0.mlp.fc1.weight"] // 1024
    assert cg[o] == 0 and cg[st.offset["encoder.blocks.0.mlp.fc1.bias"] // 1024] == 1 and cg[st.offset["mask_token"] // 1024] == 1
    decayed = sum(int(np.prod(s)) for n, s in st.shape.items() if len(s) > 1 and n != "mask_token")
    assert (cg == 0).sum() * 1024 >= decayed
    # version tracking: in-place write through a Parameter is seen
    v0 = st._version()
    with torch.no_grad():
        model.mask_token.add_(1.0)
    assert st._version() != v0
    # zero_grad(set_to_none) detaches; re-attach restores the views
    for p in model.parameters():
        p.grad = None
    assert not st.grads_attached()
    st.attach_grads()
    assert st.grads_attached()


def test_weight_gradient_group_size_by_shape():
    """encoder blocks per grouped weight-gradient launch = the smallest group that fills whole rounds of the 768 resident 128 x 128
    tiles best: ViT-B (432 tiles per block) takes three blocks per launch, ViT-L (768 = one round exactly) one"""
    from mofo_amd.runtime import _pick_wgrad_blocks
    assert _pick_wgrad_blocks(768, 3072) == 3
    assert _pick_wgrad_blocks(1024, 4096) == 2
    assert _pick_wgrad_blocks(128, 512) in (1, 2, 3)
    # one process (no gradient buckets to hand over): the 256 x 128 ring kernel takes up to seven blocks per launch and ViT-B's 216
    # units per block fill 5.9 rounds of 256 with seven (84 % with 1 ... 6); ViT-L's 384 per block already fill whole rounds
    assert _pick_wgrad_blocks(768, 3072, 12, bucketed=False) == 7
    assert _pick_wgrad_blocks(1024, 4096, 24, bucketed=False) == 2     # 768 units = three rounds exactly
    assert _pick_wgrad_blocks(768, 3072, 12, bucketed=True) == 3
    assert _pick_wgrad_blocks(1024, 4096, 24, bucketed=True) == 2


def test_gradient_segments_tile_the_buffer():
    model = _tiny_model()
    rt, st = _cpu_runtime(model)
    segs = rt.segments
    assert rt._enc_buckets(12) == [5, 3, 2, 1, 1] and rt._enc_buckets(5) == [2, 2, 1]   # shrinking: the exposed one is the smallest
    assert rt._enc_buckets(12, 3) == [6, 3, 2, 1] and rt._enc_buckets(5, 3) == [2, 2, 1] and rt._enc_buckets(24, 1) == rt._enc_buckets(24)
    assert rt._enc_buckets(24, 3) == [12, 6, 3, 2, 1] and all(sum(rt._enc_buckets(n, g)) == n for n in range(1, 40) for g in (1, 2, 3))
    assert len(segs) == 4                                             # decoder group + 3 encoder groups (5 blocks -> 2 + 2 + 1)
    ranges = sorted(segs)
    assert ranges[0][0] == 0 and ranges[-1][1] == st.total
    for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
        assert a1 == b0                                                # contiguous, no gap, no overlap
    # completion order = decoder first, then encoder from the top
    assert segs[0][1] == st.total and segs[-1][0] == 0


# ------------------------------------------------------------------------------------------------ DP on two gloo ranks
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, out):
    import torch.distributed as dist
    from mofo_amd.dist import GradSync
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                                      # different init per rank on purpose
    model = _tiny_model()
    rt, st = _cpu_runtime(model)
    model.runtime = lambda: rt
    dist.broadcast(st.params, src=0)                                   # what DataParallel.__init__ does
    sync = GradSync(model).install()
    assert sync.enabled
    g = torch.Generator().manual_seed(7 + rank)
    local = torch.randn(st.total, generator=g)
    st.grads.copy_(local / world)                                      # loss kernel pre-scales d(loss) by 1/world
    for idx in range(len(rt.segments)):                                # backward completes segment after segment
        rt._seg_now(idx)
    launched = list(sync.launched)
    sync.finish()
    torch.save({"params": st.params.clone(), "grads": st.grads.clone(), "local": local, "launched": launched}, out + f".{rank}")
    dist.destroy_process_group()


def test_data_parallel_gradient_mean_two_ranks(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    out = str(tmp_path / "dp")
    mp.spawn(_dp_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r = [torch.load(out + f".{i}") for i in range(world)]
    assert torch.equal(r[0]["params"], r[1]["params"])                 # replicas start identical (rank 0's weights)
    mean = (r[0]["local"] + r[1]["local"]) / world
    for i in range(world):
        assert torch.allclose(r[i]["grads"], mean, rtol=1e-6, atol=1e-7)   # SUM of pre-scaled grads == DDP mean
    assert torch.equal(r[0]["grads"], r[1]["grads"])                   # bit-identical on every rank
    assert [x[0] for x in r[0]["launched"]] == [0, 1, 2, 3]            # one all-reduce per segment (decoder + 2, 2, 1 encoder blocks), in completion order
    cover = sorted((lo, hi) for _, lo, hi in r[0]["launched"])
    assert cover[0][0] == 0 and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))


def _dp_plan_worker(rank, world, port, out):
    """two gloo ranks switch the encoder's bucket / group plan after construction (runtime.set_enc_plan, what bench.py's data-parallel
    A/B does): the new ranges tile the buffer, the exchange follows them, every rank ends with the same plan and the same mean"""
    import torch.distributed as dist
    from mofo_amd.dist import GradSync
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(3)
    model = _tiny_model()                                              # encoder depth 5
    rt, st = _cpu_runtime(model)                                       # built inside a 2-rank group: bucketed, scratch cap for large groups
    model.runtime = lambda: rt
    sync = GradSync(model).install()
    rec = {"cap": rt._enc_group_cap, "default": (rt.enc_buckets(), rt.wgrad_blocks, list(rt.segments))}
    rt.set_enc_plan([3, 2], 3)
    rec["switched"] = (rt.enc_buckets(), rt.wgrad_blocks, list(rt.segments))
    g = torch.Generator().manual_seed(7 + rank)
    local = torch.randn(st.total, generator=g)
    st.grads.copy_(local / world)
    for idx in range(len(rt.segments)):
        rt._seg_now(idx)
    rec["launched"] = list(sync.launched)
    sync.finish()
    rec["grads"], rec["local"] = st.grads.clone(), local
    try:
        rt.set_enc_plan([4, 2], 3)                                     # does not sum to the depth
        rec["bad_plan"] = "accepted"
    except ValueError as exc:
        rec["bad_plan"] = str(exc)
    rt.set_enc_plan(None, 2)
    rec["back"] = (rt.enc_buckets(), rt.wgrad_blocks)
    torch.save(rec, out + f".{rank}")
    dist.destroy_process_group()


def test_enc_plan_switch_two_ranks(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    out = str(tmp_path / "plan")
    mp.spawn(_dp_plan_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r = [torch.load(out + f".{i}") for i in range(world)]
    assert r[0]["cap"] == 5 and r[0]["default"][:2] == r[1]["default"][:2]
    for k in ("default", "switched", "back", "launched", "bad_plan"):
        assert r[0][k] == r[1][k], k                                   # every rank: the same plan, the same ranges, in the same order
    buckets, blocks, segs = r[0]["switched"]
    assert buckets == [3, 2] and blocks == 3 and len(segs) == 3        # decoder + two encoder buckets
    cover = sorted(segs)
    assert cover[0][0] == 0 and all(a[1] == b[0] for a, b in zip(cover, cover[1:])) and cover[-1][1] == r[0]["grads"].numel()
    assert [x[0] for x in r[0]["launched"]] == [0, 1, 2] and [(lo, hi) for _, lo, hi in r[0]["launched"]] == segs
    mean = (r[0]["local"] + r[1]["local"]) / world
    assert torch.allclose(r[0]["grads"], mean, rtol=1e-6, atol=1e-7) and torch.equal(r[0]["grads"], r[1]["grads"])
    assert "sum to the encoder depth" in r[0]["bad_plan"] and r[0]["back"][1] == 2


def _dp_check_worker(rank, world, port, out):
    """GradSync.value_check on two gloo ranks: the production order passes; a backward that finishes a range AFTER handing it to
    the all-reduce (what a too-early side-stream hand-off would amount to) is caught"""
    import torch.distributed as dist
    from mofo_amd.dist import GradSync
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(3)
    model = _tiny_model()
    rt, st = _cpu_runtime(model)
    model.runtime = lambda: rt
    sync = GradSync(model).install()
    g = torch.Generator().manual_seed(7 + rank)
    local = torch.randn(st.total, generator=g)
    lo1, hi1 = rt.segments[1]

    def good():
        st.grads.copy_(local / world)
        for idx in range(len(rt.segments)):
            rt._seg_now(idx)

    def early():                       # range 1 is handed over while its "last weight gradient" is still missing
        st.grads.copy_(local / world)
        st.grads[lo1:hi1].zero_()
        for idx in range(len(rt.segments)):
            rt._seg_now(idx)
        for h in sync.handles:
            h.wait()
        st.grads[lo1:hi1] += local[lo1:hi1] / world

    ok = sync.value_check(good)
    bad = sync.value_check(early)
    torch.save({"ok": ok, "bad": bad, "seg1": (lo1, hi1)}, out + f".{rank}")
    dist.destroy_process_group()


def test_allreduce_value_check_two_ranks(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    out = str(tmp_path / "chk")
    mp.spawn(_dp_check_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for i in range(world):
        r = torch.load(out + f".{i}")
        assert r["ok"]["ok"] and r["ok"]["max_rel"] < 1e-6 and r["ok"]["ranges"] == 4
        assert not r["bad"]["ok"] and r["bad"]["max_rel"] > 0.1
        assert tuple(r["bad"]["worst_range"][1:]) == tuple(r["seg1"])          # ... and names the range that was early


def _dp8_worker(rank, world, port, out):
    """eight gloo ranks (the node the multi-GPU records are taken on): the bucket plan, one all-reduce per bucket in completion
    order, the mean on every rank, and the self-check of the exchange -- with f32 and with bf16 gradient transport"""
    import torch.distributed as dist
    from mofo_amd.dist import GradSync
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    torch.manual_seed(3)
    model = _tiny_model()
    rt, st = _cpu_runtime(model)
    model.runtime = lambda: rt
    g = torch.Generator().manual_seed(7 + rank)
    local = torch.randn(st.total, generator=g)
    res = {}
    for name, env in (("f32", "0"), ("bf16", "1")):
        os.environ["MOFO_GRAD_BF16"] = env
        sync = GradSync(model, narrow=lambda a, b: b.copy_(a), widen=lambda a, b: b.copy_(a)).install()
        assert sync.enabled and sync.world_size == world and sync.bf16 == (env == "1")

        def backward():
            st.grads.copy_(local / world)
            for idx in range(len(rt.segments)):
                rt._seg_now(idx)

        backward()
        launched = list(sync.launched)
        sync.finish()
        res[name] = {"grads": st.grads.clone(), "launched": launched,
                     "check": sync.value_check(backward, tol=1e-4 if env == "0" else 2e-2)}
    res["local"] = local
    torch.save(res, out + f".{rank}")
    os.environ.pop("MOFO_GRAD_BF16", None)
    dist.destroy_process_group()


def test_data_parallel_eight_ranks_f32_and_bf16_transport(tmp_path):
    import torch.multiprocessing as mp
    world = 8
    out = str(tmp_path / "dp8")
    mp.spawn(_dp8_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r = [torch.load(out + f".{i}") for i in range(world)]
    mean = sum(x["local"] for x in r) / world
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    for i in range(world):
        assert torch.allclose(r[i]["f32"]["grads"], mean, rtol=1e-5, atol=1e-6)      # SUM of pre-scaled gradients == DDP's mean
        assert torch.equal(r[i]["f32"]["grads"], r[0]["f32"]["grads"])               # the same bits on every rank
        assert rel(r[i]["bf16"]["grads"], mean) <= 1e-2                              # bf16 on the wire: within 1e-2 of the f32 transport
        assert torch.equal(r[i]["bf16"]["grads"], r[0]["bf16"]["grads"])
        for kind in ("f32", "bf16"):
            assert [x[0] for x in r[i][kind]["launched"]] == [0, 1, 2, 3]            # one exchange per bucket, in completion order
            cover = sorted((lo, hi) for _, lo, hi in r[i][kind]["launched"])
            assert cover[0][0] == 0 and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
            assert r[i][kind]["check"]["ok"] and r[i][kind]["check"]["ranges"] == 4
        assert r[i]["f32"]["check"]["max_rel"] < 1e-6 and 0 < r[i]["bf16"]["check"]["max_rel"] <= 2e-2


def _dp_raise_worker(rank, world, port, out):
    """value_check when ONE rank's backward raises: every rank gets the error result, nobody is left inside a collective"""
    import torch.distributed as dist
    from mofo_amd.dist import GradSync
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _tiny_model()
    rt, st = _cpu_runtime(model)
    model.runtime = lambda: rt
    sync = GradSync(model).install()

    def backward():
        if rank == 1:
            raise RuntimeError("rank 1 is out of memory")
        st.grads.fill_(1.0)
        for idx in range(len(rt.segments)):
            rt._seg_now(idx)

    res = sync.value_check(backward)
    t = torch.ones(1)
    dist.all_reduce(t)                          # the group is still in step: the next collective completes on both ranks
    torch.save({"res": res, "sum": float(t.item())}, out + f".{rank}")
    dist.destroy_process_group()


def test_value_check_survives_a_rank_that_raises(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "raise")
    mp.spawn(_dp_raise_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    for i in range(2):
        r = torch.load(out + f".{i}")
        assert not r["res"]["ok"] and "error" in r["res"] and r["sum"] == 2.0
    assert "out of memory" in torch.load(out + ".1")["res"]["error"]


def test_device_mask_generator_rank_defaults_and_state(monkeypatch):
    """rank / world of the device-side mask generator default to the launcher's environment (utils.py:277-296 reads the same
    variables), and its position in the mask stream is checkpointable"""
    from mofo_amd.masking_generator import DeviceTubeMaskingGenerator
    monkeypatch.setenv("RANK", "3")
    monkeypatch.setenv("WORLD_SIZE", "8")
    g = DeviceTubeMaskingGenerator((8, 14, 14), 0.9, seed=5)
    assert (g.rank, g.world_size) == (3, 8)
    g2 = DeviceTubeMaskingGenerator((8, 14, 14), 0.9, seed=5, rank=0, world_size=1)
    assert (g2.rank, g2.world_size) == (0, 1)
    monkeypatch.delenv("RANK")
    monkeypatch.delenv("WORLD_SIZE")
    g3 = DeviceTubeMaskingGenerator((8, 14, 14), 0.9, seed=5)
    assert (g3.rank, g3.world_size) == (0, 1)
    g3.clips_drawn = 96
    g4 = DeviceTubeMaskingGenerator((8, 14, 14), 0.9, seed=0)
    g4.load_state_dict(g3.state_dict())
    assert (g4.seed, g4.clips_drawn) == (5, 96)


# ------------------------------------------------------------------------------------------------ "next" rows
def test_finetune_model_schema_and_checkpoint_mapping():
    """state_dict schema of the fine-tune model = the reference's (oracle.finetune_param_shapes, pinned by finetune_*.npz), a
    pretraining checkpoint maps onto it the way run_class_finetuning.py:362-411 does, and the forward refuses the CPU"""
    from mofo_amd import modeling_finetune as ft
    from mofo_amd import modeling_pretrain as mp
    from oracle import pretrain_oracle as O
    m = ft.vit_base_patch16_224(num_classes=400, all_frames=16, tubelet_size=2)
    shapes = O.finetune_param_shapes(O.VIT_B, 400)
    sd = m.state_dict()
    assert list(sd) == list(shapes) and all(tuple(sd[k].shape) == shapes[k] for k in sd)
    assert isinstance(m.norm, torch.nn.Identity) and m.get_num_layers() == 12 and m.patch_embed.num_patches == 1568
    pre = mp.pretrain_videomae_base_patch16_224(decoder_depth=1)
    r = m.load_pretrained_encoder(pre.state_dict())
    assert sorted(r.missing_keys) == ["fc_norm.bias", "fc_norm.weight", "head.bias", "head.weight"] and not r.unexpected_keys
    assert torch.equal(m.state_dict()["blocks.3.attn.qkv.weight"], pre.state_dict()["encoder.blocks.3.attn.qkv.weight"])
    with pytest.raises(RuntimeError, match="GPU only"):
        m(torch.zeros(1, 3, 16, 224, 224))
    for kw in (dict(use_mean_pooling=False), dict(init_values=0.1), dict(use_learnable_pos_emb=True)):
        with pytest.raises(NotImplementedError):
            ft.vit_base_patch16_224(num_classes=400, **kw)
    feat = ft.vit_base_patch16_224_feature_ext(num_classes=400)
    assert type(feat).__name__ == "VisionTransformer_feat_ext"


def test_mask_generator_state_travels_with_the_checkpoint(tmp_path):
    """a run that draws its masks on the device (masking_generator.DeviceTubeMaskingGenerator, args.mask_generator) resumes the mask
    stream where the checkpoint left it; a resume under another world size says so"""
    import types
    import warnings
    from functools import partial
    from mofo_amd import modeling_pretrain as mp, utils
    from mofo_amd.masking_generator import DeviceTubeMaskingGenerator
    model = mp.PretrainVisionTransformer(img_size=32, encoder_embed_dim=128, encoder_depth=2, encoder_num_heads=2, decoder_embed_dim=64,
                                         decoder_depth=1, decoder_num_heads=1, decoder_num_classes=1536, qkv_bias=True,
                                         norm_layer=partial(torch.nn.LayerNorm, eps=1e-6))
    opt = types.SimpleNamespace(state_dict=lambda: {"state": {}, "param_groups": []}, load_state_dict=lambda s: None)
    gen = DeviceTubeMaskingGenerator((8, 14, 14), 0.9, seed=11, rank=1, world_size=2)
    gen.clips_drawn = 640
    args = types.SimpleNamespace(output_dir=str(tmp_path), mask_generator=gen)
    utils.save_model(args, 4, model, model, opt, utils.NativeScalerWithGradNormCount())
    ck = torch.load(str(tmp_path / "checkpoint-4.pth"), map_location="cpu", weights_only=False)
    assert ck["mask_generator"] == {"seed": 11, "clips_drawn": 640, "world_size": 2}
    assert not hasattr(ck["args"], "mask_generator") and args.mask_generator is gen      # the object stays out of the file, the caller keeps it
    # the file unpickles where this package does not exist (the reference's loaders: any interpreter without mofo_amd on its path)
    import subprocess
    import sys
    code = ("import sys, torch; sys.modules['mofo_amd'] = None; ck = torch.load(sys.argv[1], map_location='cpu', weights_only=False); "
            "print(sorted(ck)); assert 'mofo_amd' not in repr(type(ck['args']))")
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path / "checkpoint-4.pth")], capture_output=True, text=True, cwd=str(tmp_path), timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    assert "'args', 'epoch', 'mask_generator', 'model', 'optimizer', 'scaler'" in r.stdout
    gen2 = DeviceTubeMaskingGenerator((8, 14, 14), 0.9, seed=0, rank=1, world_size=2)
    args2 = types.SimpleNamespace(output_dir=str(tmp_path), auto_resume=True, resume="", mask_generator=gen2)
    utils.auto_load_model(args2, model, model, opt, utils.NativeScalerWithGradNormCount())
    assert (gen2.seed, gen2.clips_drawn, args2.start_epoch) == (11, 640, 5)
    gen3 = DeviceTubeMaskingGenerator((8, 14, 14), 0.9, seed=0, rank=1, world_size=4)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        gen3.load_state_dict(ck["mask_generator"])
    assert any("world_size 2" in str(x.message) for x in w)


def test_bucket_override_is_read_once_and_checked(monkeypatch):
    """MOFO_ENC_BUCKETS is parsed when the runtime is built (plan_segments and encoder_backward use the same list) and a malformed
    value is an error at construction, not a silent fall-back"""
    model = _tiny_model()                      # encoder depth 5
    monkeypatch.setenv("MOFO_ENC_BUCKETS", "3,2")
    rt, st = _cpu_runtime(model)
    assert rt.enc_buckets() == [3, 2] and len(rt.segments) == 3
    monkeypatch.setenv("MOFO_ENC_BUCKETS", "1,1,1,1,1")
    assert rt.enc_buckets() == [3, 2]                                  # a later change of the variable does not move the bucket ends
    for bad in ("3,x", "4,2", "5,0"):
        monkeypatch.setenv("MOFO_ENC_BUCKETS", bad)
        with pytest.raises(ValueError, match="MOFO_ENC_BUCKETS"):
            _cpu_runtime(_tiny_model())


def test_saved_checkpoint_feeds_the_reference_finetune_loader(tmp_path):
    """what mofo_amd.utils.save_model writes goes through the reference's fine-tuning loader logic unchanged
    (run_class_finetuning.py:355-381 restated: pick checkpoint['model'], drop a mismatching head, strip 'backbone.' /
    'encoder.' prefixes) and lands on every backbone key of the fine-tune model; and the file's key order is the order the
    reference's own writer produced (tests/golden/ckpt_tiny.npz, written by the reference's utils.save_model)"""
    import types
    from collections import OrderedDict
    from functools import partial
    from mofo_amd import modeling_pretrain as mp, utils
    from oracle import pretrain_oracle as O
    cfg = O.TINY
    model = mp.PretrainVisionTransformer(img_size=32, encoder_embed_dim=128, encoder_depth=2, encoder_num_heads=2, decoder_embed_dim=64,
                                         decoder_depth=1, decoder_num_heads=1, decoder_num_classes=1536, qkv_bias=True,
                                         norm_layer=partial(torch.nn.LayerNorm, eps=1e-6))
    opt = types.SimpleNamespace(state_dict=lambda: {"state": {}, "param_groups": []})     # the optimizer entry is exercised on the GPU
    args = types.SimpleNamespace(output_dir=str(tmp_path))
    utils.save_model(args, 3, model, model, opt, utils.NativeScalerWithGradNormCount())
    checkpoint = torch.load(str(tmp_path / "checkpoint-3.pth"), map_location="cpu", weights_only=False)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ckpt_tiny.npz"))
    assert sorted(checkpoint) == [str(k) for k in g["top_keys"]]
    assert list(checkpoint["model"]) == [str(k) for k in g["model_keys"]]
    # --- the reference loader's steps
    checkpoint_model = None
    for model_key in "model|module".split("|"):
        if model_key in checkpoint:
            checkpoint_model = checkpoint[model_key]
            break
    new_dict = OrderedDict()
    for key in list(checkpoint_model.keys()):
        if key.startswith("backbone."):
            new_dict[key[9:]] = checkpoint_model[key]
        elif key.startswith("encoder."):
            new_dict[key[8:]] = checkpoint_model[key]
        else:
            new_dict[key] = checkpoint_model[key]
    want = O.finetune_param_shapes(cfg, 10)
    backbone = [k for k in want if not k.startswith(("head.", "fc_norm."))]
    assert all(k in new_dict and tuple(new_dict[k].shape) == want[k] for k in backbone)
    assert torch.equal(new_dict["blocks.1.mlp.fc2.weight"], model.state_dict()["encoder.blocks.1.mlp.fc2.weight"])


def test_launcher_flags_and_synthetic_dataset_contract():
    """mofo_amd.run_mae_pretraining: the reference's flag names / defaults for what the path reads (run_mae_pretraining.py:22-131)
    and a dataset whose items have VideoMAE.__getitem__'s layout (kinetics.py:492-495), normalised exactly like the
    reference's transforms (oracle.ingest_uint8 is pinned to them by tests/golden/ingest.npz)"""
    from mofo_amd import run_mae_pretraining as R
    from oracle import pretrain_oracle as O
    a = R.get_args([])
    ref_defaults = dict(batch_size=12, epochs=800, save_ckpt_freq=50, model="pretrain_videomae_base_patch16_224", decoder_depth=4,
                        mask_type="tube", mask_ratio=0.9, input_size=224, drop_path=0.0, normlize_target=True, opt="adamw",
                        opt_eps=1e-8, opt_betas=(0.9, 0.95), clip_grad=None, weight_decay=0.05, weight_decay_end=None, lr=1.5e-4,
                        warmup_lr=1e-6, min_lr=1e-5, warmup_epochs=40, warmup_steps=-1, num_frames=16, sampling_rate=2, seed=0,
                        resume="", auto_resume=True, start_epoch=0, pin_mem=True, world_size=1, local_rank=-1, dist_url="env://")
    for k, v in ref_defaults.items():
        assert getattr(a, k) == v, k
    b = R.get_args(["--no_auto_resume", "--no_pin_mem", "--normlize_target", "False", "--opt_betas", "0.8", "0.9", "--mask_ratio_BB", "0.75"])
    assert (b.auto_resume, b.pin_mem, b.normlize_target, b.opt_betas, b.mask_ratio_BB) == (False, False, False, [0.8, 0.9], 0.75)
    np.random.seed(4)
    ds = R.SyntheticClips(5, 8, 64, (4, 4, 4), 0.9, seed=3)
    clip, mask = ds[2]
    assert clip.shape == (3, 8, 64, 64) and clip.dtype == torch.float32 and clip.is_contiguous()
    assert mask.shape == (64,) and mask.dtype == np.float64 and mask.sum() == 4 * 14
    assert torch.equal(clip, ds[2][0]) and not torch.equal(clip, ds[3][0])                # deterministic per index
    raw = R.SyntheticClips(5, 8, 64, (4, 4, 4), 0.9, uint8_frames=True, seed=3)[2][0]
    assert raw.shape == (64, 64, 24) and raw.dtype == torch.uint8
    assert torch.equal(O.ingest_uint8(raw[None])[0], clip)                                # same pixels, reference normalisation
    clip_b, boxes, mask_b = R.SyntheticClips(5, 8, 64, (4, 4, 4), 0.9, mask_ratio_BB=0.75, seed=3)[1]
    assert boxes.shape == (8, 4) and mask_b.sum() == 4 * 14 and torch.equal(boxes[0], boxes[7])
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=2)))                     # default collate, as the reference uses
    assert batch[0].shape == (2, 3, 8, 64, 64) and batch[1].shape == (2, 64) and batch[1].dtype == torch.float64
