"""CPU-side checks: the C-ABI library loads and exports every declared symbol, the host model
exposes the reference's state_dict keys, host samplers are bit-exact against the golden vectors,
and the product path refuses to run without a GPU (no silent fallback)."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from audio8_amd import _lib as L
    lib = L.lib()
    hdr = open(os.path.join(ROOT, "include", "a8hip.h")).read()
    declared = sorted(set(re.findall(r"\b(a8_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations found"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/a8hip.h but not exported"
    assert sorted(L.EXPORTS) == declared
    assert lib.a8_version() == L.ABI_VERSION == int(re.search(r"#define A8HIP_VERSION (\d+)", hdr).group(1))


def test_state_dict_keys_match_reference_names():
    from audio8_amd import wav2vec2 as W
    from oracle import ref_cpu as R
    m = W.create_model(num_layers=2)
    want = R.init_state_dict(num_layers=2)
    got = m.state_dict()
    assert set(got.keys()) == set(want.keys())
    for k in want:
        assert tuple(got[k].shape) == tuple(want[k].shape), k
    assert sum(p.numel() for p in m.feature_extractor.parameters()) == 4200448
    a = W.create_acoustic_model(32, num_layers=1)
    wantc = R.init_state_dict(num_layers=1, ctc_labels=32)
    assert set(a.state_dict().keys()) == set(wantc.keys())
    assert a.freeze is True and hasattr(a, "proj") and hasattr(a, "encoder")
    full = W.create_model()
    assert sum(p.numel() for p in full.parameters()) == 95044608      # wav2vec2-base, SURVEY 2.2 C1 ("~95.0 M")


def test_fairseq_key_maps_cover_model():
    from audio8_amd import wav2vec2 as W
    m = W.create_model(num_layers=2)
    sd = m.state_dict()
    fake = {}
    for i in range(2):
        for k, v in W.W2V_NESTED_MAP.items():
            fake[k.format(i)] = sd[v.format(i)].clone()
    for k, v in W.W2V2_FLAT_MAP.items():
        fake[k] = sd[v].clone()
    mapped = W.convert_keys(2, dict(fake), W.W2V_MAP.nested, W.W2V_MAP.flat)
    assert set(mapped.keys()) <= set(sd.keys())
    res = m.load_state_dict(mapped, strict=False)
    assert not res.unexpected_keys
    a = W.create_acoustic_model(32, num_layers=1)
    sda = a.state_dict()
    for k, v in list(W.W2V_CTC_MAP[16].nested.items()):
        assert v.format(0) in sda, v
    for k, v in W.W2V_CTC_MAP[16].flat.items():
        assert v in sda, v
    assert len(W.W2V2_CTC_FLAT_MAP_8) == len(W.W2V2_CTC_FLAT_MAP_16) - 1


def test_host_samplers_match_golden(golden_dir):
    from audio8_amd import wav2vec2 as W
    g = np.load(os.path.join(golden_dir, "g1_masks.npz"))
    for key in g["keys"]:
        key = str(key)
        seed = int(key.split("_")[0][1:])
        B, T = (int(v) for v in g[key + "_shape"])
        p = int(key.split("_p")[1].split("_")[0]) / 100.0
        ml = int(key.split("_l")[1])
        np.random.seed(seed)
        m = W.create_mask((B, T), p, ml)
        assert np.array_equal(m, np.unpackbits(g[key])[: B * T].reshape(B, T).astype(bool)), key
        assert np.random.rand() == float(g[key + "_tail"])
    g2 = np.load(os.path.join(golden_dir, "g2_negatives.npz"))
    for key in g2["keys"]:
        key = str(key)
        seed = int(key.split("_")[0][1:])
        B = int(key.split("_b")[1].split("_")[0]); Tm = int(key.split("_t")[1].split("_")[0]); N = int(key.split("_n")[1])
        np.random.seed(seed)
        assert np.array_equal(W.Sampler(N).indices(B, Tm), g2[key].astype(np.int64)), key


def test_quantizer_temperature_quirk():
    from audio8_amd import wav2vec2 as W
    m = W.create_model(num_layers=1)
    assert m.quantizer.curr_temperature == 0.5
    m.set_num_updates(1000)
    assert m.quantizer.curr_temperature == 2


def test_product_path_has_no_cpu_fallback():
    from audio8_amd import wav2vec2 as W
    m = W.create_model(num_layers=1)
    with pytest.raises(RuntimeError, match="GPU|cuda|CUDA"):
        m(torch.zeros(1, 4000))


def test_ctc_metrics_and_edit_distance():
    from audio8_amd import ctc as Cc
    vocab = {0: '<s>', 1: '<pad>', 2: '</s>', 3: '<unk>', 4: '|', 5: 'a', 6: 'b', 7: 'c'}
    seq = [5, 5, 0, 5, 4, 6, 6, 7]
    lp = torch.full((1, len(seq), 8), -10.0)
    for t, s in enumerate(seq):
        lp[0, t, s] = 0.0
    tgt = torch.tensor([[5, 5, 4, 6, 6, 2, 1, 1]])
    assert Cc.ctc_metrics(lp, tgt, [10 ** 6], vocab) == dict(wv_errors=1, w_errors=1, w_total=2, c_errors=1, c_total=5)
    assert Cc.edit_distance("kitten", "sitting") == 3 and Cc.edit_distance([], [1, 2]) == 2


def test_crop_to_min_batching_matches_oracle_incl_dropped_sample_quirk():
    """audio8_amd.onramp.crop_to_min_batches against the oracle's restatement of AudioFileDataset.__iter__
    (audio8/data.py:409-426) on ragged inputs; the sample that arrives when the budget is full is dropped."""
    from audio8_amd.onramp import crop_to_min_batches
    from oracle import ref_cpu as R
    rng = np.random.default_rng(0)
    for trial in range(5):
        lens = rng.integers(50, 400, size=60)
        samples = [np.arange(n, dtype=np.float32) + 1000 * i for i, n in enumerate(lens)]
        got = [np.stack(b) for b in crop_to_min_batches(iter(samples), 300, 1200)]
        want = R.crop_to_min_batches(samples, 300, 1200)
        assert len(got) == len(want) >= 2
        for g, w in zip(got, want):
            np.testing.assert_array_equal(g, w)
    # the quirk, explicitly: with budget 2 * 100 the third sample closes the batch and never appears anywhere
    s = [np.full(100, i, dtype=np.float32) for i in range(6)]
    b = [np.stack(x) for x in crop_to_min_batches(iter(s), 100, 200)]
    assert [sorted(set(x[:, 0].tolist())) for x in b] == [[0.0, 1.0], [3.0, 4.0]]


def test_driver_surface_schedules_checkpoint_names_and_aliases(tmp_path):
    """The eight_mile names the unchanged drivers import (audio8/pretrain.py:17-23, train.py:15-19), the warm-up/cosine LR
    composite of audio8/utils.py:5-15, and the checkpoint naming that pretrain.py:135-137 parses the step back from."""
    import importlib
    import sys
    import torch
    from audio8_amd import driver as D
    lr = 2e-4
    s = D.create_lrs(lr, 1000, warmup_steps=100, plateau_steps=50)
    assert s(0) == 0.0 and abs(s(50) - lr / 2) < 1e-12 and s(100) == lr and s(149) == lr            # linear warm-up, plateau
    assert s(150) == lr and abs(s(650) - lr / 2) < 1e-12 and abs(s(1150)) < 1e-12 and abs(s(10 ** 6)) < 1e-12      # cosine, held at the end
    assert abs(D.create_lrs(lr, 1000, alpha=0.1, warmup_steps=0)(1000) - 0.1 * lr) < 1e-12
    m = torch.nn.Linear(3, 2)
    base = str(tmp_path / "ckpt" / "checkpoint")
    for step in (10, 250, 90):
        D.save_checkpoint(m, base, step, tick_type="step")
    name, step = D.find_latest_checkpoint(str(tmp_path / "ckpt"))
    assert name.endswith("checkpoint-step-250.pth") and step == 250
    assert int(name.split("-")[-1].split(".")[0]) == 250                    # the parse of audio8/pretrain.py:135-137
    sd = torch.load(name)
    assert sorted(sd) == ["bias", "weight"] and torch.equal(sd["weight"], m.weight.detach())
    wrapped = D.PassThroughDDP(m)                                            # `.module` is unwrapped like DDP's
    D.save_checkpoint(wrapped, base, 300, tick_type="step")
    assert sorted(torch.load(base + "-step-300.pth")) == ["bias", "weight"]
    a = D.Average("average_train_loss")
    a.update(2.0); a.update(4.0)
    assert a.avg == 3.0 and "average_train_loss" in str(a)
    assert D.str2bool("True") is True and D.str2bool("0") is False and D.revlut({"a": 1}) == {1: "a"}
    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in ("eight_mile", "audio8")}
    ddp = torch.nn.parallel.DistributedDataParallel
    try:
        D.install_aliases()
        want = {"eight_mile.utils": ["str2bool", "Average", "get_num_gpus_multiworker", "Offsets", "revlut"],
                "eight_mile.pytorch.optz": ["OptimizerManager"],
                "eight_mile.pytorch.layers": ["save_checkpoint", "init_distributed", "sequence_mask", "find_latest_checkpoint", "Dense",
                                              "TransformerEncoderStack", "Conv1DSame", "pytorch_conv1d", "pytorch_linear"],
                "eight_mile.optz": ["create_lr_scheduler", "CompositeLRScheduler"],
                "audio8.utils": ["create_lrs"],
                "audio8.wav2vec2": ["create_loss", "create_model", "load_fairseq_bin", "create_acoustic_model"],
                "audio8.ctc": ["CTCLoss", "ctc_metrics", "postproc_bpe", "postproc_letters"]}
        for modname, names in want.items():
            mod = importlib.import_module(modname)
            for n in names:
                assert hasattr(mod, n), (modname, n)
        ns = {}
        exec("from eight_mile.pytorch.optz import *", ns)                    # audio8/pretrain.py:20
        assert "OptimizerManager" in ns
        assert torch.nn.parallel.DistributedDataParallel is D.PassThroughDDP
    finally:
        torch.nn.parallel.DistributedDataParallel = ddp
        for k in [k for k in sys.modules if k.split(".")[0] in ("eight_mile", "audio8")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_g9_key_maps_and_convert_keys_equal_the_reference(golden_dir):
    """G9: audio8_amd.wav2vec2's fairseq key maps are generated programmatically; they must equal the reference's literal
    tables (audio8/wav2vec2.py:39-151) entry for entry, and convert_keys must rename / order / pass through exactly as the
    reference's (audio8/wav2vec2.py:154-168)."""
    import os
    import numpy as np
    from audio8_amd import wav2vec2 as W
    g = np.load(os.path.join(golden_dir, "g9_keymaps.npz"))
    for name in ("W2V_CTC_NESTED_MAP", "W2V2_CTC_FLAT_MAP_16", "W2V2_CTC_FLAT_MAP_8", "W2V_NESTED_MAP", "W2V2_FLAT_MAP"):
        want = dict(zip((str(k) for k in g[name + "_src"]), (str(v) for v in g[name + "_dst"])))
        assert getattr(W, name) == want, name
    for tag, cm in (("pretrain", W.W2V_MAP), ("ctc16", W.W2V_CTC_MAP[16]), ("ctc8", W.W2V_CTC_MAP[8])):
        src = {str(k): i for i, k in enumerate(g[f"{tag}_in_keys"])}
        out = W.convert_keys(int(g[f"{tag}_layers"]), dict(src), cm.nested, cm.flat)
        # ORDERED: the fixture stores the reference's output key order (its dict is built flat map first, then layer by layer in the
        # nested map's order, audio8/wav2vec2.py:154-168) -- state_dict consumers may iterate it
        assert list(out.items()) == list(zip((str(k) for k in g[f"{tag}_out_keys"]), (int(v) for v in g[f"{tag}_out_vals"]))), tag
        broken = dict(src)
        broken.pop(next(iter(cm.flat)))
        with pytest.raises(KeyError):
            W.convert_keys(int(g[f"{tag}_layers"]), broken, cm.nested, cm.flat)


def test_native_negative_sampler_is_numpy_bit_for_bit():
    """a8_host_negative_indices restates numpy's legacy randint (MT19937, masked rejection) on a copy of the global state: same indices
    as the reference's numpy statement (audio8/wav2vec2.py:962-970) and the SAME generator state afterwards (the next draws of the global
    RNG agree), for ranges on both sides of a power of two, across several state regenerations, from every starting position."""
    import numpy as np
    from audio8_amd import wav2vec2 as W
    for seed, B, T, N in ((0, 8, 355, 100), (1, 3, 2, 5), (2, 1, 3, 7), (3, 2, 257, 10), (4, 2, 258, 10), (5, 4, 1025, 3), (6, 5, 64, 100)):
        s = W.Sampler(N)
        for burn in (0, 1, 623, 624, 1000):
            np.random.seed(seed)
            if burn:
                np.random.randint(0, 2 ** 31 - 1, burn)
            want = s.indices_numpy(B, T)
            tail_want = (np.random.random(), np.random.randint(0, 1000, 5).tolist(), np.random.normal())
            np.random.seed(seed)
            if burn:
                np.random.randint(0, 2 ** 31 - 1, burn)
            got = s.indices(B, T)
            tail_got = (np.random.random(), np.random.randint(0, 1000, 5).tolist(), np.random.normal())
            assert got.dtype == (np.int32 if T >= 3 else want.dtype)
            assert np.array_equal(got, want), (seed, B, T, N, burn)
            assert tail_got == tail_want, (seed, B, T, N, burn)
    # a pending cached Gaussian survives the round trip through get_state / set_state
    np.random.seed(9)
    np.random.normal()
    a = W.Sampler(4).indices(2, 50)
    g1 = np.random.normal()
    np.random.seed(9)
    np.random.normal()
    b = W.Sampler(4).indices_numpy(2, 50)
    g2 = np.random.normal()
    assert np.array_equal(a, b) and g1 == g2


def test_pin_host_cores_takes_the_ranks_own_slice():
    """trainer.pin_host_cores(n, rank, world): rank r of W gets n cores from the r-th of W equal slices of the allowed CPU list (what bench.py
    does before importing torch); different ranks never share a core when the slices are at least n wide."""
    import torch
    from audio8_amd.trainer import pin_host_cores
    before, threads = sorted(os.sched_getaffinity(0)), torch.get_num_threads()
    try:
        if len(before) < 2:
            pytest.skip("needs two CPUs")
        world = 2
        per = len(before) // world
        n = min(2, per)
        got = []
        for r in range(world):                               # (one call per process in real use: the slice is taken from the CURRENT mask)
            os.sched_setaffinity(0, before)
            got.append(pin_host_cores(n, rank=r, world=world))
            assert sorted(os.sched_getaffinity(0)) == got[-1] and torch.get_num_threads() == n
        assert all(len(g) == n for g in got) and not (set(got[0]) & set(got[1]))
        assert got[0] == before[:n] and got[1] == before[per:per + n]
    finally:
        os.sched_setaffinity(0, before)
        torch.set_num_threads(threads)


def test_flat_layout_keeps_late_gradients_out_of_the_first_buckets():
    """trainer.FlatState lays parameters out in forward-use order: exchange buckets are cut from the END of the layout and leave in that
    order, so the encoder's input LayerNorm (`ln`: registered after the layer stack, audio8/wav2vec2.py:613-623, but applied before it --
    its gradient arrives when the whole stack's backward is through) must sit in FRONT of the stack's parameters, next to the positional
    conv; each block's w_Q / w_K / w_V stay adjacent (the fused QKV product addresses them as one matrix)."""
    from audio8_amd import wav2vec2 as W
    from audio8_amd.trainer import FlatState
    for model in (W.create_model(16, num_layers=3), W.create_acoustic_model(32, 16, num_layers=3)):
        for p in model.parameters():
            p.requires_grad_(True)
        names = {id(p): n for n, p in model.named_parameters()}
        order = [names[id(p)] for p in FlatState._ordered(model)]
        assert sorted(order) == sorted(names.values())                       # a permutation: nothing lost, nothing twice
        ln = [k for k, n in enumerate(order) if n.endswith("encoder.ln.weight") or n.endswith("encoder.ln.bias")]
        stack = [k for k, n in enumerate(order) if "transformer.encoders." in n]
        posconv = [k for k, n in enumerate(order) if "pos_conv" in n]
        assert len(ln) == 2 and max(posconv) < min(ln) and max(ln) < min(stack), order
        assert stack == list(range(min(stack), max(stack) + 1))              # the stack's parameters stay one contiguous run
        for k, n in enumerate(order):
            if n.endswith("w_Q.layer.weight"):
                stem = n[:-len("w_Q.layer.weight")]
                assert order[k:k + 6] == [stem + f"w_{c}.layer.{leaf}" for leaf in ("weight", "bias") for c in "QKV"]


def test_abi_descriptor_sizes_and_tuning_round_trip():
    """ABI 120 (include/a8hip.h): every descriptor starts with its own sizeof and an entry point refuses another layout BEFORE it touches
    the GPU (status 1 + a8_last_error text, no launch: runs on a box without one); a8_set_tuning / a8_get_tuning round trip, unknown keys
    are errors; the binding's struct sizes are the library's (the size checks pass for the binding's own structs up to the first
    argument check)."""
    import ctypes as C
    from audio8_amd import _lib as L
    lib = L.lib()
    assert lib.a8_version() == L.ABI_VERSION == 120
    d = L.GemmDesc()
    d.size = L.GEMM_DESC_SIZE - 8                                   # a binding built against a shorter struct
    assert lib.a8_gemm(C.byref(d), None) == 1 and b"descriptor size" in lib.a8_last_error()
    d.size = L.GEMM_DESC_SIZE
    d.dtype = 7                                                     # the right size gets as far as the argument checks
    assert lib.a8_gemm(C.byref(d), None) == 1 and b"bad dtype" in lib.a8_last_error()
    for cls, fn, size in ((L.EncoderLayerDesc, lib.a8_encoder_layer_fwd, L.ENC_DESC_SIZE), (L.EncoderLayerBwdDesc, lib.a8_encoder_layer_bwd, L.ENC_BWD_DESC_SIZE)):
        e = cls()
        e.size = size + 8
        assert fn(C.byref(e), None) == 1 and b"descriptor size" in lib.a8_last_error()
        e.size = size
        assert fn(C.byref(e), None) == 1 and b"bad sizes" in lib.a8_last_error()
    f = L.EncoderLayerBwdFullDesc()
    f.size = 16
    assert lib.a8_encoder_layer_bwd_full(C.byref(f), None, None, None) == 1 and b"descriptor size" in lib.a8_last_error()
    f.size, f.nhalves = L.ENC_BWD_FULL_DESC_SIZE, 3
    assert lib.a8_encoder_layer_bwd_full(C.byref(f), None, None, None) == 1 and b"nhalves" in lib.a8_last_error()
    v = C.c_int(-1)
    assert lib.a8_get_tuning(5, C.addressof(v)) == 0
    found = v.value
    try:
        assert lib.a8_set_tuning(5, 12345) == 0 and lib.a8_get_tuning(5, C.addressof(v)) == 0 and v.value == 12345
    finally:
        assert lib.a8_set_tuning(5, found) == 0
    assert lib.a8_set_tuning(99, 1) == 1 and lib.a8_get_tuning(-1, C.addressof(v)) == 1 and lib.a8_get_tuning(5, None) == 1
    assert lib.a8_delay_us(C.c_float(-1.0), None) == 1              # (argument check; a positive delay would launch)
