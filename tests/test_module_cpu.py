"""CPU: the drop-in module's parameter surface (names, shapes, init stream, checkpoint handling)."""
import os

import pytest
import torch

from conftest import GOLDEN
from oracle import esrgan_oracle as O
import super_resolution_amd as S


def test_state_dict_matches_reference_manifest():
    lines = open(os.path.join(GOLDEN, "g5_manifest.txt")).read().strip().splitlines()
    sd = S.GeneratorRRDB().state_dict()
    assert len(sd) == 702 == len(lines)
    for line, (k, v) in zip(lines, sd.items()):
        name, rest = line.split(" ", 1)
        assert name == k
        assert rest == f"{tuple(v.shape)} float32"


def test_same_seed_gives_reference_weights(golden):
    # the reference constructor under torch.manual_seed(2) drew the weights summarised in G3
    g = golden("g3_gen2.npz")
    torch.manual_seed(int(g["seed"]))
    net = S.GeneratorRRDB(num_blocks=2)
    sd = net.state_dict()
    ora = O.make_generator_state(int(g["seed"]), num_blocks=2)
    assert list(sd) == list(ora)
    for k in sd:
        assert torch.equal(sd[k], ora[k]), k
        if k != "conv_last.bias":
            assert sd[k].double().sum().item() == float(g[f"wsum/{k}"])


def test_constructor_signature_and_alias():
    net = S.GeneratorRRDB(3, 3, 64, 1, 32)
    assert S.RRDBNet is S.GeneratorRRDB
    assert sum(p.numel() for p in net.parameters()) == 1728 + 64 + 3 * 239808 + 4 * 36928 + 1731
    assert len(list(net.buffers())) == 0
    with pytest.raises(ValueError):
        S.GeneratorRRDB(precision="fp8")


def test_dataparallel_prefixed_checkpoint_loads_like_test_py():
    # esrgan/test.py:41-50 strips 'module.' and loads with strict=False
    src = S.GeneratorRRDB(num_blocks=1)
    ckpt = {"netG": {"module." + k: v for k, v in src.state_dict().items()}, "iter": 5}
    sd = ckpt.get("netG", ckpt)
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    dst = S.GeneratorRRDB(num_blocks=1)
    missing, unexpected = dst.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    for a, b in zip(src.parameters(), dst.parameters()):
        assert torch.equal(a, b)


def test_cpu_tensors_fail_loudly():
    net = S.GeneratorRRDB(num_blocks=1)
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.rand(1, 3, 8, 8))
    with pytest.raises(RuntimeError, match="3 channels"):
        net(torch.rand(1, 4, 8, 8))


def test_batched_pair_calls_have_no_cpu_fallback_either():
    """Discriminator.forward_pair / VGGFeatureExtractor.forward_pair (the two calls of esrgan/train.py:200-201 and
    :214-216 as one pass) refuse CPU tensors and mismatched batches like the plain calls, and the multi-block weight
    gradient refuses an empty or over-long block list before it touches the library."""
    from super_resolution_amd import ops
    d, v = S.Discriminator(), S.VGGFeatureExtractor(pretrained=False)
    a, b = torch.rand(2, 3, 32, 32), torch.rand(2, 3, 32, 32)
    for net in (d, v):
        with pytest.raises(RuntimeError, match="only on a ROCm device"):
            net.forward_pair(a, b)
    with pytest.raises(RuntimeError):
        ops.launch_dense_wgrad_blocks([])
    with pytest.raises(RuntimeError):
        ops.launch_dense_wgrad_blocks([None] * (ops.DENSE_WGRAD_MAX_BLOCKS + 1))
    plan = v._plan()
    assert [k for k, _m, _r in plan].count("conv") == 16 and [k for k, _m, _r in plan].count("pool") == 4
    assert plan[-1][0] == "conv" and plan[-1][2] is False          # features[:35] ends in conv5_4 without its ReLU


def test_discriminator_state_dict_matches_reference_manifest():
    """Keys, shapes and dtypes of the reference Discriminator's state_dict (tests/golden/g6_disc_manifest.txt)."""
    net = S.Discriminator()
    got = [f"{k} {tuple(v.shape)} {str(v.dtype).replace('torch.', '')}" for k, v in net.state_dict().items()]
    want = open(os.path.join(os.path.dirname(__file__), "golden", "g6_disc_manifest.txt")).read().split("\n")
    assert got == [w for w in want if w]
    net2 = S.Discriminator(in_channels=1, base_channels=16)
    assert net2.model[0].in_channels == 1 and net2.model[-1].in_channels == 128


def test_same_seed_gives_reference_discriminator_weights(golden):
    g = golden("g6_disc.npz")
    torch.manual_seed(int(g["seed"]))
    net = S.Discriminator()
    for k, v in net.state_dict().items():
        if v.dtype.is_floating_point and "running" not in k:
            assert abs(v.double().sum().item() - float(g[f"wsum/{k}"])) < 1e-6, k


def test_vgg_extractor_keys_follow_torchvision_numbering():
    from oracle import esrgan_oracle as O
    net = S.VGGFeatureExtractor(pretrained=False)
    assert list(net.state_dict().keys()) == list(O.make_vgg_state(0).keys())
    assert len(net.features) == 35 and isinstance(net.features[34], torch.nn.Conv2d)
    assert all(not p.requires_grad for p in net.parameters())
    assert len(S.VGGFeatureExtractor(feature_layer=5, pretrained=False).features) == 5
    with pytest.raises(ImportError):
        S.VGGFeatureExtractor()            # ImageNet weights come from torchvision, absent here


def test_blocked_unit_arithmetic():
    from super_resolution_amd import conv_core as cc
    assert [cc.units_for(c) for c in (1, 3, 8, 9, 16, 17, 64, 512)] == [2, 2, 2, 2, 2, 4, 8, 64]


def test_cyclegan_generator_sr_surface():
    """CycleGAN/CycleGAN_w_v1/CycleGAN.py:103-127: constructor arguments and state_dict keys / shapes of the
    reference class (manifest written from it); CPU input is refused."""
    import os
    from super_resolution_amd.cyclegan import GeneratorSR, ResidualBlock
    net = GeneratorSR()
    lines = open(os.path.join(os.path.dirname(__file__), "golden", "g9_cyclegan_sr_manifest.txt")).read().splitlines()
    want = [(ln.split(" ", 1)[0], ln.split(" ", 1)[1]) for ln in lines]
    got = [(k, str(tuple(v.shape))) for k, v in net.state_dict().items()]
    assert got == want
    assert len(GeneratorSR(num_residual=2, base_channels=32).res_blocks) == 2
    with pytest.raises(RuntimeError):
        net(torch.rand(1, 3, 8, 8))
    with pytest.raises(RuntimeError):
        ResidualBlock(64)(torch.rand(1, 64, 8, 8))
    with pytest.raises(ValueError):
        GeneratorSR(base_channels=12)


def test_conditional_vae_surface():
    """"condtional VAE/model.py":53-147: constructor argument, methods and state_dict keys / shapes of the reference
    class (manifest written from it at the default latent_dim); CPU input is refused."""
    import os
    from super_resolution_amd.cvae import CVAE
    net = CVAE()
    lines = open(os.path.join(os.path.dirname(__file__), "golden", "g10_cvae_manifest.txt")).read().splitlines()
    want = [(ln.split(" ", 1)[0], ln.split(" ", 1)[1]) for ln in lines]
    got = [(k, str(tuple(v.shape))) for k, v in net.state_dict().items()]
    assert got == want
    assert net.fc_mu.out_features == 256 and CVAE(latent_dim=32).decoder_input.in_features == 32
    for name in ("encode", "reparameterize", "decode", "forward", "sample"):
        assert callable(getattr(net, name))
    with pytest.raises(RuntimeError):
        net(torch.rand(1, 3, 32, 32))
    with pytest.raises(RuntimeError):
        net.sample(1, torch.rand(1, 3, 32, 32))


def test_conv_parameter_list_follows_the_module_tree():
    """The packing code lists the 702 parameters itself (state_dict order); a replaced submodule or parameter must
    show up in the next listing, as it does in ``parameters()``."""
    import torch.nn as nn
    net = S.GeneratorRRDB(num_blocks=2)
    assert [id(p) for p in net.conv_parameters()] == [id(p) for p in net.parameters()]
    net.body[1].rdb2.conv4 = nn.Conv2d(160, 32, 3, 1, 1)
    net.conv_last.bias = nn.Parameter(torch.zeros(3))
    assert [id(p) for p in net.conv_parameters()] == [id(p) for p in net.parameters()]
    assert [k for k, _ in net.named_parameters()] == [f"{n}.{s}" for n, m in net.named_modules()
                                                       if isinstance(m, nn.Conv2d) for s in ("weight", "bias")]


def test_replicas_never_reuse_a_cached_pack(monkeypatch):
    """An nn.DataParallel replica's parameters are fresh broadcast outputs every forward: version 0, recycled
    addresses.  (version, address) therefore cannot key a cache for them: a replica must pack on every call, and the
    owning module keeps caching."""
    from super_resolution_amd import esrgan_model as M, ops
    calls = []
    monkeypatch.setattr(ops, "generator_pack_now", lambda w, b, cfg: calls.append("fwd") or object())
    monkeypatch.setattr(ops, "generator_pack_bwd_now", lambda w, cfg: calls.append("bwd") or object())
    monkeypatch.setattr(ops, "pack_weight", lambda w, b, dt: calls.append("rdb") or (None, None))
    net = S.GeneratorRRDB(num_blocks=1)
    # the owner caches: one pack for two calls, a new one after an in-place update
    a, b = net.packed_parameters(), net.packed_parameters()
    assert a is b and calls == ["fwd"]
    with torch.no_grad():
        net.conv_first.weight.add_(1.0)
    assert net.packed_parameters() is not a and calls == ["fwd", "fwd"]
    # a replica as nn.parallel.replicate builds it: same __dict__ (so the SAME cache objects), parameters as plain
    # attributes at unchanged addresses with version counters at 0
    del calls[:]
    replica = net._replicate_for_data_parallel()
    for name, mod in net.named_modules():
        if name:
            parent = replica if "." not in name else replica.get_submodule(name.rsplit(".", 1)[0])
            rep = mod._replicate_for_data_parallel()
            parent._modules[name.rsplit(".", 1)[-1]] = rep
            for pname, p in mod._parameters.items():
                setattr(rep, pname, p.detach())
    assert M._is_replica(replica) and not M._is_replica(net)
    p1, p2 = replica.packed_parameters(), replica.packed_parameters()
    assert p1 is not p2 and calls == ["fwd", "fwd"]
    replica.packed_bwd_parameters(), replica.packed_bwd_parameters()
    assert calls == ["fwd", "fwd", "bwd", "bwd"]
    rdb = replica.body[0].rdb1
    rdb._packed(0), rdb._packed(0)
    assert calls.count("rdb") == 10
    # the replica left the owner's entry alone
    n = len(calls)
    net.packed_parameters()
    assert len(calls) == n
