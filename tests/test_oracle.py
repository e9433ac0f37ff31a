"""CPU: the oracle restatement against vectors produced by the reference classes themselves
(tests/golden/make_golden.py).  Bit-exact where the same torch ops run in the same order."""
import os
import numpy as np
import pytest
import torch

from oracle import esrgan_oracle as O


def t(a):
    return torch.from_numpy(np.asarray(a))


def sub_state(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def check_grad_summary(g, grads, stride=97, rtol=1e-5):
    names = [k[3:] for k in g if k.startswith("gs/")]
    assert names
    for name in names:
        flat = grads[name].reshape(-1)
        np.testing.assert_allclose(flat[::stride].numpy(), g[f"gs/{name}"], rtol=rtol, atol=1e-7)
        assert abs(flat.double().abs().sum().item() - float(g[f"gabs/{name}"])) <= 1e-6 * max(1.0, float(g[f"gabs/{name}"]))
    for k in g:
        if k.startswith("gfull/") and "[" not in k:
            np.testing.assert_allclose(grads[k[6:]].numpy(), g[k], rtol=rtol, atol=1e-7)


def test_weight_recipe_matches_reference_constructor(golden):
    # G3/G4/G7 store float64 sums of every weight the reference constructor drew from its seed.
    for name, nb in (("g3_gen2.npz", 2), ("g4_gen23.npz", 23)):
        g = golden(name)
        sd = O.make_generator_state(int(g["seed"]), num_blocks=nb)
        sd = O.center_outputs(sd, float(g["bias_shift"]))
        for k, v in sd.items():
            assert v.double().sum().item() == float(g[f"wsum/{k}"]) if f"wsum/{k}" in g else True
    g = golden("g4_gen23.npz")
    sd = O.make_generator_state(3)
    assert len(sd) == int(g["n_keys"]) == 702
    assert sum(v.numel() for v in sd.values()) == int(g["n_params"]) == 16697987


def test_state_dict_manifest(golden):
    import os
    from conftest import GOLDEN
    lines = open(os.path.join(GOLDEN, "g5_manifest.txt")).read().strip().splitlines()
    sd = O.make_generator_state(0)
    assert len(lines) == len(sd) == 702
    for line, (k, v) in zip(lines, sd.items()):
        name, rest = line.split(" ", 1)
        assert name == k and rest.startswith(str(tuple(v.shape))) and rest.endswith("float32")


def test_dense_block_g1(golden):
    g = golden("g1_rdb.npz")
    full = O.make_generator_state(0, num_blocks=0)  # not used: block weights are drawn alone below
    gen = torch.Generator().manual_seed(int(g["seed"]))
    sd = O.make_conv_state([(f"conv{k + 1}", 64 + 32 * k, 32 if k < 4 else 64) for k in range(5)], gen)
    for k, v in sd.items():
        assert v.double().sum().item() == float(g[f"wsum/{k}"])
    x = t(g["x"]).requires_grad_(True)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.dense_block_forward(leaves, "", x)
    assert torch.equal(out.detach(), t(g["out"]))
    (out * t(g["r"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["grad_x"], rtol=1e-5, atol=1e-7)
    check_grad_summary(g, {k: v.grad for k, v in leaves.items()})


def test_rrdb_g2(golden):
    g = golden("g2_rrdb.npz")
    gen = torch.Generator().manual_seed(int(g["seed"]))
    specs = [(f"rdb{r}.conv{k + 1}", 64 + 32 * k, 32 if k < 4 else 64) for r in (1, 2, 3) for k in range(5)]
    sd = O.make_conv_state(specs, gen)
    x = t(g["x"]).requires_grad_(True)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.rrdb_forward(leaves, "", x)
    assert torch.equal(out.detach(), t(g["out"]))
    (out * t(g["r"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["grad_x"], rtol=1e-5, atol=1e-7)
    check_grad_summary(g, {k: v.grad for k, v in leaves.items()})


def test_generator_g3_forward_backward(golden):
    g = golden("g3_gen2.npz")
    sd = O.center_outputs(O.make_generator_state(int(g["seed"]), num_blocks=2), float(g["bias_shift"]))
    out, grads, gx = O.generator_grads(sd, t(g["x"]), t(g["r"]), input_grad=True)
    assert torch.equal(out, t(g["out"]))
    assert out.shape == (1, 3, 48, 40)
    np.testing.assert_allclose(gx.numpy(), g["grad_x"], rtol=1e-5, atol=1e-7)
    check_grad_summary(g, grads)


def test_generator_g4_full(golden):
    g = golden("g4_gen23.npz")
    sd = O.center_outputs(O.make_generator_state(int(g["seed"])), float(g["bias_shift"]))
    x = t(g["x"])
    with torch.no_grad():
        out = O.generator_forward(sd, x)
    assert torch.equal(out, t(g["out"]))
    frac = O.inside_unit_fraction(out)
    assert frac > 0.9, frac      # the bias shift recipe keeps outputs inside (0,1)
    # gradients under L1-to-zeros
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    o = O.generator_forward(leaves, x)
    torch.nn.functional.l1_loss(o, torch.zeros_like(o)).backward()
    grads = {k: v.grad for k, v in leaves.items()}
    check_grad_summary(g, grads)
    np.testing.assert_allclose(grads["body.22.rdb3.conv5.weight"][:, :8].numpy(),
                               g["gfull/body.22.rdb3.conv5.weight[:, :8]"], rtol=1e-5, atol=1e-8)


def test_discriminator_g6(golden):
    g = golden("g6_disc.npz")
    sd = O.make_discriminator_state(int(g["seed"]))
    for k, v in sd.items():
        # running stats in the fixture were saved after the train-mode forward: compare parameters only
        if f"wsum/{k}" in g and "running" not in k:
            assert v.double().sum().item() == float(g[f"wsum/{k}"]), k
    img = torch.rand(2, 3, 128, 128, generator=torch.Generator().manual_seed(int(g["img_seed"])))
    train_sd = {k: v.clone() for k, v in sd.items()}
    out_train = O.discriminator_forward(train_sd, img, training=True)
    assert out_train.shape == (2, 1, 8, 8)
    np.testing.assert_allclose(out_train.numpy(), g["out_train"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(train_sd["model.3.running_mean"].numpy(), g["running_mean_2"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(train_sd["model.3.running_var"].numpy(), g["running_var_2"], rtol=1e-5, atol=1e-7)
    out_eval = O.discriminator_forward(train_sd, img, training=False)
    np.testing.assert_allclose(out_eval.numpy(), g["out_eval"], rtol=1e-4, atol=1e-5)


def test_pixel_step_g7(golden):
    g = golden("g7_pixel_step.npz")
    sd = O.make_generator_state(int(g["seed"]), num_blocks=2)
    loss, grads = O.pixel_step_grads(sd, t(g["lr"]), t(g["hr"]))
    assert abs(loss - float(g["loss_pix"])) < 1e-7
    check_grad_summary(g, grads)


def test_tiled_forward_converges_to_full_image():
    sd = O.make_generator_state(7, num_blocks=2)
    x = torch.rand(1, 3, 40, 24, generator=torch.Generator().manual_seed(8))
    with torch.no_grad():
        full = O.generator_forward(sd, x)
        tiled = O.generator_forward_tiled(sd, x, bands=3, halo=35)   # halo >= receptive field: exact
        narrow = O.generator_forward_tiled(sd, x, bands=3, halo=2)
    assert tiled.shape == full.shape
    assert (tiled - full).abs().max().item() < 1e-5
    assert (narrow - full).abs().max().item() > (tiled - full).abs().max().item()


def test_bf16_emulation_is_close_but_not_equal():
    sd = O.center_outputs(O.make_generator_state(11, num_blocks=3))
    x = torch.rand(1, 3, 16, 16, generator=torch.Generator().manual_seed(12))
    with torch.no_grad():
        ref = O.generator_forward(sd, x)
        emu = O.generator_forward(sd, x, emulate_bf16=True)
    assert not torch.equal(ref, emu)
    assert O.psnr(ref.clamp(0, 1), emu.clamp(0, 1)) > 45.0


def test_adversarial_iteration_matches_reference_run(golden):
    """G8: the reference generator + discriminator run through one adversarial iteration of esrgan/train.py:196-223
    (perceptual term on the oracle's seeded vgg19 restatement, the reference's own being unbuildable offline)."""
    g = golden("g8_adversarial_step.npz")
    sd_g = O.make_generator_state(int(g["seed_g"]), num_blocks=1)
    sd_d = O.make_discriminator_state(int(g["seed_d"]))
    sd_v = O.make_vgg_state(int(g["seed_vgg"]))
    r = O.adversarial_step(sd_g, sd_d, sd_v, torch.from_numpy(g["lr"]), torch.from_numpy(g["hr"]))
    for k in ("loss_D", "loss_G", "loss_G_pix", "loss_G_per", "loss_G_gan"):
        assert abs(r[k] - float(g[k])) <= 1e-5 * abs(float(g[k])) + 1e-7, k
    for name, grad in r["grads_D"].items():
        ref_sum, ref_abs = float(g[f"d_gsum/{name}"]), float(g[f"d_gabs/{name}"])
        assert abs(grad.double().abs().sum().item() - ref_abs) <= 1e-4 * ref_abs + 1e-9, name
        assert abs(grad.double().sum().item() - ref_sum) <= 1e-4 * ref_abs + 1e-9, name
        np.testing.assert_allclose(grad.reshape(-1)[::193].numpy(), g[f"d_gs/{name}"], rtol=2e-3, atol=1e-7)
    for name, grad in r["grads_G"].items():
        ref_abs = float(g[f"gabs/{name}"])
        assert abs(grad.double().abs().sum().item() - ref_abs) <= 1e-4 * ref_abs + 1e-9, name
    np.testing.assert_allclose(r["grads_G"]["conv_first.weight"].numpy(), g["gfull/conv_first.weight"], rtol=2e-3, atol=1e-8)
    np.testing.assert_allclose(r["state_D"]["model.3.running_mean"].numpy(), g["running_mean_3"], rtol=1e-4, atol=1e-7)
    assert int(r["state_D"]["model.3.num_batches_tracked"]) == int(g["num_batches_tracked_3"]) == 3
    for name, v in r["state_G"].items():          # parameters after both Adam steps
        ref = float(g[f"wsum/{name}"])
        assert abs(v.double().sum().item() - ref) <= 1e-5 * max(1.0, abs(ref)), name


def test_cyclegan_generator_sr_g9(golden):
    """G9: CycleGAN/CycleGAN_w_v1/CycleGAN.py GeneratorSR(num_residual=3) built under torch.manual_seed(9): the
    oracle draws the same weights, computes the same output and the same gradients."""
    g = golden("g9_cyclegan_sr.npz")
    sd = O.make_generator_sr_state(int(g["seed"]), num_residual=int(g["num_residual"]))
    for name, v in sd.items():
        ref = float(g[f"wsum/{name}"])
        assert abs(v.double().sum().item() - ref) <= 1e-9 * max(1.0, abs(ref)), name
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    out = O.generator_sr_forward(params, x)
    np.testing.assert_allclose(out.detach().numpy(), g["out"], rtol=0, atol=2e-6)
    (out * torch.from_numpy(g["r"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["grad_x"], rtol=1e-4, atol=1e-6)
    for name, p in params.items():
        ref_abs = float(g[f"gabs/{name}"])
        assert abs(p.grad.double().abs().sum().item() - ref_abs) <= 1e-4 * ref_abs + 1e-9, name
    np.testing.assert_allclose(params["conv_out.bias"].grad.numpy(), g["gfull/conv_out.bias"], rtol=1e-4, atol=1e-6)
    keys = [ln.split(" ", 1)[0] for ln in open(os.path.join(os.path.dirname(__file__), "golden",
                                                           "g9_cyclegan_sr_manifest.txt")).read().splitlines()]
    assert keys == list(O.make_generator_sr_state(0).keys())


def test_conditional_vae_g10(golden):
    """G10: "condtional VAE/model.py" CVAE(latent_dim=32) built under torch.manual_seed(10): same weights, same
    encoder outputs and same decoder output for a fixed latent, in eval and in training mode (batch statistics)."""
    g = golden("g10_cvae.npz")
    sd = O.make_cvae_state(int(g["seed"]), int(g["latent_dim"]))
    for name, v in sd.items():
        if v.dtype.is_floating_point and "running_" not in name:     # (the fixture's buffers are post-training-pass)
            ref = float(g[f"wsum/{name}"])
            assert abs(v.double().sum().item() - ref) <= 1e-9 * max(1.0, abs(ref)), name
    x, z = torch.from_numpy(g["x"]), torch.from_numpy(g["z"])
    for mode in ("eval", "train"):
        with torch.no_grad():
            mu, logvar, skips = O.cvae_encode(sd, x, mode == "train")
            out = O.cvae_decode(sd, z, x, skips, mode == "train")
        np.testing.assert_allclose(mu.numpy(), g[f"{mode}/mu"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(logvar.numpy(), g[f"{mode}/logvar"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(out.numpy(), g[f"{mode}/out"], rtol=0, atol=5e-5)
        for i, s_ in enumerate(skips):
            ref = float(g[f"{mode}/skip{i + 1}_abs"])
            assert abs(s_.double().abs().sum().item() - ref) <= 1e-5 * ref
    np.testing.assert_allclose(sd["enc_conv2.1.running_mean"].numpy(), g["running_mean_enc2"], rtol=1e-4, atol=1e-7)
    keys = [ln.split(" ", 1)[0] for ln in open(os.path.join(os.path.dirname(__file__), "golden",
                                                           "g10_cvae_manifest.txt")).read().splitlines()]
    assert keys == list(O.make_cvae_state(0).keys())


def test_generator_down_oracle_matches_reference_vector(golden):
    """G11: the oracle's restatement of CycleGAN.py:140-163 against the output of the reference class, its weights from
    the same seed (checked through the stored weight sums) and the key schema of the default network."""
    g = golden("g11_cyclegan_down.npz")
    sd = O.make_generator_down_state(int(g["seed"]), base_channels=int(g["base_channels"]))
    for k, v in sd.items():
        assert abs(v.double().sum().item() - float(g[f"wsum/{k}"])) < 1e-9 * max(1.0, abs(float(g[f"wsum/{k}"]))) + 1e-6, k
    out = O.generator_down_forward(sd, torch.from_numpy(g["x"]))
    assert (out - torch.from_numpy(g["out"])).abs().max().item() < 1e-6
    import os
    manifest = open(os.path.join(os.path.dirname(__file__), "golden", "g11_cyclegan_down_manifest.txt")).read().split("\n")
    keys = [ln.split(" ")[0] for ln in manifest if ln]
    assert keys == list(O.make_generator_down_state(0).keys())
