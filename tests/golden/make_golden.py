"""Generate tests/golden/*.npz by running the REFERENCE classes on seeded inputs.

Runs only in the build container (needs /root/reference); the outputs are committed.
The reference module imports torchvision at top level (esrgan_model.py:4) although the
generator and discriminator use only torch.nn, so an empty ``torchvision.models`` module
is placed in sys.modules before loading the file by path.  Nothing of the reference's
text is stored: fixtures hold seeds, inputs, outputs and gradients only.

    cd /tmp && python /root/repo/tests/golden/make_golden.py        # all fixtures
    cd /tmp && python /root/repo/tests/golden/make_golden.py g8     # only g8_adversarial_step.npz
    cd /tmp && python /root/repo/tests/golden/make_golden.py g9     # only g9_cyclegan_sr.npz (+ manifest)
    cd /tmp && python /root/repo/tests/golden/make_golden.py g10    # only g10_cvae.npz (+ manifest)
    cd /tmp && python /root/repo/tests/golden/make_golden.py g11    # only g11_cyclegan_down.npz (+ manifest)
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/esrgan/models/esrgan_model.py"


def load_reference():
    for m in ("torchvision", "torchvision.models"):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    spec = importlib.util.spec_from_file_location("ref_esrgan_model", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def summarize(t: torch.Tensor, stride: int = 97):
    """A gradient tensor as (strided sample, float64 sum, float64 abs-sum)."""
    flat = t.detach().reshape(-1)
    return (flat[::stride].numpy().copy(),
            np.float64(flat.double().sum().item()),
            np.float64(flat.double().abs().sum().item()))


def put_grads(store: dict, module: torch.nn.Module, full=()):
    for name, p in module.named_parameters():
        s, a, b = summarize(p.grad)
        store[f"gs/{name}"] = s
        store[f"gsum/{name}"] = a
        store[f"gabs/{name}"] = b
        if name in full:
            store[f"gfull/{name}"] = p.grad.detach().numpy().copy()


def weight_sums(store: dict, module: torch.nn.Module):
    for name, p in module.state_dict().items():
        if p.dtype.is_floating_point:
            store[f"wsum/{name}"] = np.float64(p.double().sum().item())


def make_g8(ref):
    # G8: adversarial iteration (esrgan/train.py:196-223) with the reference generator and discriminator.
    # The reference's VGGFeatureExtractor cannot be built here (torchvision absent, ImageNet weights need the
    # network): the perceptual term uses the oracle's vgg19 restatement with seeded stand-in weights.
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import esrgan_oracle as O
    torch.manual_seed(6)
    gen = ref.GeneratorRRDB(num_blocks=1)
    torch.manual_seed(8)
    disc = ref.Discriminator()
    disc.train()
    vsd = O.make_vgg_state(7)
    g = torch.Generator().manual_seed(106)
    lr = torch.rand(2, 3, 8, 8, generator=g)
    hr = torch.rand(2, 3, 32, 32, generator=g)
    optim_G = torch.optim.Adam(gen.parameters(), lr=1e-4, betas=(0.9, 0.999))
    optim_D = torch.optim.Adam(disc.parameters(), lr=1e-4, betas=(0.9, 0.999))
    mse, l1 = torch.nn.MSELoss(), torch.nn.L1Loss()
    optim_D.zero_grad()
    fake = gen(lr).detach()
    pred_real = disc(hr)
    pred_fake = disc(fake)
    loss_D = 0.5 * (mse(pred_real, torch.ones_like(pred_real)) + mse(pred_fake, torch.zeros_like(pred_fake)))
    loss_D.backward()
    st = {"seed_g": 6, "seed_d": 8, "seed_vgg": 7, "lr": lr.numpy(), "hr": hr.numpy(),
          "loss_D": np.float64(loss_D.item())}
    for name, p in disc.named_parameters():
        s_, a_, b_ = summarize(p.grad, stride=193)
        st[f"d_gs/{name}"], st[f"d_gsum/{name}"], st[f"d_gabs/{name}"] = s_, a_, b_
    optim_D.step()
    optim_G.zero_grad()
    fake = gen(lr)
    pred_fake = disc(fake)
    loss_G_gan = mse(pred_fake, torch.ones_like(pred_fake))
    loss_G_pix = l1(fake, hr)
    loss_G_per = l1(O.vgg_features(vsd, fake), O.vgg_features(vsd, hr))
    loss_G = loss_G_pix + 0.01 * loss_G_per + 1e-3 * loss_G_gan
    loss_G.backward()
    st.update(loss_G=np.float64(loss_G.item()), loss_G_gan=np.float64(loss_G_gan.item()),
              loss_G_pix=np.float64(loss_G_pix.item()), loss_G_per=np.float64(loss_G_per.item()))
    put_grads(st, gen, full=("conv_first.weight", "conv_last.bias"))
    optim_G.step()
    st["running_mean_3"] = disc.model[3].running_mean.numpy().copy()
    st["num_batches_tracked_3"] = np.int64(disc.model[3].num_batches_tracked.item())
    for name, p in list(gen.state_dict().items()) + [("D." + k, v) for k, v in disc.state_dict().items()]:
        if p.dtype.is_floating_point:
            st[f"wsum/{name}"] = np.float64(p.double().sum().item())
    np.savez_compressed(os.path.join(HERE, "g8_adversarial_step.npz"), **st)

def make_g9():
    # G9: the CycleGAN super-resolution generator (CycleGAN/CycleGAN_w_v1/CycleGAN.py:89-137).  The file imports
    # torchvision.transforms at top level (used by its dataset class only): an empty stand-in module lets it load;
    # its training entry point sits behind `if __name__ == "__main__"`.
    for m in ("torchvision", "torchvision.transforms"):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    spec = importlib.util.spec_from_file_location("ref_cyclegan_v1", "/root/reference/CycleGAN/CycleGAN_w_v1/CycleGAN.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    torch.manual_seed(9)
    net = mod.GeneratorSR(num_residual=3)
    g = torch.Generator().manual_seed(109)
    x = (torch.rand(2, 3, 12, 10, generator=g) * 2 - 1).requires_grad_(True)      # the dataset normalises to [-1, 1]
    r = torch.randn(2, 3, 48, 40, generator=g)
    out = net(x)
    (out * r).sum().backward()
    st = {"seed": 9, "num_residual": 3, "x": x.detach().numpy(), "r": r.numpy(), "out": out.detach().numpy(),
          "grad_x": x.grad.numpy()}
    put_grads(st, net, full=("conv1.0.bias", "res_blocks.2.conv2.bias", "conv_out.bias"))
    weight_sums(st, net)
    np.savez_compressed(os.path.join(HERE, "g9_cyclegan_sr.npz"), **st)
    with open(os.path.join(HERE, "g9_cyclegan_sr_manifest.txt"), "w") as f:
        for k, v in mod.GeneratorSR().state_dict().items():
            f.write(f"{k} {tuple(v.shape)}\n")


def make_g11():
    # G11: the CycleGAN HR -> LR generator (CycleGAN/CycleGAN_w_v1/CycleGAN.py:140-163), loaded like G9's class.
    for m in ("torchvision", "torchvision.transforms"):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    spec = importlib.util.spec_from_file_location("ref_cyclegan_v1", "/root/reference/CycleGAN/CycleGAN_w_v1/CycleGAN.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    torch.manual_seed(11)
    net = mod.GeneratorDown(base_channels=16)
    g = torch.Generator().manual_seed(111)
    x = (torch.rand(2, 3, 24, 20, generator=g) * 2 - 1).requires_grad_(True)
    r = torch.randn(2, 3, 6, 5, generator=g)
    out = net(x)
    (out * r).sum().backward()
    st = {"seed": 11, "base_channels": 16, "x": x.detach().numpy(), "r": r.numpy(), "out": out.detach().numpy(),
          "grad_x": x.grad.numpy()}
    put_grads(st, net, full=("down.0.bias", "res_blocks.1.conv2.bias", "res_blocks.0.conv1.weight", "out_conv.bias"))
    weight_sums(st, net)
    np.savez_compressed(os.path.join(HERE, "g11_cyclegan_down.npz"), **st)
    with open(os.path.join(HERE, "g11_cyclegan_down_manifest.txt"), "w") as f:
        for k, v in mod.GeneratorDown().state_dict().items():
            f.write(f"{k} {tuple(v.shape)}\n")


def make_g10():
    # G10: the conditional VAE ("condtional VAE/model.py":53-215).  The file imports torchvision.models at top level
    # (for its VGG loss class, which needs ImageNet weights and is not built): an empty stand-in module lets it load.
    # The weights are stored (the state is small only where it matters: the two 8192-wide linear layers are kept as
    # seeds of the reference constructor, so the fixture stores the state_dict's per-tensor sums and the test
    # re-creates the module state from the reference-ordered construction in the oracle's recipe below).
    for m in ("torchvision", "torchvision.models"):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    spec = importlib.util.spec_from_file_location("ref_cvae_model", "/root/reference/condtional VAE/model.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    torch.manual_seed(10)
    net = mod.CVAE(latent_dim=32)
    g = torch.Generator().manual_seed(110)
    x = torch.rand(3, 3, 32, 32, generator=g) * 2 - 1
    z = torch.randn(3, 32, generator=g)
    st = {"seed": 10, "latent_dim": 32, "x": x.numpy(), "z": z.numpy()}
    # the convolution / normalisation / linear parameters the HIP module needs (the fixture stays under 1 MB with
    # latent_dim 32; the two transposed convs dominate and are stored as float16-exact? no: stored as fp32 slices)
    for mode in ("eval", "train"):
        net.train(mode == "train")
        with torch.no_grad():
            mu, logvar, skips = net.encode(x)
            out = net.decode(z, x, skips)
        st[f"{mode}/mu"], st[f"{mode}/logvar"], st[f"{mode}/out"] = mu.numpy(), logvar.numpy(), out.numpy()
        for i, s_ in enumerate(skips):
            st[f"{mode}/skip{i + 1}_sum"] = np.float64(s_.double().sum().item())
            st[f"{mode}/skip{i + 1}_abs"] = np.float64(s_.double().abs().sum().item())
    st["running_mean_enc2"] = net.enc_conv2[1].running_mean.numpy().copy()
    weight_sums(st, net)
    np.savez_compressed(os.path.join(HERE, "g10_cvae.npz"), **st)
    with open(os.path.join(HERE, "g10_cvae_manifest.txt"), "w") as f:
        for k, v in mod.CVAE().state_dict().items():
            f.write(f"{k} {tuple(v.shape)}\n")


def main():
    if sys.argv[1:] == ["g10"]:
        torch.set_num_threads(8)
        make_g10()
        return
    if sys.argv[1:] == ["g9"]:
        torch.set_num_threads(8)
        make_g9()
        return
    if sys.argv[1:] == ["g11"]:
        torch.set_num_threads(8)
        make_g11()
        return
    ref = load_reference()
    torch.set_num_threads(8)
    if sys.argv[1:] == ["g8"]:          # only the newest fixture; the others are unchanged
        make_g8(ref)
        return

    # G1: one dense block ---------------------------------------------------------
    torch.manual_seed(0)
    rdb = ref.ResidualDenseBlock(64, 32)
    g = torch.Generator().manual_seed(100)
    x = torch.randn(1, 64, 8, 8, generator=g).requires_grad_(True)
    r = torch.randn(1, 64, 8, 8, generator=g)
    out = rdb(x)
    (out * r).sum().backward()
    st = {"seed": 0, "x": x.detach().numpy(), "r": r.numpy(), "out": out.detach().numpy(),
          "grad_x": x.grad.numpy()}
    put_grads(st, rdb, full=("conv1.bias", "conv5.bias", "conv1.weight"))
    weight_sums(st, rdb)
    np.savez_compressed(os.path.join(HERE, "g1_rdb.npz"), **st)

    # G2: one RRDB ----------------------------------------------------------------
    torch.manual_seed(1)
    rrdb = ref.RRDB(64, 32)
    g = torch.Generator().manual_seed(101)
    x = torch.randn(1, 64, 9, 7, generator=g).requires_grad_(True)
    r = torch.randn(1, 64, 9, 7, generator=g)
    out = rrdb(x)
    (out * r).sum().backward()
    st = {"seed": 1, "x": x.detach().numpy(), "r": r.numpy(), "out": out.detach().numpy(),
          "grad_x": x.grad.numpy()}
    put_grads(st, rrdb, full=("rdb3.conv5.bias", "rdb1.conv1.bias"))
    weight_sums(st, rrdb)
    np.savez_compressed(os.path.join(HERE, "g2_rrdb.npz"), **st)

    # G3: 2-block generator, non-square input -------------------------------------
    torch.manual_seed(2)
    gen = ref.GeneratorRRDB(num_blocks=2)
    with torch.no_grad():
        gen.conv_last.bias += 0.5           # recipe: re-centre outputs into [0,1]
    g = torch.Generator().manual_seed(102)
    x = torch.rand(1, 3, 12, 10, generator=g).requires_grad_(True)
    r = torch.randn(1, 3, 48, 40, generator=g)
    out = gen(x)
    (out * r).sum().backward()
    st = {"seed": 2, "bias_shift": 0.5, "x": x.detach().numpy(), "r": r.numpy(),
          "out": out.detach().numpy(), "grad_x": x.grad.numpy()}
    put_grads(st, gen, full=("conv_first.weight", "conv_first.bias", "conv_last.weight",
                             "conv_last.bias", "body.1.rdb3.conv5.bias", "trunk_conv.bias"))
    weight_sums(st, gen)
    np.savez_compressed(os.path.join(HERE, "g3_gen2.npz"), **st)

    # G4: full 23-block generator -------------------------------------------------
    torch.manual_seed(3)
    gen = ref.GeneratorRRDB()
    with torch.no_grad():
        gen.conv_last.bias += 0.5
    g = torch.Generator().manual_seed(103)
    x = torch.rand(1, 3, 16, 16, generator=g)
    out = gen(x)
    torch.nn.functional.l1_loss(out, torch.zeros_like(out)).backward()
    st = {"seed": 3, "bias_shift": 0.5, "x": x.numpy(), "out": out.detach().numpy(),
          "n_keys": len(gen.state_dict()),
          "n_params": sum(p.numel() for p in gen.parameters())}
    put_grads(st, gen, full=("conv_first.weight", "conv_last.bias"))
    st["gfull/body.22.rdb3.conv5.weight[:, :8]"] = \
        gen.body[22].rdb3.conv5.weight.grad[:, :8].numpy().copy()
    # keep the file small: drop the per-tensor samples of the 690 body tensors except block 0 and 22
    for k in list(st):
        if k.startswith("gs/body.") and not (k.startswith("gs/body.0.") or k.startswith("gs/body.22.")):
            del st[k]
    np.savez_compressed(os.path.join(HERE, "g4_gen23.npz"), **st)

    # G5: state-dict manifest -----------------------------------------------------
    with open(os.path.join(HERE, "g5_manifest.txt"), "w") as f:
        for k, v in gen.state_dict().items():
            f.write(f"{k} {tuple(v.shape)} {str(v.dtype).replace('torch.', '')}\n")

    # G6: discriminator, train and eval mode --------------------------------------
    torch.manual_seed(4)
    disc = ref.Discriminator()
    g = torch.Generator().manual_seed(104)
    img = torch.rand(2, 3, 128, 128, generator=g)
    disc.train()
    with torch.no_grad():
        out_train = disc(img)
    st = {"seed": 4, "img_seed": 104, "out_train": out_train.numpy(),
          "running_mean_2": disc.model[3].running_mean.numpy().copy(),
          "running_var_2": disc.model[3].running_var.numpy().copy()}
    disc.eval()
    with torch.no_grad():
        st["out_eval"] = disc(img).numpy()
    weight_sums(st, disc)
    with open(os.path.join(HERE, "g6_disc_manifest.txt"), "w") as f:
        for k, v in disc.state_dict().items():
            f.write(f"{k} {tuple(v.shape)} {str(v.dtype).replace('torch.', '')}\n")
    np.savez_compressed(os.path.join(HERE, "g6_disc.npz"), **st)

    # G7: pixel pre-training step (esrgan/train.py:156-161) ------------------------
    torch.manual_seed(5)
    gen = ref.GeneratorRRDB(num_blocks=2)
    g = torch.Generator().manual_seed(105)
    lr = torch.rand(2, 3, 8, 8, generator=g)
    hr = torch.rand(2, 3, 32, 32, generator=g)
    loss = torch.nn.L1Loss()(gen(lr), hr)
    loss.backward()
    st = {"seed": 5, "lr": lr.numpy(), "hr": hr.numpy(), "loss_pix": np.float64(loss.item())}
    put_grads(st, gen, full=("conv_first.weight", "conv_last.weight", "body.0.rdb1.conv1.bias"))
    np.savez_compressed(os.path.join(HERE, "g7_pixel_step.npz"), **st)

    make_g8(ref)

    for fn in sorted(os.listdir(HERE)):
        print(fn, os.path.getsize(os.path.join(HERE, fn)))


if __name__ == "__main__":
    main()
